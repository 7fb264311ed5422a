#!/bin/bash
# rocprofv3 --kernel-trace --stats of the config-5 training iteration on the GPU box:
#   bash tools/profile_train.sh <round tag> <precision>     -> gpurun_out/<tag>_train_<precision>_stats/
tag=${1:-r04}; prec=${2:-f16}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/${tag}_train_${prec}_stats
rm -rf $out; mkdir -p $out
(cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --train --precision $prec --steps 10 --warmup 3 > $out/bench_line.json 2> $out/bench.err) || { tail -5 $out/bench.err; exit 1; }
find $out -name '*kernel_trace*' -delete; find $out -name '*.db' -delete
echo done
