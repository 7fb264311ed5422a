#!/bin/bash
# Round profiles of the headline bench on the GPU box: (1) rocprofv3 --kernel-trace --stats of the SAME command the driver
# runs (its per-kernel average must agree with the line's roofline.avg_launch_ms), (2) three separate --pmc passes
# (FETCH_SIZE | WRITE_SIZE | SQ set) of the headline-only run for roofline.traffic and MFMA-busy.
#   bash tools/profile_bench.sh <round tag, e.g. r03>      -> gpurun_out/<tag>_stats/, gpurun_out/pmc_*/
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/${tag}_stats
rm -rf $out; mkdir -p $out
(cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err) || { tail -5 $out/bench.err; exit 1; }
find $out -name '*kernel_trace*' -delete; find $out -name '*.db' -delete
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    o=$root/gpurun_out/pmc_${set%% *}
    rm -rf $o
    (cd $root && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $o -o p -- python3 bench.py --no-cpu-baseline --no-alt --steps 10 --warmup 2 > $o.log 2>&1) || { echo "pass '$set' failed"; tail -5 $o.log; exit 1; }
    find $o -name '*kernel_trace*' -delete; find $o -name '*.db' -delete
done
echo done
