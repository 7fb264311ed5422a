# rocprofv3 kernel trace of one rank's share (512 rows) of the strong-scaled config-5 iteration: one model pass, whole iteration
# replayed from one HIP graph (python bench.py --train --precision f16 --rows-per-gpu 512 --single-pass --graphed)
mkdir -p gpurun_out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_share_512sp -o share -- python3 $root/bench.py --train --precision f16 --rows-per-gpu 512 --single-pass --graphed --steps 20 --warmup 5 --extra-file $root/gpurun_out/prof_share_512sp.json > $root/gpurun_out/prof_share_512sp.line 2>&1
cd $root; ls gpurun_out/prof_share_512sp/
