# rocprofv3 kernel stats of one rank's share (512 rows) of the strong-scaled config-5 iteration, and of the full 4096-row one
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for rows in 512 4096; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_share_$rows -o share -- python3 $GRAFT_REPO_ROOT/bench.py --train --precision f16 --rows-per-gpu $rows --steps 20 --warmup 5 --extra-file $GRAFT_REPO_ROOT/gpurun_out/prof_share_$rows.json > $GRAFT_REPO_ROOT/gpurun_out/prof_share_$rows.line 2>&1
done
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_share_512 gpurun_out/prof_share_4096 -name "*kernel_stats.csv" | head; find gpurun_out/prof_share_512 -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out/prof_share_4096 -name "*kernel_trace.csv" -size +30M -delete
