# 24 captures of the whole iteration right after eager all-reduces (a one-rank RCCL group): the NCCL watchdog must not take the
# process down (harness.CAPTURE_MODE).  Prints the exit codes.
fails=0
for i in $(seq 1 24); do
  python bench.py --train --precision f16 --rows-per-gpu 512 --force-collective --graphed --steps 4 --warmup 1 --settle-seconds 0.05 --extra-file gpurun_out/stress.json > /dev/null 2> gpurun_out/stress.err || { fails=$((fails+1)); tail -3 gpurun_out/stress.err; }
done
echo "capture stress: $fails of 24 runs failed"
