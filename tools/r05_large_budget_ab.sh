#!/bin/bash
# one rank's share (512 / 1024 rows, one pass, graphed; and 4096 rows sub-batched) with the large weight-gradient class's workgroup
# budget scaled by the block count: SNERF_LARGE_MIN_BLOCKS 0 (fixed 256, as before) / 8 / 16 (shipped) / 32
mkdir -p gpurun_out; out=gpurun_out/r05_large_budget_ab.jsonl; : > $out
for round in 0 1; do for cfg in "512 1" "512 0" "1024 1" "4096 0"; do set -- $cfg
for lib in gpurun_abl_lmb0.so gpurun_abl_lmb8.so simplenerf_amd/libsimplenerf_hip.so gpurun_abl_lmb32.so; do
  timeout -k 10 200 python tools/probes/share_ab.py $lib $1 0 $2 1 f16 20 2>/dev/null | tail -1 >> $out || exit 1
done; done; done
cat $out
