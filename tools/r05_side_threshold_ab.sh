#!/bin/bash
# levels side by side up to 65 536 (shipped) / 131 072 / 262 144 coarse samples per call: the 4096-row iteration sub-batched (131 072 per
# call) and as one pass (262 144), 2048 rows as one pass (131 072), f16 and bf16s8, graphed
mkdir -p gpurun_out; out=gpurun_out/r05_side_threshold_ab.jsonl; : > $out
for round in 0 1; do for cfg in "4096 0 f16" "4096 1 f16" "2048 1 f16" "4096 0 bf16s8"; do set -- $cfg
for lib in simplenerf_amd/libsimplenerf_hip.so gpurun_abl_side128k.so gpurun_abl_side256k.so; do
  timeout -k 10 200 python tools/probes/share_ab.py $lib $1 0 $2 1 $3 20 2>/dev/null | tail -1 >> $out || exit 1
done; done; done
cat $out
