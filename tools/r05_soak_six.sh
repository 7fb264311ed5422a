#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r05_soak_six_levels.jsonl; : > $out
for prec in bf16 f16s8 f16x3; do for lib in simplenerf_amd/libsimplenerf_hip.so gpurun_abl_noside.so; do
  timeout -k 10 280 python tools/probes/soak_six_levels.py $lib 600 $prec 2>/dev/null | tail -1 >> $out || exit 1
done; done
cat $out
