#!/bin/bash
# the 16x16x32 rendering kernel with (shipped) and without (gpurun_abl_m16unpaired.so) the opaque pair in its encoding; precisions f16 (2), bf16 (3), f16x3 (1)
mkdir -p gpurun_out; : > gpurun_out/m16_pair_ab.txt
for round in 0 1 2; do for prec in 2 3 1; do for lib in gpurun_abl_m16unpaired.so simplenerf_amd/libsimplenerf_hip.so; do
  timeout -k 10 120 python tools/probes/time_mlp.py $lib $prec 2>/dev/null | tail -1 >> gpurun_out/m16_pair_ab.txt || exit 1
done; done; done
cat gpurun_out/m16_pair_ab.txt
