#!/bin/bash
# bisect the side-by-side backward: SNERF_SIDE_MODE 0 (as shipped) 1 2 3 4 (see render.hip, SNERF_PROBE_SIDE_MODES)
mkdir -p gpurun_out
: > gpurun_out/side_modes.txt
for mode in 0 1 2 3 4; do
  echo "== mode $mode" >> gpurun_out/side_modes.txt
  SNERF_SIDE_MODE=$mode timeout -k 10 200 python tools/probes/side_by_side_determinism.py gpurun_abl_modes.so bf16 config3f ctypes 120 2>&1 | grep -v "^rep" >> gpurun_out/side_modes.txt || exit 1
done
cat gpurun_out/side_modes.txt | cut -c1-150 | grep -v "pts_linears\|views_linears\|feature_linear\|output_linear"
