#!/bin/bash
# after the packed-multiply fix: the stand-alone reproducer, the repetition probe on the shipped library, the GPU suite
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_opsel_hazard tools/probes/pk_opsel_hazard.hip 2>/dev/null || exit 1
timeout -k 10 300 /tmp/pk_opsel_hazard > gpurun_out/pk_opsel_hazard.txt 2>&1 || exit 1
grep -c "mismatches" gpurun_out/pk_opsel_hazard.txt
for p in bf16 f16s8 f16x3; do
  timeout -k 10 300 python tools/probes/side_by_side_determinism.py simplenerf_amd/libsimplenerf_hip.so $p config3f ctypes 150 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/determinism_after.txt || exit 1
done
cat gpurun_out/determinism_after.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_after.log 2>&1
echo "pytest rc $?"
tail -3 gpurun_out/gputest_after.log
