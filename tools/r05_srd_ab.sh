mkdir -p gpurun_out; out=gpurun_out/r05_srd_ab2.txt; : > $out
for round in 0 1; do for prec in 3 2 1; do for lib in gpurun_abl_before_srd.so gpurun_abl_srd_builtin.so simplenerf_amd/libsimplenerf_hip.so; do
python tools/probes/time_mlp.py $lib $prec 2>/dev/null | tail -1 | sed "s#$(pwd)/##" >> $out
done; done; done
cat $out
python -m pytest tests/test_gpu_f16.py tests/test_gpu_bf16.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -2
