mkdir -p gpurun_out
python -m pytest tests/test_gpu_optim.py tests/test_gpu_grads.py tests/test_gpu_bindings.py tests/test_gpu_f16.py tests/test_gpu_dataparallel.py tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/r05_tests4.log 2>&1; echo tests rc $?; tail -2 gpurun_out/r05_tests4.log
for rows in 4096 512; do for mode in "" "--single-pass"; do
tag=f16_${rows}_graphed$(echo $mode | tr -d ' -')
python bench.py --train --precision f16 --rows-per-gpu $rows --force-collective --graphed $mode --steps 20 --warmup 5 --extra-file gpurun_out/r05_share3_${tag}.json > gpurun_out/r05_share3_${tag}.line 2>&1; python - <<PY
import json
d=json.load(open('gpurun_out/r05_share3_${tag}.json'))
print('${tag}', 'ms/step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['timing']['step_ms']['p50'])
PY
done; done
