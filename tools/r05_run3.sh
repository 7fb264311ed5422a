mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_tests3.log 2>&1; echo tests rc $?; tail -3 gpurun_out/r05_tests3.log
for rows in 4096 512; do for p in f16; do for mode in "" "--single-pass"; do
tag=${p}_${rows}_graphed$(echo $mode | tr -d ' -')
python bench.py --train --precision $p --rows-per-gpu $rows --force-collective --graphed $mode --steps 20 --warmup 5 --extra-file gpurun_out/r05_share2_${tag}.json > gpurun_out/r05_share2_${tag}.line 2>&1; python - <<PY
import json
d=json.load(open('gpurun_out/r05_share2_${tag}.json'))
print('${tag}', 'ms/step %.3f' % d['ms_per_step'], 'p50 %.3f' % d['timing']['step_ms']['p50'])
PY
done; done; done
python bench.py --train --precision f16 --rows-per-gpu 512 --force-collective --steps 20 --warmup 5 --extra-file gpurun_out/r05_share2_f16_512_eager.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/r05_share2_f16_512_eager.json')); print('f16_512_eager ms/step %.3f' % d['ms_per_step'])"
