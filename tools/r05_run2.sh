mkdir -p gpurun_out
python -m pytest tests/test_gpu_dataparallel.py tests/test_gpu_f16.py tests/test_gpu_generic.py tests/test_gpu_train_quality.py tests/test_gpu_dist.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r05_tests2.log 2>&1; echo tests rc $?; tail -25 gpurun_out/r05_tests2.log
for rows in 4096 512; do for p in f16 bf16s8; do
python bench.py --train --precision $p --rows-per-gpu $rows --force-collective --steps 20 --warmup 5 --extra-file gpurun_out/r05_share_${p}_${rows}.json > gpurun_out/r05_share_${p}_${rows}.line 2>&1; tail -c 400 gpurun_out/r05_share_${p}_${rows}.line; echo
python bench.py --train --precision $p --rows-per-gpu $rows --force-collective --graphed --steps 20 --warmup 5 --extra-file gpurun_out/r05_share_${p}_${rows}_graphed.json > gpurun_out/r05_share_${p}_${rows}_graphed.line 2>&1; tail -c 400 gpurun_out/r05_share_${p}_${rows}_graphed.line; echo
done; done
