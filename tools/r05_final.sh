# Round-5 records on one lease: the driver's bench command, the GPU suite, the long secondary set, rocprof kernel stats and PMC
# passes of the headline and of the config-5 iteration.   bash tools/r05_final.sh [tag]
tag=${1:-r05}
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 --extra-file gpurun_out/${tag}_bench_final_extra.json > gpurun_out/${tag}_bench_final.json 2> gpurun_out/${tag}_bench_final.err; echo bench rc $? $(wc -c < gpurun_out/${tag}_bench_final.json) bytes $(wc -c < gpurun_out/${tag}_bench_final.err) stderr bytes
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputest.log 2>&1; echo tests rc $?; tail -2 gpurun_out/${tag}_gputest.log
python bench.py --extras --verbose --extra-file gpurun_out/${tag}_bench_extras.json > gpurun_out/${tag}_bench_extras.line 2> gpurun_out/${tag}_bench_extras.err; echo extras rc $?; tail -3 gpurun_out/${tag}_bench_extras.err
bash tools/profile_bench.sh ${tag} > gpurun_out/${tag}_profile_bench.log 2>&1; echo profile_bench rc $?
bash tools/profile_train.sh ${tag} f16 > gpurun_out/${tag}_profile_train.log 2>&1; echo profile_train rc $?
bash tools/pmc_passes.sh ${tag}_pmc_train_f16 bench.py --train --precision f16 --steps 8 --warmup 2 > gpurun_out/${tag}_pmc_train.log 2>&1; echo pmc_train rc $?
bash tools/pmc_passes.sh ${tag}_pmc_train_bf16s8 bench.py --train --precision bf16s8 --steps 8 --warmup 2 > gpurun_out/${tag}_pmc_train_s8.log 2>&1; echo pmc_train_s8 rc $?
