# Fold the outputs of tools/r05_final.sh <tag> (merged into gpurun_out/ by gpurun) into profiles/.   bash tools/r05_collect.sh <tag> <commit the lease ran>
tag=${1:-r05c}; commit=${2:-$(git rev-parse --short HEAD)}
g=gpurun_out
python tools/collect_pmc.py $g profiles/pmc_traffic.json $commit > /dev/null && cp profiles/pmc_traffic.json profiles/r05_pmc_fp32_forward.json
cp $g/${tag}_stats/b_kernel_stats.csv profiles/r05_bench_kernel_stats.csv
cp $g/${tag}_stats/bench_line.json profiles/r05_bench_under_rocprof_line.json
cp $g/${tag}_train_f16_stats/t_kernel_stats.csv profiles/r05_train_f16_kernel_stats.csv
python tools/collect_pmc_kernels.py $g/${tag}_pmc_train_f16 profiles/pmc_traffic_train_f16.json "bench.py --train --precision f16 --steps 8 --warmup 2" --iterations auto > /dev/null
python tools/collect_pmc_kernels.py $g/${tag}_pmc_train_bf16s8 profiles/pmc_traffic_train_bf16s8.json "bench.py --train --precision bf16s8 --steps 8 --warmup 2" --iterations auto > /dev/null
for prec in f16 f16x3; do
  [ -d $g/${tag}_pmc_lds_$prec ] && python tools/collect_pmc_generic.py $g/${tag}_pmc_lds_$prec profiles/r05_pmc_lds_${prec}_forward_m16.json "bench.py --precision $prec --no-cpu-baseline --no-alt --steps 10 --warmup 2" mlp_forward_m16 | head -3
done
python - "$commit" <<'PY'
import json, sys
for n in ('pmc_traffic_train_f16', 'pmc_traffic_train_bf16s8', 'r05_pmc_lds_f16_forward_m16', 'r05_pmc_lds_f16x3_forward_m16'):
    p = f'profiles/{n}.json'
    try:
        d = json.load(open(p))
    except FileNotFoundError:
        continue
    d['commit'] = sys.argv[1]
    json.dump(d, open(p, 'w'), indent=1)
    print(n, d.get('iterations'), d.get('hbm_gb_per_iteration'))
PY
cp profiles/pmc_traffic_train_f16.json profiles/r05_pmc_train_f16.json; cp profiles/pmc_traffic_train_bf16s8.json profiles/r05_pmc_train_bf16s8.json
cp $g/${tag}_bench_final.json profiles/r05_bench_final.json; cp $g/${tag}_bench_final_extra.json profiles/r05_bench_final_extra.json
cp $g/${tag}_bench_extras.json profiles/r05_bench_extras.json; cp $g/${tag}_gputest.log profiles/r05_gputest_observed_parity.log
echo collected $tag at $commit
