root=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pv_$3; rm -rf $out; mkdir -p $out
(cd $root && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/probes/train_ab.py $1 $2 10 > $out/line.json 2> $out/err.txt) || { tail -5 $out/err.txt; exit 1; }
find $out -name '*kernel_trace*' -delete; find $out -name '*.db' -delete
python3 - $out/t_kernel_stats.csv "$4" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Name']: print(r['Name'].split('wgrad16_kernel')[1][:36], r['Calls'], round(float(r['AverageNs'])/1e3,1), end=' | ')
print()
PY
