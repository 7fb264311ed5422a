#!/bin/bash
# Kernel timeline of the last headline steps (gaps and durations), with or without the library's timing events:
#   bash tools/probes/trace_step.sh <precision> bench|plain        (run on the GPU box; output gpurun_out/trace_step/)
prec=${1:-f16}; mode=${2:-bench}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/trace_step/${prec}_$mode
rm -rf $out; mkdir -p $out
if [ "$mode" == "bench" ]; then cmd="bench.py --precision $prec --no-alt --no-cpu-baseline --steps 6 --warmup 3"; else cmd="tools/probes/trace_step.py $prec"; fi
(cd $root && rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $cmd > $out.log 2>&1) || { tail -5 $out.log; exit 1; }
python3 - <<P
import csv,glob
f=glob.glob('$out/**/t_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))[-16:]
prev=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%8.1f us gap %6.1f  dur %7.1f  %s'%((s-int(rows[0]['Start_Timestamp']))/1e3,(s-prev)/1e3 if prev else 0,(e-s)/1e3,r['Kernel_Name'].replace('(anonymous namespace)::','')[:60]))
    prev=e
P
