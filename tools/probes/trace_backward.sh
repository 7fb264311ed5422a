#!/bin/bash
# Kernel-by-kernel device time of snerf_mlp_backward (main 8x256 MLP) at the two config-5 sizes, from a rocprofv3 kernel trace:
#   bash tools/probes/trace_backward.sh <precision 0|1|2> [lib.so]         -> gpurun_out/trace_backward/
prec=${1:-2}; lib=${2:-simplenerf_amd/libsimplenerf_hip.so}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/trace_backward/p$prec
rm -rf $out; mkdir -p $out
(cd $root && SNERF_CHILD=1 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 tools/probes/time_backward_sizes.py $prec $lib > $out.log 2>&1) || { tail -5 $out.log; exit 1; }
python3 - <<P
import csv,glob,collections
f=glob.glob('$out/**/t_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
# one backward call = the kernels between two clear_words_kernel launches; keep the last 40 calls (20 per size, timed rounds)
calls=[];cur=None
for r in rows:
    name=r['Kernel_Name'].replace('(anonymous namespace)::','')
    if name.startswith('clear_words_kernel') and int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',0))>=0:
        if cur: calls.append(cur)
        cur=[]
    if cur is not None: cur.append((name.split('(')[0][:44],int(r['Start_Timestamp']),int(r['End_Timestamp'])))
for label,sel in (('second size (last 20 calls)',calls[-20:]),('first size (20 calls before the last 100)',calls[-120:-100])):
    agg=collections.OrderedDict()
    span=0
    for c in sel:
        span+=c[-1][2]-c[0][1]
        for name,s,e in c:
            agg.setdefault(name,[0,0]); agg[name][0]+=e-s; agg[name][1]+=1
    print(label,'call span %.1f us'%(span/len(sel)/1e3))
    for name,(t,n) in agg.items(): print('   %-46s x%-3.1f %8.1f us per call'%(name,n/len(sel),t/len(sel)/1e3))
P
