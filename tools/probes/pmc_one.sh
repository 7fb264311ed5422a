#!/bin/bash
# One rocprofv3 --pmc pass (SQ set) of tools/probes/time_mlp.py for a given library build, folded into one line per MLP
# forward kernel: duration, cycles, effective clock, MFMA-busy fraction, wave-state fractions.  Run on the GPU box:
#   bash tools/probes/pmc_one.sh <tag> <lib.so> [precision, default 2]     -> gpurun_out/pmc_one/<tag>/
# (--pmc is never combined with the trace domains gpurun refuses; the program itself follows `--`.)
tag=$1; lib=$2; prec=${3:-2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_one/$tag
rm -rf $out; mkdir -p $out
(cd $root && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out -o p -- python3 tools/probes/time_mlp.py $lib $prec > $out.log 2>&1) || { echo "pass failed"; tail -5 $out.log; exit 1; }
find $out -name '*.db' -delete
python3 - <<P
import csv,glob,collections
f=glob.glob('$out/**/p_counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'mlp_forward' not in k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k][r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,c in agg.items():
    m={a:sum(b)/len(b) for a,b in c.items()}
    us=sum(dur[k].values())/len(dur[k])/1e3
    cyc=m['GRBM_GUI_ACTIVE']/8
    w=m['SQ_WAVE_CYCLES']
    print('$tag',k.split('<')[0][-28:], 'dispatches',len(dur[k]),'duration %.1f us'%us,'cycles %.0f'%cyc,'clock %.2f GHz'%(cyc/us/1e3),
          'mfma busy %.3f'%(m['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024)),'waiting %.3f'%(m['SQ_WAIT_ANY']/w),'issue-stalled %.3f'%(m['SQ_WAIT_INST_ANY']/w),'issuing %.3f'%(m['SQ_ACTIVE_INST_ANY']/w))
P
grep ms $out.log
