#!/bin/bash
# rocprofv3 --pmc passes (one counter set each) of tools/probes/time_backward_sizes.py, folded into one line per backward
# kernel and pass: averages over dispatches, wave-state fractions relative to SQ_WAVE_CYCLES.  On the GPU box:
#   bash tools/probes/pmc_backward.sh <precision 0|1|2> [lib.so]        -> gpurun_out/pmc_backward/
prec=${1:-2}; lib=${2:-simplenerf_amd/libsimplenerf_hip.so}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1)); out=$root/gpurun_out/pmc_backward/pass$i
    rm -rf $out; mkdir -p $out
    (cd $root && SNERF_CHILD=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 tools/probes/time_backward_sizes.py $prec $lib > $out.log 2>&1) || { echo "pass $i failed"; tail -3 $out.log; continue; }
    find $out -name '*.db' -delete
    python3 - <<P
import csv,glob,collections
f=glob.glob('$out/**/p_counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    if 'chain' not in k and 'wgrad' not in k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k][r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,c in agg.items():
    m={a:sum(b)/len(b) for a,b in c.items()}
    us=sum(dur[k].values())/len(dur[k])/1e3
    w=m.get('SQ_WAVE_CYCLES',1)
    print('pass$i',k[:52],'%.0f us'%us,' '.join('%s %.3g (%.3f of wave cycles)'%(a.replace('SQ_',''),v,v/w) for a,v in sorted(m.items()) if a!='SQ_WAVE_CYCLES'),'WAVE_CYCLES %.3g'%w)
P
done
