#!/bin/bash
# Three separate rocprofv3 --pmc passes of one python command (SQ set | FETCH_SIZE | WRITE_SIZE), as
# tools/collect_pmc_kernels.py expects them.  Run on the GPU box:   tools/pmc_passes.sh <tag> <script.py> [args...]
# Output: gpurun_out/<tag>_{SQ_VALU_MFMA_BUSY_CYCLES,FETCH_SIZE,WRITE_SIZE}/p_counter_collection.csv
# (--pmc is never combined with the trace domains gpurun refuses; the program itself follows `--`, not a wrapper.)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    out=$root/gpurun_out/${tag}_${set%% *}
    rm -rf $out
    (cd $root && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 "$@" > $out.log 2>&1) || { echo "pass '$set' failed"; tail -5 $out.log; exit 1; }
    # keep only the counter table (the traces are large)
    find $out -name '*kernel_trace*' -delete; find $out -name '*.db' -delete
    ls $out
done
