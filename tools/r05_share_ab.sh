mkdir -p gpurun_out; out=gpurun_out/r05_share_ab.jsonl; : > $out
for round in 0 1; do for rows in 512 4096; do for sp in 0 1; do for lib in simplenerf_amd/libsimplenerf_hip.so gpurun_abl_noside.so; do for pp in 1 0; do
python tools/probes/share_ab.py $lib $rows $pp $sp 1 f16 20 2>/dev/null | tail -1 >> $out
done; done; done; done; done
cat $out
