for lib in simplenerf_amd/libsimplenerf_hip.so gpurun_abl_gnoload.so gpurun_abl_gnostore.so gpurun_abl_gneither.so; do echo $lib; SNERF_LIB=$lib python tools/probes/time_layered.py 2>/dev/null | grep "8x512" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['mlp'], round(d['forward_ms'],2), round(d['forward_keeping_ms'],2), round(d['backward_ms'],2), round(d['fraction_of_fp32_mfma_peak_forward'],3))
"; done
