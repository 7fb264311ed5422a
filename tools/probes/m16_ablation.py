"""What separates the 16-bit inference kernel (csrc/mlp_forward_m16.hip) from its skeleton (tools/probes/m64_skeleton.hip: 1.65
PFLOP/s)?  Times one fine-size launch (262 144 samples) of the shipped library and of ablation variants of that ONE translation
unit (tools/probes/build_variant.py <name> --only mlp_forward_m16 <switches>; wrong results by design, timing only):
    m16_nodma            -DSNERF_ABL_NODMA              no LDS-DMA instruction issued (the ring is never filled)
    m16_nobarrier        -DSNERF_ABL_NOBARRIER          no workgroup barrier at the unit hand-over
    m16_nodma_nobarrier  both
    m16_noencode         -DSNERF_PROBE_M16_NOENCODE     no positional encoding (lane-dependent constants instead)
    m16_bare             all three
    m16_novmwait         -DSNERF_PROBE_NO_VMWAIT        DMA issued, never waited for
(variants that are not built are skipped)
Each in a fresh process, alternated over two rounds.   usage: python tools/probes/m16_ablation.py [out.txt]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r05_m16_ablation.txt')
libs = [os.path.join(ROOT, 'simplenerf_amd', 'libsimplenerf_hip.so')] + [
    os.path.join(ROOT, f'gpurun_abl_m16_{name}.so') for name in ('novmwait', 'nodma', 'nobarrier', 'nodma_nobarrier', 'noencode', 'bare')]
lines = []
for rnd in range(2):
    for precision, tag in ((3, 'bf16'), (2, 'f16'), (1, 'f16x3')):     # (bf16: the single-product kernel without a range watch)
        for lib in libs:
            if not os.path.exists(lib):
                continue
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'probes', 'time_mlp.py'), lib, str(precision)],
                               capture_output=True, text=True, timeout=300)
            text = (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]
            lines.append(f'round {rnd} {tag}: ' + text.replace(ROOT + os.sep, ''))
            print(lines[-1], flush=True)
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, 'w') as f:
    f.write('\n'.join(lines) + '\n')
