#!/bin/bash
# Per-kernel register / spill / instruction statistics of one library source, from the compiler's own assembly
# (same flags as simplenerf_amd/build.py).   tools/isa_stats.sh mlp_forward_m16 [extra -D flags]
src=$1; shift
out=${ISA_OUT:-/tmp/isa}
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-function -Wno-inline-asm -I$(dirname $0)/../include "$@" --cuda-device-only -S \
  -o $out/$src.s $(dirname $0)/../simplenerf_amd/csrc/$src.hip 2>/dev/null
python3 - "$out/$src.s" <<'PY'
import re, sys, subprocess
text = open(sys.argv[1]).read()
# kernel bodies: from "name:" to "s_endpgm"
meta = {}
blocks = text[text.index('amdhsa.kernels'):].split('\n  - .agpr_count')
for b in blocks[1:]:
    b = '.agpr_count' + b
    name = re.search(r'\n    \.name:\s+(\S+)', b).group(1)
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, b) or [0, '0'])[1]
    meta[name] = dict(vgpr=g('vgpr_count'), agpr=g('agpr_count'), sgpr=g('sgpr_count'), spill=g('vgpr_spill_count'), sspill=g('sgpr_spill_count'), scratch=g('private_segment_fixed_size'), lds=g('group_segment_fixed_size'))
for name, d in meta.items():
    start = text.find('\n' + name + ':')
    end = text.find('s_endpgm', start)
    body = text[start:end]
    count = lambda pat: len(re.findall(pat, body))
    try:
        pretty = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pretty = name
    pretty = re.sub(r'^void \(anonymous namespace\)::', '', pretty)
    ins = len([l for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')])
    print(f"{pretty[:90]:90s} vgpr {d['vgpr']:>3} agpr {d['agpr']:>3} spill {d['spill']:>3} scratch {d['scratch']:>5} | instr {ins:6d} mfma {count(r'v_mfma'):5d} "
          f"valu {count(chr(10)+chr(9)+'v_')-count(r'v_mfma'):5d} salu {count(chr(10)+chr(9)+'s_'):5d} ds {count(chr(10)+chr(9)+'ds_'):4d} pkmax3 {count('v_pk_maximum3'):3d} scratch_ops {count('scratch_'):3d}")
PY
