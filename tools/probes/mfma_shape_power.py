"""Driver of mfma_shape_power.hip: builds it on the GPU box, runs each MFMA shape for a few seconds while sampling the
board's power and shader clock (sysfs hwmon, as power_clock.py), prints one line per shape.
usage: mfma_shape_power.py [seconds, default 3]"""
import glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
seconds = sys.argv[1] if len(sys.argv) > 1 else '3'
exe = os.path.join(ROOT, 'gpurun_out', 'mfma_shape_power')
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-w', '-o', exe, os.path.join(ROOT, 'tools/probes/mfma_shape_power.hip')], check=True)


def read(p):
    try:
        return float(open(p).read())
    except (OSError, ValueError):
        return float('nan')


nodes = [hw for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*') if os.path.exists(os.path.join(hw, 'power1_input'))]
# (shape, waves per workgroup, workgroups per CU): 16x16x32 with 2 resp. 4 MFMAs per fragment at 4, 2 and 1 waves per SIMD
CASES = (('32', '8', '2'), ('16', '8', '2'), ('64', '8', '2'), ('16', '8', '1'), ('64', '8', '1'), ('16', '4', '1'), ('64', '4', '1'), ('16', '8', '2'), ('64', '8', '2'))
for shape, waves, wgs in CASES:
    rows, on = [], [True]

    def loop():
        while on[0]:
            rows.append([(read(os.path.join(hw, 'power1_input')), read(os.path.join(hw, 'freq1_input'))) for hw in nodes])
            time.sleep(0.02)
    th = threading.Thread(target=loop, daemon=True); th.start()
    out = subprocess.run([exe, shape, seconds, waves, wgs], capture_output=True, text=True)
    on[0] = False; th.join()
    rows = rows[len(rows) // 3:]
    # our GPU = the node with the highest mean power (the box shows all eight)
    best = max(range(len(nodes)), key=lambda i: sum(r[i][0] for r in rows)) if nodes and rows else None
    extra = ''
    if best is not None:
        extra = '  power %.0f W  sclk %.0f MHz' % (sum(r[best][0] for r in rows) / len(rows) / 1e6, sum(r[best][1] for r in rows) / len(rows) / 1e6)
    print(out.stdout.strip() + extra, flush=True)
    if out.returncode != 0:
        print(out.stderr[-500:]); sys.exit(1)
