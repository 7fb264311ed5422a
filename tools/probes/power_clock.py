"""Is the fp16 path power-limited?  Samples the GPU's power sensor and shader clock (sysfs hwmon, every 20 ms) while the
fused PE+MLP forward of the main 8x256 MLP runs back to back for a few seconds in each arithmetic mode, and prints the
averages next to the power cap.  usage: power_clock.py [seconds per mode, default 3]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simplenerf_amd import ops, synth
from simplenerf_amd.synth import abi_param_list
from tests import util


def sensors():
    out = {}
    for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        for name in ('power1_average', 'power1_input', 'power1_cap', 'freq1_input', 'freq2_input', 'temp1_input', 'temp2_input'):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(hw, {})[name] = p
    return out


def read(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return float('nan')


class Sampler(threading.Thread):
    def __init__(self, paths):
        super().__init__(daemon=True)
        self.paths, self.rows, self.on = paths, [], True

    def run(self):
        while self.on:
            self.rows.append({k: read(p) for k, p in self.paths.items()})
            time.sleep(0.02)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    found = sensors()
    print('hwmon nodes:', len(found))
    if not found:
        print('no readable sensors')
        return
    # sysfs shows every GPU of the host, HIP only ours: find it by the PCI address torch reports, else sample them all
    # and keep the one whose power moves
    props = torch.cuda.get_device_properties(0)
    mine = None
    if hasattr(props, 'pci_bus_id'):
        want = '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), props.pci_bus_id, getattr(props, 'pci_device_id', 0))
        for hw in found:
            if want in os.path.realpath(os.path.join(hw, '..', '..')):
                mine = hw
    print('device', torch.cuda.get_device_name(0), 'hwmon', mine)
    if mine is None:
        cand = {}
        probe = torch.rand(8192, 8192, device='cuda')
        for hw, pp in found.items():
            cand[hw] = read(pp.get('power1_average', pp.get('power1_input')))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.5:
            probe @ probe
        torch.cuda.synchronize()
        mine = max(found, key=lambda hw: read(found[hw].get('power1_average', found[hw].get('power1_input'))) - cand[hw])
        print('picked by power rise:', mine)
    paths = found[mine]
    cap = read(paths['power1_cap']) / 1e6 if 'power1_cap' in paths else float('nan')
    cfg = synth.mlp_config(128)
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 1)
    mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()}))
    n, s = 4096, 256
    o = torch.rand(n, 3, device='cuda'); d = torch.rand(n, 3, device='cuda'); v = d / d.norm(dim=1, keepdim=True)
    z = torch.sort(torch.rand(n, s, device='cuda'), 1)[0]
    flop = n * s * 2 * 593408

    def summarise(tag, rows, rate=None):
        if not rows:
            print(tag, 'no samples'); return
        rows = rows[len(rows) // 4:]      # steady state: drop the first quarter
        mean = lambda k: sum(r[k] for r in rows if r.get(k) == r.get(k)) / max(1, sum(1 for r in rows if r.get(k) == r.get(k)))
        pk = 'power1_average' if 'power1_average' in paths else 'power1_input'
        line = f'{tag:16s} power {mean(pk) / 1e6:7.1f} W (cap {cap:.0f} W)'
        for fk, label in (('freq1_input', 'sclk'), ('freq2_input', 'mclk')):
            if fk in paths:
                line += f'  {label} {mean(fk) / 1e6:6.0f} MHz'
        for tk in ('temp1_input', 'temp2_input'):
            if tk in paths:
                line += f'  {tk[:5]} {mean(tk) / 1e3:4.0f} C'
        if rate:
            line += f'  {rate:7.1f} TFLOP/s algorithmic'
        print(line, f'({len(rows)} samples)', flush=True)

    sm = Sampler(paths); sm.start(); time.sleep(1.0); sm.on = False; sm.join()
    summarise('idle', sm.rows)
    for tag, prec in (('fp32', 0), ('f16x3', 1), ('f16', 2), ('fp32', 0)):
        for _ in range(3): mlp.forward(o, d, v, z, precision=prec)
        torch.cuda.synchronize()
        sm = Sampler(paths); sm.start()
        t0 = time.perf_counter(); it = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20): mlp.forward(o, d, v, z, precision=prec)
            torch.cuda.synchronize(); it += 20
        dt = time.perf_counter() - t0
        sm.on = False; sm.join()
        summarise(tag, sm.rows, flop * it / dt / 1e12)
    # the training kernels of the 16-bit mode and of f16x3: storing forward, then backward (chain + weight gradients)
    plist = abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()})
    shapes = [tuple(p.shape) for p in plist]
    n2, s2 = 2048, 192
    o2, d2, v2, z2 = o[:n2], d[:n2], v[:n2], z[:n2, :s2].contiguous()
    gs, gr = torch.randn(n2, s2, 1, device='cuda') * 1e-4, torch.randn(n2, s2, 3, device='cuda') * 1e-4
    for tag, prec in (('f16', 2), ('f16x3', 1), ('fp32', 0)):
        sigma, rgb, saved = mlp.forward_train(o2, d2, v2, z2, None, prec)
        for what in ('fwd-train', 'backward'):
            run = (lambda: mlp.forward_train(o2, d2, v2, z2, None, prec)) if what == 'fwd-train' else \
                  (lambda: mlp.backward(saved, sigma, rgb, gs, gr, shapes, prec))
            for _ in range(2): run()
            torch.cuda.synchronize()
            sm = Sampler(paths); sm.start()
            t0 = time.perf_counter(); it = 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(10): run()
                torch.cuda.synchronize(); it += 10
            dt = time.perf_counter() - t0
            sm.on = False; sm.join()
            mult = 1 if what == 'fwd-train' else 2
            summarise(f'{tag} {what}', sm.rows, n2 * s2 * 2 * 593408 * mult * it / dt / 1e12)


if __name__ == '__main__':
    main()
