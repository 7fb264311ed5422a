"""Do an HBM-bound kernel sequence and an issue-bound one overlap when they run on different streams?  (the
16-bit iteration is ~4.7 ms of forward + chain kernels that use < half the HBM bandwidth and ~3.5 ms of weight-gradient
kernels that are HBM-bound with the matrix pipe a third busy.)
Two independent snerf_mlp_backward calls (main 8x256 MLP, 2048 x 192 samples each) and a forward_train + backward pair:
  serial      both on one stream
  two         each on its own (ordinary) stream
  halves      each on its own stream restricted to half of the CUs by hipExtStreamCreateWithCUMask (alternating CU pairs, so
              both halves span every XCD)
Prints ms per pair (median of 5 rounds of 10).   python tools/probes/overlap_backward.py [precision 0|1|2, default 2]"""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from simplenerf_amd import ops, synth
from tests import util
from simplenerf_amd.synth import abi_param_list

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


cfg = synth.mlp_config(128)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 1)
plist = abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()})
shapes = [tuple(p.shape) for p in plist]
n, s = 2048, 192
work = []
for i in range(2):
    mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(plist)
    o = torch.rand(n, 3, device='cuda'); d = torch.rand(n, 3, device='cuda'); v = d / d.norm(dim=1, keepdim=True)
    z = torch.sort(torch.rand(n, s, device='cuda'), 1)[0]
    sigma, rgb, saved = mlp.forward_train(o, d, v, z, None, prec)
    into = [torch.zeros(sh, device='cuda') for sh in shapes]
    work.append(dict(mlp=mlp, o=o, d=d, v=v, z=z, sigma=sigma, rgb=rgb, saved=saved, gs=torch.randn(n, s, 1, device='cuda') * 1e-4,
                     gr=torch.randn(n, s, 3, device='cuda') * 1e-4, into=into))
torch.cuda.synchronize()


def backward(w):
    w['mlp'].backward(w['saved'], w['sigma'], w['rgb'], w['gs'], w['gr'], shapes, prec, into=w['into'])


def forward(w):
    w['mlp'].forward_train(w['o'], w['d'], w['v'], w['z'], None, prec)


def timed(fn_a, fn_b, sa, sb):
    times = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            with torch.cuda.stream(sa):
                fn_a(work[0])
            with torch.cuda.stream(sb):
                fn_b(work[1])
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 10 * 1e3)
    return statistics.median(times)


main = torch.cuda.current_stream()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    backward(work[0]); forward(work[1]); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
# 256 CUs = 8 mask words; alternate pairs of CUs between the two halves
h1, h2 = masked_stream([0x33333333] * 8), masked_stream([0xCCCCCCCC] * 8)
for label, fa, fb in (('backward + backward', backward, backward), ('forward_train + backward', forward, backward)):
    print(f'{label}: serial {timed(fa, fb, main, main):.3f} ms, two streams {timed(fa, fb, s1, s2):.3f} ms, '
          f'two half-GPU streams {timed(fa, fb, h1, h2):.3f} ms', flush=True)
