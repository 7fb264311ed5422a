#!/usr/bin/env python3
"""The summation orders of the two reductions in sample_pdf (src/models/SimpleNeRF01.py:333-334) that K5 reproduces
(simplenerf_amd/csrc/resample_device.h: torch_row_sum and the serial fp64 running sum), restated in numpy and checked against
torch on this host:

  torch.sum(x, -1) of a contiguous float row      ATen SumKernel.cpp: 8-float vectors (sum_stub has no AVX-512 build, so an
                                                  AVX-512 host runs the AVX2 one), four interleaved accumulators, 16-step cascade
  torch.cumsum(x, -1) of a float row              ATen ReduceOpsKernel.cpp: sequential, accumulated in DOUBLE, every entry
                                                  rounded to float

    python tools/check_torch_sum_order.py            # prints mismatches per row length (expected: 0 everywhere)
    ATEN_CPU_CAPABILITY=avx2 python tools/check_torch_sum_order.py

tests/test_host_logic.py runs the same check on a few sizes, so a torch build that sums in another order is noticed.
"""
import numpy as np

f32 = np.float32


def _ceil_log2(x):
    return 1 if x <= 2 else int(x - 1).bit_length()


def _multi_row_sum(load, rows, size, width):
    levels = 4
    level_power = max(4, _ceil_log2(size) // levels)
    level_step = 1 << level_power
    level_mask = level_step - 1
    acc = [[np.zeros(width, f32) for _ in range(rows)] for _ in range(levels)]
    i = 0
    while i + level_step <= size:
        for _ in range(level_step):
            for k in range(rows):
                acc[0][k] = acc[0][k] + load(i, k)
            i += 1
        for j in range(1, levels):
            for k in range(rows):
                acc[j][k] = acc[j][k] + acc[j - 1][k]
                acc[j - 1][k] = np.zeros(width, f32)
            if (i & (level_mask << (j * level_power))) != 0:
                break
    while i < size:
        for k in range(rows):
            acc[0][k] = acc[0][k] + load(i, k)
        i += 1
    for j in range(1, levels):
        for k in range(rows):
            acc[0][k] = acc[0][k] + acc[j][k]
    return acc[0]


def _row_sum(vectors, width):
    ilp = 4
    groups = len(vectors) // ilp
    partial = _multi_row_sum(lambda i, k: vectors[i * ilp + k], ilp, groups, width)
    for i in range(groups * ilp, len(vectors)):
        partial[0] = partial[0] + vectors[i]
    for k in range(1, ilp):
        partial[0] = partial[0] + partial[k]
    return partial[0]


def torch_sum_row(x, width=8):
    """torch.sum of one contiguous float32 row, ATen's order."""
    x = np.asarray(x, f32)
    n = len(x)
    if n < width:
        return _row_sum([x[i:i + 1] for i in range(n)], 1)[0]
    whole = n // width
    lanes = _row_sum([x[i * width:(i + 1) * width] for i in range(whole)], width)
    total = f32(0)
    for k in range(whole * width, n):
        total = f32(total + x[k])
    for k in range(width):
        total = f32(total + lanes[k])
    return total


def torch_cumsum_row(x):
    """torch.cumsum of one float32 row on the CPU: a double accumulator, each entry rounded to float."""
    run = np.float64(0)
    out = np.empty(len(x), f32)
    for i, v in enumerate(np.asarray(x, f32)):
        run = run + np.float64(v)
        out[i] = f32(run)
    return out


def mismatches(n, rows=200, seed=0):
    import torch
    rng = np.random.RandomState(seed)
    x = (rng.rand(rows, n) ** 4).astype(f32) + f32(1e-5)
    t = torch.from_numpy(x)
    ref_sum = torch.sum(t, -1, keepdim=True).numpy()[:, 0]
    ref_cum = torch.cumsum(t / torch.sum(t, -1, keepdim=True), -1).numpy()
    pdf = (t / torch.sum(t, -1, keepdim=True)).numpy()
    bad_sum = int(sum(torch_sum_row(r) != s for r, s in zip(x, ref_sum)))
    bad_cum = int(sum((torch_cumsum_row(r) != c).any() for r, c in zip(pdf, ref_cum)))
    return bad_sum, bad_cum


if __name__ == '__main__':
    import torch
    print('torch', torch.__version__, 'capability', torch.backends.cpu.get_cpu_capability())
    worst = 0
    for n in (5, 14, 30, 62, 126, 190, 254, 510, 1022, 2000, 5000):
        bad = mismatches(n)
        worst = max(worst, *bad)
        print(f'row length {n}: sum mismatches {bad[0]}, cumsum mismatches {bad[1]} of 200 rows')
    raise SystemExit(1 if worst else 0)
