#!/usr/bin/env python3
"""What the vendor library's fp32 GEMM reaches on the layered path's product shapes (torch.matmul -> hipBLASLt / rocBLAS, fp32 in,
fp32 out, TF32 off): a yardstick for csrc/mlp_generic.hip's gemm kernels, not a dependency."""
import json

import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = 'cuda:0'


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for m, n, k, what in ((262144, 512, 512, 'forward / input gradient of a 512-wide layer: X (M x K) . W^T'),
                      (262144, 256, 256, '256-wide layer'),
                      (512, 512, 262144, 'weight gradient of a 512-wide layer: dZ^T (M x K) . X')):
    if k > m:
        a = torch.randn(k, m, device=dev).t()       # dZ^T: contiguous along m
        b = torch.randn(k, n, device=dev)
    else:
        a = torch.randn(m, k, device=dev)
        b = torch.randn(n, k, device=dev).t()       # W^T with W (N x K) row-major
    ms = timed(lambda: torch.matmul(a, b))
    print(json.dumps({'M': m, 'N': n, 'K': k, 'what': what, 'ms': round(ms, 4), 'tflops': round(2.0 * m * n * k / ms / 1e9, 1),
                      'fraction_of_fp32_mfma_peak': round(2.0 * m * n * k / ms / 1e9 / 157.3, 3)}), flush=True)
