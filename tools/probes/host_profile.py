#!/usr/bin/env python3
"""Where the HOST time of an eager config-5 iteration goes (cProfile): one rank's share is enqueue-bound when issued eagerly
(2.7-3.0 ms per 512-row iteration against 1.5-1.9 from a graph).   usage: host_profile.py [rows] [precision] [iterations]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
precision = sys.argv[2] if len(sys.argv) > 2 else 'f16'
iterations = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
step, n = bench.training_step(precision, 0, 1, dev, False, False, False, rows)
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iterations):
    step()
host = (time.perf_counter() - t0) / iterations * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / iterations * 1e3
print(f'{n} rows, {precision}, eager: {host:.3f} ms of enqueue per iteration, {total:.3f} ms per iteration')
prof = cProfile.Profile()
prof.enable()
for _ in range(iterations):
    step()
prof.disable()
torch.cuda.synchronize()
stats = pstats.Stats(prof)
stats.sort_stats('tottime').print_stats(35)
stats.sort_stats('cumulative').print_stats(45)
