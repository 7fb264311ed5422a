#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the ray-marching hot path (coarse+fine, 128+128 samples, 8x256 MLPs).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (--train: BASELINE config 5 instead, see train_bench)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of 1024 rays per GPU: on-device ray generation for the
rank's pixel block of a fern frame, coarse depths, coarse MLP, compositing, inverse-CDF resampling + merge, fine MLP,
compositing (eval mode, every reference output incl. alpha), and -- for N > 1 -- the single gather of the per-ray
colour/depth to rank 0.  Inputs (camera, weights) are resident in HBM before the timed region.  Weak scaling: each
rank renders its own 1024-ray block; `value` is the whole-job rays/s = N*1024*K / max-over-ranks time.

Printed JSON (rank 0, one line) also carries
  roofline      fp32-MFMA roofline of the dominant kernel (fused PE+MLP forward): algorithmic FLOPs of its launches
                inside the timed region / their HIP-event durations, against 157.3 TFLOP/s (MI355X_MICROARCH.md)
  cpu_baseline  the oracle (torch CPU fp32 restatement of the reference path, reference chunking) timed on this
                host on the same 1024-ray batch, best of 5 after one warm-up
"""
import argparse
import json
import os
import sys
import time

import numpy
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from simplenerf_amd import harness, ops, synth  # noqa: E402
from simplenerf_amd.models.ModelFactory import get_model  # noqa: E402

RAYS_PER_GPU = 1024
FLOP_PER_SAMPLE = 2 * 593408          # main 8x256 MLP, Linear layers only
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_FP16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)
WORKLOAD = ('headline: 1024 rays/GPU x (128 coarse + 128 fine -> 256 merged) samples, 8x256 coarse+fine MLPs, '
            'LLFF fern NDC rays, eval')


def synthetic_model(configs, seed, device, precision='fp32'):
    configs = synth.with_overrides(configs, hip_precision=precision)
    model = get_model(configs, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed, sigma_gain=200.0, sigma_shift=8.0).items()})
    return model.to(device).eval()


def host_cores():
    """Threads for the CPU leg: the cores this process may run on, capped at the GPU box's per-GPU CPU share (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(configs, camera, first_ray):
    """Oracle on the host CPU, same rays and weights as the GPU step; bounded: 1 warm-up + 5 runs of 1024 rays."""
    from oracle import nerf_oracle, raygen_oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    shapes = {k: tuple(v.shape) for k, v in get_model(configs, None).state_dict().items()}
    params = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, sigma_gain=200.0, sigma_shift=8.0).items()}
    full = raygen_oracle.full_frame_batch(camera['resolution'], camera['intrinsic'], camera['pose'], camera['near'],
                                          camera['far'], True, camera['near_ndc'], camera['far_ndc'])
    batch = {k: torch.from_numpy(numpy.ascontiguousarray(v[first_ray:first_ray + RAYS_PER_GPU])) for k, v in full.items()}
    best = float('inf')
    with torch.no_grad():
        for i in range(6):
            t0 = time.perf_counter()
            nerf_oracle.render(params, configs, batch, training=False)
            dt = time.perf_counter() - t0
            if i > 0:
                best = min(best, dt)
    return {'value': RAYS_PER_GPU / best, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{RAYS_PER_GPU} rays of the same workload, best of 5 after 1 warm-up ({best:.3f} s each), '
                      f'torch {torch.__version__} CPU fp32, chunk 4096 / netchunk 16384'}


def pmc_traffic(precision):
    """HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run (profiles/), or None."""
    path = os.path.join(REPO, 'profiles', 'pmc_traffic.json' if precision == 'fp32' else f'pmc_traffic_{precision}.json')
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f).get('mlp_forward_hbm_bytes_per_launch')
    return None


def train_bench(args, rank, world, device, dist):
    """--train: BASELINE config 5 instead of the headline metric.  One step = the reference's training iteration
    (Trainer.train_one_iter) with every stage on the device: batch assembly (2048 pixel + 2048 sparse-depth rows per GPU,
    each rank a slice of one global index stream), four MLPs forward, nine losses, backward, ONE all-reduce of the
    flattened gradients (RCCL) for N > 1, Adam with the decayed rate.  Weak scaling: 4096 rows per GPU."""
    from simplenerf_amd import optim
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer
    rows = 2048
    cfg = synth.training_configs(args.precision, num_rays=rows * world, num_sparse=rows * world)
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    model = model.to(device).train()
    batcher = BatchAssembler(cfg, synth.training_scene(), device, rank=rank, world_size=world)
    losses = LossComputer(cfg)
    opt = optim.Adam(list(model.parameters()), lr=cfg['optimizer']['lr_initial'],
                     betas=(cfg['optimizer']['beta1'], cfg['optimizer']['beta2']))
    decayer = get_lr_decayer(cfg)
    state = {'iter': 20000}

    def step():
        it = state['iter']
        state['iter'] += 1
        for group in opt.param_groups:
            group['lr'] = decayer.get_updated_learning_rate(it)
        return harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it), cfg['sub_batch_size'], world,
                                      single_pass=args.single_pass)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        per_gpu = 2 * rows
        flop = per_gpu * (64 * 2 * (593408 + 577280 + 492032) + 192 * 2 * 593408) * 3      # forward x3 (dgrad + wgrad)
        print(json.dumps({
            'metric': 'training rays/sec (config 5: forward + backward + optimiser, 4 MLPs, 9 losses)',
            'value': per_gpu * world * args.steps / elapsed, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': {'fp32': 'f32', 'f16x3': 'f16x3', 'f16': 'f16 (bf16 layer gradients)'}[args.precision], 'data': 'synthetic',
            'config': {'workload': 'config 5: 2048 pixel + 2048 sparse-depth rows per GPU in two sub-batches, main coarse+fine '
                                   '+ points-aug + views-aug MLPs (64 + 192 samples), nine shipped losses, Adam, NeRF LR decay',
                       'rows_per_gpu': per_gpu, 'single_pass': bool(args.single_pass), 'parallelism': f'row-shard x{world}' + (' + 1 gradient all-reduce/step' if world > 1 else '')},
            'algorithmic_tflops': flop * world * args.steps / elapsed / 1e12}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train', action='store_true',
                    help='measure BASELINE config 5 (training iteration) instead of the headline render metric')
    ap.add_argument('--single-pass', action='store_true',
                    help='--train: one model forward/backward over the whole batch, losses still normalised per sub-batch '
                         '(harness.train_one_iter single_pass)')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the secondary f16x3 / f16 measurements')
    ap.add_argument('--precision', choices=('fp32', 'f16x3', 'f16'), default='fp32',
                    help="arithmetic of the fused MLP kernel: fp32 MFMA; fp16 hi/lo split with 3 MFMAs per product "
                         "(fp32-grade results, same parity tests); or f16 = one fp16 MFMA per product with 16-bit saved "
                         "tensors (BASELINE config 5's 16-bit training mode, own tolerances: tests/test_gpu_f16.py)")
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP renderer has no CPU path')
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL ('nccl') on a multi-GPU node; SNERF_DIST_BACKEND=gloo lets the same code path be rehearsed with several
        # ranks sharing one GPU (RCCL refuses two ranks on one device)
        backend = os.environ.get('SNERF_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.train:
        train_bench(args, rank, world, device, dist)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    configs = synth.make_configs('headline')
    camera = synth.camera('fern', 0)
    h, w = camera['resolution']
    model = synthetic_model(configs, 7, device, args.precision)
    # rank r renders pixels [base + r*1024, base + (r+1)*1024) from the middle of the frame
    base = (h // 2) * w
    first = base + rank * RAYS_PER_GPU
    keys = ('rgb_fine', 'depth_fine')

    def step():
        batch = harness.frame_batch(camera, True, device, first, RAYS_PER_GPU)
        out = model(batch)
        local = {k: out[k] for k in keys}
        if world > 1:
            return harness.gather_rays(local, world * RAYS_PER_GPU, rank, world)
        return local

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        ops.PackedMlp.event_log = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    log, ops.PackedMlp.event_log = ops.PackedMlp.event_log, None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        kernel_ms = sum(a.elapsed_time(b) for a, b, _ in log)
        kernel_flop = sum(n for _, _, n in log) * FLOP_PER_SAMPLE
        achieved = kernel_flop / (kernel_ms * 1e-3) / 1e12
        if args.precision == 'fp32':
            peak, dtype, kernel_name = PEAK_FP32_MFMA_TFLOPS, 'f32', 'mlp_forward_kernel<8,4,true,false,false>'
            note = 'fp32 MFMA: one pass per algorithmic FLOP'
        elif args.precision == 'f16x3':
            peak, dtype, kernel_name = PEAK_FP16_MFMA_TFLOPS, 'f16x3 (fp16 hi/lo split, fp32 accumulate)', \
                'mlp_forward_f16x3_kernel<8,4,true,false,false,3>'
            note = ('achieved counts ALGORITHMIC FLOPs; the kernel issues 3 fp16 MFMA passes per product, so its ceiling is '
                    'peak/3 = 833 TFLOP/s and MFMA-pipe utilisation = 3 x frac')
        else:
            peak, dtype, kernel_name = PEAK_FP16_MFMA_TFLOPS, 'f16 (fp16 MFMA, fp32 accumulate)', \
                'mlp_forward_f16x3_kernel<8,4,true,false,false,1>'
            note = 'one fp16 MFMA pass per product; NOT within the fp32 parity bar (sigma ~1e-3 relative)'

        result = {
            'metric': 'rays/sec (coarse+fine, 128+128 samples)',
            'value': world * RAYS_PER_GPU * args.steps / elapsed,
            'unit': 'rays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': WORKLOAD, 'rays_per_gpu': RAYS_PER_GPU, 'samples': '128+128',
                       'parallelism': f'ray-shard x{world}' + (' + 1 gather/step' if world > 1 else '')},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': pmc_traffic(args.precision),
                         'kernel': kernel_name, 'note': note, 'launches': len(log),
                         'avg_launch_ms': kernel_ms / max(1, len(log)),
                         'kernel_share_of_step': kernel_ms / (elapsed * 1e3)},
        }
        if world == 1 and args.precision == 'fp32' and not args.no_alt:
            # the same step with the other two arithmetic modes of the fused MLP kernel, as secondary measurements
            def measure(precision):
                alt_model = synthetic_model(configs, 7, device, precision)
                with torch.no_grad():
                    for _ in range(args.warmup):
                        alt_model(harness.frame_batch(camera, True, device, first, RAYS_PER_GPU))
                    torch.cuda.synchronize()
                    ops.PackedMlp.event_log = []
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        alt_model(harness.frame_batch(camera, True, device, first, RAYS_PER_GPU))
                    torch.cuda.synchronize()
                    alt_elapsed = time.perf_counter() - t0
                alt_log, ops.PackedMlp.event_log = ops.PackedMlp.event_log, None
                alt_ms = sum(a.elapsed_time(b) for a, b, _ in alt_log)
                alt_tf = sum(n for _, _, n in alt_log) * FLOP_PER_SAMPLE / (alt_ms * 1e-3) / 1e12
                return alt_elapsed, alt_tf

            alt_elapsed, alt_tf = measure('f16x3')
            result['also_measured'] = {
                'precision': 'f16x3 (fp16 hi/lo split, 3 MFMA passes per product, fp32 accumulate; same parity tests)',
                'value': RAYS_PER_GPU * args.steps / alt_elapsed, 'unit': 'rays/s', 'ms_per_step': alt_elapsed / args.steps * 1e3,
                'roofline': {'bound': 'mfma', 'achieved': alt_tf, 'peak': PEAK_FP16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': alt_tf / PEAK_FP16_MFMA_TFLOPS, 'kernel': 'mlp_forward_f16x3_kernel<8,4,true,false,false,3>'}}
            alt_elapsed, alt_tf = measure('f16')
            result['also_measured_16bit'] = {
                'precision': 'f16 (one fp16 MFMA pass per product, fp32 accumulate; OUTSIDE the fp32 parity bar -- colour ~1e-4, '
                             'depth ~6e-4 from the fp32 path, tests/test_gpu_f16.py; BASELINE config 5 names this mode for training)',
                'value': RAYS_PER_GPU * args.steps / alt_elapsed, 'unit': 'rays/s', 'ms_per_step': alt_elapsed / args.steps * 1e3,
                'roofline': {'bound': 'mfma', 'achieved': alt_tf, 'peak': PEAK_FP16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': alt_tf / PEAK_FP16_MFMA_TFLOPS, 'kernel': 'mlp_forward_f16x3_kernel<8,4,true,false,false,1>'}}
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(configs, camera, first)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
