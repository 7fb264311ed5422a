#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the ray-marching hot path (coarse+fine, 128+128 samples, 8x256 MLPs).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (--train: BASELINE config 5 instead, see train_bench)

``--gpus N`` with N > 1 needs no wrapper: when no launcher has set WORLD_SIZE, bench.py starts N fresh rank processes
itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) BEFORE anything in
this process touches the GPU, relays rank 0's JSON line and exits with the ranks' status.  Under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N``
the launcher's environment is used as it is.  One rank per GPU, RCCL (torch.distributed backend "nccl") between them.

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
  collective    (N > 1) backend and number of ranks torch.distributed reports
  also_measured*        the same step in the other two arithmetic modes (N = 1)
  also_measured_train   BASELINE config 5 (the reference's training iteration, 4096 rows per GPU) in fp32 and in the
                        16-bit mode: ms per iteration, algorithmic TFLOP/s, fraction of the matching MFMA peak
  sustained     the headline step repeated for ~1 s of device time (the K timed steps alone are ~65 ms)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# (simplenerf_amd is imported inside the rank-side functions: the self-launching parent must not load the HIP library)

RAYS_PER_GPU = 1024
FLOP_PER_SAMPLE = 2 * 593408          # main 8x256 MLP, Linear layers only
TRAIN_FLOP_PER_RAY = (64 * 2 * (593408 + 577280 + 492032) + 192 * 2 * 593408) * 3   # config 5: forward x3 (dgrad + wgrad)
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_FP16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)
WORKLOAD = ('headline: 1024 rays/GPU x (128 coarse + 128 fine -> 256 merged) samples, 8x256 coarse+fine MLPs, '
            'LLFF fern NDC rays, eval')


def launch_ranks(num_ranks: int) -> int:
    """Self-launch for ``--gpus N`` without a launcher: N child processes, one per GPU, each a fresh interpreter running
    this file with the rank environment set.  The parent never initialises the GPU (no HIP call, no library load) and
    never exec()s: it waits, forwards rank 0's output and returns the worst exit status.  If a rank dies the others are
    stopped (they would wait in the rendezvous forever)."""
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs, logs = [], []
    for rank in range(num_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(num_ranks), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        log = tempfile.TemporaryFile(mode='w+')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=log,
                                      stderr=None if rank == 0 else subprocess.STDOUT))
    codes = [None] * num_ranks
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()          # our own children, by handle
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    failed = [i for i, c in enumerate(codes) if c != 0]
    for i, log in enumerate(logs):
        log.seek(0)
        text = log.read()
        if i == 0:
            sys.stdout.write(text)
        elif i in failed and text.strip():
            sys.stderr.write(f'[rank {i}] ' + text[-4000:] + '\n')
        log.close()
    sys.stdout.flush()
    return 0 if not failed else next(c for c in codes if c != 0)


def _pkg():
    from simplenerf_amd import harness, ops, synth
    from simplenerf_amd.models.ModelFactory import get_model
    return harness, ops, synth, get_model


def synthetic_model(configs, seed, device, precision='fp32'):
    _, _, synth, get_model = _pkg()
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


def cpu_baseline(configs, camera, first_ray, runs=9):
    """Oracle on the host CPU, same rays and weights as the GPU step; bounded: 1 warm-up + ``runs`` passes over the 1024
    rays (~10 s of CPU work on 16 cores)."""
    from oracle import nerf_oracle, raygen_oracle
    _, _, synth, get_model = _pkg()
    cores = host_cores()
    torch.set_num_threads(cores)
    shapes = {k: tuple(v.shape) for k, v in get_model(configs, None).state_dict().items()}
    params = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, sigma_gain=200.0, sigma_shift=8.0).items()}
    full = raygen_oracle.full_frame_batch(camera['resolution'], camera['intrinsic'], camera['pose'], camera['near'],
                                          camera['far'], True, camera['near_ndc'], camera['far_ndc'])
    batch = {k: torch.from_numpy(numpy.ascontiguousarray(v[first_ray:first_ray + RAYS_PER_GPU])) for k, v in full.items()}
    times = []
    with torch.no_grad():
        for i in range(runs + 1):
            t0 = time.perf_counter()
            nerf_oracle.render(params, configs, batch, training=False)
            if i > 0:
                times.append(time.perf_counter() - t0)
    best = min(times)
    return {'value': RAYS_PER_GPU / best, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{RAYS_PER_GPU} rays of the same workload, best of {runs} after 1 warm-up ({best:.3f} s best, '
                      f'{sum(times) / len(times):.3f} s mean; {sum(times):.1f} s of CPU work), torch {torch.__version__} CPU fp32, '
                      f'chunk 4096 / netchunk 16384'}


def pmc_traffic(precision):
    """HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run (profiles/), or None."""
    path = os.path.join(REPO, 'profiles', 'pmc_traffic.json' if precision == 'fp32' else f'pmc_traffic_{precision}.json')
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f).get('mlp_forward_hbm_bytes_per_launch')
    return None


# ---------------------------------------------------------------------------------------------- config 5 (training)
def training_step(precision, rank, world, device, single_pass=False):
    """BASELINE config 5: a callable running ONE iteration of the reference's training loop (Trainer.train_one_iter,
    src/Trainer01.py:60-107) with every stage on the device: batch assembly (2048 pixel + 2048 sparse-depth rows per GPU,
    each rank a slice of one global index stream), four MLPs forward, nine losses, backward, ONE all-reduce of the
    flattened gradients (RCCL) for N > 1, Adam with the decayed rate.  Weak scaling: 4096 rows per GPU."""
    harness, _, synth, get_model = _pkg()
    from simplenerf_amd import optim
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer
    rows = 2048
    cfg = synth.training_configs(precision, num_rays=rows * world, num_sparse=rows * world)
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
                                      single_pass=single_pass)

    return step, 2 * rows


TRAIN_WORKLOAD = ('config 5: 2048 pixel + 2048 sparse-depth rows per GPU in two sub-batches, main coarse+fine + points-aug + '
                  'views-aug MLPs (64 + 192 samples), nine shipped losses, Adam, NeRF LR decay')
TRAIN_DTYPE = {'fp32': 'f32', 'f16x3': 'f16x3', 'f16': 'f16 (bf16 layer gradients)'}


def time_training(precision, device, steps, warmup, single_pass=False, board_seconds=0.0):
    """Config 5 on one GPU: (ms per iteration, ms of MLP forward launches, ms of MLP backward calls) from ``steps`` timed
    iterations after ``warmup``."""
    _, ops, _, _ = _pkg()
    step, rows = training_step(precision, 0, 1, device, single_pass)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ops.profile_enable(64 * steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fwd, _ = ops.profile_collect(ops.PROFILE_MLP_FORWARD)
    bwd, _ = ops.profile_collect(ops.PROFILE_MLP_BACKWARD)
    ops.profile_enable(0)
    time_training.board = None
    if board_seconds > 0:      # the same iteration for about a second with the board's power / clock sensors sampled
        sampler = BoardSampler(device.index or 0)
        with sampler:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < board_seconds:
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
        time_training.board = sampler.summary()
    return elapsed / steps * 1e3, sum(fwd) / steps, sum(bwd) / steps, rows


def training_record(device, steps=10, warmup=3):
    """The ``also_measured_train`` object of the default bench line: config 5 at 4096 rows on this GPU in the fp32 mode
    (the reference's arithmetic), the fp16-split mode (same parity tests) and the 16-bit mode BASELINE config 5 names, each
    against its own MFMA ceiling."""
    dominant = {'fp32': 'wgrad_kernel<2,8,false> (weight gradients); forward mlp_forward_kernel<8,4,true,false,true>',
                'f16x3': 'wgrad_kernel<2,8,true> (weight gradients); chain mlp_backward_chain_f16x3_kernel<8,4,true,3,8>',
                'f16': 'chain mlp_backward_chain_f16x3_kernel<8,4,true,1,8>; wgrad16_kernel<2,8> (weight gradients, HBM-bound, '
                       '6.4 TB/s); forward mlp_forward_f16x3_kernel<8,4,true,false,true,1,8>'}
    out = {'workload': TRAIN_WORKLOAD, 'rows_per_gpu': 4096, 'steps': steps, 'warmup': warmup, 'modes': {}}
    # (f16x3 issues three fp16 MFMA passes per algorithmic product: its ceiling is a third of the fp16 peak)
    for precision, peak in (('fp32', PEAK_FP32_MFMA_TFLOPS), ('f16x3', PEAK_FP16_MFMA_TFLOPS / 3), ('f16', PEAK_FP16_MFMA_TFLOPS)):
        ms, fwd_ms, bwd_ms, rows = time_training(precision, device, steps, warmup, board_seconds=1.0)
        tflops = rows * TRAIN_FLOP_PER_RAY / (ms * 1e-3) / 1e12
        out['modes'][precision] = {
            'board': time_training.board,
            'dtype': TRAIN_DTYPE[precision], 'ms_per_step': ms, 'value': rows / (ms * 1e-3), 'unit': 'rays/s',
            'algorithmic_tflops': tflops, 'peak_tflops': peak, 'frac_of_peak': tflops / peak,
            'mlp_forward_ms_per_step': fwd_ms, 'mlp_backward_ms_per_step': bwd_ms,
            'mlp_share_of_step': (fwd_ms + bwd_ms) / ms, 'dominant_kernels': dominant[precision]}
    return out


def train_bench(args, rank, world, device, dist):
    """--train: BASELINE config 5 instead of the headline metric (see training_step)."""
    step, per_gpu = training_step(args.precision, rank, world, device, args.single_pass)

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
        line = {
            'metric': 'training rays/sec (config 5: forward + backward + optimiser, 4 MLPs, 9 losses)',
            'value': per_gpu * world * args.steps / elapsed, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': TRAIN_DTYPE[args.precision], 'data': 'synthetic',
            'config': {'workload': TRAIN_WORKLOAD, 'rows_per_gpu': per_gpu, 'single_pass': bool(args.single_pass),
                       'parallelism': f'row-shard x{world}' + (' + 1 gradient all-reduce/step' if world > 1 else '')},
            'algorithmic_tflops': per_gpu * TRAIN_FLOP_PER_RAY * world * args.steps / elapsed / 1e12}
        if world > 1:
            line['collective'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size()}
        print(json.dumps(line), flush=True)


# ---------------------------------------------------------------------------------------------- stand-in (tests only)
def standin_bench(args, rank, world, dist):
    """SNERF_BENCH_STANDIN=1 (tests/test_dist_gloo.py only): the launcher, rendezvous, per-rank block, gather, barrier,
    max-over-ranks timing and JSON line of the render bench with the renderer replaced by a CPU stand-in, so that the
    N > 1 path is exercised without GPUs.  The line says so (``"data": "stand-in"``); it is not a measurement."""
    from simplenerf_amd import harness
    first = rank * RAYS_PER_GPU

    def step():
        idx = torch.arange(first, first + RAYS_PER_GPU, dtype=torch.float32)
        local = {'rgb_fine': torch.stack([idx, 2 * idx, 3 * idx], 1), 'depth_fine': idx + 0.5}
        return harness.gather_rays(local, world * RAYS_PER_GPU, rank, world) if world > 1 else local

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ref = torch.arange(world * RAYS_PER_GPU, dtype=torch.float32)
        assert torch.equal(full['depth_fine'], ref + 0.5) and torch.equal(full['rgb_fine'][:, 2], 3 * ref)
        line = {'metric': 'rays/sec (coarse+fine, 128+128 samples)', 'value': world * RAYS_PER_GPU * args.steps / elapsed,
                'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'none', 'data': 'stand-in', 'config': {'workload': 'launcher rehearsal, no renderer'}}
        if world > 1:
            line['collective'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size()}
        print(json.dumps(line), flush=True)


# ---------------------------------------------------------------------------------------------- headline
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
    ap.add_argument('--no-alt', action='store_true', help='skip the secondary measurements (other precisions, training, sustained)')
    ap.add_argument('--precision', choices=('fp32', 'f16x3', 'f16'), default='fp32',
                    help="arithmetic of the fused MLP kernel: fp32 MFMA; fp16 hi/lo split with 3 MFMAs per product "
                         "(fp32-grade results, same parity tests); or f16 = one fp16 MFMA per product with 16-bit saved "
                         "tensors (BASELINE config 5's 16-bit training mode, own tolerances: tests/test_gpu_f16.py)")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))       # nothing above this line touches the GPU

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher set WORLD_SIZE={world}')
    standin = os.environ.get('SNERF_BENCH_STANDIN') == '1'
    dist = None
    device = None
    if not standin:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: the HIP renderer has no CPU path')
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL ('nccl') on a multi-GPU node; SNERF_DIST_BACKEND=gloo lets the same code path be rehearsed with several
        # ranks sharing one GPU (RCCL refuses two ranks on one device) or, with the stand-in step, on the CPU
        backend = os.environ.get('SNERF_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    try:
        if standin:
            standin_bench(args, rank, world, dist)
        elif args.train:
            train_bench(args, rank, world, device, dist)
        else:
            render_bench(args, rank, world, device, dist)
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


class BoardSampler:
    """Board power and shader clock of THIS process's GPU while a measurement runs (sysfs hwmon, one reading every 20 ms;
    no privileges needed).  The fp16 modes run the board at its power cap with the clock throttled below the 2.4 GHz the
    nominal peaks assume, so the line carries what the board did next to each fraction.  All fields None when the sensors
    are not readable."""

    def __init__(self, device_index=0):
        import glob
        import threading
        self._threading = threading
        self.paths = None
        try:
            props = torch.cuda.get_device_properties(device_index)
            want = '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), props.pci_bus_id, getattr(props, 'pci_device_id', 0))
        except Exception:
            return
        for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
            if want not in os.path.realpath(os.path.join(hw, '..', '..')):
                continue
            paths = {k: os.path.join(hw, f) for k, f in (('power', 'power1_average'), ('power', 'power1_input'),
                                                          ('cap', 'power1_cap'), ('sclk', 'freq1_input'))
                     if os.path.exists(os.path.join(hw, f))}
            if 'power' in paths or 'sclk' in paths:
                self.paths = paths

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def __enter__(self):
        self.rows, self._on = [], True
        if self.paths:
            def loop():
                while self._on:
                    self.rows.append({k: self._read(p) for k, p in self.paths.items() if k != 'cap'})
                    time.sleep(0.02)
            self._thread = self._threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._on = False
        if self.paths:
            self._thread.join()

    def summary(self):
        """mean over the last three quarters of the readings (the first quarter is the ramp)"""
        if not self.paths or len(self.rows) < 4:
            return {'power_w': None, 'power_cap_w': None, 'sclk_mhz': None, 'readings': len(getattr(self, 'rows', []))}
        rows = self.rows[len(self.rows) // 4:]

        def mean(key):
            vals = [r[key] for r in rows if r.get(key) is not None]
            return sum(vals) / len(vals) if vals else None
        power, sclk = mean('power'), mean('sclk')
        cap = self._read(self.paths['cap']) if 'cap' in self.paths else None
        return {'power_w': None if power is None else power / 1e6, 'power_cap_w': None if cap is None else cap / 1e6,
                'sclk_mhz': None if sclk is None else sclk / 1e6, 'readings': len(rows)}


def render_bench(args, rank, world, device, dist):
    harness, ops, synth, _ = _pkg()
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
        # the library brackets every fused PE+MLP launch with HIP events on the launch stream (snerf_profile_enable)
        ops.profile_enable(4 * args.steps + 16)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    launch_ms, launch_samples = ops.profile_collect(ops.PROFILE_MLP_FORWARD)
    ops.profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return

    kernel_ms = sum(launch_ms)
    kernel_flop = sum(launch_samples) * FLOP_PER_SAMPLE
    achieved = kernel_flop / (kernel_ms * 1e-3) / 1e12
    if args.precision == 'fp32':
        peak, dtype, kernel_name = PEAK_FP32_MFMA_TFLOPS, 'f32', 'mlp_forward_kernel<8,4,true,false,false>'
        note = 'fp32 MFMA: one pass per algorithmic FLOP'
    elif args.precision == 'f16x3':
        peak, dtype, kernel_name = PEAK_FP16_MFMA_TFLOPS, 'f16x3 (fp16 hi/lo split, fp32 accumulate)', \
            'mlp_forward_m16_kernel<3,8>'
        note = ('achieved counts ALGORITHMIC FLOPs; the kernel issues 3 fp16 MFMA passes per product, so its ceiling is '
                'peak/3 = 833 TFLOP/s and MFMA-pipe utilisation = 3 x frac')
    else:
        peak, dtype, kernel_name = PEAK_FP16_MFMA_TFLOPS, 'f16 (fp16 MFMA, fp32 accumulate)', \
            'mlp_forward_m16_kernel<1,8>'
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
                     'kernel': kernel_name, 'note': note, 'launches': len(launch_ms),
                     'avg_launch_ms': kernel_ms / max(1, len(launch_ms)),
                     'kernel_share_of_step': kernel_ms / (elapsed * 1e3)},
    }
    if world > 1:
        result['collective'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                                'pattern': 'one gather of (rgb, depth) = 16 B/ray to rank 0 per step'}
    if world == 1 and args.precision == 'fp32' and not args.no_alt:
        # the same step with the other two arithmetic modes of the fused MLP kernel, as secondary measurements
        def measure(precision, steps):
            alt_model = model if precision == 'fp32' else synthetic_model(configs, 7, device, precision)
            with torch.no_grad():
                for _ in range(args.warmup):
                    alt_model(harness.frame_batch(camera, True, device, first, RAYS_PER_GPU))
                torch.cuda.synchronize()
                ops.profile_enable(4 * steps + 16)
                t0 = time.perf_counter()
                for _ in range(steps):
                    alt_model(harness.frame_batch(camera, True, device, first, RAYS_PER_GPU))
                torch.cuda.synchronize()
                alt_elapsed = time.perf_counter() - t0
            alt_ms, alt_samples = ops.profile_collect(ops.PROFILE_MLP_FORWARD)
            ops.profile_enable(0)
            alt_tf = sum(alt_samples) * FLOP_PER_SAMPLE / (sum(alt_ms) * 1e-3) / 1e12
            measure.kernel_share = sum(alt_ms) / (alt_elapsed * 1e3)
            return alt_elapsed, alt_tf

        def board_state(precision, achieved_tflops, nominal_peak, seconds=1.2):
            """the same step back to back for ~1.2 s with the board's sensors sampled: power, cap, shader clock, and the
            fraction of the peak AT THAT CLOCK (the nominal peaks are quoted at 2.4 GHz)"""
            alt_model = model if precision == 'fp32' else synthetic_model(configs, 7, device, precision)
            sampler = BoardSampler(device.index or 0)
            with torch.no_grad(), sampler:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < seconds:
                    for _ in range(25):
                        alt_model(harness.frame_batch(camera, True, device, first, RAYS_PER_GPU))
                    torch.cuda.synchronize()
            state = sampler.summary()
            if state['sclk_mhz']:
                state['frac_of_peak_at_this_clock'] = achieved_tflops / (nominal_peak * state['sclk_mhz'] / 2400.0)
            return state

        sustained_steps = 300       # ~1 s of device time: long enough for the clocks to settle under the load
        s_elapsed, s_tf = measure('fp32', sustained_steps)
        result['sustained'] = {'steps': sustained_steps, 'value': RAYS_PER_GPU * sustained_steps / s_elapsed, 'unit': 'rays/s',
                               'ms_per_step': s_elapsed / sustained_steps * 1e3, 'achieved': s_tf,
                               'frac': s_tf / PEAK_FP32_MFMA_TFLOPS, 'timed_region_s': s_elapsed,
                               'board': board_state('fp32', s_tf, PEAK_FP32_MFMA_TFLOPS)}
        alt_elapsed, alt_tf = measure('f16x3', args.steps)
        result['also_measured'] = {
            'precision': 'f16x3 (fp16 hi/lo split, 3 MFMA passes per product, fp32 accumulate; same parity tests)',
            'value': RAYS_PER_GPU * args.steps / alt_elapsed, 'unit': 'rays/s', 'ms_per_step': alt_elapsed / args.steps * 1e3,
            'roofline': {'bound': 'mfma', 'achieved': alt_tf, 'peak': PEAK_FP16_MFMA_TFLOPS / 3, 'unit': 'TFLOP/s',
                         'frac': alt_tf / (PEAK_FP16_MFMA_TFLOPS / 3), 'frac_of_fp16_peak': alt_tf / PEAK_FP16_MFMA_TFLOPS,
                         'note': 'peak = fp16 dense MFMA peak / 3: the kernel issues three fp16 MFMA passes per algorithmic '
                                 'product, achieved counts algorithmic FLOPs',
                         'kernel': 'mlp_forward_m16_kernel<3,8>',
                         'kernel_share_of_step': measure.kernel_share},
            'board': board_state('f16x3', alt_tf, PEAK_FP16_MFMA_TFLOPS / 3)}
        alt_elapsed, alt_tf = measure('f16', args.steps)
        result['also_measured_16bit'] = {
            'precision': 'f16 (one fp16 MFMA pass per product, fp32 accumulate; OUTSIDE the fp32 parity bar -- colour ~1e-4, '
                         'depth ~6e-4 from the fp32 path, tests/test_gpu_f16.py; BASELINE config 5 names this mode for training)',
            'value': RAYS_PER_GPU * args.steps / alt_elapsed, 'unit': 'rays/s', 'ms_per_step': alt_elapsed / args.steps * 1e3,
            'roofline': {'bound': 'mfma', 'achieved': alt_tf, 'peak': PEAK_FP16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': alt_tf / PEAK_FP16_MFMA_TFLOPS, 'kernel': 'mlp_forward_m16_kernel<1,8>',
                         'kernel_share_of_step': measure.kernel_share},
            'board': board_state('f16', alt_tf, PEAK_FP16_MFMA_TFLOPS)}
        result['also_measured_train'] = training_record(device)
    if world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(configs, camera, first)
    print(json.dumps(result), flush=True)


if __name__ == '__main__':
    main()
