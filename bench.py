#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the ray-marching hot path (coarse+fine, 128+128 samples, 8x256 MLPs).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --frame re10k [--gpus N]      BASELINE config 4: ONE full frame strong-scaled over N ranks + gather
    python bench.py --train [--precision f16]     BASELINE config 5: the training iteration (see train_bench)

``--gpus N`` with N > 1 needs no wrapper: when no launcher has set WORLD_SIZE, bench.py starts N fresh rank processes
itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) BEFORE anything in
this process touches the GPU, relays rank 0's JSON line and exits with the ranks' status.  Under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N``
the launcher's environment is used as it is.  One rank per GPU, RCCL (torch.distributed backend "nccl") between them;
a launch with more ranks than the node has GPUs is refused.  ``--force-collective`` runs the N > 1 protocol (process group,
barrier, the per-step gather, max-over-ranks reduction) with N = 1, so that every RCCL call of the multi-GPU line executes
on a one-GPU box.  (GPU-less rehearsals of the launcher and of the N > 1 protocol live in tests/bench_rehearsal.py, which
calls ``main`` with a stand-in renderer and the gloo backend; this file has no such switch.)

One "step" = one pass of the whole hot path over one batch of 1024 rays per GPU: on-device ray generation for the
rank's pixel block of a fern frame, coarse depths, coarse MLP, compositing, inverse-CDF resampling + merge, fine MLP,
compositing (eval mode, every reference output incl. alpha), and -- for N > 1 -- the single gather of the per-ray
colour/depth to rank 0.  Inputs (camera, weights) are resident in HBM before the timed region.  Weak scaling: each
rank renders its own 1024-ray block; `value` is the whole-job rays/s = N*1024*K / max-over-ranks time.

Timing protocol (every mode, every precision; ``timed_steps``):
  settle    the step's device work back to back for >= 0.6 s, untimed -- a fresh GPU lease starts from idle clocks, an
            empty allocator and cold code paths, which a handful of 3-ms warm-up steps does not absorb (round 2's driver
            run lost 22 ms of a 88-ms timed region that way)
  warm-up   W steps of the full step (incl. the collective), untimed, with the library's event hooks already on
  timed     barrier + synchronize, EXACTLY K steps, barrier + synchronize; `value` = work / that whole wall time.  Every
            step also gets one event on the launch stream and one host stamp after its enqueue, reported as
            ``step_ms {p50, p90, max, first, argmax}`` (device time between consecutive end-of-step events),
            ``enqueue_ms`` (host time per step) and ``idle_ms_per_step`` (wall per step minus the event-timed fused
            PE+MLP launches: device idle + the ~1 % of small kernels), so that a one-off stall is visible and attributable
            to the host or the device; ``step_trace_ms`` holds the raw per-step device times when K <= 64

Output contract (round 5).  stdout carries exactly ONE line of at most 2 KB, printed LAST, holding only
  metric value unit n_gpus steps warmup ms_per_step higher_is_better scaling vs_baseline dtype data config
  roofline      {bound, achieved, peak, unit, frac, traffic, traffic_algorithmic, traffic_source, kernel, launches,
                avg_launch_ms}: the MFMA roofline of the dominant kernel (fused PE+MLP forward) -- algorithmic FLOPs of its
                launches inside the timed region / their HIP-event durations on the launch stream, against the dense MFMA
                peak of the arithmetic used (MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 matrix, 2 500 TFLOP/s fp16 / bf16)
  cpu_baseline  {value, unit, cores, kind, sample}: the oracle (torch CPU fp32 restatement of the reference path, reference
                chunking) timed on this host on the same 1024-ray batch, with as many threads as the job OWNS cores
                (affinity mask capped by the cgroup CPU quota), bounded to ~10 s of CPU work
  collective    (N > 1) {backend, ranks, bytes, gather_ms_p50}
  also          (N = 1 default run) a handful of scalars: the same step in f16x3 / f16 / bf16 and the config-5 iteration in
                the 16-bit modes, each with its fraction of the matching MFMA ceiling
  extra         path of the side file (JSON) with EVERYTHING else: per-step timing, per-rank tables, the full records of the
                secondary measurements.  ``--extras`` adds the long secondary set to it (sustained run, board power / clock,
                whole frames of configs 2 and 4, config 5 in six precisions and three issue modes, rank share)
Nothing is written to stderr unless ``--verbose`` (library chatter and Python warnings go to a log that is replayed on
stderr only when the run fails), so the last non-empty line of stdout + stderr, however a caller merges them, is the result.
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
SETTLE_SECONDS = 0.6
METRIC = 'rays/sec (coarse+fine, 128+128 samples)'

# per-precision constants of the headline kernel: (ceiling in algorithmic TFLOP/s, dtype string, kernel, note)
PRECISION_INFO = {
    'fp32': (PEAK_FP32_MFMA_TFLOPS, 'f32', 'mlp_forward_kernel<8,4,true,false,false,false>', 'fp32 MFMA: one pass per algorithmic FLOP'),
    'f16x3': (PEAK_FP16_MFMA_TFLOPS / 3, 'f16x3 (fp16 hi/lo split, fp32 accumulate)', 'mlp_forward_m16_kernel<3,8>',
              'peak = fp16 dense MFMA peak / 3: the kernel issues three fp16 MFMA passes per algorithmic product, achieved '
              'counts algorithmic FLOPs'),
    'f16': (PEAK_FP16_MFMA_TFLOPS, 'f16 (fp16 MFMA, fp32 accumulate)', 'mlp_forward_m16_kernel<1,8>',
            'one fp16 MFMA pass per product; NOT within the fp32 parity bar (sigma ~1e-3 relative)'),
    'f16s8': (PEAK_FP16_MFMA_TFLOPS, 'f16 (fp16 MFMA, fp32 accumulate; training keeps the trunk activations as fp8)', 'mlp_forward_m16_kernel<1,8>',
              "rendering in this mode IS the f16 mode; only what the training forward saves differs"),
    'bf16': (PEAK_FP16_MFMA_TFLOPS, 'bf16 (bf16 MFMA, fp32 accumulate)', 'mlp_forward_m16_kernel<1,8,true>',
             'one bf16 MFMA pass per product (BASELINE config 5\'s literal dtype; no range limit); NOT within the fp32 parity bar '
             '(sigma ~1e-2 relative, tests/test_gpu_bf16.py)'),
    'bf16s8': (PEAK_FP16_MFMA_TFLOPS, 'bf16 (bf16 MFMA, fp32 accumulate; training keeps the trunk activations as fp8)', 'mlp_forward_m16_kernel<1,8,true>',
               "rendering in this mode IS the bf16 mode; only what the training forward saves differs"),
}

# whole-frame workloads (BASELINE configs 2 and 4; SURVEY 8d resolves the resolutions): name -> (scene, camera kwargs, text)
FRAMES = {
    'fern': ('fern', {}, 'config 2, reference-native frame: LLFF fern 1008x756'),
    'fern504': ('fern', {'downscale': 2}, 'config 2 as BASELINE names it: LLFF fern 504x378'),
    're10k': ('re10k', {'resolution': (756, 1008)}, 'config 4 as BASELINE names it: RealEstate-10K camera at 1008x756'),
    're10k_native': ('re10k', {}, "config 4, the scene's own frame size: RealEstate-10K 1024x576"),
}
FRAME_SAMPLES = 64 + 192              # config 2 / 4 evaluate 64 coarse + 192 merged fine samples per ray
FRAME_DISPLAY_BYTES = 3 + 4 * 4       # uint8 colour + depth, depth_var, depth_ndc, depth_var_ndc per pixel leave the device
FRAME_GATHER_BYTES = 4 * (3 + 4)      # fp32 colour + the four depth columns per ray cross xGMI to rank 0


def launch_ranks(num_ranks: int, script=None) -> int:
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
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=log,
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


_RESULT_FD = None
_QUIET_LOG = None        # (file object, saved stderr descriptor) while stderr is held back
VERBOSE = False
EXTRA_FILE = None        # side file of the full record (set in main)

LINE_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
             'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'collective', 'also', 'extra')
ROOFLINE_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_algorithmic', 'traffic_source', 'kernel',
                 'launches', 'avg_launch_ms')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'sample')
COLLECTIVE_KEYS = ('backend', 'ranks', 'bytes', 'gather_ms_p50')
LINE_LIMIT = 2048


def claim_stdout(quiet=True):
    """stdout carries exactly ONE line, the result: libraries write there too (RCCL prints a five-line version banner on
    stdout when the first communicator is created), so the rank keeps the real stdout for ``emit`` and points descriptor 1 --
    whatever C or Python code prints from here on -- at stderr.  ``quiet`` (the default run): stderr itself is held back in a
    log file that ``release_stderr(failed=True)`` replays, so that a successful run leaves nothing behind the result line
    however the caller merges the two streams (round 4's driver record did not parse with progress lines after it)."""
    global _RESULT_FD, _QUIET_LOG
    if _RESULT_FD is None:
        sys.stdout.flush()
        sys.stderr.flush()
        _RESULT_FD = os.dup(1)
        if quiet and _QUIET_LOG is None:
            log = tempfile.TemporaryFile(mode='w+b')
            _QUIET_LOG = (log, os.dup(2))
            os.dup2(log.fileno(), 2)
        os.dup2(2, 1)


def release_stderr(failed):
    """give stderr back; replay what was held back if the run failed"""
    global _QUIET_LOG
    if _QUIET_LOG is None:
        return
    log, saved = _QUIET_LOG
    _QUIET_LOG = None
    sys.stderr.flush()
    os.dup2(saved, 2)
    os.close(saved)
    if failed:
        log.seek(0)
        os.write(2, log.read()[-20000:])
    log.close()


def _short(value, digits=6):
    """floats to ``digits`` significant digits (the line has a byte budget; the side file keeps full precision)"""
    if isinstance(value, float):
        return float(f'%.{digits}g' % value)
    if isinstance(value, dict):
        return {k: _short(v, digits) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_short(v, digits) for v in value]
    return value


def compact_line(full, extra_path=None):
    """The ONE stdout line out of the full record: the contract's keys only, sub-objects cut to theirs (module docstring)."""
    line = {k: full[k] for k in LINE_KEYS if k in full}
    if isinstance(line.get('roofline'), dict):
        line['roofline'] = {k: line['roofline'][k] for k in ROOFLINE_KEYS if k in line['roofline']}
    if isinstance(line.get('cpu_baseline'), dict):
        line['cpu_baseline'] = {k: line['cpu_baseline'].get(k) for k in CPU_KEYS}
    if isinstance(line.get('collective'), dict):
        c = line['collective']
        line['collective'] = {k: c[k] for k in COLLECTIVE_KEYS if k in c}
        if isinstance(c.get('gather_ms'), dict):
            line['collective']['gather_ms_p50'] = c['gather_ms'].get('p50')
    if extra_path:
        line['extra'] = extra_path
    line = _short(line)
    text = json.dumps(line, separators=(', ', ': '))
    if len(text) > LINE_LIMIT:       # never happens with the fields above; rather lose the optional scalars than the parse
        line.pop('also', None)
        text = json.dumps(line, separators=(', ', ': '))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def write_extra(full):
    """The full record to the side file.  -> the path as the line names it (relative to the repo when inside it), or None."""
    if not EXTRA_FILE:
        return None
    try:
        os.makedirs(os.path.dirname(EXTRA_FILE) or '.', exist_ok=True)
        with open(EXTRA_FILE, 'w') as f:
            json.dump(full, f, indent=1)
    except OSError:
        return None
    rel = os.path.relpath(EXTRA_FILE, REPO)
    return EXTRA_FILE if rel.startswith('..') else rel


def emit(full):
    """Side file first, then the ONE line, last thing this process prints."""
    text = compact_line(full, write_extra(full))
    sys.stderr.flush()
    data = (text + '\n').encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def progress(text):
    """--verbose: one line on stderr per phase (a long --extras run shows where it is).  Silent otherwise."""
    if VERBOSE:
        print(f'[bench {time.strftime("%H:%M:%S")}] {text}', file=sys.stderr, flush=True)


def _pkg():
    from simplenerf_amd import harness, ops, synth
    from simplenerf_amd.models.ModelFactory import get_model
    return harness, ops, synth, get_model


def synthetic_model(configs, seed, device, precision='fp32', fused=False):
    _, _, synth, get_model = _pkg()
    configs = synth.with_overrides(configs, hip_precision=precision, hip_fused_render=bool(fused))
    model = get_model(configs, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed, sigma_gain=200.0, sigma_shift=8.0).items()})
    return model.to(device).eval()


def cgroup_cpu_quota():
    """Cores the job's cgroup may use at once (cgroup v2 ``cpu.max``, v1 ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``), as a
    float, or None when there is no quota.  The GPU boxes report 256 CPUs and a 256-wide affinity mask but schedule the job on
    a 16-core share: this file is where that shows."""
    def read(path):
        try:
            with open(path) as f:
                return f.read().split()
        except OSError:
            return None
    v2 = read('/sys/fs/cgroup/cpu.max')
    if v2 and len(v2) == 2 and v2[0] != 'max':
        try:
            return float(v2[0]) / float(v2[1])
        except (ValueError, ZeroDivisionError):
            return None
    quota, period = read('/sys/fs/cgroup/cpu/cpu.cfs_quota_us'), read('/sys/fs/cgroup/cpu/cpu.cfs_period_us')
    if quota and period:
        try:
            q, p = float(quota[0]), float(period[0])
            return q / p if q > 0 and p > 0 else None
        except ValueError:
            return None
    return None


CPU_LEG_MAX_THREADS = 16           # the pool's per-GPU CPU share; also the cap where no quota is visible


def host_cores(affinity=None, quota='read', cap=CPU_LEG_MAX_THREADS):
    """Threads for the CPU leg = the cores this job OWNS: its affinity mask (os.cpu_count() where the platform has none),
    capped by the cgroup quota and by ``cap`` (a box that shows neither limit still schedules 16 cores per GPU; round 4's
    256-thread leg never finished a pass there).  The arguments exist for the test."""
    if affinity is None:
        try:
            affinity = len(os.sched_getaffinity(0))
        except AttributeError:
            affinity = os.cpu_count() or 1
    if quota == 'read':
        quota = cgroup_cpu_quota()
    n = max(1, int(affinity))
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    return min(n, cap) if cap else n


CPU_LEG_TIMEOUT_S = 60.0           # hard limit of the one CPU leg (about 15 s on an idle 16-core share)


def cpu_leg(kind, first_ray, threads, budget_seconds=9.0, max_runs=9):
    """One leg of ``cpu_baseline`` (runs in a child process: ``bench.py --cpu-leg THREADS``): the oracle on the host CPU with
    ``threads`` torch threads, same rays and weights as the GPU step; 1 warm-up + up to ``max_runs`` passes over the 1024 rays
    or ``budget_seconds`` of CPU work, whichever comes first."""
    from oracle import nerf_oracle, raygen_oracle
    _, _, synth, get_model = _pkg()
    configs = synth.make_configs(kind)
    camera = synth.camera('fern', 0)
    shapes = {k: tuple(v.shape) for k, v in get_model(configs, None).state_dict().items()}
    params = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, sigma_gain=200.0, sigma_shift=8.0).items()}
    full = raygen_oracle.full_frame_batch(camera['resolution'], camera['intrinsic'], camera['pose'], camera['near'],
                                          camera['far'], True, camera['near_ndc'], camera['far_ndc'])
    batch = {k: torch.from_numpy(numpy.ascontiguousarray(v[first_ray:first_ray + RAYS_PER_GPU])) for k, v in full.items()}
    torch.set_num_threads(threads)
    times = []
    with torch.no_grad():
        nerf_oracle.render(params, configs, batch, training=False)        # warm-up
        while len(times) < max_runs and (not times or sum(times) < budget_seconds):
            t0 = time.perf_counter()
            nerf_oracle.render(params, configs, batch, training=False)
            times.append(time.perf_counter() - t0)
            # (one line per finished pass: a parent that runs out of patience keeps what there is)
            print(json.dumps({'threads': threads, 'value': RAYS_PER_GPU / min(times), 'unit': 'rays/s', 'runs': len(times),
                              'best_s': min(times), 'mean_s': sum(times) / len(times)}), flush=True)
    return {'threads': threads, 'value': RAYS_PER_GPU / min(times), 'unit': 'rays/s', 'runs': len(times),
            'best_s': min(times), 'mean_s': sum(times) / len(times)}


def cpu_baseline(kind, first_ray, threads=None):
    """The oracle timed on this host, on the same 1024-ray batch as the GPU step: ONE bounded leg (a child process with a hard
    time limit that reports after every finished pass, so a slow host still yields a figure) with ``host_cores()`` threads --
    the cores the job owns, not the 256 the box advertises (BASELINE.md section 3 says os.cpu_count(); on this pool that leg
    oversubscribes a 16-core share and does not finish one pass in 45 s: DESIGN.md section 6)."""
    n = int(threads or host_cores())
    host = {'os_cpu_count': os.cpu_count(), 'cgroup_quota_cores': cgroup_cpu_quota(), 'threads_used': n}
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-leg', str(n), '--cpu-leg-args', kind, str(first_ray)]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(OMP_NUM_THREADS=str(n), MKL_NUM_THREADS=str(n))
    t0 = time.perf_counter()
    leg = None
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=CPU_LEG_TIMEOUT_S, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode == 0 and lines:
            leg = json.loads(lines[-1])
        else:
            host['failed'] = (r.stderr or r.stdout)[-300:]
    except subprocess.TimeoutExpired as late:
        so_far = late.stdout.decode() if isinstance(late.stdout, bytes) else (late.stdout or '')
        lines = [ln for ln in so_far.splitlines() if ln.startswith('{') and ln.rstrip().endswith('}')]
        leg = json.loads(lines[-1]) if lines else None       # the passes it did finish, if any
        host['timed_out_after_s'] = round(time.perf_counter() - t0, 1)
    if leg is None:
        return {'value': None, 'unit': 'rays/s', 'cores': n, 'kind': 'port', 'sample': 'no pass finished', 'host': host}
    return {'value': leg['value'], 'unit': 'rays/s', 'cores': n, 'kind': 'port',
            'sample': f"{RAYS_PER_GPU} rays of the headline batch, best of {leg['runs']} passes after 1 warm-up ({leg['best_s']:.2f} s best), "
                      f"torch CPU fp32, chunk 4096 / netchunk 16384",
            'leg': leg, 'host': host}


def pmc_traffic(precision):
    """HBM bytes per launch of the dominant kernel and where the figure comes from.  bench.py cannot read the memory
    counters of its own process (they need rocprofv3 around it): the figure is the one a separate ``rocprofv3 --pmc`` run of
    this same command wrote to profiles/ (tools/pmc_passes.sh + tools/collect_pmc.py; FETCH_SIZE | WRITE_SIZE with the guide's
    gfx950 corrections), named in ``traffic_source`` with the commit it was collected at.  -> (bytes or None, source or None)"""
    name = 'pmc_traffic.json' if precision == 'fp32' else f'pmc_traffic_{precision}.json'
    path = os.path.join(REPO, 'profiles', name)
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        record = json.load(f)
    source = f"profiles/{name} @ {record.get('commit', 'unrecorded')} (separate rocprofv3 --pmc passes, not this run)"
    return record.get('mlp_forward_hbm_bytes_per_launch'), source


def pmc_train_traffic(precision):
    """HBM bytes per training iteration (config 5) from a separate rocprofv3 --pmc run of `bench.py --train` (profiles/
    pmc_traffic_train_<precision>.json: tools/pmc_passes.sh + tools/collect_pmc_kernels.py --iterations N) -- in the
    16-bit modes the weight gradients stream at the HBM rate and this figure / 6.3 TB/s is the floor under the iteration.  -> dict or None"""
    name = f'pmc_traffic_train_{precision}.json'
    path = os.path.join(REPO, 'profiles', name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        record = json.load(f)
    if 'hbm_gb_per_iteration' not in record:
        return None
    return {'hbm_gb_per_iteration': record['hbm_gb_per_iteration'],
            'source': f"profiles/{name} @ {record.get('commit', 'unrecorded')} (separate rocprofv3 --pmc passes of `bench.py --train --precision "
                      f"{precision}`, all kernels, not this run)"}


# ---------------------------------------------------------------------------------------------- timing protocol
class _Mark:
    """A point on the launch stream: a HIP event on torch's current stream (the stream every library call of this
    process is enqueued on), or a host clock reading for the GPU-less stand-in."""

    def __init__(self, gpu: bool):
        self.event = torch.cuda.Event(enable_timing=True) if gpu else None
        self.host = 0.0

    def record(self):
        if self.event is not None:
            self.event.record()
        self.host = time.perf_counter()

    def ms_until(self, later: '_Mark') -> float:
        if self.event is not None:
            return self.event.elapsed_time(later.event)
        return (later.host - self.host) * 1e3


def settle(run, seconds=None, gpu=True, chunk=8):
    """Untimed: ``run`` (the step's device work, no collective: ranks run different counts) back to back for at least
    ``seconds`` (default: SETTLE_SECONDS).  -> (runs, seconds spent)"""
    seconds = SETTLE_SECONDS if seconds is None else seconds
    t0 = time.perf_counter()
    runs = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(chunk):
            run()
        runs += chunk
        if gpu:
            torch.cuda.synchronize()
    return runs, time.perf_counter() - t0


def timed_steps(step, steps, fence, gpu=True):
    """EXACTLY ``steps`` steps bracketed by ``fence()`` (barrier + synchronize) on both sides.  -> (elapsed seconds over
    the whole region, per-step device milliseconds between consecutive end-of-step marks, per-step host enqueue ms)."""
    marks = [_Mark(gpu) for _ in range(steps + 1)]
    # nothing may run for the first time inside the region (on a fresh lease a first call can fault its code in from a
    # lazily loaded image): one throw-away pair of marks is recorded and read before the opening fence
    dry = [_Mark(gpu), _Mark(gpu)]
    dry[0].record()
    dry[1].record()
    fence()
    dry[0].ms_until(dry[1])
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    device_ms = [marks[i].ms_until(marks[i + 1]) for i in range(steps)]
    enqueue_ms = [(marks[i + 1].host - marks[i].host) * 1e3 for i in range(steps)]
    return elapsed, device_ms, enqueue_ms


def _quantiles(values):
    s = sorted(values)
    pick = lambda q: s[min(len(s) - 1, int(round(q * (len(s) - 1))))]
    return {'p50': pick(0.5), 'p90': pick(0.9), 'max': s[-1], 'first': values[0], 'argmax': values.index(s[-1])}


def step_summary(elapsed, device_ms, enqueue_ms, kernel_ms_total=None):
    """The attribution fields of a timed region (see the module docstring)."""
    steps = len(device_ms)
    out = {'step_ms': _quantiles(device_ms), 'enqueue_ms': _quantiles(enqueue_ms),
           'idle_ms_per_step': None if kernel_ms_total is None else (elapsed * 1e3 - kernel_ms_total) / steps}
    if steps <= 64:
        out['step_trace_ms'] = [round(v, 4) for v in device_ms]
    return out


def headline_line(world, steps, warmup, precision, elapsed, device_ms, enqueue_ms, launch_ms, launch_samples,
                  dropped=0, settle_info=None, data='synthetic'):
    """The bench line of the headline metric from one timed region's measurements (pure: tests build it from fake
    numbers).  `value` = all rays of the K steps / the whole elapsed time -- no step is left out."""
    peak, dtype, kernel_name, note = PRECISION_INFO[precision]
    kernel_ms = float(sum(launch_ms))
    line = {
        'metric': METRIC,
        'value': world * RAYS_PER_GPU * steps / elapsed,
        'unit': 'rays/s',
        'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': elapsed / steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': dtype, 'data': data,
        'config': {'workload': WORKLOAD, 'rays_per_gpu': RAYS_PER_GPU, 'samples': '128+128',
                   'parallelism': f'ray-shard x{world}' + (' + 1 gather/step' if world > 1 else '')},
    }
    if launch_ms:
        achieved = sum(launch_samples) * FLOP_PER_SAMPLE / (kernel_ms * 1e-3) / 1e12
        line['roofline'] = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                            'traffic': pmc_traffic(precision)[0], 'traffic_source': pmc_traffic(precision)[1],
                            'traffic_algorithmic': sum(launch_samples) * 20 / len(launch_ms) + 595844 * 4, 'kernel': kernel_name, 'note': note, 'launches': len(launch_ms),
                            'avg_launch_ms': kernel_ms / len(launch_ms), 'kernel_share_of_step': kernel_ms / (elapsed * 1e3),
                            'launches_not_timed': int(dropped),
                            'step_frac': world * RAYS_PER_GPU * steps * (128 + 256) * FLOP_PER_SAMPLE / elapsed / 1e12 / peak / world}
    line['timing'] = step_summary(elapsed, device_ms, enqueue_ms, kernel_ms if launch_ms else None)
    if settle_info is not None:
        line['timing']['settle'] = {'runs': settle_info[0], 'seconds': settle_info[1]}
    return line


def max_over_ranks(value, dist, device):
    """the contract's max-over-ranks of a scalar (one all-reduce; the tensor lives on the rank's GPU under RCCL)"""
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def per_rank_table(values, dist, device):
    """``values`` (a short list of floats) of every rank, as a list of lists indexed by rank (one all-gather)."""
    if dist is None:
        return [list(map(float, values))]
    where = device if dist.get_backend() == 'nccl' else None         # gloo gathers host tensors
    mine = torch.tensor(values, dtype=torch.float64, device=where)
    table = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(table, mine)
    return [[float(v) for v in row.cpu()] for row in table]


# ---------------------------------------------------------------------------------------------- renderers
class HipRenderer:
    """The product path for one rank: the drop-in model on this rank's GPU."""
    gpu = True

    def __init__(self, precision, device, rank, world, kind='headline', collective=None, fused=False):
        harness, ops, synth, _ = _pkg()
        self.harness, self.ops, self.synth = harness, ops, synth
        self.precision, self.device, self.rank, self.world = precision, device, rank, world
        self.fused = bool(fused)         # configs['model']['hip_fused_render']: the render as one launch (csrc/render_fused.hip)
        self.collective = world > 1 if collective is None else collective      # True with one rank under --force-collective
        self.gather_marks = None         # [(before, after)] per step while the gather is being timed
        self.configs = synth.with_overrides(synth.make_configs(kind), hip_precision=precision)
        self.model = synthetic_model(synth.make_configs(kind), 7, device, precision, fused)
        self.camera = synth.camera('fern', 0)
        h, w = self.camera['resolution']
        # rank r renders pixels [base + r*1024, base + (r+1)*1024) from the middle of the frame
        self.first = (h // 2) * w + rank * RAYS_PER_GPU

    def local(self):
        batch = self.harness.frame_batch(self.camera, True, self.device, self.first, RAYS_PER_GPU)
        out = self.model(batch)
        return {'rgb_fine': out['rgb_fine'], 'depth_fine': out['depth_fine']}

    def step(self):
        local = self.local()
        if not self.collective:
            return local
        if self.gather_marks is None:
            return self.harness.gather_rays(local, self.world * RAYS_PER_GPU, self.rank, self.world)
        # the gather's own time: events on the launch stream either side of the call (torch.distributed makes this stream
        # wait for the collective's, so the second event fires when the gathered data may be used)
        before, after = _Mark(self.gpu), _Mark(self.gpu)
        before.record()
        full = self.harness.gather_rays(local, self.world * RAYS_PER_GPU, self.rank, self.world)
        after.record()
        self.gather_marks.append((before, after))
        return full

    def frame_camera(self, name):
        scene, kwargs, _ = FRAMES[name]
        return self.synth.camera(scene, 0, **kwargs)

    def frame(self, name):
        """Tester.predict_frame for this rank's block of the frame; the five display outputs as host arrays on rank 0."""
        return self.harness.predict_frame(self.model, self.configs, self.frame_camera(name), self.device, self.rank, self.world,
                                          collective=self.collective)

    def frame_block(self, name, rays=65536):
        cam = self.frame_camera(name)
        self.model(self.harness.frame_batch(cam, True, self.device, 0, min(rays, cam['resolution'][0] * cam['resolution'][1])))

    def profile(self, capacity):
        self.ops.profile_enable(capacity)

    def profile_reset(self):
        self.ops.profile_reset()

    def profile_collect(self):
        ms, samples = self.ops.profile_collect(self.ops.PROFILE_MLP_FORWARD)
        return ms, samples, self.ops.profile_dropped()


# ---------------------------------------------------------------------------------------------- config 5 (training)
def training_step(precision, rank, world, device, single_pass=False, graphed=False, collective=False, rows_per_gpu=4096, graph_scope=None):
    """BASELINE config 5: a callable running ONE iteration of the reference's training loop (Trainer.train_one_iter,
    src/Trainer01.py:60-107) with every stage on the device: batch assembly (pixel rows + as many sparse-depth rows, each rank
    a slice of one global index stream), four MLPs forward, nine losses, backward, ONE all-reduce of the flattened gradients
    (RCCL) for N > 1, Adam with the decayed rate.  ``rows_per_gpu`` = 4096: weak scaling, the reference's whole batch on every
    GPU.  ``rows_per_gpu`` = 4096 / N: BASELINE config 5 as stated -- ONE 4096-row batch over the N ranks (8 ranks x
    512 rows); the reference cuts its batch into two 2048-row sub-batches BEFORE DataParallel scatters each over the devices
    (Trainer01.py:82-93), so a rank's share is processed as two sub-batches of rows_per_gpu / 2."""
    harness, _, synth, get_model = _pkg()
    from simplenerf_amd import optim
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer
    if rows_per_gpu % 2 or rows_per_gpu <= 0:
        raise SystemExit('--rows-per-gpu: an even number of rows (half pixel rows, half sparse-depth rows)')
    rows = rows_per_gpu // 2
    cfg = synth.training_configs(precision, num_rays=rows * world, num_sparse=rows * world)
    cfg['sub_batch_size'] = rows          # this rank's share of each of the reference's two sub-batches
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    model = model.to(device).train()
    # the sparse-depth epoch is a whole number of global batches (largest such count <= 1.5 M of the 2.29 M pixels), so that
    # every iteration really has 2048 sparse rows per GPU.  (Until round 3 the scene held ~4 570 sparse points: every third
    # iteration ran a 476-row sparse batch and the reported mean -- 8.39 ms in the 16-bit mode -- understated the full-size
    # iteration, 9.3 ms, by 10 %.  `short_batches` in the line counts iterations of the timed region that were not full.)
    sparse_points = (1500000 // (rows * world)) * rows * world
    batcher = BatchAssembler(cfg, synth.training_scene(sparse_points=sparse_points), device, rank=rank, world_size=world)
    losses = LossComputer(cfg)
    opt = optim.Adam(list(model.parameters()), lr=cfg['optimizer']['lr_initial'],
                     betas=(cfg['optimizer']['beta1'], cfg['optimizer']['beta2']))
    decayer = get_lr_decayer(cfg)
    state = {'iter': 20000}

    # --graphed.  scope 'iteration' (default for one rank): the WHOLE iteration -- batch assembly, draws, pass, all-reduce, Adam --
    # replayed from ONE HIP graph (harness.GraphedIteration).  scope 'pass' (default for N > 1): the model pass -- re-pack, forwards,
    # losses, backward -- from one graph (harness.GraphedTrainStep), batch assembly / gradient all-reduce / Adam enqueued around it:
    # no collective inside a capture, so it runs on any backend and any number of ranks (a captured all-reduce has only ever run with
    # one rank: ADVICE r4), and the host still saves the ~100 launches of the pass -- which is what bounds a small per-rank share.
    graph, pass_graph = None, None
    scope = (graph_scope or ('iteration' if world == 1 else 'pass')) if graphed else None
    if scope == 'iteration':
        if world > 1:
            raise SystemExit("--graph-scope iteration captures the gradient all-reduce, which has only been exercised with one rank; "
                             "N > 1 runs --graph-scope pass")
        # (with single_pass: ONE model pass, i.e. the reference's loop with sub_batch_size = batch size -- Trainer01.py:82)
        graph = harness.GraphedIteration(model, losses, opt, batcher, decayer, sub_batch_size=None if single_pass else cfg['sub_batch_size'],
                                         force_collective=collective)

    def step():
        it = state['iter']
        state['iter'] += 1
        if graph is not None:
            out = graph(it)
            step.short_batches += int(graph.last_was_short)
            return out
        for group in opt.param_groups:
            group['lr'] = decayer.get_updated_learning_rate(it)
        batch = batcher.get_next_batch(it)
        step.short_batches += int(batch['rays_o'].shape[0] != 2 * rows)
        if scope == 'pass':
            if step.pass_graph is None:       # captured on the first (full) batch; a short batch at an epoch's end runs eagerly inside it
                step.pass_graph = harness.GraphedTrainStep(model, losses, batch, sub_batch_size=None if single_pass else cfg['sub_batch_size'])
            totals = step.pass_graph(batch)
            harness.allreduce_gradients(model.parameters(), world, force=collective)
            opt.step()
            return totals
        return harness.train_one_iter(model, losses, opt, batch, cfg['sub_batch_size'], world, single_pass=single_pass,
                                      force_collective=collective)

    step.pass_graph = pass_graph
    step.short_batches = 0
    return step, 2 * rows


TRAIN_WORKLOAD = ('config 5: 2048 pixel + 2048 sparse-depth rows per GPU in two sub-batches, main coarse+fine + points-aug + '
                  'views-aug MLPs (64 + 192 samples), nine shipped losses, Adam, NeRF LR decay')


def train_workload(rows_per_gpu, world, strong):
    half = rows_per_gpu // 2
    if strong:
        return (f'config 5 as BASELINE states it: ONE {rows_per_gpu * world}-row batch over {world} rank(s) = {half} pixel + {half} '
                'sparse-depth rows per GPU in two sub-batches, 4 MLPs (64 + 192 samples), nine losses, one gradient all-reduce, Adam')
    return TRAIN_WORKLOAD.replace('2048 pixel + 2048', f'{half} pixel + {half}')
TRAIN_DTYPE = {'fp32': 'f32', 'f16x3': 'f16x3', 'f16': 'f16 (bf16 layer gradients)', 'bf16': 'bf16',
               'f16s8': 'f16 (bf16 layer gradients, fp8 e4m3 saved trunk activations)',
               'bf16s8': 'bf16 (fp8 e4m3 saved trunk activations)'}


def train_rows_per_gpu(args, world):
    """rows of the training batch one GPU holds: --rows-per-gpu R, or --global-rows G (strong scaling: G / N), else 4096"""
    if args.global_rows:
        if args.global_rows % (2 * world):
            raise SystemExit(f'--global-rows {args.global_rows}: not a multiple of 2 x {world} ranks')
        return args.global_rows // world, True
    return (args.rows_per_gpu or 4096), False


def train_bench(args, rank, world, device, dist):
    """--train: BASELINE config 5 instead of the headline metric (see training_step).  Default: weak scaling, 4096 rows per
    GPU.  ``--global-rows 4096``: the config as BASELINE states it -- ONE 4096-row batch over the N ranks (`scaling: strong`);
    ``--rows-per-gpu 512 --force-collective`` on one GPU is what one rank of eight then does, all-reduce included."""
    per_gpu, strong = train_rows_per_gpu(args, world)
    step, per_gpu = training_step(args.precision, rank, world, device, args.single_pass, graphed=args.graphed,
                                  collective=dist is not None, rows_per_gpu=per_gpu, graph_scope=args.graph_scope)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):       # (the all-reduce keeps the ranks in step: the settle phase counts steps here)
        step()
    fence()
    if world == 1:
        settle(step, None, chunk=2)
    step.short_batches = 0
    elapsed, device_ms, enqueue_ms = timed_steps(step, args.steps, fence)
    ranks = per_rank_table([_quantiles(device_ms)['p50'], elapsed], dist, device)
    elapsed = max_over_ranks(elapsed, dist, device)
    if rank == 0:
        line = {
            'metric': 'training rays/sec (config 5: forward + backward + optimiser, 4 MLPs, 9 losses)',
            'value': per_gpu * world * args.steps / elapsed, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None, 'dtype': TRAIN_DTYPE[args.precision], 'data': 'synthetic',
            'config': {'workload': train_workload(per_gpu, world, strong), 'rows_per_gpu': per_gpu, 'global_rows': per_gpu * world,
                       'single_pass': bool(args.single_pass),
                       'graphed': (args.graph_scope or ('iteration' if world == 1 else 'pass')) if args.graphed else False,
                       'parallelism': f'row-shard x{world}' + (' + 1 gradient all-reduce/step' if dist is not None else '')},
            'algorithmic_tflops': per_gpu * TRAIN_FLOP_PER_RAY * world * args.steps / elapsed / 1e12,
            'timing': step_summary(elapsed, device_ms, enqueue_ms, None)}
        line['timing']['short_batches'] = step.short_batches   # timed iterations with fewer than rows_per_gpu rows (epoch ends)
        traffic = pmc_train_traffic(args.precision) if per_gpu == 4096 else None
        line['roofline'] = {'bound': 'mfma',      # (what `achieved` / `peak` are; the HBM side of the iteration is `traffic`)
                            'achieved': line['algorithmic_tflops'] / world,
                            'peak': {'fp32': PEAK_FP32_MFMA_TFLOPS, 'f16x3': PEAK_FP16_MFMA_TFLOPS / 3}.get(args.precision, PEAK_FP16_MFMA_TFLOPS),
                            'unit': 'TFLOP/s', 'traffic': None if traffic is None else traffic['hbm_gb_per_iteration'] * 1e9,
                            'traffic_source': None if traffic is None else traffic['source'],
                            'kernel': 'whole iteration (all kernels)',
                            'note': 'whole-iteration wall time against the MFMA peak of the mode; `traffic` = HBM bytes per iteration, '
                                    'all kernels: / 6.3 TB/s it is the HBM floor (DESIGN.md says what bounds each kernel above it)'}
        line['roofline']['frac'] = line['roofline']['achieved'] / line['roofline']['peak']
        if dist is not None:
            line['collective'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size(), 'bytes': 2265488 * 4,
                                  'pattern': 'one all-reduce of the flattened parameter gradients per iteration',
                                  'per_rank': [{'rank': i, 'step_ms_p50': r[0], 'elapsed_s': r[1]} for i, r in enumerate(ranks)]}
        emit(line)


# ---------------------------------------------------------------------------------------------- whole frames (configs 2, 4)
def time_frames(renderer, name, frames, warmup, fence, world, dist=None, device=None):
    """``frames`` whole frames through Tester.predict_frame's path on this rank's share of the pixels, each rank's wall
    time fenced on both sides; -> (max-over-ranks seconds for all frames, rays per frame, last frame's outputs on rank 0)."""
    cam = renderer.frame_camera(name)
    rays = int(cam['resolution'][0]) * int(cam['resolution'][1])
    if world == 1:
        settle(lambda: renderer.frame_block(name), None, renderer.gpu, chunk=1)
    out = None
    for _ in range(warmup):
        out = renderer.frame(name)
    fence()
    t0 = time.perf_counter()
    for _ in range(frames):
        out = renderer.frame(name)
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, device)
    return elapsed, rays, out


def frame_entry(name, precision, elapsed, rays, frames, world):
    peak = PRECISION_INFO[precision][0]
    tflops = rays * FRAME_SAMPLES * FLOP_PER_SAMPLE * frames / elapsed / 1e12
    return {'frame': name, 'workload': FRAMES[name][2], 'rays': rays, 'samples': '64+128 (192 merged)', 'precision': precision,
            'frames': frames, 'ms_per_frame': elapsed / frames * 1e3, 'value': rays * frames / elapsed, 'unit': 'rays/s',
            'algorithmic_tflops': tflops, 'peak_tflops': peak * world, 'frac_of_peak': tflops / (peak * world)}


FRAME_PATH = ('on-device ray generation per 65536-ray block -> coarse+fine render -> display conversion (uint8 colour, '
              'clipped depths) -> D2H of the five display outputs (Tester.predict_frame, src/Tester01.py:57-66); wall time')


def frame_bench(args, rank, world, device, dist, make_renderer, data):
    """--frame NAME: the frame is the unit of work (BASELINE config 4: 'full-frame render, ray-batch shard across 8 GPUs +
    RCCL gather').  STRONG scaling: the frame's rays are block-sharded over the N ranks (harness.shard_range), every rank
    generates and renders its own block, rank 0 receives the frame through one gather, converts and copies it to the host.
    `value` = frame rays x K frames / max-over-ranks wall time."""
    def fence():
        if dist is not None:
            dist.barrier()
        if data == 'synthetic':
            torch.cuda.synchronize()

    renderer = make_renderer(args.precision, 'config2')
    elapsed, rays, out = time_frames(renderer, args.frame, args.steps, args.warmup, fence, world, dist, device)
    if rank != 0:
        return
    assert out is not None and all(numpy.asarray(v).size > 0 for v in out.values())
    entry = frame_entry(args.frame, args.precision, elapsed, rays, args.steps, world)
    per = -(-rays // world)
    line = {'metric': 'rays/sec, full frame (64+128 samples, 8x256 coarse+fine) incl. raygen, gather and display output',
            'value': entry['value'], 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': entry['ms_per_frame'], 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': PRECISION_INFO[args.precision][1] if data == 'synthetic' else 'none', 'data': data,
            'config': {'workload': FRAMES[args.frame][2] + ': one frame per step, ' + FRAME_PATH, 'rays_per_frame': rays,
                       'rays_per_gpu': per, 'samples': '64+128',
                       'parallelism': f'ray-shard x{world}' + (' + 1 gather/frame' if dist is not None else '')},
            'roofline': {'bound': 'mfma', 'achieved': entry['algorithmic_tflops'], 'peak': entry['peak_tflops'], 'unit': 'TFLOP/s',
                         'frac': entry['frac_of_peak'], 'traffic': None,
                         'note': 'whole-frame wall time (not a kernel time): algorithmic FLOPs of the frame / wall, against N x the per-GPU peak'}}
    if dist is not None:
        line['collective'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                              'bytes': per * FRAME_GATHER_BYTES, 'pattern': 'one gather of the five per-ray outputs to rank 0 per frame'}
    emit(line)


# ---------------------------------------------------------------------------------------------- headline
def measure_headline(renderer, steps, warmup, fence, world, do_settle=True):
    """settle -> warm-up (event hooks on) -> timed region.  -> dict of raw measurements for ``headline_line``."""
    with torch.no_grad():
        settle_info = settle(renderer.local, None, renderer.gpu) if do_settle else None
        renderer.profile(4 * (steps + warmup) + 16)
        for _ in range(warmup):
            renderer.step()
        fence()
        renderer.profile_reset()
        if renderer.collective:
            renderer.gather_marks = []          # from here on every step brackets its gather with two events
        elapsed, device_ms, enqueue_ms = timed_steps(renderer.step, steps, fence, renderer.gpu)
    gather_ms = [a.ms_until(b) for a, b in (renderer.gather_marks or [])]
    renderer.gather_marks = None
    launch_ms, launch_samples, dropped = renderer.profile_collect()
    renderer.profile(0)
    return {'elapsed': elapsed, 'device_ms': device_ms, 'enqueue_ms': enqueue_ms, 'launch_ms': launch_ms,
            'launch_samples': launch_samples, 'dropped': dropped, 'settle_info': settle_info, 'gather_ms': gather_ms}


def render_bench(args, rank, world, device, dist, make_renderer, data):
    gpu = data == 'synthetic'

    collective = dist is not None          # N > 1, or N = 1 under --force-collective

    def fence():
        if collective:
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    renderer = make_renderer(args.precision, 'headline')
    m = measure_headline(renderer, args.steps, args.warmup, fence, world)
    if rank == 0:
        progress(f'headline timed: {m["elapsed"] / args.steps * 1e3:.3f} ms/step')
    ranks = None
    if collective:
        # what each rank saw, so that a sub-linear point can be attributed: its own step time, the MLP kernels' share of
        # it, the gather's own time, its elapsed seconds (the line's value uses the maximum)
        g = _quantiles(m['gather_ms']) if m['gather_ms'] else {'p50': 0.0, 'max': 0.0}
        ranks = per_rank_table([_quantiles(m['device_ms'])['p50'], sum(m['launch_ms']) / max(args.steps, 1), g['p50'], g['max'],
                                m['elapsed']], dist, device)
        m['elapsed'] = max_over_ranks(m['elapsed'], dist, device)
    result = None
    if rank == 0:
        result = headline_line(world, args.steps, args.warmup, args.precision, m['elapsed'], m['device_ms'], m['enqueue_ms'],
                               m['launch_ms'], m['launch_samples'], m['dropped'], m['settle_info'], data)
        if not gpu:
            result['dtype'] = 'none'
            result['config'] = {'workload': 'launcher rehearsal, no renderer'}
        if collective:
            result['collective'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                                    'bytes': RAYS_PER_GPU * 16, 'pattern': 'one gather of (rgb, depth) = 16 B/ray to rank 0 per step',
                                    'gather_ms': {'p50': max(r[2] for r in ranks), 'max': max(r[3] for r in ranks),
                                                  'how': 'events either side of the gather on each rank, worst rank'},
                                    'per_rank': [{'rank': i, 'step_ms_p50': r[0], 'mlp_kernel_ms_per_step': r[1],
                                                  'gather_ms_p50': r[2], 'gather_ms_max': r[3], 'elapsed_s': r[4]}
                                                 for i, r in enumerate(ranks)]}
            result['config']['parallelism'] = f'ray-shard x{world} + 1 gather/step'
    alt = args.precision == 'fp32' and not args.no_alt
    if collective and alt:
        frames = secondary().frame_records_sharded(make_renderer, fence, rank, world, dist, device)     # every rank takes part
        if rank == 0:
            result['also_measured_frame'] = frames
    if rank != 0:
        return
    if not collective and alt and gpu:
        result['also'] = secondary().quick_also(args, device, make_renderer, fence, result)
        progress('secondary scalars done')
        if args.extras:
            secondary().long_extras(args, rank, world, device, make_renderer, fence, renderer, result)
    if world == 1 and gpu and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline('headline', renderer.first, args.cpu_threads)
        progress('cpu baseline done')
    emit(result)


def secondary():
    """bench_secondary.py (the `also` scalars, the --extras set, the sharded frame, the board sampler), bound to THIS module --
    which is `__main__` when bench.py runs as a script and `bench` when it is imported."""
    import bench_secondary
    bench_secondary.bind(_THIS)
    return bench_secondary


class _ThisModule:
    """This module's live globals as attributes (sys.modules need not hold it: tests load bench.py under other names)."""

    def __getattr__(self, name):
        try:
            return globals()[name]
        except KeyError:
            raise AttributeError(name) from None


_THIS = _ThisModule()


_SECONDARY_NAMES = ('BoardSampler', 'time_training', 'training_record', 'rank_share_record', 'one_rank_group', 'frame_records_single',
                    'frame_records_sharded', 'quick_also', 'long_extras')


def __getattr__(name):            # bench.BoardSampler, bench.time_training ...: what moved to bench_secondary.py stays reachable
    if name in _SECONDARY_NAMES:
        return getattr(secondary(), name)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')


def visible_gpus():
    """GPUs this process can use (does not initialise HIP on this image)."""
    return torch.cuda.device_count()


def main(argv=None, renderer_cls=None, backend='nccl', share_devices=False, script=None):
    """``renderer_cls`` / ``backend`` / ``share_devices`` / ``script`` are for tests/bench_rehearsal.py only (a GPU-less
    stand-in renderer, gloo instead of RCCL, several ranks on one GPU, and the file the self-launcher starts its ranks from);
    `python bench.py` always runs HipRenderer over RCCL with one rank per GPU."""
    global SETTLE_SECONDS, VERBOSE, EXTRA_FILE
    ap = argparse.ArgumentParser()
    ap.add_argument('--train', action='store_true',
                    help='measure BASELINE config 5 (training iteration) instead of the headline render metric')
    ap.add_argument('--frame', choices=sorted(FRAMES), default=None,
                    help='measure whole frames of that camera, strong-scaled over the ranks (BASELINE config 2 / 4), instead '
                         'of the headline step; --steps counts frames (default 5)')
    ap.add_argument('--single-pass', action='store_true',
                    help='--train: one model forward/backward over the whole batch, losses still normalised per sub-batch '
                         '(harness.train_one_iter single_pass)')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default 50; 5 frames with --frame)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default 5; 1 frame with --frame)')
    ap.add_argument('--settle-seconds', type=float, default=None,
                    help=f'untimed settle phase before the warm-up steps (default {SETTLE_SECONDS}; 0 reproduces the cold start)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the secondary measurements (other precisions, frames, training, sustained)')
    ap.add_argument('--force-collective', action='store_true',
                    help='run the N > 1 protocol -- process group, barriers, the per-step gather (or gradient all-reduce), the '
                         'max-over-ranks reduction -- with whatever N is, including 1: every RCCL call of the multi-GPU line on a '
                         'one-GPU box')
    ap.add_argument('--precision', choices=('fp32', 'f16x3', 'f16', 'bf16', 'f16s8', 'bf16s8'), default='fp32',
                    help="arithmetic of the fused MLP kernel: fp32 MFMA; fp16 hi/lo split with 3 MFMAs per product "
                         "(fp32-grade results: same rendering parity tests, looser class for training gradients); or f16 = one fp16 MFMA per product with 16-bit saved "
                         "tensors (BASELINE config 5's 16-bit training mode, own tolerances: tests/test_gpu_f16.py)")
    ap.add_argument('--extras', action='store_true',
                    help='N = 1: also run the long secondary set (sustained run with board power / clock, the other precisions with '
                         'their boards, the fused render kernel, whole frames of configs 2 and 4, config 5 in every precision and '
                         'issue mode, the rank share of the strong-scaled iteration) -- into the side file, never onto the line')
    ap.add_argument('--extra-file', default=None,
                    help='side file for the full record (default: gpurun_out/bench_extra.json under the repo)')
    ap.add_argument('--verbose', action='store_true', help='progress lines and library chatter on stderr (default: none)')
    ap.add_argument('--cpu-threads', type=int, default=None, help='threads of the CPU baseline leg (default: the cores this job owns)')
    ap.add_argument('--global-rows', type=int, default=None,
                    help='--train: ONE batch of that many rows over the N ranks (BASELINE config 5 as stated: 4096; scaling: strong)')
    ap.add_argument('--rows-per-gpu', type=int, default=None,
                    help="--train: rows of the batch per GPU (default 4096); 512 with --force-collective = one rank's share of eight")
    ap.add_argument('--graphed', action='store_true', help='--train: replay the iteration from a HIP graph (see --graph-scope)')
    ap.add_argument('--graph-scope', choices=('iteration', 'pass'), default=None,
                    help="--train --graphed: 'iteration' = batch assembly, draws, pass, all-reduce and Adam in ONE graph (default for N = 1; "
                         "one rank only); 'pass' = the model pass in one graph, batch assembly / all-reduce / Adam around it (default for N > 1)")
    ap.add_argument('--cpu-leg', type=int, default=None, help=argparse.SUPPRESS)          # child of cpu_baseline: THREADS
    ap.add_argument('--cpu-leg-args', nargs=2, default=None, help=argparse.SUPPRESS)       # workload kind, first ray
    args = ap.parse_args(argv)
    if args.cpu_leg is not None:           # no GPU, no process group: one CPU leg, one JSON line
        print(json.dumps(cpu_leg(args.cpu_leg_args[0], int(args.cpu_leg_args[1]), args.cpu_leg)), flush=True)
        return
    VERBOSE = bool(args.verbose)
    EXTRA_FILE = os.path.abspath(args.extra_file) if args.extra_file else os.path.join(REPO, 'gpurun_out', 'bench_extra.json')
    if args.settle_seconds is not None:
        SETTLE_SECONDS = max(0.0, args.settle_seconds)
    if args.steps is None:
        args.steps = 5 if args.frame else 50
    if args.warmup is None:
        args.warmup = 1 if args.frame else 5
    standin = renderer_cls is not None

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        if backend == 'nccl' and not share_devices and args.gpus > visible_gpus():
            raise SystemExit(f'--gpus {args.gpus}: this node has {visible_gpus()} GPU(s); RCCL needs one device per rank')
        raise SystemExit(launch_ranks(args.gpus, script))       # nothing above this line touches the GPU

    claim_stdout(quiet=not VERBOSE)
    failed = True
    try:
        _run_rank(args, renderer_cls, backend, share_devices, standin)
        failed = False
    finally:
        release_stderr(failed)


def _run_rank(args, renderer_cls, backend, share_devices, standin):
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher set WORLD_SIZE={world}')
    dist = None
    device = None
    if not standin:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: the HIP renderer has no CPU path')
        count = torch.cuda.device_count()
        if share_devices:
            local_rank = local_rank % count         # rehearsal only: several ranks on the one GPU of a test box (gloo)
        elif world > count or local_rank >= count:
            # one rank per GPU, no wrap-around: two RCCL ranks on one device fail late and obscurely (and two ranks sharing a
            # GPU would halve each other's throughput silently with any other backend)
            raise SystemExit(f'rank {rank}: LOCAL_RANK={local_rank}, WORLD_SIZE={world} but this node shows {count} GPU(s); '
                             f'bench.py runs one rank per GPU')
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:       # --force-collective without a launcher: a one-rank group of our own
            with socket.socket() as sock:
                sock.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
        if backend == 'nccl':                     # RCCL, bound to this rank's device
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def make_renderer(precision, kind):
        return (renderer_cls or HipRenderer)(precision, device, rank, world, kind, collective=dist is not None)

    data = 'stand-in' if standin else 'synthetic'
    try:
        if args.train:
            if standin:
                raise SystemExit('--train has no stand-in')
            train_bench(args, rank, world, device, dist)
        elif args.frame:
            frame_bench(args, rank, world, device, dist, make_renderer, data)
        else:
            render_bench(args, rank, world, device, dist, make_renderer, data)
    finally:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
