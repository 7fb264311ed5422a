"""world_size-2 CPU (gloo) check of the multi-GPU path's host logic: block partition of the rays and the single
gather that reassembles the frame on rank 0.  The per-rank renderer is replaced by a deterministic stand-in (the HIP
kernels need a GPU); what is tested is exactly the code bench.py / harness.render_frame run around them."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simplenerf_amd import harness


def test_shard_range_covers_every_ray_once():
    for n in (0, 1, 7, 1024, 762048, 190512):
        for world in (1, 2, 3, 8):
            spans = [harness.shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for first, count in spans:
                assert first == pos or count == 0
                pos += count
            assert max(c for _, c in spans) - min(c for _, c in spans) <= -(-n // world)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        first, count = harness.shard_range(n, rank, world)
        idx = torch.arange(first, first + count, dtype=torch.float32)
        local = {'rgb_fine': torch.stack([idx, idx * 2, idx * 3], 1), 'depth_fine': idx + 0.5}
        full = harness.gather_rays(local, n, rank, world)
        if rank == 0:
            ref = torch.arange(n, dtype=torch.float32)
            assert torch.equal(full['depth_fine'], ref + 0.5)
            assert torch.equal(full['rgb_fine'], torch.stack([ref, ref * 2, ref * 3], 1))
            open(tmp, 'w').write('ok')
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [1000, 1001, 3])
def test_two_rank_gather_reassembles_frame(tmp_path, n):
    marker = str(tmp_path / 'done')
    mp.spawn(_worker, args=(2, _free_port(), n, marker), nprocs=2, join=True)
    assert open(marker).read() == 'ok'


def _grad_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10
        shard = x[rank * 4:(rank + 1) * 4]
        (net(shard) ** 2).mean().backward()
        harness.allreduce_gradients(net.parameters(), world)
        got = [p.grad.clone() for p in net.parameters()]
        net.zero_grad()
        (net(x) ** 2).mean().backward()  # the same loss over the whole batch on one rank
        for g, p in zip(got, net.parameters()):
            assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-7)
        if rank == 0:
            open(tmp, 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch(tmp_path):
    marker = str(tmp_path / 'done')
    mp.spawn(_grad_worker, args=(2, _free_port(), marker), nprocs=2, join=True)
    assert open(marker).read() == 'ok'


class _StandInModel(torch.nn.Module):
    """The renderer's interface (dict in, dict out) around a tiny CPU network; records the row offsets it is given."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
        self.seen = []

    def forward(self, batch):
        self.seen.append((int(batch['row_offset']), batch['rays_o'].shape[0]))
        return {'rgb_coarse': torch.sigmoid(self.net(batch['rays_o']))}


class _StandInLosses:
    def compute_losses(self, batch, out):
        mse = torch.mean(torch.square(out['rgb_coarse'] - batch['target_rgb']))
        return {'MSE01': {'loss_value': mse}, 'TotalLoss': mse}


def _train_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        gen = torch.Generator().manual_seed(3)
        full = {'rays_o': torch.randn(16, 3, generator=gen), 'target_rgb': torch.rand(16, 3, generator=gen), 'iter_num': 0,
                'common_data': {'resolution': (4, 4)}}
        # one process, whole batch, two sub-batches of 8
        ref = _StandInModel()
        ref_opt = torch.optim.SGD(ref.parameters(), lr=0.1)
        ref_totals = harness.train_one_iter(ref, _StandInLosses(), ref_opt, full, sub_batch_size=8)
        assert ref.seen == [(0, 8), (8, 8)]
        # two ranks: rank r holds rows [4r, 4r+4) of each sub-batch, i.e. the union over ranks is the same batch
        rows = torch.cat([torch.arange(4 * rank, 4 * rank + 4), 8 + torch.arange(4 * rank, 4 * rank + 4)])
        mine = {k: (v[rows] if isinstance(v, torch.Tensor) else v) for k, v in full.items()}
        mine['row_offset'] = 100 * rank
        model = _StandInModel()
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        totals = harness.train_one_iter(model, _StandInLosses(), opt, mine, sub_batch_size=4, world_size=world)
        assert model.seen == [(100 * rank, 4), (100 * rank + 4, 4)]
        for a, b in zip(model.parameters(), ref.parameters()):     # averaged gradients == full-batch gradients
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
        both = totals['TotalLoss'].clone()
        dist.all_reduce(both)
        assert torch.allclose(both / world, ref_totals['TotalLoss'], rtol=1e-5)
        if rank == 0:
            open(tmp, 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_two_rank_training_iteration_equals_single_process(tmp_path):
    """harness.train_one_iter (sub-batching, loss accumulation, one gradient all-reduce, optimiser step) on two ranks
    that each hold half of every sub-batch reproduces the single-process parameters."""
    marker = str(tmp_path / 'done')
    mp.spawn(_train_worker, args=(2, _free_port(), marker), nprocs=2, join=True)
    assert open(marker).read() == 'ok'


def _run_bench(ranks, *flags):
    """-> (the ONE stdout line, the full record from the side file); tests/util.run_bench holds the run to the output contract"""
    import sys
    from tests import util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return util.run_bench([sys.executable, os.path.join(repo, 'tests', 'bench_rehearsal.py'), '--standin', '--backend', 'gloo', '--',
                           '--gpus', str(ranks), *flags], timeout=600)


@pytest.mark.parametrize('ranks', [2, 3, 8])
def test_bench_self_launches_its_ranks(ranks):
    """``python bench.py --gpus N`` with no launcher around it (how the driver calls it): bench.py starts its N rank
    processes itself, they rendezvous over torch.distributed (gloo here, RCCL on a GPU node), every rank contributes its
    1024-ray block to the gather, and the parent relays ONE JSON line whose ``collective`` object reports the ranks.  The
    renderer is replaced by the CPU stand-in (tests/bench_rehearsal.py calls bench.main with it) -- what is exercised is the launcher and the N > 1
    protocol (settle, warm-up, fenced timed region with per-step stamps, max over ranks), which needs no GPU.  The default
    N > 1 record also carries BASELINE config 4's frame, strong-scaled over the same ranks (``also_measured_frame``).  Round 5:
    the stdout line is the short contract line (tests/util.run_bench), everything else is read from the side file it names;
    ``ranks = 8`` is the driver's ``--gpus 8`` command line run once without hardware (eight 1024-ray blocks, the 8-way
    ``per_rank`` table, 190-ray frame shards)."""
    short, line = _run_bench(ranks, '--steps', '3', '--warmup', '1')
    assert short['collective'] == {'backend': 'gloo', 'ranks': ranks, 'bytes': 1024 * 16,
                                   'gather_ms_p50': pytest.approx(line['collective']['gather_ms']['p50'], rel=1e-4, abs=1e-9)}
    assert short['value'] == pytest.approx(line['value'], rel=1e-4) and 'timing' not in short and 'also_measured_frame' not in short
    assert line['n_gpus'] == ranks and line['steps'] == 3 and line['warmup'] == 1 and line['scaling'] == 'weak'
    assert line['collective']['backend'] == 'gloo' and line['collective']['ranks'] == ranks
    assert line['collective']['bytes'] == 1024 * 16
    # attribution of a sub-linear point: every rank's own step time, kernel time, gather time and elapsed seconds
    per_rank = line['collective']['per_rank']
    assert [r['rank'] for r in per_rank] == list(range(ranks))
    assert all(r['step_ms_p50'] > 0 and r['gather_ms_p50'] >= 0 and r['elapsed_s'] > 0 for r in per_rank)
    assert line['collective']['gather_ms']['max'] >= line['collective']['gather_ms']['p50'] >= 0
    assert line['ms_per_step'] * 1e-3 * 3 == pytest.approx(max(r['elapsed_s'] for r in per_rank), rel=1e-9)
    assert line['data'] == 'stand-in' and line['value'] > 0
    timing = line['timing']
    assert set(timing['step_ms']) == {'p50', 'p90', 'max', 'first', 'argmax'} and len(timing['step_trace_ms']) == 3
    assert timing['settle']['seconds'] >= 0.5 and timing['settle']['runs'] > 0
    frames = line['also_measured_frame']
    assert frames['n_gpus'] == ranks and frames['scaling'] == 'strong' and frames['collective']['ranks'] == ranks
    rays = 37 * 41
    assert [e['rays'] for e in frames['entries']] == [rays, rays] and all(e['value'] > 0 for e in frames['entries'])
    assert frames['collective']['bytes_per_rank_and_frame'] == -(-rays // ranks) * 28


@pytest.mark.parametrize('ranks', [1, 2, 3, 8])
def test_bench_frame_mode_strong_scales_one_frame(ranks):
    """``python bench.py --gpus N --frame re10k`` (BASELINE config 4: one full frame, rays block-sharded over the ranks, one
    gather to rank 0): the stand-in frame has 1517 rays -- ragged over 2 and 3 ranks -- and rank 0 checks every gathered
    frame element by element inside bench.py."""
    short, line = _run_bench(ranks, '--frame', 're10k', '--steps', '2', '--warmup', '1')
    rays = 37 * 41
    assert short['scaling'] == 'strong' and short['config']['rays_per_gpu'] == -(-rays // ranks)
    assert line['n_gpus'] == ranks and line['steps'] == 2 and line['scaling'] == 'strong' and line['data'] == 'stand-in'
    assert line['config']['rays_per_frame'] == rays and line['config']['rays_per_gpu'] == -(-rays // ranks)
    assert line['value'] == pytest.approx(rays / (line['ms_per_step'] * 1e-3), rel=1e-9)
    if ranks > 1:
        assert line['collective'] == {'backend': 'gloo', 'ranks': ranks, 'bytes': -(-rays // ranks) * 28,
                                      'pattern': 'one gather of the five per-ray outputs to rank 0 per frame'}


@pytest.mark.parametrize('mode', [(), ('--frame', 're10k')])
def test_bench_force_collective_runs_the_multi_rank_protocol_with_one_rank(mode):
    """``--force-collective`` at N = 1: a one-rank process group, barriers in the fences, the gather in every step and the
    max-over-ranks reduction -- the code the N > 1 line runs (under RCCL on a GPU box: tests/test_gpu_dist.py)."""
    _, line = _run_bench(1, '--force-collective', '--steps', '2', '--warmup', '1', *mode)
    assert line['n_gpus'] == 1 and line['collective']['backend'] == 'gloo' and line['collective']['ranks'] == 1
    if not mode:
        assert len(line['collective']['per_rank']) == 1 and 'gather' in line['config']['parallelism']
        assert line['also_measured_frame']['collective']['ranks'] == 1


def test_bench_refuses_more_ranks_than_gpus():
    """bench.py proper (no rehearsal switches): ``--gpus 2`` on a node with fewer than two GPUs is an error BEFORE any rank is
    started, and a launcher-provided rank whose LOCAL_RANK has no device of its own is an error before the process group is
    created -- no wrap-around of ranks onto one device (until round 3: ``local_rank % device_count``)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip('a multi-GPU node')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and 'GPU(s)' in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    env.update(RANK='1', LOCAL_RANK='1', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and ('GPU(s)' in r.stderr or 'needs an MI355X' in r.stderr)


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that dies takes the launch down with a non-zero status instead of leaving the others in the rendezvous."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(repo, 'tests', 'bench_rehearsal.py'), '--standin', '--backend', 'no-such-backend',
                        '--', '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_bench_under_the_drivers_launcher():
    """The driver's N > 1 command (``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W``) with the rehearsal wrapper in bench.py's place, which hands
    the same argument list to ``bench.main`` -- the ranks come from the launcher (RANK / LOCAL_RANK /
    WORLD_SIZE in the environment) instead of bench.py's own, and rank 0 alone prints the ONE JSON line."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    from tests import util
    short, line = util.run_bench([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                                  '127.0.0.1', '--master-port', str(port), os.path.join(repo, 'tests', 'bench_rehearsal.py'), '--standin',
                                  '--backend', 'gloo', '--', '--gpus', '2', '--steps', '3', '--warmup', '1'], env=env, timeout=600)
    assert short['n_gpus'] == 2 and short['collective']['ranks'] == 2
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1 and line['scaling'] == 'weak'
    assert line['collective']['ranks'] == 2 and line['value'] > 0 and len(line['timing']['step_trace_ms']) == 3
