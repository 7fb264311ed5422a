"""Shared helpers for the test-suite: golden loading, weight reconstruction, error metrics."""
import os

import numpy
import torch

from simplenerf_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

MLP_PREFIXES = ('coarse_model.', 'fine_model.', 'pts_aug_coarse_model.', 'pts_aug_fine_model.',
                'views_aug_coarse_model.', 'views_aug_fine_model.')


def load(name):
    with numpy.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def mlp_param_shapes(mlp_cfg: dict, prefix: str = '') -> dict:
    """Parameter shapes of one MLP, derived from its config the way the reference's constructor does
    (src/models/SimpleNeRF01.py:567-609).  Independent restatement used by the tests to build weights."""
    dp, wp = mlp_cfg['points_net_depth'], mlp_cfg['points_net_width']
    dv, wv = mlp_cfg['views_net_depth'], mlp_cfg['views_net_width']
    full_pe = 3 + 6 * mlp_cfg['points_positional_encoding_degree']
    pts_in = full_pe
    views_in = (3 + 6 * mlp_cfg['views_positional_encoding_degree']) if mlp_cfg['use_view_dirs'] else 0
    if 'points_sigma_positional_encoding_degree' in mlp_cfg:
        pts_in = (2 * mlp_cfg['points_sigma_positional_encoding_degree'] + 1) * 3
        views_in += full_pe - pts_in
    shapes = {}

    def lin(name, fin, fout):
        shapes[f'{prefix}{name}.weight'] = (fout, fin)
        shapes[f'{prefix}{name}.bias'] = (fout,)

    lin('pts_linears.0', pts_in, wp)
    for i in range(dp - 1):
        lin(f'pts_linears.{i + 1}', wp + pts_in if i == 4 else wp, wp)
    view_dep = mlp_cfg['view_dependent_rgb']
    if view_dep:
        lin('views_linears.0', views_in + wp, wv)
        for i in range(dv - 1):
            lin(f'views_linears.{i + 1}', wv, wv)
    lin('pts_output_linear', wp, 1 if view_dep else 4)
    if view_dep:
        lin('feature_linear', wp, wp)
        lin('views_output_linear', wv, 4 if mlp_cfg.get('predict_visibility') else 3)
    return shapes


def model_param_shapes(configs: dict) -> dict:
    m = configs['model']
    shapes = {}
    if 'coarse_mlp' in m:
        shapes.update(mlp_param_shapes(m['coarse_mlp'], 'coarse_model.'))
    if 'fine_mlp' in m:
        shapes.update(mlp_param_shapes(m['fine_mlp'], 'fine_model.'))
    for key, short in (('points_augmentation', 'pts_aug'), ('views_augmentation', 'views_aug')):
        if key in m:
            for level in ('coarse', 'fine'):
                if f'{level}_mlp' in m[key]:
                    shapes.update(mlp_param_shapes(m[key][f'{level}_mlp'], f'{short}_{level}_model.'))
    return shapes


def golden_params(configs: dict, golden: dict) -> dict:
    """Weights for an e2e golden: synth(seed) + the density-head overrides stored in the fixture."""
    sd = synth.synth_state_dict(model_param_shapes(configs), int(golden['seed']))
    for k, v in golden.items():
        if k.startswith('ovr_'):
            sd[k[4:]] = v
    if int(golden.get('fine_equals_coarse', 0)):
        for k in list(sd):
            if k.startswith('coarse_model.'):
                sd['fine_model.' + k[len('coarse_model.'):]] = sd[k]
    return {k: torch.from_numpy(numpy.ascontiguousarray(v)) for k, v in sd.items()}


def golden_batch(golden: dict) -> dict:
    return {k[3:]: torch.from_numpy(v) for k, v in golden.items() if k.startswith('in_')}


# observed parity figures of the GPU tests, printed in pytest's terminal summary (tests/conftest.py) so that the test
# record shows how far the gates are from what was measured: [(tag, text)]
OBSERVED = []


def observe(tag: str, text: str) -> None:
    OBSERVED.append((tag, text))


def linf(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else numpy.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(numpy.max(numpy.abs(a.astype(numpy.float64) - b.astype(numpy.float64))))


def rel_linf(a, b, floor=1.0):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else numpy.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = numpy.abs(a.astype(numpy.float64) - b.astype(numpy.float64))
    return float(numpy.max(d / numpy.maximum(numpy.abs(b.astype(numpy.float64)), floor)))


def outlier_fraction(a, b, tol):
    """Fraction of entries that differ by more than ``tol`` (resampled depths sit on rounding-sensitive
    thresholds, SURVEY 8a row 8: compare them outlier-tolerantly)."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else numpy.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else numpy.asarray(b)
    return float(numpy.mean(numpy.abs(a.astype(numpy.float64) - b.astype(numpy.float64)) > tol))


GRAD_SAMPLE_STRIDE = 61


def grad_loss(out):
    """The fixed scalar loss of tools/make_golden.py (G7): mean squares of every rgb_* / depth_* output the shipped
    losses read, depths scaled by 0.01."""
    loss = 0.
    for k in sorted(out):
        base = k.replace('points_augmentation_', '').replace('views_augmentation_', '')
        if base in ('rgb_coarse', 'rgb_fine'):
            loss = loss + (out[k] ** 2).mean()
        elif base in ('depth_coarse', 'depth_fine'):
            loss = loss + 0.01 * (out[k] ** 2).mean()
    return loss


LOSS_OUTPUT_KEYS = ('rgb_coarse', 'rgb_fine', 'points_augmentation_rgb_coarse', 'views_augmentation_rgb_coarse',
                    'depth_coarse', 'depth_fine', 'points_augmentation_depth_coarse', 'views_augmentation_depth_coarse')


def loss_case(golden: dict, device='cpu'):
    """Rebuild (configs, input_dict, output_dict) of a G8 loss fixture from its recorded seeds.  ``common_data`` is
    un-replicated (what LossComputer sees after it has taken [0])."""
    scene = synth.synth_scene(int(golden['scene_seed']))
    batch = synth.loss_batch(scene, int(golden['num_rays']), int(golden['num_sparse']), int(golden['batch_seed']))
    sparse = bool(golden['sparse_in_batch'])
    configs = synth.make_configs('config3')
    configs['losses'] = synth.loss_configs()
    if sparse:
        configs['data_loader']['sparse_depth'] = {}
    t = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).to(device)
    input_dict = {
        'iter_num': int(golden['iter_num']),
        'rays_o': t(batch['rays_o']), 'rays_d': t(batch['rays_d']), 'pixel_id': t(batch['pixel_id']),
        'target_rgb': t(batch['target_rgb']), 'indices_mask_nerf': t(batch['indices_mask_nerf']),
        'common_data': {'poses': t(scene['poses']), 'images': t(scene['images']),
                        'intrinsics': t(scene['intrinsics']), 'resolution': scene['resolution']},
    }
    if sparse:
        input_dict['indices_mask_sparse_depth'] = t(batch['indices_mask_sparse_depth'])
        input_dict['sparse_depth_values'] = t(batch['sparse_depth_values'])
    output_dict = {k: t(batch[k]).clone().requires_grad_(True) for k in LOSS_OUTPUT_KEYS}
    return configs, input_dict, output_dict


# ---------------------------------------------------------------------------------------------------- bench.py's output contract
BENCH_LINE_KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                   'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'collective', 'also', 'extra'}
BENCH_ROOFLINE_KEYS = {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_algorithmic', 'traffic_source', 'kernel',
                       'launches', 'avg_launch_ms'}


def run_bench(command, env=None, timeout=900):
    """Run a bench command (``[python, bench.py | tests/bench_rehearsal.py ..., flags]``) the way a driver might capture it --
    stdout and stderr MERGED into one stream -- and hold it to the output contract (bench.py's docstring):
    the last non-empty line of that stream is the ONE JSON line, it is shorter than 2 KB, carries only the contract's keys, and
    names the side file with the full record.  -> (line, full record)"""
    import json
    import subprocess
    import tempfile
    env = dict(os.environ if env is None else env)
    for key in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(key, None)
    with tempfile.TemporaryDirectory() as tmp:
        side = os.path.join(tmp, 'extra.json')
        r = subprocess.run(list(command) + ['--extra-file', side], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=timeout, env=env)
        assert r.returncode == 0, r.stdout[-6000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert lines and lines[-1].startswith('{'), r.stdout[-3000:]
        assert len([ln for ln in lines if ln.startswith('{"metric"')]) == 1, r.stdout[-3000:]
        text = lines[-1]
        assert len(text) < 2048, len(text)
        line = json.loads(text)
        assert set(line) <= BENCH_LINE_KEYS, set(line) - BENCH_LINE_KEYS
        if 'roofline' in line:
            assert set(line['roofline']) <= BENCH_ROOFLINE_KEYS, set(line['roofline']) - BENCH_ROOFLINE_KEYS
        assert line['extra'] == side
        with open(side) as f:
            full = json.load(f)
    for key in ('metric', 'n_gpus', 'steps', 'warmup', 'scaling', 'data'):
        assert line[key] == full[key], key
    return line, full
