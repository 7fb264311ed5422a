"""ORACLE -- test infrastructure, not product code.

CPU restatement (PyTorch CPU tensor ops, fp32) of the reference's ray-marching path, written as flat functions
over plain tensors and a ``{name: tensor}`` parameter dictionary.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this module; the shipped HIP path never does.

Pinned against the reference: ``tools/make_golden.py`` imports ``/root/reference/src/models/SimpleNeRF01.py``
in the build container and stores its outputs for seeded inputs in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those vectors.

Every function cites the reference lines (relative to ``/root/reference/src/models/SimpleNeRF01.py``) whose
arithmetic it follows.  Where rounding order matters for parity (sample positions feed sin(512 x)), the op order
of the reference expression is kept.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SKIP_AFTER_LAYER = 4  # reference hard-codes skips=[4] (:580)


# --------------------------------------------------------------------------------------------------
# sampling along the ray
# --------------------------------------------------------------------------------------------------
def coarse_depths(near: Tensor, far: Tensor, num_samples: int, lindisp: bool = False,
                  t_rand: Optional[Tensor] = None) -> Tensor:
    """Evenly spaced (or stratified, when ``t_rand`` in [0,1) is supplied) depths; (N,1),(N,1) -> (N,S).

    Follows get_z_vals_coarse :272-302.  ``t_rand`` stands for the reference's ``torch.rand(N,S)`` draw (:299).
    """
    t = torch.linspace(0., 1., steps=num_samples)
    if not lindisp:
        z = near * (1. - t) + far * t
    else:
        z = 1. / (1. / near * (1. - t) + 1. / far * t)
    z = z.expand([near.shape[0], num_samples])
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def ray_points(origins: Tensor, directions: Tensor, depths: Tensor) -> Tensor:
    """o + d*z with the multiply and the add rounded separately (:140-142, :204-206). -> (N,S,3)"""
    return origins[..., None, :] + directions[..., None, :] * depths[..., :, None]


def resample_depths(z_coarse: Tensor, weights_coarse: Tensor, num_fine: int, u: Optional[Tensor] = None) -> Tensor:
    """Inverse-CDF samples from the coarse weights merged with the coarse depths, sorted; -> (N, S_c+S_f).

    Follows get_z_vals_fine :304-315 and sample_pdf :329-361.  ``u`` (N,S_f) stands for the reference's
    ``torch.rand`` draw (:341); ``None`` selects the deterministic ``linspace`` of :338.
    """
    bins = .5 * (z_coarse[..., 1:] + z_coarse[..., :-1])
    w = weights_coarse[..., 1:-1] + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = torch.linspace(0., 1., steps=num_fine).expand(list(cdf.shape[:-1]) + [num_fine])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    samples = (bin_b + t * (bin_a - bin_b)).detach()  # no gradient through the sample positions (:312)
    z_fine, _ = torch.sort(torch.cat([z_coarse, samples], -1), -1)
    return z_fine


# --------------------------------------------------------------------------------------------------
# positional encoding + MLP
# --------------------------------------------------------------------------------------------------
def pos_encode(x: Tensor, degree: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] -> (..., 3+6L).  (:533-557, :612-624)"""
    freqs = 2. ** torch.linspace(0., degree - 1, steps=degree)
    parts = [x]
    for f in freqs:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def mlp_layout(params: Dict[str, Tensor], prefix: str) -> dict:
    """Recover the layer structure of one MLP from the parameter shapes alone (cf. MLP.__init__ :561-609)."""
    depth = 0
    while f'{prefix}pts_linears.{depth}.weight' in params:
        depth += 1
    views_depth = 0
    while f'{prefix}views_linears.{views_depth}.weight' in params:
        views_depth += 1
    return {
        'depth': depth,
        'views_depth': views_depth,
        'width': params[f'{prefix}pts_linears.0.weight'].shape[0],
        'pts_in': params[f'{prefix}pts_linears.0.weight'].shape[1],
        'pts_out': params[f'{prefix}pts_output_linear.weight'].shape[0],
        'view_dependent': views_depth > 0,
    }


def mlp_forward(params: Dict[str, Tensor], prefix: str, mlp_cfg: dict, pts: Tensor, view_dirs: Optional[Tensor],
                sigma_noise: Optional[Tensor] = None, view_dirs2: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """One NeRF MLP on flat points.  pts (B,3), view_dirs (B,3)|None -> sigma (B,1), rgb (B,3).

    Follows MLP.forward :626-654, get_view_independent_outputs :656-685, get_view_dependent_outputs :687-715.
    ``sigma_noise`` (B,1) stands for ``randn * raw_noise_std`` (:670) and is added before the ReLU.
    """
    lay = mlp_layout(params, prefix)
    enc = pos_encode(pts, mlp_cfg['points_positional_encoding_degree'])
    trunk_in = enc[:, :lay['pts_in']]
    h = trunk_in
    for i in range(lay['depth']):
        h = F.relu(F.linear(h, params[f'{prefix}pts_linears.{i}.weight'], params[f'{prefix}pts_linears.{i}.bias']))
        if i == SKIP_AFTER_LAYER:
            h = torch.cat([trunk_in, h], -1)
    head = F.linear(h, params[f'{prefix}pts_output_linear.weight'], params[f'{prefix}pts_output_linear.bias'])
    sigma = head[..., 0:1]
    if sigma_noise is not None:
        sigma = sigma + sigma_noise
    sigma = F.relu(sigma)
    out = {'sigma': sigma}
    if lay['pts_out'] == 4:
        out['rgb_view_independent'] = torch.sigmoid(head[..., 1:4])
        out['rgb'] = out['rgb_view_independent']
    if lay['view_dependent']:
        feature = F.linear(h, params[f'{prefix}feature_linear.weight'], params[f'{prefix}feature_linear.bias'])
        feature = torch.cat([feature, enc[:, lay['pts_in']:]], dim=1)
        enc_views = pos_encode(view_dirs, mlp_cfg['views_positional_encoding_degree'])


        def views_head(encoded):       # get_view_dependent_outputs :687-715; `encoded` (B,27) or (B,K,27)
            feat = feature if encoded.dim() == 2 else feature[:, None, :].repeat([1, encoded.shape[1], 1])
            hv = torch.cat([feat, encoded], -1)
            for i in range(lay['views_depth']):
                hv = F.relu(F.linear(hv, params[f'{prefix}views_linears.{i}.weight'],
                                     params[f'{prefix}views_linears.{i}.bias']))
            return F.linear(hv, params[f'{prefix}views_output_linear.weight'], params[f'{prefix}views_output_linear.bias'])

        vout = views_head(enc_views)
        out['rgb_view_dependent'] = torch.sigmoid(vout[..., 0:3])
        if mlp_cfg.get('predict_visibility', False):          # 4th row of the views head (:708-712)
            out['visibility'] = torch.sigmoid(vout[..., 3:4])
            if view_dirs2 is not None:                          # the same head per secondary direction (:646-649)
                vout2 = views_head(pos_encode(view_dirs2, mlp_cfg['views_positional_encoding_degree']))
                out['visibility2'] = torch.sigmoid(vout2[..., 3:4])
        out['rgb'] = out['rgb_view_dependent']
    return out


def run_mlp(params, prefix, mlp_cfg, pts: Tensor, view_dirs: Optional[Tensor], netchunk: Optional[int],
            sigma_noise: Optional[Tensor] = None, view_dirs2: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """(N,S,3) points through the MLP in ``netchunk``-row pieces (run_network :363-392, batchify :394-428).
    ``view_dirs2`` (N,S,K,3): per-sample secondary directions of a predict_visibility MLP."""
    n, s = pts.shape[:2]
    flat = pts.reshape(-1, 3)
    vflat = None
    if view_dirs is not None and mlp_cfg['use_view_dirs']:
        vflat = view_dirs[:, None].expand(pts.shape).reshape(-1, 3)
    nflat = None if sigma_noise is None else sigma_noise.reshape(-1, 1)
    v2flat = None
    if view_dirs2 is not None and mlp_cfg['use_view_dirs'] and mlp_cfg.get('predict_visibility', False):
        v2flat = view_dirs2.reshape(-1, view_dirs2.shape[-2], 3)
    step = flat.shape[0] if netchunk is None else netchunk
    pieces: Dict[str, list] = {}
    for i in range(0, flat.shape[0], step):
        piece = mlp_forward(params, prefix, mlp_cfg, flat[i:i + step], None if vflat is None else vflat[i:i + step],
                            None if nflat is None else nflat[i:i + step], None if v2flat is None else v2flat[i:i + step])
        for k, v in piece.items():
            pieces.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0).reshape([n, s] + list(v[0].shape[1:])) for k, v in pieces.items()}


def other_view_dirs(z: Tensor, rays_o: Tensor, rays_d: Tensor, rays_o2: Tensor, ndc: bool) -> Tensor:
    """compute_other_view_dirs :317-326: unit directions from the secondary camera centres (N,K,3) to every sample."""
    if ndc:
        tn = -(1 + rays_o[..., 2]) / rays_d[..., 2]
        z = (((rays_o[..., None, 2] + tn[..., None] * rays_d[..., None, 2]) / (1 - z + 1e-6)) - rays_o[..., None, 2]) / rays_d[..., None, 2]
    pts = rays_o[..., None, :] + z[..., None] * rays_d[..., None, :]
    dirs = pts[:, :, None] - rays_o2[..., None, :, :]
    return dirs / torch.norm(dirs, dim=-1, keepdim=True)


def secondary_origins(batch: Dict[str, Tensor]) -> Tensor:
    """rays_o2 of render_rays :120-133: given, or the centres of the other training views per ray."""
    if 'rays_o2' in batch:
        return batch['rays_o2']
    poses = batch['common_data']['poses']
    image_id = batch['pixel_id'][:, 0].long()
    cols = []
    for i in range(int(batch['num_frames']) - 1):
        other = i + (i >= image_id).long()
        cols.append(poses[other][:, :3, 3])
    return torch.stack(cols, dim=1)


# --------------------------------------------------------------------------------------------------
# alpha compositing
# --------------------------------------------------------------------------------------------------
def ndc_to_world_depth(z_ndc: Tensor, rays_o: Tensor, rays_d: Tensor) -> Tensor:
    """convert_depth_from_ndc :486-502 (near plane hard-coded to 1 there)."""
    oz = rays_o[..., 2:3]
    dz = rays_d[..., 2:3]
    tn = -(1 + oz) / dz
    c = torch.where(z_ndc == 1., 1e-3, 0.)
    return (oz + tn * dz) / dz * (1 / (1 - z_ndc + c) - 1) + tn


def composite(sigma: Tensor, rgb: Tensor, z: Tensor, march_dirs: Tensor, ndc: bool, white_bkgd: bool = False,
              rays_o: Optional[Tensor] = None, rays_d: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """sigma (N,S), rgb (N,S,3), z (N,S) -> per-ray colour/opacity/depth + per-sample alpha, visibility, weights.

    ``march_dirs`` are the directions the depths are measured along (world rays_d, or rays_d_ndc when ``ndc``);
    ``rays_o``/``rays_d`` (world) are needed only for the NDC -> world depth conversion.
    Follows volume_rendering :430-483.
    """
    far_cap = torch.tensor([1. if ndc else 1e10])
    z1 = torch.cat([z, far_cap.expand(z[..., :1].shape)], -1)
    delta = (z1[..., 1:] - z1[..., :-1]) * torch.norm(march_dirs[..., None, :], dim=-1)
    alpha = 1. - torch.exp(-sigma * delta)
    vis = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * vis
    rgb_map = torch.sum(weights[..., None] * rgb, dim=-2)
    acc = torch.sum(weights, dim=-1)
    out = {'acc': acc, 'alpha': alpha, 'visibility': vis, 'weights': weights}
    if not ndc:
        depth = torch.sum(weights * z, dim=-1) / (acc + 1e-6)
        out['depth'] = depth
        out['depth_var'] = torch.sum(weights * torch.square(z - depth[..., None]), dim=-1)
    else:
        depth_ndc = torch.sum(weights * z, dim=-1) / (acc + 1e-6)
        out['depth_ndc'] = depth_ndc
        out['depth_var_ndc'] = torch.sum(weights * torch.square(z - depth_ndc[..., None]), dim=-1)
        zw = ndc_to_world_depth(z, rays_o, rays_d)
        depth = torch.sum(weights * zw, dim=-1) / (acc + 1e-6)
        out['depth'] = depth
        out['depth_var'] = torch.sum(weights * torch.square(zw - depth[..., None]), dim=-1)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc[..., None])
    out['rgb'] = rgb_map
    return out


# --------------------------------------------------------------------------------------------------
# orchestration
# --------------------------------------------------------------------------------------------------
_AUG = (('points_augmentation', 'pts_aug'), ('views_augmentation', 'views_aug'))


def render_chunk(params: Dict[str, Tensor], configs: dict, batch: Dict[str, Tensor], training: bool, retraw: bool,
                 rand: Optional[dict] = None, sec_views_vis: bool = False) -> Dict[str, Tensor]:
    """One ``chunk`` of rays through coarse (+augmented, when training) and fine passes.  (render_rays :108-270)

    ``rand`` carries the draws the reference takes from the CPU generator, by name:
    ``t_rand`` (N,S_c); ``u`` (N,S_f); ``noise_coarse``, ``noise_points_augmentation``,
    ``noise_views_augmentation`` (N,S_c,1); ``noise_fine`` (N,S_c+S_f,1) -- already scaled by raw_noise_std.
    Missing entries mean "no perturbation / no noise" for that draw.
    """
    rand = rand or {}
    mcfg = configs['model']
    ndc = configs['data_loader']['ndc']
    netchunk = mcfg['netchunk']
    rays_o, rays_d = batch['rays_o'], batch['rays_d']
    if ndc:
        march_o, march_d = batch['rays_o_ndc'], batch['rays_d_ndc']
        near, far = batch['near_ndc'], batch['far_ndc']
    else:
        march_o, march_d = rays_o, rays_d
        near, far = batch['near'], batch['far']
    view_dirs = batch.get('view_dirs')
    model_predicts = mcfg['coarse_mlp'].get('predict_visibility', False) or \
        ('fine_mlp' in mcfg and mcfg['fine_mlp'].get('predict_visibility', False))
    rays_o2 = secondary_origins(batch) if model_predicts and sec_views_vis else None

    def shade(prefix, mlp_cfg, z, noise, dirs2=None):
        pts = ray_points(march_o, march_d, z)
        raw = run_mlp(params, prefix, mlp_cfg, pts, view_dirs, netchunk, noise, dirs2)
        comp = composite(raw['sigma'][..., 0], raw['rgb'], z, march_d, ndc, mcfg['white_bkgd'], rays_o, rays_d)
        if model_predicts and sec_views_vis and 'visibility2' in raw:      # volume_rendering :479-482
            comp['visibility2'] = torch.sum(comp['weights'][..., None] * raw['visibility2'][..., 0], dim=-2) / (comp['acc'][..., None] + 1e-6)
        return raw, comp

    def dirs2_for(level_cfg, z):   # render_rays :150-152, :213-215: only the MAIN model's flag decides
        if rays_o2 is None or not level_cfg.get('predict_visibility', False):
            return None
        return other_view_dirs(z, rays_o, rays_d, rays_o2, ndc)

    out: Dict[str, Tensor] = {}

    def emit(tag, level, raw, comp):
        for k, v in comp.items():
            out[f'{tag}{k}_{level}'] = v
        if retraw:
            for k, v in raw.items():
                out[f'{tag}raw_{k}_{level}'] = v

    z_coarse = coarse_depths(near, far, mcfg['coarse_mlp']['num_samples'], mcfg['lindisp'],
                             rand.get('t_rand') if training else None)
    dirs2_coarse = dirs2_for(mcfg['coarse_mlp'], z_coarse)
    raw, comp = shade('coarse_model.', mcfg['coarse_mlp'], z_coarse, rand.get('noise_coarse'), dirs2_coarse)
    weights_coarse = comp['weights']
    out['z_vals_coarse'] = z_coarse
    emit('', 'coarse', raw, comp)
    if training:
        for cfg_key, short in _AUG:
            if cfg_key in mcfg and 'coarse_mlp' in mcfg[cfg_key]:
                raw, comp = shade(f'{short}_coarse_model.', mcfg[cfg_key]['coarse_mlp'], z_coarse,
                                  rand.get(f'noise_{cfg_key}'), dirs2_coarse)
                emit(f'{cfg_key}_', 'coarse', raw, comp)
    if 'fine_mlp' in mcfg:
        z_fine = resample_depths(z_coarse, weights_coarse, mcfg['fine_mlp']['num_samples'],
                                 rand.get('u') if training else None)
        dirs2_fine = dirs2_for(mcfg['fine_mlp'], z_fine)
        raw, comp = shade('fine_model.', mcfg['fine_mlp'], z_fine, rand.get('noise_fine'), dirs2_fine)
        out['z_vals_fine'] = z_fine
        emit('', 'fine', raw, comp)
        if training:        # fine-level augmentation MLPs on the same fine points (render_rays :234-263)
            for cfg_key, short in _AUG:
                if cfg_key in mcfg and 'fine_mlp' in mcfg[cfg_key]:
                    raw, comp = shade(f'{short}_fine_model.', mcfg[cfg_key]['fine_mlp'], z_fine,
                                      rand.get(f'noise_{cfg_key}_fine'), dirs2_fine)
                    emit(f'{cfg_key}_', 'fine', raw, comp)
    if not retraw:
        for level in ('coarse', 'fine'):
            for k in ('z_vals', 'visibility', 'weights'):
                out.pop(f'{k}_{level}', None)
    return out


def render(params: Dict[str, Tensor], configs: dict, batch: Dict[str, Tensor], training: bool = False,
           retraw: bool = False, rand_per_chunk: Optional[list] = None, sec_views_vis: bool = False) -> Dict[str, Tensor]:
    """Whole batch in ``chunk``-ray pieces, outputs concatenated (forward :67-75, batchify_rays :81-106)."""
    retraw = retraw or training
    sec_views_vis = sec_views_vis or training
    n = batch['rays_o'].shape[0]
    chunk = configs['model']['chunk']
    pieces: Dict[str, list] = {}
    for ci, i in enumerate(range(0, n, chunk)):
        sub = {k: (v[i:i + chunk] if isinstance(v, torch.Tensor) and v.shape[0] == n else v) for k, v in batch.items()}
        res = render_chunk(params, configs, sub, training, retraw, None if rand_per_chunk is None else rand_per_chunk[ci],
                           sec_views_vis)
        for k, v in res.items():
            pieces.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in pieces.items()}


# --------------------------------------------------------------------------------------------------
# the reference's random-draw order, replayed (SimpleNeRF01.py:299, :670, :341)
# --------------------------------------------------------------------------------------------------
def replay_reference_draws(configs: dict, num_rays: int, seed: int) -> list:
    """Re-draw, from a CPU generator seeded like ``torch.manual_seed(seed)``, exactly the tensors a training-mode
    reference forward consumes, in its order: per ray chunk -- rand(N,S_c); randn per netchunk for the main coarse
    MLP, then points-aug, then views-aug; rand(N,S_f); randn per netchunk for the fine MLP.  Returns one ``rand``
    dict per chunk in the form ``render_chunk`` takes.  Draws that the config switches off are skipped, as the
    reference skips them (perturb == 0 -> no rand; raw_noise_std == 0 -> no randn)."""
    gen = torch.Generator()
    gen.manual_seed(seed)
    mcfg = configs['model']
    s_c = mcfg['coarse_mlp']['num_samples']
    std = mcfg['raw_noise_std']
    netchunk = mcfg['netchunk']

    def noise(rows):
        if not std > 0.:
            return None
        step = rows if netchunk is None else netchunk
        return torch.cat([torch.randn((min(step, rows - i), 1), generator=gen) * std for i in range(0, rows, step)], 0)

    per_chunk = []
    for i in range(0, num_rays, mcfg['chunk']):
        n = min(mcfg['chunk'], num_rays - i)
        rand = {}
        if mcfg['perturb'] > 0.:
            rand['t_rand'] = torch.rand((n, s_c), generator=gen)
        rand['noise_coarse'] = noise(n * s_c)
        for cfg_key, _ in _AUG:
            if cfg_key in mcfg and 'coarse_mlp' in mcfg[cfg_key]:
                rand[f'noise_{cfg_key}'] = noise(n * s_c)
        if 'fine_mlp' in mcfg:
            s_f = mcfg['fine_mlp']['num_samples']
            if mcfg['perturb'] > 0.:
                rand['u'] = torch.rand((n, s_f), generator=gen)
            rand['noise_fine'] = noise(n * (s_c + s_f))
        for k in list(rand):
            if rand[k] is None:
                del rand[k]
            elif k.startswith('noise_'):
                rand[k] = rand[k].reshape(n, -1, 1)
        per_chunk.append(rand)
    return per_chunk
