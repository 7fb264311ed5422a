"""CPU restatement (numpy) of training-batch assembly, the shuffled index stream and the training draws -- TEST
INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Pinned:
  * ``build_ray_cache`` / ``assemble_batch`` follow the reference's cache-then-gather path
    (src/data_preprocessors/DataPreprocessor01.py:284-349, :586-704) and are held bit-for-bit to
    ``tests/golden/batch_assembly.npz`` (batches produced by the reference's own ``DataPreprocessor``);
  * ``philox4x32_10`` is held to the Random123 known-answer vectors (Salmon et al., "Parallel random numbers: as easy
    as 1, 2, 3", SC'11; kat_vectors of the Random123 distribution);
  * ``shuffled_indices`` restates THIS build's index stream (the reference shuffles a host array with numpy's global
    Mersenne Twister, :269, :562 -- not reproducible on a device); its pins are the permutation properties the
    reference's epoch relies on (every candidate exactly once per epoch) plus bit-equality with the HIP kernel.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy

from . import raygen_oracle

f32 = numpy.float32
M32 = 0xFFFFFFFF


# ----------------------------------------------------------------------------------------------------------------
def build_ray_cache(poses: numpy.ndarray, intrinsics: numpy.ndarray, resolution, near: float, ndc: bool) -> Dict[str, numpy.ndarray]:
    """preprocess_nerf_data :284-349: per-view get_rays (+ get_ndc_rays), stacked and flattened to (V*h*w, .)."""
    h, w = resolution
    rays_o, rays_d, pixel_id, o_ndc, d_ndc = [], [], [], [], []
    for i in range(poses.shape[0]):
        o, d = raygen_oracle.camera_rays(resolution, intrinsics[i], poses[i])
        rays_o.append(o)
        rays_d.append(d)
        gx, gy = numpy.meshgrid(numpy.arange(w, dtype=f32), numpy.arange(h, dtype=f32), indexing='xy')
        pixel_id.append(numpy.stack([i * numpy.ones_like(gx), gx, gy], axis=2))
        if ndc:
            on, dn = raygen_oracle.ndc_rays(o, d, resolution, intrinsics[i], near)
            o_ndc.append(on)
            d_ndc.append(dn)
    rays_d_all = numpy.stack(rays_d, 0)
    cache = {
        'rays_o': numpy.stack(rays_o, 0).reshape(-1, 3).astype(f32), 'rays_d': rays_d_all.reshape(-1, 3).astype(f32),
        'view_dirs': raygen_oracle.unit_dirs(rays_d_all).reshape(-1, 3).astype(f32),
        'pixel_id': numpy.stack(pixel_id, 0).reshape(-1, 3).astype(numpy.int32),
    }
    if ndc:
        cache['rays_o_ndc'] = numpy.stack(o_ndc, 0).reshape(-1, 3).astype(f32)
        cache['rays_d_ndc'] = numpy.stack(d_ndc, 0).reshape(-1, 3).astype(f32)
    return cache


def assemble_batch(indices: numpy.ndarray, num_pixel_rays: int, cache: Dict[str, numpy.ndarray], images: numpy.ndarray,
                   near: float, far: float, ndc: bool, near_ndc: float = 0.0, far_ndc: float = 1.0,
                   sparse_depths: Optional[numpy.ndarray] = None, sparse_errors: Optional[numpy.ndarray] = None,
                   sparse_depths_ndc: Optional[numpy.ndarray] = None) -> Dict[str, numpy.ndarray]:
    """load_nerf_cached_batch :586-636 + load_sparse_depth_cached_batch :655-704: everything starts at -1; pixel rows
    are filled from the cache, then the sparse-depth rows get their rays and depth values (not a target colour)."""
    n = indices.shape[0]
    is_pixel = numpy.arange(n) < num_pixel_rays
    out = {'indices': indices, 'indices_mask_nerf': is_pixel}
    names = ['rays_o', 'rays_d', 'view_dirs'] + (['rays_o_ndc', 'rays_d_ndc'] if ndc else [])
    for name in names:
        out[name] = cache[name][indices]              # both row kinds read the same cache
    out['pixel_id'] = cache['pixel_id'][indices]
    target = numpy.full((n, 3), -1, dtype=f32)
    target[is_pixel] = images.reshape(-1, 3).astype(f32)[indices[is_pixel]]
    out['target_rgb'] = target
    out['near'] = numpy.full((n, 1), near, dtype=f32)
    out['far'] = numpy.full((n, 1), far, dtype=f32)
    if ndc:
        out['near_ndc'] = numpy.full((n, 1), near_ndc, dtype=f32)
        out['far_ndc'] = numpy.full((n, 1), far_ndc, dtype=f32)
    if sparse_depths is not None:
        out['indices_mask_sparse_depth'] = ~is_pixel
        for key, table in (('sparse_depth_values', sparse_depths), ('sparse_depth_errors', sparse_errors),
                           ('sparse_depth_values_ndc', sparse_depths_ndc)):
            if table is not None:
                col = numpy.full((n, 1), -1, dtype=f32)
                col[~is_pixel] = table.reshape(-1, 1)[indices[~is_pixel]]
                out[key] = col
    return out


def precrop_window(height: int, width: int, fraction: float):
    """generate_indices :258-265: rows [h1,h2) x columns [w1,w2) of the central crop."""
    h1, h2 = int(round(height / 2 * (1 - fraction))), int(round(height / 2 * (1 + fraction)))
    w1, w2 = int(round(width / 2 * (1 - fraction))), int(round(width / 2 * (1 + fraction)))
    return h1, h2, w1, w2


# ----------------------------------------------------------------------------------------------------------------
def _splitmix64(z: int) -> int:
    m64 = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & m64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
    return z ^ (z >> 31)


def _mix32(h: numpy.ndarray) -> numpy.ndarray:
    h = h.astype(numpy.uint64)
    h ^= h >> 16
    h = (h * 0x85ebca6b) & M32
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & M32
    h ^= h >> 16
    return h


def shuffled_positions(seed: int, epoch: int, first: int, count: int, domain: int, rounds: int = 8) -> numpy.ndarray:
    """Positions [first, first+count) of the keyed permutation of [0, domain): balanced Feistel network over the next
    even power of two, cycle-walked into the domain."""
    bits = 2
    while (1 << bits) < domain:
        bits += 1
    bits += bits & 1
    half = bits // 2
    mask = (1 << half) - 1
    keys = [_splitmix64(seed ^ _splitmix64(epoch * rounds + r)) & M32 for r in range(rounds)]
    x = numpy.arange(first, first + count, dtype=numpy.uint64)
    todo = numpy.ones(count, dtype=bool)
    while todo.any():
        cur = x[todo]
        left, right = cur >> numpy.uint64(half), cur & numpy.uint64(mask)
        for r in range(rounds):
            nxt = left ^ (_mix32((right + numpy.uint64(keys[r])) & numpy.uint64(M32)) & numpy.uint64(mask))
            left, right = right, nxt
        cur = (left << numpy.uint64(half)) | right
        x[todo] = cur
        todo[todo] = cur >= domain
    return x.astype(numpy.int64)


def shuffled_indices(seed: int, epoch: int, first: int, count: int, domain: int, candidates: Optional[numpy.ndarray] = None,
                     num_views: int = 0, height: int = 0, width: int = 0, crop=None) -> numpy.ndarray:
    pos = shuffled_positions(seed, epoch, first, count, domain)
    if candidates is not None:
        return candidates[pos]
    y0, y1, x0, x1 = crop if crop is not None else (0, height, 0, width)
    per_view = (y1 - y0) * (x1 - x0)
    view, rest = pos // per_view, pos % per_view
    return (view * height + y0 + rest // (x1 - x0)) * width + x0 + rest % (x1 - x0)


# ----------------------------------------------------------------------------------------------------------------
def philox4x32_10(counter: numpy.ndarray, key) -> numpy.ndarray:
    """counter (..., 4) uint32, key (2,) -> (..., 4) uint32.  Philox-4x32 with 10 rounds."""
    c = [counter[..., i].astype(numpy.uint64) for i in range(4)]
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0 = c[0] * numpy.uint64(0xD2511F53)
        p1 = c[2] * numpy.uint64(0xCD9E8D57)
        c = [(p1 >> numpy.uint64(32)) ^ c[1] ^ numpy.uint64(k0), p1 & numpy.uint64(M32),
             (p0 >> numpy.uint64(32)) ^ c[3] ^ numpy.uint64(k1), p0 & numpy.uint64(M32)]
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return numpy.stack(c, -1).astype(numpy.uint32)


def _draw_words(seed: int, stream_id: int, first_row: int, num_rows: int, row_width: int) -> numpy.ndarray:
    blocks = (row_width + 3) // 4
    rows = (first_row + numpy.arange(num_rows, dtype=numpy.uint64))[:, None] * numpy.ones((1, blocks), dtype=numpy.uint64)
    ctr = numpy.stack([rows & numpy.uint64(M32), rows >> numpy.uint64(32),
                       numpy.arange(blocks, dtype=numpy.uint64)[None, :] * numpy.ones((num_rows, 1), dtype=numpy.uint64),
                       numpy.full((num_rows, blocks), stream_id, dtype=numpy.uint64)], -1)
    return philox4x32_10(ctr, (seed & M32, (seed >> 32) & M32))       # (rows, blocks, 4)


def random_uniform(seed: int, stream_id: int, first_row: int, num_rows: int, row_width: int) -> numpy.ndarray:
    words = _draw_words(seed, stream_id, first_row, num_rows, row_width)
    u = (words >> numpy.uint32(8)).astype(f32) * f32(2.0 ** -24)
    return u.reshape(num_rows, -1)[:, :row_width]


def random_normal(seed: int, stream_id: int, first_row: int, num_rows: int, row_width: int, scale: float = 1.0) -> numpy.ndarray:
    words = _draw_words(seed, stream_id, first_row, num_rows, row_width)
    u1 = ((words[..., 0::2] >> numpy.uint32(8)).astype(numpy.float64) + 1.0) * 2.0 ** -24
    u2 = (words[..., 1::2] >> numpy.uint32(8)).astype(numpy.float64) * 2.0 ** -24
    radius = numpy.sqrt(-2.0 * numpy.log(u1))
    z = numpy.stack([radius * numpy.cos(2 * numpy.pi * u2), radius * numpy.sin(2 * numpy.pi * u2)], -1)   # (rows, blocks, 2, 2)
    return (scale * z).reshape(num_rows, -1)[:, :row_width].astype(f32)
