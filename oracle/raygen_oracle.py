"""ORACLE -- test infrastructure, not product code.

numpy restatement of the reference's host-side ray generation (file:line relative to
``/root/reference/src/data_preprocessors/DataPreprocessor01.py``).  Pinned by golden vector G1
(``tests/golden/raygen_*.npz``, produced by ``tools/make_golden.py`` from the reference itself).
"""
from __future__ import annotations

import numpy


def process_pose(pose: numpy.ndarray, average_pose: numpy.ndarray, translation_scale: float) -> numpy.ndarray:
    """Raw world-to-camera 4x4 -> the camera-to-world pose ``get_rays`` consumes (test mode).

    preprocess_poses :937-976 with train_mode=False: scale the translation, recentre
    (``avg_pose @ inv(pose)`` :978-980), then flip to the (x,-y,-z) convention :982-988, :1020-1030;
    result cast to float32 (:975).  Arithmetic in float64 like the reference (poses arrive as float64).
    """
    p = numpy.array(pose, dtype=numpy.float64)
    p[:3, 3] *= translation_scale
    c = numpy.asarray(average_pose, dtype=numpy.float64) @ numpy.linalg.inv(p)
    flip = numpy.diag([1., -1., -1.])
    rot = flip.T @ c[:3, :3] @ flip
    tra = flip @ c[:3, 3:]
    out = numpy.concatenate([numpy.concatenate([rot, tra], axis=1), c[3:]], axis=0)
    return out.astype(numpy.float32)


def camera_rays(resolution, intrinsic: numpy.ndarray, pose: numpy.ndarray, pixel_centre_offset: float = 0.0):
    """Pinhole rays for every pixel; -> rays_o, rays_d of shape (h, w, 3), float32.  get_rays :351-368."""
    h, w = resolution
    x, y = numpy.meshgrid(numpy.arange(w, dtype=numpy.float32), numpy.arange(h, dtype=numpy.float32), indexing='xy')
    if pixel_centre_offset:
        x = x + numpy.float32(pixel_centre_offset)
        y = y + numpy.float32(pixel_centre_offset)
    homo = numpy.stack([x, y, numpy.ones_like(x)], axis=2)
    dirs = (numpy.linalg.inv(intrinsic)[None, None] @ homo[:, :, :, None])[:, :, :, 0]
    dirs[:, :, 1:] *= -1
    rays_d = numpy.sum(dirs[..., numpy.newaxis, :] * pose[:3, :3], -1)
    rays_o = numpy.broadcast_to(pose[:3, -1], rays_d.shape)
    return rays_o, rays_d


def unit_dirs(rays_d: numpy.ndarray) -> numpy.ndarray:
    """get_view_dirs :392-394."""
    return rays_d / numpy.linalg.norm(rays_d, ord=2, axis=-1, keepdims=True)


def ndc_rays(rays_o, rays_d, resolution, intrinsic, near):
    """Forward-facing NDC warp of rays; get_ndc_rays :371-389."""
    h, w = resolution
    fx, fy = intrinsic[0, 0], intrinsic[1, 1]
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    o0 = -1. / (w / (2. * fx)) * o[..., 0] / o[..., 2]
    o1 = -1. / (h / (2. * fy)) * o[..., 1] / o[..., 2]
    o2 = 1. + 2. * near / o[..., 2]
    d0 = -1. / (w / (2. * fx)) * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = -1. / (h / (2. * fy)) * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2. * near / o[..., 2]
    return numpy.stack([o0, o1, o2], -1), numpy.stack([d0, d1, d2], -1)


def full_frame_batch(resolution, intrinsic, pose, near, far, ndc: bool, near_ndc=0.0, far_ndc=1.0) -> dict:
    """The input dictionary of a full-frame render, flattened to (h*w, .); create_test_data :807-895
    (single pose, no view pose, no secondary poses; ``pose`` already processed)."""
    intrinsic = numpy.asarray(intrinsic).astype('float32')
    rays_o, rays_d = camera_rays(resolution, intrinsic, pose)
    view_dirs = unit_dirs(rays_d)
    ones = numpy.ones_like(rays_d[..., :1])
    batch = {
        'rays_o': numpy.ascontiguousarray(rays_o).reshape(-1, 3),
        'rays_d': rays_d.reshape(-1, 3),
        'view_dirs': view_dirs.reshape(-1, 3),
        'near': (near * ones).reshape(-1, 1),
        'far': (far * ones).reshape(-1, 1),
    }
    if ndc:
        o_ndc, d_ndc = ndc_rays(rays_o, rays_d, resolution, intrinsic, near)
        batch['rays_o_ndc'] = o_ndc.reshape(-1, 3)
        batch['rays_d_ndc'] = d_ndc.reshape(-1, 3)
        batch['near_ndc'] = (near_ndc * ones).reshape(-1, 1)
        batch['far_ndc'] = (far_ndc * ones).reshape(-1, 1)
    return batch


def to_display(rgb: numpy.ndarray, depth: numpy.ndarray):
    """post_process_image / post_process_depth :1106-1114: clip to [0,1], round(255x) -> uint8; depth clip >= 0.
    Pinned by tests/golden/display.npz (the reference's two functions on ties, out-of-range values, infinities, NaN)."""
    with numpy.errstate(invalid='ignore'):
        img = numpy.round(numpy.clip(rgb, 0, 1) * 255).astype('uint8')
    dep = numpy.clip(depth, 0, numpy.inf).astype('float32')
    return img, dep


def retrieve_inference_outputs(configs: dict, resolution, network_outputs: dict) -> dict:
    """DataPreprocessor.retrieve_inference_outputs :897-925: of all network outputs keep rgb / depth / depth_var
    (+ depth_ndc / depth_var_ndc when ndc) of the finest level present, as a display image and clipped depth maps."""
    h, w = resolution
    suffix = '_fine' if 'fine_mlp' in configs['model'] else '_coarse'
    out = {'image': to_display(numpy.asarray(network_outputs[f'rgb{suffix}']).reshape(h, w, 3), numpy.zeros(1))[0]}
    for name in ['depth', 'depth_var'] + (['depth_ndc', 'depth_var_ndc'] if configs['data_loader']['ndc'] else []):
        out[name] = to_display(numpy.zeros(3), numpy.asarray(network_outputs[f'{name}{suffix}']).reshape(h, w))[1]
    return out
