"""Training-batch assembly on the device behind the reference's ``get_next_batch`` interface.

``BatchAssembler(configs, scene).get_next_batch(iter_num, image_num=None)`` returns the dictionary the reference's
``DataPreprocessor.get_next_batch`` returns in cached-batching train mode (src/data_preprocessors/DataPreprocessor01.py
:507-551): ``indices``, ``indices_mask_nerf`` [, ``indices_mask_sparse_depth``], ``iter_num``, ``num_frames``, ``rays_o``,
``rays_d``, ``view_dirs``, ``pixel_id``, ``target_rgb``, ``near``, ``far`` [, ``rays_o_ndc``, ``rays_d_ndc``,
``near_ndc``, ``far_ndc``] [, ``sparse_depth_values``, ``sparse_depth_errors``, ``sparse_depth_values_ndc``] and
``common_data`` = {poses, intrinsics, images (each with a leading per-GPU axis), resolution}.

``scene`` is what the reference's preprocessing leaves in ``preprocessed_data_dict`` -- dataset loading, pose
recentring and sparse-depth rasterisation stay outside this build (out of scope):
    poses (V,4,4) processed camera-to-world, intrinsics (V,3,3), images (V,h,w,3) float in [0,1], resolution (h,w),
    near, far [, near_ndc, far_ndc], frame_nums (V,), and optionally the dense (V*h*w,) tables sparse_depths,
    sparse_errors, sparse_depths_ndc (<= 0 / -1 where a pixel has no sparse depth).

What differs from the reference, by design:
  * no ray cache: rows are recomputed from the cameras in the assembly kernel (bit-identical values);
  * the epoch order is a keyed on-the-fly permutation (snerf_shuffled_indices) seeded by ``configs['seed']``
    (default 0), not numpy's global Mersenne Twister; pass ``indices=`` (and ``indices_sparse=``) to replay a given
    order, e.g. the reference's;
  * with several ranks, ``rank``/``world_size`` give each rank a contiguous slice of every batch of the SAME global
    stream, so the union over ranks is the single-process batch; the returned ``global_rows`` (int64, one per row) say
    where each of the rank's rows sits in that single-process batch (pixel rows first, then sparse rows), which is what
    the renderer keys its training draws on -- every ray gets the jitter and density noise it would get in a
    single-process run, whatever the number of ranks and however the trainer cuts the batch into sub-batches.
Reference behaviour kept: batches are consecutive slices of the epoch order and the order is renewed when a slice
reaches the end (:559-563, a short last batch included); a full-image request returns pixel rays only (:564-567);
during pre-cropping the candidates are the central window (:258-268) -- and stay so for the whole run, because the
reference discards the regenerated list at ``precrop_iterations`` (:557-558; tests/golden/batch_assembly.npz
``after_precrop_count``).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops

Tensor = torch.Tensor


class BatchAssembler:
    def __init__(self, configs: dict, scene: dict, device='cuda:0', rank: int = 0, world_size: int = 1):
        loader = configs['data_loader']
        self.configs = configs
        self.device = torch.device(device)
        self.ndc = bool(loader['ndc'])
        self.num_rays = int(loader['num_rays'])
        self.sparse_depth_needed = 'sparse_depth' in loader
        self.num_rays_sparse_depth = int(loader['sparse_depth']['num_rays']) if self.sparse_depth_needed else 0
        self.seed = int(configs.get('seed', 0))
        self.rank, self.world_size = int(rank), int(world_size)
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).to(self.device).contiguous()
        self.poses, self.intrinsics, self.images = t(scene['poses']), t(scene['intrinsics']), t(scene['images'])
        self.resolution = (int(scene['resolution'][0]), int(scene['resolution'][1]))
        self.num_views = self.poses.shape[0]
        h, w = self.resolution
        if tuple(self.images.shape) != (self.num_views, h, w, 3):
            raise RuntimeError(f'images {tuple(self.images.shape)} do not match {self.num_views} views of {h}x{w}')
        self.near, self.far = float(scene['near']), float(scene['far'])
        self.near_ndc, self.far_ndc = float(scene.get('near_ndc', 0.0)), float(scene.get('far_ndc', 1.0))
        self.frame_nums = [int(f) for f in scene.get('frame_nums', range(self.num_views))]
        self.table = ops.camera_table(self.intrinsics, self.poses, self.resolution)
        self.sparse = {k: (t(scene[k]).reshape(-1) if scene.get(k) is not None else None)
                       for k in ('sparse_depths', 'sparse_errors', 'sparse_depths_ndc')}
        if self.sparse_depth_needed:
            if self.sparse['sparse_depths'] is None:
                raise RuntimeError("configs['data_loader']['sparse_depth'] is set but the scene has no 'sparse_depths' table")
            # candidates = pixels with a sparse depth (numpy.where(sparse_depths > 0), :441); built once at start-up
            self.sparse_candidates = torch.nonzero(self.sparse['sparse_depths'] > 0).reshape(-1).contiguous()
        fraction, iterations = loader.get('precrop_fraction', 1), loader.get('precrop_iterations', 0)
        self.crop = (0, h, 0, w)
        if fraction < 1 and 0 < iterations:      # generate_indices(iter_num=0), :258-265
            self.crop = (int(round(h / 2 * (1 - fraction))), int(round(h / 2 * (1 + fraction))),
                         int(round(w / 2 * (1 - fraction))), int(round(w / 2 * (1 + fraction))))
        self.domain = self.num_views * (self.crop[1] - self.crop[0]) * (self.crop[3] - self.crop[2])
        self.i_batch = self.epoch = 0
        self.i_batch_sparse_depth = self.epoch_sparse = 0

    # --------------------------------------------------------------------------------------------------
    def _next_slice(self, cursor: int, epoch: int, size: int, domain: int):
        """(first, count, new cursor, new epoch): indices[cursor:cursor+size] of the epoch order, then the wrap rule."""
        first, count = cursor, max(0, min(size, domain - cursor))
        cursor += size
        if cursor >= domain:
            cursor, epoch = 0, epoch + 1
        return first, count, cursor, epoch

    def _shard(self, first: int, count: int):
        """This rank's part of a slice of ``count`` positions starting at ``first``: (first, count, offset in the slice)."""
        per = -(-count // self.world_size)
        lo = min(count, self.rank * per)
        return first + lo, min(count, lo + per) - lo, lo

    def next_positions(self) -> Dict[str, tuple]:
        """Advance the epoch cursors by one batch (reference :559-563) and return where it sits: ``{'pixel': (epoch, first,
        count)[, 'sparse': (epoch, first, count)]}`` -- ``count`` is shorter than the configured batch at the end of an epoch."""
        first, total, self.i_batch, epoch = self._next_slice(self.i_batch, self.epoch, self.num_rays, self.domain)
        positions = {'pixel': (self.epoch, first, total)}
        self.epoch = epoch
        if self.sparse_depth_needed:
            domain = self.sparse_candidates.shape[0]
            first, total_sd, self.i_batch_sparse_depth, epoch = self._next_slice(self.i_batch_sparse_depth, self.epoch_sparse,
                                                                                 self.num_rays_sparse_depth, domain)
            positions['sparse'] = (self.epoch_sparse, first, total_sd)
            self.epoch_sparse = epoch
        return positions

    def is_full(self, positions: Dict[str, tuple]) -> bool:
        return positions['pixel'][2] == self.num_rays and (not self.sparse_depth_needed or
                                                           positions['sparse'][2] == self.num_rays_sparse_depth)

    def select_batch_indices(self, iter_num: int, image_num: Optional[int] = None, positions: Optional[Dict[str, tuple]] = None,
                             at: Optional[Tensor] = None):
        """-> (indices int64 GPU tensor, number of pixel-ray rows, sparse rows present?, (first global pixel row, first global
        sparse row))   (reference :553-584).  ``positions``: where the batch sits (default: the next one, advancing the
        cursors).  ``at``: the device-resident iteration record to read epoch and position from instead (a FULL batch inside a
        captured graph: the counts are the configured ones)."""
        h, w = self.resolution
        if image_num is not None:
            image_index = self.frame_nums.index(int(image_num))
            first, count, lo = self._shard(image_index * h * w, h * w)
            return torch.arange(first, first + count, dtype=torch.int64, device=self.device), count, False, (lo, lo + count)
        if at is not None:
            epoch, first, total = 0, 0, self.num_rays
        else:
            positions = positions if positions is not None else self.next_positions()
            epoch, first, total = positions['pixel']
        first, count, lo = self._shard(first, total)
        indices = ops.shuffled_indices(self.seed, epoch, first, count, self.domain, self.device, num_views=self.num_views,
                                       resolution=self.resolution, crop=self.crop, at=at)
        if not self.sparse_depth_needed:
            return indices, count, False, (lo, lo + count)
        domain = self.sparse_candidates.shape[0]
        epoch, first, total_sd = (0, 0, self.num_rays_sparse_depth) if at is not None else positions['sparse']
        first, n_sd, lo_sd = self._shard(first, total_sd)
        sparse = ops.shuffled_indices(self.seed + 1, epoch, first, n_sd, domain, self.device, candidates=self.sparse_candidates,
                                      at=at, sparse=True)
        # global rows: the single-process batch is [all pixel rows (total) | all sparse rows (total_sd)]
        return torch.cat([indices, sparse]), count, True, (lo, total + lo_sd)

    def get_next_batch(self, iter_num: int, image_num: Optional[int] = None, indices: Optional[Tensor] = None,
                       indices_sparse: Optional[Tensor] = None, positions: Optional[Dict[str, tuple]] = None,
                       at: Optional[Tensor] = None) -> Dict[str, object]:
        if indices is not None:     # replay a given order (parity with the reference's numpy stream)
            with_sparse = indices_sparse is not None
            num_pixel = indices.shape[0]
            indices = torch.cat([indices, indices_sparse]) if with_sparse else indices
            indices = indices.to(self.device, torch.int64)
            first_rows = (0, num_pixel)
        else:
            indices, num_pixel, with_sparse, first_rows = self.select_batch_indices(iter_num, image_num, positions, at)
        sp = self.sparse if with_sparse else {k: None for k in self.sparse}
        batch = ops.assemble_batch(indices, num_pixel, self.table, self.resolution, self.images, self.ndc, self.near, self.far,
                                   self.near_ndc, self.far_ndc, sp['sparse_depths'], sp['sparse_errors'],
                                   sp['sparse_depths_ndc'] if self.ndc else None, with_sparse_mask=with_sparse,
                                   first_pixel_row=first_rows[0], first_sparse_row=first_rows[1])
        out: Dict[str, object] = {'common_data': {}, 'indices': indices, 'indices_mask_nerf': batch.pop('indices_mask_nerf')}
        if with_sparse:
            out['indices_mask_sparse_depth'] = batch.pop('indices_mask_sparse_depth')
        out.update(iter_num=iter_num, num_frames=self.num_views)
        out.update(batch)
        # shared tensors with the leading per-GPU axis of the reference (:545-550); a view, not a copy
        out['common_data'] = {'poses': self.poses[None], 'intrinsics': self.intrinsics[None], 'images': self.images[None],
                              'resolution': self.resolution}
        return out
