"""Minimal counterpart of the reference's Tester call path around the model (last row):
full-frame input batch -> ``model(batch)`` under no_grad -> display post-processing, plus the ray-sharded
multi-GPU variant (one process per GPU, contiguous block of rays per rank, one gather of the per-ray outputs).

Reference call sites restated here: DataPreprocessor.create_test_data (src/data_preprocessors/DataPreprocessor01.py
:807-895), NerfTester.predict_frame (src/Tester01.py:57-66), post_process_image/depth (:1106-1114).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import torch

from . import ops
from .models.SimpleNeRFHip01 import global_rows

Tensor = torch.Tensor
DEFAULT_KEYS = ('rgb_fine', 'depth_fine', 'depth_var_fine')
# Stream-capture mode of the two graphed steps below.  torch's default, 'global', makes EVERY thread's capture-unsafe HIP call
# fail while the capture is open -- and with a process group alive, torch.distributed's NCCL watchdog thread polls the events of
# earlier collectives (hipEventQuery) whenever it likes: if a poll falls inside the capture window the watchdog dies with
# hipErrorStreamCaptureUnsupported and takes the process with it (round 5: one of three `bench.py --extras` runs, in the
# rank-share leg, right after an eager all-reduce).  'thread_local' confines the check to the capturing thread, whose calls are
# the stream-ordered launches the capture is there to record.
CAPTURE_MODE = 'thread_local'


def shard_range(num_rays: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns rays [r*ceil(N/R), min(N, (r+1)*ceil(N/R)))  (SURVEY 8e)."""
    per = -(-num_rays // world_size)
    first = min(num_rays, rank * per)
    return first, min(num_rays, first + per) - first


_columns: Dict[tuple, Tensor] = {}


def _constant_column(value: float, rows: int, device: torch.device) -> Tensor:
    """(rows,1) tensor filled with ``value`` (the near/far columns of a frame's batch are constants: built once per
    value and size, shared read-only between batches instead of being re-filled every step)."""
    key = (value, rows, device)
    col = _columns.get(key)
    if col is None:
        if len(_columns) > 64:
            _columns.clear()
        col = _columns[key] = torch.full((rows, 1), value, dtype=torch.float32, device=device)
    return col


def frame_batch(camera: dict, ndc: bool, device, first_ray: int = 0, num_rays: Optional[int] = None) -> Dict[str, Tensor]:
    """Input dictionary for rays [first_ray, first_ray+num_rays) of a frame, generated on the device (K1)."""
    h, w = camera['resolution']
    if num_rays is None:
        num_rays = h * w - first_ray
    batch = ops.generate_rays((h, w), camera['intrinsic'], camera['pose'], camera['near'], ndc, device, first_ray, num_rays)
    col = lambda v: _constant_column(float(v), num_rays, torch.device(device))
    batch['near'], batch['far'] = col(camera['near']), col(camera['far'])
    if ndc:
        batch['near_ndc'], batch['far_ndc'] = col(camera.get('near_ndc', 0.0)), col(camera.get('far_ndc', 1.0))
    return batch


@torch.no_grad()
def render_rays_blockwise(model, camera: dict, ndc: bool, device, first_ray: int, num_rays: int,
                          keys: Iterable[str] = DEFAULT_KEYS, ray_block: int = 65536) -> Dict[str, Tensor]:
    """Render a ray range in blocks (bounding the per-sample buffers), keeping only ``keys``."""
    keys = tuple(keys)
    pieces = {k: [] for k in keys}
    # a rank that owns no rays (ceil-division shards of a tiny frame) still runs one zero-ray call, so that its
    # outputs have the right trailing shapes for the gather
    for start in (range(first_ray, first_ray + num_rays, ray_block) if num_rays > 0 else (first_ray,)):
        count = min(ray_block, first_ray + num_rays - start)
        out = model(frame_batch(camera, ndc, device, start, count))
        for k in keys:
            pieces[k].append(out[k])
    return {k: torch.cat(v, 0) for k, v in pieces.items()}


def gather_rays(local: Dict[str, Tensor], num_rays: int, rank: int, world_size: int, dst: int = 0,
                group=None) -> Optional[Dict[str, Tensor]]:
    """One gather of the per-ray outputs of every rank's block to ``dst`` (all keys packed into one buffer, so it
    is a single collective).  Returns the full-frame tensors on ``dst`` and None elsewhere."""
    import torch.distributed as dist
    per = -(-num_rays // world_size)
    keys = sorted(local)
    # (rows, width) view of each key; a zero-row tensor keeps its trailing shape (reshape(0, -1) is ambiguous)
    cols = [local[k].reshape(local[k].shape[0], int(torch.Size(local[k].shape[1:]).numel())) for k in keys]
    widths = [c.shape[1] for c in cols]
    packed = torch.cat(cols, 1) if cols else torch.empty((0, 0))
    pad = torch.zeros((per, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    pad[:packed.shape[0]] = packed
    bucket = [torch.empty_like(pad) for _ in range(world_size)] if rank == dst else None
    dist.gather(pad, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.cat(bucket, 0)[:num_rays]
    out, c0 = {}, 0
    for k, wdt in zip(keys, widths):
        out[k] = full[:, c0:c0 + wdt].reshape((num_rays,) + tuple(local[k].shape[1:]))
        c0 += wdt
    return out


@torch.no_grad()
def render_frame(model, camera: dict, ndc: bool, device, keys: Iterable[str] = DEFAULT_KEYS, rank: int = 0,
                 world_size: int = 1, ray_block: int = 65536, collective: Optional[bool] = None) -> Optional[Dict[str, Tensor]]:
    """Full frame, (h*w, .) per key.  With world_size > 1 every rank renders its block and rank 0 receives the
    frame through one gather (returns None on the other ranks).  ``collective=True`` issues that gather with a single rank
    too (a one-rank process group: the RCCL call path of the multi-GPU run on a one-GPU box)."""
    h, w = camera['resolution']
    n = h * w
    first, count = shard_range(n, rank, world_size)
    local = render_rays_blockwise(model, camera, ndc, device, first, count, keys, ray_block)
    if not (world_size > 1 if collective is None else collective):
        return local
    return gather_rays(local, n, rank, world_size)


def to_display(rgb: Tensor, depth: Tensor):
    """clip -> round(255 x) -> uint8 colour; depth clipped at 0 (post_process_image / post_process_depth), fused on
    the device so that a frame is copied to the host as 3 B/pixel."""
    return ops.to_display(rgb, depth)


def retrieve_inference_outputs(configs: dict, resolution, network_outputs: Dict[str, Tensor]) -> dict:
    """What leaves the device after a frame and in what form: DataPreprocessor.retrieve_inference_outputs
    (src/data_preprocessors/DataPreprocessor01.py:897-925).  The reference copies EVERY network output to the host
    (post_process_output, incl. ~1 KB/ray of alpha) and then keeps five of them; here only those five are converted on the
    device (K: snerf_to_display -- uint8 colour, depths clipped at 0) and copied: 3 + 16 B per pixel.

    Returns numpy arrays under the reference's keys, in its order: ``image`` (h,w,3) uint8, ``depth``, ``depth_var``
    (h,w) float32 and, for NDC scenes, ``depth_ndc``, ``depth_var_ndc``; the suffix is ``_fine`` when the model has a
    fine MLP, else ``_coarse`` (:900-905); ``visibility2`` (nf-1, h, w) float32 when the outputs carry it
    (predict_visibility with secondary views, :919-924)."""
    h, w = int(resolution[0]), int(resolution[1])
    if 'fine_mlp' in configs['model']:
        suffix = '_fine'
    elif 'coarse_mlp' in configs['model']:
        suffix = '_coarse'
    else:
        raise RuntimeError('retrieve_inference_outputs: the model has neither a fine nor a coarse MLP')
    image, depth = ops.to_display(network_outputs[f'rgb{suffix}'].reshape(h * w, 3), network_outputs[f'depth{suffix}'].reshape(h * w))
    out = {'image': image.reshape(h, w, 3), 'depth': depth.reshape(h, w)}
    names = ['depth_var'] + (['depth_ndc', 'depth_var_ndc'] if configs['data_loader']['ndc'] else [])
    for name in names:
        out[name] = ops.to_display(network_outputs[f'rgb{suffix}'].reshape(h * w, 3),
                                   network_outputs[f'{name}{suffix}'].reshape(h * w), colour=False)[1].reshape(h, w)
    result = {k: v.cpu().numpy() for k, v in out.items()}
    if f'visibility2{suffix}' in network_outputs:
        vis2 = network_outputs[f'visibility2{suffix}'].reshape(h, w, -1).permute(2, 0, 1).contiguous()
        result['visibility2'] = vis2.float().cpu().numpy()         # post_process_visibility: a float32 cast (:1116-1119)
    return result


@torch.no_grad()
def predict_frame(model, configs: dict, camera: dict, device, rank: int = 0, world_size: int = 1,
                  ray_block: int = 65536, collective: Optional[bool] = None) -> Optional[dict]:
    """NerfTester.predict_frame (src/Tester01.py:57-66): full-frame batch -> model under no_grad -> the five display
    outputs.  Rays are generated on the device per block; with world_size > 1 each rank renders its block of the frame
    and rank 0 receives it through one gather (None on the other ranks)."""
    ndc = bool(configs['data_loader']['ndc'])
    suffix = '_fine' if 'fine_mlp' in configs['model'] else '_coarse'
    keys = [f'rgb{suffix}', f'depth{suffix}', f'depth_var{suffix}'] + ([f'depth_ndc{suffix}', f'depth_var_ndc{suffix}'] if ndc else [])
    frame = render_frame(model, camera, ndc, device, keys, rank, world_size, ray_block, collective)
    if frame is None:
        return None
    return retrieve_inference_outputs(configs, camera['resolution'], frame)


def allreduce_gradients(parameters, world_size: int, group=None, force: bool = False) -> None:
    """Training with rays sharded over ranks: average every parameter gradient across ranks with ONE collective
    (all gradients flattened into a single buffer: 2 265 488 floats = 9.06 MB for the 4-MLP model).
    Equivalent to the reference's DataParallel, which gathers the per-device outputs and takes the loss mean over the
    whole batch (src/Trainer01.py:93-96), when every rank holds the same number of rays.  ``force``: issue the collective
    with a single rank too (one-rank process group -- the RCCL call path on a one-GPU box)."""
    if world_size == 1 and not force:
        return
    import torch.distributed as dist
    # EVERY trainable parameter takes part, zeros standing in for a missing gradient: a rank whose shard of a short last
    # batch is empty (or whose loss terms skipped an MLP) must still flatten the same layout as its peers
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    pieces = [piece.view_as(p) for piece, p in zip(flat.split([p.numel() for p in params]), params)]
    held = [(p, piece) for p, piece in zip(params, pieces) if p.grad is not None]
    if held:     # one multi-tensor copy back into the gradient buffers (their addresses stay: they may be a graph's static ones)
        torch._foreach_copy_([p.grad for p, _ in held], [piece for _, piece in held])
    for p, piece in zip(params, pieces):
        if p.grad is None:
            p.grad = piece.clone()


def train_one_iter(model, loss_computer, optimizer, input_batch: dict, sub_batch_size: Optional[int] = None,
                   world_size: int = 1, group=None, single_pass: bool = False, force_collective: bool = False) -> Dict[str, Tensor]:
    """One optimisation step over ``input_batch`` the way the reference's trainer does it (Trainer.train_one_iter,
    src/Trainer01.py:60-107): gradients cleared, the batch cut into consecutive sub-batches, ``model`` ->
    ``compute_losses`` -> ``TotalLoss.backward()`` per sub-batch (gradients accumulate: the step minimises the SUM of
    the sub-batch means), one ``optimizer.step()``.  With ``world_size`` > 1 the accumulated gradients are averaged
    over ranks by one all-reduce before the step.  Returns the summed loss values as device tensors (the reference
    calls ``.item()`` on each, a host sync per loss per sub-batch; the caller decides when to read them).

    ``single_pass``: the same objective -- every loss still normalised over its own sub-batch, the sub-batch totals
    summed -- with ONE model forward/backward over the whole batch instead of one per sub-batch.  The reference cuts the
    batch because of device memory; the renderer treats rays independently, so evaluating them together changes nothing
    but the number of launches and the gradient-accumulation adds (4096 rows: 10.2 -> 9.1 ms in the 16-bit mode).  Only
    the rounding of the sums differs from the sub-batched run (the draws are keyed by iteration and global row)."""
    optimizer.zero_grad(set_to_none=True)
    n = input_batch['rays_o'].shape[0]
    sub = int(sub_batch_size or n)
    totals: Dict[str, Tensor] = {}
    # global rows (keys of the training draws): the batch's per-row ``global_rows`` tensor is sliced with the other
    # tensors; without it the rows are row_offset + r
    base = int(input_batch.get('row_offset', 0))
    if single_pass and sub < n:
        whole = dict(input_batch)
        output = model(whole)
        # one split per output tensor: its backward is a single concatenation of the sub-batch gradients (slicing per
        # sub-batch would zero-fill and add a full-size gradient per slice)
        pieces = {k: (v.split(sub) if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else None)
                  for k, v in output.items()}
        objective = None
        for index, start in enumerate(range(0, n, sub)):
            piece = {}
            for key, value in input_batch.items():
                if isinstance(value, torch.Tensor):
                    piece[key] = value[start:start + sub]
                elif key == 'common_data':
                    piece[key] = dict(value)
                else:
                    piece[key] = value
            piece['row_offset'] = base + start
            out_piece = {k: (pieces[k][index] if pieces[k] is not None else v) for k, v in output.items()}
            losses = loss_computer.compute_losses(piece, out_piece)
            objective = losses['TotalLoss'] if objective is None else objective + losses['TotalLoss']
            for name, entry in losses.items():
                value = entry['loss_value'] if isinstance(entry, dict) else entry
                value = value.detach() if isinstance(value, torch.Tensor) else torch.as_tensor(float(value))
                totals[name] = totals[name] + value if name in totals else value
        objective.backward()
        allreduce_gradients(model.parameters(), world_size, group, force_collective)
        optimizer.step()
        return totals
    for start in range(0, n, sub):
        piece = {}
        for key, value in input_batch.items():
            if isinstance(value, torch.Tensor):
                piece[key] = value[start:start + sub]
            elif key == 'common_data':
                piece[key] = dict(value)
            else:
                piece[key] = value
        piece['row_offset'] = base + start
        losses = loss_computer.compute_losses(piece, model(piece))
        losses['TotalLoss'].backward()
        for name, entry in losses.items():
            value = entry['loss_value'] if isinstance(entry, dict) else entry
            value = value.detach() if isinstance(value, torch.Tensor) else torch.as_tensor(float(value))
            totals[name] = totals[name] + value if name in totals else value
    allreduce_gradients(model.parameters(), world_size, group, force_collective)
    optimizer.step()
    return totals


class GraphedTrainStep:
    """The device work of one training pass -- weight re-pack, every MLP forward, compositing and resampling, the loss
    table, and the whole backward down to the parameter gradients -- captured ONCE as a HIP graph and replayed per
    iteration (``torch.cuda.graph``; every kernel of the library only enqueues on the current stream, so the capture
    sees them like torch's own).  What changes per iteration stays outside the graph and is written into its static
    inputs first: the batch (copied in), the Philox draws (generated in place by ``model.draw_training_randomness``)
    and, after the replay, the optimiser step.  ~150 launches and their Python glue become one graph launch: the host
    cost of an iteration drops from ~5.4 ms to well under 1 ms, which is what bounds small per-rank batches.

    ``sub_batch_size`` (round 2): the captured work is the reference's iteration exactly -- the batch cut into consecutive
    sub-batches, model -> losses -> backward per sub-batch, gradients accumulated by the backward kernels
    (src/Trainer01.py:82-96) -- so a replay is bit-identical to ``train_one_iter(..., sub_batch_size)`` (same draws too:
    they are generated per sub-batch, each counted as one training forward).  Without it the whole batch is one pass:
    the MSE, sparse-depth and augmentation-depth terms come out the same, the coarse-fine consistency term's patch
    statistics see the whole batch at once and differ by ~10 % (tools/probes/sub_batch_terms.py).  ``p.grad`` of every parameter is a static
    buffer the graph overwrites each replay: do not call ``zero_grad(set_to_none=True)`` between iterations.  The loss
    weights of the iteration are baked into the graph; it is re-captured when ``LossComputer.get_loss_weight`` changes
    them (the shipped schedule: once, at iteration 10000).
    """

    def __init__(self, model, loss_computer, sample_batch: Dict[str, object], warmup: int = 2,
                 sub_batch_size: Optional[int] = None):
        self.model, self.losses = model, loss_computer
        self.static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample_batch.items() if k != 'common_data'}
        self.common = dict(sample_batch.get('common_data', {}))
        self.n = self.static['rays_o'].shape[0]
        self.device = self.static['rays_o'].device
        self.draws = {k: torch.empty(shape, dtype=torch.float32, device=self.device)
                      for k, shape in model.training_draw_shapes(self.n).items()}
        self.warmup = warmup
        self.sub = int(sub_batch_size) if sub_batch_size else self.n
        self.graph = None
        self.weights_key = None
        self.totals: Dict[str, Tensor] = {}

    @staticmethod
    def _rows(batch, n):
        first, rows = global_rows(batch, n)
        return rows if rows is not None else first

    def _weights(self, iter_num):
        return tuple(self.losses.get_loss_weight(cfg, iter_num) for cfg in self.losses.losses.values())

    def _pass(self):
        totals: Dict[str, Tensor] = {}
        for start in range(0, self.n, self.sub):
            cut = slice(start, start + self.sub)
            self.model.set_random_draws({k: v[cut] for k, v in self.draws.items()})
            piece = {k: (v[cut] if isinstance(v, torch.Tensor) else v) for k, v in self.static.items()}
            piece['common_data'] = dict(self.common)
            losses = self.losses.compute_losses(piece, self.model(piece))
            losses['TotalLoss'].backward()
            for name, entry in losses.items():
                value = (entry['loss_value'] if isinstance(entry, dict) else entry).detach()
                totals[name] = totals[name] + value if name in totals else value
        return totals

    def _draw(self, batch):
        """This iteration's draws into the static buffers: per sub-batch, each counted as one training forward -- what the
        eager sub-batched iteration draws."""
        rows = self._rows(batch, self.n)
        for start in range(0, self.n, self.sub):
            count = min(self.sub, self.n - start)
            part = rows[start:start + count] if isinstance(rows, torch.Tensor) else rows + start
            self.model.draw_training_randomness(count, part, self.device, out={k: v[start:start + count] for k, v in self.draws.items()},
                                                iter_num=batch.get('iter_num'))

    def _capture(self):
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):          # eager warm-up on a side stream, as torch's graph capture requires
            for _ in range(self.warmup):
                self._pass()
        torch.cuda.current_stream(self.device).wait_stream(side)
        for p in self.model.parameters():
            p.grad = None                      # gradients are (re)allocated inside the capture: static addresses
        self.model.invalidate_packed()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
            self.totals = self._pass()

    def __call__(self, batch: Dict[str, object]) -> Dict[str, Tensor]:
        rows = batch['rays_o'].shape[0]
        if rows != self.n:
            # a short batch at the end of an epoch (the reference's slicing, DataPreprocessor01.py:559-563): the same
            # sub-batched pass as train_one_iter (ceil(rows / sub) model -> losses -> backward rounds, the sub-batch
            # totals summed, draws per sub-batch), launched directly, accumulating into the (zeroed) static gradient
            # buffers so that their addresses survive
            for p in self.model.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            all_rows = self._rows(batch, rows)
            totals: Dict[str, Tensor] = {}
            for start in range(0, rows, self.sub):
                count = min(self.sub, rows - start)
                part = all_rows[start:start + count] if isinstance(all_rows, torch.Tensor) else all_rows + start
                self.model.set_random_draws(self.model.draw_training_randomness(count, part, self.device, iter_num=batch.get('iter_num')))
                piece = {k: (v[start:start + count] if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
                piece['common_data'] = dict(batch.get('common_data', {}))
                losses = self.losses.compute_losses(piece, self.model(piece))
                losses['TotalLoss'].backward()
                for name, entry in losses.items():
                    value = (entry['loss_value'] if isinstance(entry, dict) else entry).detach()
                    totals[name] = totals[name] + value if name in totals else value
            return totals
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                self.static[k].copy_(v)
            elif k != 'common_data':
                self.static[k] = v
        self._draw(batch)
        key = self._weights(batch['iter_num'])
        if self.graph is None or key != self.weights_key:
            self._capture()                    # records the work (on these inputs); nothing is computed until the replay
            self.weights_key = key
        self.graph.replay()
        # a replay bypasses the entry points that report an fp16 range violation of an earlier launch: ask here (a host read
        # of a pinned word, no synchronisation -- like the entry points it sees launches that have finished)
        if self.model.precision in ops.FP16_RANGE_PRECISIONS and ops.range_status(clear=True):
            raise ops.Fp16RangeError("GraphedTrainStep: an earlier replay met a value outside the fp16 range (|v| > 65504); its "
                                     "gradients are invalid -- use hip_precision 'fp32' for this model")
        return self.totals


class GraphedIteration:
    """The WHOLE training iteration as ONE HIP graph (round 3): batch assembly (index permutation + ray
    assembly), the training draws, weight re-pack, every MLP forward, compositing / resampling, the loss table, the complete
    backward AND the Adam update -- ``GraphedTrainStep`` replayed only the pass in between, with batch copy, draws and optimiser
    enqueued from Python around it.  What changes from iteration to iteration cannot be a kernel argument of a replayed graph,
    so it lives in device memory: the graph's first node (``snerf_iteration_advance``) copies the next record of a pinned host
    ring -- epoch positions of the pixel and sparse-depth slices, the iteration number that keys the Philox draws, Adam's two
    step-dependent factors with the decayed learning rate folded in -- into the device-resident record that
    ``snerf_shuffled_indices_at`` / ``snerf_random_*_at`` / ``snerf_adam_step_at`` read.  Per iteration the host writes one
    64-byte record and launches one graph: parameters after N iterations are bit-identical to N eager
    ``train_one_iter`` iterations (tests/test_gpu_optim.py).

    A short batch at the end of an epoch (other tensor shapes) runs eagerly into the graph's static gradient buffers, as
    ``GraphedTrainStep`` does; the graph is re-captured when ``LossComputer.get_loss_weight`` changes the loss weights.  The host
    may run at most ``slots - 1`` replays ahead of the device: the slot about to be overwritten is guarded by an event.

    Several ranks (round 4): ``batcher`` built with (rank, world_size) hands every rank its slice of the same global batch, and
    the ONE all-reduce of the flattened gradients (``allreduce_gradients``) is captured between the last sub-batch's backward
    and the Adam update -- RCCL collectives record into a HIP graph like kernels, so a replay still costs the host one record
    and one launch.  It needs the ``nccl`` backend (gloo's collectives run on the host and cannot be captured);
    ``force_collective`` captures the all-reduce with a single rank too (the test of exactly that on a one-GPU box)."""

    def __init__(self, model, loss_computer, optimizer, batcher, lr_decayer=None, sub_batch_size: Optional[int] = None,
                 slots: int = 8, warmup: int = 1, group=None, force_collective: bool = False):
        self.world = int(getattr(batcher, 'world_size', 1))
        self.group, self.force_collective = group, bool(force_collective)
        if self.world > 1 or self.force_collective:
            import torch.distributed as dist
            if not dist.is_initialized() or dist.get_backend(group) != 'nccl':
                raise NotImplementedError("GraphedIteration over several ranks captures the gradient all-reduce: it needs an initialised "
                                          "'nccl' (RCCL) process group -- with gloo use GraphedTrainStep + allreduce_gradients")
        if len(optimizer.param_groups) != 1:
            raise NotImplementedError('GraphedIteration replays ONE (step size, bias correction) record: a single parameter group')
        self.model, self.losses, self.opt, self.batcher, self.decayer = model, loss_computer, optimizer, batcher, lr_decayer
        self.device = batcher.device
        self.ring = ops.IterationRing(self.device, slots)
        self.events = [None] * slots
        self.sub = int(sub_batch_size) if sub_batch_size else None
        self.warmup = warmup
        self.graph = None
        self.weights_key = None
        self.totals: Dict[str, Tensor] = {}
        self.wait_seconds = 0.0           # time the host spent waiting for a ring slot (it was >= slots - 1 replays ahead)

    def _weights(self, iter_num):
        return tuple(self.losses.get_loss_weight(cfg, iter_num) for cfg in self.losses.losses.values())

    def _passes(self, batch, draws_at) -> Dict[str, Tensor]:
        """model -> losses -> backward per sub-batch (gradients accumulate in the kernels), totals summed."""
        n = batch['rays_o'].shape[0]
        sub = self.sub or n
        rows = batch['global_rows']
        totals: Dict[str, Tensor] = {}
        for start in range(0, n, sub):
            count = min(sub, n - start)
            self.model.set_random_draws(self.model.draw_training_randomness(count, rows[start:start + count], self.device,
                                                                            iter_num=batch.get('iter_num'), at=draws_at))
            piece = {k: (v[start:start + count] if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            piece['common_data'] = dict(batch.get('common_data', {}))
            losses = self.losses.compute_losses(piece, self.model(piece))
            losses['TotalLoss'].backward()
            for name, entry in losses.items():
                value = (entry['loss_value'] if isinstance(entry, dict) else entry).detach()
                totals[name] = totals[name] + value if name in totals else value
        return totals

    def _body(self, iter_num: int, whole: bool) -> Dict[str, Tensor]:
        if whole:
            self.ring.advance()
        batch = self.batcher.get_next_batch(iter_num, at=self.ring.current)
        totals = self._passes(batch, self.ring.current)
        if not whole:      # the eager warm-up: which trainable parameters does the backward leave without a gradient?
            self._gradless = [p for p in self.model.parameters() if p.requires_grad and p.grad is None]
        allreduce_gradients(self.model.parameters(), self.world, self.group, self.force_collective)
        if whole:
            self.opt.step_at(self.ring.current)
        return totals

    def _capture(self, iter_num: int):
        # optimiser state must exist BEFORE the capture (created inside it, its zero fill would be replayed every iteration)
        for group in self.opt.param_groups:
            for p in group['params']:
                if p.requires_grad and len(self.opt.state[p]) == 0:
                    self.opt.state[p]['step'] = torch.tensor(0.0, dtype=torch.float32)
                    self.opt.state[p]['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    self.opt.state[p]['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):          # eager warm-up on a side stream, as torch's graph capture requires: the pass
            for _ in range(self.warmup):       # only -- no ring advance, no optimiser step
                self._body(iter_num, whole=False)
        torch.cuda.current_stream(self.device).wait_stream(side)
        for p in self.model.parameters():
            p.grad = None                      # gradients are (re)allocated inside the capture: static addresses
        if self.world > 1 or self.force_collective:
            # ... except for parameters the backward never reaches: the all-reduce gives those a zero stand-in (every rank must
            # flatten the same layout), and ASSIGNING ``p.grad`` inside a capture would hand the optimiser a buffer of the graph's
            # private pool that no replay re-creates.  They get their static zero buffer here, before the capture;
            # inside it the all-reduce only copies into buffers that exist.
            for p in getattr(self, '_gradless', ()):
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
        self.model.invalidate_packed()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
            self.totals = self._body(iter_num, whole=True)

    def __call__(self, iter_num: int) -> Dict[str, Tensor]:
        positions = self.batcher.next_positions()
        lr = None
        if self.decayer is not None:
            lr = self.decayer.get_updated_learning_rate(iter_num)
            for group in self.opt.param_groups:
                group['lr'] = lr
        lr = self.opt.param_groups[0]['lr'] if lr is None else lr
        self.last_was_short = not self.batcher.is_full(positions)
        if self.last_was_short:
            # a short batch at the end of an epoch: the same sub-batched pass launched directly into the (zeroed) static
            # gradient buffers, then the ordinary optimiser step
            for p in self.model.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            totals = self._passes(self.batcher.get_next_batch(iter_num, positions=positions), None)
            allreduce_gradients(self.model.parameters(), self.world, self.group, self.force_collective)
            self.opt.step()
            return totals
        key = self._weights(iter_num)
        if self.graph is None or key != self.weights_key:
            self._capture(iter_num)            # records the work; nothing of it runs (and no ring record is consumed) until the replay
            self.weights_key = key
        slot = self.ring.filled % self.ring.slots
        if self.events[slot] is not None and not self.events[slot].query():
            import time
            t0 = time.perf_counter()
            self.events[slot].synchronize()    # the replay that read this slot has finished
            self.wait_seconds += time.perf_counter() - t0
        neg_step_size, bias2_sqrt = self.opt.graph_factors(self.opt.next_count(), lr)
        sparse = positions.get('sparse', (0, 0, 0))
        self.ring.fill(iter_num, positions['pixel'][:2], sparse[:2], neg_step_size, bias2_sqrt)
        self.graph.replay()
        event = torch.cuda.Event()
        event.record()
        self.events[slot] = event
        self.opt.count_step()
        if self.model.precision in ops.FP16_RANGE_PRECISIONS and ops.range_status(clear=True):
            raise ops.Fp16RangeError("GraphedIteration: an earlier replay met a value outside the fp16 range (|v| > 65504); its "
                                     "update is invalid -- use hip_precision 'fp32' for this model")
        return self.totals


# ---------------------------------------------------------------------------------------------------------------
# frame writer: what the reference's Tester does with a predicted frame (src/Tester01.py:69-92), without
# its skimage dependency -- PNG through zlib, .npy through numpy
def _png_bytes(array) -> bytes:
    import struct
    import zlib
    import numpy
    a = numpy.ascontiguousarray(array)
    if a.dtype != numpy.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise RuntimeError(f'PNG writer takes uint8 (h,w) or (h,w,3) arrays, got {a.dtype} {a.shape}')
    h, w = a.shape[:2]
    rows = numpy.concatenate([numpy.zeros((h, 1), dtype=numpy.uint8), a.reshape(h, -1)], axis=1)   # filter type 0 per row

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)

    header = struct.pack('>IIBBBBB', w, h, 8, 2 if a.ndim == 3 else 0, 0, 0, 0)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', header) + chunk(b'IDAT', zlib.compress(rows.tobytes(), 6)) + chunk(b'IEND', b'')


def save_image(path, image) -> None:
    """``image``: uint8 (h,w,3) array or tensor (e.g. ``to_display`` output reshaped); ``.png`` or ``.npy`` by suffix."""
    import os
    import numpy
    path = os.fspath(path)
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    array = image.cpu().numpy() if isinstance(image, torch.Tensor) else numpy.asarray(image)
    if path.endswith('.png'):
        with open(path, 'wb') as f:
            f.write(_png_bytes(array))
    elif path.endswith('.npy'):
        numpy.save(path, array)
    else:
        raise RuntimeError(f'Unknown image format: {path}')


def save_depth(path, depth, as_png: bool = False) -> None:
    """``depth``: float (h,w); ``.npy`` keeps the values (plus a PNG preview when ``as_png``), ``.png`` stores
    round(depth / depth.max() * 255) like the reference (src/Tester01.py:80-92)."""
    import os
    import numpy
    path = os.fspath(path)
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    array = depth.cpu().numpy() if isinstance(depth, torch.Tensor) else numpy.asarray(depth)
    preview = numpy.round(array / array.max() * 255).astype('uint8')
    if path.endswith('.png'):
        with open(path, 'wb') as f:
            f.write(_png_bytes(preview))
    elif path.endswith('.npy'):
        numpy.save(path, array)
        if as_png:
            with open(os.path.splitext(path)[0] + '.png', 'wb') as f:
                f.write(_png_bytes(preview))
    else:
        raise RuntimeError(f'Unknown depth format: {path}')
