"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the plugin factory
follows the reference's naming rule, and the parameter container reproduces the reference's state_dict layout."""
import ctypes
import json

import numpy
import os
import re

import pytest
import torch

from simplenerf_amd import _lib, ops, synth
from simplenerf_amd.models.ModelFactory import get_model
from tests import util

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    headers = sorted(f for f in os.listdir(os.path.join(REPO, 'include')) if f.endswith('.h'))
    assert headers == ['simplenerf_hip.h', 'simplenerf_train.h']
    header = ''.join(open(os.path.join(REPO, 'include', f)).read() for f in headers)
    declared = set(re.findall(r'\b(snerf_[a-z_0-9]+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.snerf_abi_version() == _lib.ABI_VERSION


def test_descriptor_queries_need_no_gpu():
    lib = _lib.load()
    d = ops.mlp_desc(synth.mlp_config(64))
    assert lib.snerf_mlp_num_params(ctypes.byref(d)) == 24
    assert lib.snerf_mlp_packed_floats(ctypes.byref(d)) > 593408  # at least the MAC count of the main MLP
    d = ops.mlp_desc(synth.mlp_config(64, use_view_dirs=False, view_dependent_rgb=False))
    assert lib.snerf_mlp_num_params(ctypes.byref(d)) == 18
    # a shape outside the fused kernels' set goes to the layered path (round 4): its "packed" buffer is the parameters themselves
    layered = synth.mlp_config(64, width=192, views_width=96, views_depth=2)
    d = ops.mlp_desc(layered)
    from tests import util
    shapes = util.mlp_param_shapes(layered)
    assert lib.snerf_mlp_num_params(ctypes.byref(d)) == len(shapes) == 26
    count = sum(-(-int(numpy.prod(s)) // 4) * 4 for s in shapes.values())         # each tensor padded to 16 bytes
    assert lib.snerf_mlp_packed_floats(ctypes.byref(d)) == count
    assert lib.snerf_mlp_saved_floats(ctypes.byref(d), 10, 7) > 70 * (8 * 192 + 63)
    bad = ops.mlp_desc(synth.mlp_config(64, depth=5))                              # the reference itself cannot build depth 5
    assert lib.snerf_mlp_packed_floats(ctypes.byref(bad)) == 0
    assert b'points_net_depth' in lib.snerf_last_error()
    bad = ops.mlp_desc(synth.mlp_config(64, width=192, predict_visibility=True))
    assert lib.snerf_mlp_num_params(ctypes.byref(bad)) == 0 and b'predict_visibility' in lib.snerf_last_error()


def test_invalid_arguments_return_errors_without_touching_the_gpu():
    lib = _lib.load()
    st = lib.snerf_coarse_depths(None, None, 4, 8, 0, None, None, None)
    assert st == -1 and b'NULL' in lib.snerf_last_error()
    st = lib.snerf_resample_depths(ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 2, 8, None, ctypes.c_void_p(16), None)
    assert st == -1 and b'coarse' in lib.snerf_last_error()


@pytest.mark.parametrize('name', ['SimpleNeRFHip01', 'SimpleNeRF01'])
def test_factory_naming_rule(name):
    cfg = synth.make_configs('config3', model_name=name)
    model = get_model(cfg, None)
    assert type(model).__name__ == 'SimpleNeRFHip'
    assert isinstance(model, torch.nn.Module)
    with pytest.raises(RuntimeError, match='Unknown model'):
        get_model(synth.make_configs('config1', model_name='NoSuchModel07'), None)


@pytest.mark.parametrize('kind', ['config1', 'config2', 'config3'])
def test_state_dict_layout_matches_reference(kind):
    cfg = synth.make_configs(kind)
    model = get_model(cfg, None)
    expect = util.model_param_shapes(cfg)
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert got == expect
    # checkpoints written through DataParallel carry the 'module.' prefix (src/Trainer01.py:352-366)
    wrapped = torch.nn.DataParallel(model) if torch.cuda.is_available() else None
    if wrapped is not None:
        assert all(k.startswith('module.') for k in wrapped.state_dict())


def test_default_init_consumes_rng_like_reference_constructor_order():
    """Same seed -> same initial weights as a model whose Linear layers are created in the reference's order."""
    cfg = synth.make_configs('config2')
    torch.manual_seed(3)
    a = get_model(cfg, None).state_dict()
    torch.manual_seed(3)
    lin = torch.nn.Linear
    first = lin(63, 256)
    assert torch.equal(a['coarse_model.pts_linears.0.weight'], first.weight)


def test_loss_computer_host_logic():
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    configs = synth.make_configs('config3')
    configs['losses'] = synth.loss_configs()
    lc = LossComputer(configs)
    assert list(lc.losses) == [c['name'] for c in configs['losses']]
    assert lc.get_loss_weight(configs['losses'][6], 0) == 0 and lc.get_loss_weight(configs['losses'][6], 9999) == 0
    assert lc.get_loss_weight(configs['losses'][6], 10000) == 0.1 and lc.get_loss_weight(configs['losses'][0], 5) == 1
    with pytest.raises(RuntimeError, match='loss_weight is None'):
        lc.get_loss_weight({'name': 'MSE01'}, 0)
    with pytest.raises(RuntimeError, match='Unknown Loss Function'):
        LossComputer({**configs, 'losses': [{'name': 'VisibilityLoss01', 'weight': 1}]})
    # no CPU path: host tensors are refused, not silently evaluated by torch
    import torch
    inp = {'iter_num': 0, 'rays_o': torch.zeros(4, 3), 'indices_mask_nerf': torch.ones(4, dtype=torch.bool),
           'target_rgb': torch.zeros(4, 3)}
    out = {'rgb_coarse': torch.zeros(4, 3), 'rgb_fine': torch.zeros(4, 3)}
    with pytest.raises(RuntimeError, match='expected a tensor on the GPU'):
        LossComputer({**configs, 'losses': [{'name': 'MSE01', 'weight': 1}]}).compute_losses(inp, out)


def test_lr_decayers_follow_the_reference():
    from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer
    g = util.load('optim_adam.npz')
    nerf = get_lr_decayer({'optimizer': {'lr_decayer_name': 'NeRFLearningRateDecayer01', 'lr_initial': 5e-4, 'lr_decay': 250}})
    mip = get_lr_decayer({'num_iterations': 500000,
                          'optimizer': {'lr_decayer_name': 'MipNeRFLearningRateDecayer01', 'lr_initial': 5e-4,
                                        'lr_final': 5e-6, 'lr_decay_steps': 2500, 'lr_decay_mult': 0.01}})
    for it, a, b in zip(g['probe_iters'], g['nerf_lr'], g['mip_lr']):
        assert nerf.get_updated_learning_rate(int(it)) == float(a)          # bit-identical doubles
        assert mip.get_updated_learning_rate(int(it)) == float(b)
    dense = util.load('lr_schedules.npz')       # every 37th iteration up to 500 000, both schedules: the same doubles
    for it, a, b in zip(dense['iters'], dense['nerf_lr'], dense['mip_lr']):
        assert nerf.get_updated_learning_rate(int(it)) == float(a), it
        assert mip.get_updated_learning_rate(int(it)) == float(b), it
    with pytest.raises(RuntimeError, match='Unknown lr decayer'):
        get_lr_decayer({'optimizer': {'lr_decayer_name': 'CosineDecayer01'}})


def test_optimizer_state_and_checkpoints_interchange_with_the_reference_format(tmp_path):
    """state_dict layout = torch.optim.Adam's (what a reference checkpoint holds); model keys carry the DataParallel
    ``module.`` prefix on disk and load with or without it."""
    from simplenerf_amd import checkpoint, optim
    g = util.load('optim_adam.npz')
    configs = synth.make_configs('config1')
    model = get_model(configs, None)
    ours = optim.Adam(list(model.parameters()), lr=5e-4, betas=(0.9, 0.999))
    ref = torch.optim.Adam(list(model.parameters()), lr=5e-4, betas=(0.9, 0.999))
    assert sorted(ours.state_dict()['param_groups'][0]) == list(g['state_dict_group_keys'])
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    ref.step()                                          # CPU torch Adam creates the reference-format state
    ours.load_state_dict(ref.state_dict())
    sd = ours.state_dict()
    assert sorted(sd['state'][0]) == list(g['state_dict_state_keys']) and float(sd['state'][0]['step']) == 1.0
    ref2 = torch.optim.Adam(list(model.parameters()), lr=1e-3)
    ref2.load_state_dict(sd)                            # and back
    assert ref2.param_groups[0]['lr'] == 5e-4
    with pytest.raises(RuntimeError, match='GPU'):      # stepping needs the HIP library and device tensors
        ours.step()
    with pytest.raises(NotImplementedError):
        optim.Adam(list(model.parameters()), weight_decay=0.1)

    path = tmp_path / 'Model_Iter000010.tar'
    checkpoint.save_checkpoint(path, 10, model, ours)
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {'iteration_num', 'model_state_dict', 'optimizer_state_dict'}
    assert all(k.startswith('module.') for k in raw['model_state_dict'])
    other = get_model(configs, None)
    assert checkpoint.load_checkpoint(path, other, optim.Adam(list(other.parameters()))) == 10
    for (k, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    torch.save({'iteration_num': 3, 'model_state_dict': model.state_dict()}, path)   # un-prefixed names load too
    assert checkpoint.load_checkpoint(path, other) == 3


def test_batch_assembler_epoch_and_shard_arithmetic():
    """Host logic of BatchAssembler without a GPU: epoch slicing (the reference's indices[i:i+n] with the wrap rule,
    DataPreprocessor01.py:559-563) and the per-rank slice of a batch."""
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    asm = BatchAssembler.__new__(BatchAssembler)
    cursor = epoch = 0
    seen = []
    for _ in range(7):
        first, count, cursor, epoch = asm._next_slice(cursor, epoch, 300, 1000)
        seen.append((first, count, cursor, epoch))
    assert seen[:4] == [(0, 300, 300, 0), (300, 300, 600, 0), (600, 300, 900, 0), (900, 100, 0, 1)]   # short last slice
    assert seen[4] == (0, 300, 300, 1)
    assert asm._next_slice(700, 0, 300, 1000) == (700, 300, 0, 1)                                     # exact fit wraps too
    for world in (1, 2, 3, 8):
        covered = []
        for rank in range(world):
            asm.rank, asm.world_size = rank, world
            first, count, offset = asm._shard(40, 100)
            assert offset == first - 40
            covered += list(range(first, first + count))
        assert covered == list(range(40, 140))


def test_benchmark_scene_has_an_epoch_of_whole_batches():
    """``synth.training_scene(sparse_points=...)`` puts a sparse depth on EXACTLY that many pixels, and the count bench.py
    asks for is a whole number of global batches for every rank count the driver uses -- so that no timed iteration of
    config 5 is a short one (the default scene's ~4 570 points made every third 2048-row batch 476 rows)."""
    from simplenerf_amd import synth
    scene = synth.training_scene(num_views=2, height=24, width=32, sparse_points=128)
    assert int((scene['sparse_depths'] >= 0).sum()) == 128 == int((scene['sparse_errors'] >= 0).sum())
    assert int((synth.training_scene(num_views=2, height=24, width=32)['sparse_depths'] >= 0).sum()) != 128   # default: a fraction
    for world in (1, 2, 4, 8):
        rows = 2048
        points = (1500000 // (rows * world)) * rows * world        # bench.training_step
        assert points % (rows * world) == 0 and points <= 3 * 756 * 1008 and points >= 64 * rows * world


def test_frame_writer_round_trips(tmp_path):
    """harness.save_image / save_depth (the Tester's frame writer, reference src/Tester01.py:69-92): the PNG decodes
    back to the same pixels (parsed here with zlib only), .npy keeps the values, unknown suffixes raise."""
    import struct
    import zlib
    import numpy
    from simplenerf_amd import harness

    def decode(path):
        data = open(path, 'rb').read()
        assert data[:8] == b'\x89PNG\r\n\x1a\n'
        pos, chunks = 8, {}
        while pos < len(data):
            (n,), tag = struct.unpack('>I', data[pos:pos + 4]), data[pos + 4:pos + 8]
            body = data[pos + 8:pos + 8 + n]
            assert struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
            chunks[tag] = chunks.get(tag, b'') + body
            pos += 12 + n
        w, h, depth, colour = struct.unpack('>IIBB', chunks[b'IHDR'][:10])
        channels = 3 if colour == 2 else 1
        rows = numpy.frombuffer(zlib.decompress(chunks[b'IDAT']), dtype=numpy.uint8).reshape(h, 1 + w * channels)
        assert depth == 8 and (rows[:, 0] == 0).all()
        return rows[:, 1:].reshape((h, w, 3) if channels == 3 else (h, w))

    rng = numpy.random.RandomState(0)
    image = rng.randint(0, 256, size=(37, 53, 3)).astype(numpy.uint8)
    harness.save_image(tmp_path / 'frames' / '0001.png', torch.from_numpy(image))
    assert numpy.array_equal(decode(tmp_path / 'frames' / '0001.png'), image)
    harness.save_image(tmp_path / 'frames' / '0001.npy', image)
    assert numpy.array_equal(numpy.load(tmp_path / 'frames' / '0001.npy'), image)
    depth = rng.uniform(0.5, 9.0, size=(37, 53)).astype(numpy.float32)
    harness.save_depth(tmp_path / 'depth' / '0001.npy', depth, as_png=True)
    assert numpy.array_equal(numpy.load(tmp_path / 'depth' / '0001.npy'), depth)
    assert numpy.array_equal(decode(tmp_path / 'depth' / '0001.png'), numpy.round(depth / depth.max() * 255).astype('uint8'))
    with pytest.raises(RuntimeError, match='Unknown image format'):
        harness.save_image(tmp_path / 'x.jpg', image)


def test_global_rows_key_and_shard_arithmetic():
    """Global-row bookkeeping of the training draws (a rank's rows are a pixel shard followed by a sparse
    shard, not one contiguous range of the single-process batch; and the reference's trainer slices every TENSOR of the
    batch into sub-batches, so the rows travel as a per-row tensor)."""
    from simplenerf_amd.models.SimpleNeRFHip01 import global_rows
    assert global_rows({}, 7) == (0, None) and global_rows({'row_offset': 40}, 7) == (40, None)
    rows = torch.arange(7, dtype=torch.int64) + 100
    first, got = global_rows({'global_rows': rows, 'row_offset': 5}, 7)
    assert first == 0 and got is rows
    with pytest.raises(RuntimeError):
        global_rows({'global_rows': rows}, 6)            # a batch cut without cutting its rows
    # the union over ranks of the global rows of a sharded [pixel | sparse] batch is every row exactly once
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    asm = BatchAssembler.__new__(BatchAssembler)
    total_p, total_s, world = 1000, 37, 3
    seen = []
    for rank in range(world):
        asm.rank, asm.world_size = rank, world
        _, count, lo = asm._shard(0, total_p)
        _, count_s, lo_s = asm._shard(0, total_s)
        seen += list(range(lo, lo + count)) + list(range(total_p + lo_s, total_p + lo_s + count_s))
    assert sorted(seen) == list(range(total_p + total_s))


def test_bench_board_sampler_without_sensors():
    """bench.py's power / clock sampler degrades to nulls when the board's sensors are not readable (this container has no
    GPU at all): the bench line must still be produced."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sampler = bench.BoardSampler(0)
    with sampler:
        pass
    state = sampler.summary()
    assert state['power_w'] is None and state['sclk_mhz'] is None and state['power_cap_w'] is None


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(REPO, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_line_carries_the_attribution_fields():
    """The headline line is built from one timed region's raw measurements: `value` is ALL the work over the WHOLE elapsed
    time (a stalled step is not left out), and the per-step fields make such a stall visible and attributable (round 2's
    driver run lost 22 ms in a 88-ms region and the line could not say where)."""
    bench = _bench_module()
    steps = 20
    device_ms = [3.27] * steps
    device_ms[4] = 25.3                      # one stalled step
    enqueue_ms = [0.4] * steps
    enqueue_ms[4] = 22.4                     # ... during which the host did not enqueue
    elapsed = sum(device_ms) * 1e-3
    launch_ms = [1.07, 2.18] * steps
    samples = [131072, 262144] * steps
    line = bench.headline_line(1, steps, 5, 'fp32', elapsed, device_ms, enqueue_ms, launch_ms, samples, dropped=0,
                               settle_info=(184, 0.61))
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'timing'):
        assert key in line, key
    assert line['value'] == pytest.approx(1024 * steps / elapsed) and line['ms_per_step'] == pytest.approx(elapsed / steps * 1e3)
    timing = line['timing']
    assert timing['step_ms']['max'] == 25.3 and timing['step_ms']['argmax'] == 4 and timing['step_ms']['p50'] == 3.27
    assert timing['step_ms']['first'] == 3.27 and timing['enqueue_ms']['max'] == 22.4 and timing['enqueue_ms']['argmax'] == 4
    assert timing['idle_ms_per_step'] == pytest.approx((elapsed * 1e3 - sum(launch_ms)) / steps)
    assert timing['step_trace_ms'] == [round(v, 4) for v in device_ms] and timing['settle'] == {'runs': 184, 'seconds': 0.61}
    roof = line['roofline']
    assert roof['launches'] == 2 * steps and roof['launches_not_timed'] == 0 and roof['bound'] == 'mfma' and roof['peak'] == 157.3
    assert roof['achieved'] == pytest.approx(sum(samples) * 2 * 593408 / (sum(launch_ms) * 1e-3) / 1e12)
    assert roof['kernel_share_of_step'] == pytest.approx(sum(launch_ms) / (elapsed * 1e3))
    # step-level fraction: every algorithmic FLOP of the K steps over the whole wall time
    assert roof['step_frac'] == pytest.approx(1024 * steps * 384 * 2 * 593408 / elapsed / 1e12 / 157.3)
    # long regions keep the quantiles and drop the raw trace
    long = bench.step_summary(1.0, [3.0] * 300, [0.3] * 300, 900.0)
    assert 'step_trace_ms' not in long and long['idle_ms_per_step'] == pytest.approx(100.0 / 300)


def test_bench_timed_region_protocol_on_the_cpu():
    """settle() runs for at least the asked time; timed_steps() runs EXACTLY K steps between two fences and returns one
    device interval and one host interval per step."""
    import time
    bench = _bench_module()
    calls = []
    runs, spent = bench.settle(lambda: calls.append('s'), 0.05, gpu=False, chunk=4)
    assert runs == len(calls) and runs % 4 == 0 and spent >= 0.05
    log = []
    elapsed, device_ms, enqueue_ms = bench.timed_steps(lambda: (log.append('step'), time.sleep(0.002)), 5, lambda: log.append('fence'), gpu=False)
    assert log == ['fence'] + ['step'] * 5 + ['fence']      # (the dry marks run before the opening fence)
    assert len(device_ms) == len(enqueue_ms) == 5 and all(v >= 2.0 for v in device_ms) and elapsed >= 0.01
    assert sum(device_ms) <= elapsed * 1e3 + 1e-6


def test_torch_library_binding_loads_and_rejects_cpu_tensors():
    """The TORCH_LIBRARY extension (csrc_torch/snerf_torch.cpp, built by torch.utils.cpp_extension) registers
    torch.ops.snerf.render against the same C ABI; without a GPU it must load, report the ABI version and refuse CPU tensors
    with a c10::Error (RuntimeError) instead of computing anything."""
    from simplenerf_amd import _torch_ext
    snerf = _torch_ext.load()
    assert int(snerf.abi_version()) == _lib.ABI_VERSION
    schema = str(torch.ops.snerf.render.default._schema)
    assert 'Tensor?[] packed' in schema and 'Tensor[] params' in schema and schema.endswith('-> Tensor[]')
    d = ops.mlp_desc(synth.mlp_config(64))
    descs = [d.points_net_depth, d.points_net_width, d.views_net_depth, d.views_net_width, d.points_pe_degree, d.views_pe_degree,
             d.sigma_pe_degree, d.use_view_dirs, d.view_dependent_rgb, d.predict_visibility] + [0] * 50
    cfg = [0, 0, 0, 64, 0, 0, 1, 0]
    rays = [torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 3), None, None, torch.ones(4, 1), torch.ones(4, 1), None]
    packed = [torch.zeros(8)] + [None] * 5
    with pytest.raises(RuntimeError, match='GPU'):
        snerf.render(cfg, descs, packed, rays, [None] * 9, [], [0] * 6, False, False)
    with pytest.raises(RuntimeError, match='cfg holds'):
        snerf.render(cfg[:3], descs, packed, rays, [None] * 9, [], [0] * 6, False, False)


def test_graphed_step_count_follows_the_parameters_that_hold_gradients():
    """``Adam.next_count`` used to read the count of the FIRST parameter whether or not it takes part in the steps;
    with a frozen / unused first parameter every graph replay was then fed step 1's bias-correction factors."""
    import torch
    from simplenerf_amd import optim
    frozen, used = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
    opt = optim.Adam([frozen, used], lr=1e-3)
    used.grad = torch.ones(3)
    assert opt.next_count() == 1
    for expected in (1, 2, 3):
        assert opt.next_count() == expected
        assert opt.count_step() == expected          # what a graph replay does after the captured step_at
    assert opt.next_count() == 4 and opt._count(frozen) == 0
    # one record serves every parameter of a replay: differing counts, betas or learning rates are refused
    other = torch.nn.Parameter(torch.zeros(2))
    other.grad = torch.ones(2)
    opt.add_param_group({'params': [other]})
    with pytest.raises(RuntimeError, match='same step count'):
        opt.next_count()
    opt.param_groups[1]['lr'] = 5e-4
    with pytest.raises(RuntimeError, match='same betas and learning rate'):
        opt.graph_factors(4, 1e-3)


def test_torch_sum_and_cumsum_orders_that_the_resampling_kernel_reproduces():
    """K5 (csrc/resample_device.h) adds sample_pdf's normaliser in torch.sum's order and runs its CDF as torch.cumsum does; both
    orders are restated in numpy by tools/check_torch_sum_order.py -- if a torch build summed differently, this fails here (and
    the bit-exact GPU resampling test would fail there)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'check_torch_sum_order.py')
    spec = importlib.util.spec_from_file_location('check_torch_sum_order', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for n in (5, 62, 126, 190, 700):
        assert mod.mismatches(n, rows=40) == (0, 0), n


def test_bench_stdout_line_is_short_whatever_the_record_holds(tmp_path, capfd):
    """BENCH_r04.json had ``parsed: null`` -- the stdout line had grown to 21.7 KB and progress lines followed it.
    ``emit`` now writes the full record to a side file and prints ONE line < 2 KB holding only the contract's keys; a record
    the size of round 4's (every secondary measurement attached) comes out the same short line."""
    bench = _bench_module()
    steps = 50
    full = bench.headline_line(1, steps, 5, 'fp32', 0.164, [3.28] * steps, [0.4] * steps, [1.07, 2.18] * steps, [131072, 262144] * steps,
                               dropped=0, settle_info=(184, 0.61))
    full['cpu_baseline'] = {'value': 1353.123456789, 'unit': 'rays/s', 'cores': 16, 'kind': 'port', 'sample': 'x' * 120,
                            'leg': {'runs': 9}, 'host': {'os_cpu_count': 256}}
    full['also'] = {k: {'rays_s': 2671234.5678, 'frac': 0.5431234} for k in ('f16x3', 'f16', 'bf16')}
    full['collective'] = {'backend': 'nccl', 'ranks': 8, 'bytes': 16384, 'pattern': 'p' * 80, 'gather_ms': {'p50': 0.0251234, 'max': 0.1},
                          'per_rank': [{'rank': i, 'step_ms_p50': 3.3, 'elapsed_s': 0.1} for i in range(8)]}
    for key in ('also_measured', 'also_measured_train', 'also_measured_frame', 'sustained'):      # round 4's ballast
        full[key] = {'blob': [{'trace': list(range(64)), 'text': 'y' * 200} for _ in range(12)]}
    assert len(json.dumps(full)) > 20000
    bench.EXTRA_FILE = str(tmp_path / 'side' / 'extra.json')
    bench.emit(full)
    out, err = capfd.readouterr()
    assert err == '' and out.endswith('\n') and out.count('\n') == 1
    assert len(out) < 2048, len(out)
    line = json.loads(out)
    assert set(line) == {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                         'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'collective', 'also', 'extra'}
    assert set(line['roofline']) == {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_algorithmic', 'traffic_source',
                                     'kernel', 'launches', 'avg_launch_ms'}
    assert set(line['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'sample'} and line['cpu_baseline']['cores'] == 16
    assert line['collective'] == {'backend': 'nccl', 'ranks': 8, 'bytes': 16384, 'gather_ms_p50': 0.0251234}
    assert line['roofline']['frac'] == pytest.approx(full['roofline']['frac'], rel=1e-5)
    assert line['value'] == pytest.approx(full['value'], rel=1e-5)
    # the side file holds the full record, untouched
    with open(line['extra']) as f:
        assert json.load(f) == json.loads(json.dumps(full))
    # the measured-traffic source names the commit the counters were collected at
    if line['roofline']['traffic'] is not None:
        assert 'profiles/pmc_traffic.json @ ' in line['roofline']['traffic_source']


def test_bench_cpu_leg_uses_the_cores_the_job_owns():
    """The CPU leg's thread count: affinity mask, capped by the cgroup quota and by 16 (round 4's 256-thread leg on a 16-core
    share never finished a pass and cost the default run 45 s)."""
    bench = _bench_module()
    assert bench.host_cores(affinity=256, quota=16.0) == 16
    assert bench.host_cores(affinity=256, quota=None) == 16          # no quota visible: the pool's per-GPU share
    assert bench.host_cores(affinity=8, quota=None) == 8
    assert bench.host_cores(affinity=8, quota=2.5) == 2 + (1 if 2.5 + 0.5 >= 3 else 0)
    assert bench.host_cores(affinity=4, quota=0.3) == 1
    assert 1 <= bench.host_cores() <= 16
    quota = bench.cgroup_cpu_quota()
    assert quota is None or quota > 0


def test_bench_strong_scaling_rows(monkeypatch):
    """``--train --global-rows 4096`` is BASELINE config 5 as stated: ONE 4096-row batch over the N ranks (512 rows per rank at
    N = 8), not 4096 rows per GPU."""
    import argparse
    bench = _bench_module()
    ns = lambda **kw: argparse.Namespace(global_rows=kw.get('g'), rows_per_gpu=kw.get('r'))
    assert bench.train_rows_per_gpu(ns(g=4096), 8) == (512, True)
    assert bench.train_rows_per_gpu(ns(g=4096), 1) == (4096, True)
    assert bench.train_rows_per_gpu(ns(), 8) == (4096, False)
    assert bench.train_rows_per_gpu(ns(r=512), 1) == (512, False)
    with pytest.raises(SystemExit):
        bench.train_rows_per_gpu(ns(g=4096 + 2), 8)
    assert '256 pixel + 256' in bench.train_workload(512, 8, True) and '4096-row batch over 8' in bench.train_workload(512, 8, True)


def test_every_diagnostic_switch_is_guarded():
    """the ablation / probe switches inside the product kernels (wrong results by design) cannot reach the shipped
    library.  (i) every such macro the sources test is listed in csrc/probe_guard.h, which is force-included into every
    translation unit and #errors unless SNERF_PROBE_BUILD is defined; (ii) build.py refuses them for the shipped library's name
    wherever they come from (extra flags, HIPCC, HIPCC_COMPILE_FLAGS_APPEND, CXXFLAGS ...); (iii) a real compile of a library
    source with -DSNERF_PROBE_HALF_X fails, and passes once SNERF_PROBE_BUILD is there too."""
    import subprocess
    from simplenerf_amd import build
    csrc = build.CSRC
    guard = open(os.path.join(csrc, 'probe_guard.h')).read()
    used = set()
    for name in os.listdir(csrc):
        if name.endswith(('.hip', '.h')) and name != 'probe_guard.h':
            used |= set(re.findall(r'\b(SNERF_(?:ABL|PROBE)_[A-Z0-9_]+|SNERF_CLOCK_STAMP)\b', open(os.path.join(csrc, name)).read()))
    used.discard('SNERF_PROBE_BUILD')
    assert used and all(f'defined({m})' in guard for m in used), sorted(m for m in used if f'defined({m})' not in guard)
    assert '-include' in build.FLAGS and build.FLAGS[build.FLAGS.index('-include') + 1].endswith('probe_guard.h')
    # (ii)
    with pytest.raises(RuntimeError, match='refusing to build the shipped library'):
        build.check_shipped_build(['-DSNERF_PROBE_HALF_X'], build.LIB, environ={})
    for var in ('HIPCC', 'HIPCC_COMPILE_FLAGS_APPEND', 'CXXFLAGS'):
        with pytest.raises(RuntimeError, match=var):
            build.check_shipped_build([], build.LIB, environ={var: '/opt/rocm/bin/hipcc -DSNERF_ABL_NODMA'})
    with pytest.raises(RuntimeError, match='SNERF_PROBE_BUILD'):
        build.check_shipped_build(['-DSNERF_PROBE_HALF_X'], '/tmp/variant.so', environ={})
    build.check_shipped_build(['-DSNERF_PROBE_BUILD', '-DSNERF_PROBE_HALF_X'], '/tmp/variant.so', environ={})
    build.check_shipped_build([], build.LIB, environ={'CXXFLAGS': '-O2'})
    with pytest.raises(RuntimeError, match='refusing'):
        build.build_library(extra_flags=['-DSNERF_PROBE_HALF_X'])
    # (iii) the compiler itself (host pass only: seconds)
    source = os.path.join(csrc, 'display.hip')
    cmd = [build.HIPCC, *build.FLAGS, '--cuda-host-only', '-fsyntax-only', source]
    bad = subprocess.run(cmd + ['-DSNERF_PROBE_HALF_X'], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'diagnostic switch' in bad.stderr, bad.stderr[-2000:]
    for extra in ([], ['-DSNERF_PROBE_HALF_X', '-DSNERF_PROBE_BUILD']):
        ok = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=300)
        assert ok.returncode == 0, ok.stderr[-2000:]


def test_no_packed_fp32_instruction_reads_the_high_register_into_the_low_result(tmp_path):
    """Round 5: ``v_pk_{mul,add,fma}_f32`` with ``op_sel:[0,1...]`` -- the low result reading the HIGH register of the second
    source -- returns wrong low halves beside another kernel's MFMAs on MI355X (tools/probes/pk_opsel_hazard.hip,
    profiles/r05_pk_opsel_hazard.txt; found as gradients that differed between repetitions once levels ran side by side).
    (i) the shipped library holds no such instruction (build.py checks the same after every link and refuses the library);
    (ii) the scan itself finds the form in a kernel written to contain it."""
    import subprocess
    from simplenerf_amd import build
    assert os.path.exists(build.OBJDUMP), build.OBJDUMP
    found = build.hazardous_packed_forms(build.LIB)
    assert not found, found[:10]
    assert build.FILE_FLAGS['composite'] == ['-fno-slp-vectorize'] and build.FILE_FLAGS['losses'] == ['-fno-slp-vectorize']
    source = tmp_path / 'form.hip'
    source.write_text('#include <hip/hip_runtime.h>\n'
                      'typedef float f32x2 __attribute__((ext_vector_type(2)));\n'
                      '__global__ void has_the_form(f32x2* p) {\n'
                      '    f32x2 a = p[threadIdx.x], b = p[threadIdx.x + 64], d;\n'
                      '    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));\n'
                      '    p[threadIdx.x] = d;\n'
                      '}\n'
                      '__global__ void has_a_safe_form(f32x2* p) {\n'
                      '    f32x2 a = p[threadIdx.x], b = p[threadIdx.x + 64], d;\n'
                      '    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));\n'
                      '    p[threadIdx.x] = d;\n'
                      '}\n')
    lib = tmp_path / 'form.so'
    done = subprocess.run([build.HIPCC, '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', str(source), '-o', str(lib)],
                          capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    seen = build.hazardous_packed_forms(str(lib))
    assert len(seen) == 1 and 'has_the_form' in seen[0][0] and 'op_sel:[0,1]' in seen[0][1], seen
