"""The C ABI used without PyTorch: a small C++ program (tests/native/c_abi_smoke.cpp) is compiled against include/*.h,
linked with libsimplenerf_hip.so and run on the GPU."""
import os
import shutil
import subprocess

import pytest

from simplenerf_amd import _lib, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, 'tests', 'native', 'c_abi_smoke.cpp')


def compile_smoke(out_dir):
    exe = os.path.join(out_dir, 'c_abi_smoke')
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    cmd = [build.HIPCC, '--offload-arch=gfx950', '-O1', '-std=c++17', f'-I{os.path.join(REPO, "include")}', SRC, '-o', exe,
           f'-L{lib_dir}', '-lsimplenerf_hip', f'-Wl,-rpath,{lib_dir}']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which(build.HIPCC) is None and not os.path.exists(build.HIPCC), reason='hipcc not available')
def test_c_abi_program_compiles_and_links(tmp_path):
    """CPU-side half: headers are valid C++ for an outside consumer and every symbol it uses links."""
    build.build_library()
    assert os.path.exists(compile_smoke(str(tmp_path)))


@pytest.mark.gpu
def test_c_abi_program_runs_without_torch(tmp_path):
    exe = compile_smoke(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'c_abi_smoke: OK' in r.stdout, r.stdout + r.stderr


def test_per_device_attribute_flags(tmp_path):
    """csrc/device_once.h on the CPU (g++, fake device ordinals): the dynamic-LDS cap of a kernel is raised once per
    DEVICE, not once per process -- a process that drives two devices must configure both."""
    exe = str(tmp_path / 'device_once_test')
    src = os.path.join(REPO, 'tests', 'native', 'device_once_test.cpp')
    r = subprocess.run(['g++', '-O1', '-std=c++17', '-pthread', src, '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and 'device_once_test: OK' in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_forward_from_two_threads_is_reentrant():
    """SURVEY 8b "Threading": DataParallel runs one Python thread per device into forward; the library holds no global
    mutable state, so two threads rendering concurrently (here on one device, two streams) get what one gets."""
    import threading
    import torch
    from simplenerf_amd import harness, synth
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.make_configs('config2')
    dev = torch.device('cuda', 0)
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    model = model.to(dev).eval()
    cam = synth.camera('fern', 0)
    with torch.no_grad():
        want = [model(harness.frame_batch(cam, True, dev, 1000 * (i + 1), 777))['rgb_fine'].clone() for i in range(2)]
    torch.cuda.synchronize()
    got, errors = [None, None], []

    def work(i):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream), torch.no_grad():
                for _ in range(5):
                    got[i] = model(harness.frame_batch(cam, True, dev, 1000 * (i + 1), 777))['rgb_fine']
            stream.synchronize()
        except Exception as e:                      # noqa: BLE001  (reported below)
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        assert torch.equal(got[i], want[i])
