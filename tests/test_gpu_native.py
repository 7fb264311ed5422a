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
