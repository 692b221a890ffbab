"""The C ABI used from plain C++/HIP (no Python, no torch): builds tests/native/native_check.cpp
against libqd_hip.so (+ the C oracle as checker) and runs it on the GPU."""
import os
import shutil
import subprocess

import pytest

from oracle import oracle_c
from quantized_distillation_amd import _lib
from quantized_distillation_amd import build as qb

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_cpp_client_of_the_c_abi():
    qb.build_extension()
    oracle_so = oracle_c.build()
    exe = os.path.join(ROOT, 'build', 'native_check')
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, 'tests', 'native', 'native_check.cpp')
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    libdir, odir = os.path.dirname(_lib.LIB_PATH), os.path.dirname(oracle_so)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'include'), src,
                           '-L', libdir, '-lqd_hip', '-L', odir, '-lqd_oracle',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,' + odir, '-o', exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'NATIVE CHECK PASSED' in out.stdout
