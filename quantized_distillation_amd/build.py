"""Builds libqd_hip.so for gfx950 with hipcc (cross-compiles without a GPU, a few seconds)."""
import os
import shutil
import subprocess

from . import _lib

HIPCC_FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
    '-ffp-contract=off',                            # every fp32 op rounded separately, as the reference does
    '-fhip-fp32-correctly-rounded-divide-sqrt',     # IEEE division: the level index must be bit exact
    '-fno-fast-math',
]


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (looked on PATH and in /opt/rocm/bin)')
    return exe


def build_extension(force=False, verbose=False):
    srcs = [os.path.join(_lib.CSRC, 'qd_kernels.hip'), os.path.join(_lib.CSRC, 'qd_codec.hip'),
            os.path.join(_lib.CSRC, 'qd_multi_dq.hip'), os.path.join(_lib.CSRC, 'qd_abs.hip'),
            os.path.join(_lib.CSRC, 'qd_multi_global.hip'), os.path.join(_lib.CSRC, 'qd_select.hip')]
    deps = srcs + [os.path.join(_lib.CSRC, 'qd_common.h'), os.path.join(_lib.INCLUDE, 'qd_hip.h')]
    out = _lib.LIB_PATH
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = [hipcc()] + HIPCC_FLAGS + ['-I', _lib.INCLUDE] + srcs + ['-o', out + '.tmp']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + '.tmp', out)
    return out


def build_glue(force=False, verbose=False):
    """_qd_glue.so: the CPython/ATen binding of the per-call entry points (csrc/qd_torch_glue.cpp), host code only
    (g++), linked against libqd_hip.so ($ORIGIN rpath) and libtorch."""
    import sysconfig

    import torch
    src = os.path.join(_lib.CSRC, 'qd_torch_glue.cpp')
    out = _lib.GLUE_PATH
    deps = [src, os.path.join(_lib.INCLUDE, 'qd_hip.h'), _lib.LIB_PATH]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    tdir = os.path.dirname(torch.__file__)
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        raise RuntimeError('g++ not found')
    cmd = [cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-w', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           '-I', _lib.INCLUDE, '-I', os.path.join(tdir, 'include'),
           '-I', os.path.join(tdir, 'include', 'torch', 'csrc', 'api', 'include'), '-I', '/opt/rocm/include',
           '-I', sysconfig.get_paths()['include'], src, '-o', out + '.tmp',
           '-L', os.path.dirname(_lib.LIB_PATH), '-l:libqd_hip.so', '-L', os.path.join(tdir, 'lib'),
           '-ltorch_python', '-ltorch', '-ltorch_cpu', '-ltorch_hip', '-lc10', '-lc10_hip',
           '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + os.path.join(tdir, 'lib')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + '.tmp', out)
    return out


if __name__ == '__main__':
    print(build_extension(force=True, verbose=True))
    print(build_glue(force=True, verbose=True))
