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


def _headers():
    """What every object / assembly file depends on besides its own source (ONE list for build_extension and
    device_assembly: qd_transform.h holds almost all kernel code)."""
    return [os.path.join(_lib.CSRC, 'qd_common.h'), os.path.join(_lib.CSRC, 'qd_transform.h'), os.path.join(_lib.INCLUDE, 'qd_hip.h'),
            os.path.abspath(__file__)]


def _tmp(path):
    """Per-process temporary name next to `path`: concurrent first-import builds (N ranks under torchrun) must not write
    the same file; each finishes with an atomic os.replace of a complete file."""
    return '%s.tmp.%d' % (path, os.getpid())


SOURCES = ['qd_kernels.hip', 'qd_nearest.hip', 'qd_reductions.hip', 'qd_scale.hip', 'qd_codec.hip', 'qd_multi_dq.hip', 'qd_abs.hip', 'qd_multi_global.hip',
           'qd_select.hip', 'qd_selftest.hip']
OBJ_DIR = os.path.join(os.path.dirname(_lib.INCLUDE), 'build', 'obj')              # git-ignored; objects are rebuilt from source when stale


def _compile_flags():
    return [f for f in HIPCC_FLAGS if f != '-shared'] + ['-I', _lib.INCLUDE]


def build_extension(force=False, verbose=False, save_temps_dir=None):
    """One object per .hip file (compiled in parallel, reused while its sources are older), then one link.
    save_temps_dir: also keep the device assembly of every file there (tests/test_abi.py reads the kernels' register /
    scratch / LDS metadata from it)."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(_lib.CSRC, f) for f in SOURCES]
    hdrs = _headers()
    out = _lib.LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = [os.path.join(OBJ_DIR, os.path.splitext(f)[0] + '.o') for f in SOURCES]

    def stale(target, deps):
        return force or not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)

    def compile_one(pair):
        src, obj = pair
        if not stale(obj, [src] + hdrs):
            return
        tmp = _tmp(obj)
        cmd = [hipcc()] + _compile_flags() + ['-c', src, '-o', tmp]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, obj)

    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    if save_temps_dir is not None:
        device_assembly(save_temps_dir, verbose=verbose)
    if stale(out, objs):
        tmp = _tmp(out)
        cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', tmp]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    return out


def device_assembly(out_dir, verbose=False):
    """gfx950 assembly of every .hip file (hipcc -S --cuda-device-only) under out_dir; returns the .s paths.  Reused while
    newer than the sources."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(out_dir, exist_ok=True)
    hdrs = _headers()

    def one(f):
        src = os.path.join(_lib.CSRC, f)
        dst = os.path.join(out_dir, os.path.splitext(f)[0] + '.s')
        if os.path.exists(dst) and all(os.path.getmtime(dst) >= os.path.getmtime(d) for d in [src] + hdrs):
            return dst
        tmp = _tmp(dst)
        cmd = [hipcc()] + [x for x in _compile_flags() if x != '-fPIC'] + ['-S', '--cuda-device-only', src, '-o', tmp]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, dst)
        return dst

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        return list(ex.map(one, SOURCES))


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
           '-I', sysconfig.get_paths()['include'], src, '-o', _tmp(out),
           '-L', os.path.dirname(_lib.LIB_PATH), '-l:libqd_hip.so', '-L', os.path.join(tdir, 'lib'),
           '-ltorch_python', '-ltorch', '-ltorch_cpu', '-ltorch_hip', '-lc10', '-lc10_hip',
           '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + os.path.join(tdir, 'lib')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(_tmp(out), out)
    return out


HOST_FLAGS = ['-O3', '-std=c++17', '-fPIC', '-shared', '-fopenmp',
              '-ffp-contract=off', '-fno-fast-math',     # as the device build: every fp32 op rounded separately, no reassociation
              '-fno-trapping-math', '-fno-math-errno',   # value-neutral: lets the compiler if-convert and vectorise the compare / select chains
              '-fvisibility=hidden',                      # only the extern "C" entry points leave the library ...
              '-Wl,-Bsymbolic-functions']                 # ... and its own calls to them never resolve into libqd_hip.so (same names)


def build_host(force=False, verbose=False):
    """libqd_host.so: the per-call entry points of include/qd_hip.h for CPU tensors (csrc/host/qd_host.cpp), g++ + OpenMP."""
    src = os.path.join(_lib.CSRC, 'host', 'qd_host.cpp')
    out = _lib.HOST_LIB_PATH
    deps = [src, os.path.join(_lib.INCLUDE, 'qd_hip.h'), os.path.abspath(__file__)]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        raise RuntimeError('g++ not found')
    cmd = [cxx] + HOST_FLAGS + ['-I', _lib.INCLUDE, src, '-o', _tmp(out)]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(_tmp(out), out)
    return out


if __name__ == '__main__':
    print(build_extension(force=True, verbose=True))
    print(build_glue(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
