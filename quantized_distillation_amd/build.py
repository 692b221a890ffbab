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
            os.path.join(_lib.CSRC, 'qd_multi_global.hip')]
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


if __name__ == '__main__':
    print(build_extension(force=True, verbose=True))
