#!/usr/bin/env python3
"""Build libqd_hip.so from the kernel sources of ANOTHER git revision (for A/B timings on one box: boxes of the pool
differ by up to 20 % on write-heavy kernels, so before/after numbers must come from the same lease).
    python tools/build_rev_lib.py <rev> build/libqd_hip_<name>.so
Tools that take QD_LIB=<path> (side_output_probe.py, ab_lib.py, bench_kernels.py) then load that library instead."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantized_distillation_amd import build as qb  # noqa: E402

rev, out = sys.argv[1], os.path.abspath(sys.argv[2])
with tempfile.TemporaryDirectory() as td:
    os.makedirs(os.path.join(td, 'q', 'csrc'))
    os.makedirs(os.path.join(td, 'include'))
    files = subprocess.check_output(['git', 'ls-tree', '--name-only', rev, 'quantized_distillation_amd/csrc/'], cwd=ROOT, text=True).split()
    srcs = []
    for f in files + ['include/qd_hip.h']:
        data = subprocess.check_output(['git', 'show', '%s:%s' % (rev, f)], cwd=ROOT)
        dst = os.path.join(td, 'include', 'qd_hip.h') if f.startswith('include/') else os.path.join(td, 'q', 'csrc', os.path.basename(f))
        open(dst, 'wb').write(data)
        if dst.endswith('.hip'):
            srcs.append(dst)
    cmd = [qb.hipcc()] + qb.HIPCC_FLAGS + ['-I', os.path.join(td, 'include')] + srcs + ['-o', out]
    print(' '.join(cmd))
    subprocess.check_call(cmd)
print(out)
