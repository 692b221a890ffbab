#!/usr/bin/env python3
"""A/B of compile-time variants of the codec kernels on one box (round 4's record: profiles/r04_ab_codec.txt; the variants
themselves were removed from csrc/qd_codec.hip once the narrow form had won -- the macro is what a future A/B would
re-introduce): builds csrc/qd_codec.hip on its own with -DQD_UNPACK_VARIANT=<v> for every variant given, times qd_unpack_uniform_f32 (4-bit, bucket 256, 64 Mi elements, HIP events,
3 rotating outputs) and checks that every variant decodes to the same bits.   python tools/ab_codec.py 0 1 2 3 4"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from harness.kernel_bench import time_row  # noqa: E402
from quantized_distillation_amd import _lib, build, codec  # noqa: E402

variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]
out_dir = os.path.join(ROOT, 'build', 'ab')
os.makedirs(out_dir, exist_ok=True)
dev = torch.device('cuda:0')
N = 1 << 26
xs = [torch.randn(N, device=dev) for _ in range(3)]
ref = None
for bits, s in ((4, 16), (8, 256), (2, 4)):
    pks = [codec.pack_uniform(x, s, 256, bits=bits) for x in xs]
    for v in variants:
        so = os.path.join(out_dir, 'libcodec_v%d.so' % v)
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_lib.CSRC, 'qd_codec.hip')):
            subprocess.check_call([build.hipcc()] + build.HIPCC_FLAGS + ['-I', _lib.INCLUDE, '-DQD_UNPACK_VARIANT=%d' % v,
                                                                        os.path.join(_lib.CSRC, 'qd_codec.hip'), '-o', so])
        lib = ctypes.CDLL(so)
        fn = lib.qd_unpack_uniform_f32
        fn.restype = ctypes.c_int
        fn.argtypes = _lib.SIGNATURES['qd_unpack_uniform_f32'][1]
        ys = [torch.empty(N, device=dev) for _ in range(3)]

        def call(i):
            pk = pks[i % 3]
            rc = fn(pk.packed.data_ptr(), N, 256, s, bits, pk.alpha.data_ptr(), pk.beta.data_ptr(), ys[i % 3].data_ptr(), _lib.stream_ptr())
            assert rc == 0, rc
        us, lo, hi = time_row(call)
        call(0)
        torch.cuda.synchronize()
        if v == variants[0]:
            ref = ys[0].clone()
        same = torch.equal(ys[0], ref)
        byt = (4 + bits / 8) * N
        print('bits %d variant %d: %7.2f us (%.2f..%.2f)  %6.0f GB/s  %.1f%% of 8 TB/s  same bits as variant %d: %s'
              % (bits, v, us, lo, hi, byt / us / 1e3, byt / us / 1e3 / 80, variants[0], same), flush=True)
