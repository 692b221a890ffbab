#!/usr/bin/env python3
"""A few dispatches of the nearest-point calls round 3 changed, for SQ-counter passes (rocprofv3 --pmc ...), N = 64 Mi:
the pre-processed forward (K5, uint8 indices) at k = 256 / 16 / 4 on the vector kernel (bucket 256), at k = 4 on the chunk
kernel (bucket 100) and on chunk_any (bucket 33).  QD_LIB=<path> loads another build of the library (before / after)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from quantized_distillation_amd import _lib  # noqa: E402

if os.environ.get('QD_LIB'):
    _lib.LIB_PATH = os.environ['QD_LIB']
lib = _lib.load()
N = 1 << 26
dev = torch.device('cuda:0')
us = [torch.rand(N, device=dev) for _ in range(2)]
q = torch.empty(N, device=dev)
idx = torch.empty(N, dtype=torch.uint8, device=dev)
ws = _lib.workspace(dev)
st = torch.cuda.current_stream().cuda_stream
for b, k in ((256, 256), (256, 16), (256, 4), (100, 4), (33, 4), (100, 256)):
    pts = torch.sort(torch.rand(k, device=dev))[0].contiguous()
    nb = lib.qd_num_buckets(N, b)
    ab = torch.ones(2, nb, device=dev)
    ab[1].zero_()
    for i in range(3):
        _lib.check(lib.qd_nearest_point_f32(us[i % 2].data_ptr(), 1, pts.data_ptr(), k, 1, q.data_ptr(), idx.data_ptr(), 1, N, b,
                                            ab[0].data_ptr(), ab[1].data_ptr(), None, 0, 0.0, ws.data_ptr(), ws.numel(), st))
    torch.cuda.synchronize()
print('ok')
