#!/usr/bin/env python3
"""Per-kernel times of the radix select (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import quantization.help_functions as qhf  # noqa: E402

n = int(os.environ.get('N', 17842176))
m = int(os.environ.get('M', 8))
kind = os.environ.get('KIND', 'gauss')
u = torch.rand(n, device='cuda') if kind == 'unit' else (torch.randn(n, device='cuda') * 0.12 + 0.5).clamp_(0, 1)
ranks = np.linspace(0, n - 1, m).astype(np.int64)
for _ in range(20):
    qhf.order_statistics(u, ranks)
torch.cuda.synchronize()
