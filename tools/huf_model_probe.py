import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, quantization
import quantization.help_functions as qhf
from harness import models
dev = torch.device('cuda:0')
torch.manual_seed(0)
params = [p.detach().to(dev) for p in models.WideResNet(16, 22).parameters()]
n = sum(p.numel() for p in params)
fn = lambda t: quantization.uniformQuantization(t, 16, bucket_size=256)
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = qhf.get_huffman_encoding_mean_bit_length(iter(params), fn, 'uniform', s=16)
    torch.cuda.synchronize(); return time.perf_counter() - t0, v
for tag in ('one-pass (round 6)', 'two-kernel form'):
    if tag != 'one-pass (round 6)':
        qhf._fused_rescale_counts = lambda *a, **k: None
    run()
    ts = [run() for _ in range(7)]
    print('%-22s get_huffman_encoding_mean_bit_length on WRN-16-22 (60 tensors, %.1f M): median %.2f ms, min %.2f ms, %.6f bits/weight' % (
        tag, n / 1e6, sorted(t for t, _ in ts)[3] * 1e3, min(t for t, _ in ts) * 1e3, ts[0][1]))
