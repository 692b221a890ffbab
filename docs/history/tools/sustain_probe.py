#!/usr/bin/env python3
"""Why is a long back-to-back run of the headline kernel slower from Python than from kbench?
Runs 400 launches back to back in several configurations, timing chunks of 20 with events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import _lib  # noqa: E402

N = 64 * 1024 * 1024
dev = torch.device('cuda:0')
lib = _lib.load()
gen = torch.Generator().manual_seed(0)
xs_host = [torch.randn(N, generator=gen).to(dev) for _ in range(4)]
xs_dev = [torch.randn(N, device=dev) for _ in range(4)]
outs = [torch.empty(N, device=dev) for _ in range(4)]
ab = torch.empty(2, N // 256, device=dev)
ws = _lib.workspace(dev)
live = [None] * 4


def api(xs):
    def f(i):
        q, _ = quantization.uniformQuantization(xs[i % 4], 16, bucket_size=256)
        live[i % 4] = q
    return f


def cabi(xs, with_ab, stream=None):
    def f(i):
        st = stream if stream is not None else _lib.stream_ptr()
        lib.qd_uniform_f32(xs[i % 4].data_ptr(), outs[i % 4].data_ptr(), N, 256, 16,
                           ab[0].data_ptr() if with_ab else None, ab[1].data_ptr() if with_ab else None,
                           None, None, 0, 0.0, 0, 0, ws.data_ptr(), ws.numel(), st)
    return f


def run(name, fn, launches=400, chunk=20):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(launches // chunk + 1)]
    evs[0].record()
    for c in range(launches // chunk):
        for i in range(chunk):
            fn(c * chunk + i)
        evs[c + 1].record()
    torch.cuda.synchronize()
    t = [evs[c].elapsed_time(evs[c + 1]) * 1e3 / chunk for c in range(launches // chunk)]
    print('%-44s' % name, ' '.join('%5.0f' % v for v in t))


run('api, host-generated randn', api(xs_host))
run('api, device-generated randn', api(xs_dev))
run('c-abi fixed outs, alpha/beta, host data', cabi(xs_host, True))
run('c-abi fixed outs, no alpha/beta, host data', cabi(xs_host, False))
run('c-abi fixed outs, no alpha/beta, dev data', cabi(xs_dev, False))
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    run('c-abi, side stream, alpha/beta, host data', cabi(xs_host, True, s2.cuda_stream))
run('api again, host-generated randn', api(xs_host))
print(torch.cuda.memory_summary(abbreviated=True)[:600])
