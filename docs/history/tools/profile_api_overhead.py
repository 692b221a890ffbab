#!/usr/bin/env python3
"""Host-side cost of the drop-in per-tensor API (the reference's loop shape,
cnn_models/conv_forward_model.py:235-247): wall time per call in the launch-bound regime and of
the whole per-parameter loop for the config models, next to the multi-tensor launch.
Writes a table to stdout (-> profiles/rNN_api_overhead.txt)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from harness import models  # noqa: E402
from quantized_distillation_amd.multi_tensor import MultiTensorQuantizer  # noqa: E402

DEV = torch.device('cuda:0')


def wall(fn, reps):
    for _ in range(max(20, reps // 10)):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


def host_only(fn, reps):
    """Host time per call with the device queue kept shallow (sync every 64 calls, sync time excluded)."""
    fn()
    torch.cuda.synchronize()
    tot = 0.0
    done = 0
    while done < reps:
        t0 = time.perf_counter()
        for _ in range(64):
            fn()
        tot += time.perf_counter() - t0
        done += 64
        torch.cuda.synchronize()
    return tot / done * 1e6


print('torch', torch.__version__, torch.cuda.get_device_name(0))
x500, x800k = torch.randn(500, device=DEV), torch.randn(800000, device=DEV)
pts = torch.tensor([0.0, 0.42, 0.58, 1.0], device=DEV)
print('%-62s %8s %8s' % ('call', 'wall us', 'host us'))
for name, fn in (
        ('uniformQuantization(500 el, s=16, bucket=256)', lambda: quantization.uniformQuantization(x500, 16, bucket_size=256)),
        ('uniformQuantization(800000 el, s=16, bucket=256)', lambda: quantization.uniformQuantization(x800k, 16, bucket_size=256)),
        ('uniformQuantization(500 el, s=16, bucket=None)', lambda: quantization.uniformQuantization(x500, 16)),
        ('uniformQuantization(800000 el, s=16, bucket=None)', lambda: quantization.uniformQuantization(x800k, 16)),
        ('torch.empty_like(500 el)  [allocation alone]', lambda: torch.empty_like(x500)),
        ('x.clamp(-1, 1) on 500 el  [one torch op, for scale]', lambda: x500.clamp(-1, 1)),
):
    print('%-62s %8.2f %8.2f' % (name, wall(fn, 3000), host_only(fn, 3000)))
fnv = quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=x800k)
g = torch.randn(800000, device=DEV)
fnv.forward(None, pts)
for name, fn in (('nonUniformQuantization_variable.forward (800000 el, k=4)', lambda: fnv.forward(None, pts)),
                 ('nonUniformQuantization_variable.backward (800000 el, k=4)', lambda: fnv.backward(g))):
    print('%-62s %8.2f %8.2f' % (name, wall(fn, 3000), host_only(fn, 3000)))

from quantized_distillation_amd import _lib  # noqa: E402
for name, t in (('500 el', x500), ('800000 el', x800k)):
    a, b, c = _lib.glue().host_cost_probe(t, 16, 256, 6400)
    print('host cost inside the binding, %-10s: bare C-ABI launch %.2f us, two output allocations %.2f us, both %.2f us'
          % (name, a, b, c))
print()
print('%-62s %10s %10s' % ('per-parameter loop (reference shape) vs one multi-tensor launch', 'loop us', 'multi us'))
for name, net in (('CIFAR10 student (22 tensors, 1.0 M)', models.student()), ('WideResNet-16-22 (60 tensors, 82.7 M)', models.WideResNet(16, 22)),
                  ('resnet18 k=1.5 (62 tensors, 25.9 M)', models.ResNetK((2, 2, 2, 2), 1.5)), ('LSTM seq2seq (22 tensors, 28.8 M)', models.Seq2SeqLSTM())):
    ps = [p.detach().to(DEV).contiguous() for p in net.parameters()]
    for bucket in (256, None):
        def loop():
            for p in ps:
                quantization.uniformQuantization(p, 16, bucket_size=bucket)
        mt = MultiTensorQuantizer(ps, 16, bucket)
        print('%-62s %10.1f %10.1f' % ('%s bucket=%s' % (name, bucket), wall(loop, 200), wall(lambda: mt.quantize(check_pointers=False), 200)))
    del ps

if '--cprofile' in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5000):
        quantization.uniformQuantization(x500, 16, bucket_size=256)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
