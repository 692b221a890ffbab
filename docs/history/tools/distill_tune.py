#!/usr/bin/env python3
"""Harness-side knobs for the CIFAR distillation step (fp32 throughout): MIOpen find mode,
channels_last, teacher forward on its own stream.  Prints steps/s for each combination."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from harness import models  # noqa: E402
from harness.distill import DistillTrainer, synthetic_batch  # noqa: E402

dev = torch.device('cuda:0')


def run(bench, cl, steps=150, teacher_stream=True):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    st, te = models.student(), models.teacher()
    if cl:
        st, te = st.to(memory_format=torch.channels_last), te.to(memory_format=torch.channels_last)
    tr = DistillTrainer(st, te, dev, num_bits=4, bucket_size=256, mode='multi', teacher_stream=teacher_stream)
    batches = [synthetic_batch(50, dev, seed=i) for i in range(4)]
    if cl:
        batches = [(x.contiguous(memory_format=torch.channels_last), y) for x, y in batches]
    for i in range(30):
        tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


for rep in range(2):
    for ts in (False, True):
        print('teacher_stream=%s : %.1f steps/s' % (ts, run(False, False, steps=300, teacher_stream=ts)), flush=True)
if '--all' not in sys.argv:
    sys.exit(0)
for bench in (False, True):
    for cl in (False, True):
        try:
            print('cudnn.benchmark=%s channels_last=%s : %.1f steps/s' % (bench, cl, run(bench, cl)))
        except Exception as e:  # noqa: BLE001
            print('cudnn.benchmark=%s channels_last=%s : failed %r' % (bench, cl, e))
