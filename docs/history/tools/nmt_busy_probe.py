#!/usr/bin/env python3
"""configs[4] step (2-layer LSTM seq2seq, batch 64): wall time of 10 eager steps, to be compared with the summed kernel time of the
same run under `rocprofv3 --kernel-trace --stats` -- how much of the step is the host issuing launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from harness import models  # noqa: E402
from harness.distill import DistillTrainer, seq2seq_kd_loss_fn, synthetic_token_batch  # noqa: E402

dev = torch.device('cuda:0')
batches = [synthetic_token_batch(64, dev, seed=i) for i in range(2)]
torch.manual_seed(0)
tr = DistillTrainer(models.Seq2SeqLSTM(), models.Seq2SeqLSTM(), dev, num_bits=4, bucket_size=256, lr=1.0, momentum=0.0, nesterov=False,
                    weight_decay=0.0, loss_fn=seq2seq_kd_loss_fn, clip_norm=5.0, grad_chunks=4, overlap_allreduce=True,
                    quantize_from_first_step=False)
for i in range(4):
    tr.step(*batches[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    tr.step(*batches[i % 2])
torch.cuda.synchronize()
print('WALL_MS_PER_STEP %.2f (14 steps run in all: 4 warm-up + 10 timed)' % ((time.perf_counter() - t0) * 100))
