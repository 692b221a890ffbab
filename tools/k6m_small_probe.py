"""K6m (qd_multi_point_grad_f32) on the 22-tensor CIFAR student and the 60-tensor WRN-16-22 lists: HIP-event time per call and the
scratch it needs; run under rocprofv3 --kernel-trace for the two kernels own durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from harness import kernel_bench
from quantized_distillation_amd.multi_tensor import MultiTensorDiffQuant
dev = torch.device('cuda:0')
for model in ('student', 'wrn'):
    shapes = kernel_bench.model_shapes(model)
    g = torch.Generator(device=dev).manual_seed(0)
    ts = [torch.randn(s, device=dev, generator=g).view(-1) for s in shapes]
    outs = [torch.empty_like(t) for t in ts]; grads = [torch.randn(t.shape, device=dev, generator=g) for t in ts]
    for k in (4, 16):
        mt = MultiTensorDiffQuant(ts, outs, grads, k, 256)
        pts = torch.sort(torch.rand(len(ts), k, device=dev, generator=g), dim=1)[0].contiguous()
        mt.forward(pts)
        out = torch.empty(len(ts), k, device=dev)
        for _ in range(200): mt.backward(out)
        torch.cuda.synchronize()
        reps = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): mt.backward(out)
            e1.record(); torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) * 10)
        print('%-8s k=%-3d K6m backward: median %.2f us (%.2f..%.2f), scratch %.2f MB' % (model, k, sorted(reps)[2], min(reps), max(reps), mt._scratch.numel() * 4 / 1e6))
