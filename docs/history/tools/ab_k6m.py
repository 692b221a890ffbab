"""A/B of the multi-tensor point-gradient sweep (K6m, qd_multi_point_grad_f32) on the WRN-16-22 shape list (the per-step call
of BASELINE configs[2]): the library as built against build/libqd_hip_prev.so (the previous build, kept by hand before a
kernel change).  Per library: its own plan, HIP-event time (median of 5 x 40 launches, 3 rotating gradient sets > 256 MiB),
error against a float64 reference relative to sum |g alpha|, bit-identity over 20 launches.

    python tools/ab_k6m.py [k ...]          -> stdout (tools/gpu_session.sh abk6m -> profiles/r05_ab_k6m.txt)
"""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from harness import kernel_bench  # noqa: E402
from quantized_distillation_amd import _lib  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(path)
    for name in ('qd_multi_dq_plan', 'qd_multi_point_grad_f32'):
        res, args = _lib.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith('--models=')]
    models = ([a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--models=')] or ['wrn,student,one64Mi'])[0].split(',')
    ks = [int(a) for a in argv] or [4, 16]
    dev = torch.device('cuda', 0)
    libs = [('current', _lib.LIB_PATH)]
    prev = os.path.join(ROOT, 'build', 'libqd_hip_prev.so')
    if os.path.exists(prev):
        libs.insert(0, ('previous', prev))
    for model in models:
        # diagnostic shape lists besides the two config models: one tensor; the WRN list without its 43 tensors below one tile
        # (wrn_big); eight equal tensors of the same total (eq8); the WRN list with every tensor a view of ONE flat buffer per
        # kind, 256-byte aligned slots, as the data-parallel harness lays its gradients out (wrn_flat)
        base = model.replace('_flat', '').replace('_big', '')
        if model == 'one64Mi':
            shapes = [(1 << 26,)]
        elif model == 'eq8':
            shapes = [(82746890 // 8,)] * 8
        else:
            shapes = kernel_bench.model_shapes(base)
        ns = [int(np.prod(s)) for s in shapes]
        if model.endswith('_big'):
            ns = [n for n in ns if n >= 1024]
        flat = model.endswith('_flat')
        tot = sum(ns)
        g = torch.Generator().manual_seed(0)
        nset = 64 if model == 'student' else 3
        tot = sum(ns)
        for k in ks:
            sets = []
            for _ in range(nset):
                if flat:
                    offs = np.cumsum([0] + [-(-n // 64) * 64 for n in ns])
                    fg = torch.randn(int(offs[-1]), generator=g).to(dev)
                    fi = torch.randint(0, k, (int(offs[-1]),), generator=g, dtype=torch.uint8).to(dev)
                    grads = [fg[o:o + n] for o, n in zip(offs, ns)]
                    idx = [fi[o:o + n] for o, n in zip(offs, ns)]
                else:
                    grads = [torch.randn(n, generator=g).to(dev) for n in ns]
                    idx = [torch.randint(0, k, (n,), generator=g, dtype=torch.uint8).to(dev) for n in ns]
                alpha = [(torch.rand(-(-n // 256) if n > 256 else 1, generator=g) + 0.5).to(dev) for n in ns]
                sets.append((grads, idx, alpha))
            # float64 reference of set 0
            ref = np.zeros((len(ns), k))
            scale = np.zeros(len(ns))
            for t, (gr, ix, al) in enumerate(zip(*sets[0])):
                n = gr.numel()
                a = al.double().repeat_interleave(256)[:n] if n > 256 else al.double().expand(n)
                prod = gr.double() * a
                ref[t] = torch.zeros(k, dtype=torch.float64, device=dev).index_add_(0, ix.long(), prod).cpu().numpy()
                scale[t] = float(prod.abs().sum())
            for tag, path in libs:
                lib = bind(path)
                tables, blocks = [], None
                for grads, idx, alpha in sets:
                    host = (_lib.QdDiffQuantDesc * len(ns))()
                    for i in range(len(ns)):
                        host[i].grad, host[i].idx, host[i].alpha, host[i].n = grads[i].data_ptr(), idx[i].data_ptr(), alpha[i].data_ptr(), ns[i]
                    b = ctypes.c_int64(0)
                    lib.qd_multi_dq_plan(host, len(ns), 256, ctypes.byref(b))
                    blocks = int(b.value)
                    tables.append(torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev))
                scratch = torch.empty(blocks * k, device=dev)
                out = torch.empty(len(ns), k, device=dev)

                def launch(i):
                    rc = lib.qd_multi_point_grad_f32(tables[i % nset].data_ptr(), len(ns), blocks, 256, k, out.data_ptr(), scratch.data_ptr(),
                                                     scratch.numel() * 4, _lib.stream_ptr(dev))
                    assert rc == 0, rc
                launch(0)
                first = out.clone()
                same = True
                for _ in range(20):
                    launch(0)
                    same = same and torch.equal(out, first)
                err = float(np.max(np.abs(first.cpu().numpy().astype(np.float64) - ref).max(axis=1) / np.maximum(scale, 1e-30)))
                us, lo, hi = kernel_bench.time_row(launch, iters=40, reps=5)
                print('%-8s k=%-3d %-8s rows %5d | %8.2f us (%.2f-%.2f) | %6.0f GB/s (5 B/elem) = %.3f of 8 TB/s | max err / sum|g a| %.2e | 20 launches bit-identical: %s'
                      % (model, k, tag, blocks, us, lo, hi, 5 * tot / us / 1e3, 5 * tot / us / 1e3 / 8000.0, err, same), flush=True)
            if model == 'one64Mi':                           # the single-tensor entry point on the same data
                lib = bind(_lib.LIB_PATH)
                res, args = _lib.SIGNATURES['qd_point_grad_f32']
                lib.qd_point_grad_f32.restype, lib.qd_point_grad_f32.argtypes = res, args
                ws = _lib.workspace(dev)
                out1 = torch.empty(k, device=dev)

                def single(i):
                    gr, ix, al = sets[i % nset]
                    rc = lib.qd_point_grad_f32(gr[0].data_ptr(), ix[0].data_ptr(), 1, al[0].data_ptr(), ns[0], 256, k, out1.data_ptr(), ws.data_ptr(),
                                               ws.numel(), _lib.stream_ptr(dev))
                    assert rc == 0, rc
                us, lo, hi = kernel_bench.time_row(single, iters=40, reps=5)
                print('%-8s k=%-3d %-8s            | %8.2f us (%.2f-%.2f) | %6.0f GB/s (5 B/elem) = %.3f of 8 TB/s  (qd_point_grad_f32, the single-tensor call)'
                      % (model, k, 'single', us, lo, hi, 5 * tot / us / 1e3, 5 * tot / us / 1e3 / 8000.0), flush=True)
            del sets
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
