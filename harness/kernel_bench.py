"""Achieved algorithmic bandwidth of every kernel on the path, measured the same way everywhere.

(The `kernel` column names what rocprofv3 shows for the call -- tools/dispatch_map.py, docs/history/profiles/r04_dispatch_map.txt; template
arguments: mode 0 = quantize-dequantize, 1 = scale_down, 2 = nearest point.)

One table of rows (the headline's secondary rows and the per-kernel byte bases of
K2 ... K9), one timing routine, two users: bench.py's `kernels` leg (so that every row is in the
driver's record, not only in builder-run profiles) and tools/bench_kernels.py (which adds the
bucket-size sweeps and writes profiles/rNN_kernels.txt).

Method, per row: >= 100 ms of untimed back-to-back launches (past the idle-to-busy clock
transient and the slow first tens of milliseconds on fresh allocations), then REPS repetitions of
ITERS launches bracketed by HIP events on the launch stream (torch's current stream -- the one
_lib.stream_ptr() hands to the C ABI); the MEDIAN repetition is the row's time, min and max are
listed.  Every row rotates >= 3 input/output sets whose bytes per call exceed the 256 MiB
Infinity Cache, so each launch streams from HBM.  Rows go through the public Python API or the C
ABI exactly as the product calls them (allocation of results included where the API allocates).

`bytes_per_elem` is the ALGORITHMIC traffic of the row, not what the kernel
happens to move; frac = GB/s / 8000 (HBM3E peak, MI355X_MICROARCH.md).
"""
import statistics
import time

import torch

HBM_PEAK_GBPS = 8000.0
ITERS = 40
REPS = 3
PRECONDITION_S = 0.1


def model_shapes(name):
    """Parameter shapes of a BASELINE config model, without allocating it (meta device)."""
    from . import models
    with torch.device('meta'):
        if name == 'wrn':
            m = models.WideResNet(16, 22)
        elif name == 'student':
            m = models.student()
        else:
            raise ValueError(name)
    return [tuple(p.shape) for p in m.parameters()]


def time_row(fn, iters=ITERS, reps=REPS, precondition_s=PRECONDITION_S):
    """(median, min, max) microseconds per call of fn(i) -- see the module docstring."""
    t0 = time.perf_counter()
    i = 0
    while precondition_s > 0:
        for _ in range(20):
            fn(i)
            i += 1
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > precondition_s:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    samples = []
    for _ in range(reps):
        e0.record()
        for j in range(iters):
            fn(i + j)
        e1.record()
        torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(samples), min(samples), max(samples)


class Rows(object):
    """Collects rows; `marker` (optional) is called with the row name before a row starts (tools/dispatch tracing)."""

    def __init__(self, marker=None, verbose=False, only=None):
        self.rows, self.marker, self.verbose, self.only = [], marker, verbose, only

    def add(self, name, kernel, fn, bytes_per_elem, n, iters=ITERS, note=None):
        if self.only is not None and not self.only(name):
            return
        if self.marker is not None:
            self.marker(name)
        us, lo, hi = time_row(fn, iters=iters)
        gbps = bytes_per_elem * n / us / 1e3
        row = {'name': name, 'kernel': kernel, 'us': round(us, 2), 'us_min': round(lo, 2), 'us_max': round(hi, 2),
               'bytes_per_elem': bytes_per_elem, 'n': int(n), 'GBps': round(gbps, 1), 'frac': round(gbps / HBM_PEAK_GBPS, 4)}
        if note:
            row['note'] = note
        self.rows.append(row)
        if self.verbose:
            print(format_row(row), flush=True)


def format_row(r):
    return '%-44s %-34s %9.2f us (%7.2f..%7.2f)  %5.1f B/elem  %8.1f GB/s  %5.1f%% of 8 TB/s  %s' % (
        r['name'], r['kernel'], r['us'], r['us_min'], r['us_max'], r['bytes_per_elem'], r['GBps'], 100 * r['frac'], r.get('note', ''))


def flat_row(r, width=118):
    """One row as a short string (the driver's record keeps scalars of the roofline object, not nested lists, and cuts strings):
    the numbers first, the kernel name last and shortened if need be."""
    s = '%s | %.2f us | %g B/el | %.0f GB/s | %.3f | ' % (r['name'], r['us'], r['bytes_per_elem'], r['GBps'], r['frac'])
    return (s + r['kernel'])[:width]


def run(dev, log2n=26, sweeps=False, marker=None, verbose=False, only=None, hist_log2n=30):
    """All rows on device `dev`; returns the list of row dicts.  sweeps=True adds the bucket-size / point-count sweeps
    (tools only).  Peak device memory: about 9 GiB."""
    import quantization
    from quantized_distillation_amd import _lib, codec
    from quantized_distillation_amd.multi_tensor import MultiTensorDiffQuant, MultiTensorQuantizer
    lib = _lib.load()
    N = 1 << log2n
    R = 4
    out = Rows(marker, verbose, only)
    add = out.add
    gen = torch.Generator(device=dev).manual_seed(0)

    def randn(n):
        return torch.randn(n, device=dev, generator=gen)

    xs = [randn(N) for _ in range(R)]
    live = [None] * R

    def keep(i, v):
        live[i % R] = v

    def uq(src, s, b, **kw):
        return lambda i: keep(i, quantization.uniformQuantization(src[i % R], s, bucket_size=b, **kw)[0])

    # ---- K1: the headline call and SURVEY 8d's secondary rows
    add('K1 uniform 4-bit b256 (headline)', 'k_bucket_vec<0,16,4,1>', uq(xs, 16, 256), 8, N)
    add('K1 uniform 2-bit (s=4) b256', 'k_bucket_vec<0,16,4,1>', uq(xs, 4, 256), 8, N)
    xw = [x * 0.05 for x in xs]
    add('K1 4-bit b256 weight-like 0.05*randn', 'k_bucket_vec<0,16,4,1>', uq(xw, 16, 256), 8, N)
    del xw
    xr = [randn(N + 17) for _ in range(R)]
    add('K1 4-bit b256 ragged N=64Mi+17', 'k_bucket_vec<0,16,4,1>', uq(xr, 16, 256), 8, N + 17)
    del xr
    if log2n == 26:
        xd = [x[:64000000] for x in xs]
        add('K1 4-bit b256 N=64,000,000', 'k_bucket_vec<0,16,4,1>', uq(xd, 16, 256), 8, 64000000)
        del xd
    add('K1g uniform 4-bit bucket_size=None', 'k_minmax_partial+k_minmax_final+k_single_apply<0>', uq(xs, 16, None), 12, N,
        note='3 launches: reduce, fold, apply')
    if sweeps:
        add('K1s uniform 4-bit b256 stochastic', 'k_bucket_vec<0,16,4,1>', uq(xs, 16, 256, stochastic_rounding=True), 8, N)
        for b in (64, 128, 512, 1024, 2048, 4096, 8192, 100, 36, 33, 50, 250, 513, 1000, 1001, 2000, 3000, 5000, 8000):
            add('K1 uniform 4-bit bucket %d' % b, '(dispatch map)', uq(xs, 16, b), 8, N, iters=ITERS if b in (64, 128, 512, 1024, 2048) else 12)

    # ---- K2 / K3
    sf = quantization.ScalingFunction('linear', False, False, 256)
    add('K2 scale_down b256', 'k_bucket_vec<1,16,4,1>', lambda i: keep(i, sf.scale_down(xs[i % R])), 8, N)
    us_ = [sf.scale_down(x) for x in xs[:3]]
    add('K3 inv_scale_down b256', 'k_inv_scale<false>', lambda i: keep(i, sf.inv_scale_down(us_[i % 3])), 8, N)
    del us_

    # ---- K4 / K5 / K6
    gs = [randn(N) for _ in range(R)]
    for k in ((4, 16, 256) if sweeps else (4, 16)):
        pts = torch.sort(torch.rand(k, device=dev, generator=gen))[0]
        add('K4 nonUniform k=%d b256 (int64 idx)' % k, 'k_bucket_vec<2,16,4,1>',
            lambda i, pts=pts: keep(i, quantization.nonUniformQuantization(xs[i % R], pts, bucket_size=256)[:2]), 16, N,
            note='q and the int64 indices of the last 4 calls stay alive; 177-201 us box to box with one binary (docs/history/profiles/r04_ab_idx_stores.txt)')
        if k == 4:
            add('K4 nonUniform k=4 b256 (uint8 idx: index_dtype opt-in)', 'k_bucket_vec<2,16,4,1>',
                lambda i, pts=pts: keep(i, quantization.nonUniformQuantization(xs[i % R], pts, bucket_size=256, index_dtype=torch.uint8)[:2]), 9, N,
                note="the same call with one-byte indices: 9 instead of 16 B/element")
        fns = [quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=xs[j]) for j in range(3)]
        add('K5 diff-quant forward k=%d (u resident, u8 idx)' % k, 'k_nearest_prescaled_stream<false>',
            lambda i, pts=pts, fns=fns: fns[i % 3].forward(None, pts), 9, N)
        for f in fns:
            f.forward(None, pts)
        add('K6 point gradient k=%d (u8 idx)' % k, 'k_point_grad_fast<%d,1,1,4,..>+k_point_grad_final' % (4 if k <= 4 else 0),
            lambda i, fns=fns: fns[i % 3].backward(gs[i % R]), 5, N)
        del fns

    # ---- K7 / K8
    fq = [quantization.uniformQuantization_variable(16, bucket_size=256) for _ in range(R)]

    def k7(i):
        f = fq[i % R]
        f.saved_for_backward = {'input': xs[i % R]}
        keep(i, f.backward(gs[(i + 1) % R]))
    add("K7 'complicated' STE backward b256", 'k_ste_backward_vec<16,4>', k7, 12, N)
    st_ptr = _lib.stream_ptr
    add('K8 truncated STE mask, 32% of |w| > 1', 'k_truncated_ste',
        lambda i: lib.qd_truncated_ste_f32(xs[i % R].data_ptr(), gs[i % R].data_ptr(), N, 1.0, st_ptr()), 12, N,
        note="adversarial density; SURVEY 8d's basis (w read, g read + written). Zero stores that never read g take the same time "
             '(profiles/r06_ab_k8.txt): a partly written line is read-modified-written at the memory side; PMC: ~10 B/elem moved; the '
             'minimal 4 + 4 x 0.32 = 5.3 B/elem would read 0.30')
    ws_ = [x * 0.2 for x in xs]
    add('K8 truncated STE mask, nothing masked', 'k_truncated_ste',
        lambda i: lib.qd_truncated_ste_f32(ws_[i % R].data_ptr(), gs[i % R].data_ptr(), N, 1.0, st_ptr()), 4, N, note='w read only')
    add('K8 clamp to [-1,1], nothing out of range', 'k_clamp', lambda i: lib.qd_clamp_f32(ws_[i % R].data_ptr(), N, 1.0, st_ptr()), 4, N,
        note='w read only')
    del ws_, gs, fq

    # ---- codec: pack / unpack / histogram
    pks = [None] * R
    add('PK pack 4-bit levels + alpha/beta b256', 'k_pack_vec<16,4,4>', lambda i: pks.__setitem__(i % R, codec.pack_uniform(xs[i % R], 16, 256)), 4.5, N)
    pk = [codec.pack_uniform(xs[j], 16, 256) for j in range(R)]
    add('UPK unpack 4-bit -> fp32 b256', 'k_unpack<4>', lambda i: keep(i, pk[i % R].unpack()), 4.5, N)
    del pks, pk
    add('LVH level histogram of x, s=16 b256', 'k_level_hist_vec<16,4>+k_hist_fold', lambda i: codec.level_histogram(xs[i % R], 16, 256), 4, N,
        note='levels counted in the kernel that computes them: 4 B read, nothing written')
    # the Huffman accounting's per-tensor step (help_functions.py:215-223): re-scale the QUANTIZED tensor, digitize, count
    import quantization.help_functions as qhf
    qs = [quantization.uniformQuantization(x, 16, bucket_size=256)[0] for x in xs]
    e16 = torch.from_numpy(qhf._digitize_edges(16, 1e-5)).to(dev)
    sf = quantization.ScalingFunction('linear', False, False, bucket_size=256)
    add('HUF re-scale + digitize + count of q, s=16 b256 (one pass)', 'k_scale_digitize_hist_vec<16,4>+k_hist_fold',
        lambda i: qhf._fused_rescale_counts(qs[i % R], sf, 16, e16), 4, N, note='4 B read, nothing written')
    add('HUF2 the two-kernel form it replaces (scale_down, then digitize + count)', 'k_bucket_vec<1,16,4,1>+k_hist_sym<0>+k_hist_fold',
        lambda i: qhf._device_counts('digitize', sf.scale_down(qs[i % R]).view(-1), 16, e16), 12, N, note='4 r + 4 w, then 4 r')
    del qs
    live[:] = [None] * R
    del xs
    NH = 1 << hist_log2n
    lev8 = [torch.randint(0, 16, (NH,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(3)]
    for k in (16, 256):
        add('HST histogram of u8 levels k=%d, %d Mi symbols' % (k, NH >> 20), 'k_hist_atomic<2>+k_hist_fold', lambda i, k=k: codec.histogram_u8(lev8[i % 3], k), 1, NH,
            iters=12)
    del lev8

    # ---- multi-tensor kernels on the BASELINE config shape lists
    def multi_rows(tag, shapes, iters):
        sets = []
        tot = sum(torch.Size(s).numel() for s in shapes)
        nset = 3 if tot * 8 > (64 << 20) else 1
        for _ in range(nset):
            masters = [randn(torch.Size(s).numel()).view(s) for s in shapes]
            sets.append((masters, MultiTensorQuantizer(masters, 16, 256)))
        add('K9 multi-tensor uniform 4-bit, %s' % tag, 'k_multi_uniform<256>', lambda i: sets[i % nset][1].quantize(check_pointers=False), 8, tot, iters=iters)
        if sweeps:
            add('   same tensors, per-tensor API loop', '(per tensor)', lambda i: [quantization.uniformQuantization(m, 16, bucket_size=256) for m in sets[i % nset][0]],
                8, tot, iters=10)
        return sets, tot, nset

    sets, tot, nset = multi_rows('WRN-16-22 60 tensors 82.7 M', model_shapes('wrn'), ITERS)
    mdq = []
    for masters, _mt in sets:
        qs = [torch.empty_like(m) for m in masters]
        gr = [torch.randn_like(m) for m in masters]
        mdq.append(MultiTensorDiffQuant(masters, qs, gr, 4, 256))
    ptsm = torch.sort(torch.rand(len(sets[0][0]), 4, device=dev, generator=gen), dim=1)[0].contiguous()
    add('K5m multi-tensor assign k=4, WRN-16-22', 'k_multi_nearest<256>', lambda i: mdq[i % nset].forward(ptsm), 9, tot)
    add('K6m multi-tensor point gradient k=4, WRN-16-22', 'k_multi_point_grad+k_multi_point_grad_final', lambda i: mdq[i % nset].backward(), 5, tot)
    del mdq, sets
    multi_rows('CIFAR student 22 tensors 1.0 M', model_shapes('student'), 200)
    return out.rows
