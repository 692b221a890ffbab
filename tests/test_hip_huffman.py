"""The Huffman size accounting behind the reference's own function (quantization/help_functions.py:175-232) on the device:
digitize + histogram kernels against numpy, the boundary function against the staged REFERENCE running on the host, the
levels-only form of the quantize kernel, and the bookkeeping of tensors that kernels write in place."""
import numpy as np
import pytest
import torch

import quantization
import quantization.help_functions as qhf
from oracle import oracle_c as oc
from oracle import ref_stage
from quantized_distillation_amd import _lib, codec

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def digitize_hist(v, edges):
    """Device counters of qd_digitize_histogram_f32 for a device fp32 tensor and host float64 edges."""
    e = torch.from_numpy(np.asarray(edges, dtype=np.float64)).to(DEV)
    h = qhf._device_counts('digitize', v, len(edges), e)
    assert h is not None and h.dtype == torch.int64 and h.numel() == len(edges) + 1
    return host(h)


def want_digitize(v, edges):
    c = np.digitize(v, np.asarray(edges, dtype=np.float64))
    return np.bincount(c, minlength=len(edges) + 1)


def test_digitize_histogram_equals_numpy():
    rng = np.random.RandomState(0)
    for s in (2, 4, 16, 200, 256):
        edges = qhf._digitize_edges(s, 1e-5)
        for n in (1, 3, 4, 5, 4099, (1 << 21) + 7):
            # values ON the level positions, just below / above the edges, outside [0, 1], NaN, +-inf
            lev = rng.randint(0, s, size=n)
            v = (lev / (s - 1)).astype(np.float32)
            jitter = rng.choice([0.0, 1e-5, -1e-5, 2e-5, -2e-5, 1e-7, -1e-7], size=n).astype(np.float32)
            v = (v + jitter).astype(np.float32)
            if n > 10:
                v[1], v[2], v[3], v[4], v[5] = np.nan, np.inf, -np.inf, -0.5, 1.5
                v[6:10] = np.float32(edges[min(3, s - 1)])            # the float32 nearest an edge
            for off in (0, 1, 3):                                       # views that start 0 / 4 / 12 bytes into a 16-byte granule
                if off >= n:
                    continue
                got = digitize_hist(dev(v)[off:], edges)
                assert np.array_equal(got, want_digitize(v[off:], edges)), (s, n, off)
    # the float32 neighbourhood of every edge: the kernel compares in float32 against thr[j] = the smallest float32 whose
    # promotion is >= edges[j]; one ulp either side of every edge must land where numpy's float64 comparison puts it
    for s in (2, 16, 255, 256):
        edges = qhf._digitize_edges(s, 1e-5)
        f = edges.astype(np.float32)
        v = np.concatenate([f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf)),
                            np.nextafter(np.nextafter(f, np.float32(np.inf)), np.float32(np.inf))]).astype(np.float32)
        assert np.array_equal(digitize_hist(dev(v), edges), want_digitize(v, edges)), s
    # edges float32 cannot hold: beyond its range, inside its denormals, zeros of either sign, exactly representable ones
    edges = np.array([-1e300, -3.5e38, -1.0, -1e-320, -0.0, 1e-320, 1e-46, 1.401298464324817e-45, 1e-40, 0.1, 0.5, 1.0, 3.4028234663852886e38,
                      3.5e38, 1e300], dtype=np.float64)
    f = edges.astype(np.float32)
    v = np.concatenate([f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf)),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4028235e38, -3.4028235e38], dtype=np.float32)]).astype(np.float32)
    with np.errstate(all='ignore'):
        assert np.array_equal(digitize_hist(dev(v), edges), want_digitize(v, edges))
    # edges that are not evenly spaced (the walk from the linear guess), duplicates of the first / last region
    edges = np.sort(np.concatenate([rng.rand(40) ** 3, [0.0, 0.5, 0.5000001, 1.0]]))
    v = rng.rand(1 << 18).astype(np.float32) * 1.2 - 0.1
    assert np.array_equal(digitize_hist(dev(v), edges), want_digitize(v, edges))
    one = np.array([0.25])
    assert np.array_equal(digitize_hist(dev(v), one), want_digitize(v, one))
    # empty input: zeroed counters
    assert digitize_hist(torch.empty(0, device=DEV), edges).sum() == 0


def test_one_pass_rescale_digitize_histogram_equals_the_two_kernel_form_and_numpy():
    """qd_scale_digitize_histogram_f32 (re-scale + digitize + count in one pass over the quantized tensor) against (a) the
    two-kernel form it replaces -- ScalingFunction.scale_down on the device, then qd_digitize_histogram_f32 -- and (b) numpy:
    np.digitize of the oracle's scale_down, the reference's own steps (help_functions.py:215-218).  Bit-exact counters for
    every register-resident bucket size, ragged lengths, quantized tensors and adversarial ones (raw values, buckets a few
    ulps wide, constant buckets, NaN / inf buckets, denormal ranges)."""
    from oracle import oracle_np as onp
    rng = np.random.RandomState(11)
    for bucket in qhf.FUSED_DIGITIZE_BUCKETS:
        for s in (2, 4, 16, 256):
            edges = qhf._digitize_edges(s, 1e-5)
            edges_dev = torch.from_numpy(edges).to(DEV)
            for n in (1, bucket - 1, bucket, bucket + 1, 7 * bucket + 5, 64 * bucket, (1 << 20) + 3 * bucket + 17):
                x = rng.randn(n).astype(np.float32) * np.float32(rng.choice([1e-3, 0.05, 1.0, 300.0]))
                kinds = [('quantized', host(quantization.uniformQuantization(dev(x), s, bucket_size=bucket)[0]))]
                if s == 16:
                    raw = x.copy()
                    nb = n // bucket
                    if nb >= 6:
                        raw[0:bucket] = np.float32(1.0) + np.spacing(np.float32(1.0)) * rng.randint(0, 4, size=bucket).astype(np.float32)
                        raw[bucket:2 * bucket] = 0.25                                    # constant: alpha -> 1
                        raw[2 * bucket + 3] = np.nan                                     # the whole bucket becomes NaN
                        raw[3 * bucket + 1] = np.inf
                        raw[4 * bucket + 2] = -np.inf
                        raw[5 * bucket:6 * bucket] = (rng.rand(bucket) * 1e-41).astype(np.float32)   # denormal range: IEEE division path
                    kinds.append(('raw', raw))
                for kind, q in kinds:
                    qd = dev(q)
                    sf = quantization.ScalingFunction('linear', False, False, bucket_size=bucket)
                    got = qhf._fused_rescale_counts(qd, sf, s, edges_dev)
                    assert got is not None, (bucket, s, n, kind)
                    got = host(got)
                    two = qhf._device_counts('digitize', sf.scale_down(qd).view(-1)[0:n], s, edges_dev)
                    assert np.array_equal(got, host(two)), (bucket, s, n, kind, got, host(two))
                    with np.errstate(all='ignore'):
                        u = onp.scale_down(q, bucket)['u'].reshape(-1)[:n]
                    assert np.array_equal(got, want_digitize(u, edges)), (bucket, s, n, kind)
                    assert got.sum() == n
    # what the one-pass form does not take: the caller falls back to the two-kernel form
    q = dev(rng.randn(4096).astype(np.float32))
    e16 = torch.from_numpy(qhf._digitize_edges(16, 1e-5)).to(DEV)
    assert qhf._fused_rescale_counts(q, quantization.ScalingFunction('linear', False, False, bucket_size=100), 16, e16) is None
    assert qhf._fused_rescale_counts(q, quantization.ScalingFunction('linear', False, False, bucket_size=None), 16, e16) is None
    assert qhf._fused_rescale_counts(q, quantization.ScalingFunction('linear', False, True, bucket_size=256), 16, e16) is None
    assert qhf._fused_rescale_counts(q, quantization.ScalingFunction('linear', 0.5, False, bucket_size=256), 16, e16) is None
    assert qhf._fused_rescale_counts(q[1:], quantization.ScalingFunction('linear', False, False, bucket_size=256), 16, e16) is None   # 4-byte offset view
    assert qhf._fused_rescale_counts(q.cpu(), quantization.ScalingFunction('linear', False, False, bucket_size=256), 16, e16) is None
    empty = qhf._fused_rescale_counts(q[:0], quantization.ScalingFunction('linear', False, False, bucket_size=256), 16, e16)
    assert empty is not None and int(empty.sum()) == 0


def test_index_histogram_equals_numpy():
    rng = np.random.RandomState(1)
    for k_used in (1, 4, 16, 255, 256):
        for n in (1, 2, 3, 1001, (1 << 20) + 3):
            idx = rng.randint(0, k_used, size=n).astype(np.int64)
            for off in (0, 1):
                if off >= n:
                    continue
                h = qhf._device_counts('index', dev(idx)[off:], 256)
                assert h.numel() == 257
                assert np.array_equal(host(h), np.bincount(idx[off:], minlength=257)), (k_used, n, off)
    idx = rng.randint(-5, 300, size=50000).astype(np.int64)                # outside the table: counted in the last entry
    h = host(qhf._device_counts('index', dev(idx), 256))
    inside = (idx >= 0) & (idx < 256)
    assert np.array_equal(h[:256], np.bincount(idx[inside], minlength=256)) and h[256] == (~inside).sum()
    assert qhf._device_counts('index', dev(idx).to(torch.int32), 256) is None     # other dtypes: the host path
    assert qhf._device_counts('digitize', torch.zeros(4), 16, None) is None       # host tensors: the host path


def _student_like_params():
    from harness import models
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(3)
    out = []
    for p in models.student().parameters():
        t = p.detach().clone()
        if t.dim() <= 1:
            t = t + 0.01 * torch.randn(t.shape, generator=g)
        out.append(t.contiguous())
    return out


def test_boundary_function_equals_the_reference_on_the_student_shapes(monkeypatch):
    """get_huffman_encoding_mean_bit_length of this package (device tensors, counting on the device) against the staged
    reference's own function on the same values on the host: to 1e-12, for the 4-bit / 2-bit / 8-bit uniform settings of the
    reference's drivers (cifar10_test.py:300-330) with and without buckets, and for non-uniform points.  While it runs,
    nothing larger than the counters may be copied to the host."""
    refq = ref_stage.load()
    assert refq is not None, 'oracle/_ref is not staged (run __graft_entry__.build() where /root/reference exists)'
    import importlib
    refqhf = importlib.import_module(refq.__name__ + '.help_functions')
    params = _student_like_params()
    params_d = [p.to(DEV) for p in params]
    copied = []
    real_cpu = torch.Tensor.cpu

    def spy_cpu(self, *a, **kw):
        if self.is_cuda:
            copied.append(self.numel())
        return real_cpu(self, *a, **kw)
    one_pass = []
    real_fused = qhf._fused_rescale_counts

    def spy_fused(*a, **kw):
        h = real_fused(*a, **kw)
        one_pass.append(h is not None)
        return h
    for s, bucket in ((16, 256), (4, 256), (256, 256), (16, None), (4, None), (16, 100), (16, 64), (4, 2048)):
        want = refqhf.get_huffman_encoding_mean_bit_length(iter(params), lambda t: refq.uniformQuantization(t, s, bucket_size=bucket),
                                                           'uniform', s=s)
        monkeypatch.setattr(torch.Tensor, 'cpu', spy_cpu)
        monkeypatch.setattr(qhf, '_fused_rescale_counts', spy_fused)
        got = qhf.get_huffman_encoding_mean_bit_length(iter(params_d), lambda t: quantization.uniformQuantization(t, s, bucket_size=bucket),
                                                       'uniform', s=s)
        monkeypatch.undo()
        assert abs(got - want) < 1e-12, (s, bucket, got, want)
        assert copied and max(copied) <= s + 1, (s, bucket, max(copied))
        # bucketed with a register-resident size: re-scale + digitize + count is ONE kernel per tensor; otherwise the two-kernel form
        assert len(one_pass) == len(params) and all(one_pass) == (bucket in qhf.FUSED_DIGITIZE_BUCKETS) and any(one_pass) == all(one_pass), (s, bucket)
        del copied[:], one_pass[:]
        # the level histogram of the codec gives the same lengths on such (non-degenerate) tensors
        if bucket in (256, None):
            assert abs(codec.huffman_mean_bit_length_uniform(params_d, s, bucket) - want) < 1e-12
    pts = [0.0, 0.21, 0.48, 0.52, 0.77, 1.0]
    want = refqhf.get_huffman_encoding_mean_bit_length(iter(params), lambda t: refq.nonUniformQuantization(t, pts, bucket_size=256), 'nonuniform')
    monkeypatch.setattr(torch.Tensor, 'cpu', spy_cpu)
    got = qhf.get_huffman_encoding_mean_bit_length(iter(params_d), lambda t: quantization.nonUniformQuantization(t, pts, bucket_size=256), 'nonuniform')
    monkeypatch.undo()
    assert abs(got - want) < 1e-12
    assert len(copied) == 1 and copied[0] == 257 * len(params)      # ONE copy for the model: the [tensors][256 + 1] counters
    # a list of functions (one per tensor), and more symbols than the device tables hold: counted on the host, same result
    fns_ref = [lambda t, s=s: refq.uniformQuantization(t, s, bucket_size=256) for s in [1000] * len(params)]
    fns = [lambda t, s=s: quantization.uniformQuantization(t, s, bucket_size=256) for s in [1000] * len(params)]
    want = refqhf.get_huffman_encoding_mean_bit_length(iter(params), fns_ref, 'uniform', s=1000)
    assert abs(qhf.get_huffman_encoding_mean_bit_length(iter(params_d), fns, 'uniform', s=1000) - want) < 1e-12


def test_boundary_function_follows_the_reference_where_levels_and_digitized_rescaling_differ():
    """The reference does not count level indices: it re-scales the QUANTIZED tensor and digitizes that (:215-217).  In a bucket
    whose range is a few ulps of its offset the two differ (the quantized values collapse onto a coarse grid); the boundary
    function must follow the reference there, whatever the level histogram says."""
    refq = ref_stage.load()
    assert refq is not None
    import importlib
    refqhf = importlib.import_module(refq.__name__ + '.help_functions')
    rng = np.random.RandomState(5)
    x = rng.randn(64 * 256).astype(np.float32)
    base = np.float32(1.0)
    ulp = np.spacing(base)
    for b in range(0, 64, 3):                                   # every third bucket: values within 3 ulps of 1.0
        x[b * 256:(b + 1) * 256] = base + ulp * rng.randint(0, 4, size=256).astype(np.float32)
    x[5 * 256:6 * 256] = 0.25                                   # a constant bucket (alpha -> 1)
    t = torch.from_numpy(x)
    for s in (4, 16):
        want = refqhf.get_huffman_encoding_mean_bit_length(iter([t]), lambda v: refq.uniformQuantization(v, s, bucket_size=256), 'uniform', s=s)
        got = qhf.get_huffman_encoding_mean_bit_length(iter([t.to(DEV)]), lambda v: quantization.uniformQuantization(v, s, bucket_size=256),
                                                       'uniform', s=s)
        assert abs(got - want) < 1e-12, (s, got, want)


def test_levels_only_form_of_the_quantize_kernel():
    """qd_uniform_f32 with q == NULL: level indices (and alpha / beta when asked for), no quantized tensor written."""
    lib = _lib.load()
    oc.build()
    rng = np.random.RandomState(2)
    ws = _lib.workspace(DEV)
    for bucket in (64, 128, 256, 512, 1024, 2048):
        for n in (bucket * 37, bucket * 37 + 5, bucket - 1, 3):
            for s in (2, 4, 16, 256):
                x = rng.randn(n).astype(np.float32)
                xd = dev(x)
                lev = torch.full((n + 8,), 255, dtype=torch.uint8, device=DEV)
                nb = 1 if n < bucket else -(-n // bucket)
                ab = torch.zeros(2, nb, device=DEV)
                rc = lib.qd_uniform_f32(xd.data_ptr(), None, n, bucket, s, ab[0].data_ptr(), ab[1].data_ptr(), lev.data_ptr(), None, 0, 0.0,
                                        0, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
                assert rc == 0, (bucket, n, s)
                ref = oc.uniform_quantize(x, s, bucket)
                assert np.array_equal(host(lev)[:n], ref['lev']), (bucket, n, s)
                assert np.all(host(lev)[n:] == 255), 'wrote past the end'
                assert np.array_equal(host(ab[0]), ref['alpha'].reshape(-1)) and np.array_equal(host(ab[1]), ref['beta'].reshape(-1))
                # alpha / beta are optional
                lev2 = torch.empty(n, dtype=torch.uint8, device=DEV)
                assert lib.qd_uniform_f32(xd.data_ptr(), None, n, bucket, s, None, None, lev2.data_ptr(), None, 0, 0.0, 0, 0,
                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr()) == 0
                assert np.array_equal(host(lev2), ref['lev'])
    x = dev(rng.randn(4096).astype(np.float32))
    lev = torch.empty(4096, dtype=torch.uint8, device=DEV)
    args = (None, None, lev.data_ptr(), None, 0, 0.0)
    # outside the form's geometry: refused, not computed some other way
    assert lib.qd_uniform_f32(x.data_ptr(), None, 4096, 100, 16, *args, 0, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr()) == -3
    assert lib.qd_uniform_f32(x.data_ptr(), None, 4096, 0, 16, *args, 0, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr()) == -3
    assert lib.qd_uniform_f32(x.data_ptr(), None, 4096, 256, 16, *args, 1, 7, ws.data_ptr(), ws.numel(), _lib.stream_ptr()) == -3   # stochastic
    assert lib.qd_uniform_f32(x.data_ptr() + 4, None, 4092, 256, 16, *args, 0, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr()) == -3  # misaligned
    assert lib.qd_uniform_f32(x.data_ptr(), None, 4096, 256, 16, None, None, None, None, 0, 0.0, 0, 0, ws.data_ptr(), ws.numel(),
                              _lib.stream_ptr()) == -1                                                                                # nothing to write
    # codec.level_histogram uses it where it applies and the q-writing form elsewhere: same counts
    xh = rng.randn(100003).astype(np.float32)
    for s, bucket in ((16, 256), (16, 100), (4, None), (256, 2048)):
        assert np.array_equal(host(codec.level_histogram(dev(xh), s, bucket)), np.bincount(oc.uniform_quantize(xh, s, bucket)['lev'], minlength=s))
    assert np.array_equal(host(codec.level_histogram(dev(xh)[1:], 16, 256)), np.bincount(oc.uniform_quantize(xh[1:], 16, 256)['lev'], minlength=16))
    # non-finite buckets: the one-pass form (qd_level_histogram_f32) and the write-the-levels-then-count form must give the same
    # counts -- a NaN anywhere in a bucket, +inf, -inf, both, in full buckets and in the short last one
    xn = rng.randn(256 * 40 + 77).astype(np.float32)
    xn[5] = np.nan
    xn[256 * 3 + 17] = np.inf
    xn[256 * 7 + 1] = -np.inf
    xn[256 * 9 + 2], xn[256 * 9 + 200] = np.inf, -np.inf
    xn[256 * 11:256 * 12] = np.inf
    xn[-3] = np.inf
    xd = dev(xn)
    one_pass = host(codec.level_histogram(xd, 16, 256))
    lev = torch.empty(xn.size, dtype=torch.uint8, device=DEV)
    q = torch.empty_like(xd)
    _lib.check(lib.qd_uniform_f32(xd.data_ptr(), q.data_ptr(), xn.size, 256, 16, None, None, lev.data_ptr(), None, 0, 0.0, 0, 0,
                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    two_pass = host(codec.histogram_u8(lev, 16))
    assert one_pass.sum() == xn.size and np.array_equal(one_pass, two_pass), (one_pass, two_pass)
    assert np.array_equal(two_pass, np.bincount(host(lev), minlength=16))
    # the buckets that hold a non-finite value count as level 0 throughout
    finite_buckets = np.ones(41, bool)
    finite_buckets[[0, 3, 7, 9, 11, 40]] = False
    ref = oc.uniform_quantize(xn[:256 * 40].reshape(40, 256)[finite_buckets[:40]].reshape(-1), 16, 256)['lev']
    want = np.bincount(ref, minlength=16)
    want[0] += xn.size - ref.size
    assert np.array_equal(one_pass, want), (one_pass, want)


def test_kernels_that_write_in_place_bump_the_version_counter():
    """A kernel that writes over a tensor through its raw pointer must leave the tensor's version counter where an in-place
    torch op would: ScalingFunction's lazily computed arg indices rely on it (as does autograd's saved-tensor check)."""
    from quantized_distillation_amd import ste
    x = torch.randn(4096, device=DEV)
    pts = torch.tensor([0.0, 0.3, 0.6, 1.0], device=DEV)
    # a scaling function that lazily retains x ...
    _, sf = quantization.uniformQuantization(x, 16, bucket_size=256)
    want_min = None
    keep = x.clone()
    v0 = x._version
    # ... then an in-place non-uniform call overwrites x: the lazy indices must refuse, not return those of the new data
    quantization.nonUniformQuantization(x, pts, bucket_size=256, modify_in_place=True)
    assert x._version > v0
    assert not torch.equal(x, keep)
    with pytest.raises(RuntimeError, match='modified in place'):
        sf.idx_min_rows
    # read BEFORE the overwrite they are right and stay available
    x = keep.clone()
    _, sf = quantization.uniformQuantization(x, 16, bucket_size=256)
    want_min = sf.idx_min_rows.clone()
    quantization.nonUniformQuantization(x, pts, bucket_size=256, modify_in_place=True)
    assert torch.equal(sf.idx_min_rows, want_min)
    assert torch.equal(want_min.view(-1), keep.view(-1, 256).argmin(dim=1))
    # every other in-place entry point
    for call in (lambda t: quantization.uniformQuantization(t, 16, bucket_size=256, modify_in_place=True),
                 lambda t: quantization.uniformQuantization(t, 16, modify_in_place=True),
                 lambda t: quantization.ScalingFunction('linear', False, False, 256, modify_in_place=True).scale_down(t),
                 lambda t: ste.clamp_(t, 1.0),
                 lambda t: ste.truncated_ste_(t, keep, 1.0),
                 lambda t: ste.ste_bucket_backward(keep, t, 256, 16, out=t)):
        t = keep.clone()
        view = t.view(16, 256)                      # a view shares the counter of its base
        v = t._version
        call(t)
        assert t._version > v and view._version == t._version
    sf = quantization.ScalingFunction('linear', False, False, 256, modify_in_place=True)
    u = sf.scale_down(keep.clone())
    v = u._version
    sf.inv_scale_down(u)
    assert u._version > v
    # the multi-tensor launch writes its outputs through a device table
    from quantized_distillation_amd.multi_tensor import MultiTensorQuantizer
    outs = [torch.zeros(4096, device=DEV), torch.zeros(100, device=DEV)]
    mt = MultiTensorQuantizer([keep, keep[:100].clone()], 16, 256, outputs=outs)
    v = [o._version for o in outs]
    mt.quantize()
    assert all(o._version > v0 for o, v0 in zip(outs, v))
    # inference tensors carry no counter: the call must simply work
    with torch.inference_mode():
        t = keep.clone()
        quantization.uniformQuantization(t, 16, bucket_size=256, modify_in_place=True)
        ste.clamp_(t, 1.0)


def test_api_on_a_device_that_is_not_current():
    """The launch-geometry caches are per device: a call on another device than the first one used must not inherit its
    numbers (one process may drive several GPUs)."""
    if torch.cuda.device_count() < 2:
        pytest.skip('one HIP device visible')
    x0 = torch.randn(100003, device='cuda:0')
    x1 = x0.to('cuda:1')
    with torch.cuda.device(0):
        q0, _ = quantization.uniformQuantization(x0, 16, bucket_size=256)
        g0, _ = quantization.uniformQuantization(x0, 16)
    with torch.cuda.device(1):
        q1, _ = quantization.uniformQuantization(x1, 16, bucket_size=256)
        g1, _ = quantization.uniformQuantization(x1, 16)
    q1b, _ = quantization.uniformQuantization(x1, 16, bucket_size=256)        # current device 0, tensor on 1
    assert torch.equal(q0.cpu(), q1.cpu()) and torch.equal(g0.cpu(), g1.cpu()) and torch.equal(q1.cpu(), q1b.cpu())


def test_boundary_function_property_vs_the_reference():
    """Random models (1-4 tensors of random sizes and value kinds -- ties, constant buckets, heavy tails, denormals, mixed
    scales), s in {2, 4, 16, 256}, bucket in {None, 256, 64, 1024, 100, 33, 7}: get_huffman_encoding_mean_bit_length of this package
    on the device equals the staged reference's on the host to 1e-12, whatever the values are."""
    import importlib
    import os

    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    from test_hip_property import make
    refq = ref_stage.load()
    assert refq is not None
    refqhf = importlib.import_module(refq.__name__ + '.help_functions')
    soak = int(os.environ.get('QD_SOAK', '1'))

    @settings(max_examples=25 * soak, deadline=None, suppress_health_check=list(HealthCheck), derandomize=(soak == 1), database=None)
    @given(sizes=st.lists(st.integers(1, 30000), min_size=1, max_size=4), s=st.sampled_from([2, 4, 16, 256]),
           bucket=st.sampled_from([None, 256, 256, 64, 1024, 100, 33, 7]), seed=st.integers(0, 2 ** 31 - 1), kind=st.integers(0, 8))
    def check(sizes, s, bucket, seed, kind):
        params = [torch.from_numpy(make(n, seed + i, kind if i % 2 == 0 else 0)) for i, n in enumerate(sizes)]
        with np.errstate(all='ignore'):
            want = refqhf.get_huffman_encoding_mean_bit_length(iter(params), lambda t: refq.uniformQuantization(t, s, bucket_size=bucket), 'uniform', s=s)
        got = qhf.get_huffman_encoding_mean_bit_length(iter([p.to(DEV) for p in params]),
                                                       lambda t: quantization.uniformQuantization(t, s, bucket_size=bucket), 'uniform', s=s)
        assert abs(got - want) < 1e-12, (sizes, s, bucket, seed, kind, got, want)
    check()
