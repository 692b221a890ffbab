"""GPU tests of the native common-case entry point of uniformQuantization (csrc/qd_torch_glue.cpp:
uniform_common): the configuration the reference's training loops use on every parameter tensor every step
(ref: cnn_models/conv_forward_model.py:216-221, 235-247) goes through ONE native call that also builds the
ScalingFunction and takes alpha / beta from a slab shared by many calls.  What is pinned here: the object it
returns behaves like one built by the general path (bit-exact fields, same shapes, lazy fields, inverse), slices
of the slab never alias -- not across calls, slab roll-overs or streams -- and everything outside that
configuration still reaches the general path with the reference's exceptions."""
import numpy as np
import pytest
import torch

import quantization
from oracle import oracle_c as oc
from quantization.quant_functions import ScalingFunction
from quantized_distillation_amd import _lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    _lib.load()
    oc.build()


def host(t):
    return t.detach().cpu().numpy()


def took_common_path(sf):
    return '_ab_slab' in sf.__dict__ or ('type_scaling' not in sf.__dict__ and '_ab' in sf.__dict__)


@pytest.mark.parametrize('n,bucket', [(100003, 256), (500, 256), (7, 256), (4096, 64), (100003, None), (3, None),
                                      (77777, 100), (1 << 20, 33)])
def test_common_path_matches_the_oracle_and_the_general_path(n, bucket):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    xd = x.to(DEV).view(-1)
    q, sf = quantization.uniformQuantization(xd, 16, bucket_size=bucket)
    assert isinstance(sf, ScalingFunction) and took_common_path(sf)
    ref = oc.uniform_quantize(x.numpy(), 16, bucket)
    assert np.array_equal(host(q), ref['q'])
    assert np.array_equal(host(sf.alpha).reshape(-1), ref['alpha'].reshape(-1))
    assert np.array_equal(host(sf.beta).reshape(-1), ref['beta'].reshape(-1))
    # the same object through the general path (modify_in_place=False spelled as a non-default clamp is not available, so
    # take stochastic_rounding=False + subtract_mean=False via the ScalingFunction API instead)
    sf2 = ScalingFunction('linear', False, False, bucket)
    u2 = sf2.scale_down(xd)
    assert sf.alpha.shape == sf2.alpha.shape and sf.beta.shape == sf2.beta.shape
    assert torch.equal(sf.alpha, sf2.alpha) and torch.equal(sf.beta, sf2.beta)
    assert sf.expected_tensor_size == sf2.expected_tensor_size
    assert sf.original_tensor_length == n and tuple(sf.original_tensor_size) == (n,)
    assert sf.type_scaling == 'linear' and sf.max_element is False and sf.subtract_mean is False
    assert sf.bucket_size == bucket and sf.modify_in_place is True and sf.mean_tensor == 0 and sf.tol_diff_zero == 1e-10
    assert torch.equal(sf.idx_min_rows, sf2.idx_min_rows) and torch.equal(sf.idx_max_rows, sf2.idx_max_rows)
    # the inverse works on it and equals the general object's
    assert torch.equal(sf.inv_scale_down(u2.clone()), sf2.inv_scale_down(u2))


def test_shapes_and_views_are_kept():
    x = torch.randn(12, 50, 3, device=DEV)
    q, sf = quantization.uniformQuantization(x, 4, bucket_size=256)
    assert q.shape == x.shape and q.is_contiguous() and q.data_ptr() != x.data_ptr()
    assert tuple(sf.original_tensor_size) == (12, 50, 3)
    assert tuple(sf.alpha.shape) == (8, 1) and tuple(sf.beta.shape) == (8, 1)
    q1, sf1 = quantization.uniformQuantization(x, 4)
    assert tuple(sf1.alpha.shape) == (1,) and tuple(sf1.beta.shape) == (1,)
    assert float(sf1.alpha) == float(x.max() - x.min()) and float(sf1.beta) == float(x.min())
    # p.data = q, as the reference loop rebinds it (ref: conv_forward_model.py:243)
    p = torch.nn.Parameter(x.clone())
    p.data = quantization.uniformQuantization(p.data, 16, bucket_size=256)[0]
    assert p.data.shape == x.shape


def test_slab_slices_never_alias_across_calls_and_rollovers():
    """Every retained ScalingFunction keeps ITS alpha / beta: small calls share a slab, a slab that fills up is replaced
    (the old one lives on through the objects carved from it), requests above a quarter slab get their own tensor."""
    g = torch.Generator().manual_seed(5)
    kept = []
    sizes = [300, 5000, 256 * 20000, 1000, 256 * 30000 + 5, 64, 256 * 40000, 256 * 20000, 17, 256 * 70000, 256 * 25000, 999]
    for rep in range(3):
        for n in sizes:
            x = torch.randn(n, generator=g) * float(1 + rep)
            q, sf = quantization.uniformQuantization(x.to(DEV), 16, bucket_size=256)
            kept.append((x, q, sf))
    # where each object's pair lives, BEFORE it is read (reading copies the 2 * nb floats out and lets the slab go)
    slabs = set()
    by_slab = {}
    for x, q, sf in kept:
        nb = (x.numel() + 255) // 256
        if sf.__dict__.get('_ab_slab') is not None:
            slabs.add(sf._ab_slab.data_ptr())
            by_slab.setdefault(sf._ab_slab.data_ptr(), []).append((sf._ab_off, sf._ab_off + 2 * nb))
        else:
            assert 2 * nb > (1 << 18) // 4                    # the big ones were allocated on their own
    assert len(slabs) >= 3, 'the sizes above fill more than two 1 MiB slabs'
    # ranges carved from one slab are disjoint
    for ranges in by_slab.values():
        ranges.sort()
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 <= b0
    for x, q, sf in kept:
        ref = oc.uniform_quantize(x.numpy(), 16, 256)
        assert np.array_equal(host(q), ref['q'])
        assert np.array_equal(host(sf.alpha).reshape(-1), ref['alpha'].reshape(-1))
        assert np.array_equal(host(sf.beta).reshape(-1), ref['beta'].reshape(-1))
        assert sf.__dict__.get('_ab_slab') is None            # read: the slab is released


def test_slab_is_per_stream():
    x = torch.randn(70001, device=DEV)
    q0, sf0 = quantization.uniformQuantization(x, 16, bucket_size=256)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        q1, sf1 = quantization.uniformQuantization(x, 16, bucket_size=256)
    side.synchronize()
    torch.cuda.synchronize()
    assert sf0._ab_slab.data_ptr() != sf1._ab_slab.data_ptr()          # (before the pairs are read: reading releases the slab)
    assert torch.equal(q0, q1) and torch.equal(sf0.alpha, sf1.alpha) and torch.equal(sf0.beta, sf1.beta)


def test_everything_else_takes_the_general_path():
    x = torch.randn(5000, device=DEV)
    want_q, want_sf = quantization.uniformQuantization(x, 16, bucket_size=256)
    for s in (np.int64(16), 16.0):                               # not an exact int: general path, same result
        q, sf = quantization.uniformQuantization(x, s, bucket_size=256)
        assert torch.equal(q, want_q) and torch.equal(sf.alpha, want_sf.alpha)
    xt = torch.randn(64, 80, device=DEV).t()                     # not contiguous
    q, sf = quantization.uniformQuantization(xt, 16, bucket_size=256)
    assert q.shape == xt.shape
    ref = oc.uniform_quantize(host(xt.contiguous()).reshape(-1), 16, 256)
    assert np.array_equal(host(q).reshape(-1), ref['q'])
    q, sf = quantization.uniformQuantization(torch.empty(0, device=DEV), 16, bucket_size=256)
    assert q.numel() == 0
    for bad in (True, 0, -3, 2.0, np.int64(256)):
        with pytest.raises(ValueError):
            quantization.uniformQuantization(x, 16, bucket_size=bad)
    with pytest.raises(ValueError):
        quantization.uniformQuantization(x, 1, bucket_size=256)
    # a CPU tensor: computed by libqd_host.so, result on the CPU, the same bits as the device's
    q_cpu, sf_cpu = quantization.uniformQuantization(x.cpu(), 16, bucket_size=256)
    assert q_cpu.device.type == 'cpu' and torch.equal(q_cpu, want_q.cpu()) and torch.equal(sf_cpu.alpha, want_sf.alpha.cpu())
    with pytest.raises(TypeError):
        quantization.uniformQuantization(x.double(), 16, bucket_size=256)
    # in place, clamp, mean, stochastic: general path, untouched behaviour
    y = x.clone()
    q, sf = quantization.uniformQuantization(y, 16, bucket_size=256, modify_in_place=True)
    assert q.data_ptr() == y.data_ptr() and torch.equal(q, want_q)


def test_lazy_arg_indices_guard_still_holds():
    x = torch.randn(3000, device=DEV)
    q, sf = quantization.uniformQuantization(x, 16, bucket_size=256)
    x.add_(1.0)
    with pytest.raises(RuntimeError):
        sf.idx_min_rows


def test_lazy_arg_indices_on_a_noncontiguous_view_of_a_modified_base():
    """Round-2 advisor finding: the guard compared the version counter of the ORIGINAL tensor with the one recorded for
    the contiguous copy actually retained, so `w.t()` of a parameter that had ever been written in place raised a
    spurious 'modified in place' error.  The guard now follows the retained tensor."""
    y = torch.randn(64, 48, device=DEV)
    y.add_(1.0)                                              # version counter > 0
    q, sf = quantization.uniformQuantization(y.t(), 16, bucket_size=256)
    want = oc.uniform_quantize(host(y.t().contiguous()), 16, 256)
    assert np.array_equal(host(q), want['q'])
    sf2 = ScalingFunction('linear', False, False, 256)
    sf2.scale_down(y.t().contiguous())
    assert torch.equal(sf.idx_min_rows, sf2.idx_min_rows) and torch.equal(sf.idx_max_rows, sf2.idx_max_rows)
    # and through the general path's other entry (max_element set -> not the common path)
    q3, sf3 = quantization.uniformQuantization(y.t(), 16, bucket_size=256, max_element=10.0)
    assert torch.equal(sf3.idx_min_rows, sf2.idx_min_rows)


def test_inference_mode_tensors_quantize():
    """Round-2 advisor finding: inference tensors do not track a version counter (`._version` raises), and the common
    path declines them, so uniformQuantization / scale_down / nonUniformQuantization failed outright under
    torch.inference_mode()."""
    with torch.inference_mode():
        x = torch.randn(5000, device=DEV)
        q, sf = quantization.uniformQuantization(x, 16, bucket_size=256)
        want = oc.uniform_quantize(host(x), 16, 256)
        assert np.array_equal(host(q), want['q'])
        assert np.array_equal(host(sf.alpha).reshape(-1), want['alpha'].reshape(-1))
        sf2 = ScalingFunction('linear', False, False, 256)
        u = sf2.scale_down(x)
        assert np.array_equal(host(u).reshape(-1)[:5000], oc.scale_down(host(x), 256)['u'])
        assert sf2.idx_min_rows.shape == (20, 1) and torch.equal(sf.idx_min_rows, sf2.idx_min_rows)
        pts = torch.tensor([0.0, 0.3, 0.7, 1.0], device=DEV)
        qn, idx, _ = quantization.nonUniformQuantization(x, pts, bucket_size=256)
        r = oc.nonuniform_quantize(host(x), host(pts), 256)
        assert np.array_equal(host(qn), r['q']) and np.array_equal(host(idx), r['idx'])


def test_retained_scaling_functions_do_not_pin_slabs():
    """alpha / beta of the common path live in a 1 MiB slab shared by many calls; an object that is KEPT (the Huffman
    accounting keeps one ScalingFunction per tensor, ref: help_functions.py:200-233) copies its 2 * nb floats out when they
    are first read and lets the slab go.  64 kept objects, each carved from a different slab: device memory held by them
    drops from 64 MiB to a few KiB once their alpha has been read."""
    torch.cuda.synchronize()
    small = torch.randn(200, device=DEV)
    filler = torch.randn(256 * 30000, device=DEV)            # 30000 buckets = 60000 floats: a fresh slab every 4 calls
    kept = []
    for i in range(64):
        kept.append(quantization.uniformQuantization(small, 16, bucket_size=256)[1])
        for _ in range(5):
            quantization.uniformQuantization(filler, 16, bucket_size=256)
    assert all(took_common_path(sf) for sf in kept)
    slabs = {sf._ab_slab.data_ptr() for sf in kept}
    assert len(slabs) >= 32, 'the test did not spread the kept objects over many slabs'
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    for sf in kept:
        assert sf.alpha.shape == (1, 1) and sf._ab_slab is None
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated()
    assert before - after >= (len(slabs) - 2) * (1 << 20), (before, after, len(slabs))
    want = oc.uniform_quantize(host(small), 16, 256)
    assert all(np.array_equal(host(sf.alpha).reshape(-1), want['alpha'].reshape(-1)) for sf in kept)
