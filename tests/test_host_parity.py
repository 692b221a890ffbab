"""CPU tensors through the same API ("CPU tensors must keep working"; the reference's functions accept them,
quantization/quant_functions.py:186,254,283-284): computed by libqd_host.so (csrc/host/qd_host.cpp), the per-call entry
points of include/qd_hip.h for host pointers, selected by the tensor's device.

The device-independent parity tests of tests/test_hip_parity.py -- the golden vectors produced by running the reference, the
non-finite cases, the option sweeps, the oracle comparisons -- are run again here with every tensor on the CPU: the same test
bodies, `DEV` switched.  Plus what is specific to the host library: it never touches libqd_hip.so, the staged reference
agrees bit for bit on fresh random tensors, the device-only entry points refuse CPU tensors, and a CPU tensor and the
stochastic-rounding generator of the device draw the same numbers."""
import inspect
import itertools

import numpy as np
import pytest
import torch

import quantization
import quantization.help_functions as qhf
import test_hip_parity as P
from oracle import oracle_c as oc
from oracle import ref_stage
from quantized_distillation_amd import _lib, ste

CPU_TESTS = [
    P.test_subtract_mean_where_the_last_bit_of_the_mean_flips_levels, P.test_uniform_golden, P.test_scale_down_and_inverse_golden,
    P.test_roundtrip_golden, P.test_modify_in_place_and_views, P.test_uniform_random_sweep_vs_c_oracle, P.test_nonuniform_golden,
    P.test_nonuniform_random_vs_c_oracle, P.test_search_sorted_handle_query, P.test_init_points_and_huffman_golden,
    P.test_ste_complicated_golden, P.test_ste_complicated_large_vs_c_oracle, P.test_truncated_ste_kernels,
    P.test_nonfinite_inputs_golden, P.test_nonuniform_options_golden, P.test_lazy_arg_indices_raise_after_the_source_was_modified,
    P.test_nonuniform_single_bucket_of_exactly_bucket_size, P.test_quantize_bit_exact_at_extreme_scales,
    P.test_scale_down_bit_exact_at_extreme_scales, P.test_stochastic_rounding_statistics,
]


def _cases():
    """(test function, {argname: value}) for every parametrisation of the listed tests."""
    out = []
    for fn in CPU_TESTS:
        marks = [m for m in getattr(fn, 'pytestmark', []) if m.name == 'parametrize']
        if not marks:
            out.append(pytest.param(fn, {}, id=fn.__name__))
            continue
        axes = []
        for m in marks:
            names = [a.strip() for a in m.args[0].split(',')]
            axes.append([dict(zip(names, v if len(names) > 1 else (v,))) for v in m.args[1]])
        for combo in itertools.product(*axes):
            kw = {}
            for d in combo:
                kw.update(d)
            out.append(pytest.param(fn, kw, id='%s[%s]' % (fn.__name__, '-'.join(str(v) for v in kw.values()))))
    return out


@pytest.fixture(scope='module', autouse=True)
def _built():
    _lib.host()
    oc.build()


@pytest.mark.parametrize('fn,params', _cases())
def test_parity_suite_on_cpu_tensors(fn, params, monkeypatch, request):
    monkeypatch.setattr(P, 'DEV', 'cpu')
    # on the GPU `x.to(DEV)` / dev(a) give the test a COPY it may overwrite (modify_in_place cases) while it keeps x for the
    # expected values; `.to('cpu')` of a CPU tensor is the tensor itself, so here the copy is made explicitly
    real_to = torch.Tensor.to

    def to_copy(self, *a, **k):
        r = real_to(self, *a, **k)
        return r.clone() if (a and isinstance(a[0], str) and a[0] == 'cpu' and r.data_ptr() == self.data_ptr()) else r
    monkeypatch.setattr(torch.Tensor, 'to', to_copy)
    kwargs = dict(params)
    for name in inspect.signature(fn).parameters:
        if name not in kwargs:
            kwargs[name] = request.getfixturevalue(name)
    fn(**kwargs)


def test_cpu_tensors_never_touch_the_hip_library(monkeypatch):
    """One library per device: with libqd_hip.so and the native glue made unloadable, every CPU call still works -- and a
    device-only entry point says so instead of computing somewhere else."""
    def boom(*a, **k):
        raise AssertionError('a CPU tensor reached the HIP library')
    monkeypatch.setattr(_lib, 'load', boom)
    monkeypatch.setattr(_lib, 'glue', boom)
    x = torch.randn(5000)
    q, sf = quantization.uniformQuantization(x, 16, bucket_size=256)
    assert q.device.type == 'cpu' and sf.alpha.device.type == 'cpu' and sf.idx_min_rows.device.type == 'cpu'
    quantization.uniformQuantization(x.clone(), 16, bucket_size=256, modify_in_place=True, subtract_mean=True, max_element=1.5)
    pts = torch.tensor([0.0, 0.3, 0.7, 1.0])
    qn, idx, _ = quantization.nonUniformQuantization(x, pts, bucket_size=256)
    assert idx.dtype == torch.int64 and idx.device.type == 'cpu'
    fn = quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=x)
    fn.forward(None, pts)
    _, gp = fn.backward(torch.randn(5000))
    assert gp.shape == (4,) and gp.device.type == 'cpu'
    fu = quantization.uniformQuantization_variable(16, bucket_size=256)
    fu.forward(x)
    assert fu.backward(torch.randn(5000)).shape == x.shape
    ste.clamp_(x.clone(), 1.0)
    ste.truncated_ste_(torch.randn(5000), x, 1.0)
    qhf.initialize_quantization_points(x, quantization.ScalingFunction('linear', False, False, 256), 8)
    # the entry points that exist for device tensors only refuse, loudly
    from quantized_distillation_amd import codec
    from quantized_distillation_amd.multi_tensor import MultiTensorQuantizer
    with pytest.raises(RuntimeError, match='HIP device'):
        MultiTensorQuantizer([x], 16, 256)
    with pytest.raises(RuntimeError, match='HIP device'):
        codec.level_histogram(x, 16, 256)
    with pytest.raises(NotImplementedError, match='HIP device'):
        quantization.uniformQuantization(x, 16, type_of_scaling='absmax', bucket_size=256)


def test_a_missing_hip_library_is_still_an_error_for_device_work(monkeypatch, tmp_path):
    """The host library is not a fallback: with libqd_hip.so absent, loading the device library fails loudly even though
    libqd_host.so is there."""
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libqd_hip.so'))
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(_lib.QdLibraryMissing, match='no CPU fallback'):
        _lib.load()
    assert _lib.host() is not None


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_fresh_random_tensors_against_the_staged_reference(seed):
    """Not only the committed goldens: the reference itself (oracle/_ref, staged bytecode) on tensors generated now."""
    refq = ref_stage.load()
    assert refq is not None, 'oracle/_ref is not staged (run __graft_entry__.build() where /root/reference exists)'
    g = torch.Generator().manual_seed(seed)
    for n, bucket, s, k in [(100003, 256, 16, 4), (5000, 100, 4, 16), (300, 256, 256, 2), (70000, None, 16, 33), (1 << 18, 256, 16, 16),
                            (255, 256, 16, 4), (513, 256, 3, 5)]:
        x = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-3, 3, (1,), generator=g)))
        q, sf = quantization.uniformQuantization(x, s, bucket_size=bucket)
        qr, sfr = refq.uniformQuantization(x, s, bucket_size=bucket)
        assert torch.equal(q, qr) and torch.equal(sf.alpha, sfr.alpha) and torch.equal(sf.beta, sfr.beta), (n, bucket, s)
        assert torch.equal(sf.idx_min_rows, sfr.idx_min_rows) and torch.equal(sf.idx_max_rows, sfr.idx_max_rows), (n, bucket, s)
        assert sf.expected_tensor_size == sfr.expected_tensor_size and sf.original_tensor_length == sfr.original_tensor_length
        pts = torch.sort(torch.rand(k, generator=g))[0]
        qn, idx, _ = quantization.nonUniformQuantization(x, pts, bucket_size=bucket)
        qnr, idxr, _ = refq.nonUniformQuantization(x, pts, bucket_size=bucket)
        assert torch.equal(qn, qnr) and torch.equal(idx, idxr) and idx.dtype == idxr.dtype, (n, bucket, k)
        q8, idx8, _ = quantization.nonUniformQuantization(x, pts, bucket_size=bucket, index_dtype=torch.uint8)
        assert torch.equal(idx8.long(), idxr) and torch.equal(q8, qnr)
        a, b = quantization.ScalingFunction('linear', False, False, bucket), refq.ScalingFunction('linear', False, False, bucket)
        u, ur = a.scale_down(x), b.scale_down(x)
        assert torch.equal(u, ur) and torch.equal(a.inv_scale_down(u), b.inv_scale_down(ur)), (n, bucket)
        fa = quantization.nonUniformQuantization_variable(bucket_size=bucket, pre_process_tensors=True, tensor=x)
        fb = refq.nonUniformQuantization_variable(bucket_size=bucket, pre_process_tensors=True, tensor=x)
        assert torch.equal(fa.forward(None, pts), fb.forward(None, pts)), (n, bucket, k)
        assert torch.equal(fa.savedForBackward['indices'], fb.savedForBackward['indices'])
        gr = torch.randn(n, generator=g)
        _, gp = fa.backward(gr)
        _, gpr = fb.backward(gr)
        alpha_e = sfr.alpha.reshape(-1).repeat_interleave(bucket)[:n] if (bucket and n >= bucket) else sfr.alpha.reshape(-1)[0].expand(n)
        scale = float((gr.double() * alpha_e.double()).abs().sum())
        assert float((gp.double() - gpr.double()).abs().max()) <= 1e-6 * scale, (n, bucket, k)     # the reference sums in fp32
        init = qhf.initialize_quantization_points(x, quantization.ScalingFunction('linear', False, False, bucket), k)
        import importlib
        refh = importlib.import_module(refq.__name__ + '.help_functions')
        assert torch.equal(init, refh.initialize_quantization_points(x, refq.ScalingFunction('linear', False, False, bucket), k))


def test_stochastic_rounding_draws_the_device_generators_numbers():
    """csrc/host/qd_host.cpp carries the Philox4x32-7 of csrc/qd_common.h: element e rounds up iff uniform(seed, e) <= frac.
    Checked against a Python statement of the same generator (the device path is checked against the same statement in
    tests/test_hip_parity.py::test_stochastic_rounding_bit_exact_on_every_kernel_path through the C oracle's levels)."""
    def philox4(seed, block):
        M0, M1, mask = 0xD2511F53, 0xCD9E8D57, 0xFFFFFFFF
        c = [block & mask, (block >> 32) & mask, 0x51ed270b, 0x2545f491]
        k0, k1 = seed & mask, (seed >> 32) & mask
        for _ in range(7):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> 32) ^ c[3] ^ k1) & mask, p0 & mask]
            k0, k1 = (k0 + 0x9E3779B9) & mask, (k1 + 0xBB67AE85) & mask
        return [np.float32(v >> 8) * np.float32(1.0 / 16777216.0) for v in c]
    n, s = 1000, 16
    x = torch.linspace(0, 1, n)
    seed = 0x1234567890ABCDEF
    q = torch.empty(n)
    ab = torch.empty(2, 1)
    _lib.check(_lib.host().qd_uniform_f32(x.data_ptr(), q.data_ptr(), n, 0, s, ab[0].data_ptr(), ab[1].data_ptr(), None, None, 0, 0.0,
                                         1, seed, None, 0, None))
    xs, sm1 = x.numpy(), np.float32(s - 1)
    a, b = np.float32(ab[0, 0]), np.float32(ab[1, 0])
    for e in range(n):
        u = np.float32(np.float32(xs[e] - b) / a)
        t = np.float32(u * sm1)
        lo = np.floor(t)
        up = philox4(seed, e >> 2)[e & 3] <= np.float32(t - lo)
        w = np.float32(np.float32(lo / sm1) + (np.float32(np.float32(1.0) / sm1) if up else np.float32(0.0)))
        want = np.float32(np.float32(np.float32(w * a) + b) + np.float32(0.0))
        assert np.float32(q[e]) == want, e


@pytest.mark.parametrize('name', ['test_uniform_matches_oracle', 'test_nonuniform_matches_oracle', 'test_ste_backward_matches_oracle',
                                  'test_uniform_subtract_mean_matches_oracle'])
def test_property_suite_on_cpu_tensors(name, monkeypatch):
    """The device-independent hypothesis properties of tests/test_hip_property.py (random sizes, 35 bucket sizes, level / point
    counts, nine value distributions incl. denormals, huge ranges and all-ties, clamp, mean) with every tensor on the CPU."""
    import test_hip_property as PP
    monkeypatch.setattr(PP, 'DEV', 'cpu')
    getattr(PP, name)()
