"""One API, two libraries: the same call on a tensor that lives on the HIP device (libqd_hip.so) and on its CPU copy
(libqd_host.so) gives the same bits -- values, alpha / beta, arg indices, point indices, and (through the C ABI with one seed) the
stochastic-rounding branch; the two fp32 reductions agree to the tolerance both are held to against float64."""
import numpy as np
import pytest
import torch

import quantization
from quantized_distillation_amd import _lib, ste

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('n,bucket', [(100003, 256), (5000, 100), (300, 256), (70001, None), (1 << 20, 256), (4099, 33)])
def test_device_and_host_library_agree(n, bucket):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 0.3
    xd = x.to(DEV)
    for s in (16, 4, 256):
        qc, sc = quantization.uniformQuantization(x, s, bucket_size=bucket)
        qd, sd = quantization.uniformQuantization(xd, s, bucket_size=bucket)
        assert torch.equal(qd.cpu(), qc) and torch.equal(sd.alpha.cpu(), sc.alpha) and torch.equal(sd.beta.cpu(), sc.beta), (s,)
        assert torch.equal(sd.idx_min_rows.cpu(), sc.idx_min_rows) and torch.equal(sd.idx_max_rows.cpu(), sc.idx_max_rows)
    qc, sc = quantization.uniformQuantization(x, 16, bucket_size=bucket, max_element=0.5, subtract_mean=True)
    qd, sd = quantization.uniformQuantization(xd, 16, bucket_size=bucket, max_element=0.5, subtract_mean=True)
    assert float(sd.mean_tensor) == float(sc.mean_tensor) and torch.equal(qd.cpu(), qc)          # both: float64 sum, one rounding
    pts = torch.sort(torch.rand(11, generator=g))[0]
    qc, ic, _ = quantization.nonUniformQuantization(x, pts, bucket_size=bucket)
    qd, idv, _ = quantization.nonUniformQuantization(xd, pts, bucket_size=bucket)
    assert torch.equal(qd.cpu(), qc) and torch.equal(idv.cpu(), ic)
    a, b = quantization.ScalingFunction('linear', False, False, bucket), quantization.ScalingFunction('linear', False, False, bucket)
    uc, ud = a.scale_down(x), b.scale_down(xd)
    assert torch.equal(ud.cpu(), uc) and torch.equal(b.inv_scale_down(ud).cpu(), a.inv_scale_down(uc))
    fc = quantization.nonUniformQuantization_variable(bucket_size=bucket, pre_process_tensors=True, tensor=x)
    fd = quantization.nonUniformQuantization_variable(bucket_size=bucket, pre_process_tensors=True, tensor=xd)
    assert torch.equal(fd.forward(None, pts.to(DEV)).cpu(), fc.forward(None, pts))
    assert torch.equal(fd.savedForBackward['indices'].cpu(), fc.savedForBackward['indices'])
    gr = torch.randn(n, generator=g)
    gpc, gpd = fc.backward(gr)[1], fd.backward(gr.to(DEV))[1].cpu()
    alpha_e = sc.alpha.reshape(-1)
    scale = float(gr.abs().double().sum()) * float(a.alpha.abs().max())
    assert float((gpc.double() - gpd.double()).abs().max()) <= 1e-6 * scale
    if bucket is not None:
        uc_, ud_ = quantization.uniformQuantization_variable(16, bucket_size=bucket), quantization.uniformQuantization_variable(16, bucket_size=bucket)
        uc_.forward(x), ud_.forward(xd)
        oc, od = uc_.backward(gr), ud_.backward(gr.to(DEV)).cpu()
        assert float((oc.double() - od.double()).abs().max()) <= 1e-6 * float(gr.abs().double().sum())
        assert int((oc != od).sum()) <= 2 * (-(-n // bucket))          # only the two touched elements of a bucket can differ (summation order)
    w = torch.randn(n, generator=g)
    assert torch.equal(ste.clamp_(w.clone().to(DEV), 1.0).cpu(), ste.clamp_(w.clone(), 1.0))
    assert torch.equal(ste.truncated_ste_(gr.clone().to(DEV), w.to(DEV), 1.0).cpu(), ste.truncated_ste_(gr.clone(), w, 1.0))
    # stochastic rounding: one seed through the C ABI of both libraries -> the same draws, the same bits
    seed = 0xC0FFEE123456789
    nb = 1 if (bucket is None or n < bucket) else -(-n // bucket)
    qh, abh = torch.empty(n), torch.empty(2, nb)
    qg, abg = torch.empty(n, device=DEV), torch.empty(2, nb, device=DEV)
    ws = _lib.workspace(torch.device(DEV))
    _lib.check(_lib.host().qd_uniform_f32(x.data_ptr(), qh.data_ptr(), n, bucket or 0, 16, abh[0].data_ptr(), abh[1].data_ptr(), None, None, 0, 0.0,
                                         1, seed, None, 0, None))
    _lib.check(_lib.load().qd_uniform_f32(xd.data_ptr(), qg.data_ptr(), n, bucket or 0, 16, abg[0].data_ptr(), abg[1].data_ptr(), None, None, 0, 0.0,
                                         1, seed, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    assert torch.equal(qg.cpu(), qh)
    assert not torch.equal(qh, quantization.uniformQuantization(x, 16, bucket_size=bucket)[0])      # (and it IS the stochastic branch)


@pytest.mark.parametrize('tie_mode', [0, 1])        # QD_STE_TIE_REFERENCE, QD_STE_TIE_TRUE_ARG
def test_ste_backward_on_buckets_that_hold_a_nan_or_an_infinity(tie_mode):
    """torch's min / max propagate a NaN and report it at its FIRST position, for the minimum and the maximum alike.  A bucket
    of x that holds a NaN (or an infinity: alpha = inf) therefore quantizes to NaN as a whole, its sum S is NaN, and the
    reference's 'complicated' backward adds and subtracts S at one element: out = NaN there, g everywhere else
    (quant_functions.py:350,383-400).  Both libraries do exactly that; in the reference's tie mode the result is compared with
    the staged reference itself (the two shape fixes of SURVEY 8c applied), NaN pattern and values."""
    from oracle import ref_stage
    n, bucket = 64 * 256 + 77, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g)
    nan_buckets, inf_buckets = list(range(0, 64, 3)), list(range(1, 64, 6))
    first_nan = {}
    for b in nan_buckets:
        pos = sorted(int(v) for v in torch.randint(0, bucket, (2,), generator=g))
        x[b * bucket + pos[0]] = x[b * bucket + pos[1]] = float('nan')
        first_nan[b] = b * bucket + pos[0]
    for b in inf_buckets:
        x[b * bucket + 7] = float('inf') if b % 2 else float('-inf')
    x[64 * bucket + 5] = float('nan')                                            # the short last bucket too
    first_nan[64] = 64 * bucket + 5
    oh, od = torch.empty(n), torch.empty(n, device=DEV)
    xd, gd = x.to(DEV), gr.to(DEV)
    _lib.check(_lib.host().qd_ste_bucket_backward_f32(x.data_ptr(), gr.data_ptr(), oh.data_ptr(), n, bucket, 16, tie_mode, None))
    _lib.check(_lib.load().qd_ste_bucket_backward_f32(xd.data_ptr(), gd.data_ptr(), od.data_ptr(), n, bucket, 16, tie_mode, _lib.stream_ptr()))
    od = od.cpu()
    assert torch.equal(torch.isnan(oh), torch.isnan(od))
    for b, j in first_nan.items():
        sl = slice(b * bucket, min((b + 1) * bucket, n))
        j = b * bucket if tie_mode == 0 else j             # reference mode: the first NaN of the QUANTIZED bucket, which is NaN as a whole
        want = gr[sl].clone()
        want[j - b * bucket] = float('nan')
        for o in (oh, od):
            assert torch.equal(torch.isnan(o[sl]), torch.isnan(want)) and torch.equal(torch.nan_to_num(o[sl]), torch.nan_to_num(want)), (b, j)
    for b in inf_buckets:
        if b in first_nan:
            continue
        sl = slice(b * bucket, (b + 1) * bucket)
        assert int(torch.isnan(od[sl]).sum()) == (1 if tie_mode == 0 else 2), b          # one element (+S - S), or the arg-max and the arg-min
    fin = ~torch.isnan(oh)
    assert float((oh[fin].double() - od[fin].double()).abs().max()) <= 1e-6 * float(gr.abs().double().sum())
    if tie_mode == 0:
        patched = ref_stage.load_patched()
        assert patched is not None, 'oracle/_ref/patched is not staged (run __graft_entry__.build() where /root/reference exists)'
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fn = patched.uniformQuantization_variable(16, bucket_size=bucket)
            fn.forward(x.clone())
            want = fn.backward(gr.clone()).view(-1)
        assert torch.equal(torch.isnan(want), torch.isnan(od))
        assert float((want[fin].double() - od[fin].double()).abs().max()) <= 1e-6 * float(gr.abs().double().sum())
        # the same through the public API on the device tensor
        fd = quantization.uniformQuantization_variable(16, bucket_size=bucket)
        fd.forward(xd)
        assert torch.equal(torch.isnan(fd.backward(gd).cpu().view(-1)), torch.isnan(want))
