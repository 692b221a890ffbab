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
