"""BASELINE configs[0] as a drop-in test: "CIFAR10 ConvolForwardNet student, 4-bit uniform quant, CPU reference path".

The reference's own `train_model(..., quantizeWeights=True)` (cnn_models/conv_forward_model.py:165-393, staged as bytecode by
oracle/ref_stage.py with the one torch >= 0.5 fix) and its own `optimize_quantization_points` (:395-592), run UNCHANGED twice
on the CPU with the same initial weights and batches:

    A   `import quantization` resolves to this repository's package -> CPU tensors -> libqd_host.so
    B   `import quantization` resolves to the reference's own package (its CPU path is what the reference ships)

Everything but the quantizer is the same code, and the quantizer is bit-exact, so the two trainings must agree bit for bit:
every parameter after training, the loss history, the quantization points.  (The GPU counterpart, with side B fed host
copies, is tests/test_hip_dropin_reference_loop.py.)  Runs where the loop stays on the CPU (`USE_CUDA` false: this container)."""
import contextlib
import copy
import io

import pytest
import torch

import quantization as product_quantization
from oracle import ref_stage

pytestmark = [pytest.mark.skipif(not (ref_stage.loop_is_staged() and ref_stage.is_staged()),
                                 reason='reference loop / quantizer not staged under oracle/_ref'),
              pytest.mark.skipif(torch.cuda.is_available(), reason="the reference's loop moves everything to the GPU when there is one")]


def _batches(n, batch):
    g = torch.Generator().manual_seed(4242)
    return [(torch.randn(batch, 3, 32, 32, generator=g), torch.randint(0, 10, (batch,), generator=g)) for _ in range(n)]


def _train(loop, state, batches, epochs=2, **kw):
    torch.manual_seed(0)
    model = loop.ConvolForwardNet(**loop.smallerModelSpec, useBatchNorm=True, useAffineTransformInBatchNorm=True)
    model.load_state_dict(state)
    torch.manual_seed(123)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        model, info = loop.train_model(model, batches, batches[:1], epochs_to_train=epochs, print_every=1,
                                       quantizeWeights=True, use_distillation_loss=False, **kw)
    assert info['errorFlag'] is False, out.getvalue()[-2000:]
    return model, info


@pytest.mark.parametrize('kw', [
    dict(numBits=4, bucket_size=256),
    dict(numBits=4, bucket_size=256, quantize_first_and_last_layer=False),
    dict(numBits=2, bucket_size=None),
    dict(numBits=4, bucket_size=256, backprop_quantization_style='truncated'),
    dict(numBits=8, bucket_size=100, estimate_quant_grad_every=2),
], ids=['4bit-b256', '4bit-b256-skip-first-last', '2bit-nobucket', '4bit-truncated', '8bit-b100-every2'])
def test_reference_train_model_runs_unchanged_on_cpu_tensors(kw, monkeypatch):
    from quantized_distillation_amd import _lib

    def boom(*a, **k):
        raise AssertionError('a CPU training loop reached the HIP library')
    monkeypatch.setattr(_lib, 'load', boom)
    monkeypatch.setattr(_lib, 'glue', boom)
    refq = ref_stage.load()
    loop_ours = ref_stage.load_loop(product_quantization)
    loop_ref = ref_stage.load_loop(refq)
    assert loop_ours.quantization is product_quantization and loop_ref.quantization is refq and not loop_ours.USE_CUDA
    torch.manual_seed(7)
    init = loop_ref.ConvolForwardNet(**loop_ref.smallerModelSpec, useBatchNorm=True, useAffineTransformInBatchNorm=True)
    state = copy.deepcopy(init.state_dict())
    batches = _batches(3, 8)
    m_a, info_a = _train(loop_ours, state, batches, **kw)
    m_b, info_b = _train(loop_ref, state, batches, **kw)
    assert info_a['numEpochsTrained'] == info_b['numEpochsTrained'] == 2
    assert info_a['lossSaved'] == info_b['lossSaved'], (info_a['lossSaved'], info_b['lossSaved'])
    for (na, pa), (nb, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert na == nb and not pa.is_cuda
        assert torch.equal(pa, pb), (na, float((pa - pb).abs().max()))


@pytest.mark.parametrize('kw', [dict(numPointsPerTensor=4, bucket_size=256),
                                dict(numPointsPerTensor=8, bucket_size=256, assignBitsAutomatically=True)], ids=['k4-b256', 'k8-auto-bits'])
def test_reference_optimize_quantization_points_runs_unchanged_on_cpu_tensors(kw):
    """The differentiable-quantization loop (ref: conv_forward_model.py:395-592) on CPU tensors: percentile initialisation,
    nonUniformQuantization_variable forward / backward per tensor per step, SGD on the points -- this package (libqd_host.so)
    against the reference's own package.  Every forward is bit-identical; the point gradients differ in the order of one fp32
    sum per point (the reference sums in fp32 with torch's blocking, the host library in float64), and the loop amplifies that
    step by step (it is not contractive, see the GPU counterpart), so the trained points agree to 1e-4 relative."""
    refq = ref_stage.load()
    loop_ours = ref_stage.load_loop(product_quantization)
    # side B: the reference's package as it is, except that its assign_bits_automatically gets Python floats (round(tensor)
    # raises on torch 2.x -- an incompatibility of the reference with this torch, not a difference between the quantizers)
    import types
    side_b = types.ModuleType('quantization')
    side_b.__dict__.update({k: v for k, v in refq.__dict__.items() if not k.startswith('__')})
    hf = types.ModuleType('quantization.help_functions')
    hf.__dict__.update({k: v for k, v in refq.help_functions.__dict__.items() if not k.startswith('__')})
    hf.assign_bits_automatically = lambda importance, *a, **k: refq.help_functions.assign_bits_automatically(
        [float(x) for x in importance], *a, **k)
    side_b.help_functions = hf
    loop_ref = ref_stage.load_loop(side_b)
    batches = _batches(3, 8)
    results = []
    for loop in (loop_ours, loop_ref):
        torch.manual_seed(7)
        model = loop.ConvolForwardNet(**loop.smallerModelSpec, useBatchNorm=True, useAffineTransformInBatchNorm=True)
        g = torch.Generator().manual_seed(99)                # (constant tensors make the loop chaotic: as in the GPU test)
        with torch.no_grad():
            for prm in model.parameters():
                if prm.dim() == 1:
                    prm.add_(0.05 * torch.randn(prm.shape, generator=g))
        torch.manual_seed(11)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            state, points, info = loop.optimize_quantization_points(
                model, batches, batches[:1], initial_learning_rate=1e-5, epochs_to_train=1, print_every=1,
                use_distillation_loss=False, **kw)
        results.append((state, [p.detach().cpu() for p in points], info))
    (sa, pa, ia), (sb, pb, ib) = results
    assert [p.numel() for p in pa] == [p.numel() for p in pb], 'same number of points per tensor'
    assert ia['numEpochsTrained'] == ib['numEpochsTrained'] == 1
    for i, (x, y) in enumerate(zip(pa, pb)):
        assert torch.all(x[1:] >= x[:-1]), 'points stay sorted'
        assert torch.allclose(x, y, rtol=1e-4, atol=2e-6), (i, x, y, float((x - y).abs().max()))
    total = flips = 0
    for k in sa:
        a, b = sa[k].detach().float().view(-1), sb[k].detach().float().view(-1)
        total += a.numel()
        flips += int((~torch.isclose(a, b, rtol=1e-4, atol=1e-6)).sum())
    assert flips <= max(3, total * 2e-5), (flips, total)  # the returned quantized weights: equal up to a vanishing fraction of assignment flips
    assert abs(ia['lossSaved'][-1] - ib['lossSaved'][-1]) <= 1e-4 * max(1.0, abs(ib['lossSaved'][-1]))
