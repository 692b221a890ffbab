"""THE drop-in claim, end to end: the reference's own `train_model(..., quantizeWeights=True)`
(cnn_models/conv_forward_model.py:165-393, staged as bytecode by oracle/ref_stage.py with the one torch >= 0.5 fix
`loss.data[0]` -> `loss.item()`), run UNCHANGED twice on the same MI355X with the same initial weights and batches:

    A   `import quantization` resolves to this repository's package   (HIP kernels behind the C ABI)
    B   `import quantization` resolves to the reference's own package, its functions fed with host copies of the
        weights (on torch 2.x the reference's quantizer raises on device tensors: it mixes CPU and device tensors)

Everything but the quantizer is the same code on the same device, and the quantizer is bit-exact, so the two
trainings must agree: every parameter after training, the loss history, and the quantized weights the loop returns.
Skipped when nothing is staged (run __graft_entry__.build() where the reference checkout exists)."""
import contextlib
import copy
import io

import pytest
import torch

import quantization as product_quantization
from oracle import ref_stage

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ref_stage.loop_is_staged() and ref_stage.is_staged()),
                                 reason='reference loop / quantizer not staged under oracle/_ref')]
DEV = torch.device('cuda:0')


def _reference_on_host(refq):
    """A module that looks like the reference's `quantization` package to the loop and runs the reference's own
    functions on host copies of the tensors, handing the results back on the device they came from."""
    import types
    m = types.ModuleType('quantization')
    m.USE_CUDA = refq.USE_CUDA
    m.help_functions, m.quant_functions = refq.help_functions, refq.quant_functions
    m.ScalingFunction = refq.ScalingFunction
    m.nonUniformQuantization = refq.nonUniformQuantization
    m.uniformQuantization_variable = refq.uniformQuantization_variable
    m.nonUniformQuantization_variable = refq.nonUniformQuantization_variable

    def uniformQuantization(tensor, *args, **kwargs):
        q, sf = refq.uniformQuantization(tensor.cpu(), *args, **kwargs)
        return q.to(tensor.device), sf
    m.uniformQuantization = uniformQuantization
    return m


def _batches(n, batch):
    g = torch.Generator().manual_seed(4242)
    return [(torch.randn(batch, 3, 32, 32, generator=g), torch.randint(0, 10, (batch,), generator=g)) for _ in range(n)]


def _train(loop, state, batches, **kw):
    torch.manual_seed(0)
    model = loop.ConvolForwardNet(**loop.smallerModelSpec, useBatchNorm=True, useAffineTransformInBatchNorm=True)
    model.load_state_dict(state)
    model = model.to(DEV)
    torch.manual_seed(123)                                   # dropout masks, if any
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        model, info = loop.train_model(model, batches, batches[:1], epochs_to_train=2, print_every=1,
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
def test_reference_train_model_runs_unchanged_on_our_quantizer(kw):
    refq = ref_stage.load()
    loop_ours = ref_stage.load_loop(product_quantization)
    host_ref = _reference_on_host(refq)
    loop_ref = ref_stage.load_loop(host_ref)
    assert loop_ours.quantization is product_quantization and loop_ref.quantization is host_ref
    assert loop_ours.USE_CUDA and loop_ours.cnn_hf.USE_CUDA
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    torch.manual_seed(7)
    init = loop_ref.ConvolForwardNet(**loop_ref.smallerModelSpec, useBatchNorm=True, useAffineTransformInBatchNorm=True)
    state = copy.deepcopy(init.state_dict())
    batches = _batches(4, 16)
    m_a, info_a = _train(loop_ours, state, batches, **kw)
    m_b, info_b = _train(loop_ref, state, batches, **kw)
    assert info_a['numEpochsTrained'] == info_b['numEpochsTrained'] == 2
    assert info_a['lossSaved'] == info_b['lossSaved'], (info_a['lossSaved'], info_b['lossSaved'])
    worst = 0.0
    for (na, pa), (nb, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert na == nb and pa.is_cuda
        worst = max(worst, float((pa - pb).abs().max()))
        assert torch.equal(pa, pb), (na, worst)
    # the loop returns the model with QUANTIZED weights (conv_forward_model.py:384-385): they sit on a grid of 2^bits levels
    w = [p for n, p in m_a.named_parameters() if p.numel() == 800000][0].detach().view(-1)
    if kw.get('bucket_size') == 256 and kw.get('quantize_first_and_last_layer', True):
        rows = w[:256 * 100].view(100, 256)
        assert all(len(torch.unique(r)) <= 2 ** kw['numBits'] for r in rows)
