"""The SECOND caller `north_star` names, replayed statement by statement: the quantizer call sites of the reference's
seq2seq training loop, translation_models/model.py

    :184,243      step_since_last_grad_quant_estimation starts at 0 -> the FIRST batch is NOT quantized
    :242-258      state_dict() saved, then per parameter (first / last skipped on request):
                    'truncated'   p.data.clamp_(-1, 1); p.data = quantization.uniformQuantization(p.data, s, ...)[0]
                    'none'        p.data = quantization.uniformQuantization(p.data, s, ...)[0]
                    'complicated' p.data = quantizeFunctions[idx].forward(p.data)       (uniformQuantization_variable)
    :260          forward / backward on the quantized weights
    :262-279      load_state_dict(saved); 'truncated': p.grad.data[p.data.abs() > 1] = 0;
                  'complicated': p.grad.data = quantizeFunctions[idx].backward(p.grad.data)
    :282          optimizer step on the full-precision weights
    :301-312      after training every parameter is quantized once more; 'complicated' deletes and resets
                  saved_for_backward (:309-310)

onmt itself needs torchtext 0.1.1 and cannot run here, so the loop around these statements is a stand-in (synthetic
token batches, the harness' 2-layer LSTM encoder-decoder with the reference's parameter shape list, plain SGD); the
statements themselves are executed twice from the same initial weights and batches --

    A   `quantization` = this repository's package (HIP kernels behind the C ABI), tensors stay on the MI355X
    B   `quantization` = the reference's own package from the staged bytecode (oracle/ref_stage.py), fed host copies

-- and must leave bit-identical parameters after 3 SGD steps for 'none' / 'truncated' (the quantizer is bit-exact and
everything else is the same code on the same device) and K7-tolerance agreement for 'complicated', where side B runs the
reference's backward with the two shape fixes of SURVEY.md 8c (as shipped it raises for more than one bucket).
"""
import copy

import numpy as np
import pytest
import torch

import quantization as product_quantization
from harness import models
from harness.distill import synthetic_token_batch
from oracle import ref_stage

import errlog

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ref_stage.is_staged() and ref_stage.patched_is_staged()),
                                 reason='reference quantizer not staged under oracle/_ref (run __graft_entry__.build())')]
DEV = torch.device('cuda:0')


class _HostVariable:
    """The reference's uniformQuantization_variable on host copies: same life cycle (forward saves the input, backward
    consumes and resets saved_for_backward), tensors handed back on the device they came from.  Flattened: the shipped
    backward handles 1-D tensors only (its .view logic mixes the flat and the original shape), and buckets are cut
    from the flattened tensor anyway."""

    def __init__(self, ref_fn):
        self.fn = ref_fn

    @property
    def saved_for_backward(self):
        return self.fn.saved_for_backward

    @saved_for_backward.setter
    def saved_for_backward(self, v):
        self.fn.saved_for_backward = v

    @saved_for_backward.deleter
    def saved_for_backward(self):
        del self.fn.saved_for_backward

    def forward(self, t):
        return self.fn.forward(t.detach().cpu().reshape(-1)).reshape(t.shape).to(t.device)

    def backward(self, g):
        return self.fn.backward(g.detach().cpu().reshape(-1)).reshape(g.shape).to(g.device)


def _reference_on_host(refq):
    import types
    m = types.ModuleType('quantization')

    def uniformQuantization(tensor, *args, **kwargs):
        q, sf = refq.uniformQuantization(tensor.cpu(), *args, **kwargs)
        return q.to(tensor.device), sf
    m.uniformQuantization = uniformQuantization
    m.uniformQuantization_variable = lambda *a, **k: _HostVariable(refq.uniformQuantization_variable(*a, **k))
    return m


def _loss(model, batch):
    src, tgt = batch
    logits = model(src, tgt[:-1])
    return torch.nn.functional.cross_entropy(logits, tgt[1:].reshape(-1), ignore_index=1, reduction='sum') / src.size(1)


def _replay(quantization, state, batches, style, numBits=4, bucket_size=256, quantize_first_and_last_layer=True,
            num_estimate_quant_grad=1, lr=0.5, model_kw=None, grads_out=None):
    """The call-site sequence of translation_models/model.py:184-312 around a stand-in forward/backward/step."""
    torch.manual_seed(0)
    model = models.Seq2SeqLSTM(**(model_kw or {}))
    model.load_state_dict(state)
    model = model.to(DEV)
    optim = torch.optim.SGD(model.parameters(), lr=lr)
    step_since_last_grad_quant_estimation = 0                                   # :184
    num_param_model = sum(1 for _ in model.parameters())
    s = 2 ** numBits                                                             # :194
    if style in ('none', 'truncated'):                                           # :199-205
        def quantizeFunctions(x):
            return quantization.uniformQuantization(x, s, type_of_scaling='linear', stochastic_rounding=False,
                                                    max_element=False, subtract_mean=False, modify_in_place=False,
                                                    bucket_size=bucket_size)[0]
    else:                                                                        # :207-213
        quantizeFunctions = [quantization.uniformQuantization_variable(s, type_of_scaling='linear',
                                                                       stochastic_rounding=False, max_element=False,
                                                                       subtract_mean=False, modify_in_place=False,
                                                                       bucket_size=bucket_size)
                             for _ in model.parameters()]
    quantized_steps = 0
    for batch in batches:
        model.zero_grad()
        quantize_now = step_since_last_grad_quant_estimation >= num_estimate_quant_grad
        if quantize_now:                                                         # :243-258
            model_state_dict = model.state_dict()                                # no copy: p.data is REBOUND below, :245
            for idx, p in enumerate(model.parameters()):
                if quantize_first_and_last_layer is False and (idx == 0 or idx == num_param_model - 1):
                    continue
                if style == 'truncated':
                    p.data.clamp_(-1, 1)
                if style in ('none', 'truncated'):
                    p.data = quantizeFunctions(p.data)
                else:
                    p.data = quantizeFunctions[idx].forward(p.data)
            quantized_steps += 1
        _loss(model, batch).backward()                                           # :260
        if quantize_now:                                                         # :262-279
            model.load_state_dict(model_state_dict)
            del model_state_dict
            if style in ('truncated', 'complicated'):
                for idx, p in enumerate(model.parameters()):
                    if quantize_first_and_last_layer is False and (idx == 0 or idx == num_param_model - 1):
                        continue
                    if style == 'truncated':
                        p.grad.data[p.data.abs() > 1] = 0
                    else:
                        p.grad.data = quantizeFunctions[idx].backward(p.grad.data)
                        assert quantizeFunctions[idx].saved_for_backward is None
        if grads_out is not None:
            grads_out.append([p.grad.detach().clone() for p in model.parameters()])
        optim.step()                                                             # :282
        if step_since_last_grad_quant_estimation >= num_estimate_quant_grad:
            step_since_last_grad_quant_estimation = 0
        step_since_last_grad_quant_estimation += 1
    trained = [p.detach().clone() for p in model.parameters()]
    for idx, p in enumerate(model.parameters()):                                 # :301-312
        if style == 'truncated':
            p.data.clamp_(-1, 1)
        if style in ('none', 'truncated'):
            p.data = quantizeFunctions(p.data)
        else:
            p.data = quantizeFunctions[idx].forward(p.data)
            del quantizeFunctions[idx].saved_for_backward
            quantizeFunctions[idx].saved_for_backward = None
    return trained, [p.detach().clone() for p in model.parameters()], quantized_steps


def _setup(model_kw, nbatch, batch, seq):
    torch.manual_seed(77)
    m = models.Seq2SeqLSTM(**model_kw)
    with torch.no_grad():                                    # some weights beyond [-1, 1]: the 'truncated' clamp and mask bite
        for p in m.parameters():
            if p.dim() > 1:
                p.mul_(1.0 + 3.0 * (torch.rand(p.shape, generator=torch.Generator().manual_seed(p.numel() % 9973)) > 0.995).float())
    state = copy.deepcopy(m.state_dict())
    v_src, v_tgt = model_kw.get('v_src', 18000), model_kw.get('v_tgt', 10000)
    batches = [synthetic_token_batch(batch, DEV, seed=10 + i, v_src=v_src, v_tgt=v_tgt, max_len=seq) for i in range(nbatch)]
    return state, batches


@pytest.mark.parametrize('style,kw', [
    ('none', dict(numBits=4, bucket_size=256)),
    ('truncated', dict(numBits=4, bucket_size=256)),
    ('none', dict(numBits=4, bucket_size=256, quantize_first_and_last_layer=False)),
    ('none', dict(numBits=2, bucket_size=None)),
], ids=['none-4bit-b256', 'truncated-4bit-b256', 'none-skip-first-last', 'none-2bit-nobucket'])
def test_seq2seq_call_sites_bit_identical_to_the_reference_quantizer(style, kw):
    """The reference's full LSTM parameter shape list ((18000,500), (10000,500), (2000,1000), ... 22 tensors, 28.8 M):
    4 batches = 1 un-quantized + 3 quantized SGD steps; every parameter after training and every weight the loop returns
    (the final quantization, :301-312) bit-identical between our package and the reference's own."""
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.backends.cudnn.deterministic = True
    try:
        model_kw = {}
        state, batches = _setup(model_kw, 4, 16, 24)
        refq = ref_stage.load()
        a_tr, a_q, a_steps = _replay(product_quantization, state, batches, style, model_kw=model_kw, **kw)
        b_tr, b_q, b_steps = _replay(_reference_on_host(refq), state, batches, style, model_kw=model_kw, **kw)
    finally:
        torch.use_deterministic_algorithms(False)
    assert a_steps == b_steps == 3, 'the first batch is not quantized (ref: :184,243)'
    assert len(a_tr) == 22
    for i, (x, y) in enumerate(zip(a_tr, b_tr)):
        assert torch.equal(x, y), (style, kw, i, tuple(x.shape), float((x - y).abs().max()))
    initial = list(state.values())
    assert sum(int(not torch.equal(x.cpu(), w)) for x, w in zip(a_tr, initial)) >= 20, 'training moved the weights'
    for i, (x, y) in enumerate(zip(a_q, b_q)):
        assert torch.equal(x, y), (style, kw, 'final quantization', i, tuple(x.shape))
    # the returned weights are quantized: at most 2^numBits distinct values per bucket of the big embedding
    big = a_q[0].reshape(-1)
    if kw.get('bucket_size'):
        assert all(len(torch.unique(big[j * 256:(j + 1) * 256])) <= 2 ** kw['numBits'] for j in range(0, 2000, 97))
    else:
        assert len(torch.unique(big)) <= 2 ** kw['numBits']


def test_seq2seq_call_sites_complicated_style_matches_the_patched_reference():
    """'complicated' (uniformQuantization_variable.forward / .backward per parameter, the saved_for_backward delete/reset
    of :309-310): side B is the reference's own backward with the two 8c shape fixes on host copies.  The forward is
    bit-exact; each backward differs from the reference's only in the order of one fp32 sum per bucket, so the first
    quantized step's gradients agree to the K7 tolerance (1e-6 of sum|terms| per bucket) and the parameters after three
    steps to fp32 round-off of that size.  A smaller model (the reference's backward builds N x N sparse matrices on the
    host): same 22-tensor structure, vocabulary 1200 / 900, hidden size 64."""
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.backends.cudnn.deterministic = True
    kw = dict(numBits=4, bucket_size=256)
    model_kw = dict(v_src=1200, v_tgt=900, emb=64, hidden=64)
    try:
        state, batches = _setup(model_kw, 4, 8, 22)
        refq = ref_stage.load_patched()
        ga, gb = [], []
        a_tr, a_q, _ = _replay(product_quantization, state, batches, 'complicated', model_kw=model_kw, grads_out=ga, lr=0.05, **kw)
        b_tr, b_q, _ = _replay(_reference_on_host(refq), state, batches, 'complicated', model_kw=model_kw, grads_out=gb, lr=0.05, **kw)
    finally:
        torch.use_deterministic_algorithms(False)
    # batch 0 is not quantized: identical code, identical gradients
    for x, y in zip(ga[0], gb[0]):
        assert torch.equal(x, y)
    # batch 1: same quantized weights (bit-exact forward) -> same incoming gradients; the STE backward's bucket sums differ
    # in summation order only.  Check every tensor against the float64 oracle AND against the reference's fp32 output.
    names = list(state.keys())
    for i, (x, y) in enumerate(zip(ga[1], gb[1])):
        ratio = float((x - y).abs().max() / (y.abs().max() + 1e-30))
        assert torch.allclose(x, y, rtol=0, atol=2e-6 * float(y.abs().max()) + 1e-12), (i, names[i], ratio)
    for i, (x, y) in enumerate(zip(a_tr, b_tr)):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6), (i, names[i], float((x - y).abs().max()))
    for i, (x, y) in enumerate(zip(a_q, b_q)):
        # the final quantization of nearly identical weights: the same levels except where a weight sits on a rounding boundary
        close = torch.isclose(x, y, rtol=1e-4, atol=1e-5)
        assert int((~close).sum()) <= max(2, x.numel() // 2000), (i, names[i], int((~close).sum()))


def test_complicated_backward_object_life_cycle_vs_patched_reference():
    """One tensor at a time, the object protocol the loop relies on (quant_functions.py:306-318, 329-406): forward saves a
    CLONE of the input, backward recomputes from it, deletes it and resets saved_for_backward to None, a second backward
    raises, a new forward re-arms -- and the result matches the patched reference's within 1e-6 of sum|terms| per bucket
    at the LSTM's shapes (2000 x 1000 decoder cell, 500-element bias, ragged embedding slice)."""
    refq = ref_stage.load_patched()
    rng = np.random.RandomState(5)
    for shape, bucket, s in (((2000, 1000), 256, 16), ((500,), 256, 16), ((18000 * 5 + 123,), 256, 16), ((2000, 500), 100, 256)):
        x = (rng.randn(*shape) * 0.08).astype(np.float32)
        g = rng.randn(*shape).astype(np.float32)
        xd, gd = torch.from_numpy(x).to(DEV), torch.from_numpy(g).to(DEV)
        fn = product_quantization.uniformQuantization_variable(s, bucket_size=bucket)
        rf = refq.uniformQuantization_variable(s, bucket_size=bucket)
        q = fn.forward(xd)
        qr = rf.forward(torch.from_numpy(x).reshape(-1))
        assert np.array_equal(q.cpu().numpy().reshape(-1), qr.numpy())
        assert fn.saved_for_backward is not None and fn.saved_for_backward['input'].data_ptr() != xd.data_ptr()
        xd.add_(1.0)                                          # the loop restores the weights before backward: the clone is what counts
        out = fn.backward(gd)
        out_ref = rf.backward(torch.from_numpy(g).reshape(-1))
        assert fn.saved_for_backward is None and rf.saved_for_backward is None
        errlog.check_ste("K7 vs the (patched) reference's own fp32 backward at LSTM shapes", out.cpu().numpy(), x, g, s, bucket,
                         (shape, bucket, s), ref_out=out_ref.numpy())
        errlog.check_ste('K7 bucket sum vs float64 oracle at LSTM shapes', out.cpu().numpy(), x, g, s, bucket, (shape, bucket, s))
        with pytest.raises(ValueError):
            fn.backward(gd)
        fn.forward(xd)
        assert fn.saved_for_backward is not None
        del fn.saved_for_backward                             # :309-310
        fn.saved_for_backward = None
