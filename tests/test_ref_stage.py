"""oracle/ref_stage.py: the reference's own quantization package, compiled to bytecode under the
git-ignored oracle/_ref/, is what bench.py times as `cpu_baseline` (kind "reference") on the GPU
box.  Here: it loads without disturbing this repository's `quantization` package, and it IS the
reference -- its outputs equal the golden vectors (which gen_golden.py produced from the sources)
and the oracle's."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from oracle import ref_stage

pytestmark = pytest.mark.skipif(not (ref_stage.is_staged() or os.path.exists(ref_stage.REF_ROOT)),
                                reason='nothing staged and no reference checkout')


def test_staged_reference_loads_beside_the_product_package():
    import quantization as product
    refq = ref_stage.load()
    assert refq is not None and refq is not product
    assert os.path.abspath(refq.__file__).startswith(ref_stage.STAGE_DIR)
    import sys
    assert sys.modules['quantization'] is product, 'the product package must stay registered under its name'
    assert refq.uniformQuantization.__module__ == 'quantization.quant_functions'
    assert not os.path.exists(os.path.join(ref_stage.PKG_DIR, 'quant_functions.py')), 'bytecode only: no source is copied'
    man = ref_stage.manifest()
    assert man and set(man['files']) == set(ref_stage.FILES)


def test_staged_reference_reproduces_golden_vectors(golden_uniform):
    refq = ref_stage.load()
    G = golden_uniform
    checked = 0
    for i, c in enumerate(G.meta):
        if c.get('stochastic') or c.get('type', 'linear') != 'linear':
            continue
        x = torch.from_numpy(G.arr('u', i, 'x').copy())
        kw = dict(bucket_size=c.get('bucket'), max_element=c.get('max_element', False),
                  subtract_mean=c.get('subtract_mean', False))
        q, sf = refq.uniformQuantization(x, c['s'], **kw)
        assert np.array_equal(q.numpy(), G.arr('u', i, 'q')), i
        checked += 1
        if checked >= 60:
            break
    assert checked >= 30


@pytest.mark.parametrize('n,bucket,s', [(100003, 256, 16), (4097, None, 4), (1000, 33, 16), (5, 256, 2)])
def test_staged_reference_equals_oracle(n, bucket, s):
    refq = ref_stage.load()
    x = torch.randn(n, generator=torch.Generator().manual_seed(n))
    q, sf = refq.uniformQuantization(x, s, bucket_size=bucket)
    want = onp.uniform_quantize(x.numpy(), s, bucket)
    assert np.array_equal(q.numpy(), want['q'])
    assert np.array_equal(sf.alpha.numpy().reshape(-1), want['alpha'].reshape(-1))


@pytest.mark.skipif(not (ref_stage.loop_is_staged() or os.path.exists(ref_stage.REF_ROOT)), reason='reference loop not staged')
def test_staged_reference_training_loop_runs_on_cpu_with_the_reference_quantizer():
    """oracle/_ref/loop: the reference's train_model (bytecode, `loss.data[0]` -> `loss.item()`) imports with
    `quantization` bound to the package handed to load_loop, leaves sys.modules as it found it, and trains."""
    import contextlib
    import io
    import sys
    import quantization as product
    refq = ref_stage.load()
    loop = ref_stage.load_loop(refq)
    assert loop is not None and loop.quantization is refq
    assert sys.modules['quantization'] is product and 'cnn_models' not in sys.modules
    assert ref_stage.load_loop(refq) is not loop, 'every call gives fresh module objects'
    torch.manual_seed(0)
    model = loop.ConvolForwardNet(**loop.smallerModelSpec, useBatchNorm=True, useAffineTransformInBatchNorm=True)
    g = torch.Generator().manual_seed(1)
    batches = [(torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)) for _ in range(2)]
    if loop.USE_CUDA:
        pytest.skip('a GPU is visible: tests/test_hip_dropin_reference_loop.py covers that case')
    with contextlib.redirect_stdout(io.StringIO()):
        _, info = loop.train_model(model, batches, batches[:1], epochs_to_train=1, print_every=1, quantizeWeights=True,
                                   numBits=4, bucket_size=256)
    assert info['errorFlag'] is False and info['numEpochsTrained'] == 1 and len(info['lossSaved']) == 1


@pytest.mark.skipif(not (ref_stage.patched_is_staged() or os.path.exists(ref_stage.REF_ROOT)), reason='patched reference not staged')
def test_staged_patched_reference_reproduces_the_ste_golden_vectors(golden_ste):
    """oracle/_ref/patched: the reference with the two shape fixes of SURVEY 8c applied to its source text in memory before
    compiling (ref_stage.stage_patched) -- what the GPU tests of the 'complicated' style compare with.  It IS the patched
    reference tests/golden/gen_golden.py ran: its backward reproduces every golden STE vector bit for bit; everything else
    in it is the unpatched reference (same forward, same golden vectors); and the UNPATCHED package still raises for more
    than one bucket, which is why the patched one exists (quant_functions.py:369-371,398-400)."""
    import sys
    import quantization as product
    patched = ref_stage.load_patched()
    plain = ref_stage.load()
    assert patched is not None and patched is not plain and patched is not product
    assert sys.modules['quantization'] is product
    assert os.path.abspath(patched.__file__).startswith(ref_stage.PATCHED_DIR)
    assert not os.path.exists(os.path.join(ref_stage.PATCHED_PKG, 'quant_functions.py')), 'bytecode only'
    G = golden_ste
    for i, c in enumerate(G.meta):
        x, g = torch.from_numpy(G.arr('s', i, 'x').copy()), torch.from_numpy(G.arr('s', i, 'g').copy())
        fn = patched.uniformQuantization_variable(c['s'], bucket_size=c['bucket'])
        q = fn.forward(x)
        assert np.array_equal(q.numpy(), G.arr('s', i, 'q')), (i, c)
        out = fn.backward(g.clone())
        assert fn.saved_for_backward is None
        assert np.array_equal(out.numpy(), G.arr('s', i, 'gout')), (i, c)
    x = torch.randn(1000)
    fn = plain.uniformQuantization_variable(16, bucket_size=256)
    fn.forward(x)
    with pytest.raises(RuntimeError):
        fn.backward(torch.randn(1000))
