"""The C oracle (oracle/qd_oracle.c) must agree bit-for-bit with the numpy oracle, with the
golden vectors produced by the reference, and with the reference's full-size checksums."""
import numpy as np
import pytest

from oracle import oracle_c as oc
from oracle import oracle_np as onp

import errlog


@pytest.fixture(scope='module', autouse=True)
def _build():
    oc.build()


def test_c_uniform_vs_golden(golden_uniform):
    G = golden_uniform
    for i, c in enumerate(G.meta):
        x = G.arr('u', i, 'x')
        r = oc.uniform_quantize(x, c['s'], c['bucket'], c['max_element'], c['subtract_mean'],
                                mean=c['mean'] if c['subtract_mean'] else None)
        tag = 'case %d %r' % (i, c)
        n = x.size
        assert np.array_equal(r['q'], G.arr('u', i, 'q')), tag
        assert np.array_equal(r['alpha'], G.arr('u', i, 'alpha').reshape(-1)), tag
        assert np.array_equal(r['beta'], G.arr('u', i, 'beta').reshape(-1)), tag
        assert np.array_equal(r['imin'], G.arr('u', i, 'imin').reshape(-1)), tag
        assert np.array_equal(r['imax'], G.arr('u', i, 'imax').reshape(-1)), tag
        assert np.array_equal(r['lev'], G.arr('u', i, 'lev').reshape(-1)[:n]), tag
        sd = oc.scale_down(x, c['bucket'], c['max_element'], c['subtract_mean'],
                           mean=c['mean'] if c['subtract_mean'] else None)
        assert np.array_equal(sd['u'], G.arr('u', i, 'u').reshape(-1)[:n]), tag


def test_c_nonuniform_vs_golden(golden_nonuniform):
    G = golden_nonuniform
    for i, c in enumerate(G.meta):
        x, pts = G.arr('n', i, 'x'), G.arr('n', i, 'pts')
        r = oc.nonuniform_quantize(x, pts, c['bucket'], 'distance')
        assert np.array_equal(r['idx'], G.arr('n', i, 'idx')) and np.array_equal(r['q'], G.arr('n', i, 'q')), (i, c)
        r = oc.nonuniform_quantize(x, pts, c['bucket'], 'midpoint')
        assert np.array_equal(r['idx'], G.arr('n', i, 'idx_pre')) and np.array_equal(r['q'], G.arr('n', i, 'q_pre')), (i, c)
        got, ab = oc.point_grad(G.arr('n', i, 'g'), G.arr('n', i, 'idx_pre'), G.arr('n', i, 'alpha'), c['bucket'], c['k'])
        ref = G.arr('n', i, 'gp').astype(np.float64)
        # the oracle sums in float64; the golden value is the reference's own fp32 sum: this is the REFERENCE's distance
        # from exact summation, which must itself sit well inside north_star's 1e-6
        errlog.check_sum("oracle (C, float64) vs the reference's fp32 gradPointTensor (golden)", got, ref, ab, (i, c))


def test_c_ste_vs_numpy_and_golden(golden_ste):
    G = golden_ste
    for i, c in enumerate(G.meta):
        x, g = G.arr('s', i, 'x'), G.arr('s', i, 'g')
        out = oc.ste_complicated_backward(x, g, c['s'], c['bucket'])
        assert np.array_equal(out, onp.ste_complicated_backward(x, g, c['s'], c['bucket'])), (i, c)
        errlog.check_ste("oracle (C) vs the (patched) reference's fp32 STE output (golden)", out, x, g, c['s'], c['bucket'], (i, c),
                         ref_out=G.arr('s', i, 'gout'))


def test_c_big_checksums_from_reference(golden_big):
    """Full pipelines at 100k..1M elements: histograms of the integer level path must match the
    reference exactly; float64 checksums of q to ~1e-12 relative (same fp32 values summed)."""
    import torch
    for c in golden_big:
        x = torch.randn(c['n'], generator=torch.Generator().manual_seed(c['seed'])).numpy()
        if c['op'] == 'uniform':
            assert abs(float(x.astype(np.float64).sum()) - c['x_sum']) < 1e-6     # same input stream
            r = oc.uniform_quantize(x, c['s'], c['bucket'])
            assert np.bincount(r['lev'], minlength=c['s']).tolist() == c['hist']
            s1, s2 = oc.checksum(r['q'])
            assert abs(s1 - c['sum_q']) <= 1e-9 * max(1.0, abs(c['sum_q2'])) and abs(s2 - c['sum_q2']) <= 1e-9 * c['sum_q2']
            assert [float(v) for v in r['q'][:5]] == c['q_head'] and [float(v) for v in r['q'][-3:]] == c['q_tail']
        else:
            pts = np.array(c['points'], dtype=np.float32)
            r = oc.nonuniform_quantize(x, pts, c['bucket'], 'distance')
            assert np.bincount(r['idx'], minlength=c['k']).tolist() == c['hist']
            s1, _ = oc.checksum(r['q'])
            assert abs(s1 - c['sum_q']) <= 1e-6 * max(1.0, abs(c['sum_q']))


def test_c_random_vs_numpy():
    rng = np.random.RandomState(0)
    for n, bucket, s in [(1, 256, 16), (255, 256, 4), (256, 256, 2), (10000, 256, 16), (10007, 100, 16),
                         (5000, None, 256), (777, 7, 3)]:
        x = rng.randn(n).astype(np.float32)
        a = onp.uniform_quantize(x, s, bucket)
        b = oc.uniform_quantize(x, s, bucket)
        assert np.array_equal(a['q'], b['q']) and np.array_equal(a['lev'].reshape(-1)[:n], b['lev'])
        assert np.array_equal(a['imin'].reshape(-1), b['imin']) and np.array_equal(a['imax'].reshape(-1), b['imax'])
    assert oc.max_threads() >= 1


# ---- property-based: the C oracle (the checker of the GPU property tests) against the numpy oracle (pinned to
# the reference's golden vectors line by line) on random sizes / buckets / levels / distributions, CPU only
from hypothesis import HealthCheck, given, settings          # noqa: E402
from hypothesis import strategies as st                       # noqa: E402

_buckets = st.sampled_from([None, 256, 64, 100, 7, 1, 33, 1000, 4, 2048, 513])
_sizes = st.one_of(st.integers(1, 600), st.integers(600, 20000))


def _make(n, seed, kind):
    rng = np.random.RandomState(seed)
    x = [rng.randn(n), rng.randn(n) * 1e-3 + 5.0, rng.randint(-4, 5, size=n).astype(np.float64), np.full(n, 0.25),
         rng.standard_cauchy(n), rng.rand(n) * 1e-30][kind]
    return x.astype(np.float32)


@settings(max_examples=60, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(n=_sizes, bucket=_buckets, s=st.sampled_from([2, 3, 4, 16, 255, 256, 1000]), seed=st.integers(0, 2 ** 31 - 1),
       kind=st.integers(0, 5), clamp=st.sampled_from([False, 0.5, 2.0]), sub=st.booleans())
def test_c_uniform_property_vs_numpy(n, bucket, s, seed, kind, clamp, sub):
    x = _make(n, seed, kind)
    mean = float(np.float32(x.astype(np.float64).mean())) if sub else None
    a = onp.uniform_quantize(x, s, bucket, clamp, sub, mean=mean)
    b = oc.uniform_quantize(x, s, bucket, clamp, sub, mean=mean)
    assert np.array_equal(a['q'], b['q'])
    assert np.array_equal(a['alpha'].reshape(-1), b['alpha']) and np.array_equal(a['beta'].reshape(-1), b['beta'])
    assert np.array_equal(a['imin'].reshape(-1), b['imin']) and np.array_equal(a['imax'].reshape(-1), b['imax'])
    assert np.array_equal(a['lev'].reshape(-1)[:n], b['lev'])


@settings(max_examples=40, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(n=_sizes, bucket=_buckets, k=st.sampled_from([1, 2, 3, 4, 16, 17, 64, 200]), seed=st.integers(0, 2 ** 31 - 1),
       kind=st.integers(0, 3), dup=st.booleans())
def test_c_nonuniform_property_vs_numpy(n, bucket, k, seed, kind, dup):
    x = _make(n, seed, kind)
    rng = np.random.RandomState(seed ^ 0x5bd1)
    pts = np.sort(rng.rand(k)).astype(np.float32)
    if dup and k > 2:
        pts[1], pts[-1] = pts[0], pts[-2]
    for mode in ('distance', 'midpoint'):
        a = onp.nonuniform_quantize(x, pts, bucket, mode)
        b = oc.nonuniform_quantize(x, pts, bucket, mode)
        assert np.array_equal(np.asarray(a['idx']).reshape(-1)[:n], b['idx'].reshape(-1)), mode
        assert np.array_equal(a['q'], b['q']), mode
    g = rng.randn(n).astype(np.float32)
    got, absum = oc.point_grad(g, b['idx'], b['alpha'], bucket, k)
    want, _ = onp.point_grad(g, b['idx'], b['alpha'], bucket, k)
    errlog.check_sum('oracle C vs oracle numpy (both float64 sums)', got, np.asarray(want, dtype=np.float64).reshape(-1), absum,
                     (n, bucket, k), tol=1e-12)


@settings(max_examples=30, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(n=_sizes, bucket=st.sampled_from([64, 256, 100, 7, 1024]), s=st.sampled_from([2, 4, 16, 256]),
       seed=st.integers(0, 2 ** 31 - 1), kind=st.integers(0, 3))
def test_c_ste_property_vs_numpy(n, bucket, s, seed, kind):
    x = _make(n, seed, kind)
    g = np.random.RandomState(seed ^ 77).randn(n).astype(np.float32)
    assert np.array_equal(oc.ste_complicated_backward(x, g, s, bucket), onp.ste_complicated_backward(x, g, s, bucket))
