"""Pin the CPU oracle (oracle/oracle_np.py) against vectors produced by the reference itself
(tests/golden/gen_golden.py).  Bit-exact everywhere except where the reference's own fp32
summation order is involved (mean, point-gradient sums, the sparse-mm of the 'complicated' STE)."""
import numpy as np

from oracle import oracle_np as onp

import errlog


def _b(c):
    return c['bucket']


def test_uniform_all_cases(golden_uniform):
    G = golden_uniform
    assert len(G.meta) > 200
    for i, c in enumerate(G.meta):
        x = G.arr('u', i, 'x')
        mean = c['mean'] if c['subtract_mean'] else None
        r = onp.uniform_quantize(x, c['s'], _b(c), c['max_element'], c['subtract_mean'], mean=mean)
        tag = 'case %d %r' % (i, c)
        assert np.array_equal(r['q'], G.arr('u', i, 'q')), tag
        assert np.array_equal(r['alpha'].reshape(-1), G.arr('u', i, 'alpha').reshape(-1)), tag
        assert np.array_equal(r['beta'].reshape(-1), G.arr('u', i, 'beta').reshape(-1)), tag
        assert np.array_equal(r['imin'].reshape(-1), G.arr('u', i, 'imin').reshape(-1)), tag
        assert np.array_equal(r['imax'].reshape(-1), G.arr('u', i, 'imax').reshape(-1)), tag
        assert np.array_equal(r['u'], G.arr('u', i, 'u')), tag
        assert np.array_equal(r['lev'], G.arr('u', i, 'lev')), tag
        assert list(r['u'].shape) == c['expected_tensor_size'], tag
        assert r['alpha'].shape == G.arr('u', i, 'alpha').shape, tag
        if c['subtract_mean']:
            # the oracle's own (float64) mean agrees with torch's fp32 mean to fp32 rounding
            m = onp.scale_down(x, _b(c), c['max_element'], True)['mean']
            assert abs(float(m) - c['mean']) <= 2e-7 * max(1.0, abs(c['mean'])) + 1e-9


def test_nonuniform_all_cases(golden_nonuniform):
    G = golden_nonuniform
    for i, c in enumerate(G.meta):
        x, pts = G.arr('n', i, 'x'), G.arr('n', i, 'pts')
        tag = 'case %d %r' % (i, c)
        r = onp.nonuniform_quantize(x, pts, _b(c), mode='distance')
        assert np.array_equal(r['idx'], G.arr('n', i, 'idx')), tag
        assert np.array_equal(r['q'], G.arr('n', i, 'q')), tag
        assert r['idx'].dtype == np.int64 and r['idx'].shape == x.shape
        r2 = onp.nonuniform_quantize(x, pts, _b(c), mode='midpoint')
        assert np.array_equal(r2['idx'], G.arr('n', i, 'idx_pre')), tag
        assert np.array_equal(r2['q'], G.arr('n', i, 'q_pre')), tag
        r3 = onp.nonuniform_quantize(x, G.arr('n', i, 'pts2'), _b(c), mode='midpoint')
        assert np.array_equal(r3['idx'], G.arr('n', i, 'idx_pre2')), tag
        assert np.array_equal(r3['q'], G.arr('n', i, 'q_pre2')), tag


def test_point_grad_all_cases(golden_nonuniform):
    G = golden_nonuniform
    for i, c in enumerate(G.meta):
        g = G.arr('n', i, 'g')
        for idx_key, gp_key in (('idx_pre', 'gp'), ('idx', 'gp_np'), ('idx_pre2', 'gp2')):
            got, absum = onp.point_grad(g, G.arr('n', i, idx_key), G.arr('n', i, 'alpha'), _b(c), c['k'])
            ref = G.arr('n', i, gp_key).astype(np.float64)
            errlog.check_sum("oracle (numpy, float64) vs the reference's fp32 gradPointTensor (golden)", got, ref, absum, (i, c, idx_key))


def test_ste_complicated_all_cases(golden_ste):
    G = golden_ste
    for i, c in enumerate(G.meta):
        x, g = G.arr('s', i, 'x'), G.arr('s', i, 'g')
        out = onp.ste_complicated_backward(x, g, c['s'], _b(c), tie_mode='reference')
        ref = G.arr('s', i, 'gout')
        errlog.check_ste("oracle (numpy) vs the (patched) reference's fp32 STE output (golden)", out, x, g, c['s'], _b(c), (i, c),
                         ref_out=ref)
        # elements that are neither argmax nor argmin of their bucket pass through untouched
        untouched = out == g
        assert untouched.sum() >= g.size - 2 * (-(-g.size // c['bucket']))


def test_roundtrip_and_layout(golden_misc):
    G = golden_misc
    for i, c in enumerate(G.meta['roundtrip']):
        x = G.z['rt%d_x' % i]
        sd = onp.scale_down(x, c['bucket'])
        assert np.array_equal(sd['u'], G.z['rt%d_u' % i])
        back = onp.inv_scale_down(sd['u'], sd['alpha'], sd['beta'], sd['mean'], sd['n'], sd['shape'])
        assert np.array_equal(back, G.z['rt%d_back' % i])


def test_init_points(golden_misc):
    G = golden_misc
    for i, c in enumerate(G.meta['init_points']):
        p = onp.init_points_percentile(G.z['ip%d_x' % i], c['bucket'], c['k'])
        assert np.array_equal(p, G.z['ip%d_p' % i]), (c, p, G.z['ip%d_p' % i])


def test_kats_from_survey():
    """Hand-checkable known answers (generated from the reference)."""
    x = np.array([0, 0.1, 0.25, 0.5, 0.75, 0.9, 1, -1], dtype=np.float32)
    r = onp.uniform_quantize(x, 4, None)
    t = np.float32(0.3333333730697632)
    assert np.array_equal(r['q'], np.array([t, t, t, t, 1, 1, 1, -1], dtype=np.float32))
    assert r['alpha'].tolist() == [2.0] and r['beta'].tolist() == [-1.0]
    assert r['imin'].tolist() == [7] and r['imax'].tolist() == [6]
    r = onp.uniform_quantize(x, 4, 4)
    assert np.array_equal(r['q'], np.array([0, 0.1666666716337204, 0.3333333432674408, 0.5, 1, 1, 1, -1], dtype=np.float32))
    r = onp.uniform_quantize(np.array([1, 2, 3, 4, 5, 7], dtype=np.float32), 4, 4)
    assert r['q'].tolist() == [1, 2, 3, 4, 5, 7] and r['u'].shape == (2, 4)
    assert r['alpha'].reshape(-1).tolist() == [3, 2] and r['beta'].reshape(-1).tolist() == [1, 5]
    r = onp.uniform_quantize(np.array([1, 2, 4], dtype=np.float32), 4, 256)
    assert r['q'].tolist() == [1, 2, 4] and r['u'].shape == (1, 3) and r['alpha'].shape == (1, 1)
    r = onp.uniform_quantize(np.full(5, 0.3, dtype=np.float32), 16, None)
    assert r['alpha'].tolist() == [1.0] and np.all(r['q'] == np.float32(0.3))
    assert onp.uniform_quantize(np.array([0, 0.5, 1], dtype=np.float32), 2, None)['q'].tolist() == [0, 0, 1]
    assert onp.uniform_quantize(np.array([0, 0.5, 1.5, 2.5, 3], dtype=np.float32), 4, None)['q'].tolist() == [0, 0, 2, 2, 3]
    pts = np.array([0, 0.25, 0.75, 1], dtype=np.float32)
    xx = np.array([0, 0.125, 0.3, 0.5, 0.74, 0.875, 1], dtype=np.float32)
    for mode in ('distance', 'midpoint'):
        r = onp.nonuniform_quantize(xx, pts, None, mode)
        assert r['idx'].tolist() == [0, 1, 1, 2, 2, 3, 3]
        assert r['q'].tolist() == [0, 0.25, 0.25, 0.75, 0.75, 1, 1]
    gp, _ = onp.point_grad(np.arange(1, 8, dtype=np.float32), r['idx'], r['alpha'], None, 4)
    assert gp.tolist() == [1, 5, 9, 13]
    r = onp.nonuniform_quantize(np.array([2, 4, 6, 10, -1, 0, 1, 3], dtype=np.float32), pts, 4)
    assert r['q'].tolist() == [2, 4, 8, 10, -1, 0, 2, 3] and r['idx'].tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    gp, _ = onp.point_grad(np.ones(8, dtype=np.float32), r['idx'], r['alpha'], 4, 4)
    assert gp.tolist() == [12, 12, 12, 12]


def test_bucket_geometry():
    assert onp.bucket_geometry(1000, 256) == (4, 256, 1024)
    assert onp.bucket_geometry(1024, 256) == (4, 256, 1024)
    assert onp.bucket_geometry(3, 256) == (1, 3, 3)
    assert onp.bucket_geometry(77, None) == (1, 77, 77)


def test_torch_port(golden_uniform):
    """oracle/torch_port.py (the op-for-op torch CPU port timed by bench.py) against the golden set."""
    import torch
    from oracle.torch_port import uniform_quantize_torch_ops
    G = golden_uniform
    for i, c in enumerate(G.meta):
        if c['subtract_mean'] or c['max_element'] is not False:
            continue
        q, alpha, beta = uniform_quantize_torch_ops(torch.from_numpy(G.arr('u', i, 'x')), c['s'], c['bucket'])
        assert np.array_equal(q.numpy(), G.arr('u', i, 'q')), (i, c)
        assert np.array_equal(alpha.numpy().reshape(-1), G.arr('u', i, 'alpha').reshape(-1)), (i, c)


def test_nonfinite_inputs(golden_nonfinite):
    """NaN poisons its bucket (torch min/max propagate NaN); +-inf does too, through alpha/beta."""
    from oracle import oracle_c as oc
    oc.build()
    G = golden_nonfinite
    for i, c in enumerate(G.meta):
        x = G.arr('f', i, 'x')
        for r in (onp.uniform_quantize(x, c['s'], c['bucket']), oc.uniform_quantize(x, c['s'], c['bucket'])):
            assert np.array_equal(r['q'], G.arr('f', i, 'q'), equal_nan=True), (i, c)
            assert np.array_equal(r['alpha'].reshape(-1), G.arr('f', i, 'alpha').reshape(-1), equal_nan=True), (i, c)
            assert np.array_equal(r['beta'].reshape(-1), G.arr('f', i, 'beta').reshape(-1), equal_nan=True), (i, c)
        assert np.isnan(G.arr('f', i, 'q')).any()


def test_nonuniform_options(golden_nonuniform_options):
    """max_element / subtract_mean on the non-uniform path; bit-exact given the reference's mean."""
    G = golden_nonuniform_options
    for i, c in enumerate(G.meta):
        x, pts = G.arr('o', i, 'x'), G.arr('o', i, 'pts')
        mean = c['mean'] if c['subtract_mean'] else None
        for mode, qk, ik in (('distance', 'q', 'idx'), ('midpoint', 'q_pre', 'idx_pre')):
            r = onp.nonuniform_quantize(x, pts, c['bucket'], mode, c['max_element'], c['subtract_mean'], mean=mean)
            assert np.array_equal(r['idx'], G.arr('o', i, ik)), (i, c, mode)
            assert np.array_equal(r['q'], G.arr('o', i, qk)), (i, c, mode)


def test_mean_options_oracle_reproduces_every_reference_run():
    """tests/golden/mean_options.npz: the reference at 1 / 2 / 4 / 8 torch threads on inputs whose elements sit on rounding
    boundaries.  Given the mean of a run, the oracle reproduces that run bit for bit -- so the only thing that separates two
    runs (or the device from a run) is the mean scalar.  Also pins what the file is for: the reference disagrees with itself."""
    from conftest import load_golden
    G = load_golden('mean_options.npz')
    self_disagreements = whole_level = 0
    for i, c in enumerate(G.meta):
        x = G.arr('m', i, 'x')
        runs = {}
        for th in c['q_stored_for_threads']:
            mean = c['mean_by_threads'][str(th)]
            r = onp.uniform_quantize(x, c['s'], c['bucket'], False, True, mean=mean)
            assert np.array_equal(r['q'], G.arr('m', i, 'q_t%d' % th)), (i, th)
            runs[th] = r['q']
        if len(runs) > 1:
            self_disagreements += 1
            a, *rest = runs.values()
            step = float(np.abs(x).max()) / c['s'] / 4
            whole_level += sum(int((np.abs(a.astype(np.float64) - b) > step).sum()) for b in rest)
    assert self_disagreements >= 5 and whole_level > 1000
