#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by RUNNING the reference.

Run in the authoring container only (it needs /root/reference, which does not exist on the
GPU box):

    python tests/golden/gen_golden.py

The reference package is imported from /root/reference *unmodified* (torch 2.10 CPU) and called
through its public API; nothing of it is copied here.  The only place where the shipped code
cannot be used as-is is uniformQuantization_variable.backward (it raises for more than one
bucket, quantization/quant_functions.py:369-370 and :398-400): for that case the two shape fixes
described in SURVEY.md section 8c are applied to the *source text at run time* (string replace on
inspect.getsource, then exec) so the golden output is still produced by the reference's code.

Outputs (all small, committed):
  uniform.npz      uniformQuantization + ScalingFunction side outputs over a case grid
  nonuniform.npz   nonUniformQuantization (plain and pre-processed) + point gradients
  ste.npz          patched 'complicated' backward
  nonuniform_options.npz   nonUniformQuantization with max_element / subtract_mean
  nonfinite.npz    uniformQuantization on inputs holding NaN / +-inf
  misc.npz         scale_down / inv_scale_down round trips, initialize_quantization_points,
                   assign_bits_automatically, huffman mean bit length
  coords.npz       cart2hyperspherical / hypershperical2cart / invert_pytorch_vector / findFirstNonZeroIndex
  big_checksums.json   float64 checksums / histograms of larger runs (no tensors stored)
  mean_options.npz subtract_mean=True on inputs built so that a last-bit change of the mean flips levels, with the
                   reference run at several torch thread counts (its fp32 mean depends on them)

    python tests/golden/gen_golden.py [name ...]      only the named outputs (uniform, nonuniform, ste, misc,
                                                      nonuniform_options, nonfinite, coords, big, mean_options)
"""
import inspect
import json
import os
import sys

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

# the reference package is literally called `quantization`; make sure it is the one we import
sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or '.') != os.path.abspath(os.path.join(HERE, '..', '..'))]
import quantization as refq                      # noqa: E402
import quantization.help_functions as refqhf     # noqa: E402
import quantization.quant_functions as refqf     # noqa: E402

assert os.path.abspath(refq.__file__).startswith(REF), refq.__file__
torch.set_num_threads(1)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def make_input(kind, shape, seed):
    g = gen(seed)
    n = int(np.prod(shape))
    if kind == 'randn':
        x = torch.randn(n, generator=g)
    elif kind == 'weights':
        x = 0.05 * torch.randn(n, generator=g)
    elif kind == 'const':
        x = torch.full((n,), 0.3)
    elif kind == 'ints':           # many exact ties for min/max first-occurrence and half-even
        x = torch.randint(-3, 4, (n,), generator=g).float()
    elif kind == 'halves':         # values on k+0.5 grid: exercises round-half-even
        x = torch.randint(0, 31, (n,), generator=g).float() * 0.5
    elif kind == 'uniform01':
        x = torch.rand(n, generator=g)
    else:
        raise ValueError(kind)
    return x.view(*shape).contiguous()


UNIFORM_CASES = []
_shapes = [(1,), (3,), (255,), (256,), (257,), (1000,), (4096,), (50, 75, 5), (500, 16), (10,), (33, 7, 3)]
_seed = 100
for shape in _shapes:
    for s in (2, 4, 16, 256):
        for bucket in (None, 256, 4, 100):
            n = int(np.prod(shape))
            if n > 5000 and not ((s == 16 and bucket in (None, 256, 100)) or (s == 4 and bucket == 256)):
                continue
            _seed += 1
            UNIFORM_CASES.append(dict(kind='randn', shape=shape, s=s, bucket=bucket, seed=_seed,
                                      max_element=False, subtract_mean=False))
for kind in ('weights', 'const', 'ints', 'halves', 'uniform01'):
    for shape in [(1000,), (257,), (64, 9)]:
        for s, bucket in ((16, 256), (4, None), (16, 4), (3, 100), (7, 256)):
            _seed += 1
            UNIFORM_CASES.append(dict(kind=kind, shape=shape, s=s, bucket=bucket, seed=_seed,
                                      max_element=False, subtract_mean=False))
for shape in [(1000,), (513,)]:
    for bucket in (None, 256):
        _seed += 1
        UNIFORM_CASES.append(dict(kind='randn', shape=shape, s=16, bucket=bucket, seed=_seed,
                                  max_element=0.5, subtract_mean=False))
        _seed += 1
        UNIFORM_CASES.append(dict(kind='randn', shape=shape, s=16, bucket=bucket, seed=_seed,
                                  max_element=False, subtract_mean=True))
        _seed += 1
        UNIFORM_CASES.append(dict(kind='randn', shape=shape, s=8, bucket=bucket, seed=_seed,
                                  max_element=1.25, subtract_mean=True))


def run_uniform():
    out = {}
    meta = []
    for i, c in enumerate(UNIFORM_CASES):
        x = make_input(c['kind'], c['shape'], c['seed'])
        x0 = x.clone()
        q, sf = refq.uniformQuantization(x, c['s'], type_of_scaling='linear', bucket_size=c['bucket'],
                                         max_element=c['max_element'], subtract_mean=c['subtract_mean'])
        assert torch.equal(x, x0)            # input untouched
        alpha, beta = sf.alpha.clone(), sf.beta.clone()
        imin, imax = sf.idx_min_rows.clone(), sf.idx_max_rows.clone()
        mean = float(sf.mean_tensor)
        # scale_down alone (fresh object, reference semantics) -> u in bucket layout, and the
        # level index rint(u*(s-1)) computed by the same torch ops the reference uses
        sf2 = refq.ScalingFunction('linear', c['max_element'], c['subtract_mean'], c['bucket'])
        u = sf2.scale_down(x)
        lev = torch.round(u.clone().mul_(c['s'] - 1)).to(torch.int32)
        k = 'u%03d_' % i
        out[k + 'x'] = x.numpy()
        out[k + 'q'] = q.numpy()
        out[k + 'alpha'] = alpha.numpy()
        out[k + 'beta'] = beta.numpy()
        out[k + 'imin'] = imin.numpy()
        out[k + 'imax'] = imax.numpy()
        out[k + 'u'] = u.numpy()
        out[k + 'lev'] = lev.numpy()
        m = dict(c)
        m['shape'] = list(c['shape'])
        m['mean'] = mean
        m['expected_tensor_size'] = list(sf.expected_tensor_size)
        m['original_tensor_length'] = int(sf.original_tensor_length)
        meta.append(m)
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, 'uniform.npz'), **out)
    print('uniform cases:', len(meta))


NONUNIFORM_CASES = []
_seed = 5000
for shape in [(7,), (256,), (257,), (1000,), (50, 75, 5), (3, 5, 7)]:
    for k in (2, 3, 4, 7, 16, 64):
        for bucket in (None, 256, 4, 100):
            n = int(np.prod(shape))
            if n > 5000 and (bucket in (4, 100) or k in (3, 7, 64)):
                continue
            _seed += 1
            NONUNIFORM_CASES.append(dict(shape=shape, k=k, bucket=bucket, seed=_seed, points='random'))
for k in (4, 16):
    for bucket in (None, 256):
        _seed += 1
        NONUNIFORM_CASES.append(dict(shape=(2000,), k=k, bucket=bucket, seed=_seed, points='grid'))
        _seed += 1
        NONUNIFORM_CASES.append(dict(shape=(2000,), k=k, bucket=bucket, seed=_seed, points='dups'))
        _seed += 1
        NONUNIFORM_CASES.append(dict(shape=(2000,), k=k, bucket=bucket, seed=_seed, points='percentile'))


def make_points(c, x):
    g = gen(c['seed'] + 77)
    k = c['k']
    if c['points'] == 'random':
        p = torch.sort(torch.rand(k, generator=g))[0]
    elif c['points'] == 'grid':          # x will be placed on midpoints -> exact ties
        p = torch.arange(k).float() / (k - 1)
    elif c['points'] == 'dups':          # duplicate points / empty bins
        p = torch.sort(torch.rand(k, generator=g))[0]
        p[1] = p[0]
        if k > 4:
            p[k // 2] = p[k // 2 - 1]
    elif c['points'] == 'percentile':
        sf = refq.ScalingFunction('linear', False, False, c['bucket'], False)
        p = refqhf.initialize_quantization_points(x, sf, k)
    return p.float().contiguous()


def run_nonuniform():
    out = {}
    meta = []
    for i, c in enumerate(NONUNIFORM_CASES):
        n = int(np.prod(c['shape']))
        if c['points'] == 'grid':
            # inputs already in [0,1] with min 0 and max 1 per bucket so that u == x exactly and
            # many elements sit exactly on midpoints between grid points
            g = gen(c['seed'])
            k = c['k']
            x = torch.randint(0, 2 * (k - 1) + 1, (n,), generator=g).float() / (2 * (k - 1))
            b = c['bucket'] or n
            x[0::b] = 0.0
            x[1::b] = 1.0
            x = x.view(*c['shape'])
        else:
            x = make_input('randn', c['shape'], c['seed'])
        pts = make_points(c, x)
        q, idx, sf = refq.nonUniformQuantization(x, pts, bucket_size=c['bucket'])
        # list form must give the same answer
        q_l, idx_l, _ = refq.nonUniformQuantization(x, [float(v) for v in pts], bucket_size=c['bucket'])
        assert torch.equal(q, q_l) and torch.equal(idx, idx_l)
        # pre-processed path (SearchSorted.query, midpoint formulation), first and second query
        fn = refq.nonUniformQuantization_variable(bucket_size=c['bucket'], pre_process_tensors=True, tensor=x)
        q_pre = fn.forward(None, pts).clone()
        idx_pre = fn.savedForBackward['indices'].clone()
        g = make_input('randn', c['shape'], c['seed'] + 1)
        _, gp = fn.backward(g)
        pts2 = torch.sort((pts + 0.03 * torch.randn(c['k'], generator=gen(c['seed'] + 5))).clamp(0, 1))[0]
        q_pre2 = fn.forward(None, pts2).clone()
        idx_pre2 = fn.savedForBackward['indices'].clone()
        _, gp2 = fn.backward(g)
        # non-preprocessed variable path must agree with the function
        fn_np = refq.nonUniformQuantization_variable(bucket_size=c['bucket'])
        q_v = fn_np.forward(x, pts)
        assert torch.equal(q_v, q)
        _, gp_np = fn_np.backward(g)
        kk = 'n%03d_' % i
        out[kk + 'x'] = x.numpy()
        out[kk + 'pts'] = pts.numpy()
        out[kk + 'q'] = q.numpy()
        out[kk + 'idx'] = idx.numpy().astype(np.int64)
        out[kk + 'alpha'] = sf.alpha.numpy()
        out[kk + 'beta'] = sf.beta.numpy()
        out[kk + 'q_pre'] = q_pre.numpy()
        out[kk + 'idx_pre'] = idx_pre.numpy().astype(np.int64)
        out[kk + 'g'] = g.numpy()
        out[kk + 'gp'] = gp.numpy()          # from the pre-processed indices
        out[kk + 'gp_np'] = gp_np.numpy()    # from the distance-rule indices
        out[kk + 'pts2'] = pts2.numpy()
        out[kk + 'q_pre2'] = q_pre2.numpy()
        out[kk + 'idx_pre2'] = idx_pre2.numpy().astype(np.int64)
        out[kk + 'gp2'] = gp2.numpy()
        m = dict(c)
        m['shape'] = list(c['shape'])
        m['pre_equals_plain'] = bool(torch.equal(idx_pre, idx))
        meta.append(m)
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, 'nonuniform.npz'), **out)
    print('nonuniform cases:', len(meta), 'pre==plain in', sum(m['pre_equals_plain'] for m in meta))


def patched_backward():
    """uniformQuantization_variable.backward with the two shape fixes of SURVEY.md 8c applied to
    the reference source text at run time (nothing is stored in this repository)."""
    src = inspect.getsource(refqf.uniformQuantization_variable.backward)
    a = "adder_for_buckets = torch.arange(0, self.bucket_size * total_num_buckets, self.bucket_size).long()"
    assert a in src
    src = src.replace(a, a + ".view(-1, 1)")
    b = "(grad_output*(quantized_tensor_unscaled-(tensor-beta)/alpha).view(-1)).view(-1,1))"
    assert b in src
    src = src.replace(b, b + ".view(-1)")
    import textwrap
    ns = dict(vars(refqf))
    exec(textwrap.dedent(src), ns)
    return ns['backward']


STE_CASES = []
_seed = 9000
for shape in [(256,), (1000,), (1024,), (257,), (4099,)]:     # 1-D only: the reference's view logic breaks for N-d
    for s in (4, 16):
        for bucket in (256, 64, 100):
            _seed += 1
            STE_CASES.append(dict(shape=shape, s=s, bucket=bucket, seed=_seed, kind='randn'))
_seed += 1
STE_CASES.append(dict(shape=(512,), s=16, bucket=256, seed=_seed, kind='ints'))


def run_ste():
    bw = patched_backward()
    out = {}
    meta = []
    for i, c in enumerate(STE_CASES):
        x = make_input(c['kind'], c['shape'], c['seed'])
        g = make_input('randn', c['shape'], c['seed'] + 1)
        fn = refq.uniformQuantization_variable(c['s'], bucket_size=c['bucket'])
        q = fn.forward(x)
        gout = bw(fn, g.clone())
        kk = 's%03d_' % i
        out[kk + 'x'] = x.numpy()
        out[kk + 'g'] = g.numpy()
        out[kk + 'q'] = q.numpy()
        out[kk + 'gout'] = gout.numpy()
        m = dict(c)
        m['shape'] = list(c['shape'])
        meta.append(m)
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, 'ste.npz'), **out)
    print('ste cases:', len(meta))


def run_misc():
    out = {}
    meta = {}
    # scale_down / inv_scale_down round trip, including the padded bucket layout
    rt = []
    for i, (shape, bucket) in enumerate([((1000,), 256), ((1000,), None), ((7, 11), 4), ((3,), 256), ((512,), 256)]):
        x = make_input('randn', shape, 300 + i)
        sf = refq.ScalingFunction('linear', False, False, bucket)
        u = sf.scale_down(x)
        back = sf.inv_scale_down(u)
        out['rt%d_x' % i] = x.numpy()
        out['rt%d_u' % i] = u.numpy()
        out['rt%d_back' % i] = back.numpy()
        rt.append(dict(shape=list(shape), bucket=bucket))
    meta['roundtrip'] = rt
    # percentile initialisation of the points
    ip = []
    for i, (n, bucket, k) in enumerate([(100003, 256, 4), (100003, None, 16), (5000, 256, 3), (777, 100, 8)]):
        x = torch.randn(n, generator=gen(1234 + i))
        sf = refq.ScalingFunction('linear', False, False, bucket, False)
        p = refqhf.initialize_quantization_points(x, sf, k)
        out['ip%d_x' % i] = x.numpy()
        out['ip%d_p' % i] = p.numpy()
        ip.append(dict(n=n, bucket=bucket, k=k))
    meta['init_points'] = ip
    # bit allocation heuristic
    ab = []
    rng = np.random.RandomState(7)
    for (m, init, is_point) in [(5, 4, False), (22, 16, True), (7, [2, 4, 4, 8, 2, 4, 4], False), (60, 4, True)]:
        norms = [float(v) for v in rng.rand(m) * 3 + 0.01]
        res = refqhf.assign_bits_automatically(norms, init, input_is_point=is_point)
        ab.append(dict(norms=norms, init=init, input_is_point=is_point, result=[int(v) for v in res]))
    meta['assign_bits'] = ab
    # huffman accounting through the public helper, uniform and nonuniform
    params = [torch.randn(n, generator=gen(4000 + n)) for n in (5000, 300, 64, 1000)]
    hf = []
    for s, bucket in ((16, 256), (4, None), (4, 256)):
        f = lambda t, s=s, bucket=bucket: refq.uniformQuantization(t, s, bucket_size=bucket)   # noqa: E731
        mbl = refqhf.get_huffman_encoding_mean_bit_length(iter(params), f, 'uniform', s=s)
        hf.append(dict(kind='uniform', s=s, bucket=bucket, mean_bit_length=float(mbl)))
    pts = torch.tensor([0.0, 0.4, 0.6, 1.0])
    f = lambda t: refq.nonUniformQuantization(t, pts, bucket_size=256)                       # noqa: E731
    mbl = refqhf.get_huffman_encoding_mean_bit_length(iter(params), f, 'nonuniform')
    hf.append(dict(kind='nonuniform', points=[0.0, 0.4, 0.6, 1.0], bucket=256, mean_bit_length=float(mbl)))
    for j, p in enumerate(params):
        out['hf_p%d' % j] = p.numpy()
    meta['huffman'] = hf
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, 'misc.npz'), **out)
    print('misc done')


def run_nonuniform_options():
    """nonUniformQuantization / nonUniformQuantization_variable with max_element and subtract_mean
    (plain and pre-processed): options no driver passes, but part of the signatures."""
    out = {}
    meta = []
    cases = []
    sd = 12000
    for shape in [(1000,), (513,), (40, 25)]:
        for bucket in (None, 256, 100):
            for (me, sm) in ((0.5, False), (False, True), (1.25, True)):
                sd += 1
                cases.append(dict(shape=list(shape), bucket=bucket, max_element=me, subtract_mean=sm, k=5, seed=sd))
    for i, c in enumerate(cases):
        x = make_input('randn', tuple(c['shape']), c['seed'])
        pts = torch.sort(torch.rand(c['k'], generator=gen(c['seed'] + 9)))[0].float()
        q, idx, sf = refq.nonUniformQuantization(x, pts, max_element=c['max_element'], subtract_mean=c['subtract_mean'],
                                                 bucket_size=c['bucket'])
        fn = refq.nonUniformQuantization_variable(max_element=c['max_element'], subtract_mean=c['subtract_mean'],
                                                  bucket_size=c['bucket'], pre_process_tensors=True, tensor=x)
        q_pre = fn.forward(None, pts).clone()
        idx_pre = fn.savedForBackward['indices'].clone()
        k = 'o%03d_' % i
        out[k + 'x'] = x.numpy()
        out[k + 'pts'] = pts.numpy()
        out[k + 'q'] = q.numpy()
        out[k + 'idx'] = idx.numpy().astype(np.int64)
        out[k + 'q_pre'] = q_pre.numpy()
        out[k + 'idx_pre'] = idx_pre.numpy().astype(np.int64)
        out[k + 'alpha'] = sf.alpha.numpy()
        m = dict(c)
        m['mean'] = float(sf.mean_tensor)
        meta.append(m)
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, 'nonuniform_options.npz'), **out)
    print('nonuniform option cases:', len(meta))


def run_nonfinite():
    """NaN / +-inf inputs: torch's min/max propagate NaN and an infinite alpha or beta turns the whole
    bucket into NaN through the arithmetic -- behaviour worth pinning because v_min/v_max on the
    GPU drop NaNs unless told otherwise."""
    out = {}
    meta = []
    cases = []
    for bucket in (256, None, 100):
        for s in (16, 4):
            cases.append(dict(n=1000, bucket=bucket, s=s, nan_at=[5], inf_at=[], ninf_at=[]))
            cases.append(dict(n=1000, bucket=bucket, s=s, nan_at=[], inf_at=[300], ninf_at=[]))
            cases.append(dict(n=1000, bucket=bucket, s=s, nan_at=[], inf_at=[], ninf_at=[600]))
            cases.append(dict(n=1000, bucket=bucket, s=s, nan_at=[999], inf_at=[10], ninf_at=[11]))
    cases.append(dict(n=40000, bucket=None, s=16, nan_at=[39999], inf_at=[], ninf_at=[]))
    cases.append(dict(n=40000, bucket=None, s=16, nan_at=[], inf_at=[123], ninf_at=[]))
    for i, c in enumerate(cases):
        x = torch.randn(c['n'], generator=gen(7000 + i))
        for j in c['nan_at']:
            x[j] = float('nan')
        for j in c['inf_at']:
            x[j] = float('inf')
        for j in c['ninf_at']:
            x[j] = float('-inf')
        q, sf = refq.uniformQuantization(x, c['s'], bucket_size=c['bucket'])
        k = 'f%03d_' % i
        out[k + 'x'] = x.numpy()
        out[k + 'q'] = q.numpy()
        out[k + 'alpha'] = sf.alpha.numpy()
        out[k + 'beta'] = sf.beta.numpy()
        meta.append(c)
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, 'nonfinite.npz'), **out)
    print('nonfinite cases:', len(meta))


def run_coords():
    """The hyperspherical-coordinate helpers of help_functions.py:8-64 (not on the quantization path; kept
    for a complete module surface).  Vectors with trailing zeros, a single non-zero head, all zeros, a
    negative last coordinate."""
    out = {}
    cases = []
    g = gen(11)
    vecs = [torch.randn(5, generator=g), torch.randn(2, generator=g), torch.randn(33, generator=g),
            torch.tensor([0.3, -1.2, 0.0, 0.0]), torch.tensor([-2.0, 0.0, 0.0]), torch.tensor([1.5, 0.0]),
            torch.zeros(4), torch.tensor([0.5, 0.25, -0.75]), torch.tensor([0.0, 0.0, 2.0]),
            torch.tensor([0.0, -3.0, 0.0, 0.0, 0.0])]
    for i, v in enumerate(vecs):
        r, ang = refqhf.cart2hyperspherical(v.clone())
        back = refqhf.hypershperical2cart((r, ang))
        out['c%d_x' % i] = v.numpy()
        out['c%d_r' % i] = np.asarray(float(r), dtype=np.float32)
        out['c%d_ang' % i] = ang.numpy()
        out['c%d_back' % i] = back.numpy()
        out['c%d_inv' % i] = refqhf.invert_pytorch_vector(v).numpy()
        cases.append({'first_nonzero': int(refqhf.findFirstNonZeroIndex(v))})
    out['meta'] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, 'coords.npz'), **out)
    print('coords cases:', len(cases))


def run_big():
    """Checksums of larger runs; inputs are re-creatable from the seed with torch.randn (CPU
    generator streams are identical for the same torch build), and are also re-derivable through
    the oracle.  Float64 accumulation, as in SURVEY.md appendix B."""
    res = []
    for n, s, bucket, seed in [(100003, 16, 256, 1234), (100003, 16, None, 1234), (1 << 20, 16, 256, 0),
                               ((1 << 20) + 17, 4, 256, 3), (1 << 20, 4, None, 5)]:
        x = torch.randn(n, generator=gen(seed))
        q, sf = refq.uniformQuantization(x, s, bucket_size=bucket)
        sf2 = refq.ScalingFunction('linear', False, False, bucket)
        lev = torch.round(sf2.scale_down(x).mul_(s - 1)).view(-1)[:n].long()
        res.append(dict(op='uniform', n=n, s=s, bucket=bucket, seed=seed,
                        x_sum=float(x.double().sum()),
                        sum_q=float(q.double().sum()), sum_q2=float((q.double() ** 2).sum()),
                        hist=[int(v) for v in torch.bincount(lev, minlength=s)],
                        q_head=[float(v) for v in q[:5]], q_tail=[float(v) for v in q[-3:]]))
    for n, k, bucket, seed in [(100003, 4, 256, 1234), (1 << 20, 16, 256, 0)]:
        x = torch.randn(n, generator=gen(seed))
        sf = refq.ScalingFunction('linear', False, False, bucket, False)
        pts = refqhf.initialize_quantization_points(x, sf, k)
        q, idx, _ = refq.nonUniformQuantization(x, pts, bucket_size=bucket)
        res.append(dict(op='nonuniform', n=n, k=k, bucket=bucket, seed=seed,
                        points=[float(v) for v in pts], sum_q=float(q.double().sum()),
                        hist=[int(v) for v in torch.bincount(idx.view(-1), minlength=k)]))
    with open(os.path.join(HERE, 'big_checksums.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print('big done')


def run_mean_options():
    """subtract_mean=True (ref: quant_functions.py:66-70,148) where the LAST BIT of the mean matters.

    The reference computes the mean with torch's fp32 CPU sum, whose value depends on how the sum is blocked: on the thread
    count and the SIMD width of the box (1-2 ulps here between 1, 2, 4 and 8 threads).  These cases make that visible:
    values on a (k + 0.5) * step grid -- every element sits exactly on a rounding boundary of the level computation once
    the mean is subtracted and added back -- plus noise-free offsets whose fp32 and float64 sums differ.  For every case the
    reference is run at 1, 2, 4 and 8 threads; the mean and the quantized tensor of each run are stored, so that a test
    can tell a deviation that comes from the mean scalar (explained by one of these runs or by the correctly rounded mean)
    from an arithmetic difference."""
    out, meta = {}, []
    g = gen(4242)
    cases = []
    for i in range(20):
        n = [4099, 40003, 65536, 70001, 100003][i % 5]
        s = [16, 4, 16, 256][i % 4]
        bucket = [256, None, 100, 256][(i // 2) % 4]
        step = [1.0, 0.25, 0.1, 3.0][i % 4]
        cases.append((n, s, bucket, step, i))
    for n, s, bucket, step, i in cases:
        lev = torch.randint(0, s - 1, (n,), generator=g).float()
        # (level + 0.5) * step: the half-way points of the level grid of the range [0, (s-1)*step], which one element at each
        # end pins in every bucket; then shifted by an offset whose fp32 sum is inexact
        x = (lev + 0.5) * step
        x[::47] = 0.0
        x[1::47] = (s - 1) * step
        offset = [0.1, 1.0 / 3.0, 1e-3, 7.7][i % 4]
        x = (x + offset).contiguous()
        k = 'm%03d_' % i
        out[k + 'x'] = x.numpy()
        runs = {}
        stored = {}                                      # mean -> thread count whose q is stored (one q per distinct mean)
        for th in (1, 2, 4, 8):
            torch.set_num_threads(th)
            q, sf = refq.uniformQuantization(x, s, bucket_size=bucket, subtract_mean=True)
            runs[th] = float(sf.mean_tensor)
            if runs[th] not in stored:
                stored[runs[th]] = th
                out[k + 'q_t%d' % th] = q.numpy()
        torch.set_num_threads(1)
        meta.append(dict(n=n, s=s, bucket=bucket, step=step, offset=offset, mean_by_threads={str(t): m for t, m in runs.items()},
                         q_stored_for_threads=sorted(stored.values()), mean_f64=float(x.double().mean())))
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, 'mean_options.npz'), **out)
    distinct = sum(1 for m in meta if len(set(m['mean_by_threads'].values())) > 1)
    print('mean_options cases:', len(meta), '; the reference disagrees with itself across thread counts on', distinct)


if __name__ == '__main__':
    todo = dict(uniform=run_uniform, nonuniform=run_nonuniform, ste=run_ste, misc=run_misc, nonuniform_options=run_nonuniform_options,
                nonfinite=run_nonfinite, coords=run_coords, big=run_big, mean_options=run_mean_options)
    for name in (sys.argv[1:] or list(todo)):
        todo[name]()
