"""CPU oracle (numpy) for the fake-quantization hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain restatement of the reference's algorithm
(antspy/quantized_distillation, quantization/quant_functions.py and
quantization/help_functions.py) in numpy fp32 arithmetic, one separately rounded IEEE operation
per reference tensor op, in the reference's op order.  It exists to CHECK the HIP path; nothing
in the product (`quantized_distillation_amd/`, `quantization/`) imports it.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may use it.

Parity pin: every function here is checked in tests/test_oracle_golden.py against the vectors in
tests/golden/*.npz, which were produced by running the unmodified reference
(tests/golden/gen_golden.py).  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so "the reference executed here" is the pin.

All `ref:` citations are relative to /root/reference/.
"""
import numpy as np

F32 = np.float32
TOL_DIFF_ZERO = 1e-10          # ref: quantization/quant_functions.py:40


# ----------------------------------------------------------------------------- bucket geometry
def bucket_geometry(n, bucket):
    """(num_buckets, row_length, padded_length) of the bucket view of an n-element tensor.

    ref: quantization/help_functions.py:67-94 -- n < bucket: one short row of n elements;
    n % bucket == 0: n/bucket full rows; otherwise one extra row padded with copies of the
    last element.  bucket None: a single row holding the whole tensor (1-D in the reference).
    """
    if bucket is None:
        return 1, n, n
    mult, rest = divmod(n, bucket)
    if mult == 0:
        return 1, n, n
    if rest == 0:
        return mult, bucket, n
    return mult + 1, bucket, (mult + 1) * bucket


def bucketize(v, bucket):
    """ref: help_functions.py:67-94 (fill_values='last').  v: 1-D fp32.  Returns 2-D rows
    (or the 1-D vector when bucket is None, as quant_functions.py:79-80 does)."""
    n = v.size
    if bucket is None:
        return v
    nb, row, padded = bucket_geometry(n, bucket)
    if padded != n:
        v = np.concatenate([v, np.full(padded - n, v[-1], dtype=F32)])
    return v.reshape(nb, row)


# ----------------------------------------------------------------------------- scaling
def scale_down(x, bucket=None, max_element=False, subtract_mean=False, mean=None):
    """Linear scaling to [0,1].  ref: quant_functions.py:56-107,129.

    Returns dict(u, alpha, beta, imin, imax, mean, n, shape).  `u` has the bucket layout
    (nb, row) -- or (n,) when bucket is None -- INCLUDING the padding of a ragged last bucket.
    `mean` overrides the fp32 mean (torch's summation order is not reproducible in numpy;
    tests pass the value the implementation under test reports).
    """
    x = np.asarray(x, dtype=F32)
    shape = x.shape
    v = x.reshape(-1).copy()
    n = v.size
    if subtract_mean:                                             # :66-68
        m = F32(np.mean(v, dtype=np.float64)) if mean is None else F32(mean)
        v = (v - m).astype(F32)
    else:
        m = F32(0.0)                                              # :70
    if max_element is not False:                                  # :72-74
        me = F32(max_element)
        v = np.where(v > me, me, v)
        v = np.where(v < -me, -me, v).astype(F32)
    t = bucketize(v, bucket)                                      # :78-81
    axis = 0 if bucket is None else 1
    mn = t.min(axis=axis, keepdims=True)                          # :85-90
    mx = t.max(axis=axis, keepdims=True)
    imin = t.argmin(axis=axis).reshape(mn.shape).astype(np.int64)  # first occurrence, like torch
    imax = t.argmax(axis=axis).reshape(mx.shape).astype(np.int64)
    alpha = (mx - mn).astype(F32)                                 # :91
    beta = mn.astype(F32)                                         # :92
    alpha = np.where(alpha < F32(TOL_DIFF_ZERO), F32(1.0), alpha).astype(F32)   # :95-99
    u = ((t - beta).astype(F32) / alpha).astype(F32)              # :106-107 (sub, then true div)
    return dict(u=u, alpha=alpha, beta=beta, imin=imin, imax=imax, mean=m, n=n, shape=shape)


def inv_scale_down(u, alpha, beta, mean, n, shape):
    """ref: quant_functions.py:131-152: mul, add (two roundings), add mean, strip padding."""
    y = (u * alpha).astype(F32)                                   # :142
    y = (y + beta).astype(F32)                                    # :143
    y = (y + F32(mean)).astype(F32)                               # :148
    return y.reshape(-1)[:n].reshape(shape)                       # :149-150


# ----------------------------------------------------------------------------- uniform
def uniform_quantize(x, s, bucket=None, max_element=False, subtract_mean=False, mean=None):
    """Deterministic k-level quantize-dequantize.  ref: quant_functions.py:155-194 (:189-191).

    Returns dict(q, lev, alpha, beta, imin, imax, mean, u).  `lev` = rint(u*(s-1)) as int32 in
    the bucket layout (padding included) -- the integer path that must match bit-exactly.
    """
    sd = scale_down(x, bucket, max_element, subtract_mean, mean)
    sm1 = F32(s - 1)                                              # :172
    t = (sd['u'] * sm1).astype(F32)                               # :189
    r = np.rint(t).astype(F32)                                    # :190 round half to even
    w = (r / sm1).astype(F32)                                     # :191
    q = inv_scale_down(w, sd['alpha'], sd['beta'], sd['mean'], sd['n'], sd['shape'])   # :193
    out = dict(sd)
    out.update(q=q, lev=r.astype(np.int32))
    return out


def uniform_quantize_stochastic(x, s, rand, bucket=None, max_element=False, subtract_mean=False, mean=None):
    """Stochastic-rounding variant given the uniform [0,1) draws `rand` (bucket layout).
    ref: quant_functions.py:174-187."""
    sd = scale_down(x, bucket, max_element, subtract_mean, mean)
    sm1 = F32(s - 1)
    prob = (sm1 * sd['u']).astype(F32)                            # :179
    t = (sd['u'] * sm1).astype(F32)                               # :180
    fl = np.floor(t).astype(F32)                                  # :181
    prob = (prob - fl).astype(F32)                                # :182
    w = (fl / sm1).astype(F32)                                    # :183
    inc = ((np.asarray(rand, dtype=F32).reshape(w.shape) <= prob).astype(F32) * F32(1.0) / sm1).astype(F32)
    w = (w + inc).astype(F32)                                     # :187
    q = inv_scale_down(w, sd['alpha'], sd['beta'], sd['mean'], sd['n'], sd['shape'])
    out = dict(sd)
    out.update(q=q)
    return out


def philox4x32_7_uniform(seed, n):
    """The uniform [0,1) draw of every element index 0..n-1 under the in-kernel generator of the HIP path
    (quantized_distillation_amd/csrc/qd_common.h: philox_uniform4) -- Philox4x32 with 7 rounds, counter =
    (element >> 2, 0x51ed270b, 0x2545f491), key = the 64-bit seed, component = element & 3, 24 mantissa bits.
    This restates OUR generator (the reference draws torch.rand on the host, quant_functions.py:185-186, which
    no device generator can reproduce); it lets the stochastic branch be checked bit for bit on every kernel
    path instead of only statistically."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    blocks = np.arange((n + 3) // 4, dtype=np.uint64)
    c0, c1 = blocks & mask, blocks >> np.uint64(32)
    c2 = np.full_like(blocks, 0x51ed270b)
    c3 = np.full_like(blocks, 0x2545f491)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(7):
        p0, p1 = M0 * c0, M1 * c2                                   # 32 x 32 -> 64-bit products
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    words = np.stack([c0, c1, c2, c3], axis=1).reshape(-1)[:n]
    return ((words >> np.uint64(8)).astype(np.float32) * F32(1.0 / 16777216.0)).astype(F32)


# ----------------------------------------------------------------------------- non-uniform
def assign_distance(u, pts):
    """Nearest sorted point, ties to the UPPER point.  ref: quant_functions.py:267-273."""
    pts = np.asarray(pts, dtype=F32)
    flat = u.reshape(-1)
    i = np.searchsorted(pts, flat, side='left').clip(max=pts.size - 1)            # :267-268
    lower = (i > 0) & (np.fabs(flat - pts[np.maximum(i - 1, 0)]) < np.fabs(flat - pts[i]))   # :269-272
    return (i - lower).astype(np.int64).reshape(u.shape)                           # :273


def midpoints(pts):
    """ref: quant_functions.py:533 -- k[:-1] + diff(k)/2 in fp32."""
    pts = np.asarray(pts, dtype=F32)
    return (pts[:-1] + (np.diff(pts) / F32(2.0)).astype(F32)).astype(F32)


def assign_midpoint(u, pts):
    """Index = #{midpoints <= u}: what SearchSorted.query computes through the sorted copy.
    ref: quant_functions.py:531-573 (searchsorted-left of the midpoints INTO the sorted tensor,
    :534, means element e gets index j iff exactly j midpoints are <= e)."""
    m = midpoints(pts)
    return np.searchsorted(m, u.reshape(-1), side='right').astype(np.int64).reshape(u.shape)


def nonuniform_quantize(x, pts, bucket=None, mode='distance', max_element=False, subtract_mean=False, mean=None):
    """ref: quant_functions.py:196-290.  mode 'distance' = plain path (:267-273), 'midpoint' =
    pre-processed path (:275 -> SearchSorted.query).  Returns dict(q, idx, alpha, beta, ...);
    idx is int64 with the ORIGINAL shape (:288-289)."""
    pts = np.asarray(pts, dtype=F32)
    sd = scale_down(x, bucket, max_element, subtract_mean, mean)
    idx_full = assign_distance(sd['u'], pts) if mode == 'distance' else assign_midpoint(sd['u'], pts)
    w = pts[idx_full]                                                               # :278
    q = inv_scale_down(w, sd['alpha'], sd['beta'], sd['mean'], sd['n'], sd['shape'])   # :286-287
    idx = idx_full.reshape(-1)[:sd['n']].reshape(sd['shape'])
    out = dict(sd)
    out.update(q=q, idx=idx)
    return out


def point_grad(g, idx, alpha, bucket, k, dtype=np.float64):
    """gradPoint[j] = sum_{i: idx_i == j} g_i * alpha_bucket(i).  ref: quant_functions.py:493-503.
    The product g*alpha is one fp32 multiply (:495); the per-bin sum is accumulated in `dtype`
    (float64 by default = the exact-ish value both torch's fp32 sum and the HIP reduction are
    compared against, with a tolerance relative to sum|g*alpha|)."""
    g = np.asarray(g, dtype=F32).reshape(-1)
    idx = np.asarray(idx).reshape(-1)
    n = g.size
    if bucket is None:
        a = np.broadcast_to(np.asarray(alpha, dtype=F32).reshape(-1)[:1], (n,))
    else:
        nb, row, _ = bucket_geometry(n, bucket)
        a = np.repeat(np.asarray(alpha, dtype=F32).reshape(-1), row)[:n]
    mg = (g * a).astype(F32)
    out = np.zeros(k, dtype=dtype)
    np.add.at(out, idx, mg.astype(dtype))
    absum = np.zeros(k, dtype=np.float64)
    np.add.at(absum, idx, np.abs(mg).astype(np.float64))
    return out, absum


# ----------------------------------------------------------------------------- STE variants
def ste_complicated_backward(x, g, s, bucket, tie_mode='reference'):
    """'complicated' straight-through backward.  ref: quant_functions.py:319-406 with the two
    shape fixes of SURVEY.md 8c (the shipped code raises for > 1 bucket).

    Closed form: per bucket S_b = sum_i g_i*(qs_i - u_i) over the real (non
    padded) elements, out = g, out[jmax_b] += S_b, out[jmin_b] -= S_b.
    Reference-faithful details reproduced here:
      * :350 re-runs scale_down on the QUANTIZED tensor, so alpha/beta/jmin/jmax are those of q
        (jmax = first element at the top level; tie_mode 'reference'), and qs = (q-beta_q)/alpha_q;
      * u_i = (x_i - beta_q)/alpha_q (:400) uses the same re-derived alpha/beta.
      * the sum is what torch.mm of the sparse +-1 matrix does: fp32 accumulate; we accumulate in
        float64 and compare with tolerance.
    tie_mode 'true_arg' uses argmax/argmin of x instead.
    """
    x = np.asarray(x, dtype=F32)
    g = np.asarray(g, dtype=F32)
    shape = x.shape
    n = x.size
    uq = uniform_quantize(x, s, bucket)
    q = uq['q']
    sdq = scale_down(q, bucket)                       # :350 (overwrites alpha, beta, idx)
    nb, row, padded = bucket_geometry(n, bucket)
    qs = sdq['u'].reshape(-1)[:n]
    alpha = np.repeat(sdq['alpha'].reshape(-1), row)[:n]
    beta = np.repeat(sdq['beta'].reshape(-1), row)[:n]
    u = ((x.reshape(-1) - beta).astype(F32) / alpha).astype(F32)       # :400 (tensor-beta)/alpha
    term = (g.reshape(-1) * (qs - u).astype(F32)).astype(F32)
    if tie_mode == 'reference':
        jmax, jmin = sdq['imax'].reshape(-1), sdq['imin'].reshape(-1)
    else:
        sdx = scale_down(x, bucket)
        jmax, jmin = sdx['imax'].reshape(-1), sdx['imin'].reshape(-1)
    out = g.reshape(-1).astype(np.float64).copy()
    for b in range(nb):
        lo, hi = b * row, min((b + 1) * row, n)
        sb = term[lo:hi].astype(np.float64).sum()
        # the reference adds grad_alpha^T . term: column jmax gets +sum over rows i of the bucket
        # whose (expanded, truncated to n) idx points at it
        out[lo + jmax[b]] += sb
        out[lo + jmin[b]] -= sb
    return out.astype(F32).reshape(shape)


def ste_bucket_terms(x, g, s, bucket, tie_mode='reference'):
    """The pieces of the 'complicated' STE backward per bucket, for tolerance checks of the bucket SUM (the only
    floating-point reduction of that function): terms t_i = g_i * (qs_i - u_i) exactly as the reference rounds them in
    fp32 (quant_functions.py:350, :400), then
        sb[b]        = float64 sum of the bucket's terms,
        abs_terms[b] = float64 sum of |t_i|  (what a summation error is measured against),
        jmax[b], jmin[b] = the two positions (relative to the bucket start) the sum is added to / subtracted from.
    Vectorised (no Python loop over buckets)."""
    x = np.asarray(x, dtype=F32).reshape(-1)
    g = np.asarray(g, dtype=F32).reshape(-1)
    n = x.size
    q = uniform_quantize(x, s, bucket)['q'].reshape(-1)
    sdq = scale_down(q, bucket)
    nb, row, padded = bucket_geometry(n, bucket)
    qs = sdq['u'].reshape(-1)[:n]
    alpha = np.repeat(sdq['alpha'].reshape(-1), row)[:n]
    beta = np.repeat(sdq['beta'].reshape(-1), row)[:n]
    u = ((x - beta).astype(F32) / alpha).astype(F32)
    term = (g * (qs - u).astype(F32)).astype(F32).astype(np.float64)
    starts = np.arange(nb, dtype=np.int64) * row
    sb = np.add.reduceat(term, starts)
    abs_terms = np.add.reduceat(np.abs(term), starts)
    if tie_mode == 'reference':
        jmax, jmin = sdq['imax'].reshape(-1), sdq['imin'].reshape(-1)
    else:
        sdx = scale_down(x, bucket)
        jmax, jmin = sdx['imax'].reshape(-1), sdx['imin'].reshape(-1)
    return {'sb': sb, 'abs_terms': abs_terms, 'jmax': np.asarray(jmax, np.int64), 'jmin': np.asarray(jmin, np.int64),
            'row': row, 'nb': nb}


def truncated_ste_mask(w, grad):
    """'truncated' STE: grad[|w| > 1] = 0.  ref: cnn_models/conv_forward_model.py:263-264."""
    w = np.asarray(w, dtype=F32)
    out = np.asarray(grad, dtype=F32).copy()
    out[np.abs(w) > F32(1.0)] = F32(0.0)
    return out


# ----------------------------------------------------------------------------- setup helpers
def init_points_percentile(x, bucket, k, max_element=False, subtract_mean=False):
    """ref: help_functions.py:140-154 -- np.percentile (linear interpolation, float64) of the
    scaled tensor without padding, cast back to fp32."""
    sd = scale_down(x, bucket, max_element, subtract_mean)
    flat = sd['u'].reshape(-1)[:sd['n']]
    return np.percentile(flat, np.linspace(0, 100, num=k)).astype(F32)


# ----------------------------------------------------------------------------- absmax / absnorm
# PARITY UNPINNED: the reference's code for these scaling types (quant_functions.py:109-127,144-146)
# raises on every torch version, so there is nothing to pin against.  This restates the math
# those lines evidently intend; it only checks that the HIP kernels do what DESIGN.md says.
def scale_down_abs(x, bucket=None, kind='absmax', norm=None):
    """sign, u = |x| / norm_b with norm_b = max|x| ('absmax') or sqrt(sum x^2) ('absnorm') per bucket,
    norm < 1e-10 -> 1.  `norm` overrides the per-bucket norms (fp32 summation order of the L2 norm
    is implementation specific).  Returns dict(u, sign, norm) in the padded bucket layout."""
    x = np.asarray(x, dtype=F32)
    v = x.reshape(-1)
    n = v.size
    t = bucketize(v, bucket)
    t2 = t.reshape(1, -1) if bucket is None else t
    sign = np.sign(t2).astype(F32)
    m = np.abs(t2)
    if norm is None:
        if kind == 'absmax':
            nrm = m.max(axis=1, keepdims=True).astype(F32)
        else:
            nrm = np.sqrt((m.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(F32)
        nrm = np.where(nrm < F32(TOL_DIFF_ZERO), F32(1.0), nrm).astype(F32)
    else:
        nrm = np.asarray(norm, dtype=F32).reshape(-1, 1)
    u = (m / nrm).astype(F32)
    return dict(u=u.reshape(t.shape), sign=sign.reshape(t.shape), norm=nrm.reshape(-1), n=n, shape=x.shape)


def uniform_quantize_abs(x, s, bucket=None, kind='absmax', norm=None):
    sd = scale_down_abs(x, bucket, kind, norm)
    sm1 = F32(s - 1)
    u2 = sd['u'].reshape(sd['norm'].size, -1)
    w = (np.rint((u2 * sm1).astype(F32)).astype(F32) / sm1).astype(F32)
    y = (w * sd['norm'].reshape(-1, 1)).astype(F32)
    y = (y * sd['sign'].reshape(u2.shape)).astype(F32)
    y = (y + F32(0.0)).astype(F32)
    out = dict(sd)
    out['q'] = y.reshape(-1)[:sd['n']].reshape(sd['shape'])
    return out
