"""Tolerance of the floating-point REDUCTIONS of the path, pinned where `north_star` puts it: 1e-6, relative to the sum of
the magnitudes of the terms ("relative to sum |g alpha| rather than to the (possibly cancelling)
result").  Everything else on the path is bit-exact and never comes through here.

Every check also RECORDS the error it achieved (error / sum|terms|) as one JSON line in gpurun_out/reduction_error.jsonl,
so that the margin is known, not assumed; tools/summarize_reduction_error.py turns the log of a GPU run into
docs/history/profiles/r04_reduction_error.txt.
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, 'gpurun_out', 'reduction_error.jsonl')
TOL = 1e-6                     # north_star: "within 1e-6 relative fp32 tolerance"


def _record(kind, tag, ratio, n_terms=None, tol=TOL):
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, 'a') as f:
            f.write(json.dumps({'kind': kind, 'case': str(tag)[:160], 'err_over_sum_abs_terms': float(ratio),
                                'tol': tol, 'n_terms': None if n_terms is None else int(n_terms)}) + '\n')
    except OSError:
        pass


def check_sum(kind, got, want, abs_terms, tag='', tol=TOL, n_terms=None):
    """got: the kernel's fp32 sums; want: the same sums accumulated in float64 (the oracle); abs_terms: sum of |term| per
    sum.  Requires |got - want| <= tol * abs_terms for every sum (exact equality where abs_terms == 0) and records
    max(|got - want| / abs_terms)."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    abs_terms = np.asarray(abs_terms, dtype=np.float64).reshape(-1)
    assert got.shape == want.shape == abs_terms.shape, (got.shape, want.shape, abs_terms.shape)
    err = np.abs(got - want)
    nz = abs_terms > 0
    ratio = float(np.max(err[nz] / abs_terms[nz])) if nz.any() else 0.0
    _record(kind, tag, ratio, n_terms, tol)
    assert np.all(err[~nz] == 0), (kind, tag, 'a sum with no non-zero term must be exactly 0', got[~nz])
    assert ratio <= tol, (kind, tag, 'error / sum|terms| = %.3g > %.1g' % (ratio, tol))
    return ratio


def check_mean(kind, got, want, abs_mean, tag='', tol=TOL, n_terms=None):
    """mean(x): |got - want| <= tol * mean|x| (the terms are the elements / n)."""
    err = abs(float(got) - float(want))
    ratio = err / abs_mean if abs_mean > 0 else (0.0 if err == 0 else float('inf'))
    _record(kind, tag, ratio, n_terms, tol)
    assert ratio <= tol, (kind, tag, 'error / mean|x| = %.3g > %.1g' % (ratio, tol))
    return ratio


def check_ste(kind, out, x, g, s, bucket, tag='', tol=TOL, tie_mode='reference', ref_out=None):
    """'complicated' STE backward (K7): every position but the two touched ones per bucket equals g exactly; the touched
    ones hold g +- S_b where the fp32 bucket sum S_b may differ from the float64 sum of the same (reference-rounded) terms
    by tol * sum|terms| (+ the rounding of the final fp32 add).  Records max |S_kernel - S_oracle| / sum|terms|.
    ref_out: compare with this output (the reference's own fp32 result) instead of with the float64 sums."""
    from oracle import oracle_np as onp
    out = np.asarray(out, dtype=np.float32).reshape(-1)
    g = np.asarray(g, dtype=np.float32).reshape(-1)
    T = onp.ste_bucket_terms(x, g, s, bucket, tie_mode)
    row, nb = T['row'], T['nb']
    starts = np.arange(nb, dtype=np.int64) * row
    pmax, pmin = starts + T['jmax'], starts + T['jmin']
    touched = np.zeros(out.size, bool)
    live = pmax != pmin                                   # a constant bucket: +S and -S cancel, nothing is touched
    touched[pmax[live]] = True
    touched[pmin[live]] = True
    assert np.array_equal(out[~touched], g[~touched]), (kind, tag, 'untouched positions must equal the incoming gradient')
    g64 = g.astype(np.float64)
    want_max = g64[pmax] + T['sb']
    want_min = g64[pmin] - T['sb']
    if ref_out is not None:
        ref_out = np.asarray(ref_out, dtype=np.float64).reshape(-1)
        want_max, want_min = ref_out[pmax], ref_out[pmin]
    err = np.maximum(np.abs(out[pmax].astype(np.float64) - want_max), np.abs(out[pmin].astype(np.float64) - want_min))[live]
    # the final fp32 add rounds once more: half an ulp of the result, i.e. <= 2^-24 (|g_j| + |S|)
    denom = (T['abs_terms'] + np.maximum(np.abs(g64[pmax]), np.abs(g64[pmin])))[live]
    nz = denom > 0
    ratio = float(np.max(err[nz] / denom[nz])) if nz.any() else 0.0
    _record(kind, tag, ratio, row, tol)
    assert np.all(err[~nz] == 0), (kind, tag)
    assert ratio <= tol, (kind, tag, 'bucket-sum error / (sum|terms| + |g_j|) = %.3g > %.1g' % (ratio, tol))
    return ratio
