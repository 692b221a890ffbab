"""Proof-by-exhaustion tool for the bucket-invariant division of the quantize kernels (qd_common.h: div_alpha<true>).

The reference divides every element by its bucket's alpha with an IEEE fp32 division
(quantization/quant_functions.py:106-107); the level index rint(u (s-1)) of :189-191 must match bit for bit.  The
kernels compute y = RN(1/alpha) ONCE per bucket and then, per element,
    q = RN(n y);   r = fma(-alpha, q, n);   u = fma(r, y, q)
which is the correctly rounded quotient RN(n / alpha) for alpha in [2^-60, 2^100] and n = 0 or n >= 2^-100 (Markstein's
division by a loop invariant).  This tool checks that claim two ways:

  device (needs a GPU):  python tools/div_invariant_check.py --pairs 1e9
      runs qd_selftest_div_invariant (csrc/qd_selftest.hip: the very inline function the kernels use, compiled with the
      library's flags) over `--pairs` adversarial pairs per family and prints tested / mismatches per family;
      exit status 1 on any mismatch.  Output of the committed run: docs/history/profiles/r03_div_invariant.txt.

  host (no GPU):         python tools/div_invariant_check.py --cpu 200000
      restates the three operations in EXACT rational arithmetic (fractions.Fraction, one explicit round-to-nearest-even
      to fp32 per operation, the FMAs rounded once) and compares with the exactly rounded quotient; also shows that the
      claim FAILS outside the stated range (tiny numerators, alpha beyond 2^126), i.e. that the test can fail.
      tests/test_host_logic.py runs a slice of it on every CPU test run.
"""
import argparse
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FAMILIES = ['quantizer domain (n = x - min, alpha = max - min)', 'wide exponents, random significands',
            'all-ones / near-power-of-two significands', 'near-exact quotients (level and half-level values +- 2 ulp)',
            'edges of the stated ranges (alpha = 2^-60 / 2^100, n = 2^-100 / alpha)',
            'small normal quotients 2^-120 .. 2^-50, n >= 2^-100 (scale_down returns u itself)']


# ---------------------------------------------------------------------------------------------- exact host restatement
def rn_f32(fr):
    """Round a Fraction to the nearest fp32 (ties to even), exactly; returns a Fraction that is an fp32 value
    (overflow -> None)."""
    if fr == 0:
        return Fraction(0)
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    e = a.numerator.bit_length() - a.denominator.bit_length()      # 2^(e-1) <= a < 2^(e+1)
    if a < Fraction(2) ** e:
        e -= 1
    e = max(e, -126)                                               # denormals share the quantum of 2^-126
    quantum = Fraction(2) ** (e - 23)
    m = a / quantum
    fl = m.numerator // m.denominator
    rem = m - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
        fl += 1
    out = fl * quantum
    if out >= Fraction(2) ** 128:
        return None
    return sign * out


def div_invariant_exact(n, alpha):
    """(u from the kernel's three operations, RN(n / alpha)) for fp32 n, alpha given as Fractions."""
    y = rn_f32(Fraction(1) / alpha)
    q = rn_f32(n * y)
    r = rn_f32(n - alpha * q)                # fma(-alpha, q, n): one rounding
    u = rn_f32(q + r * y)                    # fma(r, y, q): one rounding
    return u, rn_f32(n / alpha)


def f32_fraction(bits):
    return Fraction(float(np.array([bits], dtype=np.uint32).view(np.float32)[0]))


def host_pairs(rng, count, in_range=True):
    """(n, alpha) bit patterns: alpha with exponent in [-60, 100] (or outside when in_range is False), nasty significands
    half of the time, n <= 2^8 alpha and >= 2^-100."""
    nasty = np.array([0x7FFFFF, 0x7FFFFE, 0, 1, 2, 0x400000, 0x3FFFFF, 0x555555, 0x2AAAAA, 0x7FF000], dtype=np.uint32)
    out = []
    while len(out) < count:
        ea = int(rng.randint(-60, 101)) if in_range else int(rng.choice([rng.randint(-126, -61), rng.randint(101, 127)]))
        ma = int(nasty[rng.randint(len(nasty))]) if rng.rand() < 0.5 else int(rng.randint(0, 1 << 23))
        en = ea - int(rng.randint(0, 50)) + 8
        if in_range and en < -100:
            continue
        en = max(min(en, 126), -126)
        mn = int(nasty[rng.randint(len(nasty))]) if rng.rand() < 0.5 else int(rng.randint(0, 1 << 23))
        out.append((((en + 127) << 23) | mn, ((ea + 127) << 23) | ma))
    return out


def run_host(count, seed=0, verbose=True):
    rng = np.random.RandomState(seed)
    bad = 0
    for nb, ab in host_pairs(rng, count):
        u, want = div_invariant_exact(f32_fraction(nb), f32_fraction(ab))
        if want is None:
            continue
        bad += (u != want)
    # numerators of 0 and quotients that are exact
    for ab in (0x3F800000, 0x3F7FFFFF, 0x00800000 + (67 << 23), 0x7F000000 - (27 << 23)):
        a = f32_fraction(ab)
        for n in (Fraction(0), a, a / 2, a * Fraction(3, 4)):
            u, want = div_invariant_exact(rn_f32(n), a)
            bad += (u != want)
    out_bad = 0
    outside = host_pairs(rng, max(2000, count // 20), in_range=False)
    for nb, ab in outside:
        u, want = div_invariant_exact(f32_fraction(nb), f32_fraction(ab))
        if want is None or u is None:
            out_bad += 1
            continue
        out_bad += (u != want)
    if verbose:
        print('host, exact rational arithmetic: %d pairs inside the range, %d mismatches' % (count, bad))
        print('host, outside the range (alpha < 2^-60 or > 2^100): %d of %d pairs differ (the shortcut is NOT used there)'
              % (out_bad, len(outside)))
    return bad, out_bad


def run_host_small_quotients(count, seed=0, verbose=True):
    """Pairs with alpha in [2^-50, 2^100], n >= 2^-100 and quotient exponents down to -200.  Returns the mismatches among
    (normal quotients, denormal quotients, quotients that round to zero) and how many of each were tried: the shortcut is
    exact for the first and the last class and NOT always for the denormal one, which is why scale_down -- the only caller
    that returns the quotient itself -- refuses buckets with a numerator below alpha 2^-120."""
    rng = np.random.RandomState(seed)
    nasty = np.array([0x7FFFFF, 0x7FFFFE, 0, 1, 2, 0x400000, 0x3FFFFF, 0x555555, 0x2AAAAA, 0x7FF000], dtype=np.uint32)

    def mant():
        return int(nasty[rng.randint(len(nasty))]) if rng.rand() < 0.5 else int(rng.randint(0, 1 << 23))
    stats = {'normal': [0, 0], 'denormal': [0, 0], 'zero': [0, 0]}
    for _ in range(count):
        ea = int(rng.randint(-50, 101))
        en = int(rng.randint(-100, ea - 49))                       # quotient exponent in [-100 - ea, -50]
        n, a = f32_fraction(((en + 127) << 23) | mant()), f32_fraction(((ea + 127) << 23) | mant())
        u, want = div_invariant_exact(n, a)
        cls = 'zero' if want == 0 else ('denormal' if want < Fraction(2) ** -126 else 'normal')
        stats[cls][0] += 1
        stats[cls][1] += (u != want)
    if verbose:
        print('host, small quotients (n >= 2^-100): ' + ', '.join('%s %d tried / %d differ' % (k, v[0], v[1]) for k, v in stats.items()))
    return stats


# ------------------------------------------------------------------------------------------------------- device driver
def run_device(pairs, seed=1, chunk=1 << 28, verbose=True):
    import torch
    from quantized_distillation_amd import _lib
    lib = _lib.load()
    res = torch.zeros(4, dtype=torch.int64, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    total_bad = 0
    rows = []
    for fam, name in enumerate(FAMILIES):
        tested = bad = skipped = 0
        first = 0
        done = 0
        while done < pairs:
            m = int(min(chunk, pairs - done))
            _lib.check(lib.qd_selftest_div_invariant(seed * 1000003 + fam * 7919 + done // chunk, m, fam, res.data_ptr(), st))
            t, b, f, s = [int(v) for v in res.cpu().tolist()]
            tested += t
            bad += b
            skipped += s
            first = first or f
            done += m
        rows.append((fam, name, tested, bad, skipped, first))
        total_bad += bad
        if verbose:
            extra = '' if not bad else '   first mismatch: n bits 0x%08x, alpha bits 0x%08x' % ((first >> 32) & 0xFFFFFFFF, first & 0xFFFFFFFF)
            print('family %d  %-62s tested %13d  mismatches %d  (outside the domain, skipped: %d)%s'
                  % (fam, name, tested, bad, skipped, extra))
    return total_bad, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=float, default=0, help='device: pairs per family')
    ap.add_argument('--cpu', type=int, default=0, help='host: pairs in exact rational arithmetic')
    ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    rc = 0
    if a.cpu:
        bad, _ = run_host(a.cpu, a.seed)
        rc |= 1 if bad else 0
        st = run_host_small_quotients(a.cpu, a.seed)
        rc |= 1 if (st['normal'][1] or st['zero'][1]) else 0
    if a.pairs:
        import torch
        print('device: %s, torch %s, hip %s' % (torch.cuda.get_device_name(0), torch.__version__, torch.version.hip))
        bad, _ = run_device(int(a.pairs), a.seed)
        print('TOTAL mismatches: %d' % bad)
        rc |= 1 if bad else 0
    if not a.cpu and not a.pairs:
        ap.print_help()
    return rc


if __name__ == '__main__':
    sys.exit(main())
