#!/usr/bin/env python3
"""Which of the kernel instantiations shipped in libqd_hip.so does the GPU test suite actually launch?

The launchers choose among dozens of instantiations per mode by bucket size, alignment, point count and tensor size
(csrc/qd_transform.h launch_bucketed, csrc/qd_reductions.hip); the tests are parameterised by those INPUTS, not by the kernel
that ends up running.  This tool closes the loop: it runs the parity suite under `rocprofv3 --kernel-trace --stats`, takes the
names of the kernels that were dispatched, and compares them with the kernels the shipped code objects contain
(tools/kernel_meta.py).  An instantiation nobody launches is either dead (delete it from the dispatch) or untested (give
it a test).

    python tools/launch_coverage.py --run [pytest args ...]     on the GPU box; writes gpurun_out/launch_coverage.{json,txt}
    python tools/launch_coverage.py --check FILE.json           anywhere: the committed record against the library as built NOW

tests/test_launch_coverage.py asserts --check on the committed profiles/r06_launch_coverage.json, so a dispatch change that
adds an instantiation fails the CPU suite until the coverage run has been repeated.

Kernels are matched by their demangled name without the argument list.  The same template can be instantiated in several
translation units (everything sits in anonymous namespaces: qd_transform.h is included four times); a name counts as
launched whichever unit's copy ran.
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import kernel_meta  # noqa: E402

LIB = os.path.join(ROOT, 'quantized_distillation_amd', 'libqd_hip.so')


def short_name(demangled):
    """`void (anonymous namespace)::k_x<0, 16>((anonymous namespace)::KParams) [clone .kd]` -> `k_x<0,16>`."""
    s = demangled.strip()
    s = re.sub(r'\s*\[clone [^\]]*\]\s*$', '', s)
    s = re.sub(r'\.kd$', '', s)
    if s.startswith('void '):
        s = s[5:]
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):                       # the argument list: the first '(' outside <> that is not "(anonymous namespace)"
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0 and not s.startswith('(anonymous namespace)', i):
            cut = i
            break
    s = s[:cut].replace('(anonymous namespace)::', '').replace(' ', '')
    return s


def shipped(lib=LIB):
    ks = kernel_meta.kernels(lib)
    dm = kernel_meta.demangle([k['name'] for k in ks])
    return sorted(set(short_name(dm[k['name']]) for k in ks))


def launched_from(stats_dir):
    names = {}
    files = glob.glob(os.path.join(stats_dir, '**', '*kernel_stats.csv'), recursive=True)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                nm = short_name(row.get('Name') or row.get('Kernel_Name') or '')
                names[nm] = names.get(nm, 0) + int(float(row.get('Calls') or row.get('Count') or 0))
    if not files:                                    # no stats file: fall back to the trace itself
        for f in glob.glob(os.path.join(stats_dir, '**', '*kernel_trace.csv'), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    nm = short_name(row.get('Kernel_Name', ''))
                    names[nm] = names.get(nm, 0) + 1
    return names


def report(ship, launched, meta):
    ours = {k: v for k, v in launched.items() if k in set(ship)}
    missing = [k for k in ship if k not in ours]
    rec = dict(meta)
    rec.update({'shipped_kernels': ship, 'n_shipped': len(ship), 'launched': {k: ours[k] for k in sorted(ours)},
                'n_launched': len(ours), 'unlaunched': missing})
    lines = ['# launch coverage of the GPU test suite: %d kernel names shipped in libqd_hip.so, %d launched, %d never launched'
             % (len(ship), len(ours), len(missing)), '# %s' % meta.get('command', ''), '']
    if missing:
        lines.append('NEVER LAUNCHED:')
        lines += ['  ' + k for k in missing]
        lines.append('')
    lines.append('launched (dispatches during the suite):')
    lines += ['  %8d  %s' % (ours[k], k) for k in sorted(ours)]
    return rec, '\n'.join(lines) + '\n'


def run(pytest_args):
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    out_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    # (the two files that test bench.py's PROCESS orchestration are left out: their workers abort on purpose, and a process that
    # calls abort() under rocprofv3 does not die -- the profiler's own SIGABRT handling hangs it (measured: the abort-injection
    # test of test_hip_bench_ranks.py sat there until its 900 s timeout).  They launch no kernel that the other files do not;
    # bench.py as a whole under rocprofv3 is tools/bench_under_rocprof.sh.)
    args = pytest_args or ['tests', '-q', '-m', 'gpu', '-x', '-p', 'no:cacheprovider',
                           '--ignore=tests/test_hip_bench_ranks.py', '--ignore=tests/test_hip_capture_watchdog.py']
    with tempfile.TemporaryDirectory(dir='/tmp') as td:
        cmd = [exe, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', td, '-o', 'cov', '--', sys.executable, '-m', 'pytest'] + args
        env = dict(os.environ, TMPDIR='/tmp')
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        tail = '\n'.join(r.stdout.splitlines()[-15:])
        launched = launched_from(td)
    ship = shipped()
    rec, txt = report(ship, launched, {'command': ' '.join(cmd[:9] + ['<tmp>', '-o', 'cov', '--', 'python', '-m', 'pytest'] + args),
                                       'pytest_rc': r.returncode, 'pytest_tail': tail})
    with open(os.path.join(out_dir, 'launch_coverage.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    with open(os.path.join(out_dir, 'launch_coverage.txt'), 'w') as f:
        f.write(txt)
    print(txt[:6000])
    print(tail)
    return 0 if (r.returncode == 0 and not rec['unlaunched']) else 1


def check(path, lib=LIB):
    """Problems of the committed record against the library as built now ([] = fine)."""
    with open(path) as f:
        rec = json.load(f)
    ship = shipped(lib)
    problems = []
    if rec.get('pytest_rc') != 0:
        problems.append('the recorded suite run failed (rc %r)' % rec.get('pytest_rc'))
    new = [k for k in ship if k not in set(rec['shipped_kernels'])]
    gone = [k for k in rec['shipped_kernels'] if k not in set(ship)]
    if new:
        problems.append('kernels in the library that the record has never seen (re-run tools/launch_coverage.py --run): %s' % new)
    if gone:
        problems.append('kernels in the record that are no longer shipped (re-run): %s' % gone)
    if rec['unlaunched']:
        problems.append('shipped but never launched by a test: %s' % rec['unlaunched'])
    return problems


if __name__ == '__main__':
    if '--run' in sys.argv:
        sys.exit(run([a for a in sys.argv[1:] if a != '--run']))
    if '--check' in sys.argv:
        p = check(sys.argv[sys.argv.index('--check') + 1])
        print('\n'.join(p) if p else 'coverage record is complete and current')
        sys.exit(1 if p else 0)
    print('\n'.join(shipped()))
