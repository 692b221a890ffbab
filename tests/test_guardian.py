"""bench.py's line must survive whatever happens to the process that measures (harness/guardian.py).

Round 4's driver run ended with SIGABRT from a c10d watchdog thread inside an optional leg, 50 s after the headline had been
measured: no line, no measurement.  These tests run the same two-process shape on CPU with a worker (tests/fake_bench.py)
that aborts, exits, hangs or raises on purpose, alone and as one of two gloo ranks under torch.distributed.run.
"""
import json
import os
import signal
import subprocess
import sys
import tempfile
import time

import pytest

from harness import launch, report

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, 'fake_bench.py')


def detail_file():
    fd, path = tempfile.mkstemp(suffix='.json', prefix='qd_fake_detail_')
    os.close(fd)
    os.unlink(path)
    return path


def read_detail(path):
    """The full record the guardian wrote next to the compact line (None if it wrote none)."""
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return None
    finally:
        if os.path.exists(path):
            os.unlink(path)


def check_stdout_line(text):
    """What the driver does with stdout: the last line of its tail must be ONE complete JSON object."""
    assert len(text.encode()) <= report.LINE_LIMIT, len(text)
    return json.loads(text)


def run(*argv, timeout=120, env=None):
    """-> (process, stdout lines, the full record from the detail file).  stdout carries the compact line only."""
    e = {k: v for k, v in (env or os.environ).items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    path = detail_file()
    p = subprocess.run([sys.executable, FAKE, '--detail', path] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout, env=e)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    for l in lines:
        check_stdout_line(l)
    return p, lines, read_detail(path)


def test_clean_run_prints_exactly_one_line():
    p, lines, full = run()
    assert p.returncode == 0, p.stderr
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['value'] == 123.0 and d['roofline']['frac'] == 0.8 and d['detail'].endswith('.json')
    assert d['bench_process'] == {'workers': 1, 'restarts': 0, 'wall_s': d['bench_process']['wall_s'], 'last_exit': 0}
    assert full['value'] == 123.0 and full['a'] == full['b'] == full['c'] == 'ok'
    assert full['bench_process']['restarts'] == 0 and full['bench_process']['workers'] == [{'exit': 0, 'legs_done': 4}]
    assert 'bench record (full): ' in p.stderr                        # ... and the full record is on stderr too, never on stdout


def test_a_full_size_record_is_printed_as_a_line_the_driver_can_parse():
    """Round 5's defect: the 24 KB record was printed whole and the driver's record of stdout holds less than that.  The same
    record through the guardian: one line of at most 4096 bytes that parses from the last 6000 bytes of stdout and carries
    what the contract asks of `roofline` and `cpu_baseline`; the whole record is in the detail file."""
    p, lines, full = run('--fat')
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-2000:]
    assert len(lines[0].encode()) <= 4096
    d = json.loads(p.stdout[-6000:].splitlines()[-1])
    for key in report.CONTRACT:
        assert key in d, key
    assert d['metric'] == 'quantize_dequantize_GBps_64M_fp32_4bit' and d['dtype'] == 'f32' and d['value'] > 1000
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'avg_launch_us', 'rocprof_kernel_avg_us', 'rocprof_frac',
                'traffic_over_algorithmic', 'algorithmic_bytes_per_launch', 'worst_kernel', 'worst_kernel_frac'):
        assert key in r, key
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic'] > r['algorithmic_bytes_per_launch'] * 0.98
    c = d['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample', 'cpu_model', 'os_cpu_count', 'threads'):
        assert key in c, key
    assert c['kind'] == 'reference' and c['value'] > 1 and len(c['sample']) <= 160
    assert len(d['config']['workload']) <= 120 and d['config']['n_elements_per_gpu'] == 1 << 26
    assert d['parity_bit_exact_vs_reference'] is True and d['rccl_world_size'] == 1
    assert set(d['steps_per_sec']) >= {'cfg0_cpu_reference_quantizer', 'cfg1_cifar_student', 'cfg2_diffquant_wrn', 'cfg3_imagenet_resnet18k', 'cfg4_nmt_lstm'}
    assert all(isinstance(v, float) for v in d['steps_per_sec'].values())
    assert set(d['dp']) == {'cfg1', 'cfg2', 'cfg3', 'cfg4'} and all('dp_efficiency' in v for v in d['dp'].values())
    assert 'dropped_to_fit' not in d
    assert len(full['roofline']['kernels']) == 28 and len(json.dumps(full)) > 20000       # nothing lost: the detail file has it all


@pytest.mark.parametrize('how,shown', [('abort', 'SIGABRT'), ('exit', 7), ('raise', 1)])
def test_a_worker_that_dies_in_a_leg_costs_that_leg_only(how, shown):
    p, lines, full = run('--die-in', 'b', '--how', how)
    assert p.returncode == 0, p.stderr
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d['value'] == 123.0
    assert d['bench_process']['restarts'] == 1 and d['bench_process']['workers'] == 2 and d['bench_process']['legs_lost'] == ['b']
    assert full['a'] == 'ok' and full['c'] == 'ok' and 'b' not in full
    bp = full['bench_process']
    assert bp['restarts'] == 1 and len(bp['workers']) == 2
    assert bp['workers'][0]['exit'] == shown and bp['workers'][0]['during'] == 'b' and bp['workers'][1]['exit'] == 0
    assert 'b' in bp['legs_lost_with_their_worker'] and str(shown) in bp['legs_lost_with_their_worker']['b']
    assert full['lost'] == bp['legs_lost_with_their_worker']       # the fresh worker was told


def test_the_last_leg_dying_needs_no_restart():
    p, lines, full = run('--die-in', 'c')
    assert p.returncode == 0 and len(lines) == 1
    assert json.loads(lines[0])['bench_process']['last_exit'] == 'SIGABRT'
    assert full['a'] == full['b'] == 'ok' and 'c' not in full and full['bench_process']['restarts'] == 0


def test_dying_before_the_headline_is_an_error_line_and_a_nonzero_exit():
    p, lines, full = run('--die-in', 'headline')
    assert p.returncode == 1 and len(lines) == 1
    d = json.loads(lines[0])
    assert d['value'] is None and 'SIGABRT' in d['error']


def test_a_hanging_leg_is_ended_at_the_wall_limit_with_the_line_intact():
    t0 = time.time()
    p, lines, full = run('--die-in', 'b', '--how', 'hang', '--deadline-s', '3')
    assert time.time() - t0 < 30
    assert p.returncode == 0 and len(lines) == 1, (p.stdout, p.stderr)
    d = json.loads(lines[0])
    assert d['value'] == 123.0 and 'wall limit' in d['error']
    assert full['a'] == 'ok' and 'b' not in full and full['bench_process']['workers'][-1]['during'] == 'b'


def test_multi_rank_runs_are_not_restarted_and_other_ranks_stay_silent():
    p, lines, full = run('--die-in', 'b', '--world', '2', '--rank', '0')
    assert p.returncode == 0 and len(lines) == 1
    assert json.loads(lines[0])['bench_process']['restarts'] == 0
    assert full['a'] == 'ok' and 'c' not in full
    p, lines, full = run('--die-in', 'b', '--world', '2', '--rank', '1')
    assert p.returncode == 0 and lines == [] and full is None      # a non-zero exit would make torchrun tear rank 0 down before it prints


def test_sigterm_to_the_guardian_prints_the_line_and_ends_the_worker():
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    path = detail_file()
    p = subprocess.Popen([sys.executable, FAKE, '--detail', path, '--die-in', 'c', '--how', 'hang'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, env=e)
    time.sleep(3.0)                                        # headline, a, b are done by now; c hangs
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    lines = [l for l in out.splitlines() if l.strip()]
    assert p.returncode == 0 and len(lines) == 1, (out, err)
    d = check_stdout_line(lines[0])
    full = read_detail(path)
    assert d['value'] == 123.0 and 'SIGTERM' in d['bench_process']['last_exit']
    assert full['b'] == 'ok' and 'c' not in full and 'SIGTERM' in full['bench_process']['workers'][-1]['exit']


def test_rank_1_killed_with_sigabrt_mid_leg_rank_0_still_prints_the_headline():
    """Two gloo ranks under torch.distributed.run, each a guardian + worker pair.  Rank 1's worker aborts inside leg 'a' while
    rank 0 is in that leg's all-reduce: rank 0's collective gives up at the data group's timeout (4 s here, DATA_TIMEOUT_S =
    120 s in bench.py), the leg is recorded as failed, the later collective-bearing legs are skipped, and rank 0's guardian
    prints ONE line with the headline; the launcher exits 0."""
    t0 = time.time()
    path = detail_file()
    rc, out = launch.run_ranks(FAKE, 2, ['--detail', path, '--dist', '--die-in', 'a', '--die-rank', '1'], timeout=240, capture=True)
    took = time.time() - t0
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert rc == 0, out[-3000:]
    assert len(lines) == 1, out[-3000:]
    d = check_stdout_line(lines[0])
    full = read_detail(path)
    assert d['value'] == 123.0 and d['n_gpus'] == 2
    assert 'error' in full['a'] and 1 in full['a']['failed_ranks']
    assert 'skipped' in full['b'] and 'skipped' in full['c']
    assert took < 120, took


def test_two_gloo_ranks_full_size_record_rank_0_prints_the_compact_line():
    """The same at two ranks under torch.distributed.run: rank 0's line is the only JSON on the launcher's stdout, it has the
    same shape as at N = 1 and n_gpus = 2; rank 1 prints nothing."""
    path = detail_file()
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    p = subprocess.run(launch.launcher_command(FAKE, 2, ['--detail', path, '--dist', '--fat']), env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = check_stdout_line(p.stdout[-6000:].splitlines()[-1])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['roofline']['frac'] > 0.5 and d['cpu_baseline']['value'] > 1
    one = report.compact(json.load(open(os.path.join(HERE, 'golden', 'bench_record_full.json'))))
    assert set(one) - {'bench_process', 'detail'} <= set(d)             # the shape of the line does not depend on N
    assert read_detail(path)['n_gpus'] == 2


def test_bench_py_guardian_never_imports_torch():
    """The process that owns the line must not be able to die of torch / HIP / RCCL: bench.py's guardian path imports the
    standard library and harness.guardian / harness.launch only."""
    code = ("import sys; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench, harness.guardian, harness.launch; "
            "assert 'torch' not in sys.modules, 'torch imported'; print('ok')" % ROOT)
    p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == 'ok', p.stderr


def test_bench_legs_and_budget_table_are_consistent():
    sys.path.insert(0, ROOT)
    import bench
    assert set(bench.OPTIONAL) <= set(bench.LEGS) and set(bench.DISTILL_LEGS) <= set(bench.LEGS)
    assert bench.LEGS.index('cpu_baseline') < bench.LEGS.index('kernels') < bench.LEGS.index('cifar_student')
    first_optional = min(bench.LEGS.index(x) for x in bench.OPTIONAL)
    assert first_optional > bench.LEGS.index('cifar_student')       # nothing optional before the legs the metric names
    args = bench.parse_args(['--gpus', '8'])
    off = bench.disabled_legs(args, 8)
    assert {'cifar_graph', 'rocprof', 'cpu_baseline', 'pcie_note', 'cpu_distill'} <= off       # N > 1: no capture next to live collectives
    assert 'cifar_graph' not in bench.disabled_legs(bench.parse_args(['--gpus', '8', '--graph-at-any-n']), 8)
    assert bench.disabled_legs(bench.parse_args([]), 1) == set()


def test_a_signal_handler_installed_outside_python_does_not_break_the_guardian(tmp_path):
    """Under rocprofv3 SIGTERM / SIGINT already carry a C-level handler when the interpreter starts (its preloaded tool library
    installs them): signal.signal() then returns None as the previous handler, and putting None back raises TypeError -- which
    made the guardian exit 1 AFTER printing the line, and at N > 1 made torchrun tear rank 0 down before it printed (found by
    running bench.py under rocprofv3, tools/bench_under_rocprof.sh).  Reproduced with a preloaded library whose constructor
    installs the handlers."""
    import shutil
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        pytest.skip('no C compiler')
    src = tmp_path / 'pre.c'
    src.write_text('#include <signal.h>\nstatic void h(int s) { (void)s; }\n'
                   '__attribute__((constructor)) static void init(void) { signal(SIGTERM, h); signal(SIGINT, h); }\n')
    lib = tmp_path / 'libpre.so'
    subprocess.check_call([cc, '-shared', '-fPIC', str(src), '-o', str(lib)])
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    e['LD_PRELOAD'] = str(lib)
    probe = subprocess.run([sys.executable, '-c', 'import signal; print(signal.getsignal(signal.SIGTERM))'], env=e, stdout=subprocess.PIPE, text=True)
    assert probe.stdout.strip() == 'None'                       # the situation: a handler Python did not install
    p = subprocess.run([sys.executable, FAKE], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=e)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stderr[-1500:])
    assert check_stdout_line(lines[0])['value'] == 123.0
    # ... and a dying worker is still handled under it
    p = subprocess.run([sys.executable, FAKE, '--die-in', 'b'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=e)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])['bench_process']['restarts'] == 1


def test_a_sigkilled_guardian_takes_its_worker_with_it(tmp_path):
    """A hard kill of the guardian (a driver's timeout) must not leave the measuring process behind on the GPU: the worker asks
    the kernel to be killed with its parent (PR_SET_PDEATHSIG)."""
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.Popen([sys.executable, FAKE, '--die-in', 'c', '--how', 'hang'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
    time.sleep(3.0)
    kids = subprocess.run(['pgrep', '-P', str(p.pid)], stdout=subprocess.PIPE, text=True).stdout.split()
    assert len(kids) == 1, kids
    p.kill()
    p.wait(10)
    deadline = time.time() + 10
    while time.time() < deadline and os.path.exists('/proc/%s' % kids[0]):
        try:
            if open('/proc/%s/stat' % kids[0]).read().split()[2] == 'Z':      # reaped by init soon: gone for our purposes
                break
        except OSError:
            break
        time.sleep(0.2)
    alive = os.path.exists('/proc/%s' % kids[0]) and open('/proc/%s/stat' % kids[0]).read().split()[2] != 'Z'
    assert not alive, 'the worker survived its guardian'
