"""bench.py's multi-rank flow on a one-GPU box: two ranks on device 0, collectives through gloo (QD_BENCH_BACKEND=gloo,
QD_BENCH_ONE_GPU=1) -- everything the driver's `--gpus N` run goes through except RCCL itself: the launcher, the per-leg
agreement over the control group, the data-parallel report with rank 0 running alone while the others wait, the closing
barriers, ONE JSON line from rank 0 with the contract keys and the data-parallel keys of every steps/sec leg."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

from harness import launch, report
from harness.dpbench import DP_KEYS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def detail_path():
    return os.path.join(tempfile.mkdtemp(prefix='qd_bench_detail_'), 'bench_detail.json')


def the_line(stdout):
    """What the driver does: ONE complete JSON object at the end of (the tail of) stdout, and nothing else on it."""
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout[-2000:]
    assert len(lines[0].encode()) <= report.LINE_LIMIT, len(lines[0])
    d = json.loads(stdout[-6000:].splitlines()[-1])
    for k in report.CONTRACT:
        assert k in d, k
    return d


def test_the_driver_command_prints_one_line_of_at_most_4096_bytes_with_roofline_and_cpu_baseline():
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5` (+ --quick: short steps/sec legs): the line parses from the last 6000
    bytes of stdout and carries the contract fields, `roofline` with measured traffic and `cpu_baseline`; everything else is in
    the detail file."""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    path = detail_path()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5', '--quick', '--detail', path],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = the_line(p.stdout)
    assert d['metric'] == 'quantize_dequantize_GBps_64M_fp32_4bit' and d['dtype'] == 'f32' and d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['higher_is_better'] is True
    assert '64Mi' in d['config']['workload'] and d['config']['levels'] == 16 and d['config']['bucket_size'] == 256
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and r['unit'] == 'GB/s' and 0.5 < r['frac'] < 1.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['traffic'] is not None and 0.98 <= r['traffic'] / r['algorithmic_bytes_per_launch'] <= 1.03 and r['traffic_measured_in_this_run'] is True
    assert 0.9 * r['avg_launch_us'] <= r['rocprof_kernel_avg_us'] <= 1.05 * r['avg_launch_us']      # (20 timed launches against 800 under the profiler)
    assert r['kernel_rows'] >= 20 and 0 < r['worst_kernel_frac'] < 1
    c = d['cpu_baseline']
    assert c['kind'] == 'reference' and c['value'] > 0 and c['cores'] >= 1 and c['unit'] == 'GB/s' and c['sample']
    assert d['parity_bit_exact_vs_reference'] is True and d['rccl_world_size'] == 1
    assert set(d['steps_per_sec']) >= {'cfg0_cpu_reference_quantizer', 'cfg1_cifar_student', 'cfg2_diffquant_wrn', 'cfg3_imagenet_resnet18k', 'cfg4_nmt_lstm'}
    assert all(isinstance(v, float) and v > 0 for v in d['steps_per_sec'].values()), d['steps_per_sec']
    assert 'dropped_to_fit' not in d and d['detail'] == 'bench_detail.json'
    full = json.load(open(path))
    assert len(full['roofline']['kernels']) == r['kernel_rows'] and full['value'] == d['value']
    assert report.compact(full) == json.loads(report.fit(report.compact(full)))       # the line IS the compact form of the record


def test_two_ranks_quick_run_end_to_end():
    env = dict(os.environ, QD_BENCH_BACKEND='gloo', QD_BENCH_ONE_GPU='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    path = detail_path()
    cmd = launch.launcher_command(os.path.join(ROOT, 'bench.py'), 2, ['--gpus', '2', '--steps', '5', '--warmup', '2', '--quick', '--no-cpu-baseline',
                                                                       '--no-pmc', '--no-kernels', '--precondition-s', '0.05', '--detail', path,
                                                                       '--skip-legs', 'diffquant_wrn,nmt_lstm_dp'])     # (all four legs: tools/gpu_session.sh ranks2)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = the_line(p.stdout)
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['scaling'] == 'weak' and d['collective_backend'] == 'gloo' and d['rccl_world_size'] == 2
    assert d['value'] > 0 and d['roofline']['frac'] > 0 and d['cpu_baseline'] is None
    assert abs(d['value'] - 2 * 8 * d['config']['n_elements_per_gpu'] / (d['ms_per_step'] * 1e-3) / 1e9) <= 2e-3 * d['value']
    # the data-parallel scalars of the line: numbers only, one group per config
    assert d['steps_per_sec']['cfg1_cifar_student'] > 0 and d['steps_per_sec']['cfg3_imagenet_resnet18k'] > 0
    for cfg in ('cfg1', 'cfg3'):
        for k in ('steps_per_sec', 'dp_efficiency', 'busbw_GBps', 'exposed_comm_ms', 'global_batch'):
            assert isinstance(d['dp'][cfg][k], (int, float)), (cfg, k)
        assert d['dp'][cfg]['busbw_GBps'] > 0
    full = json.load(open(path))
    legs_ = full['distill']
    assert 'legs_failed' not in legs_, legs_.get('legs_failed')
    assert 'diffquant_wrn' not in legs_ and 'nmt_lstm_dp' not in legs_
    for name in ('imagenet_resnet18k_dp',):
        leg = legs_[name]
        assert 'error' not in leg and 'skipped' not in leg, (name, leg)
        for k in DP_KEYS:
            assert k in leg, (name, k)
        assert leg['n_gpus'] == 2 and leg['global_batch'] == 2 * leg['per_gpu_batch']
        assert leg['rank_ms_per_step']['min'] <= leg['rank_ms_per_step']['max']
        assert leg['busbw_GBps'] > 0                       # 2 (N-1)/N x bytes / t with N = 2
    dp1 = legs_['cifar_student']['dp']
    for k in DP_KEYS:
        assert k in dp1, k
    assert d['dp']['cfg3']['dp_efficiency'] == legs_['imagenet_resnet18k_dp']['dp_efficiency']


def test_a_failed_preflight_collective_still_yields_the_line():
    """If the data-path communicator cannot move bytes (here: rank 1 is told to fail its pre-flight all-reduce) the run must
    not hang in its first barrier: the headline is measured per GPU, every collective-bearing leg is skipped and recorded as
    such, rank 0 prints the line with the error, every rank exits 0."""
    env = dict(os.environ, QD_BENCH_BACKEND='gloo', QD_BENCH_ONE_GPU='1', QD_BENCH_TEST_PREFLIGHT_FAIL='1', QD_BENCH_DATA_TIMEOUT_S='8')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    path = detail_path()
    cmd = launch.launcher_command(os.path.join(ROOT, 'bench.py'), 2, ['--gpus', '2', '--steps', '5', '--warmup', '2', '--quick', '--no-cpu-baseline',
                                                                       '--no-pmc', '--no-kernels', '--precondition-s', '0.05', '--detail', path])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = the_line(p.stdout)
    assert 'pre-flight' in d['error'] and 'pre-flight' in d['rccl_error'] and '1]' in d['rccl_error']     # rank 1, and rank 0 that waited for it in vain
    assert d['n_gpus'] == 2 and d['value'] > 0
    assert all(isinstance(v, str) and 'pre-flight' in v for v in d['steps_per_sec'].values()), d['steps_per_sec']
    for name, leg in json.load(open(path))['distill'].items():
        assert 'skipped' in leg and 'pre-flight' in leg['skipped'], (name, leg)


def test_a_leg_that_aborts_the_worker_costs_that_leg_only():
    """The round-4 failure, injected: the worker process dies with SIGABRT inside the optional hipGraph leg (after the headline,
    the kernel rows and the configs[1] leg).  The guardian (harness/guardian.py) records the leg as lost, starts a fresh worker
    for the legs that are left, and prints ONE line with everything else in it; exit code 0."""
    env = dict(os.environ, QD_BENCH_TEST_ABORT_IN='cifar_graph')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    path = detail_path()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--quick', '--no-pmc',
                        '--precondition-s', '0.05', '--skip-legs', 'diffquant_wrn,nmt_lstm_dp,imagenet_resnet18k_dp,kernels',
                        '--deadline-s', '300', '--detail', path],          # (the guardian's own wall limit: a hang costs minutes, not the suite)
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    line = the_line(p.stdout)
    assert line['value'] > 0 and line['roofline']['frac'] > 0 and line['cpu_baseline']['value'] > 0
    assert line['bench_process']['restarts'] == 1 and line['bench_process']['legs_lost'] == ['cifar_graph']
    assert 'SIGABRT' in line['steps_per_sec']['cfg1_hipgraph'] and line['steps_per_sec']['cfg1_cifar_student'] > 0
    d = json.load(open(path))
    bp = d['bench_process']
    assert bp['restarts'] == 1 and bp['workers'][0]['exit'] == 'SIGABRT' and bp['workers'][0]['during'] == 'cifar_graph' and bp['workers'][1]['exit'] == 0
    assert 'cifar_graph' in bp['legs_lost_with_their_worker']
    assert 'multi' in d['distill']['cifar_student']                         # measured by the first worker, kept
    assert 'SIGABRT' in d['distill']['cifar_graph']['error']                # the lost leg says so
    assert d['roofline']['pcie_inclusive_GBps_note'] > 0                    # measured by the second worker
    assert d['cpu_baseline']['distill']['steps_per_sec'] > 0
