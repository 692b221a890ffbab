"""bench.py's multi-rank flow on a one-GPU box: two ranks on device 0, collectives through gloo (QD_BENCH_BACKEND=gloo,
QD_BENCH_ONE_GPU=1) -- everything the driver's `--gpus N` run goes through except RCCL itself: the launcher, the per-leg
agreement over the control group, the data-parallel report with rank 0 running alone while the others wait, the closing
barriers, ONE JSON line from rank 0 with the contract keys and the data-parallel keys of every steps/sec leg."""
import json
import os
import subprocess
import sys

import pytest

from harness import launch
from harness.dpbench import DP_KEYS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_quick_run_end_to_end():
    env = dict(os.environ, QD_BENCH_BACKEND='gloo', QD_BENCH_ONE_GPU='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = launch.launcher_command(os.path.join(ROOT, 'bench.py'), 2, ['--gpus', '2', '--steps', '5', '--warmup', '2', '--quick', '--no-cpu-baseline',
                                                                       '--no-pmc', '--no-kernels', '--precondition-s', '0.05',
                                                                       '--skip-legs', 'diffquant_wrn,nmt_lstm_dp'])     # (all four legs: tools/gpu_session.sh ranks2)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['scaling'] == 'weak' and d['collective_backend'] == 'gloo'
    assert d['value'] > 0 and d['roofline']['frac'] > 0
    assert abs(d['value'] - 2 * 8 * d['config']['n_elements_per_gpu'] / (d['ms_per_step'] * 1e-3) / 1e9) <= 2e-3 * d['value']
    legs_ = d['distill']
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
    # the scalars the driver's record keeps
    r = d['roofline']
    assert isinstance(r['steps_cfg1'], str) and 'multi' in r['steps_cfg1']
    for key in ('dp_cfg1', 'dp_cfg3_imagenet'):
        assert isinstance(r[key], str) and len(r[key]) <= 118 and 'N=2' in r[key], (key, r.get(key))


def test_a_failed_preflight_collective_still_yields_the_line():
    """If the data-path communicator cannot move bytes (here: rank 1 is told to fail its pre-flight all-reduce) the run must
    not hang in its first barrier: the headline is measured per GPU, every collective-bearing leg is skipped and recorded as
    such, rank 0 prints the line with the error, every rank exits 0."""
    env = dict(os.environ, QD_BENCH_BACKEND='gloo', QD_BENCH_ONE_GPU='1', QD_BENCH_TEST_PREFLIGHT_FAIL='1', QD_BENCH_DATA_TIMEOUT_S='8')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = launch.launcher_command(os.path.join(ROOT, 'bench.py'), 2, ['--gpus', '2', '--steps', '5', '--warmup', '2', '--quick', '--no-cpu-baseline',
                                                                       '--no-pmc', '--no-kernels', '--precondition-s', '0.05'])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert 'pre-flight' in d['error'] and 'pre-flight' in d['rccl_error'] and '1]' in d['rccl_error']     # rank 1, and rank 0 that waited for it in vain
    assert d['n_gpus'] == 2 and d['value'] > 0
    for name, leg in d['distill'].items():
        assert 'skipped' in leg and 'pre-flight' in leg['skipped'], (name, leg)


def test_a_leg_that_aborts_the_worker_costs_that_leg_only():
    """The round-4 failure, injected: the worker process dies with SIGABRT inside the optional hipGraph leg (after the headline,
    the kernel rows and the configs[1] leg).  The guardian (harness/guardian.py) records the leg as lost, starts a fresh worker
    for the legs that are left, and prints ONE line with everything else in it; exit code 0."""
    env = dict(os.environ, QD_BENCH_TEST_ABORT_IN='cifar_graph')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--quick', '--no-pmc',
                        '--precondition-s', '0.05', '--skip-legs', 'diffquant_wrn,nmt_lstm_dp,imagenet_resnet18k_dp,kernels',
                        '--deadline-s', '300'],          # (the guardian's own wall limit: a hang costs minutes, not the suite)
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['value'] > 0 and d['roofline']['frac'] > 0 and d['cpu_baseline']['value'] > 0
    bp = d['bench_process']
    assert bp['restarts'] == 1 and bp['workers'][0]['exit'] == 'SIGABRT' and bp['workers'][0]['during'] == 'cifar_graph' and bp['workers'][1]['exit'] == 0
    assert 'cifar_graph' in bp['legs_lost_with_their_worker']
    assert 'multi' in d['distill']['cifar_student']                         # measured by the first worker, kept
    assert 'SIGABRT' in d['distill']['cifar_graph']['error']                # the lost leg says so
    assert d['roofline']['pcie_inclusive_GBps_note'] > 0                    # measured by the second worker
    assert d['cpu_baseline']['distill']['steps_per_sec'] > 0
    assert list(d)[-1] == 'roofline'
