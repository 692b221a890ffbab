"""bench.py's failure containment and data-parallel report on CPU: two gloo ranks (harness/legs.py, harness/dpbench.py)."""
import json
import os
import threading
import time

from harness import launch, legs
from harness.dpbench import DP_KEYS

HERE = os.path.dirname(os.path.abspath(__file__))


def test_a_rank_that_raises_inside_a_leg_cannot_hang_the_others():
    t0 = time.time()
    rc, out = launch.run_ranks(os.path.join(HERE, 'leg_worker.py'), 2, [], timeout=240, capture=True)
    took = time.time() - t0
    assert rc == 0, out
    d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
    assert d['fine'] == {'sum': 2.0}
    # the leg in which rank 1 raised: an error on EVERY rank (rank 0 reports it too), naming the rank
    assert 'error' in d['breaks'] and d['breaks']['failed_ranks'] in ([1], [0, 1])
    # later collective-bearing legs are skipped, legs without collectives still run
    assert 'skipped' in d['after'] and 'breaks' in d['after']['skipped']
    assert d['local'] == {'rank': 0}
    assert d['history'] and d['history'][0][0] == 'breaks'
    assert d['seconds'] < 30 and took < 120, (d['seconds'], took)


def test_data_parallel_report_keys_at_world_2():
    rc, out = launch.run_ranks(os.path.join(HERE, 'leg_worker.py'), 2, [], timeout=240, capture=True)
    assert rc == 0, out
    d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])['dp']
    for k in DP_KEYS:
        assert k in d, k
    assert d['n_gpus'] == 2 and d['global_batch'] == 16 and len(d['ms_per_step_repetitions']) == 3
    assert d['steps_per_sec_min'] <= d['steps_per_sec'] <= d['steps_per_sec_max']
    assert d['rank_ms_per_step']['min'] <= d['rank_ms_per_step']['max']
    assert d['exchanged_bytes_per_step'] == 4 << 16
    # ring bus bandwidth = 2 (N-1)/N x bytes / time = algorithmic bandwidth at N = 2
    assert abs(d['busbw_GBps'] - d['algbw_GBps']) <= 0.11
    assert abs(d['exposed_comm_ms'] - (d['ms_per_step'] - d['ms_per_step_without_exchange'])) < 2e-3
    assert abs(d['dp_efficiency'] - d['ms_per_step_one_gpu_alone'] / d['ms_per_step']) < 1e-3


def test_single_rank_runner_and_deadline():
    r = legs.LegRunner()
    assert r.world == 1 and r.ctl is None
    assert r.run('ok', lambda: 5) == 5
    bad = r.run('bad', lambda: 1 / 0)
    assert 'ZeroDivisionError' in bad['error'] and bad['failed_ranks'] == [0]
    assert r.broken is None                  # one rank: nobody can be left inside a collective
    assert r.run('next', lambda: 6) == 6
    d = legs.Deadline(60, lambda: None)
    d.cancel()
    d._thread.join(5)
    assert not d._thread.is_alive()
    assert not any(t.name == 'bench-deadline' and t.is_alive() for t in threading.enumerate())


def test_rccl_env_defaults_do_not_override_the_caller():
    env = {'TORCH_NCCL_ASYNC_ERROR_HANDLING': '1'}
    legs.rccl_env_defaults(env)
    assert env['TORCH_NCCL_ASYNC_ERROR_HANDLING'] == '1' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    env = {}
    legs.rccl_env_defaults(env)
    assert env['TORCH_NCCL_ASYNC_ERROR_HANDLING'] == '2'
