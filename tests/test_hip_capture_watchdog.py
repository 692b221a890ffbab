"""hipGraph capture next to a live RCCL process group (harness/distill.py: capture_into, quiesce_collectives).

Round 4's driver-run bench was lost here: the c10d watchdog thread polled a collective's completion event while a
GLOBAL-mode capture was open on the main thread -> "operation not permitted when stream is capturing" on the watchdog ->
std::terminate -> SIGABRT.  The product now captures in thread-local mode, after waiting for its collectives and giving
the watchdog time to retire them, and restores the caller's stream if a capture fails.  Each case runs in a child process
(tests/capture_worker.py): the failure mode is a process abort.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_worker(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return subprocess.run([sys.executable, os.path.join(HERE, 'capture_worker.py')] + list(argv), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_capture_right_after_20_allreduce_steps_30_times_over():
    """The driver's situation, x 30: one-rank NCCL group, forced collectives, 20 eager steps, capture immediately."""
    p = run_worker('--iters', '30', '--steps', '20')
    assert p.returncode == 0, 'rc %s\n%s' % (p.returncode, p.stderr[-3000:])
    assert 'CAPTURE_OK 30' in p.stdout


def test_thread_local_capture_holds_with_collectives_in_flight_and_no_settling():
    """The capture mode on its own: no settling time, and 8 un-waited all-reduces issued right before every capture, so the
    watchdog is polling completion events while the capture is open."""
    p = run_worker('--iters', '20', '--steps', '20', '--settle', '0', '--inflight', '8')
    assert p.returncode == 0, 'rc %s\n%s' % (p.returncode, p.stderr[-3000:])
    assert 'CAPTURE_OK 20' in p.stdout


def test_long_thread_local_capture_with_an_incomplete_allreduce_listed_by_the_watchdog():
    """Deterministic form: ONE capture held open for 0.5 s while an all-reduce is kept incomplete behind a spin kernel, so the
    watchdog (100 ms period) queries that work item's event several times while the capture is open.  Thread-local mode
    (what capture_into() uses): passes."""
    p = run_worker('--long-capture', '0.5', '--mode', 'thread_local')
    assert p.returncode == 0, 'rc %s\n%s' % (p.returncode, p.stderr[-3000:])
    assert 'LONG_CAPTURE_OK' in p.stdout


def test_the_same_capture_in_global_mode_is_round_4s_crash():
    """... and in torch's default 'global' mode the same situation IS the crash that cost round 4 its driver measurement: the
    watchdog thread's event query is refused ("operation not permitted when stream is capturing"), its exception ends the
    process with SIGABRT (4 of 4 runs on four boxes in round 5).  Kept as a test so that the hazard stays documented by something
    that runs; on a stack where global mode survives it skips, saying so."""
    p = run_worker('--long-capture', '0.5', '--mode', 'global')
    if p.returncode == 0 and 'LONG_CAPTURE_OK' in p.stdout:
        pytest.skip('a global-mode capture survived the watchdog on this stack: the hazard this documents is gone here')
    assert p.returncode != 0 and 'LONG_CAPTURE_OK' not in p.stdout
    assert 'stream is capturing' in p.stderr and 'watchdog' in p.stderr, p.stderr[-2000:]


@pytest.mark.parametrize('how', ['raise', 'sync', 'item'])
def test_a_failed_capture_restores_the_stream_and_leaves_the_process_usable(how):
    """capture_into(): when the captured body fails (a Python exception; a device synchronize or a blocking copy, which
    invalidate the capture so that capture_end() raises too), the stream that was current before is current again, eager
    launches work, the trainer has not switched to replay and still steps eagerly (further capture attempts on it are refused:
    it is to be discarded), and fresh captures -- a simple graph, a new trainer -- work in the same process."""
    p = run_worker('--failed-capture', how)
    assert p.returncode == 0, 'rc %s\n%s\n%s' % (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    assert 'FAILED_CAPTURE_OK' in p.stdout
