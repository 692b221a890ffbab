"""Failure containment for a benchmark made of several legs that hold collectives.

bench.py runs the headline measurement and then a handful of steps/sec legs, each with its own
RCCL all-reduces.  Two things can eat the driver's time budget on the first 8-GPU run:

  * one rank raises inside a leg (an out-of-memory condition, a shape that one rank sees and
    the others do not) while the others sit inside that leg's next collective;
  * everything is fine on every rank, but one leg hangs.

LegRunner bounds both.  Every leg ends with an agreement -- "did this rank finish the leg?",
one int per rank, summed over a SEPARATE gloo (CPU, TCP) group, so that it works whatever state
the data-path communicator is in -- and a rank that raised still enters that agreement (the
leg BODY is wrapped, not the call).  The ranks that were left inside a collective leave it
when the data group's timeout expires (`init_process_group(timeout=...)`; for RCCL with
TORCH_NCCL_ASYNC_ERROR_HANDLING=2 the watchdog aborts the communicator and the blocked call
raises instead of tearing the process down), raise, and enter the same agreement.  Once a
rank has failed INSIDE a leg, the data-path communicator may hold unmatched collectives, so
every later collective-bearing leg is skipped (recorded as such) and the final JSON line is
still printed.  The backstop for the second case is the wall limit of the process that owns the
line (harness/guardian.py: it ends the worker and prints what had been measured); `Deadline`
below is the in-process form of the same idea, kept for scripts without a guardian.

What the agreement guarantees whatever RCCL's watchdog does (mode 2 "CleanUpOnly" aborts the
communicator; whether the blocked wait() then RAISES or merely returns is not something a
one-GPU box can show): a leg's result is reported only if EVERY rank finished the leg.  A rank
whose aborted collective returned without raising may compute on garbage, but the rank that
caused the abort has said "failed" in the agreement, so every rank -- that one included --
replaces its result with the error record.  Untested on RCCL with N > 1 (no such box yet).

Device-agnostic: tests/test_legs_gloo.py runs two gloo ranks on CPU, one of which raises inside a
leg, and checks that both leave within seconds with the error in rank 0's JSON line.
"""
import datetime
import os
import sys
import threading
import traceback

import torch
import torch.distributed as dist

DATA_TIMEOUT_S = 120           # a collective of the data-path group that nobody joins gives up after this long
CTL_TIMEOUT_S = 420            # the agreement waits for the ranks that are waiting for THAT timeout, and then some


def rccl_env_defaults(env=None):
    """Environment a multi-rank RCCL run needs before init_process_group: dmabuf IPC, and a watchdog that aborts the
    communicator of a timed-out collective and lets the blocked call RAISE (mode 2, "CleanUpOnly") instead of tearing the
    process down (the default), so that the rank reaches the agreement and rank 0 its JSON line."""
    env = os.environ if env is None else env
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '2')
    return env


def miopen_env_defaults(env=None):
    """MIOpen's FAST find mode for the steps/sec legs (before the first convolution: MIOpen reads the variable once).  The
    default hybrid mode benchmarks every applicable solver the first time it sees a shape its find-db does not hold: 45 s for
    the WideResNet-16-22 shapes of configs[2] on a fresh box, inside the one command the driver times, on every rank.  FAST
    takes the find-db entry when there is one and the immediate-mode heuristic otherwise: 2.9 s, at 8.0 instead of 8.4
    steps/s on configs[2] and within 1 % on configs[3] (profiles/r06_miopen_find_mode.txt).  The convolutions are the
    callers' work, not the path measured here; the quantizer phases of those legs are timed on their own and do not change.
    Returns the mode in force (a value exported by the user wins)."""
    env = os.environ if env is None else env
    env.setdefault('MIOPEN_FIND_MODE', '2')
    return env['MIOPEN_FIND_MODE']


def data_timeout(seconds=None):
    """Timeout of the data-path group's collectives (QD_BENCH_DATA_TIMEOUT_S overrides the default: the tests use a short one)."""
    if seconds is None:
        seconds = float(os.environ.get('QD_BENCH_DATA_TIMEOUT_S', DATA_TIMEOUT_S))
    return datetime.timedelta(seconds=seconds)


class LegRunner(object):
    def __init__(self, ctl_timeout_s=CTL_TIMEOUT_S, log=None):
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if ready else 1
        self.rank = dist.get_rank() if ready else 0
        self.ctl = None
        if ready and self.world > 1:
            self.ctl = dist.new_group(backend='gloo', timeout=datetime.timedelta(seconds=ctl_timeout_s))
        self.broken = None                 # why the data-path communicator is no longer trusted
        self.history = []                  # [(leg, failed ranks)]
        self.log = log if log is not None else (lambda s: sys.stderr.write(s + '\n'))

    # ------------------------------------------------------------------ the agreement
    def agree(self, ok):
        """Every rank says whether it is fine; returns the sorted list of ranks that are not.  Goes over the gloo control
        group only.  If the control group itself fails (a rank died), every other rank counts as failed."""
        if self.ctl is None:
            return [] if ok else [self.rank]
        t = torch.zeros(self.world, dtype=torch.int32)
        t[self.rank] = 0 if ok else 1
        try:
            dist.all_reduce(t, group=self.ctl)
        except Exception as e:                                     # noqa: BLE001 -- a dead rank: nobody else can be trusted
            self.log('legs: the control group failed (%s: %s)' % (type(e).__name__, e))
            self.ctl = None
            self.broken = self.broken or 'the control group failed: a rank died'
            return [r for r in range(self.world) if r != self.rank or not ok]
        return [r for r in range(self.world) if int(t[r])]

    def rank0_says(self, flag):
        """Rank 0's yes/no, known to every rank (over the control group): decisions that depend on a rank's own clock -- "does
        this optional leg still fit the wall budget?" -- must come out the same everywhere, or one rank would skip a leg
        whose collectives the others enter."""
        if self.ctl is None:
            return bool(flag)
        t = torch.tensor([1 if (flag and self.rank == 0) else 0], dtype=torch.int32)
        try:
            dist.all_reduce(t, group=self.ctl)
        except Exception as e:                                     # noqa: BLE001
            self.log('legs: the control group failed (%s: %s)' % (type(e).__name__, e))
            self.ctl = None
            self.broken = self.broken or 'the control group failed: a rank died'
            return False
        return bool(int(t[0]))

    def barrier(self, ok=True):
        """A barrier that does not touch the data-path communicator; it carries one bit per rank (returns the ranks that
        passed ok=False), so that an in-leg wait can tell the others "do not enter the next collective"."""
        return self.agree(ok)

    # ------------------------------------------------------------------ one leg
    def run(self, name, fn, *args, collective=True, **kwargs):
        """fn(*args, **kwargs) on every rank; its result, or {'error': ...} on EVERY rank if any rank failed, or
        {'skipped': ...} if an earlier failure left the data-path communicator in an unknown state."""
        if collective and self.broken:
            return {'skipped': 'not run: %s' % self.broken}
        err, res = None, None
        try:
            res = fn(*args, **kwargs)
        except Exception as e:                                     # noqa: BLE001 -- recorded in the JSON line
            self.log(traceback.format_exc())
            err = '%s: %s' % (type(e).__name__, e)
        failed = self.agree(err is None)
        if not failed:
            return res
        self.history.append((name, failed))
        if collective and self.world > 1:
            self.broken = "leg '%s' failed on rank(s) %s: the data-path communicator may hold unmatched collectives" % (
                name, ','.join(str(r) for r in failed))
        out = {'error': err if err is not None else "rank(s) %s failed in this leg (this rank finished it)" % failed,
               'failed_ranks': failed}
        return out


class Deadline(object):
    """After `seconds`, call `on_expire()` (rank 0: print the partial JSON line) and end the process with `exit_code`.
    cancel() when the run finishes in time."""

    def __init__(self, seconds, on_expire, exit_code=3):
        self._done = threading.Event()
        self.seconds = seconds

        def watch():
            if self._done.wait(seconds):
                return
            try:
                on_expire()
            finally:
                os._exit(exit_code)
        self._thread = threading.Thread(target=watch, name='bench-deadline', daemon=True)
        self._thread.start()

    def cancel(self):
        self._done.set()
