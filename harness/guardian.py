"""The process that owns bench.py's ONE JSON line.

Round 4's driver run lost its measurement: an optional leg brought the process down with SIGABRT (a c10d watchdog
thread's exception -> std::terminate) fifty seconds after the headline had been measured, and the line was only
printed at the very end.  No Python handler can catch that.  So the measuring is done by a WORKER process and the
line belongs to this one, which imports neither torch nor HIP and therefore cannot be taken down by them:

    guardian (python, stdlib only)  --spawns-->  worker (torch, HIP, RCCL; `bench.py --worker`)
         ^                                           |
         +------- snapshots over a pipe -------------+   one JSON object per line: {"line": {...}, "progress": {...}}

The worker sends a snapshot of its record whenever a leg finishes (and a progress marker before a leg starts).  The
guardian keeps the last complete one and, when the worker ends -- however it ends: exit 0, an exception, a signal, the
wall limit -- writes the full record to `detail_path` and to stderr and prints its COMPACT form (harness/report.py:
at most 4096 bytes) exactly once, as the only thing on stdout.  If the worker died inside a leg and this is a
one-rank run, the guardian starts a fresh worker for the legs that are left (the dead leg is recorded as such and not
retried), so one bad leg costs that leg only.  At N > 1 the ranks cannot be restarted one by one; the line then holds
what rank 0 had measured.

Exit code: 0 when the printed line has a headline value, 1 otherwise.  Ranks other than 0 print nothing and exit 0
whenever their worker has ended (a non-zero exit would make torch.distributed.run tear rank 0 down before it prints).
Device-agnostic and torch-free: tests/test_guardian.py drives it on CPU with workers that abort on purpose.
"""
import json
import os
import select
import signal
import subprocess
import sys
import tempfile
import time

from . import report

REPORT_FD_ENV = 'QD_BENCH_REPORT_FD'


class Reporter(object):
    """Worker side: sends snapshots to the guardian (or, when run without one, keeps the last line for a plain print)."""

    def __init__(self):
        fd = os.environ.get(REPORT_FD_ENV)
        self.fd = int(fd) if fd else None
        self.last = None

    def send(self, line, done=(), running=None, wall=None):
        self.last = line
        if self.fd is None:
            return
        msg = json.dumps({'line': line, 'progress': {'done': list(done), 'running': running, 'wall_s': wall}}) + '\n'
        data = msg.encode()
        try:
            while data:
                n = os.write(self.fd, data)
                data = data[n:]
        except OSError:                      # the guardian is gone (EPIPE): keep measuring, the worker prints the line itself at the end
            self.fd = None


def signal_name(rc):
    if rc is not None and rc < 0:
        try:
            return signal.Signals(-rc).name
        except ValueError:
            return 'signal %d' % -rc
    return None


def _read_snapshots(proc, rfd, state, wall_limit_s, t0, log):
    """Pump the pipe until the worker has ended and the pipe is drained; returns 'limit' if the wall limit was reached."""
    buf, eof = b'', False
    while True:
        left = wall_limit_s - (time.time() - t0)
        if left <= 0:
            return 'limit'
        if not eof:
            r, _, _ = select.select([rfd], [], [], min(left, 0.25))
            if r:
                chunk = os.read(rfd, 1 << 16)
                if chunk:
                    buf += chunk
                    while b'\n' in buf:
                        raw, buf = buf.split(b'\n', 1)
                        try:
                            snap = json.loads(raw.decode())
                        except ValueError:
                            log('guardian: dropped an unparsable snapshot (%d bytes)' % len(raw))
                            continue
                        state['line'] = snap.get('line') or state.get('line')
                        state['progress'] = snap.get('progress') or {}
                    continue
                eof = True                        # every write end is closed
        if proc.poll() is not None:
            if eof:
                return None
            r, _, _ = select.select([rfd], [], [], 0)     # the worker is gone: drain what it left, then stop
            if not r:
                return None
        elif eof:
            time.sleep(0.1)                       # the worker closed the pipe but is still running: wait for it


def _die_with_parent():
    """preexec of the worker: have the kernel send it SIGKILL when the guardian dies (prctl PR_SET_PDEATHSIG).  The worker runs
    in its own session (so that the guardian can end the whole group: rocprofv3 children, data-loader processes); without this
    a guardian that is SIGKILLed -- a driver's hard timeout -- would leave a worker behind that holds the GPU."""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)        # 1 = PR_SET_PDEATHSIG
    except Exception:                                        # noqa: BLE001 -- not Linux / no libc: nothing to set
        pass


def _kill_group(proc, log):
    for sig in (signal.SIGTERM, signal.SIGKILL):
        if proc.poll() is not None:
            return
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            return
        try:
            proc.wait(5)
        except subprocess.TimeoutExpired:
            log('guardian: worker group did not end on %s' % sig.name)


def supervise(worker_cmd, all_legs, rank=0, world=1, wall_limit_s=1500.0, max_restarts=2, out=None, log=None, env=None,
              detail_path=None, compact=report.compact):
    """Run `worker_cmd(extra_args) -> argv` under supervision; print the line (rank 0) and return the exit code.

    Rank 0 writes the full record to `detail_path` (if given) and to stderr, and prints report.fit(compact(record)) on `out`.

    worker_cmd(extra) must return the argv of a worker that understands
        --resume FILE     JSON {"line": ..., "done": [...], "dead": {leg: reason}}: continue from there
    and reports through the file descriptor named by $QD_BENCH_REPORT_FD.  `all_legs` is the ordered list of leg names
    the worker runs by default (what "the legs that are left" is computed from)."""
    out = sys.stdout if out is None else out
    log = log if log is not None else (lambda s: (sys.stderr.write(s + '\n'), sys.stderr.flush()))
    t0 = time.time()
    state = {'line': None, 'progress': {}}
    dead, runs, restarts = {}, [], 0
    current = {'proc': None}
    printed = {'done': False}

    def finish(code_hint=None):
        if printed['done']:
            return
        printed['done'] = True
        line = state['line']
        info = {'workers': runs, 'restarts': restarts, 'wall_s': round(time.time() - t0, 1)}
        if dead:
            info['legs_lost_with_their_worker'] = dead
        if rank == 0:
            if line is None:
                line = {'metric': 'quantize_dequantize_GBps_64M_fp32_4bit', 'value': None, 'unit': 'GB/s', 'n_gpus': world}
            line = dict(line)
            if line.get('value') is None and 'error' not in line:
                line['error'] = 'the worker ended before the headline was measured (%s); see stderr' % (code_hint,)
            line['bench_process'] = info
            if detail_path:
                line['detail'] = os.path.basename(detail_path)
                if not report.write_detail(detail_path, line, log):
                    del line['detail']
            log('bench record (full): ' + json.dumps(line))
            out.write(report.fit(compact(line)) + '\n')
            out.flush()

    def on_signal(signum, _frame):
        log('guardian: %s -- printing what has been measured and ending the worker' % signal.Signals(signum).name)
        if current['proc'] is not None:
            _kill_group(current['proc'], log)
        runs.append({'exit': 'guardian received %s' % signal.Signals(signum).name})
        finish()
        os._exit(0 if (state['line'] or {}).get('value') else 1)

    old = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    try:
        extra = []
        resume_files = []
        while True:
            rfd, wfd = os.pipe()
            e = dict(os.environ if env is None else env)
            e[REPORT_FD_ENV] = str(wfd)
            # the worker's stdout is this process's stderr: RCCL banners, MIOpen chatter and stray prints can never
            # reach the stdout that carries the line
            proc = subprocess.Popen(worker_cmd(extra), env=e, pass_fds=(wfd,), stdout=sys.stderr.fileno(), start_new_session=True,
                                    preexec_fn=_die_with_parent)
            os.close(wfd)
            current['proc'] = proc
            why = _read_snapshots(proc, rfd, state, wall_limit_s, t0, log)
            os.close(rfd)
            if why == 'limit':
                log('guardian: wall limit of %.0f s reached, ending the worker' % wall_limit_s)
                _kill_group(proc, log)
                runs.append({'exit': 'killed at the wall limit of %.0f s' % wall_limit_s, 'during': state['progress'].get('running')})
                if state['line'] is not None:
                    state['line'] = dict(state['line'], error='wall limit of %.0f s reached: the line holds what had been measured by then'
                                                             % wall_limit_s)
                break
            rc = proc.wait()
            current['proc'] = None
            prog = state['progress']
            run = {'exit': signal_name(rc) or rc, 'legs_done': len(prog.get('done') or [])}
            runs.append(run)
            if rc == 0:
                break
            leg = prog.get('running')
            run['during'] = leg
            log('guardian: worker ended with %s during leg %r' % (run['exit'], leg))
            left = [x for x in all_legs if x not in (prog.get('done') or []) and x != leg and x not in dead]
            if leg is not None:
                dead[leg] = 'the worker process died with %s inside this leg' % (run['exit'],)
            if world != 1 or (state['line'] or {}).get('value') is None or leg is None or not left or restarts >= max_restarts:
                break
            restarts += 1
            f = tempfile.NamedTemporaryFile('w', suffix='.json', prefix='qd_bench_resume_', delete=False)
            json.dump({'line': state['line'], 'done': list(prog.get('done') or []) + [leg], 'dead': dead}, f)
            f.close()
            resume_files.append(f.name)
            extra = ['--resume', f.name]
            log('guardian: starting a fresh worker for the remaining legs: %s' % ', '.join(left))
        for f in resume_files:
            try:
                os.unlink(f)
            except OSError:
                pass
        finish(runs[-1]['exit'] if runs else None)
    finally:
        for s, h in old.items():
            if h is not None:                # None: the previous handler was not installed from Python (a profiler's, a launcher's
                signal.signal(s, h)          # C-level one -- rocprofv3 does that): nothing Python could put back, and nothing to undo
    if rank != 0:
        return 0
    return 0 if (state['line'] or {}).get('value') else 1
