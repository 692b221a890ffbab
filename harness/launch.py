"""One process per GPU: starts N ranks of a script under torch.distributed.run on this node.

`python bench.py --gpus N` without a torchrun environment calls run_ranks(): the script is
re-executed as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port <free port> <script> <the same arguments>

so that every rank gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and initialises RCCL (backend
"nccl" on ROCm) or gloo itself.  Device-agnostic: the CPU tests start gloo ranks through it.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def under_launcher(env=None):
    """True inside a rank started by torch.distributed.run (or any launcher that exports the
    rendezvous variables)."""
    env = os.environ if env is None else env
    return 'WORLD_SIZE' in env and 'RANK' in env


def launcher_command(script, nranks, argv, port=None):
    port = free_port() if port is None else port
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(nranks)),
            '--master-addr', '127.0.0.1', '--master-port', str(port), script] + list(argv)


def run_ranks(script, nranks, argv, env=None, timeout=None, capture=False):
    """Run `script argv...` as `nranks` ranks on this node; returns the launcher's exit code
    (with capture=True: (exit code, stdout text)).  stdout/stderr are inherited otherwise, so
    rank 0's single JSON line stays the last line of this process's stdout."""
    e = dict(os.environ if env is None else env)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # the host driver only supports dmabuf IPC (RCCL needs it)
    e.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // max(1, int(nranks)))))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    cmd = launcher_command(script, nranks, argv)
    if capture:
        p = subprocess.run(cmd, env=e, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return p.returncode, p.stdout
    return subprocess.run(cmd, env=e, timeout=timeout).returncode
