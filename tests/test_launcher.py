"""`python bench.py --gpus N` must start N ranks itself when no launcher environment is present
(the driver invokes it that way).  The launcher is device-agnostic, so it is tested here on CPU
with gloo ranks; bench.py's own use of it is checked structurally (no GPU in this container)."""
import json
import os
import sys

import pytest

from harness import launch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize('n', [2, 3])
def test_run_ranks_starts_n_ranks(n):
    rc, out = launch.run_ranks(os.path.join(HERE, 'rank_worker.py'), n, ['--gpus', str(n), '--steps', '5'],
                               timeout=300, capture=True)
    assert rc == 0, out
    line = [l for l in out.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['world'] == n and d['env_world'] == n
    assert d['sum'] == n * (n + 1) / 2                     # every rank took part in the all-reduce
    assert d['argv'] == ['--gpus', str(n), '--steps', '5']  # arguments reach the ranks unchanged
    assert d['master'] == '127.0.0.1'


def test_under_launcher_detection():
    assert not launch.under_launcher({})
    assert not launch.under_launcher({'WORLD_SIZE': '2'})
    assert launch.under_launcher({'WORLD_SIZE': '2', 'RANK': '0'})


def test_launcher_command_shape():
    cmd = launch.launcher_command('/x/bench.py', 8, ['--gpus', '8'], port=29511)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-3:] == ['/x/bench.py', '--gpus', '8']


def test_bench_relaunches_itself_for_n_gt_1():
    """bench.py reads --gpus, and with N > 1 outside a launcher hands over to launch.run_ranks
    BEFORE anything touches the GPU (the guardian process never does; the device count is asked of a child);
    the launched ranks then report the world size they were given."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    g = src[src.index('def guardian_main('):src.index('def worker_main(')]
    i_launch = g.index('launch.run_ranks(os.path.abspath(__file__), args.gpus, argv)')
    assert 'args.gpus > 1 and not launch.under_launcher()' in g[:i_launch]
    assert i_launch < g.index('guardian.supervise(')
    assert 'import torch' not in g.replace("'import torch; print(", '')         # only inside the child's -c string
    w = src[src.index('def worker_main('):]
    assert "'n_gpus': n_gpus" in w and "group['world'] = dist.get_world_size()" in w


def test_bench_without_gpu_with_n_gt_1_fails_loudly():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert 'HIP device' in p.stderr
