"""Data-parallel distillation step with the REAL quantizer kernels: two ranks (both on cuda:0 --
the test box has one GPU -- exchanging gradients over gloo) must hold bit-identical master
weights after every step, and the all-reduced gradient must be the mean of the two local ones."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, overlap):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from harness import models
    from harness.distill import DistillTrainer, synthetic_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)                                     # identical replicas
    tr = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi',
                        grad_chunks=3 if overlap else 1, overlap_allreduce=overlap)
    assert tr.sync.world == 2
    for step in range(3):
        x, y = synthetic_batch(8, dev, seed=100 * rank + step)       # each rank its own shard of the batch
        tr.quantize()
        tr.forward_backward(x, y)
        if overlap:
            # the pieces are already being reduced from the backward hooks: only the end state is observable
            tr.sync.sync()
        else:
            local = tr.flat_grad.clone()
            tr.sync.sync()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            want = (gathered[0] + gathered[1]) / 2
            assert torch.allclose(tr.flat_grad, want, rtol=1e-6, atol=1e-8), 'all-reduced gradient != mean of local gradients'
        both = [torch.zeros_like(tr.flat_grad) for _ in range(world)]
        dist.all_gather(both, tr.flat_grad)
        assert torch.equal(both[0], both[1]), 'ranks disagree on the reduced gradient'
        tr.opt.step()
        torch.save(tr.flat_master.cpu(), os.path.join(out_dir, 'm_r%d_s%d.pt' % (rank, step)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_two_rank_distill_step_keeps_replicas_identical(tmp_path, overlap):
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), overlap), nprocs=2, join=True)
    for step in range(3):
        a = torch.load(tmp_path / ('m_r0_s%d.pt' % step))
        b = torch.load(tmp_path / ('m_r1_s%d.pt' % step))
        assert torch.equal(a, b), 'replicas diverged at step %d' % step
    assert not torch.equal(torch.load(tmp_path / 'm_r0_s0.pt'), torch.load(tmp_path / 'm_r0_s2.pt'))


# ------------------------------------------------------------------ the RCCL call path itself (backend "nccl")
def _rccl_worker(rank, world, port, out_dir):
    """Single-rank RCCL group on cuda:0: the collectives of GradSynchronizer (plain, chunked, hook-overlapped), the
    setup broadcast and a DistillTrainer / DiffQuantTrainer step are issued through RCCL (QD_FORCE_DIST=1 removes
    the world == 1 short-circuit), so the code path the multi-GPU bench runs is executed on a one-GPU box."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', QD_FORCE_DIST='1',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    from harness import models
    from harness.diffquant import DiffQuantTrainer
    from harness.distill import DistillTrainer, synthetic_batch
    from harness.flat import GradSynchronizer
    report = {}
    # 1. bare synchroniser: sum over one rank leaves the buffer bit-identical, and the collectives were issued
    for chunks in (1, 5):
        g = torch.randn(1 << 20, device=dev)
        before = g.clone()
        s = GradSynchronizer(g, chunks=chunks)
        assert s.active and s.world == 1
        s.sync()
        torch.cuda.synchronize()
        assert torch.equal(g, before)
        assert s.collectives_issued == chunks
        report['plain_chunks_%d' % chunks] = s.collectives_issued
    # 2. the distillation step with the hook-overlapped reduction (4 groups) == the same step without any group
    torch.manual_seed(0)
    a = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi',
                       grad_chunks=4, overlap_allreduce=True)
    assert a.sync.active
    os.environ['QD_FORCE_DIST'] = '0'
    torch.manual_seed(0)
    b = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi')
    assert not b.sync.active
    os.environ['QD_FORCE_DIST'] = '1'
    torch.backends.cudnn.deterministic = True
    for step in range(4):
        x, y = synthetic_batch(16, dev, seed=step)
        issued = a.sync.collectives_issued
        a.quantize(); a.forward_backward(x, y)
        if step > 0:
            assert a.sync.collectives_issued > issued, 'groups must be launched from the backward hooks'
        a.sync.sync(); a.opt.step()
        b.step(x, y)
    torch.cuda.synchronize()
    assert torch.allclose(a.flat_master, b.flat_master, rtol=1e-4, atol=1e-6)
    assert a.sync._group_of[0] == 0, 'out_layer (parameters()[0]) gets its gradient first: it must sit in the first group'
    report['overlap_collectives'] = a.sync.collectives_issued
    # 3. differentiable quantization exchanges only the point gradients
    torch.manual_seed(0)
    d = DiffQuantTrainer(models.student(), dev, num_points=4, bucket_size=256, mode='multi')
    assert d.exchange
    x, y = synthetic_batch(8, dev, seed=3)
    assert torch.isfinite(d.step(x, y))
    torch.cuda.synchronize()
    dist.barrier()
    torch.save(report, os.path.join(out_dir, 'report.pt'))
    dist.destroy_process_group()


def test_rccl_single_rank_executes_the_collective_path(tmp_path):
    assert torch.cuda.is_available()
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    rep = torch.load(tmp_path / 'report.pt')
    assert rep['plain_chunks_1'] == 1 and rep['plain_chunks_5'] == 5 and rep['overlap_collectives'] >= 4
