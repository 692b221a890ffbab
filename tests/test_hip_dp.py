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
