"""The N>1 path on CPU: world_size-2 gloo processes exercise the data-parallel pieces that do
not need a GPU -- the flat-buffer layout, the gradient all-reduce (sum, then 1/world), unit
sharding for the micro-benchmark -- and check them against the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from harness import models
from harness.flat import ALIGN, FlatLayout, GradSynchronizer, shard_range


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, chunks, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)                                   # identical replicas on every rank
    net = models.student()
    params = list(net.parameters())
    layout = FlatLayout([p.shape for p in params])
    flat_grad = torch.zeros(layout.total)
    for g, p in zip(layout.views(flat_grad), params):
        p.grad = g
    # each rank sees its own half of the global batch (data parallel)
    gen = torch.Generator().manual_seed(123)
    images = torch.randn(8, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 10, (8,), generator=gen)
    lo, hi = shard_range(8, rank, world)
    net.eval()                                             # BN in eval mode: the loss is a plain mean over samples
    loss = torch.nn.functional.cross_entropy(net(images[lo:hi]), labels[lo:hi])
    loss.backward()
    sync = GradSynchronizer(flat_grad, chunks=abs(chunks))
    if chunks < 0:                                         # negative = overlap mode: hooks fire during backward
        sync.attach(params, layout)
        # step 1 records the order in which gradients arrive and reduces un-overlapped; from step 2 on every
        # group is all-reduced from the hook of its last gradient, while backward is still running
        for step in range(3):
            flat_grad.zero_()
            loss = torch.nn.functional.cross_entropy(net(images[lo:hi]), labels[lo:hi])
            issued = sync.collectives_issued
            loss.backward()
            if step == 0:
                assert sync.collectives_issued == issued, 'the first backward only records the arrival order'
            else:
                assert sync.collectives_issued > issued, 'groups must be launched from inside backward'
            if step < 2:
                sync.sync()
        # the plan: out_layer (parameters()[0:2]) receives its gradient FIRST, so it belongs to the first group
        assert sync._group_of[0] == 0 and sync._group_of[1] == 0
        assert len(sync._group_ranges) == abs(chunks)
        covered = sorted(r for ranges in sync._group_ranges for r in ranges)
        assert covered[0][0] == 0 and covered[-1][1] == layout.total
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), 'ranges must tile the flat buffer'
    sync.sync()
    torch.save(flat_grad, os.path.join(out_dir, 'grad_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('chunks', [1, 3, -4])
def test_allreduced_gradient_equals_single_process(tmp_path, chunks):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), chunks, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(tmp_path / 'grad_rank0.pt')
    g1 = torch.load(tmp_path / 'grad_rank1.pt')
    assert torch.equal(g0, g1), 'ranks must end the step with identical gradients'
    # single-process reference: full batch, mean loss == average of the two half-batch means
    torch.manual_seed(0)
    net = models.student().eval()
    gen = torch.Generator().manual_seed(123)
    images = torch.randn(8, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 10, (8,), generator=gen)
    torch.nn.functional.cross_entropy(net(images), labels).backward()
    layout = FlatLayout([p.shape for p in net.parameters()])
    ref = torch.zeros(layout.total)
    for v, p in zip(layout.views(ref), net.parameters()):
        v.copy_(p.grad)
    assert torch.allclose(g0, ref, rtol=1e-4, atol=1e-6)


def test_flat_layout_alignment_and_views():
    shapes = [(10, 500), (75, 3, 5, 5), (75,), (1,), (500, 1600)]
    L = FlatLayout(shapes)
    assert all(o % ALIGN == 0 for o in L.offsets)
    flat = torch.arange(L.total, dtype=torch.float32)
    vs = L.views(flat)
    assert [tuple(v.shape) for v in vs] == shapes
    assert all(v.data_ptr() == flat.data_ptr() + 4 * o for v, o in zip(vs, L.offsets))
    vs[2].zero_()
    assert flat[L.offsets[2]:L.offsets[2] + 75].abs().sum() == 0 and flat[L.offsets[2] + 75] != 0


def test_shard_range_partitions_units():
    for total in (0, 1, 7, 8, 262144):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_kd_loss_matches_reference_formula():
    """0.7*T^2*KLDiv(element-mean) + 0.3*CE, T=2 (ref: cnn_models/help_fun.py:95,124,135-139)."""
    torch.manual_seed(1)
    zs, zt = torch.randn(6, 10), torch.randn(6, 10)
    y = torch.randint(0, 10, (6,))
    want = 0.7 * 4.0 * torch.nn.KLDivLoss()(torch.log_softmax(zs / 2, 1), torch.softmax(zt / 2, 1)) \
        + 0.3 * torch.nn.CrossEntropyLoss()(zs, y)
    assert torch.allclose(models.kd_loss(zs, zt, y), want)
