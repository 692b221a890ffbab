"""Device order statistics (qd_order_stats_f32, csrc/qd_select.hip) behind initialize_quantization_points
(ref: quantization/help_functions.py:140-154): bit-exact against a host sort."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host(values, ranks):
    return np.sort(values.cpu().numpy().reshape(-1), kind='stable')[ranks]


def _same_bits(a, b):
    """Bit equality, except that +0 and -0 are the same value: among EQUAL elements a sort's order is arbitrary (numpy's
    partition and torch.sort are not specified either), and the select orders -0 before +0."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    zero = (a == 0) & (b == 0)
    return np.array_equal(np.where(zero, np.float32(0), a).view(np.uint32), np.where(zero, np.float32(0), b).view(np.uint32))


def _cases():
    g = torch.Generator().manual_seed(11)
    yield 'gauss', torch.randn(300007, generator=g)
    yield 'unit', torch.rand(70001, generator=g)
    yield 'one', torch.tensor([3.5])
    yield 'five', torch.tensor([2.0, -1.0, 0.0, 7.0, -1.0])
    yield 'constant', torch.full((40000,), 0.25)
    yield 'two-valued', (torch.rand(50000, generator=g) > 0.3).float()
    yield 'few-levels', torch.randint(0, 7, (123457,), generator=g).float() / 6
    yield 'tiny', torch.randn(65536, generator=g) * 1e-41              # denormals
    yield 'huge', torch.randn(65536, generator=g) * 1e37
    yield 'signed-zero', torch.cat([torch.zeros(100), -torch.zeros(100), torch.randn(1000, generator=g)])
    yield 'near-ties', 0.5 + torch.randint(-3, 4, (200000,), generator=g).float() * 2.0 ** -24
    yield 'large', torch.randn(5308416 + 3, generator=g) * 0.05


@pytest.mark.parametrize('name,values', list(_cases()), ids=[c[0] for c in _cases()])
def test_order_statistics_match_a_host_sort(name, values):
    import quantization.help_functions as qhf
    dev = values.cuda()
    n = dev.numel()
    rng = np.random.default_rng(3)
    for m in (1, 2, 8, 31, 32, 33, 64):
        ranks = rng.integers(0, n, size=m)
        ranks[0] = 0
        ranks[-1] = n - 1
        got = qhf.order_statistics(dev, ranks)
        assert _same_bits(got, _host(values, ranks)), (name, m)
    # sort fallback (more than 2 x 32 distinct ranks) gives the same values
    if n > 200:
        ranks = rng.permutation(n)[:150]
        assert _same_bits(qhf.order_statistics(dev, ranks), _host(values, ranks))


def test_unaligned_start_and_repeated_ranks():
    import quantization.help_functions as qhf
    g = torch.Generator().manual_seed(5)
    base = torch.randn(100003, generator=g).cuda()
    for off in (1, 2, 3):
        view = base[off:off + 99991]
        ranks = np.array([5, 5, 0, 99990, 777, 5, 50000])
        assert _same_bits(qhf.order_statistics(view, ranks), _host(view, ranks))


def test_nan_orders_last():
    import quantization.help_functions as qhf
    v = torch.randn(10000)
    v[17] = float('nan')
    got = qhf.order_statistics(v.cuda(), np.array([0, 9998, 9999]))
    want = np.sort(v.numpy())[[0, 9998, 9999]]
    assert _same_bits(got[:2], want[:2]) and np.isnan(got[2])


def test_argument_errors():
    import quantization.help_functions as qhf
    from quantized_distillation_amd import _lib
    v = torch.randn(1000).cuda()
    with pytest.raises(IndexError):
        qhf.order_statistics(v, np.array([1000]))
    with pytest.raises(TypeError):
        qhf.order_statistics(v.double(), np.array([1]))
    lib = _lib.load()
    out = torch.empty(4, device='cuda')
    ws = torch.empty(lib.qd_order_stats_workspace_bytes(2), dtype=torch.uint8, device='cuda')
    bad = np.array([5, 3], dtype=np.int64)                               # decreasing
    assert lib.qd_order_stats_f32(v.data_ptr(), 1000, bad.ctypes.data, 2, out.data_ptr(), ws.data_ptr(), ws.numel(), 0) == -1
    ok = np.array([3, 5], dtype=np.int64)
    assert lib.qd_order_stats_f32(v.data_ptr(), 1000, ok.ctypes.data, 2, out.data_ptr(), ws.data_ptr(), 16, 0) == -2
    many = np.arange(33, dtype=np.int64)
    assert lib.qd_order_stats_f32(v.data_ptr(), 1000, many.ctypes.data, 33, out.data_ptr(), ws.data_ptr(), ws.numel(), 0) == -3


def test_initialize_quantization_points_equals_numpy_percentile():
    """The reference's own formula on the host (help_functions.py:150) against the device path, model-sized."""
    import quantization
    import quantization.help_functions as qhf
    g = torch.Generator().manual_seed(2)
    for n, k in ((800000, 4), (123457, 16), (300000, 32), (50000, 256)):
        x = (torch.randn(n, generator=g) * 0.05).cuda()
        sf = quantization.ScalingFunction('linear', False, False, 256, False)
        got = qhf.initialize_quantization_points(x, sf, k).cpu().numpy()
        scaled = sf.scale_down(x).view(-1)[0:n].cpu().numpy()
        want = np.percentile(scaled, np.linspace(0, 100, num=k)).astype(np.float32)
        assert _same_bits(got, want), (n, k)
