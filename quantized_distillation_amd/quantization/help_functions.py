"""Host-side helpers of the quantizer (mirror of the reference's quantization/help_functions.py,
cited as `ref:`).  These run off the per-step path: bucket reshaping for callers that want the
view, the percentile initialisation of the quantization points, the bit-allocation heuristic
and the Huffman size accounting.  The per-tensor work they need (scaling, assignment) goes
through the HIP kernels (libqd_host.so for CPU tensors); what remains on the host is what the
reference also does on the host (np.percentile, a heap, a dict of frequencies).
"""
import math  # noqa: F401  (kept for API parity with the reference module namespace)
from collections import defaultdict
from heapq import heapify, heappop, heappush

import numpy as np
import torch


# ---- hyperspherical-coordinate helpers (ref: help_functions.py:8-64) -------------------------
# Not used by any quantization path of the reference (nothing outside this group calls them); kept so
# that the module surface is complete.  Plain torch ops on whatever device the input lives on.
def invert_pytorch_vector(pytorch_vector):
    """The 1-D vector reversed (ref: :8-14)."""
    return torch.flip(pytorch_vector, dims=[0])


def findFirstNonZeroIndex(pytorch_vector):
    """Index of the first non-zero entry, -1 if there is none (ref: :16-24)."""
    nz = torch.nonzero(pytorch_vector.reshape(-1) != 0)
    return int(nz[0]) if nz.numel() else -1


def cart2hyperspherical(cartesianCoordinates):
    """(radius, angles[n-1]) of an n-vector (ref: :27-52).  With tail[i] = sqrt(sum_{j>=i} x_j^2):
    angle_i = acos(x_i / tail[i]), 0 where the whole tail is zero, and the last angle is reflected
    to (pi, 2 pi) when the last coordinate is negative."""
    x = cartesianCoordinates
    n = x.size(0)
    tail = torch.flip(torch.sqrt(torch.flip(x ** 2, dims=[0]).cumsum(dim=0)), dims=[0])     # tail[i], i = 0..n-1
    radius = tail[0]
    head, norm = x[:n - 1], tail[:n - 1]
    live = norm != 0
    angles = torch.where(live, torch.acos(head / torch.where(live, norm, torch.ones_like(norm))), torch.zeros_like(head))
    if n > 1 and bool(x[-1] < 0):
        angles[-1] = 2 * math.pi - angles[-1]
    return radius, angles


def hypershperical2cart(sphericalCoordinates):
    """Inverse of cart2hyperspherical (ref: :54-64; the reference's spelling of the name is kept)."""
    radius, angles = sphericalCoordinates[0], sphericalCoordinates[1]
    one = torch.ones(1, dtype=angles.dtype, device=angles.device)
    cos = torch.cat((torch.cos(angles), one))
    sin_prod = torch.cat((one, torch.sin(angles).cumprod(dim=0)))
    return radius * sin_prod * cos


def create_bucket_tensor(tensor, bucket_size, fill_values='last'):
    """View `tensor` as rows of `bucket_size` elements, padding a ragged tail with copies of the
    last element (or NaN).  ref: help_functions.py:67-94.  The kernels never materialise this
    view (padding cannot change a bucket's min/max); it is provided for API compatibility."""
    if bucket_size is None:
        return tensor
    flat = tensor.reshape(-1)
    n = flat.numel()
    if fill_values == 'nan':
        fill = float('nan')
    elif fill_values == 'last':
        fill = flat[-1]
    else:
        fill = fill_values
    full, rest = divmod(n, bucket_size)
    if full == 0:
        return flat.view(1, n)
    if rest != 0:
        pad = torch.empty(bucket_size - rest, dtype=flat.dtype, device=flat.device)
        pad.fill_(fill) if not isinstance(fill, torch.Tensor) else pad.copy_(fill.expand_as(pad))
        flat = torch.cat([flat, pad])
    return flat.view(-1, bucket_size)


def assign_bits_automatically(gradient_norms, inital_bits_to_assign, input_is_point=False):
    """Redistribute a bit (or point) budget across tensors proportionally to their gradient
    norms, keeping the total fixed.  ref: help_functions.py:97-138.  Pure host arithmetic."""
    if isinstance(inital_bits_to_assign, int):
        inital_bits_to_assign = [inital_bits_to_assign] * len(gradient_norms)
    if len(inital_bits_to_assign) != len(gradient_norms):
        raise ValueError('There should be as many gradients as there are initial points.')
    gradient_norms = [float(g) for g in gradient_norms]     # accepts 0-dim device tensors
    budget = sum(inital_bits_to_assign)
    floor_alloc = [(b // 2) if input_is_point else (b - 1) for b in inital_bits_to_assign]
    spare = budget - sum(floor_alloc)
    norm_total = sum(gradient_norms)
    alloc = [base + round(g / norm_total * spare) for g, base in zip(gradient_norms, floor_alloc)]
    excess = sum(alloc) - budget
    if excess > 0:
        alloc[alloc.index(max(alloc))] -= excess
    elif excess < 0:
        alloc[alloc.index(min(alloc))] += -excess
    return alloc


def percentile_ranks(n, num_points):
    """The two neighbouring 0-based ranks and the interpolation weight of each of the `num_points` percentiles
    np.linspace(0, 100, num_points) of n values, with numpy's arithmetic (method 'linear': virtual index
    (n-1)*q with q = p/float32(100))."""
    quant = np.true_divide(np.linspace(0, 100, num=num_points), np.float32(100))
    virtual = (n - 1) * quant
    lower = np.floor(virtual)
    upper = lower + 1
    at_end = virtual >= n - 1
    lower[at_end] = n - 1
    upper[at_end] = n - 1
    gamma = virtual - np.floor(virtual)
    return lower.astype(np.int64), upper.astype(np.int64), gamma


def percentile_lerp(a, b, gamma):
    """numpy's _lerp on float32 neighbours: float32 difference, float64 lerp, the t >= 0.5 branch taken from b."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    diff = np.subtract(b, a)
    result = np.add(a, diff * gamma)
    np.subtract(b, diff * (1 - gamma), out=result, where=gamma >= 0.5)
    return result


def percentile_points_from_sorted(fetch, n, num_points):
    """np.percentile(a, np.linspace(0, 100, num_points)) (method 'linear', float32 data) given only
    random access to the SORTED data: `fetch(int64 index array) -> float32 values`.  Reproduces
    numpy's arithmetic exactly, so the result is bit-identical to the reference's host-side call
    (ref: help_functions.py:150) while only 2*num_points values ever leave the device."""
    lower, upper, gamma = percentile_ranks(n, num_points)
    return percentile_lerp(fetch(lower), fetch(upper), gamma)


ORDER_STATS_MAX_PASSES = 2       # above 2 * QD_ORDER_STATS_MAX_RANKS distinct ranks the device sort is cheaper than selecting


def order_statistics(values, ranks):
    """values[...] sorted ascending, at the (numpy int64, any order, repeats allowed) 0-based `ranks` -- as a float32
    numpy array, without sorting: a 12|10|10-bit radix select on the device (qd_order_stats_f32, csrc/qd_select.hip)
    reads `values` three times per 32 distinct ranks and only the selected values leave the device.  More than
    2 * 32 distinct ranks (num_points > 32): one device sort and a gather instead."""
    from .. import _lib
    flat = values.reshape(-1)
    if flat.dtype != torch.float32:
        raise TypeError('order_statistics needs a float32 tensor, got %s on %s' % (flat.dtype, flat.device))
    ranks = np.asarray(ranks, dtype=np.int64)
    n = flat.numel()
    if ranks.size and (ranks.min() < 0 or ranks.max() >= n):
        raise IndexError('rank out of range for %d elements' % n)
    if not flat.is_cuda:                   # a CPU tensor: the selection on the host (np.partition places exactly the asked ranks)
        wanted = np.unique(ranks)
        return np.partition(flat.numpy(), wanted)[ranks] if wanted.size else np.empty(ranks.shape, dtype=np.float32)
    distinct, inverse = np.unique(ranks, return_inverse=True)
    lib = _lib.load()
    chunk = 32                                                          # QD_ORDER_STATS_MAX_RANKS
    if distinct.size > ORDER_STATS_MAX_PASSES * chunk or n >= 2 ** 32:
        ordered = torch.sort(flat)[0]
        return ordered[torch.from_numpy(ranks).to(flat.device)].cpu().numpy()
    flat = flat.contiguous()
    out = torch.empty(distinct.size, dtype=torch.float32, device=flat.device)
    with torch.cuda.device(flat.device):
        ws_bytes = lib.qd_order_stats_workspace_bytes(min(chunk, int(distinct.size)))
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=flat.device)
        for start in range(0, distinct.size, chunk):
            part = np.ascontiguousarray(distinct[start:start + chunk])
            _lib.check(lib.qd_order_stats_f32(flat.data_ptr(), n, part.ctypes.data, int(part.size),
                                              out.data_ptr() + 4 * start, workspace.data_ptr(), ws_bytes, _lib.stream_ptr()))
    return out.cpu().numpy()[inverse.reshape(ranks.shape)]


def initialize_quantization_points(tensor, scaling_function, num_points):
    """Starting points for the non-uniform optimisation: the `num_points` evenly spaced
    percentiles of the scaled tensor.  ref: help_functions.py:140-154.

    The reference copies the whole scaled tensor to the host and runs np.percentile there.  Here
    the scaling (K2) runs on the device, the 2*num_points order statistics the interpolation needs
    are SELECTED on the device (order_statistics: three reads of the tensor, no sort) and only they
    are copied back; the interpolation itself is numpy's, so the result is bit-identical
    (tests/golden/misc.npz)."""
    n = tensor.numel()
    scaled = scaling_function.scale_down(tensor).view(-1)[0:scaling_function.original_tensor_length]
    lower, upper, gamma = percentile_ranks(n, num_points)
    found = order_statistics(scaled, np.concatenate([lower, upper]))
    values = percentile_lerp(found[:num_points], found[num_points:], gamma)
    return torch.from_numpy(values).type_as(tensor).to(tensor.device)


def huffman_encode(symb2freq):
    """Huffman code of a {symbol: weight} dict as a list of [symbol, code] sorted by code length.
    ref: help_functions.py:157-172."""
    heap = [[weight, [symbol, '']] for symbol, weight in symb2freq.items()]
    heapify(heap)
    while len(heap) > 1:
        low, high = heappop(heap), heappop(heap)
        for entry in low[1:]:
            entry[1] = '0' + entry[1]
        for entry in high[1:]:
            entry[1] = '1' + entry[1]
        heappush(heap, [low[0] + high[0]] + low[1:] + high[1:])
    return sorted(heappop(heap)[1:], key=lambda e: (len(e[-1]), e))


DEVICE_HISTOGRAM_MAX_SYMBOLS = 256      # the LDS tables of qd_digitize_histogram_f32 / qd_histogram_i64 hold this many


def _digitize_edges(s, tol):
    """The bin edges the reference digitizes against, built exactly as it builds them (ref: :213-216): Python floats."""
    return np.array([x / (s - 1) - tol for x in range(s)], dtype=np.float64)


def _device_counts(kind, values, nsym, edges_dev=None):
    """int64 device tensor of nsym + 1 counters (include/qd_hip.h: qd_digitize_histogram_f32 / qd_histogram_i64) of a flat,
    contiguous device tensor, or None when this tensor cannot take the device path (host fallback)."""
    from .. import _lib
    if not isinstance(values, torch.Tensor) or not values.is_cuda or nsym > DEVICE_HISTOGRAM_MAX_SYMBOLS:
        return None
    if kind == 'digitize' and values.dtype != torch.float32:
        return None
    if kind == 'index' and values.dtype != torch.int64:
        return None
    values = values.contiguous()
    lib = _lib.load()
    with torch.cuda.device(values.device):
        hist = torch.empty(nsym + 1, dtype=torch.int64, device=values.device)
        ws = _lib.workspace(values.device)
        if kind == 'digitize':
            rc = lib.qd_digitize_histogram_f32(values.data_ptr(), values.numel(), edges_dev.data_ptr(), nsym, hist.data_ptr(),
                                               ws.data_ptr(), ws.numel(), _lib.stream_ptr(values.device))
        else:
            rc = lib.qd_histogram_i64(values.data_ptr(), values.numel(), nsym, hist.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr(values.device))
    _lib.check(rc)
    return hist


INDEX_RETAIN_BYTES = 1 << 30            # int64 indices awaiting their counters' trip to the host, per device
FUSED_DIGITIZE_BUCKETS = (64, 128, 256, 512, 1024, 2048)   # the register-resident bucket sizes of qd_scale_digitize_histogram_f32


def _fused_rescale_counts(q_tensor, scal, nsym, edges_dev):
    """The counters of np.digitize(scal.scale_down(q_tensor)[:n], edges) from ONE pass over q_tensor
    (qd_scale_digitize_histogram_f32), or None when the re-scale is not the plain bucketed linear one -- the caller then
    re-scales and digitizes in two steps, as before.  Plain = this package's ScalingFunction, 'linear', no mean subtraction,
    no max_element, a bucket size the kernel keeps in registers, a contiguous fp32 device tensor."""
    from .. import _lib
    from .quant_functions import ScalingFunction
    if type(scal) is not ScalingFunction or scal.type_scaling != 'linear' or scal.subtract_mean or scal.max_element is not False:
        return None
    if scal.bucket_size not in FUSED_DIGITIZE_BUCKETS or nsym > DEVICE_HISTOGRAM_MAX_SYMBOLS or edges_dev is None:
        return None
    if not isinstance(q_tensor, torch.Tensor) or not q_tensor.is_cuda or q_tensor.dtype != torch.float32 or not q_tensor.is_contiguous():
        return None
    if q_tensor.data_ptr() % 16:
        return None
    lib = _lib.load()
    with torch.cuda.device(q_tensor.device):
        hist = torch.empty(nsym + 1, dtype=torch.int64, device=q_tensor.device)
        ws = _lib.workspace(q_tensor.device)
        rc = lib.qd_scale_digitize_histogram_f32(q_tensor.data_ptr(), q_tensor.numel(), int(scal.bucket_size), edges_dev.data_ptr(), nsym,
                                                 hist.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(q_tensor.device))
    if rc == _lib.QD_ERR_UNSUPPORTED:
        return None
    _lib.check(rc)
    return hist


def get_huffman_encoding_mean_bit_length(model_param_iter, quantization_functions, type_quantization='uniform',
                                         s=None):
    """Mean Huffman code length (bits/weight) of the quantization indices of a model.
    ref: help_functions.py:175-232.

    Same steps as the reference for every tensor -- call the quantization function, re-scale the quantized tensor with the
    scaling function it returned, digitize against the s level positions (uniform) or take the returned indices
    (non-uniform), count the symbols -- but the counting runs on the device: the reference copies every quantized tensor
    to the host for np.digitize + np.unique (:215-223); here the re-scale, the digitize (same float64 comparison against
    the same edges) and the histogram are ONE kernel over the quantized tensor (qd_scale_digitize_histogram_f32: 4 B read
    per element) when the scaling function is the plain bucketed linear one, two kernels otherwise (scale_down, then
    qd_digitize_histogram_f32), or one over the int64 indices (qd_histogram_i64) -- and only the s + 1 (k + 1) counters
    cross PCIe, once per device, after the last tensor.  Tensors that cannot take that path (more than 256 symbols, or a
    quantization function that returns host tensors) are counted on the host exactly as before."""
    type_quantization = type_quantization.lower()
    if type_quantization not in ('uniform', 'nonuniform'):
        raise ValueError('type_quantization not recognized')
    if s is None and type_quantization == 'uniform':
        raise ValueError('If type of quantization is uniform, you must provide s')
    if not isinstance(quantization_functions, list):
        quantization_functions = [quantization_functions]
    shared = len(quantization_functions) == 1
    counts = defaultdict(int)
    total = 0
    tol = 1e-5
    edges = _digitize_edges(s, tol) if type_quantization == 'uniform' else None
    edges_dev, device_totals = {}, {}           # per device: the edges, the int64 counters of every tensor (uniform)
    index_hists = {}                            # per device: [(counters of one tensor, its int64 indices or None)] (non-uniform)

    def host_count(bins):
        for value, count in zip(*np.unique(bins, return_counts=True)):
            counts[value] += count

    def flush_index_hists(pending):
        if not pending:
            return
        table = torch.stack([h for h, _b in pending]).cpu().numpy()            # [tensors, MAX_SYMBOLS + 1]
        fine = table[:, -1] == 0                             # last counter: indices outside the table
        h = table[fine, :-1].sum(axis=0)
        for value in np.nonzero(h)[0]:
            counts[int(value)] += int(h[value])
        for (_h, bins), ok in zip(pending, fine):            # the tensors with such indices: counted on the host, as the reference does
            if not ok:
                host_count(bins.cpu().numpy())
        del pending[:]

    for pos, param in enumerate(model_param_iter):
        param = param.clone()
        if hasattr(param, 'data'):
            param = param.data
        total += param.numel()
        fn = quantization_functions[0] if shared else quantization_functions[pos]
        if type_quantization == 'uniform':
            q_tensor, scal = fn(param)
            dev = q_tensor.device if isinstance(q_tensor, torch.Tensor) else None
            if dev is not None and dev.type == 'cuda' and dev not in edges_dev:
                edges_dev[dev] = torch.from_numpy(edges).to(dev)
            hist = _fused_rescale_counts(q_tensor, scal, s, edges_dev.get(dev))
            if hist is None:
                scaled = scal.scale_down(q_tensor).view(-1)[0:scal.original_tensor_length]
                hist = _device_counts('digitize', scaled, s, edges_dev.get(dev))
            if hist is not None:                             # hist[c] = #{digitize == c}; the reference's bin is c - 1
                device_totals.setdefault(dev, []).append(hist)   # summed once per device at the end: no add launch per tensor
            else:
                host_count(np.digitize(scaled.cpu().numpy(), edges).flatten() - 1)
        else:
            _, bins, _ = fn(param)
            bins = bins.view(-1)
            hist = _device_counts('index', bins, DEVICE_HISTOGRAM_MAX_SYMBOLS)
            if hist is None:
                host_count(bins.cpu().numpy())
            else:
                # the counters stay on the device and are fetched for many tensors at once, not one copy per tensor.  The
                # indices are kept until then -- if the table was too small for them (k > 256 points: last counter non-zero)
                # they are counted on the host, as computed: no second call of a possibly stochastic `fn`, no clone of the
                # parameter kept alive -- and a device never holds more than INDEX_RETAIN_BYTES of them
                pending = index_hists.setdefault(bins.device, [])
                pending.append((hist, bins))
                if sum(b.numel() * 8 for _h, b in pending) > INDEX_RETAIN_BYTES:
                    flush_index_hists(pending)
    for hists in device_totals.values():
        h = (hists[0] if len(hists) == 1 else torch.stack(hists).sum(dim=0)).cpu().numpy()
        for c in np.nonzero(h)[0]:
            counts[int(c) - 1] += int(h[c])
    for pending in index_hists.values():
        flush_index_hists(pending)
    assert total == sum(counts.values())
    freq = {sym: c / total for sym, c in counts.items()}
    return sum(freq[sym] * len(code) for sym, code in huffman_encode(freq))


def check_right_bits(tensor_iterator, num_quant_points, bucket_size):
    """True if no tensor has (many) more distinct scaled values than quantization points.
    ref: help_functions.py:234-262 (which notes itself that the check is approximate)."""
    from .quant_functions import ScalingFunction
    per_tensor = not isinstance(num_quant_points, int)
    scaling_function = ScalingFunction('linear', False, False, bucket_size=bucket_size)
    for pos, tensor in enumerate(tensor_iterator):
        if hasattr(tensor, 'data'):
            tensor = tensor.data
        scaled = scaling_function.scale_down(tensor)
        distinct = len(np.unique(scaled.view(-1).cpu().numpy().round(decimals=5)))
        limit = num_quant_points[pos] if per_tensor else num_quant_points
        if distinct > limit + 3:
            return False
    return True
