"""The C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
exactly the symbols include/qd_hip.h declares.  No compute calls here."""
import os
import re
import subprocess

import pytest

from quantized_distillation_amd import _lib
from quantized_distillation_amd import build as qb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    qb.build_extension()
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'qd_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(qd_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree(lib):
    syms = header_symbols()
    assert len(syms) >= 18
    assert sorted(_lib.SIGNATURES) == syms
    for s in syms:
        assert hasattr(lib, s), s


def test_exports_match_header(lib):
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r' T (qd_[a-z0-9_]+)', out)))
    assert exported == header_symbols()


def test_host_only_entry_points(lib):
    # one number in three places: the header's QD_ABI_VERSION (what the library and _qd_glue.so compile in) and the Python binding's
    header = open(os.path.join(ROOT, 'include', 'qd_hip.h')).read()
    assert int(re.search(r'#define QD_ABI_VERSION (\d+)', header).group(1)) == _lib.ABI_VERSION == lib.qd_abi_version() == 3
    assert lib.qd_target_arch() == b'gfx950'
    assert lib.qd_workspace_bytes() >= 64 * 1024
    assert lib.qd_error_string(0) == b'success'
    assert b'invalid' in lib.qd_error_string(-1)
    # bucket geometry follows help_functions.py:67-94
    assert lib.qd_num_buckets(1000, 256) == 4 and lib.qd_padded_length(1000, 256) == 1024
    assert lib.qd_num_buckets(1024, 256) == 4 and lib.qd_padded_length(1024, 256) == 1024
    assert lib.qd_num_buckets(3, 256) == 1 and lib.qd_padded_length(3, 256) == 3
    assert lib.qd_num_buckets(77, 0) == 1 and lib.qd_padded_length(77, 0) == 77


def test_every_compute_entry_point_checks_its_arguments_before_touching_the_device(lib):
    """Error behaviour of the boundary (include/qd_hip.h): null tensors with a non-zero length, zero levels / points, no
    workspace -- every compute entry point returns QD_ERR_INVALID_ARGUMENT without a launch (this runs on a box without a
    GPU), the way the reference's functions raise ValueError before any work (quant_functions.py:22-33,230-236)."""
    import ctypes
    host_only = {'qd_abi_version', 'qd_target_arch', 'qd_error_string', 'qd_workspace_bytes', 'qd_set_single_fused_mode', 'qd_num_buckets',
                 'qd_padded_length', 'qd_multi_plan', 'qd_multi_global_plan', 'qd_multi_dq_plan', 'qd_packed_bytes',
                 'qd_order_stats_workspace_bytes'}
    checked = 0
    for name, (res, args) in sorted(_lib.SIGNATURES.items()):
        if name in host_only:
            continue
        assert res is ctypes.c_int, name
        vals = [100 if a in (ctypes.c_int64, ctypes.c_uint64) else 0 if a in (ctypes.c_int, ctypes.c_size_t) else 0.0 if a is ctypes.c_float
                else None for a in args]
        assert getattr(lib, name)(*vals) == -1, name
        checked += 1
    assert checked >= 24
    # the indices-only form of the nearest-point call needs the pre-scaled input and an index output
    assert lib.qd_nearest_point_f32(None, 0, None, 4, 1, None, None, 8, 100, 256, None, None, None, 0, 0.0, None, 0, None) == -1
    assert lib.qd_selftest_div_invariant(1, 10, 6, None, None) == -1          # families 0 .. 5
    assert lib.qd_point_grad_f32(None, None, 3, None, 10, 256, 4, None, None, 0, None) == -1      # index width 1 or 8


def test_multi_plan_host(lib):
    T = (_lib.QdTensorDesc * 4)()
    for i, n in enumerate([800000, 10, 0, 1025]):
        T[i].n = n
    tiles = lib.qd_multi_plan(T, 4, 256)
    # 3125 buckets -> 782 tiles; 1 bucket -> 1 tile; empty -> 0; 5 buckets -> 2 tiles
    assert [T[i].first_tile for i in range(4)] == [0, 782, 783, 783]
    assert tiles == 785


def test_multi_dq_plan_host_and_the_partial_row_layout(lib):
    """qd_multi_dq_plan: first_tile = prefix of 4-bucket forward tiles, first_block = prefix of FULL 1024-element gradient
    tiles, first_row = prefix of partial rows with min(full tiles, W) + 1 rows per tensor, W = 2048 waves; total_blocks = their
    sum.  And the layout claim the backward kernels rest on (csrc/qd_multi_dq.hip): wave g takes full tiles g, g + W, ...; the
    waves that visit tensor ti (c full tiles from f0 on) are (f0 + i) mod W, i < min(c, W), each once, and wave g writes row
    first_row + ((g - f0) mod W) -- so the rows of a tensor are exactly 0 .. min(c, W) - 1, written once each, plus row
    min(c, W) for the extra wave (n mod 1024).  Modelled here in Python on the config shape lists and on adversarial ones
    (empty tensors, tensors below one tile, a single huge tensor, 200 small tensors: ADVICE r05's 105 MB case)."""
    import ctypes
    import numpy as np
    from harness import kernel_bench
    W = 2048

    def plan(ns, bucket=256):
        T = (_lib.QdDiffQuantDesc * len(ns))()
        for i, n in enumerate(ns):
            T[i].n = n
        blocks = ctypes.c_int64(0)
        tiles = lib.qd_multi_dq_plan(T, len(ns), bucket, ctypes.byref(blocks))
        return (tiles, blocks.value, [T[i].first_tile for i in range(len(ns))], [T[i].first_block for i in range(len(ns))],
                [T[i].first_row for i in range(len(ns))])

    tiles, rows, ft, fb, fr = plan([800000, 10, 0, 1025, 5000])
    assert ft == [0, 782, 783, 783, 785] and tiles == 790
    assert fb == [0, 781, 781, 781, 782]                      # n // 1024: 781, 0, 0, 1, 4
    assert fr == [0, 782, 783, 784, 786] and rows == 791       # 781 + 1, 1, 1, 1 + 1, 4 + 1
    assert plan([0, 0])[1] == 2                                # nothing to do: one (never read) row each
    assert plan([1 << 26])[1] == W + 1                         # 65536 full tiles: every wave visits it
    small = [int(x) for x in np.random.RandomState(1).randint(1, 70000, 200)]
    assert plan(small)[1] == sum(n // 1024 + 1 for n in small) < 200 * 70            # (round 5: 200 x 2049 rows)

    rng = np.random.RandomState(0)
    cases = [[int(np.prod(s)) for s in kernel_bench.model_shapes('wrn')], [int(np.prod(s)) for s in kernel_bench.model_shapes('student')],
             [1 << 26], [1, 1, 1, 1, 1, 1, 1, 1, 1], [0, 5, 0, 0, 7000, 0, 3, 0], [1024] * 40 + [0] + [1025] * 3,
             [int(x) for x in rng.randint(0, 300000, 200)], [int(x) for x in rng.randint(1, 3000, 64)], [3 << 20, 0, 1, 5 << 20]]
    for ns in cases:
        _tiles, rows, _ft, fb, fr = plan(ns)
        nt = len(ns)
        T = fb[-1] + ns[-1] // 1024
        assert T == sum(n // 1024 for n in ns)
        assert fr == list(np.cumsum([0] + [min(n // 1024, W) + 1 for n in ns[:-1]])) and rows == fr[-1] + min(ns[-1] // 1024, W) + 1
        tiles_t = np.arange(T)
        owner = np.searchsorted(np.asarray(fb + [T]), tiles_t, side='right') - 1         # last tensor with prefix <= t ...
        wave = tiles_t % W
        for ti in range(nt):
            touched = sorted(set(wave[owner == ti].tolist())) if T else []
            c = ns[ti] // 1024
            if c == 0:                                            # ... which never is a tensor without a full tile
                assert not touched
                continue
            written = sorted((g - fb[ti]) % W for g in touched)   # the row (relative to first_row) each visiting wave writes
            assert written == list(range(min(c, W)))              # 0 .. min(c, W) - 1, each once: what the fold reads, front to back


def test_kernels_are_gfx950_only(lib):
    """The fat binary carries exactly one device target: gfx950 (no multi-arch, no fallbacks)."""
    blob = open(_lib.LIB_PATH, 'rb').read()
    targets = set(re.findall(rb'hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}


def test_native_glue_loads_and_fails_loudly_on_cpu_tensors():
    """_qd_glue.so (csrc/qd_torch_glue.cpp) is the per-call binding the Python API uses: it must load, agree with
    libqd_hip.so on the ABI version, and refuse CPU tensors -- it is the DEVICE library's binding; CPU tensors never reach it
    (the public API sends them to libqd_host.so by the tensor's device: tests/test_host_parity.py)."""
    import pytest
    import torch
    from quantized_distillation_amd import _lib
    g = _lib.glue()
    assert g.abi_version() == _lib.load().qd_abi_version() == _lib.ABI_VERSION
    for name in ('uniform', 'nearest', 'point_grad', 'mark_written'):
        assert callable(getattr(g, name))
    with pytest.raises(RuntimeError, match='no CPU path'):
        g.uniform(torch.zeros(8), 16, 256, False, 0.0, False, 0, False, False)
    with pytest.raises(TypeError):
        g.uniform([1.0, 2.0], 16, 256, False, 0.0, False, 0, False, False)
    import quantization
    q, _ = quantization.uniformQuantization(torch.zeros(8), 16, bucket_size=4)       # ... and the API does not send them there
    assert q.device.type == 'cpu' and torch.equal(q, torch.zeros(8))


def test_host_library_exports_its_entry_points_with_the_header_signatures():
    """libqd_host.so: every symbol of _lib.HOST_SYMBOLS is exported, declared in include/qd_hip.h under the same name, and the
    library identifies itself as the host build of the same ABI version."""
    h = _lib.host()
    assert h.qd_abi_version() == _lib.ABI_VERSION and h.qd_target_arch() == b'host' and h.qd_host_max_threads() >= 1
    header = open(os.path.join(_lib.INCLUDE, 'qd_hip.h')).read()
    for name in _lib.HOST_SYMBOLS:
        assert hasattr(h, name), name
        assert re.search(r'\b%s\s*\(' % name, header), name
    exported = subprocess.run(['nm', '-D', '--defined-only', _lib.HOST_LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    names = {l.split()[-1] for l in exported.splitlines() if ' T ' in l}
    only_here = {'qd_host_max_threads', 'qd_host_set_threads'}
    assert names == set(_lib.HOST_SYMBOLS) | only_here, names ^ (set(_lib.HOST_SYMBOLS) | only_here)
    assert h.qd_num_buckets(1000, 256) == 4 and h.qd_padded_length(1000, 256) == 1024 and h.qd_num_buckets(100, 256) == 1
    assert h.qd_uniform_f32(None, None, 10, 256, 16, None, None, None, None, 0, 0.0, 0, 0, None, 0, None) == -1


def test_glue_common_path_declines_what_it_does_not_handle():
    """glue.uniform_common (the native common-case entry point of uniformQuantization) returns None -- "take the general
    path" -- for anything but a contiguous fp32 device tensor with exact-int arguments; on this CPU box that is every call,
    so the general path's errors are what a caller sees."""
    import numpy as np
    import torch
    import quantization
    from quantization.quant_functions import ScalingFunction
    g = _lib.glue()
    with pytest.raises(TypeError):
        g.register(3)
    g.register(ScalingFunction)
    x = torch.zeros(8)
    for args in ((x, 16, 256), (x, 16, None), (x.double(), 16, 4), (x, 16.0, 4), (x, np.int64(16), 4), (x, True, 4),
                 (x, 16, True), (x, 16, 0), (x, 16, -1), (x, 1, 4), (x, 2 ** 40, 4), (x, 16, 2 ** 70), ('x', 16, 4),
                 (x, 16), (x, 16, 4, 5)):
        assert g.uniform_common(*args) is None, args
    assert quantization.uniformQuantization(x, 16, bucket_size=4)[0].device.type == 'cpu'       # (CPU tensors: libqd_host.so)
    with pytest.raises(ValueError):
        quantization.uniformQuantization(x, 16, bucket_size=True)


def test_no_kernel_uses_scratch_memory(lib):
    """Every kernel of the shipped library keeps its per-lane data in registers: `private_segment_fixed_size` (scratch
    bytes per lane) and the VGPR spill count are 0 for all of them -- read from the code objects inside libqd_hip.so
    (tools/kernel_meta.py).  A spilled per-lane array in an HBM-bound kernel is extra, uncounted memory traffic; round 2
    shipped three such instantiations (272 / 528 / 228 bytes per lane in the nearest-point kernels)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import kernel_meta
    ks = kernel_meta.kernels(_lib.LIB_PATH)
    assert len(ks) >= 100, 'expected the whole kernel family, found %d' % len(ks)
    bad = [(k['name'], k.get('private_segment_fixed_size', 0), k.get('vgpr_spill_count', 0)) for k in ks
           if k.get('private_segment_fixed_size', 0) != 0 or k.get('vgpr_spill_count', 0) != 0 or k.get('uses_dynamic_stack') == 'true']
    assert not bad, bad
    assert all(k['vgpr_count'] <= 512 for k in ks)


def test_write_once_outputs_keep_their_non_temporal_hint(lib):
    """The streaming kernels store their write-once outputs with the non-temporal hint (global_store_dwordx4 ... nt).  The
    optimiser can lose it: when it merges two copies of a loop the merged store drops !nontemporal, and the kernel then
    runs with plain stores (scale_down shipped like that through round 3: 92 us against 85 us at bucket 256).  Read from
    the disassembly of the shipped code objects: every 16-byte global store of every kernel carries `nt`, except the int64
    point indices of the nearest-point kernels (plain on purpose: 175.8 us against 180.0 us with the hint, qd_transform.h
    store_side4_row) and the two in-place epilogues that rewrite only the float4s they change."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import kernel_meta
    ws = kernel_meta.wide_stores(_lib.LIB_PATH)
    dm = kernel_meta.demangle(list(ws))
    assert len(ws) >= 60, 'expected the streaming kernels, found %d with 16-byte stores' % len(ws)
    bad = []
    for sym, (total, nt) in ws.items():
        name = dm[sym].replace('(anonymous namespace)::', '')
        nearest = re.search(r'k_(bucket_\w+|single_\w+)<2,', name) or re.search(r'k_single_apply<2>', name) or 'k_nearest_prescaled_stream' in name
        in_place = name.startswith('k_clamp(') or name.startswith('k_truncated_ste(')
        if in_place:
            continue
        if nearest:
            if nt * 5 < total or nt == 0:                   # the q stores (one in five with int64 indices) keep the hint
                bad.append((name, total, nt))
        elif nt != total:
            bad.append((name, total, nt))
    assert not bad, bad


def test_no_environment_knobs_in_the_product(lib):
    """One code path per configuration: the product library reads no environment variable (round 2 shipped nine QD_*
    A/B knobs that switched between alternative kernel implementations at run time).  libqd_hip.so neither imports
    getenv nor contains a QD_* name; the only run-time switch is the documented qd_set_single_fused_mode()."""
    blob = open(_lib.LIB_PATH, 'rb').read()
    names = set(re.findall(rb'QD_[A-Z][A-Z0-9_]{2,}', blob))
    assert not names, names
    undefined = subprocess.check_output(['nm', '-D', '--undefined-only', _lib.LIB_PATH], text=True)
    assert 'getenv' not in undefined, 'libqd_hip.so imports getenv'
    for f in os.listdir(_lib.CSRC):
        if f.endswith(('.hip', '.h')):
            text = open(os.path.join(_lib.CSRC, f)).read()
            for m in re.finditer(r'getenv\s*\(', text):
                # allowed only inside an `#ifdef QD_TUNING` block (never defined by quantized_distillation_amd/build.py)
                before = text[:m.start()]
                assert before.rfind('#ifdef QD_TUNING') > before.rfind('#endif'), (f, text[m.start() - 80:m.start() + 40])
    assert 'QD_TUNING' not in ' '.join(qb.HIPCC_FLAGS)
