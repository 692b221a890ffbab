"""Host-side behaviour of the `quantization` mirror that needs no GPU: argument validation and
exceptions (same as the reference), pure-host helpers against golden values, and what happens to
tensors that are not on a HIP device: CPU fp32 tensors are computed by libqd_host.so (tests/test_host_parity.py),
anything else fails loudly -- as do the device-only entry points on CPU tensors."""
import numpy as np
import pytest
import torch

import quantization
import quantization.help_functions as qhf
from quantized_distillation_amd import _lib


def test_public_names():
    for name in ('uniformQuantization', 'ScalingFunction', 'nonUniformQuantization',
                 'uniformQuantization_variable', 'nonUniformQuantization_variable'):
        assert hasattr(quantization, name)
    assert isinstance(quantization.USE_CUDA, bool)
    import quantization.quant_functions as qf
    assert hasattr(qf, 'SearchSorted')


def test_scaling_function_validation():
    SF = quantization.ScalingFunction
    SF('Linear', False, False, None)                      # case-insensitive
    with pytest.raises(ValueError):
        SF('foo', False, False, None)
    for bad in (0, -4, 2.5, np.int64(256)):
        with pytest.raises(ValueError):
            SF('linear', False, False, bad)
    with pytest.raises(ValueError):
        SF('linear', True, False, None)                   # a bool is refused; pass a number or False
    sf = SF('linear', 0.5, True, 256, False)
    assert sf.tol_diff_zero == 1e-10 and sf.alpha is None and sf.bucket_size == 256


def test_one_library_per_device_and_loud_failures():
    """A CPU fp32 tensor is computed by libqd_host.so and the result stays on the CPU (parity: tests/test_host_parity.py);
    what has no host library -- other dtypes, meta tensors, the multi-tensor / codec entry points -- raises."""
    x = torch.randn(100)
    q, sf = quantization.uniformQuantization(x, 16, bucket_size=256)
    assert q.device.type == 'cpu' and sf.alpha.device.type == 'cpu' and q.shape == x.shape
    assert quantization.nonUniformQuantization(x, [0.0, 1.0])[1].dtype == torch.int64
    assert quantization.ScalingFunction('linear', False, False, None).scale_down(x).device.type == 'cpu'
    with pytest.raises(TypeError, match='float32'):
        quantization.uniformQuantization(x.double(), 16, bucket_size=256)
    with pytest.raises(TypeError, match='float32'):
        quantization.ScalingFunction('linear', False, False, None).scale_down(x.half())
    with pytest.raises(RuntimeError, match='HIP device or on the CPU'):
        quantization.uniformQuantization(torch.empty(100, device='meta'), 16, bucket_size=256)
    from quantized_distillation_amd import codec
    from quantized_distillation_amd.multi_tensor import MultiTensorDiffQuant, MultiTensorQuantizer
    for call in (lambda: MultiTensorQuantizer([x], 16, 256), lambda: codec.pack_uniform(x, 16, 256),
                 lambda: MultiTensorDiffQuant([x], [torch.empty(100)], [torch.empty(100)], 4, 256)):
        with pytest.raises(RuntimeError, match='HIP device'):
            call()
    assert _lib.lib_for(x) is _lib.host() and _lib.stream_for(x) is None


def test_nonuniform_argument_errors():
    x = torch.randn(10)
    with pytest.raises(ValueError):
        quantization.nonUniformQuantization(x, [0.0, 1.0], pre_processed_values=True)
    with pytest.raises(ValueError):
        quantization.nonUniformQuantization(x, [0.0, 1.0], scaling_function=object())
    with pytest.raises(ValueError):
        quantization.nonUniformQuantization_variable(pre_process_tensors=True)
    fn = quantization.nonUniformQuantization_variable()
    with pytest.raises(ValueError):
        fn.forward(x, torch.zeros(2, 2))
    with pytest.raises(ValueError):
        fn.backward(x)


def test_uniform_variable_backward_guards():
    g = torch.randn(4)
    with pytest.raises(ValueError):
        quantization.uniformQuantization_variable(16, type_of_scaling='absmax', bucket_size=4).backward(g)
    with pytest.raises(NotImplementedError):
        quantization.uniformQuantization_variable(16, subtract_mean=True, bucket_size=4).backward(g)
    with pytest.raises(NotImplementedError):
        quantization.uniformQuantization_variable(16).backward(g)
    with pytest.raises(ValueError):
        quantization.uniformQuantization_variable(16, bucket_size=4).backward(g)


def test_assign_bits_matches_reference(golden_misc):
    for c in golden_misc.meta['assign_bits']:
        got = qhf.assign_bits_automatically(c['norms'], c['init'], input_is_point=c['input_is_point'])
        assert got == c['result']
        assert sum(got) == (c['init'] * len(c['norms']) if isinstance(c['init'], int) else sum(c['init']))
    with pytest.raises(ValueError):
        qhf.assign_bits_automatically([1.0, 2.0], [4])


def test_huffman_encode():
    code = dict(qhf.huffman_encode({'a': 0.5, 'b': 0.25, 'c': 0.125, 'd': 0.125}))
    assert sorted(len(v) for v in code.values()) == [1, 2, 3, 3]
    assert len(code['a']) == 1
    # prefix free
    vals = list(code.values())
    assert not any(a != b and b.startswith(a) for a in vals for b in vals)
    with pytest.raises(ValueError):
        qhf.get_huffman_encoding_mean_bit_length([], None, 'weird')
    with pytest.raises(ValueError):
        qhf.get_huffman_encoding_mean_bit_length([], None, 'uniform')


def test_create_bucket_tensor_layout():
    x = torch.arange(10, dtype=torch.float32)
    b = qhf.create_bucket_tensor(x, 4)
    assert b.shape == (3, 4) and b[2].tolist() == [8, 9, 9, 9]
    assert qhf.create_bucket_tensor(x, 16).shape == (1, 10)
    assert qhf.create_bucket_tensor(x, None) is x
    assert qhf.create_bucket_tensor(x[:8], 4).shape == (2, 4)
    assert torch.isnan(qhf.create_bucket_tensor(x, 4, fill_values='nan')[2, 2:]).all()


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.QdLibraryMissing):
        _lib.load()


def test_percentile_from_sorted_is_numpy_exact():
    """The 2k-order-statistics percentile used by initialize_quantization_points reproduces
    np.percentile (what the reference calls, help_functions.py:150) bit for bit."""
    rng = np.random.RandomState(0)
    for trial in range(600):
        n = int(rng.choice([1, 2, 3, 5, 10, 100, 1000, 4097, 100003]))
        k = int(rng.choice([2, 3, 4, 7, 8, 16, 64, 256]))
        x = rng.rand(n).astype(np.float32)
        if trial % 3 == 0:
            x = np.round(x * 8) / 8
        s = np.sort(x)
        got = qhf.percentile_points_from_sorted(lambda i: s[i], n, k)
        assert np.array_equal(got, np.percentile(x, np.linspace(0, 100, num=k))), (n, k)


def test_checkpoint_formats_roundtrip(tmp_path):
    """The on-disk formats the reference's evaluation scripts read (cifar10_test.py:265-270,
    model_manager.py:182-190,214-249,321-347): plain pickles of Python lists/dicts + torch.save state dicts.
    tests/test_checkpoints_reference.py opens the same files with the reference's own ModelManager."""
    import pickle
    from harness import checkpoints, models
    net = models.student()
    pts = torch.sort(torch.rand(22, 4), dim=1)[0]
    pts[3, 2:] = float('inf')                          # a tensor that got 2 points from the automatic allocation
    base = str(tmp_path / 'quant_points_2bits')
    checkpoints.save_quantization_points(base, pts, {'predictionAccuracy': [0.5], 'numEpochsTrained': 1}, net.state_dict())
    with open(base, 'rb') as f:                        # exactly what the reference's reader does
        raw_points, info = pickle.load(f)
    assert isinstance(raw_points, list) and len(raw_points) == 22 and isinstance(raw_points[0][0], float)
    assert len(raw_points[3]) == 2 and len(raw_points[0]) == 4
    assert info['numEpochsTrained'] == 1
    p2, _, sd = checkpoints.load_quantization_points(base)
    assert np.allclose(np.array(p2[0], dtype=np.float32), pts[0].numpy())
    net.load_state_dict(sd)

    store = checkpoints.RunStore(str(tmp_path / 'manager'), 'cifar10', create=True)
    store.add_new_model('student', str(tmp_path / 'student'), {'spec': {'conv': [75, 50]}, 'useBatchNorm': True})
    args = {'numBits': 4, 'bucket_size': 256, 'loss_function': torch.nn.functional.cross_entropy, 'teacher_model': net,
            'learning_rate_style': 'generic', 'quantize_first_and_last_layer': False}
    assert store.append_run('student', net.state_dict(), args, {'numEpochsTrained': 0}) is None    # aborted run: nothing saved
    path = store.append_run('student', net.state_dict(), args, {'numEpochsTrained': 2, 'lossSaved': [1.0, 0.5]})
    assert path.endswith('student1') and store.get_num_training_runs('student') == 1
    again = checkpoints.RunStore(str(tmp_path / 'manager'))
    meta = again.load_metadata('student')
    assert isinstance(meta, list) and len(meta) == 2 and all(isinstance(m, dict) for m in meta)
    assert meta[0]['numBits'] == 4 and meta[0]['quantize_first_and_last_layer'] is False
    assert meta[0]['loss_function'].startswith('Name: cross_entropy. Repr: <function cross_entropy')
    assert meta[0]['teacher_model'].startswith('ConvNet(')            # nn.Module: callable without __name__ -> repr
    assert meta[1]['lossSaved'] == [1.0, 0.5]
    assert again.load_metadata('student', 0)[0]['spec'] == repr({'conv': [75, 50]})     # a dict value is repr()-ed
    assert set(again.load_model_state_dict('student')) == set(net.state_dict())
    with pytest.raises(ValueError):
        checkpoints.RunStore(str(tmp_path / 'manager'), 'x', create=True)


def test_hyperspherical_helpers_match_reference():
    """help_functions.py:8-64 (off the quantization path): outputs of the reference itself, tests/golden/coords.npz."""
    from conftest import load_golden
    import quantization.help_functions as hf
    G = load_golden('coords.npz')
    for i, c in enumerate(G.meta):
        x = torch.from_numpy(G.z['c%d_x' % i])
        r, ang = hf.cart2hyperspherical(x.clone())
        assert np.allclose(float(r), float(G.z['c%d_r' % i]), rtol=1e-6, atol=0)
        assert np.allclose(ang.numpy(), G.z['c%d_ang' % i], rtol=1e-6, atol=1e-7), i
        back = hf.hypershperical2cart((r, ang))
        assert np.allclose(back.numpy(), G.z['c%d_back' % i], rtol=1e-5, atol=1e-6), i
        assert np.array_equal(hf.invert_pytorch_vector(x).numpy(), G.z['c%d_inv' % i])
        assert hf.findFirstNonZeroIndex(x) == c['first_nonzero']


def test_bucket_invariant_division_in_exact_arithmetic():
    """The quantize kernels divide by a bucket's alpha as q = RN(n y), r = fma(-alpha, q, n), u = fma(r, y, q) with
    y = RN(1/alpha) (qd_common.h) instead of the reference's IEEE division (quant_functions.py:106-107).  Restated in
    exact rational arithmetic with one explicit fp32 rounding per operation (tools/div_invariant_check.py), the result IS
    the correctly rounded quotient on 30000 adversarial pairs inside the stated range -- and is NOT always outside it
    (so this test can fail).  The device-side check over 10^8 pairs is tests/test_hip_parity.py::
    test_division_by_bucket_invariant_alpha."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import div_invariant_check as dic
    bad, outside_bad = dic.run_host(30000, seed=3, verbose=False)
    assert bad == 0
    assert outside_bad > 0
    # scale_down returns the quotient itself: exact while the quotient is normal (or rounds to zero), not always when it is
    # denormal -- the kernels' numerator threshold max(2^-100, alpha 2^-120) keeps those buckets on the IEEE division
    st = dic.run_host_small_quotients(20000, seed=5, verbose=False)
    assert st['normal'][0] > 10000 and st['normal'][1] == 0
    assert st['zero'][1] == 0
    assert st['denormal'][0] > 500


def test_committed_bench_lines_keep_the_contract_and_are_self_consistent():
    """The bench.py runs committed under profiles/ (round 6: the stdout line as the driver sees it + the full record that went
    to bench_detail.json): the line is one JSON object of at most 4096 bytes that parses from the last 6000 bytes of stdout,
    carries every key the driver's contract names and IS the compact form of the record; the numbers agree with each other:
    roofline.achieved = algorithmic bytes / average launch time, frac = achieved / peak, value = the same bytes / ms_per_step
    (whole call, so never above the kernel's own rate), measured traffic within 2 % of the algorithmic bytes."""
    import glob
    import json
    import os
    from harness import report
    from harness.dpbench import DP_KEYS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [f for f in sorted(glob.glob(os.path.join(root, 'profiles', 'r06_bench_*.json'))) if 'detail' not in os.path.basename(f)]
    assert len([f for f in files if 'driver' in f]) >= 2        # the driver's command on at least two fresh leases
    for f in files:
        out = open(f).read()
        lines = [l for l in out.splitlines() if l.strip()]
        assert len(lines) == 1 and len(lines[0].encode()) <= report.LINE_LIMIT, (f, 'ONE line of at most 4096 bytes on stdout', len(out))
        d = json.loads(out[-6000:].splitlines()[-1])
        assert 'dropped_to_fit' not in d, f
        full = json.load(open(f.replace('r06_bench_', 'r06_bench_detail_')))
        want = json.loads(report.fit(report.compact(full)))
        assert {k: v for k, v in d.items() if k != 'detail'} == {k: v for k, v in want.items() if k != 'detail'}, f     # the line is the compact form of the record, nothing else
        for key in report.CONTRACT + ('config', 'roofline', 'cpu_baseline'):
            assert key in d, (f, key)
        assert d['metric'] == 'quantize_dequantize_GBps_64M_fp32_4bit' and d['unit'] == 'GB/s' and d['dtype'] == 'f32' and d['data'] == 'synthetic'
        assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
        assert '64Mi' in d['config']['workload'] and len(d['config']['workload']) <= 120 and d['config']['levels'] == 16 and d['config']['bucket_size'] == 256
        r = d['roofline']
        for key in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'avg_launch_us'):
            assert key in r, (f, key)
        assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and r['unit'] == 'GB/s'
        algo = r['algorithmic_bytes_per_launch']
        assert algo == 8 * d['config']['n_elements_per_gpu']
        assert abs(r['achieved'] - algo / r['avg_launch_us'] / 1e3) <= 1e-3 * r['achieved'], f
        assert abs(r['frac'] - r['achieved'] / r['peak']) <= 1e-3
        whole = d['n_gpus'] * algo / (d['ms_per_step'] * 1e-3) / 1e9
        assert abs(d['value'] - whole) <= 2e-3 * whole, (f, d['value'], whole)
        if d.get('collective_backend') != 'gloo':                # (two ranks sharing ONE GPU through gloo: the flow, not the number)
            assert d['value'] <= d['n_gpus'] * r['achieved'] * 1.001 and r['frac'] >= 0.70, (f, r['frac'])
        # the kernel's duration by rocprofv3 in the same run: what the HIP-event average (launch gaps included) has to agree with
        if r.get('rocprof_kernel_avg_us'):
            assert 0.95 * r['avg_launch_us'] <= r['rocprof_kernel_avg_us'] <= 1.01 * r['avg_launch_us'], (f, r['rocprof_kernel_avg_us'], r['avg_launch_us'])
            assert full['roofline']['rocprof_kernel_launches'] >= 500
        if r['traffic'] is not None:
            assert 0.98 * algo <= r['traffic'] <= 1.02 * algo, (f, r['traffic'])
        if 'driver' in os.path.basename(f):
            bp = full['bench_process']                           # what the guardian saw: one worker, clean exit, nothing lost
            assert bp['restarts'] == 0 and bp['workers'][-1]['exit'] == 0 and 'legs_lost_with_their_worker' not in bp, (f, bp)
            assert bp['wall_s'] <= 90, (f, bp['wall_s'])         # (round 5: 113 s, 48 of them MIOpen's first-use search)
            c = d['cpu_baseline']
            for key in ('value', 'unit', 'cores', 'kind', 'sample'):
                assert key in c, (f, key)
            assert c['kind'] == 'reference' and c['cores'] >= 1 and c['value'] < d['value'] / 100 and not c['sample'].endswith('~')
            assert d['parity_bit_exact_vs_reference'] is True and r['traffic_measured_in_this_run'] is True
            assert set(d['steps_per_sec']) >= {'cfg0_cpu_reference_quantizer', 'cfg1_cifar_student', 'cfg2_diffquant_wrn', 'cfg3_imagenet_resnet18k', 'cfg4_nmt_lstm'}
            assert all(isinstance(v, float) for v in d['steps_per_sec'].values()), (f, d['steps_per_sec'])
        # the per-kernel rows of the full record: self-consistent, and the line's one-number-per-kernel view follows them
        rows = full['roofline'].get('kernels')
        if rows is not None:
            assert isinstance(rows, list) and len(rows) >= 14 and r['kernel_rows'] == len(rows), (f, len(rows))
            for i, row in enumerate(rows):
                for key in ('name', 'kernel', 'us', 'bytes_per_elem', 'GBps', 'frac'):
                    assert key in row, (f, i, key)
                assert abs(row['GBps'] - row['bytes_per_elem'] * row['n'] / row['us'] / 1e3) <= 2e-3 * row['GBps'] + 0.2, (f, row)
                assert abs(row['frac'] - row['GBps'] / 8000.0) <= 1e-3, (f, row)
                tag = row['name'].split(' ')[0][:6]
                if row['n'] >= 1 << 24:                      # (launch-bound small rows are in the full record only)
                    assert d['kernels_frac'][tag] <= round(row['frac'], 3), (f, tag)
            headline = rows[0]
            assert abs(headline['us'] - r['avg_launch_us']) <= 0.05 * r['avg_launch_us'], (f, headline['us'], r['avg_launch_us'])
        # the steps/sec legs of the full record carry the data-parallel report, the line its scalars
        for leg, cfg in (('diffquant_wrn', 'cfg2'), ('imagenet_resnet18k_dp', 'cfg3'), ('nmt_lstm_dp', 'cfg4')):
            rec = (full.get('distill') or {}).get(leg)
            if isinstance(rec, dict) and 'steps_per_sec' in rec:
                for key in DP_KEYS:
                    assert key in rec or (key in ('allreduce_alone_ms', 'busbw_GBps', 'algbw_GBps', 'xgmi_peak_GBps_per_gpu') and rec['exchanged_bytes_per_step'] == 0), (f, leg, key)
                assert rec['n_gpus'] == d['n_gpus'] and d['dp'][cfg]['dp_efficiency'] == rec['dp_efficiency']


def test_kernel_bench_helpers_without_a_gpu():
    """harness/kernel_bench.py: the shape lists it times the multi-tensor kernels on are the parameter shapes of the BASELINE
    config models, and a row's one-line form fits what the driver's record keeps of a string."""
    from harness import kernel_bench, models
    wrn = kernel_bench.model_shapes('wrn')
    assert len(wrn) == 60 and sum(int(np.prod(s)) for s in wrn) == 82746890
    st = kernel_bench.model_shapes('student')
    assert len(st) == 22 and sum(int(np.prod(s)) for s in st) == 1000235
    assert st == [tuple(p.shape) for p in models.student().parameters()]
    row = {'name': 'K5 diff-quant forward k=16 (u resident, u8 idx)', 'kernel': 'k_nearest_prescaled_stream<false>', 'us': 91.73,
           'bytes_per_elem': 9, 'GBps': 6584.6, 'frac': 0.8231, 'n': 1 << 26}
    flat = kernel_bench.flat_row(row)
    assert len(flat) <= 118 and flat.startswith('K5 diff-quant forward k=16') and '| 91.73 us | 9 B/el | 6585 GB/s | 0.823 |' in flat
    with pytest.raises(ValueError):
        kernel_bench.model_shapes('nope')
