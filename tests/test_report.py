"""harness/report.py: the line bench.py puts on stdout is ONE JSON object of at most 4096 bytes, whatever the legs measured
or failed to measure (round 5's 24 KB line could not be parsed from the driver's record of stdout)."""
import copy
import json
import os

import pytest

from harness import report

HERE = os.path.dirname(os.path.abspath(__file__))


def full_record():
    with open(os.path.join(HERE, 'golden', 'bench_record_full.json')) as f:
        return json.load(f)


def test_compact_of_a_full_record_fits_with_room_to_spare_and_keeps_the_contract():
    full = full_record()
    c = report.compact(full)
    text = report.fit(c)
    assert len(text.encode()) <= 3500 and '\n' not in text
    d = json.loads(text)
    assert 'dropped_to_fit' not in d
    assert list(d)[:len(report.CONTRACT)] == list(report.CONTRACT)         # the contract fields lead the line
    for k in report.CONTRACT:
        assert d[k] == full[k], k
    r, fr = d['roofline'], full['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_us', 'rocprof_kernel_avg_us', 'rocprof_frac'):
        assert r[k] == fr[k], k
    assert r['traffic_measured_in_this_run'] is True and r['kernel_rows'] == 28
    worst = min((x for x in fr['kernels'] if x['n'] >= 1 << 24), key=lambda x: x['frac'])
    assert r['worst_kernel_frac'] == round(worst['frac'], 4) and worst['name'].startswith(r['worst_kernel'].rstrip('~'))
    assert d['kernels_frac']['K1'] <= fr['frac'] + 0.02 and 'K6m' in d['kernels_frac'] and 'K8' in d['kernels_frac']
    assert d['cpu_baseline']['value'] == full['cpu_baseline']['value'] and d['cpu_baseline']['cores'] == full['cpu_baseline']['cores']
    assert d['steps_per_sec']['cfg1_cifar_student'] == round(full['distill']['cifar_student']['multi']['steps_per_sec'], 2)
    assert d['dp']['cfg3']['dp_efficiency'] == full['distill']['imagenet_resnet18k_dp']['dp_efficiency']
    # nothing nested deeper than two levels, no prose beyond the two short descriptions
    for k, v in d.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                assert not isinstance(vv, list), (k, kk)
                if isinstance(vv, dict):
                    assert all(not isinstance(x, (dict, list)) for x in vv.values()), (k, kk)
    assert max(len(v) for v in _strings(d)) <= 300


def _strings(o):
    if isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, str):
        yield o


def test_partial_records_never_raise_and_say_what_is_missing():
    assert json.loads(report.fit(report.compact(None))) == {}
    assert json.loads(report.fit(report.compact({'metric': 'm', 'value': None, 'error': 'x' * 5000})))['error'].endswith('~')
    full = full_record()
    for drop in ('roofline', 'cpu_baseline', 'distill', 'config'):
        rec = copy.deepcopy(full)
        del rec[drop]
        d = json.loads(report.fit(report.compact(rec)))
        assert d['value'] == full['value']
    rec = copy.deepcopy(full)
    rec['cpu_baseline'] = None                                            # N > 1: no CPU baseline
    rec['distill']['diffquant_wrn'] = {'error': 'RuntimeError: ' + 'out of memory ' * 40}
    rec['distill']['nmt_lstm_dp'] = {'skipped': 'wall budget: 140 s spent + ~9 s expected > --budget-s 150'}
    d = json.loads(report.fit(report.compact(rec)))
    assert d['cpu_baseline'] is None
    assert d['steps_per_sec']['cfg2_diffquant_wrn'].startswith('RuntimeError') and len(d['steps_per_sec']['cfg2_diffquant_wrn']) <= 60
    assert d['steps_per_sec']['cfg4_nmt_lstm'].startswith('wall budget') and 'cfg2' not in d['dp'] and 'cfg4' not in d['dp']
    assert 'cfg0_cpu_reference_quantizer' not in d['steps_per_sec']


@pytest.mark.parametrize('limit', [4096, 2048, 1024, 400])
def test_fit_holds_any_limit_and_gives_up_the_optional_groups_first(limit):
    c = report.compact(full_record())
    c['kernels_frac'] = {'tag%03d' % i: 0.5 for i in range(400)}           # a record grown far beyond what any run produces
    c['legs_wall_s'] = 'x=1 ' * 500
    text = report.fit(c, limit)
    assert len(text.encode()) <= limit
    d = json.loads(text)
    assert d['metric'] == 'quantize_dequantize_GBps_64M_fp32_4bit' and d['value'] == c['value']
    if limit >= 2048:
        assert d['roofline']['frac'] == c['roofline']['frac'] and d['cpu_baseline']['value'] == c['cpu_baseline']['value']
        assert d['dropped_to_fit'][:2] == ['legs_wall_s', 'bench_process'] and 'kernels_frac' in d['dropped_to_fit']
    # multi-byte text is counted in bytes
    text = report.fit({'metric': 'm', 'value': 1.0, 'error': 'µ' * 3000}, 1024)
    assert len(text.encode()) <= 1024 and json.loads(text)['value'] == 1.0


def test_bench_py_workload_string_fits_the_compact_line_unshortened():
    src = open(os.path.join(os.path.dirname(HERE), 'bench.py')).read()
    i = src.index("'workload': '")
    j = src.index("',\n", i)
    workload = eval(src[i + len("'workload': "):j + 1])
    assert len(workload) <= 120 and '64Mi' in workload and 'bucket_size=256' in workload and 's=16' in workload
