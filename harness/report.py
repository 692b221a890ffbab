"""What bench.py puts on stdout: ONE JSON line of at most LINE_LIMIT bytes.

The measuring process builds a large record (28 kernel rows, the data-parallel report of every steps/sec leg, the rocprofv3
and PMC sub-records, sample descriptions).  That record goes to `bench_detail.json` next to bench.py and to stderr.  Stdout
carries `compact(record)`: the contract fields, `config`, `roofline` and `cpu_baseline` as the contract words them, and a
handful of numbers per config -- and `fit()` guarantees the size whatever a leg put into the record, by dropping the
optional groups (last first) until the line fits.  Round 5's line was 24 KB and the driver's record of stdout is shorter
than that; tests/test_report.py and tests/test_guardian.py hold the limit from now on.

Stdlib only (the guardian process imports this; it must never load torch or HIP).
"""
import json

LINE_LIMIT = 4096            # bytes, newline not counted
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data')
# optional groups of the compact line, in the order fit() gives them up
DROP_ORDER = ('legs_wall_s', 'bench_process', 'kernels_frac', 'dp', 'steps_per_sec', 'device', 'collective_backend', 'miopen_find_mode')


def _num(x, nd=4):
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (int, float)):
        return round(x, nd) if isinstance(x, float) else x
    return None


def _short(s, n):
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + '~'


def _sps(rec):
    """steps/sec of one leg record: the number, or a short reason why there is none."""
    if not isinstance(rec, dict):
        return None
    if 'steps_per_sec' in rec:
        return _num(rec['steps_per_sec'], 2)
    why = rec.get('error') or rec.get('skipped')
    return _short(why, 60) if why else None


def _dp(rec):
    """The data-parallel figures of one steps/sec leg, numbers only."""
    if not isinstance(rec, dict) or 'steps_per_sec' not in rec:
        return None
    out = {'steps_per_sec': _num(rec['steps_per_sec'], 2)}
    for key, nd in (('dp_efficiency', 4), ('busbw_GBps', 1), ('exposed_comm_ms', 3), ('allreduce_alone_ms', 4), ('global_batch', 0)):
        if key in rec:
            out[key] = _num(rec[key], nd)
    return out


def compact(line):
    """The stdout form of the full record `line` (a dict).  Never raises on a partial record: a run that ended early prints
    what it has."""
    line = line or {}
    out = {k: line.get(k) for k in CONTRACT if k in line}
    for k in ('error',):
        if line.get(k):
            out[k] = _short(line[k], 300)
    cfg = line.get('config') or {}
    if cfg:
        out['config'] = {'workload': _short(cfg.get('workload'), 120)}
        for k in ('n_elements_per_gpu', 'levels', 'bucket_size'):
            if k in cfg:
                out['config'][k] = cfg[k]
        if cfg.get('parallelism'):
            out['config']['parallelism'] = _short(cfg['parallelism'], 60)
    r = line.get('roofline') or {}
    if r:
        ro = {}
        for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_over_algorithmic',
                  'algorithmic_bytes_per_launch', 'avg_launch_us', 'rocprof_kernel_avg_us', 'rocprof_frac', 'torch_d2d_copy_GBps',
                  'pcie_inclusive_GBps_note'):
            if k in r:
                ro[k] = r[k] if isinstance(r[k], str) else _num(r[k])
        if 'kernel' in ro:
            ro['kernel'] = _short(ro['kernel'], 48)
        ro.setdefault('traffic', None)
        if 'traffic' in r and 'traffic_source' in r:
            ro['traffic_measured_in_this_run'] = str(r['traffic_source']).startswith('measured')
        rows = [x for x in (r.get('kernels') or []) if isinstance(x, dict) and 'frac' in x]
        if rows:
            worst = min((x for x in rows if x.get('n', 0) >= 1 << 24), key=lambda x: x['frac'], default=None)
            if worst:
                ro['worst_kernel'] = _short(worst.get('name'), 48)
                ro['worst_kernel_frac'] = _num(worst['frac'])
            ro['kernel_rows'] = len(rows)
        out['roofline'] = ro
        if rows:
            # one number per row, keyed by the row's tag (its first word: K1, K2, K6m, PK, ...); a tag that repeats keeps its worst.
            # Rows below 16 Mi elements (the 1 M-parameter CIFAR student: one 8 us launch) are launch-bound, a fraction of the HBM
            # peak says nothing about them: full record only, as for worst_kernel
            kf = {}
            for x in rows:
                if x.get('n', 0) < 1 << 24:
                    continue
                tag = str(x.get('name', '?')).split(' ')[0][:6]
                kf[tag] = min(kf.get(tag, 9.0), _num(x['frac'], 3))
            out['kernels_frac'] = kf
    c = line.get('cpu_baseline')
    if isinstance(c, dict):
        co = {}
        for k in ('value', 'unit', 'cores', 'kind', 'cpu_model', 'os_cpu_count', 'threads'):
            if k in c:
                co[k] = c[k] if isinstance(c[k], str) else _num(c[k], 3)
        if 'threads' not in co and 'cores' in co:
            co['threads'] = co['cores']
        if 'error' in c:
            co['error'] = _short(c['error'], 160)
        co['sample'] = _short(c.get('sample_short') or c.get('sample'), 160)
        out['cpu_baseline'] = co
    elif 'cpu_baseline' in line:
        out['cpu_baseline'] = None
    for k in ('parity_bit_exact_vs_reference', 'parity_bit_exact_vs_oracle', 'rccl_world_size'):
        if k in line:
            out[k] = line[k]
    if line.get('rccl_error'):
        out['rccl_error'] = _short(line['rccl_error'], 160)
    d = line.get('distill') or {}
    sps, dp = {}, {}
    cs = d.get('cifar_student')
    if isinstance(cs, dict):
        if 'multi' in cs or 'per_tensor' in cs:
            sps['cfg1_cifar_student'] = _sps(cs.get('multi'))
            sps['cfg1_per_tensor_calls'] = _sps(cs.get('per_tensor'))
            if _dp(cs.get('dp')):
                dp['cfg1'] = _dp(cs['dp'])
        else:
            sps['cfg1_cifar_student'] = _sps(cs)
            if _dp(cs):
                dp['cfg1'] = _dp(cs)
    cg = d.get('cifar_graph')
    if isinstance(cg, dict):
        sps['cfg1_hipgraph'] = _sps(cg.get('multi_graph')) if 'multi_graph' in cg else _sps(cg)
    for key, tag in (('diffquant_wrn', 'cfg2'), ('imagenet_resnet18k_dp', 'cfg3'), ('nmt_lstm_dp', 'cfg4')):
        if key in d:
            sps[tag + '_' + key.replace('_dp', '')] = _sps(d[key])
            if _dp(d[key]):
                dp[tag] = _dp(d[key])
    c0 = c.get('distill') if isinstance(c, dict) else line.get('cpu_distill')
    if isinstance(c0, dict):
        sps['cfg0_cpu_reference_quantizer'] = _sps(c0)
    if sps:
        out['steps_per_sec'] = sps
    if dp:
        out['dp'] = dp
    if d.get('legs_failed'):
        out['legs_failed'] = _short(json.dumps(d['legs_failed']), 200)
    for k in ('collective_backend', 'device', 'miopen_find_mode'):
        if line.get(k):
            out[k] = _short(line[k], 40)
    bp = line.get('bench_process')
    if isinstance(bp, dict):
        out['bench_process'] = {'workers': len(bp.get('workers') or []), 'restarts': bp.get('restarts'), 'wall_s': bp.get('wall_s'),
                                'last_exit': (bp.get('workers') or [{}])[-1].get('exit')}
        if bp.get('legs_lost_with_their_worker'):
            out['bench_process']['legs_lost'] = sorted(bp['legs_lost_with_their_worker'])
    w = line.get('legs_wall_s')
    if isinstance(w, dict) and w:
        out['legs_wall_s'] = ' '.join('%s=%s' % (k, v) for k, v in w.items())[:260]
    if line.get('detail'):
        out['detail'] = line['detail']
    return out


def fit(obj, limit=LINE_LIMIT):
    """json.dumps(obj) in at most `limit` bytes: optional groups are given up in DROP_ORDER, then every remaining string is
    shortened, then -- a record no run produces -- only the contract fields are kept."""
    obj = dict(obj)

    def dumps(o):
        return json.dumps(o, separators=(', ', ': '))
    s = dumps(obj)
    for key in DROP_ORDER:
        if len(s.encode()) <= limit:
            return s
        if key in obj:
            del obj[key]
            obj['dropped_to_fit'] = obj.get('dropped_to_fit', []) + [key]
            s = dumps(obj)
    if len(s.encode()) <= limit:
        return s

    def squeeze(o, n):
        if isinstance(o, dict):
            return {k: squeeze(v, n) for k, v in o.items()}
        if isinstance(o, list):
            return [squeeze(v, n) for v in o[:8]]
        if isinstance(o, str):
            return _short(o, n)
        return o
    for n in (80, 40, 16):
        s = dumps({k: (v if k in CONTRACT else squeeze(v, n)) for k, v in obj.items()})        # the contract fields stay verbatim
        if len(s.encode()) <= limit:
            return s
    s = dumps({k: obj[k] for k in CONTRACT if k in obj})
    return s if len(s.encode()) <= limit else dumps({'metric': _short(obj.get('metric'), 64), 'value': _num(obj.get('value')),
                                                      'error': 'line did not fit'})


def write_detail(path, line, log=None):
    """The full record, atomically, next to bench.py; a read-only tree costs the file, not the run."""
    import os
    try:
        tmp = '%s.%d.tmp' % (path, os.getpid())
        with open(tmp, 'w') as f:
            json.dump(line, f, indent=1)
            f.write('\n')
        os.replace(tmp, path)
        return True
    except OSError as e:
        if log:
            log('bench detail not written to %s: %s' % (path, e))
        return False
