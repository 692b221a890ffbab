#!/usr/bin/env python3
"""Which kernel does each kind of call reach?  (mode, bucket-size class, point count, alignment, tensor size) -> kernel(s).

The launchers of csrc/qd_transform.h / qd_reductions.hip / qd_codec.hip pick an instantiation from the call's geometry.
This tool makes that choice visible: it issues a labelled list of calls with a MARKER kernel between them (a tiny
torch.bitwise_xor, which nothing else here launches), under `rocprofv3 --kernel-trace`, and cuts the dispatch sequence at
the markers.

    python tools/dispatch_map.py --trace      on the GPU box: runs itself under rocprofv3, writes gpurun_out/dispatch_map.txt
    python tools/dispatch_map.py --calls      (child) issues the calls, writes the labels to $QD_DISPATCH_LABELS

The table goes to profiles/rNN_dispatch_map.txt; tools/launch_coverage.py says which instantiations the TESTS reach.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

MARK = 'BitwiseXor'          # substring of the marker kernel's name


def calls():
    import torch
    import quantization
    from quantized_distillation_amd import _lib, codec, ste
    from quantized_distillation_amd.multi_tensor import MultiTensorDiffQuant, MultiTensorQuantizer
    import quantization.help_functions as qhf
    dev = torch.device('cuda:0')
    lib = _lib.load()
    ma, mb = torch.ones(8, dtype=torch.int32, device=dev), torch.ones(8, dtype=torch.int32, device=dev)
    labels = []

    def case(label, fn):
        torch.bitwise_xor(ma, mb)            # a marker before and after: what a case's SETUP launches is not attributed to it
        labels.append(label)
        fn()
        torch.bitwise_xor(ma, mb)
    N = 1 << 22
    base = torch.randn(N + 64, device=dev)
    g = torch.randn(N, device=dev)
    x = base[:N]
    x4 = base[1:N + 1]                                  # starts 4 bytes into a 16-byte granule
    buckets = [1, 2, 3, 4, 7, 12, 33, 36, 50, 64, 100, 128, 250, 256, 300, 448, 449, 500, 512, 513, 1000, 1001, 1024, 2000, 2048,
               3000, 4096, 5000, 8000, 8192, 8200, 10000, 16384, 20000, 32768, 40000]
    for b in buckets:
        case('uniformQuantization s=16 bucket=%d N=4Mi' % b, lambda: quantization.uniformQuantization(x, 16, bucket_size=b))
    for b in (33, 100, 256, 1000):
        case('uniformQuantization s=16 bucket=%d N=4Mi view at +4 B' % b, lambda: quantization.uniformQuantization(x4, 16, bucket_size=b))
        case('uniformQuantization s=16 bucket=%d ragged N=4Mi-5' % b, lambda: quantization.uniformQuantization(x[:N - 5], 16, bucket_size=b))
    case('uniformQuantization s=256 bucket=256 (no level table)', lambda: quantization.uniformQuantization(x, 256, bucket_size=256))
    case('uniformQuantization s=16 bucket=256 stochastic', lambda: quantization.uniformQuantization(x, 16, bucket_size=256, stochastic_rounding=True))
    case('uniformQuantization s=16 bucket=256 subtract_mean', lambda: quantization.uniformQuantization(x, 16, bucket_size=256, subtract_mean=True))
    for n in (1000, 16384, 16388, 100000, 1 << 20, (1 << 20) + 4, N):
        case('uniformQuantization s=16 bucket=None N=%d' % n, lambda: quantization.uniformQuantization(x[:n], 16))
    case('uniformQuantization s=16 bucket=None N=1Mi in place', lambda: quantization.uniformQuantization(x[:1 << 20].clone(), 16, modify_in_place=True))
    for b in buckets:
        sf = quantization.ScalingFunction('linear', False, False, b)
        case('scale_down bucket=%d N=4Mi' % b, lambda: sf.scale_down(x))
    sf = quantization.ScalingFunction('linear', False, False, 256)
    u = sf.scale_down(x)
    case('inv_scale_down bucket=256', lambda: sf.inv_scale_down(u))
    sfn = quantization.ScalingFunction('linear', False, False, None)
    case('scale_down bucket=None N=4Mi', lambda: sfn.scale_down(x))
    case('idx_min_rows (lazy arg indices) bucket=256', lambda: quantization.uniformQuantization(x, 16, bucket_size=256)[1].idx_min_rows)
    case('idx_min_rows bucket=None', lambda: quantization.uniformQuantization(x, 16)[1].idx_min_rows)
    for k in (4, 16, 33, 64, 65, 256, 1000):
        pts = torch.sort(torch.rand(k, device=dev))[0]
        for b in (256, 100, 33, 1000, 5000, None):
            case('nonUniformQuantization k=%d bucket=%s (raw x, int64 idx)' % (k, b), lambda: quantization.nonUniformQuantization(x, pts, bucket_size=b))
        for b in (256, 100, 33, None):
            fn = quantization.nonUniformQuantization_variable(bucket_size=b, pre_process_tensors=True, tensor=x)
            case('diff-quant forward k=%d bucket=%s (u resident, u8/i64 idx)' % (k, b), lambda: fn.forward(None, pts))
            case('diff-quant backward (point gradient) k=%d bucket=%s' % (k, b), lambda: fn.backward(g))
    for k in (4, 16, 64, 128, 256):
        idx64 = torch.randint(0, k, (N,), device=dev)
        a = torch.rand(N // 256, device=dev)
        out = torch.empty(k, device=dev)
        ws = _lib.workspace(dev)
        case('qd_point_grad_f32 int64 idx k=%d bucket=256' % k,
             lambda: lib.qd_point_grad_f32(g.data_ptr(), idx64.data_ptr(), 8, a.data_ptr(), N, 256, k, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    fq = quantization.uniformQuantization_variable(16, bucket_size=256)
    fq.saved_for_backward = {'input': x}
    case("'complicated' STE backward bucket=256", lambda: fq.backward(g))
    for b in (100, 33, 1000):
        f2 = quantization.uniformQuantization_variable(16, bucket_size=b)
        f2.saved_for_backward = {'input': x}
        case("'complicated' STE backward bucket=%d" % b, lambda: f2.backward(g))
    w = x.clone()
    case('ste.clamp_', lambda: ste.clamp_(w, 1.0))
    case('ste.truncated_ste_', lambda: ste.truncated_ste_(g.clone(), x, 1.0))
    for s, b in ((16, 256), (4, 256), (2, 256), (256, 256), (16, 64), (16, 2048), (16, 100), (16, None)):
        case('codec.pack_uniform s=%d bucket=%s' % (s, b), lambda: codec.pack_uniform(x, s, b))
        pk = codec.pack_uniform(x, s, b)
        case('PackedUniform.unpack s=%d bucket=%s' % (s, b), lambda: pk.unpack())
    lev = torch.randint(0, 16, (N,), dtype=torch.uint8, device=dev)
    for k in (16, 256):
        case('codec.histogram_u8 k=%d' % k, lambda: codec.histogram_u8(lev, k))
    case('codec.level_histogram s=16 bucket=256 (one pass)', lambda: codec.level_histogram(x, 16, 256))
    case('codec.level_histogram s=16 bucket=100 (q-writing form)', lambda: codec.level_histogram(x, 16, 100))
    case('get_huffman_encoding_mean_bit_length uniform s=16 bucket=256',
         lambda: qhf.get_huffman_encoding_mean_bit_length(iter([x]), lambda t: quantization.uniformQuantization(t, 16, bucket_size=256), 'uniform', s=16))
    pts4 = [0.0, 0.3, 0.7, 1.0]
    case('get_huffman_encoding_mean_bit_length nonuniform k=4 bucket=256',
         lambda: qhf.get_huffman_encoding_mean_bit_length(iter([x]), lambda t: quantization.nonUniformQuantization(t, pts4, bucket_size=256), 'nonuniform'))
    for k in (4, 16, 100):
        case('initialize_quantization_points k=%d bucket=256' % k,
             lambda: qhf.initialize_quantization_points(x, quantization.ScalingFunction('linear', False, False, 256), k))
    for kind in ('absmax', 'absnorm'):
        for b in (256, None):
            case('uniformQuantization type_of_scaling=%s bucket=%s' % (kind, b), lambda: quantization.uniformQuantization(x, 16, kind, bucket_size=b))
    from harness import kernel_bench
    shapes = kernel_bench.model_shapes('student')
    masters = [torch.randn(*s, device=dev) for s in shapes]
    mt = MultiTensorQuantizer(masters, 16, 256)
    case('MultiTensorQuantizer.quantize bucket=256 (CIFAR student)', lambda: mt.quantize())
    mtg = MultiTensorQuantizer(masters, 16, None)
    case('MultiTensorQuantizer.quantize bucket=None (CIFAR student)', lambda: mtg.quantize())
    qs, gr = [torch.empty_like(m) for m in masters], [torch.randn_like(m) for m in masters]
    for k in (4, 16, 64):
        mdq = MultiTensorDiffQuant(masters, qs, gr, k, 256)
        ptsm = torch.sort(torch.rand(len(masters), k, device=dev), dim=1)[0].contiguous()
        case('MultiTensorDiffQuant.forward k=%d bucket=256' % k, lambda: mdq.forward(ptsm))
        case('MultiTensorDiffQuant.backward k=%d bucket=256' % k, lambda: mdq.backward())
    torch.cuda.synchronize()
    with open(os.environ['QD_DISPATCH_LABELS'], 'w') as f:
        json.dump(labels, f)


def trace():
    from launch_coverage import short_name
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    out_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory(dir='/tmp') as td:
        lab = os.path.join(td, 'labels.json')
        env = dict(os.environ, TMPDIR='/tmp', QD_DISPATCH_LABELS=lab)
        cmd = [exe, '--kernel-trace', '--output-format', 'csv', '-d', td, '-o', 'dm', '--', sys.executable, os.path.abspath(__file__), '--calls']
        r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 or not os.path.exists(lab):
            print(r.stdout[-3000:])
            return 1
        labels = json.load(open(lab))
        rows = []
        for f in glob.glob(os.path.join(td, '**', '*kernel_trace.csv'), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    rows.append((int(row['Start_Timestamp']), row['Kernel_Name']))
    rows.sort()
    spans, cur = [], None                    # kernels between consecutive markers: inside a case, between two cases, inside, ...
    for _t, name in rows:
        if MARK in name:
            if cur is not None:
                spans.append(cur)
            cur = []
        elif cur is not None:
            cur.append(short_name(name))
    groups = spans[0::2]                     # every case is bracketed by two markers
    if len(groups) != len(labels):
        print('marker count %d != cases %d' % (len(groups), len(labels)))
    lines = ['# call -> kernels dispatched (in order; xN = N consecutive dispatches), traced with rocprofv3 --kernel-trace', '']
    for lab_, ks in zip(labels, groups):
        folded = []
        for k in ks:
            if folded and folded[-1][0] == k:
                folded[-1][1] += 1
            else:
                folded.append([k, 1])
        lines.append('%-78s %s' % (lab_, ' ; '.join(k if c == 1 else '%s x%d' % (k, c) for k, c in folded)))
    txt = '\n'.join(lines) + '\n'
    with open(os.path.join(out_dir, 'dispatch_map.txt'), 'w') as f:
        f.write(txt)
    print(txt[:5000])
    return 0


if __name__ == '__main__':
    if '--calls' in sys.argv:
        calls()
    elif '--trace' in sys.argv:
        sys.exit(trace())
    else:
        print(__doc__)
