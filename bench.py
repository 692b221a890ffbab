#!/usr/bin/env python3
"""bench.py -- headline measurement of the fake-quantization hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch of synthetic input: one call of
quantization.uniformQuantization(x, s=16, bucket_size=256) on a 64 Mi-element fp32 tensor
(BASELINE.json: "quantize-dequantize GB/s (64M fp32, 4-bit)"), through the public Python API
(allocation of the result + one HIP kernel launch through the C ABI).  Inputs are resident in
HBM; four input tensors and four live outputs are rotated so that every call streams 512 MiB
through HBM and cannot be served from the 256 MiB Infinity Cache.

Two processes per rank (harness/guardian.py): the process the driver starts is a GUARDIAN -- plain
Python, no torch, no HIP -- that starts the WORKER (`bench.py --worker`, where everything below
is measured), receives a snapshot of the line after every leg, and prints the last one, exactly
once, when the worker has ended -- however it ended.  Round 4's driver run was lost to a SIGABRT
in an optional leg 50 s after the headline had been measured; now a dying leg costs that leg: the
guardian records it, starts a fresh worker for the legs that are left (N = 1) and prints the line.

Order of the legs (N = 1: the whole default run takes ~65 s on a fresh box; the steps/sec legs run with MIOpen's FAST find mode,
harness/legs.py miopen_env_defaults -- the default mode's first-use search for the WideResNet shapes alone took 45 s):
  headline        the timed region of the contract (+ the kernel's HIP-event time -> roofline)
  rocprof         the same kernel under rocprofv3 in child processes: --kernel-trace duration, PMC HBM traffic
  cpu_baseline    the REFERENCE's own uniformQuantization on the host cores, same workload; checker of the GPU result
      ---- from here on the guardian holds a line with value, roofline and cpu_baseline ----
  kernels         roofline.kernels: one row per kernel on the path (SURVEY 8d's secondary rows, K2 ... K9)
  cifar_student   distilled steps/sec, BASELINE configs[1], eager, + the data-parallel report
  optional, each started only while the wall budget (--budget-s) lasts:
  cifar_graph (hipGraph replay; N = 1 only), pcie_note, diffquant_wrn (configs[2]),
  imagenet_resnet18k_dp (configs[3]), nmt_lstm_dp (configs[4]), cpu_distill (configs[0] on the CPU)

Multi-GPU (one process per GPU): the path shards by tensor -- every rank quantizes its own
tensors, no collective in the data path -- so scaling is "weak" and value = total bytes of all
ranks / max-over-ranks time.  `python bench.py --gpus N` started WITHOUT a torchrun environment
launches the N ranks itself (harness/launch.py re-executes this file under
torch.distributed.run, 127.0.0.1 rendezvous, backend "nccl" = RCCL); started under torchrun it
uses the ranks it was given.  At N = 1 a single-rank RCCL group is created when the first
steps/sec leg starts (not before: the headline needs none), so the gradient all-reduce of those
legs really runs through RCCL on a one-GPU box.

Rank 0's guardian prints ONE JSON line of at most 4096 bytes (harness/report.py) -- the contract fields, `config`,
  roofline      achieved algorithmic GB/s of the dominant kernel (8 B/element: 4 read + 4 written) from
                HIP-event timing of the timed region against the 8 TB/s peak, the kernel's rocprofv3 duration and PMC
                traffic measured in the same run, the worst other kernel row
  cpu_baseline  the REFERENCE's own uniformQuantization (staged bytecode of /root/reference/quantization,
                oracle/ref_stage.py) timed on the host cores of this box on the same workload
  steps_per_sec / dp   one number per BASELINE config, and the data-parallel figures of each
-- and writes the full record (every kernel row, the legs' sub-records, sample descriptions, what the guardian saw of
its workers) to bench_detail.json next to this file (--detail PATH) and to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the default run, in order; everything after 'cifar_student' is optional (wall budget)
LEGS = ['headline', 'rocprof', 'cpu_baseline', 'kernels', 'cifar_student',
        'cifar_graph', 'pcie_note', 'diffquant_wrn', 'imagenet_resnet18k_dp', 'nmt_lstm_dp', 'cpu_distill']
OPTIONAL = {'cifar_graph': 4, 'pcie_note': 1, 'diffquant_wrn': 12, 'imagenet_resnet18k_dp': 4, 'nmt_lstm_dp': 9, 'cpu_distill': 12}
#            ^ seconds a leg is expected to take on an MI355X box (profiles/r06_bench*.json legs_wall_s): it starts only if that fits the budget
DISTILL_LEGS = ('cifar_student', 'cifar_graph', 'diffquant_wrn', 'imagenet_resnet18k_dp', 'nmt_lstm_dp')


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernels', action='store_true', help='skip the per-kernel roofline rows (roofline.kernels)')
    ap.add_argument('--no-distill', action='store_true', help='skip every steps/sec leg')
    ap.add_argument('--no-diffquant', action='store_true', help='skip the WideResNet differentiable-quantization leg')
    ap.add_argument('--no-dp-configs', action='store_true', help='skip the ImageNet-shaped and seq2seq steps/sec legs')
    ap.add_argument('--precondition-s', type=float, default=0.4, help='seconds of untimed back-to-back launches before warm-up')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 child processes (kernel duration + PMC HBM traffic of the headline kernel)')
    ap.add_argument('--budget-s', type=float, default=None,
                    help='wall budget of the run: an OPTIONAL leg starts only if its expected duration still fits (0: no limit; default: '
                         '100 s at N = 1, 240 s at N > 1, where the steps/sec legs are what the run is for and rendezvous takes its time)')
    ap.add_argument('--deadline-s', type=float, default=1500.0,
                    help='after this many seconds the guardian ends the worker and prints the line with what has been measured so far')
    ap.add_argument('--skip-legs', default='', help='comma-separated legs to leave out (%s)' % ', '.join(LEGS[1:]))
    ap.add_argument('--graph-at-any-n', action='store_true', help='run the hipGraph replay leg at N > 1 too (default: N = 1 only)')
    ap.add_argument('--cpu-distill-steps', type=int, default=60, help='steps of the configs[0] CPU leg (200 = the whole synthetic epoch)')
    ap.add_argument('--quick', action='store_true', help='short steps/sec legs (a few steps, two repetitions): for exercising the flow, not for numbers')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'bench_detail.json'), help='where rank 0 writes the full record (stdout carries the compact line)')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)          # the measuring process (started by the guardian)
    ap.add_argument('--resume', default=None, help=argparse.SUPPRESS)                 # guardian -> fresh worker: the line so far + the legs done
    ap.add_argument('--pmc-slice', action='store_true', help=argparse.SUPPRESS)       # child mode of measure_pmc_traffic()
    ap.add_argument('--pmc-slice-launches', type=int, default=12, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def disabled_legs(args, n_gpus):
    off = set(x for x in args.skip_legs.split(',') if x)
    if args.no_cpu_baseline:
        off |= {'cpu_baseline', 'cpu_distill'}
    if args.no_kernels:
        off.add('kernels')
    if args.no_pmc:
        off.add('rocprof')
    if args.no_distill:
        off |= set(DISTILL_LEGS)
    if args.no_diffquant:
        off.add('diffquant_wrn')
    if args.no_dp_configs:
        off |= {'imagenet_resnet18k_dp', 'nmt_lstm_dp'}
    if n_gpus > 1:
        off |= {'rocprof', 'cpu_baseline', 'pcie_note', 'cpu_distill'}     # rank 0 at N = 1 only (contract: cpu_baseline on rank 0 at N=1)
        if not args.graph_at_any_n:
            off.add('cifar_graph')
    return off


# ---------------------------------------------------------------------------------------------- guardian
def visible_gpus():
    """HIP devices visible to a python process -- asked of a child, so that this process never loads HIP."""
    import subprocess
    try:
        p = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)'],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, text=True)
        return int(p.stdout.strip().splitlines()[-1])
    except Exception:                                             # noqa: BLE001
        return 0


def guardian_main(args, argv):
    from harness import guardian, launch                          # stdlib only: no torch, no HIP in this process
    if args.gpus > 1 and not launch.under_launcher():
        # `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, RCCL over xGMI; every rank is
        # again a guardian + worker pair
        have = visible_gpus()
        if have < args.gpus:
            raise SystemExit('bench.py --gpus %d: only %d HIP device(s) visible' % (args.gpus, have))
        sys.stdout.flush()
        raise SystemExit(launch.run_ranks(os.path.abspath(__file__), args.gpus, argv))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    off = disabled_legs(args, world)

    def worker_cmd(extra):
        return [sys.executable, os.path.abspath(__file__), '--worker'] + list(argv) + list(extra)
    return guardian.supervise(worker_cmd, [x for x in LEGS if x not in off], rank=rank, world=world, wall_limit_s=args.deadline_s,
                              detail_path=args.detail)


# ---------------------------------------------------------------------------------------------- worker
def worker_main(args):
    import torch
    import torch.distributed as dist
    from harness import bench_legs as bl
    from harness import guardian, launch, legs
    t_start = time.time()
    legs.rccl_env_defaults()                 # (before the first HIP call: the runtime reads HSA_ENABLE_IPC_MODE_LEGACY when it starts)
    miopen_find_mode = legs.miopen_env_defaults()
    reporter = guardian.Reporter()
    N_ELEM, LEVELS, BUCKET, N_ROTATE = bl.N_ELEM, bl.LEVELS, bl.BUCKET, bl.N_ROTATE
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d\n' % (args.gpus, world, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # QD_BENCH_BACKEND=gloo + QD_BENCH_ONE_GPU=1: the WHOLE multi-rank flow (per-leg agreement, the data-parallel report with
    # rank 0 alone, the closing barriers) on a one-GPU box -- every rank on device 0, collectives through gloo instead of RCCL
    # (tests/test_hip_bench_ranks.py).  The driver never sets these.
    backend = os.environ.get('QD_BENCH_BACKEND', 'nccl')
    if os.environ.get('QD_BENCH_ONE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    n_gpus = world
    distributed = world > 1                  # the timed region's barriers: only where there is somebody to wait for
    group = {'error': None, 'world': None}

    def ensure_group():
        """The data-path process group.  N > 1: created up front (the timed region's barrier needs it).  N = 1: a one-rank
        RCCL group, created when the first steps/sec leg starts; QD_FORCE_DIST=1 makes the harness issue its collectives in
        it, so that the all-reduce path of those legs is executed and timed on this box."""
        if dist.is_initialized() or group['error']:
            return
        try:
            if launch.under_launcher():
                if backend == 'nccl':
                    dist.init_process_group('nccl', device_id=dev, timeout=legs.data_timeout())    # "nccl" is RCCL on ROCm
                else:
                    dist.init_process_group(backend, timeout=legs.data_timeout())
            else:
                dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % launch.free_port(), rank=0, world_size=1,
                                        device_id=dev, timeout=legs.data_timeout())
            if world == 1:
                os.environ['QD_FORCE_DIST'] = '1'
            group['world'] = dist.get_world_size()
        except Exception as e:                                 # noqa: BLE001 -- keep the headline measurement
            if world > 1:
                raise
            group['error'] = '%s: %s' % (type(e).__name__, e)
        line['rccl_world_size'], line['rccl_error'] = group['world'], group['error']

    off = disabled_legs(args, n_gpus)
    budget_s = args.budget_s if args.budget_s is not None else (100.0 if n_gpus == 1 else 240.0)
    resume = None
    if args.resume:
        with open(args.resume) as f:
            resume = json.load(f)
    done = list(resume['done']) if resume else []
    # the line as far as it has been measured: the guardian holds the last snapshot of it
    line = resume['line'] if resume else {
        'metric': 'quantize_dequantize_GBps_64M_fp32_4bit', 'value': None, 'unit': 'GB/s', 'n_gpus': n_gpus, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic'}
    wall = line.setdefault('legs_wall_s', {})
    if resume:
        for leg, why in (resume.get('dead') or {}).items():
            wall[leg] = 'lost'
            if leg in DISTILL_LEGS:
                line.setdefault('distill', {})
                line['distill'] = dict(line['distill'] or {}, **{leg: {'error': why}})
            else:
                line.setdefault('legs_lost', {})[leg] = why
    if distributed:
        ensure_group()
    runner = legs.LegRunner()                # the per-leg agreement runs over its own gloo group (harness/legs.py)

    def snapshot(running=None):
        line['wall_s'] = round(time.time() - t_start, 1)
        reporter.send(line, done, running, line['wall_s'])

    def want(name):
        return name not in done and name not in off

    def run_leg(name, fn, store, collective=False):
        """One leg: progress marker to the guardian, the wall budget for optional legs (rank 0 decides for everybody), the
        body through the LegRunner (an exception becomes an 'error' record on every rank), a snapshot afterwards."""
        if not want(name):
            return
        if name in OPTIONAL and budget_s > 0:
            spent = time.time() - t_start
            go = runner.rank0_says(spent + OPTIONAL[name] <= budget_s or args.quick)
            if not go:
                store({'skipped': 'wall budget: %.0f s spent + ~%d s expected > --budget-s %.0f' % (spent, OPTIONAL[name], budget_s)})
                wall[name] = 'budget'
                done.append(name)
                return
        snapshot(running=name)
        if os.environ.get('QD_BENCH_TEST_ABORT_IN') == name and not resume:        # test hook (tests/test_hip_bench_ranks.py):
            os.abort()                                                             # what a watchdog thread's std::terminate does
        t0 = time.time()
        res = runner.run(name, fn, collective=collective)
        torch.cuda.empty_cache()
        wall[name] = round(time.time() - t0, 1)
        store(res)
        done.append(name)
        snapshot()

    # ------------------------------------------------------------------ pre-flight (N > 1)
    # ONE small all-reduce through the data-path communicator before anything depends on it.  If RCCL cannot move bytes on
    # this node (no IPC path between the GPUs, a dead link) every rank learns it here, inside the group's timeout: the
    # headline then runs per GPU without the barriers of the timed region, the collective-bearing legs are skipped, and
    # rank 0's line says so -- instead of N ranks hanging in their first barrier.
    if distributed and not resume:
        pre_err = None
        try:
            if os.environ.get('QD_BENCH_TEST_PREFLIGHT_FAIL') == str(rank):      # test hook (tests/test_hip_bench_ranks.py)
                raise RuntimeError('pre-flight failure requested by the test')
            t = torch.ones(1, device=dev)
            dist.all_reduce(t)
            torch.cuda.synchronize()
            if int(t[0]) != world:
                pre_err = 'all-reduce of ones over %d ranks returned %r' % (world, float(t[0]))
        except Exception as e:                                    # noqa: BLE001
            pre_err = '%s: %s' % (type(e).__name__, e)
        failed = runner.agree(pre_err is None)
        if failed:
            group['error'] = 'pre-flight all-reduce failed on rank(s) %s%s' % (failed, ': ' + pre_err if pre_err else '')
            runner.broken = group['error']
            distributed = False              # no collective in the timed region: every rank measures its own GPU
            line['error'] = group['error'] + '; value is rank 0 alone x n_gpus (NOT a max over ranks), the steps/sec legs were skipped'

    import quantization
    from quantized_distillation_amd import _lib
    _lib.load()

    host = {}

    def x_host():
        if 'x' not in host:
            host['gen'] = torch.Generator().manual_seed(1000 * rank)
            host['x'] = torch.randn(N_ELEM, generator=host['gen'])
        return host['x']

    # ------------------------------------------------------------------ headline: the timed region of the contract
    if want('headline'):
        snapshot(running='headline')
        t_leg = time.time()
        xs = [x_host().to(dev)]
        for i in range(1, N_ROTATE):
            xs.append(torch.randn(N_ELEM, generator=host['gen']).to(dev))
        live = [None] * N_ROTATE

        def step(i):
            q, _sf = quantization.uniformQuantization(xs[i % N_ROTATE], LEVELS, bucket_size=BUCKET)
            live[i % N_ROTATE] = q            # keep the last outputs alive: rotating output buffers

        # Preconditioning (setup, untimed): from an idle GPU the first ~25 ms of back-to-back
        # launches run 10-30 % slower while the power/clock management settles
        # (docs/history/profiles/r01_sustain_probe.txt); bring the chip to its steady
        # state before the official warm-up so that short --warmup/--steps runs measure steady state.
        t_pre = time.time()
        n_pre = 0
        while time.time() - t_pre < args.precondition_s:
            for i in range(50):
                step(i)
            n_pre += 50
            torch.cuda.synchronize()

        for i in range(args.warmup):
            step(i)

        def fence():
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
                torch.cuda.synchronize()

        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            step(i)
        ev1.record()
        fence()
        elapsed = time.perf_counter() - t0
        event_ms_total = ev0.elapsed_time(ev1)

        if distributed:
            t = torch.tensor([elapsed, event_ms_total], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, event_ms_total = float(t[0]), float(t[1])

        # the same measurement over >= 200 launches whatever --steps says (the driver's --steps 20 region is 1.8 ms)
        ext_n = max(200, args.steps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(ext_n):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        ext_us = e0.elapsed_time(e1) * 1e3 / ext_n

        # reference point on this very GPU: torch's own device-to-device copy of the same 256 MiB tensors
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(3):
            live[i % N_ROTATE].copy_(xs[i % N_ROTATE])
        torch.cuda.synchronize()
        c0.record()
        for i in range(20):
            live[i % N_ROTATE].copy_(xs[i % N_ROTATE])
        c1.record()
        torch.cuda.synchronize()
        copy_gbps = bl.ALGO_BYTES_PER_ELEM * N_ELEM * 20 / (c0.elapsed_time(c1) * 1e-3) / 1e9

        # the GPU result the cpu_baseline leg checks (same tensor, seed 0 on rank 0)
        if rank == 0 and n_gpus == 1 and want('cpu_baseline'):
            q, sf = quantization.uniformQuantization(xs[0], LEVELS, bucket_size=BUCKET)
            host['q_gpu'], host['alpha_gpu'] = q.cpu().numpy(), sf.alpha.cpu().numpy().reshape(-1)
            del q, sf
        del live[:], xs[:]
        torch.cuda.empty_cache()

        bytes_per_launch = bl.ALGO_BYTES_PER_ELEM * N_ELEM
        kernel_us = event_ms_total * 1e3 / args.steps
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        committed = bl.load_pmc_traffic()
        roofline = {
            'bound': 'hbm', 'kernel': 'k_bucket_vec<0,16,4,1>',      # as rocprofv3 names it (mode 0 = quantize-dequantize; profiles/r06_bench_kernel_stats.csv)
            'achieved': round(achieved, 1), 'peak': bl.HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': round(achieved / bl.HBM_PEAK_GBPS, 4),
            'traffic': int(round(committed)) if committed else None,
            'traffic_source': 'profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on a '
                              'builder box (committed file, NOT measured in this run)',
            'traffic_committed': int(round(committed)) if committed else None,
            'extended': {'launches': ext_n, 'avg_launch_us': round(ext_us, 3),
                         'frac': round(bytes_per_launch / (ext_us * 1e-6) / 1e9 / bl.HBM_PEAK_GBPS, 4)},
            'extended_avg_launch_us': round(ext_us, 3),
            'avg_launch_us': round(kernel_us, 3), 'algorithmic_bytes_per_launch': bytes_per_launch,
            'timing': 'HIP events on the launch stream around the %d timed launches (includes inter-launch gaps)' % args.steps,
            'torch_d2d_copy_GBps': round(copy_gbps, 1),      # torch's own copy of the same bytes, same box (a hand-written NT copy reaches 6.3-6.5 TB/s: docs/history/profiles/r01_kbench.txt)
        }
        line.update({
            'value': round(bytes_per_launch * args.steps * n_gpus / elapsed / 1e9, 2),
            'ms_per_step': round(elapsed * 1e3 / args.steps, 5),
            'config': {
                'workload': 'uniformQuantization(x, s=16, bucket_size=256) on x = randn(64Mi) fp32 per GPU: 4-bit quantize-dequantize',
                'call': 'quantization.uniformQuantization(x, s=16, type_of_scaling="linear", bucket_size=256), deterministic rounding, '
                        'through the public API (result allocation + one launch through the C ABI)',
                'n_elements_per_gpu': N_ELEM, 'levels': LEVELS, 'bucket_size': BUCKET,
                'algorithmic_bytes_per_element': bl.ALGO_BYTES_PER_ELEM, 'rotating_buffers': N_ROTATE, 'precondition_launches': n_pre,
                'parallelism': 'independent tensors per rank (no data-path collective)' if n_gpus > 1 else 'single GPU',
            },
            'cpu_baseline': None, 'distill': None,
            'parity_bit_exact_vs_oracle': None, 'parity_bit_exact_vs_reference': None,
            'rccl_world_size': group['world'], 'rccl_error': group['error'], 'collective_backend': backend,
            'device': torch.cuda.get_device_name(dev), 'miopen_find_mode': miopen_find_mode,
            'roofline': roofline,
        })
        wall['headline'] = round(time.time() - t_leg, 1)
        done.append('headline')
        snapshot()
    roofline = line['roofline']
    bytes_per_launch = roofline['algorithmic_bytes_per_launch']

    # ------------------------------------------------------------------ rocprofv3 children: kernel duration + HBM traffic (rank 0, N = 1)
    def leg_rocprof():
        dur = bl.measure_rocprof_duration()
        tm = bl.measure_pmc_traffic()
        return {'dur': dur, 'traffic': tm}

    def store_rocprof(res):
        dur = (res or {}).get('dur') or {'error': (res or {}).get('error', 'not measured')}
        tm = (res or {}).get('traffic') or {'error': (res or {}).get('error', 'not measured')}
        roofline['traffic_measured'] = tm
        if tm.get('bytes_per_launch'):
            roofline['traffic'] = int(tm['bytes_per_launch'])
            roofline['traffic_source'] = 'measured in this run (traffic_measured)'
            roofline['traffic_over_algorithmic'] = tm['over_algorithmic']
            for label, rec in (tm.get('other_kernels') or {}).items():        # scalars: what the driver's record keeps
                roofline['traffic_over_algorithmic ' + label] = rec['over_algorithmic']
        roofline['rocprof_kernel'] = dur
        if dur.get('avg_us'):
            roofline['rocprof_kernel_avg_us'] = dur['avg_us']
            roofline['rocprof_kernel_launches'] = dur['launches']
            roofline['rocprof_frac'] = round(bytes_per_launch / (dur['avg_us'] * 1e-6) / 1e9 / bl.HBM_PEAK_GBPS, 4)
    run_leg('rocprof', leg_rocprof, store_rocprof)

    # ------------------------------------------------------------------ cpu_baseline (rank 0, N = 1): the reference on the host
    # cores, and in the same leg the checker of the GPU result computed above (bit-exact comparison)
    def leg_cpu_baseline():
        if 'q_gpu' not in host:                                   # a resumed worker: recompute the GPU result to be checked
            q, sf = quantization.uniformQuantization(x_host().to(dev), LEVELS, bucket_size=BUCKET)
            host['q_gpu'], host['alpha_gpu'] = q.cpu().numpy(), sf.alpha.cpu().numpy().reshape(-1)
        return bl.cpu_baseline(x_host(), host.pop('q_gpu'), host.pop('alpha_gpu'), with_ports=not args.quick)

    def store_cpu(cpu):
        if isinstance(cpu, dict) and 'error' not in cpu:
            line['parity_bit_exact_vs_oracle'] = cpu.pop('gpu_result_bit_exact', None)
            line['parity_bit_exact_vs_reference'] = cpu.pop('gpu_result_bit_exact_vs_reference', None)
        line['cpu_baseline'] = cpu
    run_leg('cpu_baseline', leg_cpu_baseline, store_cpu)
    host.pop('q_gpu', None), host.pop('alpha_gpu', None)

    # ------------------------------------------------------------------ per-kernel roofline rows (every rank its own GPU; no collective)
    def leg_kernels():
        from harness import kernel_bench
        rows = kernel_bench.run(dev)
        roofline['kernels_method'] = ('HIP events, median of %d repetitions of %d launches (12 for the 1 Gi-symbol histograms) after >= 100 ms '
                                      'of preconditioning, >= 3 rotating buffer sets' % (kernel_bench.REPS, kernel_bench.ITERS))
        return rows
    run_leg('kernels', leg_kernels, lambda rows: roofline.__setitem__('kernels', rows))

    # ------------------------------------------------------------------ steps/sec legs
    # A failure in one of them (say, an out-of-memory condition on one rank) is recorded in the JSON instead of losing
    # anything measured above, and cannot leave the other ranks inside a collective for longer than the group's timeout:
    # runner.run() wraps the leg BODY and ends every leg with an agreement over a gloo group (harness/legs.py).
    def store_distill(name):
        def store(res):
            line['distill'] = dict(line.get('distill') or {}, **{name: res})
        return store
    quick = dict(steps=3, warmup=2, reps=2) if args.quick else {}
    cifar_quick = dict(steps=20, warmup=5, repetitions=2) if args.quick else {}
    if any(want(x) for x in DISTILL_LEGS):
        ensure_group()
    run_leg('cifar_student', lambda: bl.distill_steps_per_sec(dev, rank, n_gpus, distributed, runner.barrier, **cifar_quick),
            store_distill('cifar_student'), collective=True)
    # ---- optional legs: each starts only while the wall budget lasts
    run_leg('cifar_graph', lambda: bl.distill_graph_steps_per_sec(dev, rank, n_gpus, distributed, **cifar_quick),
            store_distill('cifar_graph'), collective=True)

    def store_pcie(v):
        roofline['pcie_inclusive_GBps_note'] = v if isinstance(v, float) else None   # host buffer -> HBM -> quantize -> host buffer (never `value`)
    run_leg('pcie_note', lambda: bl.pcie_inclusive_note(x_host(), dev), store_pcie)
    host.clear()
    run_leg('diffquant_wrn', lambda: bl.diffquant_steps_per_sec(dev, rank, n_gpus, distributed, runner.barrier, **quick),
            store_distill('diffquant_wrn'), collective=True)
    # configs[3] is quoted on 8 GPUs and configs[4] on 4; both fit one GPU, so they are timed at every N
    # (weak scaling: per-GPU batch fixed) and the driver's --gpus 8 / --gpus 4 runs give BASELINE's placements
    run_leg('imagenet_resnet18k_dp', lambda: bl.dp_config_steps_per_sec('imagenet', dev, rank, n_gpus, distributed, runner.barrier, **quick),
            store_distill('imagenet_resnet18k_dp'), collective=True)
    run_leg('nmt_lstm_dp', lambda: bl.dp_config_steps_per_sec('nmt', dev, rank, n_gpus, distributed, runner.barrier, **quick),
            store_distill('nmt_lstm_dp'), collective=True)

    def store_cpu_distill(res):
        if isinstance(line.get('cpu_baseline'), dict):
            line['cpu_baseline']['distill'] = res
        else:
            line['cpu_distill'] = res
    run_leg('cpu_distill', lambda: bl.cpu_distill_baseline(steps=10 if args.quick else args.cpu_distill_steps), store_cpu_distill)
    if runner.history and isinstance(line.get('distill'), dict):
        line['distill']['legs_failed'] = [{'leg': n, 'ranks': r} for n, r in runner.history]

    snapshot()
    if reporter.fd is None and rank == 0:                          # `bench.py --worker` run by hand: no guardian to print the line
        from harness import report
        line['detail'] = os.path.basename(args.detail)
        if not report.write_detail(args.detail, line):
            del line['detail']
        print(report.fit(report.compact(line)), flush=True)
    runner.barrier()                                               # (gloo: works whatever the legs left behind)
    if dist.is_initialized():
        if runner.broken is None:
            dist.destroy_process_group()
        else:
            sys.stdout.flush(), sys.stderr.flush()
            os._exit(0)                                            # a communicator with unmatched collectives cannot be torn down cleanly


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.pmc_slice:
        from harness import bench_legs
        bench_legs.pmc_slice(args.pmc_slice_launches)
        return 0
    if args.worker:
        worker_main(args)
        return 0
    return guardian_main(args, argv)


if __name__ == '__main__':
    sys.exit(main())
