"""A stand-in for bench.py with the same two-process shape (harness/guardian.py) and no GPU: tests/test_guardian.py.

    python fake_bench.py [--die-in LEG --how abort|exit|hang|raise] [--dist]     the guardian
    python fake_bench.py --worker ...                                            the worker it starts

Legs: headline, a, b, c.  The guardian prints the COMPACT line (harness/report.py) and writes the full record to --detail.
--fat: the headline leg fills the record with a real full-size one (tests/golden/bench_record_full.json: round 5's 24 KB
record, the one the driver could not parse when it was printed whole).  --dist: the worker joins the gloo group of its torchrun environment (short timeout), every leg but
the headline holds an all-reduce and ends with the LegRunner agreement, and --die-rank says which rank dies.
"""
import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from harness import guardian  # noqa: E402

LEGS = ['headline', 'a', 'b', 'c']


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument('--worker', action='store_true')
    ap.add_argument('--resume', default=None)
    ap.add_argument('--die-in', default=None)
    ap.add_argument('--how', default='abort')
    ap.add_argument('--die-rank', type=int, default=0)
    ap.add_argument('--dist', action='store_true')
    ap.add_argument('--deadline-s', type=float, default=60.0)
    ap.add_argument('--world', type=int, default=None, help='pretend world size (guardian restart policy) without a launcher')
    ap.add_argument('--rank', type=int, default=None)
    ap.add_argument('--detail', default=None)
    ap.add_argument('--fat', action='store_true')
    return ap.parse_args(argv)


def die(how):
    if how == 'abort':
        os.abort()                            # SIGABRT: what std::terminate does
    if how == 'exit':
        os._exit(7)
    if how == 'hang':
        time.sleep(600)
    raise RuntimeError('worker raised on purpose')


def worker(args):
    rep = guardian.Reporter()
    rank = int(os.environ.get('RANK', '0'))
    resume = json.load(open(args.resume)) if args.resume else None
    done = list(resume['done']) if resume else []
    line = resume['line'] if resume else {'metric': 'fake', 'value': None, 'n_gpus': int(os.environ.get('WORLD_SIZE', '1'))}
    if resume:
        line['lost'] = resume.get('dead')
    runner = None
    if args.dist:
        import torch
        import torch.distributed as dist
        from harness import legs
        dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=4))
        runner = legs.LegRunner(ctl_timeout_s=30, log=lambda s: None)
    for leg in LEGS:
        if leg in done:
            continue
        rep.send(line, done, running=leg)
        if args.die_in == leg and rank == args.die_rank and not (resume and leg in resume['done']):
            die(args.how)
        if leg == 'headline':
            if args.fat:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bench_record_full.json')) as f:
                    line.update(json.load(f))
                line['n_gpus'] = int(os.environ.get('WORLD_SIZE', '1'))
            else:
                line['value'] = 123.0
                line['roofline'] = {'frac': 0.8}
        elif runner is not None:
            def body():
                t = torch.ones(2)
                dist.all_reduce(t)
                return float(t[0])
            line[leg] = runner.run(leg, body)
        else:
            line[leg] = 'ok'
        done.append(leg)
        rep.send(line, done, running=None)
    if rep.fd is None:
        from harness import report
        print(report.fit(report.compact(line)), flush=True)
    if runner is not None:
        runner.barrier()
        os._exit(0)


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.worker:
        worker(args)
        return 0
    world = args.world if args.world is not None else int(os.environ.get('WORLD_SIZE', '1'))
    rank = args.rank if args.rank is not None else int(os.environ.get('RANK', '0'))

    def cmd(extra):
        return [sys.executable, os.path.abspath(__file__), '--worker'] + argv + list(extra)
    return guardian.supervise(cmd, LEGS, rank=rank, world=world, wall_limit_s=args.deadline_s, detail_path=args.detail)


if __name__ == '__main__':
    sys.exit(main())
