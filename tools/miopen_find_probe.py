"""Wall time of the first-use convolution plan search per MIOPEN_FIND_MODE, and the steps/sec that follow, for the three
steps/sec legs that run convolutions / LSTM (bench.py: diffquant_wrn, imagenet_resnet18k_dp, nmt_lstm_dp).  One child process
per mode (MIOpen reads the variable once); each child sees a cold user cache (MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR
point at a fresh directory), which is what a fresh box gives the driver."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    import torch
    from harness import bench_legs as bl
    dev = torch.device('cuda', 0)
    out = {}
    for name, fn in (('diffquant_wrn', lambda: bl.diffquant_steps_per_sec(dev, 0, 1, False, lambda: None)),
                     ('imagenet', lambda: bl.dp_config_steps_per_sec('imagenet', dev, 0, 1, False, lambda: None)),
                     ('nmt', lambda: bl.dp_config_steps_per_sec('nmt', dev, 0, 1, False, lambda: None))):
        t0 = time.time()
        r = fn()
        out[name] = {'wall_s': round(time.time() - t0, 1), 'steps_per_sec': r.get('steps_per_sec'),
                     'warmup_s': [v for k, v in r.items() if k.startswith('warmup_s')][0]}
    print('RESULT ' + json.dumps(out), flush=True)
    sys.exit(0)

for mode in sys.argv[1:] or ['default', '2', '3', '5', '1']:
    env = dict(os.environ)
    d = tempfile.mkdtemp(prefix='miopen_cold_')
    env['MIOPEN_USER_DB_PATH'] = d
    env['MIOPEN_CUSTOM_CACHE_DIR'] = d
    if mode != 'default':
        env['MIOPEN_FIND_MODE'] = mode
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    res = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
    print('MIOPEN_FIND_MODE=%s rc=%d wall=%.1f s  %s' % (mode, p.returncode, time.time() - t0, res[-1][7:] if res else p.stdout[-800:]), flush=True)
