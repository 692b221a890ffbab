#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== full gpu suite"; timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/u_pytest.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/u_pytest.log
echo "== NEAREST / SCALE modes at 506..511"; timeout 300 python - <<'P'
import numpy as np, torch, quantization
from oracle import oracle_c as oc
for b in (506, 509, 511, 513, 1001):
    x = np.random.RandomState(b).randn(b*41+3).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    pts = torch.tensor([0.0, 0.3, 0.6, 1.0]).cuda()
    q, idx, sf = quantization.nonUniformQuantization(xd, pts, bucket_size=b)
    r = oc.nonuniform_quantize(x, pts.cpu().numpy(), b)
    assert np.array_equal(q.cpu().numpy(), r['q']) and np.array_equal(idx.cpu().numpy(), r['idx']), b
    sfn = quantization.ScalingFunction('linear', False, False, b)
    u = sfn.scale_down(xd)
    r2 = oc.scale_down(x, b)
    assert np.array_equal(u.cpu().numpy().reshape(-1), r2['u'].reshape(-1)), b
print('ok')
P
