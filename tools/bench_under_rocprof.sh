cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/rp1 gpurun_out/rp2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rp1 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --quick --skip-legs diffquant_wrn,imagenet_resnet18k_dp,nmt_lstm_dp > $GRAFT_REPO_ROOT/gpurun_out/rp1.json 2> $GRAFT_REPO_ROOT/gpurun_out/rp1.err); echo "N=1 under rocprofv3 rc=$? lines=$(wc -l < gpurun_out/rp1.json)"
python -c "
import json; d=json.loads(open('gpurun_out/rp1.json').read().strip().splitlines()[-1]); print(d.get('value'), d.get('error'), d.get('bench_process'), d.get('legs_wall_s'))"
grep -v "amdgpu.ids\|simple_timer\|output_stream\|Warning\|warn" gpurun_out/rp1.err | tail -15 | cut -c1-250
ls gpurun_out/rp1 | head
(cd /tmp && QD_BENCH_BACKEND=gloo QD_BENCH_ONE_GPU=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rp2 -o b -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 5 --warmup 2 --quick --no-cpu-baseline --no-pmc --no-kernels --precondition-s 0.05 --skip-legs diffquant_wrn,nmt_lstm_dp > $GRAFT_REPO_ROOT/gpurun_out/rp2.json 2> $GRAFT_REPO_ROOT/gpurun_out/rp2.err); echo "N=2 gloo under rocprofv3 rc=$? lines=$(wc -l < gpurun_out/rp2.json)"
grep -v "amdgpu.ids\|simple_timer\|output_stream\|Warning\|warn" gpurun_out/rp2.err | tail -25 | cut -c1-250
find gpurun_out/rp1 gpurun_out/rp2 -name '*.csv' -size +2M -delete
