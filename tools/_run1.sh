cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export DMVIO_BENCH_SHARE_DEVICE=1 DMVIO_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --batch 256 > gpurun_out/n2.log 2>&1
grep -E "Error|error|Traceback" -A8 gpurun_out/n2.log | head -40
tail -1 gpurun_out/n2.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], json.dumps(d['ba'])[:600])"
