cd $GRAFT_REPO_ROOT
export CASTRO_AMD_BENCH_BACKEND=gloo
for n in 4 8; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 4 --warmup 2 --ncell 64 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.read().strip()
d=json.loads(l); print(d['n_gpus'], d['config']['rank_grid'], d['ms_per_step'], d['config']['sim_time'], d['config']['host_free_steps'])"
done
python bench.py --ncell 64 --steps 4 --warmup 2 --no-cpu-baseline --no-contract-leg --stepwise 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(1, d['config']['sim_time'])"
