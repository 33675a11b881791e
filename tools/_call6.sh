cd $GRAFT_REPO_ROOT
python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q -k "host_free or fused_clean or clean_state_and_bc or hipgraph or capturable or 128_cubed" > gpurun_out/r03f_pytest.log 2>&1
tail -15 gpurun_out/r03f_pytest.log
for n in 64 128 256; do
for mode in "" "--stepwise"; do
python bench.py --ncell $n --steps 40 --warmup 5 --no-cpu-baseline --no-contract-leg $mode 2>gpurun_out/r03f_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=sum(b['ms_per_step'] for a, b in d['roofline']['kernel_utilisation'].items()); print('n=$n [$mode] ms/step %.3f  sum of kernels %.3f  host_free %s graph %s' % (d['ms_per_step'], k, d['config']['host_free_steps'], d['config']['step_graph']))" || tail -5 gpurun_out/r03f_err.log
done; done 2>&1 | tee gpurun_out/r03f_host_free.log
