cd $GRAFT_REPO_ROOT
bash tools/profile_same_box.sh r03k
bash tools/profile.sh r03k
tail -4 gpurun_out/r03k_summary.txt
bash tools/pmc.sh r03k > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/r03k_pmc_summary.txt
for n in 128 64; do python bench.py --ncell $n --steps 40 --warmup 5 --no-cpu-baseline --no-contract-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n=$n ms/step %.3f value %.4g' % (d['ms_per_step'], d['value']))"; done
python bench.py --ncell 512 --steps 6 --warmup 2 --no-cpu-baseline --no-contract-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n=512 ms/step %.3f value %.4g' % (d['ms_per_step'], d['value']))"
