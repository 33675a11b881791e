cd $GRAFT_REPO_ROOT
python -X faulthandler -m pytest tests/ -x -q -m gpu > gpurun_out/r03o_pytest_full.log 2>&1
tail -4 gpurun_out/r03o_pytest_full.log
bash tools/ab_variants.sh 2>&1 | tee gpurun_out/r03o_ab_hw_minmax.log
python tools/fuzz_parity.py 1500 77 2>&1 | tail -2
python tools/fuzz_driver.py 150 78 2>&1 | tail -2
