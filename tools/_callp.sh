cd $GRAFT_REPO_ROOT
python -X faulthandler -m pytest tests/ -x -q -m gpu > gpurun_out/r03p_pytest_full.log 2>&1
tail -3 gpurun_out/r03p_pytest_full.log
bash tools/profile_same_box.sh r03p
bash tools/profile.sh r03p
tail -2 gpurun_out/r03p_summary.txt
bash tools/pmc.sh r03p > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/r03p_pmc_summary.txt
