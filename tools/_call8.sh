cd $GRAFT_REPO_ROOT
python -X faulthandler -m pytest tests/ -x -q -m gpu > gpurun_out/r03g_pytest_full.log 2>&1
tail -8 gpurun_out/r03g_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()"
