cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/amr_profile.py 128 10 cluster > gpurun_out/amr_profile_now.txt 2>&1
python tools/amr_host_time.py >> gpurun_out/amr_profile_now.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/amrprof -o amr -- python tools/amr_host_time.py > gpurun_out/amr_rocprof.log 2>&1
f=$(ls -t gpurun_out/amrprof/*kernel_stats.csv | head -1); head -30 $f > gpurun_out/amr_kernel_stats_head.csv
python - <<'PY' >> gpurun_out/amr_profile_now.txt
import csv, glob
f = sorted(glob.glob('gpurun_out/amrprof/*kernel_stats.csv'))[-1]
tot = 0; calls = 0
for r in csv.DictReader(open(f)):
    tot += float(r['TotalDurationNs']); calls += int(r['Calls'])
print('rocprof: total kernel time %.1f ms in %d launches' % (tot/1e6, calls))
PY
rm -rf gpurun_out/amrprof
