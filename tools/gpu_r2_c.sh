#!/bin/bash
# round 2, GPU call C: LN-fold bit-exactness fix check + per-kernel stats fold vs no fold
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_lnfold.py tests/test_gpu_models.py -q -m gpu > $O/r2c_pytest.log 2>&1; tail -4 $O/r2c_pytest.log
for mode in 1 0; do
  cd /tmp && MMAMD_LN_FOLD=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$mode -o r2c -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-probe > $O/r2c_rocprof_fold$mode.log 2>&1
  f=$(find /tmp/prof_c$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2c_kernel_stats_fold$mode.csv
  grep '"metric"' $O/r2c_rocprof_fold$mode.log | cut -c1-230
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for mode in (1, 0):
    rows = list(csv.DictReader(open(f"gpurun_out/r2c_kernel_stats_fold{mode}.csv")))
    print("fold", mode)
    for r in rows[:16]:
        print("  %-95s calls %6s avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"][:95], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
