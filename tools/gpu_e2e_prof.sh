#!/bin/bash
# rocprofv3 kernel stats of the raw-input end-to-end bench (input side + towers + loss)
set +e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_e2e -o e2e -- python $GRAFT_REPO_ROOT/tools/e2e_bench.py --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/e2e_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_e2e -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/e2e_kernel_stats.csv
find gpurun_out/prof_e2e -name "*kernel_trace.csv" -delete
grep -E "resample|Name" gpurun_out/e2e_kernel_stats.csv | cut -c1-160
tail -1 gpurun_out/e2e_prof.log | cut -c1-300
