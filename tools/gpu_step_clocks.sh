#!/bin/bash
# Effective clock of every kernel of the headline step, in the step: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / 8 / duration per dispatch
# (rocprofv3 --pmc with --kernel-trace).  Writes gpurun_out/r04_step_clocks.txt
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && rm -rf /tmp/pmc_clk && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_clk -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-sample 0 --no-probe > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/r04_step_clocks.txt
import csv, collections, pathlib
cnt, dur = {}, {}
for f in pathlib.Path("/tmp/pmc_clk").rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
for f in pathlib.Path("/tmp/pmc_clk").rglob("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(list)
for d, (name, c) in cnt.items():
    if d in dur and dur[d] > 5.0:
        acc[name[:70]].append((c / 8 / dur[d] / 1e3, dur[d]))
print("# effective clock (GHz) = GRBM_GUI_ACTIVE / 8 XCDs / duration, per kernel of the headline step, IN the step (bench.py under rocprofv3 --pmc; nominal 2.4)")
print("# kernel | dispatches | mean clock | mean us")
for name, v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if len(v) >= 3:
        print(f"{name:70s} {len(v):5d}  {sum(x[0] for x in v) / len(v):5.2f} GHz  {sum(x[1] for x in v) / len(v):8.1f} us")
PY
cat $O/r04_step_clocks.txt
