#!/bin/bash
# PMC of the fp32-residual grouped GEMM gemm_bf16_nt_kernel_ppg<true,0,8> (37.5 % of the r03 step; never looked at with counters before r04), both shapes it
# runs in the step: out-projection (ViT [50432 x 768 x 768] + text [19712 x 512 x 512]) and MLP-down (ViT [50432 x 768 x 3072] + text [19712 x 512 x 2048]),
# + bias + fp32 residual read-modify-write.  FETCH_SIZE and WRITE_SIZE need separate passes (TCC: 4 slots, they cost 3 + 2).
# Writes gpurun_out/pmc_residual_kernel.json
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
rm -rf /tmp/pmc_res
for shape in "outproj 768 768 512 512" "mlpdown 768 3072 512 2048"; do
  set -- $shape
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_res/$1/$tag -o p -- python $GRAFT_REPO_ROOT/tools/one_gemm.py 50432 $2 $3 0 1 0 8 19712 $4 $5 > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, json, collections, pathlib
res = {}
shapes = {"outproj": [(50432, 768, 768), (19712, 512, 512)], "mlpdown": [(50432, 768, 3072), (19712, 512, 2048)]}
for name, probs in shapes.items():
    acc = collections.defaultdict(list)
    dur = []
    for f in pathlib.Path(f"/tmp/pmc_res/{name}").rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_nt_kernel_ppg" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in pathlib.Path(f"/tmp/pmc_res/{name}").rglob("*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_nt_kernel_ppg" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    fetch, write = m.get("FETCH_SIZE", 0) * 1024, m.get("WRITE_SIZE", 0) * 1024
    alg = sum((M * K + N * K) * 2 + 2 * M * N * 4 for M, N, K in probs)  # operands bf16 + fp32 residual read + fp32 write
    flops = sum(2 * M * N * K for M, N, K in probs)
    dur.sort()
    med = dur[len(dur) // 2] if dur else None
    res[name] = {"problems_MNK": probs, "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
                 "hbm_bytes_per_launch": int(2 * fetch + write), "algorithmic_bytes_per_launch": alg, "flops_per_launch": flops,
                 "median_us_under_profiler": med, "tflops_under_profiler": flops / med / 1e6 if med else None,
                 "mfma_busy_frac": (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)) if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m else None,
                 "tcc_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
                 "effective_clock_ghz": (m["GRBM_GUI_ACTIVE"] / 8 / med / 1e3) if "GRBM_GUI_ACTIVE" in m and med else None,  # busy cycles per XCD / duration (2.4 nominal)
                 "wait_any_frac": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
                 "wait_inst_frac": m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
                 "active_inst_frac": m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
                 "lds_bank_conflict_frac": m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else None,
                 "raw": m}
out = {"kernel": "gemm_bf16_nt_kernel_ppg<true,0>: grouped out-projection / MLP-down of both towers, + bias + fp32 residual read-modify-write (pipelined buffer-descriptor epilogue, r06)",
       "source": "tools/gpu_pmc_residual.sh (rocprofv3 --pmc, one counter group per run, mean over 8 dispatches; isolated launches on random operands)",
       "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md HBM section) -> doubled; WRITE_SIZE as reported",
       "shapes": res}
json.dump(out, open("gpurun_out/pmc_residual_kernel.json", "w"), indent=1)
for k, v in res.items():
    print(k, json.dumps({a: b for a, b in v.items() if a != "raw"}))
PY
