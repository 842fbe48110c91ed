#!/bin/bash
# PMC of the dominant kernel (grouped MLP-up GEMM of both towers, ViT [50432 x 3072 x 768] + text [19712 x 2048 x 512], + bias + QuickGELU): HBM bytes (FETCH_SIZE and WRITE_SIZE need
# separate passes: TCC has 4 slots, they cost 3 + 2), L2 hit rate, MFMA busy, waits.  Writes gpurun_out/r02_pmc_dominant.txt + .json
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
rm -rf /tmp/pmc_dom
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_dom/$tag -o p -- python $GRAFT_REPO_ROOT/tools/one_gemm.py 50432 3072 768 1 0 0 8 19712 2048 512 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmc_dom > $O/r02_pmc_dominant.txt 2>&1
python - <<'PY'
import csv, json, collections, pathlib
acc = collections.defaultdict(list)
for f in pathlib.Path("/tmp/pmc_dom").rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16_nt_kernel_ppg" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = m.get("FETCH_SIZE", 0) * 1024, m.get("WRITE_SIZE", 0) * 1024
alg = sum((M * K + N * K) * 2 + M * N * 2 for M, N, K in [(50432, 3072, 768), (19712, 2048, 512)])
out = {"kernel": "gemm_bf16_nt_kernel_ppg<false,1,8> grouped MLP-up, ViT [50432x3072x768] + text [19712x2048x512] (+bias, QuickGELU), r02",
       "source": "tools/gpu_pmc_dominant.sh (rocprofv3 --pmc, one counter group per run, mean over 8 dispatches)",
       "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
       "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md HBM section) -> doubled; WRITE_SIZE as reported",
       "hbm_bytes_per_launch": int(2 * fetch + write), "algorithmic_bytes_per_launch": alg,
       # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
       "mfma_busy_frac": (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)) if "GRBM_GUI_ACTIVE" in m else None,
       "tcc_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
       "wait_any_frac": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
       "wait_inst_frac": m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
       "raw": m}
json.dump(out, open("gpurun_out/r02_pmc_dominant.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "raw"}, indent=1))
PY
