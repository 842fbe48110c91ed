#!/bin/bash
# PMC of FLAVA's attention-probability kernels at the image-encoder shape (B = 256, S = 197, H = 12): the r05 one-pass kernel (after the flash forward) and the r04
# two-pass kernel.  One counter group per run (FETCH_SIZE and WRITE_SIZE in separate passes: the TCC has 4 slots), --kernel-trace only.
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for v in 515 514; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
    tag=$(echo $pass | cut -d' ' -f1)
    cd /tmp && rm -rf /tmp/pmcp_${v}_$tag && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcp_${v}_$tag -o p -- python $GRAFT_REPO_ROOT/tools/one_probs.py $v > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, json, collections, pathlib
out = {}
for v, needle, label in ((515, "attention_probs_lse_kernel", "one-pass kernel (r05), alone"), (515, "attention_ring_kernel", "flash forward that feeds it"),
                         (514, "attention_probs_kernel", "two-pass kernel (r04)")):
    acc = collections.defaultdict(list)
    for f in pathlib.Path("/tmp").glob(f"pmcp_{v}_*/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(x) / len(x) for k, x in acc.items()}
    B, S, H = 256, 197, 12
    alg_w = B * H * S * S * 4 if "probs" in needle else 0
    alg = alg_w + B * S * H * 64 * 2 * (2 if "lse" in needle else 4)
    d = {"kernel": needle, "what": label, "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
         "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md HBM section) -> doubled; WRITE_SIZE as reported",
         "hbm_bytes_per_launch": int(2 * m.get("FETCH_SIZE", 0) * 1024 + m.get("WRITE_SIZE", 0) * 1024), "algorithmic_bytes_per_launch": alg,
         "algorithmic_probability_bytes": alg_w,
         "tcc_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
         "wait_any_frac": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
         "lds_bank_conflict_frac": m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else None, "raw": m}
    out[label] = d
    print(json.dumps({k: v for k, v in d.items() if k not in ("raw", "correction")}))
json.dump(out, open("gpurun_out/r05_pmc_probs_kernels.json", "w"), indent=1)
PY
