#!/bin/bash
# r06: HBM/L2-miss bytes of the grouped MLP-up and qkv launches with and without the column-chunked tile order (MMAMD_GEMM_CN=-1: off)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
rm -rf /tmp/pmc_cn
for arm in auto off; do
  for shape in "mlpup 3072 768 2048 512 1" "qkv 2304 768 1536 512 0"; do
    set -- $shape
    for pass in "FETCH_SIZE" "WRITE_SIZE"; do
      cd /tmp && MMAMD_GEMM_CN=$([ $arm = off ] && echo -1 || echo 0) timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_cn/$arm/$1/$pass -o p -- python $GRAFT_REPO_ROOT/tools/one_gemm.py 50432 $2 $3 $6 0 0 8 19712 $4 $5 > /dev/null 2>&1
    done
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, json, collections, pathlib
out = {}
for arm in ("auto", "off"):
    for name in ("mlpup", "qkv"):
        acc = collections.defaultdict(list); dur = []
        for f in pathlib.Path(f"/tmp/pmc_cn/{arm}/{name}").rglob("*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "gemm_bf16_nt_kernel_ppg" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in pathlib.Path(f"/tmp/pmc_cn/{arm}/{name}").rglob("*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if "gemm_bf16_nt_kernel_ppg" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        m = {k: sum(v) / len(v) for k, v in acc.items()}
        dur.sort()
        out[f"{name}.{arm}"] = {"FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
                                "bytes_per_launch(FETCH x 2 + WRITE)": int(2 * m.get("FETCH_SIZE", 0) * 1024 + m.get("WRITE_SIZE", 0) * 1024), "median_us_under_profiler": dur[len(dur) // 2] if dur else None}
        print(name, arm, out[f"{name}.{arm}"])
json.dump({"what": "grouped MLP-up / qkv launches (both towers, B = 256): L2-miss traffic with the column-chunked tile order (auto) and without (off); rocprofv3 --pmc, separate passes, FETCH x 2 per the gfx950 correction", "launches": out}, open("gpurun_out/r06_pmc_column_chunks.json", "w"), indent=1)
PY
