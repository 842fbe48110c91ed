#!/bin/bash
# r06 energy budget: each resource of the GEMM step alone under the telemetry sampler (tools/power_clocks.py --cmd), then the step's own launches
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -o /tmp/energy_parts tools/microbench/energy_parts.hip || exit 1
S=${1:-5}
python tools/power_clocks.py --hz 20 --out gpurun_out/r06_energy_parts.json \
  --cmd "idle=sleep $S" \
  --cmd "lds_read=/tmp/energy_parts lds_read $S" --cmd "l2_dma=/tmp/energy_parts l2_dma $S" \
  --cmd "hbm_read=/tmp/energy_parts hbm_read $S" --cmd "hbm_copy=/tmp/energy_parts hbm_copy $S" \
  --cmd "valu=/tmp/energy_parts valu $S" --cmd "mfma=/tmp/energy_parts mfma $S" \
  --cmd "mfma_lds=/tmp/energy_parts mfma_lds $S" --cmd "mfma_lds_dma=/tmp/energy_parts mfma_lds_dma $S" > gpurun_out/r06_energy_parts.txt 2>&1
cat gpurun_out/r06_energy_parts.txt


python tools/power_clocks.py --seconds 5 --hz 20 --out gpurun_out/r06_power_clocks.json > gpurun_out/r06_power_clocks.txt 2>&1
cat gpurun_out/r06_power_clocks.txt
