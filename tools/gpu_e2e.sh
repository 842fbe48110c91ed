#!/bin/bash
set +e
mkdir -p gpurun_out
timeout 300 python tools/e2e_bench.py > gpurun_out/e2e_bench.log 2>&1
tail -3 gpurun_out/e2e_bench.log | cut -c1-1200
