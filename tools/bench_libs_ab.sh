#!/bin/bash
# usage: bench_libs_ab.sh rounds name=lib ...   (empty lib = in-tree build): alternating bench.py runs, ms_per_step per arm
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ -n "$lib" ]; then export MMAMD_LIB=$lib MMAMD_LIB_ALLOW_MISSING=1; else unset MMAMD_LIB MMAMD_LIB_ALLOW_MISSING; fi
    ms=$(python bench.py --steps 30 --warmup 8 --cpu-sample 0 --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['loss'])")
    echo "round $r $name $ms"
  done
done
unset MMAMD_LIB
