#!/bin/bash
# r06: what FullyShardedDataParallel (one RCCL rank, NO_SHARD, the reference trainer's wrap policy) costs the training step: alternating arms
mkdir -p gpurun_out
for i in 1 2; do
  for args in "" "--fsdp" "--fsdp --fsdp-orig-params"; do
    python tools/flava_bench.py --train --steps 6 $args 2>gpurun_out/fsdp_err.txt | grep '^{' || tail -5 gpurun_out/fsdp_err.txt
  done
  for args in "" "--fsdp" "--fsdp --fsdp-orig-params"; do
    python tools/train_bench.py --steps 6 $args 2>gpurun_out/fsdp_err.txt | grep '^{' || tail -5 gpurun_out/fsdp_err.txt
  done
done
