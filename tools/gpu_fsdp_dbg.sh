cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for orig in 0 1; do
RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 LOCAL_RANK=0 FSDP_PROBE_USE_ORIG_PARAMS=$orig timeout 300 python tests/_fsdp_probe.py 2>&1 | grep FSDP_PROBE_RESULT | tee -a gpurun_out/fsdp_dbg.txt
done
