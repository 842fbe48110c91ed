#!/bin/bash
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
echo "--- a) smoke alone"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "--- b) build then smoke"; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
echo "--- c) build, torch init, smoke"; timeout 300 python -c "
import __graft_entry__ as g; g.build()
import torch; torch.cuda.init(); print(torch.cuda.device_count()); g.smoke()" 2>&1 | tail -2
echo "--- d) lib load before torch import, then a kernel"; timeout 300 python -c "
from multimodal_amd import _lib
h=_lib.lib(); print('clear ->', h.mmamd_clear_last_hip_error())
import torch
x=torch.randn(4,8,device='cuda'); print('after torch cuda: clear ->', h.mmamd_clear_last_hip_error())
from multimodal_amd import ops
print(ops.l2_normalize(x)[0,:3])" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_torch_ops.py tests/test_gpu_coca.py tests/test_gpu_flava.py -q -m gpu > $O/r2_pytest_diag.log 2>&1; grep -E "passed|failed|^FAILED" $O/r2_pytest_diag.log
