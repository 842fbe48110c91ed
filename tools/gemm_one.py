"""Run one GEMM shape with one kernel variant a few times (target of rocprofv3 --pmc passes):
    python tools/gemm_one.py VARIANT M N K [act] [res] [iters]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import build, ops  # noqa: E402

build.build()
v, M, N, K = (int(x) for x in sys.argv[1:5])
act = int(sys.argv[5]) if len(sys.argv) > 5 else 0
res = int(sys.argv[6]) if len(sys.argv) > 6 else 0
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
bias = torch.randn(N, generator=g).cuda()
x0 = torch.randn(M, N, generator=g).cuda() if res else None
out = torch.empty(M, N, dtype=torch.float32 if res else torch.bfloat16, device="cuda")
ops.set_gemm_variant(v)
for _ in range(iters):
    ops.gemm_bf16(a, w, bias, act=act, residual=x0, out=out)
torch.cuda.synchronize()
