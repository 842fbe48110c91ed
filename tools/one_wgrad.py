#!/usr/bin/env python
"""A few launches of the TN weight-gradient GEMM (MLP-up shape) and of one codebook convolution, for rocprofv3 --pmc passes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
torch.manual_seed(0)
if what == "wgrad":
    T, M, N = 50432, 3072, 768
    y = (torch.randn(T, M) * 0.1).to(torch.bfloat16).cuda()
    x = torch.randn(T, N).to(torch.bfloat16).cuda()
    for _ in range(5):
        ops.gemm_bf16_tn_splitk(y, x)
else:  # first 3x3 convolution of the DALL-E encoder's group 1 at B = 128: 256 -> 64 channels on the 114 x 114 padded grid
    from multimodal_amd.models.flava._dalle import _Grid, DalleConv2d

    g = _Grid(128, 112, 112, torch.device("cuda"))
    conv = DalleConv2d(256, 64, 3).cuda()
    w, b = conv.packed()
    xin, out = g.new(256), g.new(64)
    xin.normal_()
    for _ in range(5):
        ops.conv_gemm_bf16(g.rows(xin), g.taps3, w, b, g.rows(out), g.M, 64, 256, g.gh, g.gw, relu_c=True)
torch.cuda.synchronize()
