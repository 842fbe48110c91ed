"""Where do the workgroups of a CU-masked stream run?  python tools/cu_census.py  (MI355X)
For each mask layout hypothesis and partition size: launch a spinning grid on the masked stream and count, per XCC, the distinct
(SE, SH, CU) ids that executed a workgroup.  The right layout is the one whose B partition shows `t` CUs on EVERY XCC."""
import sys
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import build, ops  # noqa: E402

build.build()


def census(stream):
    with torch.cuda.stream(stream):
        out = ops.cu_census(4096, 400000)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    per = defaultdict(set)
    for xcc, hw in o:
        xcc &= 0xF
        per[int(xcc)].add((int(hw) >> 8) & 0xFF)  # cu_id[11:8] | sh_id[12] | se_id[15:13]
    return {k: len(v) for k, v in sorted(per.items())}


print("default stream:", census(torch.cuda.current_stream()), "stream_cus", ops.stream_cus())
for layout in ("interleaved", "contiguous"):
    for t in (4, 8):
        ma, mb = ops.cu_partition_masks(t, layout)
        sa, sb = ops.create_cu_mask_stream(ma), ops.create_cu_mask_stream(mb)
        print(layout, "t =", t, "A:", census(sa), "cus", ops.stream_cus(sa), "| B:", census(sb), "cus", ops.stream_cus(sb), flush=True)
