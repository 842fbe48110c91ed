"""DALL-E dVAE encoder (FLAVA image codebook) fixtures from the REFERENCE:  python -m tests.golden.make_golden_flava_codebook
  flava_codebook.npz
    small.*  DalleEncoder(n_hid=256, n_blk_per_group=1, vocab_size=512) under seed 3 (weights by seed: keys / sums / abs-sums), 3 images of
             32x32: z_logits [3,512,4,4] and their argmax
    full.*   DalleVAEEncoder() architecture (8192 codes, 2 blocks per group) under seed 7 with random weights, 2 images of 112x112:
             codebook indices [2,14,14], the top-1 / top-2 logit margin per position, z_logits at every 64th code
"""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import checksums, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def main():
    _ref_shim.install()
    from torchmultimodal.models.flava.model import DalleEncoder, DalleVAEEncoder

    torch.set_num_threads(8)
    st = {}
    seed(3)
    small = DalleEncoder(n_hid=256, n_blk_per_group=1, vocab_size=512).eval()
    st["small.keys"], st["small.sums"], st["small.asums"] = checksums(small)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 3, 32, 32, generator=g)
    with torch.no_grad():
        z = small(x)
    st["small.x"], st["small.logits"], st["small.indices"] = x.numpy(), z.numpy(), torch.argmax(z, dim=1).numpy()

    seed(7)
    # DalleVAEEncoder(pretrained=True) would download OpenAI's checkpoint; the architecture with seeded random weights is what is pinned
    full = DalleVAEEncoder.__new__(DalleVAEEncoder)
    torch.nn.Module.__init__(full)
    full.image_size = 112
    from torchmultimodal.models.flava.model import DalleEncoder as Enc
    full.encoder = Enc()
    full.eval()
    st["full.keys"], st["full.sums"], st["full.asums"] = checksums(full)
    xf = torch.randn(2, 3, 112, 112, generator=g)
    with torch.no_grad():
        zf = full.encoder(xf)
        idx = full.get_codebook_indices(xf)
    top2 = torch.topk(zf, 2, dim=1).values
    st["full.x"] = xf.numpy().astype(np.float16)  # the test feeds exactly these (fp16-representable) pixels
    with torch.no_grad():
        xq = torch.from_numpy(st["full.x"].astype(np.float32))
        zf = full.encoder(xq)
        idx = full.get_codebook_indices(xq)
    top2 = torch.topk(zf, 2, dim=1).values
    st["full.indices"] = idx.numpy()
    st["full.margin"] = (top2[:, 0] - top2[:, 1]).numpy()
    st["full.logits_s64"] = zf[:, ::64].numpy()
    st["full.logit_absmax"] = np.float32(zf.abs().max())
    np.savez_compressed(OUT / "flava_codebook.npz", **st)
    print({k: getattr(v, "shape", None) for k, v in st.items()})
    print("margin quantiles", np.quantile(st["full.margin"], [0.0, 0.01, 0.05, 0.5]), "absmax", st["full.logit_absmax"])
    print("bytes", (OUT / "flava_codebook.npz").stat().st_size)


if __name__ == "__main__":
    main()
