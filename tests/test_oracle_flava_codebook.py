"""Pin the DALL-E encoder part of the numpy oracle (oracle/clip_oracle.py: conv2d_same, dalle_encoder_forward) to the reference
(fixture tests/golden/flava_codebook.npz from make_golden_flava_codebook.py).  CPU-only."""
import numpy as np
import torch

from oracle import clip_oracle as oc
from tests._util import assert_checksums
from tests.conftest import set_rng_seed


def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def test_dalle_encoder_small_logits_and_indices(golden):
    from multimodal_amd.models.flava.model import DalleEncoder

    z = golden("flava_codebook.npz")
    set_rng_seed(3)
    enc = DalleEncoder(n_hid=256, n_blk_per_group=1, vocab_size=512)  # same seeded initialisation as the reference's (checksums)
    assert_checksums(enc, _sub(z, "small."))
    sd = {k: v.numpy() for k, v in enc.state_dict().items()}
    logits = oc.dalle_encoder_forward(sd, z["small.x"])
    np.testing.assert_allclose(logits, z["small.logits"], atol=2e-5)
    assert np.array_equal(np.argmax(logits, axis=1), z["small.indices"])


def test_dalle_vae_encoder_full_size_one_image(golden):
    """Full architecture (8192 codes, 8 blocks, 112x112): one image through the oracle == the reference's indices and sampled logits."""
    from multimodal_amd.models.flava.model import DalleVAEEncoder

    z = golden("flava_codebook.npz")
    set_rng_seed(7)
    vae = DalleVAEEncoder(pretrained=False)
    assert_checksums(vae, _sub(z, "full."))
    sd = {k: v.numpy() for k, v in vae.state_dict().items()}
    x = z["full.x"][:1].astype(np.float32)
    logits = oc.dalle_encoder_forward(sd, x, "encoder.")
    np.testing.assert_allclose(logits[:, ::64], z["full.logits_s64"][:1], atol=5e-5)
    idx = np.argmax(logits, axis=1)
    safe = z["full.margin"][:1] > 1e-4
    assert np.array_equal(idx[safe], z["full.indices"][:1][safe]) and safe.mean() > 0.99
    with torch.no_grad():
        assert vae.encoder.blocks.group_1.block_1.post_gain == 1 / 64
