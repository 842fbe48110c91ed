"""Fused ViT stem (mmamd_patch_embed_gemm + mmamd_vit_cls_lnpre_ln: the GEMM gathers the patch rows from the bf16 image by its LDS-DMA source
addresses, adds the positional embedding in its epilogue; one row kernel writes the CLS rows, applies ln_pre and norm1 of the first layer)
against the three-pass path it replaces (patchify -> GEMM -> vit_assemble_ln, then layernorm): bit for bit.  Reference:
models/clip/image_encoder.py:91-106.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


@pytest.mark.parametrize("patch,width,heads,B", [(16, 768, 12, 5), (16, 768, 12, 37), (32, 768, 12, 9), (16, 256, 4, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@torch.no_grad()
def test_fused_stem_equals_three_pass_stem(patch, width, heads, B, dtype):
    from multimodal_amd import ops
    from multimodal_amd.models.clip.image_encoder import CLIPViTEncoder

    torch.manual_seed(patch + B)
    enc = CLIPViTEncoder(embedding_dim=512, patch_size=patch, image_size=224, width=width, heads=heads, layers=1).cuda().eval()
    for p in (enc.ln_pre.weight, enc.ln_pre.bias, enc.encoder.layers[0].norm1.weight, enc.encoder.layers[0].norm1.bias):
        p.data.add_(0.2 * torch.randn_like(p))
    img = torch.randn(B, 3, 224, 224).cuda().to(dtype)
    h, b, S, hn0 = enc._stem(img, want_hn0=True)
    assert hn0 is not None and (b, S) == (B, (224 // patch) ** 2 + 1)
    K = 3 * patch * patch
    h_ref, b2, S2 = enc._stem_patches(ops.patchify(img.contiguous(), patch, K))
    n1 = enc.encoder.layers[0].norm1
    hn_ref = ops.layernorm(h_ref, n1.weight.detach().float(), n1.bias.detach().float(), n1.eps)
    assert (b2, S2) == (b, S)
    assert torch.equal(h, h_ref)
    assert torch.equal(hn0, hn_ref)
    # and the whole encoder forward agrees with the same stack run on the three-pass stem
    out = enc(img)
    ref = enc._head(enc.encoder.run(h_ref.clone(), B, S, causal=False), B, S)
    assert torch.equal(out, ref)


@torch.no_grad()
def test_l14_keeps_the_patchify_path():
    from multimodal_amd.models.clip.image_encoder import CLIPViTEncoder

    torch.manual_seed(0)
    enc = CLIPViTEncoder(embedding_dim=64, patch_size=14, image_size=224, width=128, heads=2, layers=1).cuda().eval()
    h, B, S, hn0 = enc._stem(torch.randn(2, 3, 224, 224).cuda(), want_hn0=True)
    assert hn0 is None and S == 257 and torch.isfinite(h).all()
