"""FLAVA's image codebook (DALL-E dVAE encoder, reference models/flava/model.py:583-744) on an MI355X: the implicit-GEMM pipeline of
csrc/conv.hip through the C-ABI vs the reference fixture and the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc
from tests._util import assert_checksums
from tests.conftest import set_rng_seed

pytestmark = pytest.mark.gpu

# bf16 operands / bf16 residual stream through 4-8 residual blocks against an fp32 reference; logits are O(1)
LOGIT_TOL = 6e-2


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def test_conv_gemm_kernel_vs_oracle_conv():
    """One 3x3 and one 1x1 convolution on a padded grid: every output width class (N = 64 / 128 / 256+), residual, both ReLU outputs,
    fp32 output, borders stored as zeros."""
    from multimodal_amd import ops
    from multimodal_amd.models.flava._dalle import _Grid, DalleConv2d

    set_rng_seed(5)
    B, H, W = 2, 6, 10
    g = _Grid(B, H, W, torch.device("cuda"))
    for cin, cout, kw in ((64, 64, 3), (128, 128, 3), (64, 320, 3), (192, 64, 1), (256, 264, 1)):
        conv = DalleConv2d(cin, cout, kw).cuda()
        with torch.no_grad():
            conv.b.normal_()
        x = torch.randn(B, cin, H, W)
        xb = x.to(torch.bfloat16)
        res = torch.randn(B, cout, H, W).to(torch.bfloat16)
        ref = oc.conv2d_same(xb.float().numpy().astype(np.float64), host(conv.w.to(torch.bfloat16)), host(conv.b))  # same bf16 operands
        xin = g.new(cin)
        grid = torch.zeros(B, g.gh, g.gw, cin, dtype=torch.bfloat16)
        grid[:, 1:-1, 1:-1] = xb.permute(0, 2, 3, 1)
        g.rows(xin).copy_(grid.view(-1, cin).cuda())
        rin = torch.zeros(B, g.gh, g.gw, cout, dtype=torch.bfloat16)
        rin[:, 1:-1, 1:-1] = res.permute(0, 2, 3, 1)
        rin = rin.view(-1, cout).cuda()
        w, b = conv.packed()
        taps = g.taps3 if kw == 3 else g.tap1
        out, out_r = torch.full((g.M, cout), 7.0, dtype=torch.bfloat16).cuda(), torch.full((g.M, cout), 7.0, dtype=torch.bfloat16).cuda()
        ops.conv_gemm_bf16(g.rows(xin), taps, w, b, out, g.M, cout, cin, g.gh, g.gw, residual=rin, out_relu=out_r)
        o = host(out).reshape(B, g.gh, g.gw, cout)
        want = ref + res.float().numpy().astype(np.float64)
        tol = 2.0 ** -7 * max(1.0, np.abs(want).max())
        assert np.abs(o[:, 1:-1, 1:-1].transpose(0, 3, 1, 2) - want).max() <= tol, (cin, cout, kw)
        assert not o[:, 0].any() and not o[:, -1].any() and not o[:, :, 0].any() and not o[:, :, -1].any()
        assert np.array_equal(host(out_r), np.maximum(host(out), 0))
        of = torch.empty((g.M, cout), dtype=torch.float32).cuda()
        ops.conv_gemm_bf16(g.rows(xin), taps, w, b, of, g.M, cout, cin, 0, 0)
        assert np.abs(host(of).reshape(B, g.gh, g.gw, cout)[:, 1:-1, 1:-1].transpose(0, 3, 1, 2) - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())
        orl = torch.empty((g.M, cout), dtype=torch.bfloat16).cuda()
        ops.conv_gemm_bf16(g.rows(xin), taps, w, b, orl, g.M, cout, cin, g.gh, g.gw, relu_c=True)
        assert np.abs(host(orl).reshape(B, g.gh, g.gw, cout)[:, 1:-1, 1:-1].transpose(0, 3, 1, 2) - np.maximum(ref, 0)).max() <= tol


def test_pool_argmax_and_stem_kernels():
    from multimodal_amd import ops
    from multimodal_amd.models.flava._dalle import _Grid

    set_rng_seed(6)
    B, H, W, C = 2, 4, 8, 64
    g, g2 = _Grid(B, H, W, torch.device("cuda")), _Grid(B, H // 2, W // 2, torch.device("cuda"))
    x = torch.randn(B, g.gh, g.gw, C).to(torch.bfloat16)
    y, yr = g2.new(C), g2.new(C)
    ops.dalle_maxpool2(x.view(-1, C).cuda(), g2.rows(y), g2.rows(yr), B, H, W, C)
    want = x[:, 1:-1, 1:-1].float().reshape(B, H // 2, 2, W // 2, 2, C).amax(dim=(2, 4))
    got = g2.rows(y).float().cpu().view(B, g2.gh, g2.gw, C)
    assert torch.equal(got[:, 1:-1, 1:-1], want) and not got[:, 0].any() and not got[:, :, -1].any()
    assert torch.equal(g2.rows(yr).float().cpu(), torch.relu(g2.rows(y).float().cpu()))
    logits = torch.randn(B, g.gh, g.gw, 1000)
    logits[0, 1, 1, 5] = logits[0, 1, 1, 900] = 50.0  # a tie: the first maximum wins, like torch.argmax
    ids = ops.dalle_argmax(logits.view(-1, 1000).cuda(), B, H, W, 1000)
    assert torch.equal(ids.cpu(), logits[:, 1:-1, 1:-1].argmax(dim=-1)) and int(ids[0, 0, 0]) == 5
    img = torch.randn(B, 3, 8, 8)
    cols = torch.empty((B * 10 * 10, 192), dtype=torch.bfloat16).cuda()
    ops.dalle_stem_im2col(img.cuda(), 7, 192, cols)
    pad = torch.nn.functional.pad(img, (3, 3, 3, 3))
    c = cols.float().cpu().view(B, 10, 10, 192)
    for (b, y0, x0) in ((0, 0, 0), (1, 7, 3), (0, 4, 4)):
        patch = pad[b, :, y0:y0 + 7, x0:x0 + 7].to(torch.bfloat16).float().reshape(-1)
        assert torch.equal(c[b, y0 + 1, x0 + 1, :147], patch) and not c[b, y0 + 1, x0 + 1, 147:].any()


def test_small_dalle_encoder_vs_reference_fixture(golden):
    from multimodal_amd.models.flava.model import DalleEncoder

    z = golden("flava_codebook.npz")
    set_rng_seed(3)
    enc = DalleEncoder(n_hid=256, n_blk_per_group=1, vocab_size=512)
    assert_checksums(enc, _sub(z, "small."))
    enc = enc.cuda().eval()
    with torch.no_grad():
        logits = enc(torch.from_numpy(z["small.x"]).cuda())
        idx = enc.codebook_indices(torch.from_numpy(z["small.x"]).cuda())
    assert logits.shape == (3, 512, 4, 4) and idx.shape == (3, 4, 4) and idx.dtype == torch.int64
    d = np.abs(host(logits) - z["small.logits"]).max()
    ref_sorted = np.sort(z["small.logits"], axis=1)
    margin = ref_sorted[:, -1] - ref_sorted[:, -2]
    safe = margin > 2 * LOGIT_TOL
    agree = (host(idx) == z["small.indices"])
    print(f"small DALL-E encoder: max |d logits| = {d:.3e} (|logits| <= {np.abs(z['small.logits']).max():.2f}); argmax agreement {agree.mean():.3f}, "
          f"{safe.mean():.2f} of positions have a safe margin")
    assert d <= LOGIT_TOL and agree[safe].all()
    assert torch.equal(idx, logits.argmax(dim=1))  # the index kernel and the returned logits agree


def test_full_size_codebook_vs_reference_fixture(golden):
    """DalleVAEEncoder architecture (8192 codes, 112x112): indices equal the reference's wherever its top-2 margin exceeds twice the
    logit tolerance; the unfiltered agreement rate is reported."""
    from multimodal_amd import ops
    from multimodal_amd.models.flava.model import DalleVAEEncoder

    z = golden("flava_codebook.npz")
    set_rng_seed(7)
    vae = DalleVAEEncoder(pretrained=False)
    assert_checksums(vae, _sub(z, "full."))
    vae = vae.cuda().eval()
    x = torch.from_numpy(z["full.x"].astype(np.float32)).cuda()
    with torch.no_grad():
        idx = vae(x)
        logits = vae.encoder(x)
    assert idx.shape == (2, 14, 14) and logits.shape == (2, 8192, 14, 14)
    d = np.abs(host(logits[:, ::64]) - z["full.logits_s64"]).max()
    safe = z["full.margin"] > 2 * LOGIT_TOL
    agree = host(idx) == z["full.indices"]
    print(f"full-size codebook: max |d logits| (every 64th code) = {d:.3e} (|logits| <= {float(z['full.logit_absmax']):.2f}); "
          f"index agreement {agree.mean():.3f} unfiltered, {safe.mean():.2f} of positions have a safe margin")
    assert d <= LOGIT_TOL and agree[safe].all()
    with torch.no_grad():
        probs = vae.get_codebook_probs(x)
    ref_p = torch.softmax(logits.float().cpu().double(), dim=1).numpy()  # softmax of OUR logits in float64 (test-side check of the row kernel)
    assert probs.shape == (2, 8192, 14, 14) and np.abs(host(probs) - ref_p).max() <= 1e-6 and abs(float(probs[0, :, 3, 3].sum()) - 1.0) <= 1e-5
    with pytest.raises(RuntimeError):
        DalleVAEEncoder()  # pretrained=True needs the network, like the reference


def test_flava_for_pretraining_with_the_dalle_codebook():
    """flava_model_for_pretraining() end to end (models/flava/model.py:301-378, 524-544): MIM labels come from the DALL-E codebook on
    `image_for_codebook`, unmasked patches become -1, and the result equals feeding those labels to the loss by hand."""
    from multimodal_amd.models.flava.model import DalleVAEEncoder, flava_model_for_pretraining

    kw = dict(image_hidden_size=128, image_num_attention_heads=2, image_num_hidden_layers=1, image_intermediate_size=256, image_size=32,
              patch_size=16, text_hidden_size=128, text_num_attention_heads=2, text_num_hidden_layers=1, text_intermediate_size=256,
              vocab_size=200, max_position_embeddings=32, multimodal_hidden_size=128, multimodal_num_attention_heads=2,
              multimodal_num_hidden_layers=1, multimodal_intermediate_size=256, text_and_image_proj_size=64)
    set_rng_seed(12)
    pre = flava_model_for_pretraining(codebook_image_size=16, **kw)
    assert isinstance(pre.image_codebook, DalleVAEEncoder) and len(pre.image_codebook.state_dict()) == 74
    pre = pre.cuda().eval()
    B = 4
    image, img_cb = torch.randn(B, 3, 32, 32).cuda(), torch.randn(B, 3, 16, 16).cuda()
    text = torch.randint(1, 200, (B, 12)).cuda()
    masked = text.clone()
    masked[:, 3] = 103
    mlm = torch.full_like(text, -1)
    mlm[:, 3] = text[:, 3]
    pm = torch.tensor([[1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 1, 1], [0, 0, 1, 0]]).cuda()
    itm = torch.ones(B, dtype=torch.long).cuda()
    with torch.no_grad():
        out = pre(image=image, text=text, image_for_codebook=img_cb, image_patches_mask=pm, text_masked=masked, itm_labels=itm, mlm_labels=mlm)
        ids = pre.image_codebook(img_cb)
        assert ids.shape == (B, 2, 2) and int(ids.min()) >= 0 and int(ids.max()) < 8192
        labels = ids.flatten(1).clone()
        labels[pm == 0] = -1  # host-side restatement of model.py:340-343 (test only)
        fo = pre.model(image=image, text=text, image_patches_mask=pm.to(torch.bool), text_masked=masked)
        ref = pre.loss(image_sequence=fo.image.last_hidden_state, text_sequence=fo.text.last_hidden_state,
                       image_masked_sequence=fo.image_masked.last_hidden_state, text_masked_sequence=fo.text_masked.last_hidden_state,
                       multimodal_masked_sequence=fo.multimodal_masked.last_hidden_state, itm_labels=itm, mim_labels=labels, mlm_labels=mlm,
                       projected_image_embeddings=fo.projected_image_embeddings, projected_text_embeddings=fo.projected_text_embeddings)
    for name in ("mmm_image_loss", "mmm_text_loss", "itm_loss", "global_contrastive_loss"):
        a, b = getattr(out.losses, name), getattr(ref.losses, name)
        assert a is not None and abs(float(a) - float(b)) <= 1e-6, name
