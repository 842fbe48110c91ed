"""Zero-shot / retrieval read-outs on an MI355X: multimodal_amd.utils.zero_shot against the reference's torch
expressions (fixture) and the numpy oracle.  Tolerances: the kernels work in fp32 like the reference; embeddings / logits differ only
by summation order (|d| <= 1e-6 on unit vectors, 2e-4 on logits scaled by 100); hit counts and recalls are exact."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc
from tests.conftest import GOLDEN, set_rng_seed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def test_classifier_logits_and_accuracy_vs_fixture(golden):
    from multimodal_amd.utils import zero_shot as zs

    z = golden("zero_shot.npz")
    prompts = torch.from_numpy(z["prompts"]).cuda()
    C, T, E = prompts.shape
    w = zs.class_embedding(prompts.reshape(C * T, E), groups=C).t()           # [E, C] view, like zero_shot_classifier builds
    np.testing.assert_allclose(w.cpu().numpy(), z["classifier"], atol=1e-6)
    logits = zs.zero_shot_logits(torch.from_numpy(z["feats"]).cuda(), w)
    np.testing.assert_allclose(logits.cpu().numpy(), z["logits"], atol=2e-4)
    logits_c = zs.zero_shot_logits(torch.from_numpy(z["feats"]).cuda(), torch.from_numpy(z["classifier"]).cuda())  # contiguous [E, C]
    np.testing.assert_allclose(logits_c.cpu().numpy(), z["logits"], atol=2e-4)
    target = torch.from_numpy(z["target"]).cuda()
    assert zs.accuracy(torch.from_numpy(z["logits"]).cuda(), target, (1, 5, 10)) == z["acc"].tolist()
    assert zs.accuracy(logits, target, (1, 5, 10)) == z["acc"].tolist()


def test_retrieval_similarity_and_recall_vs_fixture(golden):
    from multimodal_amd.utils import zero_shot as zs

    z = golden("zero_shot.npz")
    sim = zs.retrieval_similarity(torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["txt"]).cuda())
    np.testing.assert_allclose(sim.cpu().numpy(), z["sim"], atol=1e-6)
    ref = torch.from_numpy(z["sim"]).cuda()
    got = [float(zs.compute_recall(s, k)) for s in (ref, ref.t().contiguous()) for k in (1, 5)]
    np.testing.assert_allclose(got, z["recall"], atol=1e-7)
    r = zs.compute_recall(ref, k=1)
    assert isinstance(r, torch.Tensor) and r.dim() == 0


def test_target_rank_ties_edges_and_scale():
    """Ties resolve to the lower index; out-of-range targets never hit; a 5000 x 5000 retrieval matrix against the oracle."""
    from multimodal_amd import ops
    from multimodal_amd.utils import zero_shot as zs

    s = torch.tensor([[1.0, 3.0, 3.0, 2.0], [5.0, 5.0, 5.0, 5.0], [0.0, -1.0, 7.0, 7.0]]).cuda()
    assert ops.target_rank(s, torch.tensor([2, 3, 0]).cuda()).tolist() == [1, 3, 2]
    assert ops.target_rank(s, torch.tensor([1, 0, 9]).cuda()).tolist() == [0, 0, 4]
    assert ops.target_rank(s[:, :3].contiguous(), None).tolist() == [2, 1, 0]
    set_rng_seed(0)
    a = torch.randn(5000, 64)
    b = a + 2.5 * torch.randn(5000, 64)
    sim = zs.retrieval_similarity(a.cuda(), b.cuda())
    want = oc.zero_shot_logits(a.numpy(), (b / b.norm(dim=-1, keepdim=True)).numpy().T, 1.0, np.float64)
    np.testing.assert_allclose(sim.cpu().numpy(), want, atol=2e-6)
    for k in (1, 5, 10):
        assert abs(float(zs.compute_recall(sim, k)) - oc.recall_at_k(sim.cpu().numpy(), k)) < 1e-7  # the recall is an fp32 tensor, like the reference's


def test_zero_shot_pipeline_on_clip_towers():
    """zero_shot_classifier + run_zero_shot over the CLIP towers with the BPE text transform: the read-outs equal the oracle's on
    the embeddings the towers produced (the towers themselves are covered by test_gpu_models.py)."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.transforms.clip_transform import CLIPTextTransform
    from multimodal_amd.utils import zero_shot as zs

    set_rng_seed(3)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=49408, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt).cuda().eval()
    tt = CLIPTextTransform(text_bpe_merges_path=str(GOLDEN / "clip_bpe_merges.txt.gz"))
    names = ["cat", "dog", "aeroplane", "bicycle", "teapot", "violin"]
    templates = [lambda c: f"a photo of a {c}.", lambda c: f"a blurry photo of the {c}.", lambda c: f"art of the {c}."]
    with torch.no_grad():
        w = zs.zero_shot_classifier(clip.encoder_b, tt, names, templates, "cuda")
        assert w.shape == (64, len(names))
        for c, name in enumerate(names):
            e = clip.encoder_b(tt([t(name) for t in templates]).cuda()).float().cpu().numpy()
            np.testing.assert_allclose(w[:, c].cpu().numpy(), oc.zero_shot_class_embedding(e), atol=2e-6)
        images = torch.randn(24, 3, 64, 64).cuda()
        labels = torch.randint(0, len(names), (24,))
        feats = clip.encoder_a(images)
        res = zs.run_zero_shot(clip.encoder_a, [{"image": images[:16], "label": labels[:16]}, {"image": images[16:], "label": labels[16:]}], w)
    logits = oc.zero_shot_logits(feats.float().cpu().numpy(), w.cpu().numpy())
    top2 = np.sort(logits, axis=1)[:, -2:]
    if (top2[:, 1] - top2[:, 0]).min() > 1e-3:  # no near-ties: the counts must agree exactly
        h1, h5 = oc.topk_hits(logits, labels.numpy(), (1, 5))
        assert abs(res["top1"] - h1 / 24) < 1e-9 and abs(res["top5"] - h5 / 24) < 1e-9
    assert 0.0 <= res["top1"] <= res["top5"] <= 1.0


def test_reference_example_loop_and_classifier_vs_fixture(golden):
    """The fixture holds what the REFERENCE'S OWN `_zero_shot_classifier` and `run_imagenet_zero_shot` (examples/flava/native/utils.py:
    100-160) returned for a table-lookup stand-in of the towers (tests/golden/make_golden_zero_shot.py): same classifier, same top-1 /
    top-5 rates — including the example's quirk of stopping after six batches."""
    from multimodal_amd.utils import zero_shot as zs
    from tests.golden.make_golden_zero_shot import CLASSNAMES, TEMPLATES, text_transform

    z = golden("zero_shot.npz")
    table = torch.from_numpy(z["table"]).cuda()
    w = zs.zero_shot_classifier(lambda ids: table[ids].sum(1), text_transform, CLASSNAMES, TEMPLATES)  # stand-in tower: a gather
    np.testing.assert_allclose(w.cpu().numpy(), z["classifier"], atol=1e-6)
    feats, labels = torch.from_numpy(z["batch_feats"]).cuda(), torch.from_numpy(z["batch_labels"]).cuda()
    batches = [{"image": feats[i], "label": labels[i]} for i in range(feats.shape[0])]
    res = zs.run_zero_shot(lambda x: x, batches, w, topk=(1, 5), max_batches=6)
    assert abs(res["top1"] - float(z["run_top1"])) < 1e-12 and abs(res["top5"] - float(z["run_top5"])) < 1e-12
    every = zs.run_zero_shot(lambda x: x, batches, w, topk=(1, 5))
    assert every["top1"] != res["top1"] or every["top5"] != res["top5"] or feats.shape[0] <= 6  # all 8 batches: a different denominator
