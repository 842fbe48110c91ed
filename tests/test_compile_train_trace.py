"""torch.compile(fullgraph=True) traces the CLIP TRAINING step (both towers, contrastive loss, backward): every autograd node of the path
is a torch.autograd.Function whose forward and backward bodies are single dispatcher ops (torch.ops.mmamd_train.*,
multimodal_amd/_custom_op.py) with fake implementations, so dynamo + AOT autograd meet no ctypes call and no graph break.  Runs on meta
tensors: shapes only, no GPU (the numerical twin is tests/test_gpu_compile_train.py)."""
import pytest
import torch


def _small_clip():
    from multimodal_amd.models.clip import CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.models.clip.model import CLIP
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature

    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=32, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=8, vocab_size=64, width=128, heads=2, layers=2)
    return CLIP(vit, txt).train(), ContrastiveLossWithTemperature()


@pytest.mark.parametrize("backend", ["eager", "aot_eager"])
def test_training_step_traces_without_graph_breaks(backend):
    from multimodal_amd import _torch_ops

    if not _torch_ops.try_load():
        pytest.skip("libmmamd_torch.so is not built")
    torch._dynamo.reset()
    with torch.device("meta"):
        model, loss_fn = _small_clip()
        images = torch.zeros(4, 3, 32, 32)
        ids = torch.zeros(4, 8, dtype=torch.long)

    def step(images, ids):
        out = model(images, ids)
        return loss_fn(out.embeddings_a, out.embeddings_b)

    loss = torch.compile(step, backend=backend, fullgraph=True)(images, ids)
    assert loss.shape == () and loss.requires_grad
    loss.backward()
    missing = [n for n, p in list(model.named_parameters()) + list(loss_fn.named_parameters()) if p.grad is None or p.grad.shape != p.shape]
    assert not missing, missing


def test_training_ops_are_registered_with_fake_implementations():
    import multimodal_amd.models.clip._train  # noqa: F401  (registers the ops)
    import multimodal_amd.modules.losses.contrastive_loss_with_temperature  # noqa: F401

    names = ["encoder_stack_fwd", "encoder_stack_bwd", "l2_normalize_fwd", "l2_normalize_bwd", "clip_vision_embed_fwd", "clip_vision_embed_bwd",
             "clip_text_embed_fwd", "clip_text_embed_bwd", "clip_pooled_head_fwd", "clip_pooled_head_bwd", "contrastive_fwd", "contrastive_bwd"]
    for n in names:
        assert hasattr(torch.ops.mmamd_train, n), n
    x = torch.zeros(6, 16, device="meta")
    assert torch.ops.mmamd_train.l2_normalize_fwd(x).shape == (6, 16)
    # host tensors are refused by the one implementation there is (no CPU fallback)
    from multimodal_amd import ops

    with pytest.raises(ops.MmamdError):
        torch.ops.mmamd_train.l2_normalize_fwd(torch.zeros(6, 16))


def test_compiled_inference_is_one_pair_op_with_the_eager_output_layout():
    """torch.compile of the CLIP pair in inference: ONE dispatcher op (clip_pair_fwd) whose implementation is the eager forward, so the compiled
    model runs the grouped two-tower schedule; the outputs are the two halves of one packed [B, 2E] block, as in eager mode; a deep copy of the
    model registers under its own key."""
    import copy

    from multimodal_amd import _torch_ops

    if not _torch_ops.try_load():
        pytest.skip("libmmamd_torch.so is not built")
    torch._dynamo.reset()
    with torch.device("meta"):
        model, _ = _small_clip()
        model = model.eval()
        images = torch.zeros(4, 3, 32, 32)
        ids = torch.zeros(4, 8, dtype=torch.long)
    seen = []

    def backend(gm, example_inputs):
        seen.extend(str(n.target) for n in gm.graph.nodes if n.op == "call_function")
        return gm.forward

    with torch.no_grad():
        out = torch.compile(model, backend=backend, fullgraph=True)(images, ids)
    assert any("clip_pair_fwd" in t for t in seen), seen
    assert not any("mmamd.gemm_bf16" in t or "mmamd.attn_fwd" in t for t in seen), seen
    assert out.embeddings_a.shape == (4, 64) and out.embeddings_a.stride() == (128, 1) and out.embeddings_b.storage_offset() == 64
    assert copy.deepcopy(model)._pair_key != model._pair_key
