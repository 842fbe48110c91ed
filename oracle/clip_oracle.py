"""CPU ORACLE (test infrastructure, NOT product code) for the dual-encoder contrastive hot path.

This file is a plain-numpy restatement of what the reference computes on its PyTorch CPU path for
CLIP forward + ContrastiveLossWithTemperature.  It exists only so that tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg have something to check the HIP kernels against on a box where
/root/reference does not exist.  Nothing under multimodal_amd/ may import it.

Pinned (see tests/test_oracle_golden.py and tests/golden/make_golden.py):
  * against every eval-mode known-answer vector the reference's own tests hold for this path
    (tests/models/clip/test_image_encoder.py:58-64, tests/models/clip/test_text_encoder.py:107-149,
     tests/modules/losses/test_contrastive_loss_with_temperature.py:75-82,112-123,182-184),
  * against outputs of the reference itself, imported in the build container under the 3-module shim
    (tests/golden/_ref_shim.py) and committed as fixtures under tests/golden/*.npz.

Part of the arithmetic of this path lives in a third-party dependency of the reference that is not
vendored in /root/reference: `torch` (unpinned by the reference; CI uses pytorch-nightly,
.github/workflows/unit_test.yaml:32; installed here: 2.10.0+rocm7.0).  The semantics restated from it:
  nn.TransformerEncoderLayer(norm_first=True):  x = x + sa(norm1(x)); x = x + ff(norm2(x))
      (torch/nn/modules/transformer.py:946-950), ff = linear2(act(linear1(x))) (:980-982), LN eps 1e-5
  nn.MultiheadAttention packed in-projection  q,k,v = split(x @ in_proj_weight.T + in_proj_bias)
      (torch/nn/functional.py `_in_projection_packed`), heads split along the channel dim, dh = d/h
  F.scaled_dot_product_attention  softmax(q k^T / sqrt(dh) [+ causal mask]) v
  F.layer_norm (biased variance), F.normalize (x / max(||x||_2, eps)), F.cross_entropy
      (mean reduction, label_smoothing:  (1-s)*nll + s*mean_j(-logp_j)).
Reference call sites for those: models/clip/image_encoder.py:65-77,108; models/clip/text_encoder.py:58-66,121;
models/clip/model.py:72-73; modules/losses/contrastive_loss_with_temperature.py:90-107.

All functions take / return numpy arrays.  `dtype` selects the arithmetic type (float32 mirrors the
reference CPU path; float64 is used by tests that need a tighter anchor).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

Array = np.ndarray


# ----------------------------------------------------------------------------------------------
# elementary pieces
# ----------------------------------------------------------------------------------------------
def layer_norm(x: Array, weight: Array, bias: Array, eps: float) -> Array:
    """F.layer_norm over the last dim, biased variance (Fp32LayerNorm: modules/layers/normalizations.py:13-25)."""
    mean = x.mean(axis=-1, keepdims=True)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * weight + bias


def quick_gelu(x: Array) -> Array:
    """SiLU of the reference = x * sigmoid(1.702 x) (modules/layers/activation.py:24-25)."""
    return x / (1.0 + np.exp(-x.dtype.type(1.702) * x))


def softmax_lastdim(s: Array) -> Array:
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(axis=-1, keepdims=True)


def l2_normalize(x: Array, eps: float = 1e-12) -> Array:
    """F.normalize(x) with p=2, dim=1 (models/clip/model.py:72-73)."""
    n = np.sqrt((x * x).sum(axis=1, keepdims=True))
    return x / np.maximum(n, x.dtype.type(eps))


def multi_head_self_attention(
    x: Array, in_w: Array, in_b: Array, out_w: Array, out_b: Array, heads: int, causal: bool
) -> Array:
    """nn.MultiheadAttention self-attention, batch-first [B,S,d] (see module docstring for torch sites)."""
    B, S, d = x.shape
    dh = d // heads
    qkv = x @ in_w.T + in_b  # [B,S,3d]
    q, k, v = qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :]

    def split(t):  # [B,S,d] -> [B,h,S,dh]
        return t.reshape(B, S, heads, dh).transpose(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    s = (q @ k.transpose(0, 1, 3, 2)) * x.dtype.type(1.0 / math.sqrt(dh))
    if causal:
        # text_encoder.py:74-77 mask = full(-inf).triu(1); torch uses SDPA's causal flag for it
        mask = np.triu(np.ones((S, S), dtype=bool), k=1)
        s = np.where(mask, -np.inf, s).astype(x.dtype)
    p = softmax_lastdim(s)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, S, d)
    return o @ out_w.T + out_b


def encoder_layer(x: Array, sd: Dict[str, Array], prefix: str, heads: int, causal: bool) -> Array:
    """One pre-norm torch.nn.TransformerEncoderLayer with QuickGELU (image_encoder.py:65-73)."""
    g = lambda k: sd[prefix + k]
    h = layer_norm(x, g("norm1.weight"), g("norm1.bias"), 1e-5)
    x = x + multi_head_self_attention(
        h,
        g("self_attn.in_proj_weight"),
        g("self_attn.in_proj_bias"),
        g("self_attn.out_proj.weight"),
        g("self_attn.out_proj.bias"),
        heads,
        causal,
    )
    h = layer_norm(x, g("norm2.weight"), g("norm2.bias"), 1e-5)
    h = quick_gelu(h @ g("linear1.weight").T + g("linear1.bias"))
    x = x + (h @ g("linear2.weight").T + g("linear2.bias"))
    return x


def _num_layers(sd: Dict[str, Array], prefix: str) -> int:
    n = 0
    while f"{prefix}encoder.layers.{n}.norm1.weight" in sd:
        n += 1
    return n


def _cast(sd: Dict[str, Array], dtype) -> Dict[str, Array]:
    return {k: np.asarray(v).astype(dtype, copy=False) for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# towers
# ----------------------------------------------------------------------------------------------
def patch_embed(images: Array, conv_w: Array) -> Array:
    """nn.Conv2d(3,w,k=p,s=p,bias=False) + flatten + permute (image_encoder.py:91-97) -> [B, g*g, w]."""
    B, C, H, W = images.shape
    w, _, p, _ = conv_w.shape
    gh, gw = H // p, W // p  # (rectangular inputs: layers.PatchEmbeddings accepts them, modules/layers/patch_embedding.py:58-66)
    # [B,C,gh,p,gw,p] -> [B,gh,gw,C,p,p] -> [B,gh*gw,C*p*p]; k-order (c,py,px) == conv weight flattening
    patches = images[:, :, :gh * p, :gw * p].reshape(B, C, gh, p, gw, p).transpose(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
    return patches @ conv_w.reshape(w, -1).T


def clip_vit_hidden(sd: Dict[str, Array], images: Array, heads: int, prefix: str = "") -> Array:
    """CLIPViTEncoder.forward up to and including the transformer (image_encoder.py:82-108) -> [B,S,w]."""
    x = patch_embed(images, sd[prefix + "conv.weight"])
    B = x.shape[0]
    cls = np.broadcast_to(sd[prefix + "cls_token_embedding"], (B, 1, x.shape[2]))
    x = np.concatenate([cls, x], axis=1) + sd[prefix + "positional_embedding"]
    x = layer_norm(x, sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"], 1e-5)
    for i in range(_num_layers(sd, prefix)):
        x = encoder_layer(x, sd, f"{prefix}encoder.layers.{i}.", heads, causal=False)
    return x


def clip_vit_forward(sd: Dict[str, Array], images: Array, heads: int, prefix: str = "", dtype=np.float32) -> Array:
    """CLIPViTEncoder.forward (models/clip/image_encoder.py:82-113) -> [B,E] (not normalized)."""
    sd = _cast({k: v for k, v in sd.items() if k.startswith(prefix)}, dtype)
    x = clip_vit_hidden(sd, np.asarray(images).astype(dtype), heads, prefix)
    x = layer_norm(x[:, 0, :], sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"], 1e-5)
    return x @ sd[prefix + "projection"]


def clip_text_forward(
    sd: Dict[str, Array],
    text: Array,
    heads: int,
    prefix: str = "",
    return_hidden_state: bool = False,
    dtype=np.float32,
) -> Array:
    """CLIPTextEncoder.forward (models/clip/text_encoder.py:113-134)."""
    sd = _cast({k: v for k, v in sd.items() if k.startswith(prefix)}, dtype)
    text = np.asarray(text)
    x = sd[prefix + "token_embedding.weight"][text] + sd[prefix + "positional_embedding"]
    for i in range(_num_layers(sd, prefix)):
        x = encoder_layer(x, sd, f"{prefix}encoder.layers.{i}.", heads, causal=True)
    hidden = layer_norm(x, sd[prefix + "ln_final.weight"], sd[prefix + "ln_final.bias"], 1e-5)
    if return_hidden_state:
        return hidden
    eot = text.argmax(axis=-1)  # first index of the max id, as torch.argmax
    pooled = hidden[np.arange(hidden.shape[0]), eot]
    return pooled @ sd[prefix + "projection.weight"].T


def clip_forward(sd, images, text, vision_heads: int, text_heads: int, dtype=np.float32):
    """CLIP.forward (models/clip/model.py:65-74): both towers + L2 normalize."""
    a = clip_vit_forward(sd, images, vision_heads, "encoder_a.", dtype)
    b = clip_text_forward(sd, text, text_heads, "encoder_b.", False, dtype)
    return l2_normalize(a), l2_normalize(b)


# ----------------------------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------------------------
def cross_entropy(logits: Array, labels: Array, label_smoothing: float = 0.0, reduction: str = "mean"):
    """F.cross_entropy over rows (contrastive_loss_with_temperature.py:105-106)."""
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    logp = logits - lse[:, None]
    nll = -logp[np.arange(logits.shape[0]), labels]
    if label_smoothing:
        s = logits.dtype.type(label_smoothing)
        nll = (1 - s) * nll + s * (-logp.mean(axis=1))
    if reduction == "mean":
        return nll.mean() if nll.size else np.asarray(np.nan, dtype=logits.dtype)
    if reduction == "sum":
        return nll.sum()
    return nll


def contrastive_loss_with_temperature(
    embeddings_a: Array,
    embeddings_b: Array,
    logit_scale: float,
    embeddings_a_all: Optional[Array] = None,
    embeddings_b_all: Optional[Array] = None,
    rank: int = 0,
    mask: Optional[Array] = None,
    label_smoothing: float = 0.0,
    dtype=np.float32,
):
    """contrastive_loss_with_temperature (modules/losses/contrastive_loss_with_temperature.py:50-115).

    `embeddings_*_all` are the concatenated all-gathered features ([W*B,E], :35-47); None = single process.
    Returns a dict with the fields of ContrastiveLossOutput (:17-23).
    """
    a = np.asarray(embeddings_a).astype(dtype)
    b = np.asarray(embeddings_b).astype(dtype)
    a_all = a if embeddings_a_all is None else np.asarray(embeddings_a_all).astype(dtype)
    b_all = b if embeddings_b_all is None else np.asarray(embeddings_b_all).astype(dtype)
    temperature = np.exp(np.asarray(logit_scale, dtype=dtype))  # :81
    B = a.shape[0]
    labels = B * rank + np.arange(B)  # :38-41
    logits_a = (a @ b_all.T) * temperature  # :90-92
    logits_b = (b @ a_all.T) * temperature  # :93-95
    if mask is not None:  # :97-100
        mask = np.asarray(mask).astype(bool)
        logits_a, logits_b, labels = logits_a[mask], logits_b[mask], labels[mask]
    loss_a = cross_entropy(logits_a, labels, label_smoothing)
    loss_b = cross_entropy(logits_b, labels, label_smoothing)
    return {
        "loss": (loss_a + loss_b) / 2,
        "logits_a": logits_a,
        "logits_b": logits_b,
        "loss_a": loss_a,
        "loss_b": loss_b,
    }


def clamp_logit_scale(value: float, lo: Optional[float], hi: Optional[float]) -> float:
    """ContrastiveLossWithTemperature.forward's in-place clamp (…:193)."""
    if lo is not None:
        value = max(value, lo)
    if hi is not None:
        value = min(value, hi)
    return value


# ----------------------------------------------------------------------------------------------
# algorithmic work, used by bench.py for the roofline figure (BASELINE.md §3 formula)
# ----------------------------------------------------------------------------------------------
def tower_flops(S: int, d: int, ff: int, layers: int) -> float:
    return layers * (2 * S * d * 3 * d + 4 * S * S * d + 2 * S * d * d + 4 * S * d * ff)


def clip_flops_per_pair(
    image_size=224, patch=16, vw=768, vl=12, tw=512, tff=2048, tl=12, ctx=77, E=512
) -> float:
    S = (image_size // patch) ** 2 + 1
    vision = tower_flops(S, vw, 4 * vw, vl) + 2 * (S - 1) * 3 * patch * patch * vw + 2 * vw * E
    text = tower_flops(ctx, tw, tff, tl) + 2 * tw * E
    return float(vision + text)


# ==============================================================================================
# FLAVA dual encoder (+ multimodal encoder) and its global contrastive loss
#   restated from models/flava/{model,image_encoder,text_encoder,transformer}.py,
#   modules/layers/{attention,mlp,text_embedding}.py, modules/encoders/bert_text_encoder.py,
#   modules/losses/flava.py  — all arithmetic of this part is explicit in the reference (no torch fused ops)
# ==============================================================================================
def gelu_erf(x: Array) -> Array:
    """nn.GELU() (exact erf form; models/flava/model.py:79,435,447,459 pass nn.GELU as activation)."""
    from scipy.special import erf

    return (0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))).astype(x.dtype)


def flava_attention(x: Array, sd, prefix: str, heads: int, key_mask: Optional[Array]):
    """MultiHeadAttention + SelfAttention (modules/layers/attention.py:120-241). key_mask [B,S]: 1 = attend."""
    B, S, d = x.shape
    dh = d // heads
    g = lambda k: sd[prefix + k]

    def proj(name):
        return (x @ g(name + ".weight").T + g(name + ".bias")).reshape(B, S, heads, dh).transpose(0, 2, 1, 3)

    q, k, v = proj("query"), proj("key"), proj("value")
    attn = (q @ k.transpose(0, 1, 3, 2)) / np.sqrt(x.dtype.type(dh))  # :220-221
    if key_mask is not None:  # :227-228 masked_fill(attention_mask == 0, -inf); mask broadcast [B,1,1,S]
        attn = np.where(np.asarray(key_mask)[:, None, None, :] == 0, -np.inf, attn).astype(x.dtype)
    probs = softmax_lastdim(attn)  # :230
    a = (probs @ v).transpose(0, 2, 1, 3).reshape(B, S, d)  # :239 + merge_multihead
    return a @ g("output.weight").T + g("output.bias"), probs


def flava_encoder_layer(x: Array, sd, prefix: str, heads: int, eps: float, key_mask: Optional[Array], activation=None):
    """TransformerEncoderLayer._forward_prenorm (models/flava/transformer.py:155-176); activation defaults to the erf GELU
    every FLAVA factory passes (the class default is nn.ReLU, used by the reference's layer KAT)."""
    act = gelu_erf if activation is None else activation
    g = lambda k: sd[prefix + k]
    h = layer_norm(x, g("attention_layernorm.weight"), g("attention_layernorm.bias"), eps)
    a, probs = flava_attention(h, sd, prefix + "attention.", heads, key_mask)
    x1 = a + x
    h2 = layer_norm(x1, g("feedforward_layernorm.weight"), g("feedforward_layernorm.bias"), eps)
    ff = act(h2 @ g("feedforward.model.0.weight").T + g("feedforward.model.0.bias"))
    ff = ff @ g("feedforward.model.2.weight").T + g("feedforward.model.2.bias")
    return x1 + ff, probs


def flava_transformer_encoder(x: Array, sd, prefix: str, heads: int, eps: float, key_mask: Optional[Array] = None):
    """TransformerEncoder.forward (models/flava/transformer.py:255-293) with both return flags on."""
    hidden, attns = [], []
    n = 0
    while f"{prefix}layer.{n}.attention.query.weight" in sd:
        hidden.append(x)
        x, p = flava_encoder_layer(x, sd, f"{prefix}layer.{n}.", heads, eps, key_mask)
        attns.append(p)
        n += 1
    hidden.append(x)
    return x, hidden, attns


def flava_pooler(hidden: Array, sd, prefix: str) -> Array:
    """Pooler: tanh(Linear(hidden[:, 0])) (modules/losses/flava.py:84-97); a pooler without parameters is nn.Identity (the
    reference's encoder KATs build their encoders with pooler=nn.Identity())."""
    if prefix + "dense.weight" not in sd:
        return hidden
    return np.tanh(hidden[:, 0] @ sd[prefix + "dense.weight"].T + sd[prefix + "dense.bias"])


def _cubic_weights(t: float):
    """Cubic-convolution weights, A = -0.75 (torch aten/src/ATen/native/UpSample.h get_cubic_upsample_coefficients)."""
    A = -0.75
    c1 = lambda x: ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0          # |x| <= 1
    c2 = lambda x: ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A    # 1 < |x| < 2
    return np.array([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)])


def bicubic_resize_grid(grid: Array, h0: int, w0: int, scale_h: float, scale_w: float) -> Array:
    """F.interpolate(grid[None].permute(0,3,1,2), scale_factor=(scale_h, scale_w), mode="bicubic", align_corners=False) for a
    [n, n, d] grid -> [h0, w0, d].  The algorithm is PyTorch's upsample_bicubic2d (third-party dependency of the reference, torch
    2.10, aten/src/ATen/native/UpSampleBicubic2d.cpp — not in /root/reference): source coordinate (o + 0.5)/scale - 0.5, taps at
    floor - 1 .. floor + 2 clamped to the grid, x pass then y pass.  Pinned by tests/golden/flava_cls_interp.npz (reference output)."""
    g = np.asarray(grid)
    n_y, n_x, d = g.shape
    out = np.zeros((h0, w0, d), dtype=g.dtype)
    for oy in range(h0):
        ry = (oy + 0.5) / scale_h - 0.5
        iy = int(np.floor(ry))
        wy = _cubic_weights(ry - iy)
        ys = np.clip(np.arange(iy - 1, iy + 3), 0, n_y - 1)
        for ox in range(w0):
            rx = (ox + 0.5) / scale_w - 0.5
            ix = int(np.floor(rx))
            wx = _cubic_weights(rx - ix)
            xs = np.clip(np.arange(ix - 1, ix + 3), 0, n_x - 1)
            patch = g[np.ix_(ys, xs)]                                   # [4, 4, d]
            out[oy, ox] = np.einsum("i,ijd,j->d", wy, patch, wx)
    return out


def flava_interpolate_pos_encoding(pos: Array, npatch: int, height: int, width: int, patch: int) -> Array:
    """ImageEmbeddings.interpolate_pos_encoding (models/flava/image_encoder.py:102-137).  pos [1, 1+n, d]."""
    n = pos.shape[1] - 1
    if npatch == n and height == width:
        return pos
    dim = pos.shape[-1]
    side = int(np.sqrt(n))
    h0, w0 = height // patch + 0.1, width // patch + 0.1
    grid = bicubic_resize_grid(pos[0, 1:].reshape(side, side, dim), int(h0), int(w0), h0 / np.sqrt(n), w0 / np.sqrt(n))
    return np.concatenate([pos[:, :1], grid.reshape(1, -1, dim)], axis=1)


def mlp_forward(x: Array, sd, prefix: str, n_linear: int, stride: int = 2) -> Array:
    """modules/layers/mlp.py:13-66 in eval mode with nn.ReLU between the Linears (FLAVAForClassification's classifier; Sequential
    indices advance by `stride` = 3 when a Dropout follows each activation, as in the default classifier)."""
    for i in range(n_linear):
        k = prefix + f"model.{i * stride}."
        x = x @ sd[k + "weight"].T + sd[k + "bias"]
        if i + 1 < n_linear:
            x = np.maximum(x, 0)
    return x


def flava_classification(sd, hidden_state: Array, labels: Array, n_linear: int, stride: int = 3, cls_index: int = 0):
    """FLAVAForClassification.forward after the encoder (models/flava/model.py:411-418): scores = classifier(h[:, cls_index]),
    loss = CrossEntropyLoss()(scores, labels)."""
    scores = mlp_forward(hidden_state[:, cls_index], sd, "classifier.", n_linear, stride)
    return scores, cross_entropy(scores, labels)


def flava_image_encoder(sd, prefix: str, pixel_values: Array, heads: int, image_patches_mask: Optional[Array] = None,
                        eps: float = 1e-12, dtype=np.float32, interpolate_pos_encoding: bool = False, final_eps: Optional[float] = None):
    """ImageTransformer.forward (models/flava/image_encoder.py:204-234) incl. ImageEmbeddings (:139-177)."""
    sd = _cast({k: v for k, v in sd.items() if k.startswith(prefix)}, dtype)
    x = np.asarray(pixel_values).astype(dtype)
    w = sd[prefix + "embeddings.patch_embeddings.projection.weight"]
    emb = patch_embed(x, w) + sd[prefix + "embeddings.patch_embeddings.projection.bias"]
    B = emb.shape[0]
    if image_patches_mask is not None:  # :151-156
        m = np.asarray(image_patches_mask).astype(dtype)[..., None]
        emb = emb * (1 - m) + sd[prefix + "embeddings.mask_token"].reshape(1, 1, -1) * m
    cls = np.broadcast_to(sd[prefix + "embeddings.cls_token"].reshape(1, 1, -1), (B, 1, emb.shape[2]))
    pos = sd[prefix + "embeddings.position_embeddings"].reshape(1, -1, emb.shape[2])
    if interpolate_pos_encoding:  # :170-173
        pos = flava_interpolate_pos_encoding(pos, emb.shape[1], x.shape[2], x.shape[3], w.shape[2]).astype(dtype)
    emb = np.concatenate([cls, emb], axis=1) + pos
    last, hidden, attns = flava_transformer_encoder(emb, sd, prefix + "encoder.", heads, eps)
    seq = layer_norm(last, sd[prefix + "layernorm.weight"], sd[prefix + "layernorm.bias"], eps if final_eps is None else final_eps)
    return {"last_hidden_state": seq, "pooler_output": flava_pooler(seq, sd, prefix + "pooler."), "hidden_states": hidden,
            "attentions": attns}


def flava_text_encoder(sd, prefix: str, input_ids: Array, heads: int, pad_token_id: int = 0, eps: float = 1e-12,
                       attention_mask: Optional[Array] = None, dtype=np.float32, final_eps: Optional[float] = None):
    """BERTTextEncoder.forward (modules/encoders/bert_text_encoder.py:67-120) + BERTTextEmbeddings (text_embedding.py:74-104)."""
    sd = _cast({k: v for k, v in sd.items() if k.startswith(prefix)}, dtype)
    ids = np.asarray(input_ids)
    B, S = ids.shape
    if attention_mask is None:  # :86-89
        attention_mask = (ids != pad_token_id).astype(np.int64)
    e = (sd[prefix + "embeddings.word_embeddings.weight"][ids] + sd[prefix + "embeddings.position_embeddings.weight"][np.arange(S)][None]
         + sd[prefix + "embeddings.token_type_embeddings.weight"][0][None, None])
    e = layer_norm(e, sd[prefix + "embeddings.layer_norm.weight"], sd[prefix + "embeddings.layer_norm.bias"], eps)
    last, hidden, attns = flava_transformer_encoder(e, sd, prefix + "encoder.", heads, eps, key_mask=attention_mask)
    seq = layer_norm(last, sd[prefix + "layernorm.weight"], sd[prefix + "layernorm.bias"], eps if final_eps is None else final_eps)
    return {"last_hidden_state": seq, "pooler_output": flava_pooler(seq, sd, prefix + "pooler."), "hidden_states": hidden,
            "attentions": attns}


def flava_mm_encoder(sd, prefix: str, hidden_states: Array, heads: int, eps: float = 1e-12, dtype=np.float32):
    """FLAVATransformerWithoutEmbeddings.forward (models/flava/transformer.py:47-77)."""
    sd = _cast({k: v for k, v in sd.items() if k.startswith(prefix)}, dtype)
    x = np.asarray(hidden_states).astype(dtype)
    cls = np.broadcast_to(sd[prefix + "cls_token"].reshape(1, 1, -1), (x.shape[0], 1, x.shape[2]))
    x = np.concatenate([cls, x], axis=1)
    last, hidden, attns = flava_transformer_encoder(x, sd, prefix + "encoder.", heads, eps)
    seq = layer_norm(last, sd[prefix + "layernorm.weight"], sd[prefix + "layernorm.bias"], eps)
    return {"last_hidden_state": seq, "pooler_output": flava_pooler(seq, sd, prefix + "pooler."), "hidden_states": hidden,
            "attentions": attns}


def flava_model_forward(sd, image: Array, text: Array, heads: int, mm_heads: int, image_patches_mask=None, text_masked=None,
                        dtype=np.float32):
    """FLAVAModel.forward with required_embedding='mm', skip_unmasked_mm_encoder=True (models/flava/model.py:127-231)."""
    sdc = _cast(sd, dtype)
    lin = lambda name, x: x @ sdc[name + ".weight"].T + sdc[name + ".bias"]
    img = flava_image_encoder(sd, "image_encoder.", image, heads, dtype=dtype)
    txt = flava_text_encoder(sd, "text_encoder.", text, heads, dtype=dtype)
    out = {"image": img, "text": txt,
           "projected_image_embeddings": lin("image_projection", img["last_hidden_state"][:, 0]),
           "projected_text_embeddings": lin("text_projection", txt["last_hidden_state"][:, 0])}
    img_m = flava_image_encoder(sd, "image_encoder.", image, heads, image_patches_mask, dtype=dtype)
    out["image_masked"] = img_m
    if text_masked is not None:
        txt_m = flava_text_encoder(sd, "text_encoder.", text_masked, heads, dtype=dtype)
        out["text_masked"] = txt_m
        fused = np.concatenate([lin("image_to_mm_projection", img_m["hidden_states"][-1]),
                                lin("text_to_mm_projection", txt_m["hidden_states"][-1])], axis=1)  # :294-297
        out["multimodal_masked"] = flava_mm_encoder(sd, "mm_encoder.", fused, mm_heads, dtype=dtype)
    return out


def flava_global_contrastive_loss(image_sequence: Array, text_sequence: Array, logit_scale: float, mask: Optional[Array] = None,
                                  image_all: Optional[Array] = None, text_all: Optional[Array] = None, rank: int = 0,
                                  dtype=np.float32):
    """FLAVAGlobalContrastiveLoss.forward (modules/losses/flava.py:261-293): normalise (dim=-1), clamp logit_scale to
    [0, 4.6052], contrastive_loss_with_temperature(image, text, mask)."""
    img = l2_normalize(np.asarray(image_sequence).astype(dtype))
    txt = l2_normalize(np.asarray(text_sequence).astype(dtype))
    scale = clamp_logit_scale(float(logit_scale), 0.0, 4.6052)
    o = contrastive_loss_with_temperature(img, txt, scale, image_all, text_all, rank, mask, dtype=dtype)
    return {"loss": o["loss"], "image_logits": o["logits_a"], "text_logits": o["logits_b"], "image_loss": o["loss_a"],
            "text_loss": o["loss_b"], "image_embedding": img, "text_embedding": txt, "logit_scale": scale}


# ------------------------------------------------------------------------------------------------------------------------
# FLAVA pre-training heads and loss (modules/losses/flava.py:100-484)
# ------------------------------------------------------------------------------------------------------------------------
def cross_entropy_ignore(logits: Array, labels: Array, ignore_index: int = -1):
    """nn.CrossEntropyLoss(ignore_index=...) with mean reduction over the kept rows (NaN when none is kept, like torch)."""
    logits = np.asarray(logits)
    labels = np.asarray(labels).reshape(-1)
    keep = labels != ignore_index
    lg = logits.reshape(-1, logits.shape[-1])[keep]
    lb = labels[keep]
    if lg.shape[0] == 0:
        return lg.dtype.type(np.nan)
    m = lg.max(axis=1, keepdims=True)
    lse = (m + np.log(np.exp(lg - m).sum(axis=1, keepdims=True)))[:, 0]
    return (lse - lg[np.arange(lg.shape[0]), lb]).mean(dtype=lg.dtype)


def masked_prediction_head(h: Array, sd, prefix: str, eps: float = 1e-5) -> Array:
    """MaskedPredictionHead.forward (:174-179): dense -> erf GELU -> Fp32LayerNorm -> decoder (+ the tied output bias)."""
    x = gelu_erf(h @ sd[prefix + "dense.weight"].T + sd[prefix + "dense.bias"])
    x = layer_norm(x, sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"], eps)
    return x @ sd[prefix + "decoder.weight"].T + sd[prefix + "bias"]


def masked_prediction_loss(hidden: Array, labels: Optional[Array], sd, prefix: str, ignore_index: int = -1):
    """MaskedPredictionLoss.forward (:206-238): only the labelled positions go through the head."""
    hidden = np.asarray(hidden)
    if labels is not None:
        labels = np.asarray(labels)
        keep = labels != ignore_index
        seq, lab = hidden[keep], labels[keep]
    else:
        seq, lab = hidden, None
    logits = masked_prediction_head(seq, sd, prefix + "cls.")
    if lab is None:
        return {"logits": logits, "loss": logits.dtype.type(0)}
    return {"logits": logits, "loss": cross_entropy_ignore(logits, lab, ignore_index)}


def itm_loss(hidden: Array, labels: Optional[Array], sd, prefix: str, ignore_index: int = -1):
    """ITMLoss.forward (:122-140): Pooler -> Linear(hidden, 2) -> CE."""
    pooled = flava_pooler(np.asarray(hidden), sd, prefix + "pooler.")
    scores = pooled @ sd[prefix + "cls.seq_relationship.weight"].T + sd[prefix + "cls.seq_relationship.bias"]
    if labels is None:
        return {"logits": scores, "loss": scores.dtype.type(0)}
    return {"logits": scores, "loss": cross_entropy_ignore(scores, labels, ignore_index)}


def flava_pretraining_loss(sd, image_masked_sequence=None, text_masked_sequence=None, multimodal_masked_sequence=None,
                           itm_labels=None, mim_labels=None, mlm_labels=None, projected_image_embeddings=None,
                           projected_text_embeddings=None, dtype=np.float32):
    """FLAVAPretrainingLoss.forward with all weights 1 (:370-484).  `sd` = the loss module's state_dict."""
    sd = _cast(sd, dtype)
    f = lambda a: None if a is None else np.asarray(a).astype(dtype)
    ims, tms, mms = f(image_masked_sequence), f(text_masked_sequence), f(multimodal_masked_sequence)
    out = {}
    pos_mask = None
    if ims is not None and mms is None:  # unimodal MIM (:391-402)
        start = -mim_labels.shape[1] if mim_labels is not None else 1
        out["mim"] = masked_prediction_loss(ims[:, start:, :], mim_labels, sd, "mim_loss.")
    if tms is not None and mms is None:  # unimodal MLM (:405-416)
        start = -mlm_labels.shape[1] if mlm_labels is not None else 1
        out["mlm"] = masked_prediction_loss(tms[:, start:, :], mlm_labels, sd, "mlm_loss.")
    if mms is not None:  # ITM + row filter (:418-437)
        if itm_labels is not None:
            pos = np.asarray(itm_labels) != 0
            pos_mask = pos if pos.any() else np.ones_like(pos)
        else:
            pos_mask = np.ones(mms.shape[0], dtype=bool)
        out["itm"] = itm_loss(mms, itm_labels, sd, "itm_loss.")
        mms = mms[pos_mask]
        if mlm_labels is not None:
            mlm_labels = np.asarray(mlm_labels)[pos_mask]
        if mim_labels is not None:
            mim_labels = np.asarray(mim_labels)[pos_mask]
        start = -mlm_labels.shape[1] if mlm_labels is not None else -(tms.shape[1] - 1)  # :439-452
        out["mmm_text"] = masked_prediction_loss(mms[:, start:, :], mlm_labels, sd, "mmm_loss.mlm.")
        total = mim_labels.shape[1] if mlm_labels is not None else ims.shape[1] - 1  # :454-469 (2 CLS rows skipped)
        out["mmm_image"] = masked_prediction_loss(mms[:, 2:2 + total, :], mim_labels, sd, "mmm_loss.mim.")
    if projected_image_embeddings is not None and projected_text_embeddings is not None:  # :471-482
        out["global_contrastive"] = flava_global_contrastive_loss(
            projected_image_embeddings, projected_text_embeddings, float(sd["contrastive_loss.logit_scale"]), pos_mask, dtype=dtype)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# CoCa: models/coca/*.py, modules/encoders/vision_transformer.py, modules/layers/{transformer,
# multi_head_attention,attention_pooler,patch_embedding}.py
# ------------------------------------------------------------------------------------------------------------------------
def sdpa(q: Array, k: Array, v: Array, attend: Optional[Array] = None, causal: bool = False) -> Array:
    """F.scaled_dot_product_attention on [B,H,S,dh] arrays; `attend` is a boolean mask broadcastable to [B,H,Sq,Sk]
    (True = take part), `causal` the top-left aligned lower-triangular mask."""
    s = (q @ k.transpose(0, 1, 3, 2)) / np.sqrt(q.dtype.type(q.shape[-1]))
    Sq, Sk = s.shape[-2:]
    if causal:
        s = np.where(np.tril(np.ones((Sq, Sk), dtype=bool)), s, -np.inf)
    if attend is not None:
        s = np.where(np.asarray(attend).astype(bool), s, -np.inf)
    return softmax_lastdim(s.astype(q.dtype)) @ v


def _heads(x: Array, h: int) -> Array:
    B, S, d = x.shape
    return x.reshape(B, S, h, d // h).transpose(0, 2, 1, 3)


def _merge(x: Array) -> Array:
    B, h, S, dh = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, S, h * dh)


def mh_self_attention(x: Array, sd, prefix: str, heads: int, attend=None, causal=False) -> Array:
    """MultiHeadSelfAttention.forward (modules/layers/multi_head_attention.py:38-76): packed input_proj, SDPA, output_proj."""
    qkv = x @ sd[prefix + "input_proj.weight"].T + sd[prefix + "input_proj.bias"]
    q, k, v = np.split(qkv, 3, axis=-1)
    a = _merge(sdpa(_heads(q, heads), _heads(k, heads), _heads(v, heads), attend, causal))
    return a @ sd[prefix + "output_proj.weight"].T + sd[prefix + "output_proj.bias"]


def mha_with_cache(q_in: Array, kv_in: Array, sd, prefix: str, heads: int, attend=None, causal=False, past=None, use_cache=False):
    """MultiHeadAttentionWithCache.forward (…:115-180): separate q/k/v projections (k, v from kv_in); `past` = (key, value)
    [B,H,Sp,dh] is concatenated in front of the new keys / values (:158-161); use_cache -> (output, (key, value)) (:177-179)."""
    lin = lambda name, x: x @ sd[prefix + name + ".weight"].T + (sd[prefix + name + ".bias"] if prefix + name + ".bias" in sd else 0)
    q, k, v = _heads(lin("q_proj", q_in), heads), _heads(lin("k_proj", kv_in), heads), _heads(lin("v_proj", kv_in), heads)
    if past is not None:
        k, v = np.concatenate([past[0], k], axis=2), np.concatenate([past[1], v], axis=2)
    out = lin("output_proj", _merge(sdpa(q, k, v, attend, causal)))
    return (out, (k, v)) if use_cache else out


def _ffn(x: Array, sd, prefix: str, activation=None) -> Array:
    """MLP(d, d, dim_feedforward) of the layers (modules/layers/mlp.py): Linear -> activation -> Linear; the models on the path pass
    nn.GELU (default here), the class default — used by the reference's constant-weight KATs — is nn.ReLU."""
    h = (activation or gelu_erf)(x @ sd[prefix + "model.0.weight"].T + sd[prefix + "model.0.bias"])
    return h @ sd[prefix + "model.2.weight"].T + sd[prefix + "model.2.bias"]


def relu(x: Array) -> Array:
    return np.maximum(x, 0)


def layers_encoder_layer(x: Array, sd, prefix: str, heads: int, eps: float, norm_first: bool = True, attend=None, activation=None) -> Array:
    """layers.transformer.TransformerEncoderLayer (modules/layers/transformer.py:96-156), GELU feed-forward unless `activation`."""
    ln = lambda name, t: layer_norm(t, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"], eps)
    if norm_first:
        a = mh_self_attention(ln("attention_layernorm", x), sd, prefix + "attention.", heads, attend) + x
        return a + _ffn(ln("feedforward_layernorm", a), sd, prefix + "feedforward.", activation)
    a = ln("attention_layernorm", mh_self_attention(x, sd, prefix + "attention.", heads, attend) + x)
    return ln("feedforward_layernorm", a + _ffn(a, sd, prefix + "feedforward.", activation))


def layers_encoder(x: Array, sd, prefix: str, heads: int, eps: float, norm_first: bool = True, final_eps: Optional[float] = None,
                   activation=None):
    """layers.transformer.TransformerEncoder.forward (:222-262) with return_hidden_states=True."""
    hidden, n = [], 0
    while f"{prefix}layer.{n}.attention.input_proj.weight" in sd:
        hidden.append(x)
        x = layers_encoder_layer(x, sd, f"{prefix}layer.{n}.", heads, eps, norm_first, activation=activation)
        n += 1
    hidden.append(x)
    if final_eps:
        x = layer_norm(x, sd[prefix + "final_layer_norm.weight"], sd[prefix + "final_layer_norm.bias"], final_eps)
    return x, hidden


def layers_decoder_layer(x: Array, enc: Optional[Array], sd, prefix: str, heads: int, eps: float, attend=None, past=None, use_cache=False,
                         norm_first: bool = True, activation=None):
    """TransformerDecoderLayer._forward_prenorm (:398-433): self-attention (optionally over cached keys / values, :336-359), optional
    cross-attention, feed-forward; norm_first=False = _forward_postnorm (:435-472).  use_cache -> (output, present_key_value)."""
    ln = lambda name, t: layer_norm(t, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"], eps)
    if not norm_first:
        r = mha_with_cache(x, x, sd, prefix + "attention.", heads, attend, past=past, use_cache=use_cache)
        present = None
        if use_cache:
            r, present = r
        a = ln("attention_layernorm", r + x)
        if prefix + "cross_attention.q_proj.weight" in sd:
            a = ln("cross_attention_layernorm", mha_with_cache(a, enc, sd, prefix + "cross_attention.", heads) + a)
        y = ln("feedforward_layernorm", a + _ffn(a, sd, prefix + "feedforward.", activation))
        return (y, present) if use_cache else y
    h = ln("attention_layernorm", x)
    r = mha_with_cache(h, h, sd, prefix + "attention.", heads, attend, past=past, use_cache=use_cache)
    present = None
    if use_cache:
        r, present = r
    a = r + x
    if enc is not None and prefix + "cross_attention.q_proj.weight" in sd:
        a = mha_with_cache(ln("cross_attention_layernorm", a), enc, sd, prefix + "cross_attention.", heads) + a
    y = a + _ffn(ln("feedforward_layernorm", a), sd, prefix + "feedforward.", activation)
    return (y, present) if use_cache else y


def layers_decoder(x: Array, enc: Optional[Array], sd, prefix: str, heads: int, eps: float, attend=None, final_eps=None) -> Array:
    n = 0
    while f"{prefix}layer.{n}.attention.q_proj.weight" in sd:
        x = layers_decoder_layer(x, enc, sd, f"{prefix}layer.{n}.", heads, eps, attend)
        n += 1
    if final_eps:
        x = layer_norm(x, sd[prefix + "final_layer_norm.weight"], sd[prefix + "final_layer_norm.bias"], final_eps)
    return x


def attention_pooler(x: Array, sd, prefix: str, heads: int, eps: float = 1e-5) -> Array:
    """AttentionPooler.forward (modules/layers/attention_pooler.py:49-70)."""
    k = layer_norm(x, sd[prefix + "ln_k.weight"], sd[prefix + "ln_k.bias"], eps)
    q = layer_norm(sd[prefix + "query"], sd[prefix + "ln_q.weight"], sd[prefix + "ln_q.bias"], eps)
    q = np.broadcast_to(q[None], (x.shape[0], *q.shape)).astype(x.dtype)
    out = mha_with_cache(q, k, sd, prefix + "attn.", heads)
    return layer_norm(out, sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"], eps)


def layers_patch_embeddings(pixel_values: Array, sd, prefix: str) -> Array:
    """layers.patch_embedding.PatchEmbeddings.forward (modules/layers/patch_embedding.py:98-152), eval mode, no masking."""
    emb = patch_embed(pixel_values, sd[prefix + "conv_projection.weight"]) + sd[prefix + "conv_projection.bias"]
    pos = sd[prefix + "position_embeddings"]
    if prefix + "cls_token" in sd:
        emb = emb + pos[:, 1:, :]
        cls = np.broadcast_to(sd[prefix + "cls_token"] + pos[:, :1, :], (emb.shape[0], 1, emb.shape[2]))
        return np.concatenate([cls, emb], axis=1)
    return emb + pos


def coca_text_mask(input_ids: Array, pad_idx: int = 0, padding_mask: Optional[Array] = None) -> Array:
    """CoCaTextDecoder.build_mask (models/coca/text_decoder.py:178-194) -> bool [B,1,S+1,S+1]."""
    ids = np.asarray(input_ids)
    B, S = ids.shape
    pm = (ids != pad_idx) if padding_mask is None else np.asarray(padding_mask).astype(bool)
    full = np.ones((B, S + 1, S + 1), dtype=bool)
    full[:, S, 1:] = pm  # F.pad(mask[:, None], (1, 0, S, 0), value=1): one new column on the LEFT, S new rows on top
    return (full & np.tril(np.ones((S + 1, S + 1), dtype=bool)))[:, None]


def coca_text_decoder(sd, prefix: str, input_ids: Array, heads: int, pad_idx: int = 0, eps: float = 1e-5, padding_mask=None,
                      dtype=np.float32):
    """CoCaTextDecoder.forward with embed_cls=True (…:196-252) -> (pooled [B,out], tokens [B,S,d])."""
    sd = _cast({k: v for k, v in sd.items() if k.startswith(prefix)}, dtype)
    ids = np.asarray(input_ids)
    npos = sd[prefix + "embeddings.position_embeddings"].shape[0]
    if ids.shape[1] == npos:
        ids = ids[:, :-1]
        if padding_mask is not None and np.asarray(padding_mask).shape[1] == npos:
            padding_mask = np.asarray(padding_mask)[:, :-1]
    emb = sd[prefix + "embeddings.token_embeddings.weight"][ids]
    cls = np.broadcast_to(sd[prefix + "embeddings.cls_embedding"].reshape(1, 1, -1), (ids.shape[0], 1, emb.shape[2]))
    x = np.concatenate([emb, cls], axis=1) + sd[prefix + "embeddings.position_embeddings"]
    mask = coca_text_mask(ids, pad_idx, padding_mask)
    h = layers_decoder(x, None, sd, prefix + "transformer_decoder.", heads, eps, attend=mask)
    pooled, tokens = h[:, -1], h[:, :-1]
    pooled = layer_norm(pooled, sd[prefix + "ln_final.weight"], sd[prefix + "ln_final.bias"], eps)
    return pooled @ sd[prefix + "text_projection.weight"].T, tokens


def coca_model_forward(sd, images: Array, texts: Array, vision_heads: int, text_heads: int, fusion_heads: int, pooler_heads: int,
                       cascaded: bool, pad_idx: int = 0, dtype=np.float32):
    """CoCaModel.forward (models/coca/coca_model.py:76-135) for coca_vit(...) models."""
    sdc = _cast(sd, dtype)
    x = layers_patch_embeddings(np.asarray(images).astype(dtype), sdc, "vision_encoder.embeddings.")
    img, _ = layers_encoder(x, sdc, "vision_encoder.encoder.", vision_heads, 1e-5, True, None)
    if cascaded:
        cap = attention_pooler(img, sdc, "vision_pooler.poolers.0.", pooler_heads)
        con = attention_pooler(cap, sdc, "vision_pooler.poolers.1.", pooler_heads)  # [B,1,D]: the reference keeps dim 1
    else:
        pooled = attention_pooler(img, sdc, "vision_pooler.", pooler_heads)
        con, cap = pooled[:, 0], pooled[:, 1:]
    con = con @ sdc["vision_proj.weight"].T
    con = con / np.maximum(np.linalg.norm(con, axis=-1, keepdims=True), 1e-12)
    tp, tokens = coca_text_decoder(sdc, "text_decoder.", texts, text_heads, pad_idx, dtype=dtype)
    tp = tp / np.maximum(np.linalg.norm(tp, axis=-1, keepdims=True), 1e-12)
    S = tokens.shape[1]
    mm = layers_decoder(tokens, cap, sdc, "multimodal_decoder.transformer_decoder.", fusion_heads, 1e-5,
                        attend=np.tril(np.ones((S, S), dtype=bool)), final_eps=1e-5)
    if "multimodal_decoder.output_projection.weight" in sdc:
        mm = mm @ sdc["multimodal_decoder.output_projection.weight"].T
    return {"image_pooled_output": con.astype(dtype), "text_pooled_output": tp.astype(dtype), "multimodal_embeddings": mm}


def coca_pretraining_losses(out, texts: Array, logit_scale: float, pad_idx: int = 0, dtype=np.float32):
    """CoCaForPretraining.forward (…:441-466): contrastive loss on the pooled outputs + captioning CE (ignore_index = pad)."""
    texts = np.asarray(texts)
    labels = texts[:, 1:]
    con = contrastive_loss_with_temperature(out["image_pooled_output"], out["text_pooled_output"],
                                            clamp_logit_scale(float(logit_scale), np.log(1.0), np.log(100.0)), dtype=dtype)["loss"]
    V = out["multimodal_embeddings"].shape[-1]
    cap = cross_entropy_ignore(out["multimodal_embeddings"].reshape(-1, V), labels.reshape(-1), pad_idx)
    return {"contrastive": con, "captioning": cap}


# ------------------------------------------------------------------------------------------------------------------------
# Backward of the contrastive loss (what torch autograd computes for modules/losses/contrastive_loss_with_temperature.py:81-107
# and, through the all-gather, utils/distributed.py:28-58)
# ------------------------------------------------------------------------------------------------------------------------
def contrastive_loss_backward(a: Array, b: Array, logit_scale: float, a_all: Optional[Array] = None, b_all: Optional[Array] = None,
                              rank: int = 0, mask: Optional[Array] = None, label_smoothing: float = 0.0, reduction: str = "mean",
                              grad_out3=(1.0, 0.0, 0.0), dtype=np.float64):
    """Returns d a, d b (direct terms only), d a_all, d b_all (this rank's logits' gradient w.r.t. EVERY gathered row) and
    d logit_scale, for upstream gradients grad_out3 = d(loss, loss_a, loss_b).  The caller combines them per BackpropType:
    GLOBAL -> d a += sum over ranks of d a_all[own block]; LOCAL -> own rank's d a_all[own block]; NONE -> nothing."""
    a, b = np.asarray(a, dtype=dtype), np.asarray(b, dtype=dtype)
    a_all = a if a_all is None else np.asarray(a_all, dtype=dtype)
    b_all = b if b_all is None else np.asarray(b_all, dtype=dtype)
    B, WB = a.shape[0], a_all.shape[0]
    T = np.exp(dtype(logit_scale))
    keep = np.ones(B, dtype=bool) if mask is None else np.asarray(mask).astype(bool)
    n = keep.sum() if reduction == "mean" else 1.0
    labels = rank * B + np.arange(B)

    def dlogits(logits, w):
        p = softmax_lastdim(logits)
        g = p - label_smoothing / WB
        g[np.arange(B), labels] -= 1.0 - label_smoothing
        g = g * (w / n)
        g[~keep] = 0.0
        return g

    la, lb = T * (a @ b_all.T), T * (b @ a_all.T)
    Ga = dlogits(la, 0.5 * grad_out3[0] + grad_out3[1])
    Gb = dlogits(lb, 0.5 * grad_out3[0] + grad_out3[2])
    return {"grad_a": T * (Ga @ b_all), "grad_b": T * (Gb @ a_all), "grad_a_all": T * (Gb.T @ b), "grad_b_all": T * (Ga.T @ a),
            "grad_logit_scale": (Ga * la).sum() + (Gb * lb).sum()}


# =====================================================================================================================
# FLAVA image codebook: DALL-E dVAE encoder (models/flava/model.py:583-744)
# =====================================================================================================================
def conv2d_same(x: Array, w: Array, b: Array) -> Array:
    """nn.functional.conv2d(x, w, b, padding=(kw-1)//2) for NCHW x, odd square kernels, stride 1 (DalleConv2d.forward :597-598)."""
    kw = w.shape[2]
    pad = (kw - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kw, kw), axis=(2, 3))      # [B, C, H, W, kw, kw] (a view)
    out = np.tensordot(win, w, axes=([1, 4, 5], [1, 2, 3]))                          # [B, H, W, O]
    return out.transpose(0, 3, 1, 2) + b.reshape(1, -1, 1, 1)


def dalle_encoder_block(x: Array, sd, prefix: str, post_gain: float) -> Array:
    """DalleEncoderBlock.forward (:624-625): id_path(x) + post_gain * res_path(x), res_path = (ReLU, conv) x 4 (3x3, 3x3, 3x3, 1x1)."""
    idp = conv2d_same(x, sd[prefix + "id_path.w"], sd[prefix + "id_path.b"]) if prefix + "id_path.w" in sd else x
    h = x
    for i in (1, 2, 3, 4):
        h = conv2d_same(np.maximum(h, 0), sd[prefix + f"res_path.conv_{i}.w"], sd[prefix + f"res_path.conv_{i}.b"])
    return idp + post_gain * h


def dalle_encoder_forward(sd, x: Array, prefix: str = "", dtype=np.float32) -> Array:
    """DalleEncoder.forward (:690-701): stem 7x7, 4 groups of blocks with 2x2 max pools between them, ReLU + 1x1 output conv.
    Returns z_logits [B, vocab, H/8, W/8]; group / block counts are read off the state_dict keys."""
    sd = _cast({k: v for k, v in sd.items() if k.startswith(prefix)}, dtype)
    pre = prefix + "blocks."
    groups = sorted({k[len(pre):].split(".")[0] for k in sd if k.startswith(pre + "group_")})
    n_blocks = {g: len({k[len(pre + g) + 1:].split(".")[0] for k in sd if k.startswith(pre + g + ".block_")}) for g in groups}
    n_layers = sum(n_blocks.values())
    h = conv2d_same(np.asarray(x).astype(dtype), sd[pre + "input.w"], sd[pre + "input.b"])
    for gi, g in enumerate(groups):
        for bi in range(n_blocks[g]):
            h = dalle_encoder_block(h, sd, pre + f"{g}.block_{bi + 1}.", 1.0 / n_layers**2)
        if gi + 1 < len(groups):  # every group but the last ends in nn.MaxPool2d(2) (:677, use_pool=False for group_4)
            B, C, H, W = h.shape
            h = h.reshape(B, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))
    return conv2d_same(np.maximum(h, 0), sd[pre + "output.conv.w"], sd[pre + "output.conv.b"])


def dalle_codebook_indices(sd, images: Array, prefix: str = "encoder.") -> Array:
    """DalleVAEEncoder.get_codebook_indices (:733-735): argmax over the vocabulary axis."""
    return np.argmax(dalle_encoder_forward(sd, images, prefix), axis=1)


# ----------------------------------------------------------------------------------------------------------------------
# Zero-shot / retrieval read-outs: examples/flava/native/utils.py:100-160, examples/flava/coco_zero_shot.py:24-31,84-88
def zero_shot_class_embedding(prompt_emb, dtype=np.float64):
    """utils.py:108-111: normalise the prompt embeddings [T, E], average, normalise."""
    e = np.asarray(prompt_emb, dtype)
    e = e / np.linalg.norm(e, axis=-1, keepdims=True)
    m = e.mean(0)
    return m / np.linalg.norm(m)


def zero_shot_logits(image_features, classifier, scale=100.0, dtype=np.float64):
    """utils.py:141-142: (scale * features / |features|) @ classifier [E, C]."""
    f = np.asarray(image_features, dtype)
    f = f / np.linalg.norm(f, axis=-1, keepdims=True)
    return (scale * f) @ np.asarray(classifier, dtype)


def topk_hits(output, target, topk=(1,)):
    """utils.py:117-123 `_accuracy`: for each k the number of rows whose target index is among the k largest scores
    (stable descending order: on ties the lower index first)."""
    out = np.asarray(output)
    order = np.argsort(-out, axis=1, kind="stable")
    t = np.asarray(target).reshape(-1, 1)
    return [float((order[:, :k] == t).sum()) for k in topk]


def recall_at_k(similarity, k=5):
    """coco_zero_shot.py:24-31 `compute_recall`: targets on the diagonal."""
    n = similarity.shape[0]
    return topk_hits(similarity, np.arange(n), (k,))[0] / n
