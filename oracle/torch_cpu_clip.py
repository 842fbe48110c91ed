"""TEST / MEASUREMENT INFRASTRUCTURE — never imported by multimodal_amd/ (the product path).

torch-CPU restatement of the reference's CLIP forward + contrastive loss, built from the SAME torch.nn modules the reference
composes — nn.TransformerEncoder(nn.TransformerEncoderLayer(norm_first=True)), nn.MultiheadAttention's fused CPU path,
F.layer_norm, F.normalize, F.cross_entropy — so that, timed on a host's cores, it dispatches to the same ATen kernels as the
reference itself (reference call sites: models/clip/image_encoder.py:65-77,91-113, models/clip/text_encoder.py:58-66,113-134,
models/clip/model.py:65-74, modules/losses/contrastive_loss_with_temperature.py:81-107).  It exists because /root/reference is
absent on the GPU box: bench.py's `cpu_baseline` leg times THIS there ("kind": "port", kind_detail: the restatement on torch.nn CPU modules), next to the figure of
the reference itself measured in the build container (profiles/r02_reference_cpu.json).

Pinned: tests/test_oracle_golden.py::test_torch_cpu_restatement_matches_reference_fixtures compares it with the outputs of the
reference (tests/golden/clip_b32_b8.npz, clip_b16_b4.npz) at fp32 round-off (atol 2e-5).
It consumes a state_dict with the reference's key names (the drop-in modules produce exactly those).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F
from torch import nn


class _QuickGELU(nn.Module):  # activation.py:12-25
    def forward(self, x):
        return torch.sigmoid(1.702 * x) * x


def _stack(width: int, heads: int, ff: int, layers: int, batch_first: bool) -> nn.TransformerEncoder:
    layer = nn.TransformerEncoderLayer(d_model=width, nhead=heads, dim_feedforward=ff, dropout=0.0, activation=_QuickGELU(),
                                       norm_first=True, batch_first=batch_first)
    return nn.TransformerEncoder(layer, num_layers=layers, enable_nested_tensor=False)


def _load_stack(stack: nn.TransformerEncoder, sd: Dict[str, torch.Tensor], prefix: str) -> None:
    own = stack.state_dict()
    got = {k: sd[prefix + k] for k in own}
    stack.load_state_dict(got, strict=True)


class TorchCPUCLIP:
    """Both towers of a reference-layout CLIP state_dict (ViT image tower, causal text tower) on torch's CPU kernels."""

    def __init__(self, sd: Dict[str, torch.Tensor], vision_heads: int, text_heads: int) -> None:
        sd = {k: torch.as_tensor(v).float() for k, v in sd.items()}
        self.sd = sd
        self.conv_w = sd["encoder_a.conv.weight"]
        width = self.conv_w.shape[0]
        self.patch = self.conv_w.shape[-1]
        n_img = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("encoder_a.encoder.layers."))
        n_txt = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("encoder_b.encoder.layers."))
        self.vis = _stack(width, vision_heads, sd["encoder_a.encoder.layers.0.linear1.weight"].shape[0], n_img, True).eval()  # image tower: batch_first (image_encoder.py:65-73)
        _load_stack(self.vis, sd, "encoder_a.encoder.")
        tw = sd["encoder_b.token_embedding.weight"].shape[1]
        self.txt = _stack(tw, text_heads, sd["encoder_b.encoder.layers.0.linear1.weight"].shape[0], n_txt, False).eval()  # text tower: sequence first (text_encoder.py:58-66,120)
        _load_stack(self.txt, sd, "encoder_b.encoder.")
        ctx = sd["encoder_b.positional_embedding"].shape[0]
        self.mask = torch.full((ctx, ctx), float("-inf")).triu(1)

    @torch.no_grad()
    def encode_image(self, x: torch.Tensor) -> torch.Tensor:
        sd = self.sd
        h = F.conv2d(x, self.conv_w, stride=self.patch)  # [B, w, g, g]
        h = h.flatten(2).transpose(1, 2)  # [B, g*g, w]
        cls = sd["encoder_a.cls_token_embedding"].expand(h.shape[0], 1, -1)
        h = torch.cat([cls, h], dim=1) + sd["encoder_a.positional_embedding"]
        w = h.shape[-1]
        h = F.layer_norm(h.float(), (w,), sd["encoder_a.ln_pre.weight"], sd["encoder_a.ln_pre.bias"], 1e-5)
        h = self.vis(h)
        h = F.layer_norm(h[:, 0].float(), (w,), sd["encoder_a.ln_post.weight"], sd["encoder_a.ln_post.bias"], 1e-5)
        return h @ sd["encoder_a.projection"]

    @torch.no_grad()
    def encode_text(self, ids: torch.Tensor) -> torch.Tensor:
        sd = self.sd
        h = F.embedding(ids, sd["encoder_b.token_embedding.weight"]) + sd["encoder_b.positional_embedding"]
        h = self.txt(h.transpose(0, 1), mask=self.mask, is_causal=True).transpose(0, 1)
        w = h.shape[-1]
        h = F.layer_norm(h, (w,), sd["encoder_b.ln_final.weight"], sd["encoder_b.ln_final.bias"], 1e-5)
        eot = h[torch.arange(h.shape[0]), ids.argmax(dim=-1)]
        return F.linear(eot, sd["encoder_b.projection.weight"])

    @torch.no_grad()
    def forward_loss(self, images: torch.Tensor, ids: torch.Tensor, logit_scale: float = math.log(1 / 0.07)
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """(emb_a, emb_b, logits_a, logits_b, loss) of the local (single-process) contrastive loss."""
        a = F.normalize(self.encode_image(images))
        b = F.normalize(self.encode_text(ids))
        t = math.exp(logit_scale)
        la, lb = a @ b.t() * t, b @ a.t() * t
        labels = torch.arange(a.shape[0])
        loss = 0.5 * (F.cross_entropy(la, labels) + F.cross_entropy(lb, labels))
        return a, b, la, lb, loss
