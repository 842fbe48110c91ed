"""Parameter containers + HIP execution of the pre-norm transformer stack CLIP builds from
torch.nn.TransformerEncoder(Layer) (reference: models/clip/image_encoder.py:65-77, text_encoder.py:58-66).

The containers reproduce torch's parameter NAMES (so published checkpoints and the reference's state_dicts load
with strict=True: `encoder.layers.N.self_attn.in_proj_weight`, `...out_proj.weight`, `linear1`, `linear2`,
`norm1`, `norm2`) and torch's default INITIALISATION ORDER (so a seeded construction yields bit-identical
initial weights to the reference — including nn.TransformerEncoder's deep-copy of ONE initialised layer into
all N layers).  Their forward is not torch's: `TransformerStack.run` drives the MI355X kernels:

    per layer:  LN -> QKV GEMM(+bias) -> attention -> out-proj GEMM(+bias,+residual)
                LN -> up GEMM(+bias,QuickGELU) -> down GEMM(+bias,+residual)

The residual stream x is kept in fp32 in HBM (accumulated in place by the residual epilogues); everything
that feeds an MFMA is bf16.
"""
from __future__ import annotations

import copy

import torch
from torch import nn

from ... import _torch_ops, ops
from ..._autograd import params_require_grad
from ..._packing import PackedCache

_torch_ops.try_load()

HEAD_DIM = 64  # every model on the path (CLIP B/32, B/16, L/14 and both text towers) has 64-wide heads


class SelfAttentionParams(nn.Module):
    """Parameters of nn.MultiheadAttention (packed in-projection), same names / same default init."""

    def __init__(self, embed_dim: int, num_heads: int) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)  # consumes the RNG exactly like torch's out_proj
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)


class EncoderLayerParams(nn.Module):
    """Parameters of nn.TransformerEncoderLayer(norm_first=True, activation=SiLU())."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, layer_norm_eps: float = 1e-5) -> None:
        super().__init__()
        self.self_attn = SelfAttentionParams(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)


class TransformerStack(nn.Module):
    """`layers` = N deep copies of one initialised layer (what nn.TransformerEncoder does)."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, num_layers: int) -> None:
        super().__init__()
        if d_model % nhead != 0:
            raise ValueError(f"embed_dim {d_model} must be divisible by num_heads {nhead}")
        proto = EncoderLayerParams(d_model, nhead, dim_feedforward)
        self.layers = nn.ModuleList([copy.deepcopy(proto) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.d_model = d_model
        self.nhead = nhead
        self.dim_feedforward = dim_feedforward
        self._packed = PackedCache()

    def forward(self, x: torch.Tensor, B: int, S: int, causal: bool) -> torch.Tensor:
        """Scriptable / traceable form of run(): the same kernels, called through the dispatcher ops of csrc/torch_ops.cpp
        (`torch.ops.mmamd.*`; dtype codes 0 = fp32, 1 = bf16; act code 1 = QuickGELU).  Functional: x is not updated in place."""
        H = self.nhead
        if self.d_model != 64 * H:
            raise RuntimeError("the MI355X attention kernel is built for head dim 64")
        for layer in self.layers:
            hn = torch.ops.mmamd.layernorm(x, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps, 1)
            qkv = torch.ops.mmamd.gemm_bf16(hn, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias, None, 0, 1)
            att = torch.ops.mmamd.attn_fwd(qkv, B, S, H, causal)
            x = torch.ops.mmamd.gemm_bf16(att, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias, x, 0, 0)
            hn = torch.ops.mmamd.layernorm(x, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps, 1)
            up = torch.ops.mmamd.gemm_bf16(hn, layer.linear1.weight, layer.linear1.bias, None, 1, 1)
            x = torch.ops.mmamd.gemm_bf16(up, layer.linear2.weight, layer.linear2.bias, x, 0, 0)
        return x

    @torch.jit.unused
    def run(self, x: torch.Tensor, B: int, S: int, causal: bool, first: int = 0, hn0: torch.Tensor = None) -> torch.Tensor:
        """x: fp32 [B*S, d] residual stream (updated in place and returned); layers [first:] only.  hn0 (optional, bf16 [B*S, d]): norm1 of layer
        `first` already applied to x by the producer of x (the fused ViT stem)."""
        d, H = self.d_model, self.nhead
        hd = d // H
        if hd not in (HEAD_DIM, 96):
            raise ops.MmamdError(f"the MI355X attention kernels are built for 64- and 96-wide heads, got {hd}")
        M = B * S
        dev = x.device
        pk = self._packed.get
        bf, f32 = torch.bfloat16, torch.float32
        hn = hn0 if hn0 is not None else torch.empty((M, d), dtype=bf, device=dev)
        qkv = torch.empty((M, 3 * d), dtype=bf, device=dev)
        att = torch.empty((M, d), dtype=bf, device=dev)
        up = torch.empty((M, self.dim_feedforward), dtype=bf, device=dev)
        for li in range(first, len(self.layers)):
            layer = self.layers[li]
            sa = layer.self_attn
            if li != first or hn0 is None:
                ops.layernorm(x, pk(layer.norm1.weight, f32), pk(layer.norm1.bias, f32), layer.norm1.eps, out=hn)
            ops.gemm_bf16(hn, pk(sa.in_proj_weight, bf), pk(sa.in_proj_bias, f32), out=qkv)
            if hd == HEAD_DIM:
                ops.attention_fwd(qkv, B, S, H, causal, out=att)
            else:  # 96-wide heads: the general kernel on the q / k / v column blocks of the packed projection
                ops.attention_x_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, S, S, H, hd, ops.AttnMask(causal=causal), out=att)
            ops.gemm_bf16(att, pk(sa.out_proj.weight, bf), pk(sa.out_proj.bias, f32), residual=x, out=x)
            ops.layernorm(x, pk(layer.norm2.weight, f32), pk(layer.norm2.bias, f32), layer.norm2.eps, out=hn)
            ops.gemm_bf16(hn, pk(layer.linear1.weight, bf), pk(layer.linear1.bias, f32), act=ops.ACT_QUICKGELU, out=up)
            ops.gemm_bf16(up, pk(layer.linear2.weight, bf), pk(layer.linear2.bias, f32), residual=x, out=x)
        return x


def two_stacks_groupable(sa: TransformerStack, Ma: int, sb: TransformerStack, Mb: int, cus: int = 256) -> bool:
    """Policy of the layer-locked schedule (measured, sustained runs on one box): towers of equal depth, the first one clearly larger, and
    EVERY projection pair of a layer qualifies for one grouped persistent launch (mmamd_gemm_bf16_grouped: K % 128 == 0 and at
    least two 256 x 256 tiles per CU in total).  Otherwise some pairs would run as two launches one after the other on the single stream of
    run_two_stacks and lose the overlap of the two-stream schedule: measured on ViT-B/32 at B = 256 (out-projection and MLP-down: 304 tiles),
    6.67 ms grouped vs 5.59 ms two streams."""
    def tiles(M, N):
        return ((M + 255) // 256) * ((N + 255) // 256)

    # same depth (the towers stay layer-locked to the end: ViT-L/14's upper 12 layers run alone either way, 52.7 ms grouped vs 52.5 two
    # streams) and a clearly dominant first tower (comparable towers overlap their LayerNorm / attention kernels better on two streams:
    # ViT-B/32 at B = 512, every pair grouped, 11.9 vs 10.9 ms) -- profiles/r02_two_tower_ab.txt
    if len(sa.layers) != len(sb.layers) or Ma * sa.d_model < 2 * Mb * sb.d_model:
        return False
    if sa.d_model // sa.nhead != HEAD_DIM or sb.d_model // sb.nhead != HEAD_DIM:
        return False  # (the grouped attention launch is the 64-wide ring kernel)

    for (Na, Ka), (Nb, Kb) in (((3 * sa.d_model, sa.d_model), (3 * sb.d_model, sb.d_model)), ((sa.d_model, sa.d_model), (sb.d_model, sb.d_model)),
                                 ((sa.dim_feedforward, sa.d_model), (sb.dim_feedforward, sb.d_model)),
                                 ((sa.d_model, sa.dim_feedforward), (sb.d_model, sb.dim_feedforward))):
        if Ka % 128 != 0 or Kb % 128 != 0 or tiles(Ma, Na) + tiles(Mb, Nb) < 2 * cus:
            return False
    return True


def run_two_stacks(sa: TransformerStack, xa: torch.Tensor, Ba: int, Sa: int, causal_a: bool, sb: TransformerStack, xb: torch.Tensor, Bb: int,
                   Sb: int, causal_b: bool, hn0_a: torch.Tensor = None):
    """Both towers of a dual encoder, layer-locked on ONE stream: layer i of tower A and layer i of tower B are independent until the loss
    (reference models/clip/model.py:63-75 simply runs one encoder after the other), so each of the four projections of a layer is ONE
    grouped persistent GEMM over both towers' tiles (ops.gemm_bf16_grouped): the short tower's tiles fill the partial last round of the
    long one's instead of competing with it from a second stream; LayerNorm and attention are one grouped launch each too.  Same kernels'
    arithmetic, bit-identical results to TransformerStack.run per tower.  Layers beyond the shorter stack's depth (CLIP L/14: 24 vision, 12
    text) run alone.  xa / xb are updated in place (fp32 residual read-modify-write in the GEMM epilogues)."""
    for st in (sa, sb):
        if st.d_model // st.nhead != HEAD_DIM:
            raise ops.MmamdError(f"the MI355X attention kernel is built for head dim 64, got {st.d_model // st.nhead}")
    bf, f32 = torch.bfloat16, torch.float32
    dev = xa.device
    Ma, Mb = Ba * Sa, Bb * Sb
    pa, pb = sa._packed.get, sb._packed.get

    def bufs(M, st):
        return (torch.empty((M, st.d_model), dtype=bf, device=dev), torch.empty((M, 3 * st.d_model), dtype=bf, device=dev),
                torch.empty((M, st.d_model), dtype=bf, device=dev), torch.empty((M, st.dim_feedforward), dtype=bf, device=dev))

    hna, qkva, atta, upa = bufs(Ma, sa)
    hnb, qkvb, attb, upb = bufs(Mb, sb)
    if hn0_a is not None:  # norm1 of tower A's first layer came with its stem (fused ViT stem): only tower B's is left to do
        hna = hn0_a
    n = min(len(sa.layers), len(sb.layers))

    def ln(norm_a, norm_b):  # both towers' LayerNorms in one launch
        ops.add_layernorm_grouped([(xa, None, pa(norm_a.weight, f32), pa(norm_a.bias, f32), norm_a.eps, hna),
                                   (xb, None, pb(norm_b.weight, f32), pb(norm_b.bias, f32), norm_b.eps, hnb)])

    for li in range(n):
        la, lb = sa.layers[li], sb.layers[li]
        aa, ab = la.self_attn, lb.self_attn
        if li == 0 and hn0_a is not None:
            ops.add_layernorm_grouped([(xb, None, pb(lb.norm1.weight, f32), pb(lb.norm1.bias, f32), lb.norm1.eps, hnb)])
        else:
            ln(la.norm1, lb.norm1)
        ops.gemm_bf16_grouped([(hna, pa(aa.in_proj_weight, bf), pa(aa.in_proj_bias, f32), None, qkva),
                               (hnb, pb(ab.in_proj_weight, bf), pb(ab.in_proj_bias, f32), None, qkvb)])
        ops.attention_fwd_grouped([(qkva, Ba, Sa, sa.nhead, causal_a, atta), (qkvb, Bb, Sb, sb.nhead, causal_b, attb)])
        ops.gemm_bf16_grouped([(atta, pa(aa.out_proj.weight, bf), pa(aa.out_proj.bias, f32), xa, xa),
                               (attb, pb(ab.out_proj.weight, bf), pb(ab.out_proj.bias, f32), xb, xb)], out_dtype=f32)
        ln(la.norm2, lb.norm2)
        ops.gemm_bf16_grouped([(hna, pa(la.linear1.weight, bf), pa(la.linear1.bias, f32), None, upa),
                               (hnb, pb(lb.linear1.weight, bf), pb(lb.linear1.bias, f32), None, upb)], act=ops.ACT_QUICKGELU)
        ops.gemm_bf16_grouped([(upa, pa(la.linear2.weight, bf), pa(la.linear2.bias, f32), xa, xa),
                               (upb, pb(lb.linear2.weight, bf), pb(lb.linear2.bias, f32), xb, xb)], out_dtype=f32)
    if len(sa.layers) > n:
        sa.run(xa, Ba, Sa, causal_a, first=n)
    if len(sb.layers) > n:
        sb.run(xb, Bb, Sb, causal_b, first=n)
    return xa, xb


def forbid_training_forward(module: nn.Module) -> None:
    """Standalone sub-modules (a single encoder layer, a bare tower called outside its model) have no differentiable forward: training
    runs through the stack-level autograd nodes (multimodal_amd/_autograd.py) that the models and encoders dispatch to.  Refuse, loudly,
    to return non-differentiable outputs to a training loop instead of silently detaching them."""
    if module.training and torch.is_grad_enabled() and params_require_grad(module):
        raise NotImplementedError(
            f"{type(module).__name__}: this module has no differentiable forward of its own on the MI355X path (training goes through the "
            "enclosing encoder / model); call .eval() and/or run under torch.no_grad() for inference")
