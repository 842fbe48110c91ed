"""Host-side mirror of FLAVA's image codebook, the DALL-E dVAE encoder (torchmultimodal/models/flava/model.py: DalleConv2d
:583-598, DalleEncoderBlock :601-625, DalleEncoder :628-701, DalleVAEEncoder :704-744).

Same constructors, module tree and parameter names (`blocks.group_1.block_1.res_path.conv_1.w`, ...) and the same seeded
initialisation, so state_dicts are interchangeable.  forward() does not run the module tree: every convolution is one implicit
GEMM on the matrix cores over a zero-bordered NHWC grid (csrc/conv.hip, mmamd_conv_gemm_bf16) with the bias, the identity path,
`post_gain` (folded into the last conv of a block), the border zeroing and the ReLU in front of the next convolution in its
epilogue; the 7x7 stem is an im2col + one-tap GEMM, nn.MaxPool2d(2) and the channel argmax are row kernels.  bf16 operands,
fp32 accumulation; the residual stream between blocks is bf16.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from functools import partial
from typing import Any, Dict, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from ... import ops
from ..._autograd import params_require_grad

bf, f32 = torch.bfloat16, torch.float32


class DalleConv2d(nn.Module):
    def __init__(self, n_in: int, n_out: int, kw: int) -> None:
        super().__init__()
        w = torch.empty((n_out, n_in, kw, kw), dtype=torch.float32)
        w.normal_(std=1 / math.sqrt(n_in * kw**2))
        b = torch.zeros((n_out,), dtype=torch.float32)
        self.w, self.b = nn.Parameter(w), nn.Parameter(b)
        self.kw = kw
        self._pack: Dict[Any, Tuple[tuple, Tensor, Tensor]] = {}

    def packed(self, gain: float = 1.0, stem_kpad: int = 0) -> Tuple[Tensor, Tensor]:
        """(weight bf16 [n_out, K], bias fp32 [n_out]) scaled by `gain`; K = taps*n_in in (tap, channel) order, or — stem — the
        parameter's own (channel, ky, kx) order zero-padded to stem_kpad.  Cached until the parameters change."""
        if not self.w.is_cuda:
            raise ops.MmamdError(f"parameter lives on {self.w.device}: move the module to a HIP device (.to('cuda')); there is no CPU path")
        key = (gain, stem_kpad)
        sig = (self.w.data_ptr(), self.w._version, self.b.data_ptr(), self.b._version)
        hit = self._pack.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1], hit[2]
        n_out, n_in, kw, _ = self.w.shape
        taps = kw * kw
        w32 = self.w.detach().contiguous()
        if stem_kpad:
            wp = ops.dalle_pack(w32, n_out, n_in, taps, stem_kpad, gain, False, bf)
        else:
            wp = ops.dalle_pack(w32, n_out, n_in, taps, n_in * taps, gain, True, bf)
        bp = ops.dalle_pack(self.b.detach().contiguous(), n_out, 1, 1, 1, gain, False, f32).view(n_out)
        self._pack[key] = (sig, wp, bp)
        return wp, bp

    def forward(self, x: Tensor) -> Tensor:
        raise ops.MmamdError("DalleConv2d runs as part of DalleEncoder on the MI355X path (implicit-GEMM pipeline over a padded NHWC "
                             "grid); it has no standalone NCHW forward")


class DalleEncoderBlock(nn.Module):
    def __init__(self, n_in: int, n_out: int, n_layers: int) -> None:
        super().__init__()
        n_hid = n_out // 4
        self.post_gain = 1 / (n_layers**2)
        self.id_path = DalleConv2d(n_in, n_out, 1) if n_in != n_out else nn.Identity()
        self.res_path = nn.Sequential(OrderedDict([
            ("relu_1", nn.ReLU()),
            ("conv_1", DalleConv2d(n_in, n_hid, 3)),
            ("relu_2", nn.ReLU()),
            ("conv_2", DalleConv2d(n_hid, n_hid, 3)),
            ("relu_3", nn.ReLU()),
            ("conv_3", DalleConv2d(n_hid, n_hid, 3)),
            ("relu_4", nn.ReLU()),
            ("conv_4", DalleConv2d(n_hid, n_out, 1)),
        ]))

    def forward(self, x: Tensor) -> Tensor:
        raise ops.MmamdError("DalleEncoderBlock runs as part of DalleEncoder on the MI355X path")


class _Grid:
    """Activation buffers of one resolution: bf16 [guard + B*GH*GW + guard, C]; .rows(t) is the view that starts at grid position 0.
    The guard rows only have to be addressable: they are read for border positions alone, whose outputs are stored as zeros."""

    def __init__(self, B: int, H: int, W: int, device) -> None:
        self.B, self.H, self.W = B, H, W
        self.gh, self.gw = H + 2, W + 2
        self.M = B * self.gh * self.gw
        self.guard = self.gw + 2
        self.device = device
        self.taps3 = torch.tensor([dy * self.gw + dx for dy in (-1, 0, 1) for dx in (-1, 0, 1)], dtype=torch.int64)
        self.tap1 = torch.zeros(1, dtype=torch.int64)

    def new(self, C: int, dtype=bf) -> Tensor:
        return torch.empty((self.M + 2 * self.guard, C), dtype=dtype, device=self.device)

    def rows(self, t: Tensor) -> Tensor:
        return t[self.guard:self.guard + self.M]


class DalleEncoder(nn.Module):
    def __init__(self, group_count: int = 4, n_hid: int = 256, n_blk_per_group: int = 2, input_channels: int = 3, vocab_size: int = 8192,
                 **kwargs: Any) -> None:
        super().__init__()
        self.input_channels = input_channels
        n_layers = group_count * n_blk_per_group
        output_conv = DalleConv2d(8 * n_hid, vocab_size, 1)
        self.blocks = nn.Sequential(OrderedDict([
            ("input", DalleConv2d(input_channels, 1 * n_hid, 7)),
            ("group_1", self._create_group(n_layers, n_blk_per_group, 1 * n_hid, 1 * n_hid)),
            ("group_2", self._create_group(n_layers, n_blk_per_group, 1 * n_hid, 2 * n_hid)),
            ("group_3", self._create_group(n_layers, n_blk_per_group, 2 * n_hid, 4 * n_hid)),
            ("group_4", self._create_group(n_layers, n_blk_per_group, 4 * n_hid, 8 * n_hid, use_pool=False)),
            ("output", nn.Sequential(OrderedDict([("relu", nn.ReLU()), ("conv", output_conv)]))),
        ]))

    def _create_group(self, n_layers: int, n_blk_per_group: int, n_in: int, n_hid: int, use_pool: bool = True) -> nn.Module:
        make_blk = partial(DalleEncoderBlock, n_layers=n_layers)
        blocks: "OrderedDict[str, nn.Module]" = OrderedDict()
        for i in range(n_blk_per_group):
            blocks[f"block_{i + 1}"] = make_blk(n_in, n_hid) if i == 0 else make_blk(n_hid, n_hid)
        if use_pool:
            blocks["pool"] = nn.MaxPool2d(kernel_size=2)
        return nn.Sequential(blocks)

    # ---- the MI355X pipeline -------------------------------------------------------------------------------------------------
    def _block(self, g: _Grid, blk: DalleEncoderBlock, x: Tensor, xr: Tensor, want_relu: bool) -> Tuple[Tensor, Optional[Tensor]]:
        """x, xr = the block input and its ReLU (bf16 grid buffers) -> (x_new, relu(x_new) or None)."""
        c1, c2, c3, c4 = blk.res_path.conv_1, blk.res_path.conv_2, blk.res_path.conv_3, blk.res_path.conv_4
        n_in, n_hid, n_out = c1.w.shape[1], c1.w.shape[0], c4.w.shape[0]
        h = xr
        for conv, cin in ((c1, n_in), (c2, n_hid), (c3, n_hid)):
            w, b = conv.packed()
            o = g.new(n_hid)
            ops.conv_gemm_bf16(g.rows(h), g.taps3, w, b, g.rows(o), g.M, n_hid, cin, g.gh, g.gw, relu_c=True)  # stores relu(conv(.))
            h = o
        if isinstance(blk.id_path, DalleConv2d):
            wi, bi = blk.id_path.packed()
            idp = g.new(n_out)
            ops.conv_gemm_bf16(g.rows(x), g.tap1, wi, bi, g.rows(idp), g.M, n_out, n_in, g.gh, g.gw)
        else:
            idp = x
        w4, b4 = c4.packed(gain=blk.post_gain)   # id_path(x) + post_gain * res_path(x): the gain rides in the last conv's parameters
        y = g.new(n_out)
        yr = g.new(n_out) if want_relu else None
        ops.conv_gemm_bf16(g.rows(h), g.tap1, w4, b4, g.rows(y), g.M, n_out, n_hid, g.gh, g.gw, residual=g.rows(idp),
                           out_relu=g.rows(yr) if yr is not None else None)
        return y, yr

    def _logits_grid(self, x: Tensor) -> Tuple[Tensor, _Grid]:
        if len(x.shape) != 4:
            raise ValueError(f"input shape {x.shape} is not 4d")
        if x.shape[1] != self.input_channels:
            raise ValueError(f"input has {x.shape[1]} channels but model built for {self.input_channels}")
        if not x.is_cuda:
            raise ops.MmamdError(f"images are on {x.device}: the MI355X path needs HIP device tensors (no CPU fallback)")
        if torch.is_grad_enabled() and params_require_grad(self) and self.training:
            raise ops.MmamdError("DalleEncoder on the MI355X path is inference-only (the codebook only supplies labels): call it under "
                                 "torch.no_grad() or in eval mode")
        groups = [m for n, m in self.blocks.named_children() if n.startswith("group_")]
        n_pool = sum(1 for grp in groups if any(isinstance(m, nn.MaxPool2d) for m in grp))
        B, C, H, W = x.shape
        if H % (1 << n_pool) or W % (1 << n_pool):
            raise ops.MmamdError(f"DalleEncoder: image size {H}x{W} must be divisible by {1 << n_pool}")
        xin = x.detach()
        xin = xin if xin.dtype == f32 else xin.float()
        xin = xin if xin.is_contiguous() else xin.contiguous()
        stem = self.blocks.input
        n_hid = stem.w.shape[0]
        g = _Grid(B, H, W, x.device)
        kpad = (C * stem.kw * stem.kw + 63) // 64 * 64
        cols = torch.empty((g.M, kpad), dtype=bf, device=x.device)
        ops.dalle_stem_im2col(xin, stem.kw, kpad, cols)
        ws, bs = stem.packed(stem_kpad=kpad)
        cur, cur_r = g.new(n_hid), g.new(n_hid)
        ops.conv_gemm_bf16(cols, g.tap1, ws, bs, g.rows(cur), g.M, n_hid, kpad, g.gh, g.gw, out_relu=g.rows(cur_r))
        del cols
        for grp in groups:
            mods = list(grp.children())
            blks = [m for m in mods if isinstance(m, DalleEncoderBlock)]
            pool = any(isinstance(m, nn.MaxPool2d) for m in mods)
            for i, blk in enumerate(blks):
                last = i == len(blks) - 1
                cur, cur_r = self._block(g, blk, cur, cur_r, want_relu=not (last and pool))
            if pool:
                ch = cur.shape[1]
                g2 = _Grid(B, g.H // 2, g.W // 2, x.device)
                nxt, nxt_r = g2.new(ch), g2.new(ch)
                ops.dalle_maxpool2(g.rows(cur), g2.rows(nxt), g2.rows(nxt_r), B, g.H, g.W, ch)
                g, cur, cur_r = g2, nxt, nxt_r
        oc = self.blocks.output.conv
        wo, bo = oc.packed()
        V = oc.w.shape[0]
        logits = torch.empty((g.M, V), dtype=f32, device=x.device)
        ops.conv_gemm_bf16(g.rows(cur_r), g.tap1, wo, bo, logits, g.M, V, oc.w.shape[1], 0, 0)
        return logits, g

    def forward(self, x: Tensor) -> Tensor:
        """z_logits [B, vocab, H/8, W/8] fp32 — a strided view of the NHWC grid the kernels produce (values as the reference's NCHW
        tensor; .contiguous() it if a dense NCHW copy is needed)."""
        logits, g = self._logits_grid(x)
        return logits.view(g.B, g.gh, g.gw, -1)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)

    def codebook_indices(self, x: Tensor) -> Tensor:
        logits, g = self._logits_grid(x)
        return ops.dalle_argmax(logits, g.B, g.H, g.W, logits.shape[1])


class DalleVAEEncoder(nn.Module):
    def __init__(self, image_size: Union[int, Tuple[int, int]] = 112, pretrained: bool = True):
        super().__init__()
        self.image_size = image_size
        self.encoder = DalleEncoder()
        if pretrained:
            self.load_model()

    def load_model(self) -> Any:
        # the reference downloads https://cdn.openai.com/dall-e/encoder.pkl (:714-722); there is no network here
        raise RuntimeError("DalleVAEEncoder(pretrained=True) needs network access to the DALL-E encoder checkpoint; construct with "
                           "pretrained=False and load_state_dict() the weights")

    def get_codebook_indices(self, images: Tensor) -> Tensor:
        return self.encoder.codebook_indices(images)

    def get_codebook_probs(self, images: Tensor) -> Tensor:
        """nn.Softmax(dim=1)(z_logits) (:737-739), returned as the same strided NCHW view as DalleEncoder.forward."""
        logits, g = self.encoder._logits_grid(images)
        ops.row_softmax_(logits)
        return logits.view(g.B, g.gh, g.gw, -1)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)

    def forward(self, img_seq_prob: Tensor) -> Tensor:
        return self.get_codebook_indices(img_seq_prob)
