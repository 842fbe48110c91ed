"""CLIP two-tower wrapper + factories — host-side mirror of torchmultimodal/models/clip/model.py:19-114.

`CLIP` stays tower-agnostic (any nn.Module encoders whose outputs are [B,E] HIP tensors, cf. the reference's
tests/models/clip/test_clip.py:26-56); the L2 normalisation of model.py:72-73 is the l2_normalize kernel.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
from torch import nn

from ... import _torch_ops, ops
from ..._packing import PackedModeMixin
from ...utils.common import load_module_from_url
from . import _train
from .image_encoder import CLIPViTEncoder
from .text_encoder import CLIPTextEncoder
from ...schedule import get_schedule, train_side_stream_now
from ._transformer import run_two_stacks, two_stacks_groupable


_torch_ops.try_load()
_SIDE_STREAMS = {}

# torch.compile of an INFERENCE forward: the whole pair forward is one dispatcher op whose implementation is the eager forward (grouped two-tower
# launches, fused stem, packed output), so a compiled model runs the same schedule at the same speed as the eager one (the per-op dispatcher path
# of csrc/torch_ops.cpp runs the towers one after the other: 14.7 vs 13.3 ms at B = 256, tools/compiled_vs_eager.py).  The op finds its model
# through a weak registry keyed by an integer the module carries; the parameters ride along as inputs so the graph depends on them.
import itertools
import weakref

from ..._custom_op import define as _define

_PAIR_MODELS = weakref.WeakValueDictionary()
_PAIR_KEYS = itertools.count(1)


def _pair_fwd_impl(features_a, features_b, params, key: int, E: int):
    model = _PAIR_MODELS.get(key)
    if model is None:
        raise ops.MmamdError("clip_pair_fwd: the model this compiled graph was traced from is gone")
    # the op runs the LIVE module: the `params` inputs only make the graph depend on them.  A graph replayed with substituted parameters
    # (torch.func.functional_call, export with swapped weights) would silently compute with the registered module's own: refuse instead.
    own = list(model.parameters())
    if len(own) != len(params) or any(p.data_ptr() != q.data_ptr() for p, q in zip(own, params)):
        raise ops.MmamdError("clip_pair_fwd: the parameters passed to the compiled graph are not the registered model's own (functional / substituted "
                             "parameters are not supported by this op: run the eager forward, or the per-op dispatcher path)")
    with torch.no_grad():
        out = model._forward(model.encoder_a, features_a, features_b)
    a, b = out.embeddings_a, out.embeddings_b
    base = a._base if a._base is not None else None
    if base is not None and base.shape == (a.shape[0], 2 * E) and b._base is base:
        return base  # the packed [B, 2E] block both outputs are views of
    return torch.cat([a, b], 1)


pair_fwd_op = _define("clip_pair_fwd", "(Tensor features_a, Tensor features_b, Tensor[] params, int key, int E) -> Tensor", _pair_fwd_impl,
                      lambda fa, fb, params, key, E: params[0].new_empty((fa.shape[0], 2 * E), dtype=torch.float32))


class CLIPOutput(NamedTuple):
    """L2-normalised embeddings of the two modalities (reference: models/clip/model.py:19-21).

    Layout note (inference forwards of this package): when both towers produce [B, E] outputs of one dtype, `embeddings_a` and
    `embeddings_b` are the two column halves of ONE [B, 2E] buffer — row stride 2E, i.e. NOT contiguous.  That block is exactly the message
    of the contrastive loss's packed all-gather (utils.distributed.gather_packed_features sends it without packing copies) and the loss
    kernels read the halves in place.  Values, shapes and dtypes are the reference's; code that needs a flat view should call
    `.contiguous()` first (`.reshape(-1)` works, `.view(-1)` does not).  The scripted / compiled and the training forwards return ordinary
    contiguous tensors."""

    embeddings_a: torch.Tensor
    embeddings_b: torch.Tensor


CLIP_MODEL_MAPPING = {
    "vit_b16": "https://download.pytorch.org/models/multimodal/clip/clip_vit_b16.pt",
    "vit_b32": "https://download.pytorch.org/models/multimodal/clip/clip_vit_b32.pt",
    "vit_l14": "https://download.pytorch.org/models/multimodal/clip/clip_vit_l14.pt",
}


class CLIP(PackedModeMixin, nn.Module):
    """CLIP is a model for contrastive pretraining between two modalities.

    Args:   encoder_a (nn.Module): Instantiated encoder for modality A (e.g. CLIPViTEncoder).
            encoder_b (nn.Module): Instantiated encoder for modality B (e.g. CLIPTextEncoder).

    Inputs: features_a (Tensor): Tensor containing features of modality A.
            features_b (Tensor): Tensor containing features of modality B.
    """

    def __init__(self, encoder_a: nn.Module, encoder_b: nn.Module):
        super().__init__()
        torch._C._log_api_usage_once(f"torchmultimodal.{self.__class__.__name__}")
        self.encoder_a = encoder_a
        self.encoder_b = encoder_b
        self._register_pair()

    @torch.jit.unused
    def _register_pair(self) -> None:
        self._pair_key = next(_PAIR_KEYS)
        _PAIR_MODELS[self._pair_key] = self

    def __setstate__(self, state):  # (copy.deepcopy / unpickling: the copy is another model -> its own key)
        super().__setstate__(state)
        self._register_pair()

    def forward(self, features_a: torch.Tensor, features_b: torch.Tensor) -> CLIPOutput:
        if torch.jit.is_scripting():  # dispatcher ops (csrc/torch_ops.cpp); one stream, inference only
            a = self.encoder_a(features_a)
            b = self.encoder_b(features_b)
            return CLIPOutput(embeddings_a=torch.ops.mmamd.l2_normalize(a.contiguous(), 1e-12),
                              embeddings_b=torch.ops.mmamd.l2_normalize(b.contiguous(), 1e-12))
        else:
            return self._forward_host(features_a, features_b)

    @torch.jit.unused
    def _forward_host(self, features_a: torch.Tensor, features_b: torch.Tensor) -> CLIPOutput:
        if torch.compiler.is_compiling() and not _train.wants_grad(self, features_a, features_b):
            if (type(self.encoder_a) is CLIPViTEncoder and type(self.encoder_b) is CLIPTextEncoder and self.encoder_a.projection.dtype == torch.float32
                    and self.encoder_b.projection.weight.dtype == torch.float32
                    and self.encoder_a.projection.shape[1] == self.encoder_b.projection.weight.shape[0]):
                E = self.encoder_b.projection.weight.shape[0]
                packed = pair_fwd_op(features_a, features_b, list(self.parameters()), self._pair_key, E)
                return CLIPOutput(embeddings_a=packed[:, :E], embeddings_b=packed[:, E:])
            a = self.encoder_a(features_a)
            b = self.encoder_b(features_b)
            return CLIPOutput(embeddings_a=torch.ops.mmamd.l2_normalize(a.contiguous(), 1e-12),
                              embeddings_b=torch.ops.mmamd.l2_normalize(b.contiguous(), 1e-12))
        return self._forward(self.encoder_a, features_a, features_b)

    @torch.jit.unused
    def forward_patches(self, patches_a: torch.Tensor, features_b: torch.Tensor) -> CLIPOutput:
        """Inference entry for a device-side loader (extension): modality A arrives as the bf16 im2col rows of
        transforms.clip_transform.CLIPImageTransform.patches instead of the fp32 image (CLIPViTEncoder.forward_patches);
        same towers, same two-stream schedule, same result as forward() on the corresponding image tensor."""
        if _train.wants_grad(self, patches_a):
            raise ops.MmamdError("forward_patches is an inference entry: call it under torch.no_grad() / in eval mode")
        if not hasattr(self.encoder_a, "forward_patches"):
            raise ops.MmamdError(f"{type(self.encoder_a).__name__} has no forward_patches entry")
        return self._forward(self.encoder_a.forward_patches, patches_a, features_b)

    @torch.jit.unused
    def _forward(self, tower_a, features_a: torch.Tensor, features_b: torch.Tensor) -> CLIPOutput:
        # The two towers are independent until the normalised features meet in the loss: run tower B on a side HIP
        # stream so its small-grid kernels (77-token sequences: 150-600 workgroups per GEMM) fill the CUs that tower
        # A's kernels leave idle in their last, partial wave of workgroups.  Pure stream plumbing: same kernels,
        # same results (schedule.side_stream = False disables it).
        if _train.wants_grad(self, features_a, features_b):
            # differentiable path (train mode, grad enabled): autograd nodes with HIP forward and backward
            # (not under a process group: DistributedDataParallel stashes its AccumulateGrad hooks on the stream it was constructed on, so a tower
            #  whose backward runs on a side stream pays extra syncs there and cannot be captured in a HIP graph -- ADVICE r03)
            side = self._side_stream(features_a) if train_side_stream_now() else None
            if side is None:
                embeddings_a = _train.L2NormalizeFn.apply(tower_a(features_a))
                embeddings_b = _train.L2NormalizeFn.apply(self.encoder_b(features_b))
            else:  # tower B (forward AND, through autograd's stream bookkeeping, backward) on the side stream
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    embeddings_b = _train.L2NormalizeFn.apply(self.encoder_b(features_b))
                embeddings_a = _train.L2NormalizeFn.apply(tower_a(features_a))
                main.wait_stream(side)
                embeddings_b.record_stream(main)
            return CLIPOutput(embeddings_a=embeddings_a, embeddings_b=embeddings_b)
        if self._grouped_towers(tower_a, features_a, features_b):
            embeddings_a, embeddings_b = self._towers_grouped(tower_a, features_a, features_b)
        else:
            embeddings_a, embeddings_b = self._towers_streams(tower_a, features_a, features_b)
        ea, eb = embeddings_a.detach().contiguous(), embeddings_b.detach().contiguous()
        if ea.dim() == 2 and ea.shape == eb.shape and ea.dtype == eb.dtype:
            # both outputs are views of ONE [B, 2E] block = the message of the loss's packed all-gather
            # (utils.distributed.gather_packed_features recognises the layout and gathers it without packing copies)
            E = ea.shape[1]
            packed = torch.empty((ea.shape[0], 2 * E), dtype=ea.dtype, device=ea.device)
            return CLIPOutput(embeddings_a=ops.l2_normalize(ea, eps=1e-12, out=packed[:, :E]),
                              embeddings_b=ops.l2_normalize(eb, eps=1e-12, out=packed[:, E:]))
        return CLIPOutput(embeddings_a=ops.l2_normalize(ea, eps=1e-12), embeddings_b=ops.l2_normalize(eb, eps=1e-12))

    @torch.jit.unused
    def _towers_grouped(self, tower_a, features_a, features_b):
        """Both towers layer-locked on one stream, each projection ONE grouped persistent GEMM over both towers' tiles
        (_transformer.run_two_stacks)."""
        va, tb = self.encoder_a, self.encoder_b
        from_patches = tower_a is not va
        if from_patches:
            va._check_patches(features_a)
        ids = features_b if (features_b.dtype == torch.int64 and features_b.is_contiguous()) else features_b.to(torch.int64).contiguous()
        if from_patches:
            ha, Ba, Sa = va._stem_patches(features_a)
            hn0 = None
        else:
            ha, Ba, Sa, hn0 = va._stem(features_a, want_hn0=True)
        hb = tb._stem(ids)
        Bb, Sb = ids.shape
        run_two_stacks(va.encoder, ha, Ba, Sa, False, tb.encoder, hb, Bb, Sb, True, hn0_a=hn0)
        return va._head(ha, Ba, Sa), tb._head(hb, Bb, Sb, ids)

    @torch.jit.unused
    def _towers_streams(self, tower_a, features_a, features_b):
        """Tower-agnostic schedule: tower B on a side HIP stream, forked from / joined to the caller's stream (schedule.side_stream = False:
        one after the other).  (Each tower on its own CU partition was measured and rejected: 18.8-23.0 vs 14.4 ms)"""
        side = self._side_stream(features_a)
        if side is None:
            embeddings_a = tower_a(features_a)
            embeddings_b = self.encoder_b(features_b)
        else:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                embeddings_b = self.encoder_b(features_b)
            embeddings_a = tower_a(features_a)
            main.wait_stream(side)
            embeddings_b.record_stream(main)
        return embeddings_a, embeddings_b

    @torch.jit.unused
    def _grouped_towers(self, tower_a, features_a, features_b) -> bool:
        """The grouped two-tower schedule applies to the CLIP pair of this package (ViT + text transformer) in inference, at sizes where every
        projection pair of a layer qualifies for one grouped launch (ViT-B/16 and L/14 at B = 256; not B/32, not small batches); any other
        encoder pair or size keeps the tower-agnostic two-stream path.  schedule.two_tower = "streams" / "grouped" force one of the two."""
        mode = get_schedule().two_tower  # auto | grouped (whenever the encoder pair allows it) | streams
        if mode == "streams":
            return False
        if type(self.encoder_a) is not CLIPViTEncoder or type(self.encoder_b) is not CLIPTextEncoder:
            return False
        if not (isinstance(features_a, torch.Tensor) and isinstance(features_b, torch.Tensor) and features_a.is_cuda and features_b.is_cuda):
            return False
        if features_b.dim() != 2:
            return False
        va, tb = self.encoder_a, self.encoder_b
        if va._forward_hooks or va._forward_pre_hooks or tb._forward_hooks or tb._forward_pre_hooks:
            return False  # hooks observe the encoders' own forward calls
        if features_b.size(1) != tb.context_length:
            return False  # (the encoder's forward raises the reference's error)
        g = va.image_size // va.patch_size
        if tower_a is va:
            if not (features_a.dim() == 4 and features_a.size(1) == 3 and features_a.size(2) == va.image_size and features_a.size(3) == va.image_size):
                return False
            Ba = features_a.size(0)
        elif tower_a == va.forward_patches and features_a.dim() == 2 and features_a.size(0) % (g * g) == 0:
            Ba = features_a.size(0) // (g * g)
        else:
            return False
        if mode == "grouped":
            return True
        # only when every projection pair of a layer becomes ONE persistent launch (else: the two-stream schedule overlaps better)
        return two_stacks_groupable(va.encoder, Ba * (g * g + 1), tb.encoder, features_b.size(0) * features_b.size(1))

    @torch.jit.unused
    def _side_stream(self, ref):
        if not get_schedule().side_stream or not isinstance(ref, torch.Tensor) or not ref.is_cuda:
            return None
        # (during graph capture the fork / join below is captured too: the side stream joins the capture through wait_stream)
        s = _SIDE_STREAMS.get(ref.device)  # process-wide, not a module attribute (modules stay deep-copyable/picklable)
        if s is None:
            s = torch.cuda.Stream(device=ref.device)
            _SIDE_STREAMS[ref.device] = s
        return s


def clip_vit_b16(pretrained: bool = False) -> CLIP:
    vision_encoder = CLIPViTEncoder(image_size=224, patch_size=16, layers=12, heads=12, width=768, embedding_dim=512)
    text_encoder = CLIPTextEncoder(embedding_dim=512)
    clip = CLIP(vision_encoder, text_encoder)
    if pretrained:
        load_module_from_url(clip, CLIP_MODEL_MAPPING["vit_b16"])
    return clip


def clip_vit_b32(pretrained: bool = False) -> CLIP:
    vision_encoder = CLIPViTEncoder(image_size=224, patch_size=32, layers=12, heads=12, width=768, embedding_dim=512)
    text_encoder = CLIPTextEncoder(embedding_dim=512)
    clip = CLIP(vision_encoder, text_encoder)
    if pretrained:
        load_module_from_url(clip, CLIP_MODEL_MAPPING["vit_b32"])
    return clip


def clip_vit_l14(pretrained: bool = False) -> CLIP:
    vision_encoder = CLIPViTEncoder(image_size=224, patch_size=14, layers=24, heads=16, width=1024, embedding_dim=768)
    text_encoder = CLIPTextEncoder(embedding_dim=768, width=768, dim_feedforward=3072, heads=12)
    clip = CLIP(vision_encoder, text_encoder)
    if pretrained:
        load_module_from_url(clip, CLIP_MODEL_MAPPING["vit_l14"])
    return clip
