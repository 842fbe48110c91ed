"""Host-side mirror of the dual-encoder / multimodal part of torchmultimodal/models/flava/model.py: FLAVAOutput (:38-59),
flava_multimodal_encoder (:73-98), FLAVAModel (:107-333) and the flava_model factory (:429-523).

Constructor arguments, attribute names, state_dict keys and initialisation order match the reference, so
`flava_model_unified_text_encoder.pt` loads with strict=True.  FLAVAForPreTraining / FLAVAForClassification (MLM, MIM, ITM,
DALL-E codebook) are outside the contrastive path and are not provided.

MI355X-specific: the image and text towers are issued on two HIP streams (as in models/clip/model.py); when
`image_patches_mask` is None the reference encodes the same image twice with identical results (:146-160 vs :175-181) —
here the second pass reuses the first pass's outputs.
"""
from __future__ import annotations

from collections import namedtuple
from functools import partial
from typing import Any, Callable, List, Optional, Tuple, Union

import torch
from torch import nn, Tensor
from typing_extensions import Literal

from ... import ops
from ..._packing import PackedCache, PackedModeMixin
from ...modules.layers.normalizations import Fp32LayerNorm
from ...modules.layers.transformer import TransformerOutput
from ...modules.losses.flava import cls_linear, Pooler
from ...utils.common import load_module_from_url
from ..._autograd import wants_grad
from ...schedule import get_schedule, train_side_stream_now
from ._dalle import DalleConv2d, DalleEncoder, DalleEncoderBlock, DalleVAEEncoder  # noqa: F401  (reference :583-744)
from .image_encoder import flava_image_encoder
from .text_encoder import flava_text_encoder
from .transformer import FLAVATransformerWithoutEmbeddings, TransformerEncoder

EMBEDDING_OPTIONS = Literal["image", "text", "mm"]

FLAVAOutput = namedtuple(
    "FLAVAOutput",
    ["image", "image_masked", "text", "text_masked", "multimodal", "multimodal_masked", "projected_image_embeddings",
     "projected_text_embeddings"],
    defaults=(None, None, None, None, None, None, None, None),
)
FLAVAOutput.__annotations__ = {
    "image": TransformerOutput,
    "image_masked": TransformerOutput,
    "text": TransformerOutput,
    "text_masked": TransformerOutput,
    "multimodal": TransformerOutput,
    "multimodal_masked": TransformerOutput,
}

CKPT_KEY = "flava_full"
FLAVA_FOR_PRETRAINED_MAPPING = {
    # reference model.py:75-78
    "flava_full": "https://download.pytorch.org/models/multimodal/flava/flava_for_pretraining_unified_text_encoder.pt",
}
FLAVA_MODEL_MAPPING = {
    CKPT_KEY: "https://download.pytorch.org/models/multimodal/flava/flava_model_unified_text_encoder.pt",
}

_SIDE_STREAMS = {}


def _side_stream(device: torch.device) -> torch.cuda.Stream:
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


def flava_multimodal_encoder(
    hidden_size: int = 768,
    num_attention_heads: int = 12,
    num_hidden_layers: int = 12,
    dropout: float = 0.0,
    intermediate_size: int = 3072,
    intermediate_activation: Callable[..., nn.Module] = nn.GELU,
    layer_norm_eps: float = 1e-12,
) -> FLAVATransformerWithoutEmbeddings:
    encoder = TransformerEncoder(n_layer=num_hidden_layers, d_model=hidden_size, n_head=num_attention_heads,
                                 dim_feedforward=intermediate_size, activation=intermediate_activation,
                                 layer_norm_eps=layer_norm_eps, dropout=dropout, norm_first=True)
    layernorm = Fp32LayerNorm(hidden_size, eps=layer_norm_eps)
    pooler = Pooler(hidden_size=hidden_size)
    return FLAVATransformerWithoutEmbeddings(encoder=encoder, layernorm=layernorm, pooler=pooler, hidden_size=hidden_size)


class FLAVAModel(PackedModeMixin, nn.Module):
    def __init__(
        self,
        image_encoder: nn.Module,
        text_encoder: nn.Module,
        mm_encoder: nn.Module,
        image_to_mm_projection: nn.Module,
        text_to_mm_projection: nn.Module,
        text_projection: nn.Module,
        image_projection: nn.Module,
        **kwargs: Any,
    ) -> None:
        super().__init__()
        self.image_encoder = image_encoder
        self.text_encoder = text_encoder
        self.mm_encoder = mm_encoder
        self.image_to_mm_projection = image_to_mm_projection
        self.text_to_mm_projection = text_to_mm_projection
        self.text_projection = text_projection
        self.image_projection = image_projection
        self._packed = PackedCache()

    def forward(
        self,
        image: Optional[Tensor] = None,
        text: Optional[Tensor] = None,
        image_patches_mask: Optional[Tensor] = None,
        text_masked: Optional[Tensor] = None,
        required_embedding: Optional[EMBEDDING_OPTIONS] = None,
        skip_unmasked_mm_encoder: bool = True,
    ) -> FLAVAOutput:
        training = wants_grad(self, image)
        if required_embedding is None:
            if image is not None and text is not None:
                required_embedding = "mm"
            elif image is not None:
                required_embedding = "image"
            else:
                required_embedding = "text"

        want_image = image is not None and required_embedding in ("image", "mm")
        want_text = text is not None and required_embedding in ("text", "mm")
        want_text_masked = text_masked is not None and required_embedding in ("text", "mm")

        image_outputs: TransformerOutput = TransformerOutput()
        image_masked_outputs: TransformerOutput = TransformerOutput()
        text_outputs: TransformerOutput = TransformerOutput()
        text_masked_outputs: TransformerOutput = TransformerOutput()
        projected_image_embeddings = projected_text_embeddings = None

        # text tower(s) on the side stream, image tower(s) on the caller's stream
        side = None
        dev = (image if image is not None else text).device
        if want_image and (want_text or want_text_masked) and dev.type == "cuda" and (not training or train_side_stream_now()):
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(main)
        # The unmasked and the masked pass of a tower share its weights and are independent per sample: in inference they run as ONE pass
        # over a 2B batch (the GEMMs see M = 2 B S rows: persistent 256 x 256 kernels at full rounds instead of 1.5 rounds / 128 x 128 tiles),
        # and the outputs are split back into the two TransformerOutputs as views.  Bit-identical to two passes (every kernel's per-row
        # arithmetic is independent of the row's position in the batch: tests/test_gpu_bench_size_parity.py; a patch mask of zeros blends
        # nothing: csrc/rowops.hip::flava_image_embed_kernel).  Encoders with forward hooks keep the two calls.  Training takes the one pass too
        # (schedule.flava_batched_train, r06): torch.cat / the slices are differentiable, every parameter then receives ONE gradient per step instead
        # of two that autograd adds (399 small add launches per step), and the GEMMs run at M = 2 B S -- 80.2 -> 74.7 ms per step, gradients against
        # the reference's autograd unchanged (tests/test_gpu_backward_kernels.py, both arrangements).
        batched = get_schedule().flava_batched_passes and (not training or get_schedule().flava_batched_train)
        if (batched and not training and get_schedule().flava_grouped and want_image and want_text and want_text_masked and image_patches_mask is not None
                and self._groupable(image, text, text_masked, image_patches_mask)):
            return self._forward_grouped(image, text, image_patches_mask, text_masked, required_embedding, skip_unmasked_mm_encoder)
        ctx = torch.cuda.stream(side) if side is not None else _Null()
        with ctx:
            if (batched and want_text and want_text_masked and _plain_call(self.text_encoder) and text.shape == text_masked.shape
                    and text.dtype == text_masked.dtype):
                both = self.encode_text(torch.cat([text, text_masked]))
                text_outputs, text_masked_outputs = _split_batch(both, text.shape[0])
                projected_text_embeddings = cls_linear(text_outputs.last_hidden_state, self.text_projection, self._packed)
            else:
                if want_text:
                    text_outputs, projected_text_embeddings = self.encode_text(text, projection=True)
                if want_text_masked:
                    text_masked_outputs = self.encode_text(text_masked)
        if want_image:
            if image_patches_mask is None:
                image_outputs, projected_image_embeddings = self.encode_image(image, projection=True)
                image_masked_outputs = image_outputs  # identical inputs, deterministic kernels: same values
            elif batched and _plain_call(self.image_encoder) and image_patches_mask.shape[0] == image.shape[0]:
                m = image_patches_mask.reshape(image.shape[0], -1)
                both = self.encode_image(torch.cat([image, image]), image_patches_mask=torch.cat([torch.zeros_like(m), m]))
                image_outputs, image_masked_outputs = _split_batch(both, image.shape[0])
                projected_image_embeddings = cls_linear(image_outputs.last_hidden_state, self.image_projection, self._packed)
            else:
                image_outputs, projected_image_embeddings = self.encode_image(image, projection=True)
                image_masked_outputs = self.encode_image(image, image_patches_mask=image_patches_mask)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
            for o in (text_outputs, text_masked_outputs):
                _record(o, torch.cuda.current_stream(dev))
            if projected_text_embeddings is not None:
                projected_text_embeddings.record_stream(torch.cuda.current_stream(dev))

        multimodal_outputs = TransformerOutput()
        multimodal_masked_outputs = TransformerOutput()
        if required_embedding == "mm":
            # hidden_states[-1], not last_hidden_state: FLAVA feeds the state WITHOUT the final layernorm (:187-188)
            if not skip_unmasked_mm_encoder:
                multimodal_outputs = self.encode_mm(
                    image_outputs.hidden_states[-1] if image_outputs.hidden_states else None,
                    text_outputs.hidden_states[-1] if text_outputs.hidden_states else None,
                )
            multimodal_masked_outputs = self.encode_mm(
                image_masked_outputs.hidden_states[-1] if image_masked_outputs.hidden_states else None,
                text_masked_outputs.hidden_states[-1] if text_masked_outputs.hidden_states else None,
            )

        return FLAVAOutput(
            image=image_outputs,
            image_masked=image_masked_outputs,
            text=text_outputs,
            text_masked=text_masked_outputs,
            multimodal=multimodal_outputs,
            multimodal_masked=multimodal_masked_outputs,
            projected_image_embeddings=projected_image_embeddings,
            projected_text_embeddings=projected_text_embeddings,
        )

    def _groupable(self, image, text, text_masked, image_patches_mask) -> bool:
        from ...modules.encoders.bert_text_encoder import BERTTextEncoder
        from .image_encoder import ImageTransformer
        from .transformer import TransformerEncoder, two_encoders_groupable

        ie, te = self.image_encoder, self.text_encoder
        if type(ie) is not ImageTransformer or type(te) is not BERTTextEncoder or not _plain_call(ie) or not _plain_call(te):
            return False
        if type(ie.encoder) is not TransformerEncoder or type(te.encoder) is not TransformerEncoder or te.layernorm is None or ie.layernorm is None:
            return False
        if text.shape != text_masked.shape or text.dtype != text_masked.dtype or image_patches_mask.shape[0] != image.shape[0] or text.shape[0] != image.shape[0]:
            return False
        return two_encoders_groupable(ie.encoder, te.encoder)

    def _forward_grouped(self, image, text, image_patches_mask, text_masked, required_embedding, skip_unmasked_mm_encoder) -> FLAVAOutput:
        """schedule.flava_grouped: the 2B-batch image pass and the 2B-batch text pass (schedule.flava_batched_passes) LAYER-LOCKED on one stream --
        stems, models/flava/transformer.py::run_two_encoders, heads; then the multimodal encoder as in forward().  Bit-identical outputs."""
        from .transformer import run_two_encoders

        B = image.shape[0]
        ie, te = self.image_encoder, self.text_encoder
        m = image_patches_mask.reshape(B, -1)
        xa = ie.embeddings(torch.cat([image, image]), image_patches_mask=torch.cat([torch.zeros_like(m), m]))
        ids = torch.cat([text, text_masked])
        ids = ids if ids.is_contiguous() else ids.contiguous()
        km = ops.key_mask(ids, pad_id=te.embeddings.pad_token_id) if hasattr(te.embeddings, "pad_token_id") else None
        xb = te.embeddings(input_ids=ids)
        (ya, ha, pa), (yb, hb, pb) = run_two_encoders(ie.encoder, xa, None, te.encoder, xb, km, want_probs=get_schedule().flava_attentions)
        sa = ie.layernorm(ya)
        both_i = TransformerOutput(last_hidden_state=sa, pooler_output=ie.pooler(sa) if ie.pooler is not None else None, hidden_states=ha, attentions=pa)
        sb = te.layernorm(yb)
        both_t = TransformerOutput(last_hidden_state=sb, pooler_output=te.pooler(sb) if te.pooler is not None else None, hidden_states=hb, attentions=pb)
        image_outputs, image_masked_outputs = _split_batch(both_i, B)
        text_outputs, text_masked_outputs = _split_batch(both_t, B)
        projected_image_embeddings = cls_linear(image_outputs.last_hidden_state, self.image_projection, self._packed)
        projected_text_embeddings = cls_linear(text_outputs.last_hidden_state, self.text_projection, self._packed)
        multimodal_outputs = TransformerOutput()
        multimodal_masked_outputs = TransformerOutput()
        if required_embedding == "mm":
            if not skip_unmasked_mm_encoder:
                multimodal_outputs = self.encode_mm(image_outputs.hidden_states[-1], text_outputs.hidden_states[-1])
            multimodal_masked_outputs = self.encode_mm(image_masked_outputs.hidden_states[-1], text_masked_outputs.hidden_states[-1])
        return FLAVAOutput(image=image_outputs, image_masked=image_masked_outputs, text=text_outputs, text_masked=text_masked_outputs,
                           multimodal=multimodal_outputs, multimodal_masked=multimodal_masked_outputs,
                           projected_image_embeddings=projected_image_embeddings, projected_text_embeddings=projected_text_embeddings)

    def encode_image(self, image: Tensor, image_patches_mask: Optional[Tensor] = None, projection: bool = False
                     ) -> Union[Tuple[TransformerOutput, Tensor], Optional[TransformerOutput]]:
        if image_patches_mask is not None:
            encoded_image = self.image_encoder(image, image_patches_mask)
        else:
            encoded_image = self.image_encoder(image)
        if projection:
            projected_embeddings = cls_linear(encoded_image.last_hidden_state, self.image_projection, self._packed)
            return encoded_image, projected_embeddings
        return encoded_image

    def encode_text(self, text: Tensor, text_mask: Optional[Tensor] = None, projection: bool = False
                    ) -> Union[Tuple[TransformerOutput, Tensor], Optional[TransformerOutput]]:
        encoded_text = self.text_encoder(input_ids=text, attention_mask=text_mask, return_attn_weights=get_schedule().flava_attentions,
                                         return_hidden_states=True)
        if projection:
            projected_embeddings = cls_linear(encoded_text.last_hidden_state, self.text_projection, self._packed)
            return encoded_text, projected_embeddings
        return encoded_text

    def _token_linear(self, x: Tensor, lin: nn.Linear, out: Tensor) -> None:
        """out[B, S, dm] (a strided slice of the fused sequence) = lin(x[B, S, d]) — bf16 MFMA GEMM per sample block."""
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and lin.weight.requires_grad)):
            from ._train import TokenLinearFn

            out.copy_(TokenLinearFn.apply(x, lin.weight, lin.bias))  # the placement copy is recorded by autograd (CopySlices)
            return
        B, S, d = x.shape
        h = ops.convert((x if x.is_contiguous() else x.contiguous()).view(B * S, d), torch.bfloat16)
        y = ops.gemm_bf16(h, self._packed.get(lin.weight, torch.bfloat16),
                          self._packed.get(lin.bias, torch.float32) if lin.bias is not None else None,
                          out_dtype=torch.float32)
        out.copy_(y.view(B, S, -1))  # placement into the fused [image | text] sequence: a strided device copy

    def encode_mm(self, image_embedding: Tensor, text_embedding: Tensor) -> TransformerOutput:
        if image_embedding is None or text_embedding is None:
            # nothing passed: e.g. no masked data
            return TransformerOutput()
        B, Si, _ = image_embedding.shape
        St = text_embedding.shape[1]
        dm = self.image_to_mm_projection.out_features
        fused_state = torch.empty((B, Si + St, dm), dtype=torch.float32, device=image_embedding.device)
        self._token_linear(image_embedding, self.image_to_mm_projection, fused_state[:, :Si])
        self._token_linear(text_embedding, self.text_to_mm_projection, fused_state[:, Si:])
        return self.mm_encoder(fused_state)


def _plain_call(enc: nn.Module) -> bool:
    """no forward hooks on the encoder: merging two calls into one would change what a hook observes"""
    return not (enc._forward_hooks or enc._forward_pre_hooks)


def _split_batch(o: TransformerOutput, B: int) -> Tuple[TransformerOutput, TransformerOutput]:
    """the two halves of a 2B-batch TransformerOutput (views, no copies)"""
    def half(t, k):
        return None if t is None else t[k * B:(k + 1) * B]

    def halves(k):
        return TransformerOutput(last_hidden_state=half(o.last_hidden_state, k), pooler_output=half(o.pooler_output, k),
                                 hidden_states=None if o.hidden_states is None else [half(t, k) for t in o.hidden_states],
                                 attentions=None if o.attentions is None else [half(t, k) for t in o.attentions])

    return halves(0), halves(1)


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _record(o: TransformerOutput, stream) -> None:
    for t in (o.last_hidden_state, o.pooler_output, *(o.hidden_states or ()), *(o.attentions or ())):
        if t is not None:
            t.record_stream(stream)


def flava_model(
    image_hidden_size: int = 768, image_num_attention_heads: int = 12, image_num_hidden_layers: int = 12, image_dropout: float = 0.0,
    image_intermediate_size: int = 3072, image_intermediate_activation: Callable[..., nn.Module] = nn.GELU, image_layer_norm_eps: float = 1e-12,
    use_image_masking: bool = True, image_size: int = 224, patch_size: int = 16, num_channels: int = 3,
    text_hidden_size: int = 768, text_num_attention_heads: int = 12, text_num_hidden_layers: int = 12, text_dropout: float = 0.0,
    text_intermediate_size: int = 3072, text_intermediate_activation: Callable[..., nn.Module] = nn.GELU, text_layer_norm_eps: float = 1e-12,
    vocab_size: int = 30522, pad_token_id: int = 0, type_vocab_size: int = 2, max_position_embeddings: int = 512,
    multimodal_hidden_size: int = 768, multimodal_num_attention_heads: int = 12, multimodal_num_hidden_layers: int = 6,
    multimodal_dropout: float = 0.0, multimodal_intermediate_size: int = 3072,
    multimodal_intermediate_activation: Callable[..., nn.Module] = nn.GELU, multimodal_layer_norm_eps: float = 1e-12,
    text_and_image_proj_size: int = 768, pretrained: bool = False, **kwargs: Any,
) -> FLAVAModel:
    """Factory with the reference's keyword names, defaults and positional order (models/flava/model.py:428-500; the names ARE the contract: callers and
    the example configs pass them by keyword).  The three encoder factories share one set of tower keywords (`hidden_size`, `num_attention_heads`,
    `num_hidden_layers`, `dropout`, `intermediate_size`, `intermediate_activation`, `layer_norm_eps`), prefixed per tower here; the rest belong to
    one tower each.  Construction order = the reference's (image, text, multimodal encoder, then the image->mm, text->mm, image, text
    projections): the seeded initial values are checksum-equal tensor for tensor (tests/test_host_api_flava.py)."""
    given = dict(locals())
    shared = ("hidden_size", "num_attention_heads", "num_hidden_layers", "dropout", "intermediate_size", "intermediate_activation", "layer_norm_eps")

    def tower(prefix: str, own: tuple = ()) -> dict:
        return {**{k: given[f"{prefix}_{k}"] for k in shared}, **{k: given[k] for k in own}}

    towers = {
        "image_encoder": flava_image_encoder(**tower("image", ("use_image_masking", "image_size", "patch_size", "num_channels"))),
        "text_encoder": flava_text_encoder(**tower("text", ("vocab_size", "pad_token_id", "type_vocab_size", "max_position_embeddings"))),
        "mm_encoder": flava_multimodal_encoder(**tower("multimodal")),
    }
    # (attribute name, fan-in, fan-out) in the reference's construction order: nn.Linear draws from the global RNG
    heads = {name: nn.Linear(fan_in, fan_out) for name, fan_in, fan_out in (
        ("image_to_mm_projection", image_hidden_size, multimodal_hidden_size), ("text_to_mm_projection", text_hidden_size, multimodal_hidden_size),
        ("image_projection", image_hidden_size, text_and_image_proj_size), ("text_projection", text_hidden_size, text_and_image_proj_size))}
    flava = FLAVAModel(**towers, **heads)
    if pretrained:
        load_module_from_url(flava, FLAVA_MODEL_MAPPING[CKPT_KEY])
    return flava


FLAVAForClassificationOutput = namedtuple("FLAVAForClassificationOutput", ["logits", "loss"])  # reference model.py:62-66
FLAVAForClassificationOutput.__annotations__ = {"logits": Tensor, "loss": Tensor}


class FLAVAForPreTraining(PackedModeMixin, nn.Module):
    """Mirror of models/flava/model.py:301-378: model + image codebook + FLAVAPretrainingLoss.  `image_codebook` is any module that
    maps `image_for_codebook` to integer token ids [B, h, w] (DalleVAEEncoder, ._dalle: the implicit-GEMM pipeline of csrc/conv.hip);
    the label masking
    `image_labels[~image_patches_mask] = -1` (:340-343) is mmamd_mask_labels."""

    def __init__(self, model: FLAVAModel, image_codebook: Optional[nn.Module], loss: nn.Module) -> None:
        super().__init__()
        self.model = model
        self.image_codebook = image_codebook
        self.loss = loss

    def encode_image(self, image: Tensor, cls_index: int = 0) -> Tensor:
        return self.model.encode_image(image, projection=True)[1]

    def encode_text(self, text: Tensor, text_mask: Optional[Tensor] = None, cls_index: int = 0) -> Tensor:
        return self.model.encode_text(text, text_mask, projection=True)[1]

    def forward(
        self,
        image: Optional[Tensor] = None,
        text: Optional[Tensor] = None,
        image_for_codebook: Optional[Tensor] = None,
        image_patches_mask: Optional[Tensor] = None,
        text_masked: Optional[Tensor] = None,
        required_embedding: Optional[EMBEDDING_OPTIONS] = None,
        skip_unmasked_mm_encoder: bool = True,
        itm_labels: Optional[Tensor] = None,
        mlm_labels: Optional[Tensor] = None,
    ):
        image_labels = None
        if image_for_codebook is not None:
            if self.image_codebook is None:
                raise ops.MmamdError("FLAVAForPreTraining: image_for_codebook given but the module was built without an image_codebook")
            with torch.no_grad():
                ids = self.image_codebook(image_for_codebook)
            image_labels = ids.flatten(1).to(torch.int64).contiguous().clone()
            image_patches_mask = image_patches_mask.flatten(1)
            keep = image_patches_mask if image_patches_mask.dtype == torch.uint8 else image_patches_mask.to(torch.uint8)  # flag cast only
            ops.mask_labels_(image_labels, keep.contiguous(), -1)
            image_patches_mask = image_patches_mask.to(torch.bool)
        out = self.model(image=image, text=text, image_patches_mask=image_patches_mask, text_masked=text_masked,
                         required_embedding=required_embedding, skip_unmasked_mm_encoder=skip_unmasked_mm_encoder)
        return self.loss(
            image_sequence=out.image.last_hidden_state,
            text_sequence=out.text.last_hidden_state,
            image_masked_sequence=out.image_masked.last_hidden_state,
            text_masked_sequence=out.text_masked.last_hidden_state,
            multimodal_sequence=(out.multimodal.last_hidden_state if not skip_unmasked_mm_encoder else None),
            multimodal_masked_sequence=out.multimodal_masked.last_hidden_state,
            itm_labels=itm_labels,
            mim_labels=image_labels,
            mlm_labels=mlm_labels,
            projected_image_embeddings=out.projected_image_embeddings,
            projected_text_embeddings=out.projected_text_embeddings,
        )


class FLAVAForClassification(PackedModeMixin, nn.Module):
    """Mirror of models/flava/model.py:380-422.  The classifier (modules.layers.mlp.MLP with nn.ReLU) runs on the CLS row in exact
    fp32 (mmamd_rows_linear_f32); an nn.CrossEntropyLoss `loss` is the cross-entropy kernel, any other callable is called as is."""

    def __init__(self, model: FLAVAModel, classifier: nn.Module, loss: Union[nn.Module, Callable[[Tensor, Tensor], Tensor]], **kwargs: Any) -> None:
        super().__init__()
        self.model = model
        self.classifier = classifier
        self.loss = loss

    def forward(self, image: Optional[Tensor] = None, text: Optional[Tensor] = None, required_embedding: Optional[EMBEDDING_OPTIONS] = None,
                labels: Optional[Tensor] = None, cls_index: int = 0) -> FLAVAForClassificationOutput:
        out = self.model(image=image, text=text, required_embedding=required_embedding, skip_unmasked_mm_encoder=False)
        if required_embedding == "image":
            hidden_state = out.image.last_hidden_state
        elif required_embedding == "text":
            hidden_state = out.text.last_hidden_state
        else:
            hidden_state = out.multimodal.last_hidden_state
        scores = self.classifier(hidden_state[:, cls_index])
        return FLAVAForClassificationOutput(logits=scores, loss=self._loss(scores, labels))

    def _loss(self, scores: Tensor, labels: Tensor) -> Tensor:
        if not isinstance(self.loss, nn.CrossEntropyLoss):
            return self.loss(scores, labels)
        ce = self.loss
        if ce.weight is not None or ce.reduction != "mean" or ce.label_smoothing != 0.0:
            raise ops.MmamdError("FLAVAForClassification on the MI355X path: nn.CrossEntropyLoss with class weights, a reduction other "
                                 "than 'mean' or label smoothing is not implemented")
        if labels.dtype != torch.int64:
            raise ops.MmamdError("FLAVAForClassification: class-index labels (int64) expected")
        lab = labels.contiguous()
        if torch.is_grad_enabled() and scores.requires_grad:
            from ..._autograd import CrossEntropyFn

            return CrossEntropyFn.apply(scores, lab, int(ce.ignore_index))
        sc = scores if scores.is_contiguous() else scores.contiguous()
        return ops.cross_entropy(sc, lab, int(ce.ignore_index))


def flava_model_for_pretraining(codebook_image_size: int = 112, pretrained: bool = False, image_codebook: Optional[nn.Module] = None,
                                **flava_model_kwargs: Any) -> FLAVAForPreTraining:
    """models/flava/model.py:524-544.  The codebook is a randomly initialised DalleVAEEncoder (the reference downloads OpenAI's encoder
    checkpoint in its constructor, which needs the network) unless `image_codebook` supplies one."""
    from ...modules.losses.flava import FLAVAPretrainingLoss

    if pretrained:
        raise RuntimeError("pretrained FLAVA checkpoints need network access (reference downloads them); load a state_dict instead")
    model = flava_model(**flava_model_kwargs)
    hidden_size = flava_model_kwargs.get("multimodal_hidden_size", 768)
    losses = FLAVAPretrainingLoss(hidden_size=hidden_size)
    codebook = image_codebook if image_codebook is not None else DalleVAEEncoder(image_size=codebook_image_size, pretrained=False)
    return FLAVAForPreTraining(model=model, image_codebook=codebook, loss=losses)


def flava_model_for_classification(
    num_classes: int,
    classifier_in_dim: int = 768,
    classifier_hidden_sizes: Union[int, List[int]] = 768,
    classifier_dropout: float = 0.5,
    classifier_activation: Callable[..., nn.Module] = nn.ReLU,
    classifier_normalization: Optional[Callable[..., nn.Module]] = None,
    loss_fn: Optional[Callable[..., Tensor]] = None,
    pretrained: bool = True,
    **flava_model_kwargs: Any,
) -> FLAVAForClassification:
    """models/flava/model.py:547-580 (same defaults; pretrained=True needs the network like the reference's and raises offline)."""
    from ...modules.layers.mlp import MLP

    classifier = MLP(in_dim=classifier_in_dim, out_dim=num_classes, hidden_dims=classifier_hidden_sizes, dropout=classifier_dropout,
                     activation=classifier_activation, normalization=classifier_normalization)
    model = flava_model(**flava_model_kwargs)
    if loss_fn is None:
        loss_fn = nn.CrossEntropyLoss()
    classification_model = FLAVAForClassification(model=model, classifier=classifier, loss=loss_fn)
    if pretrained:
        load_module_from_url(classification_model, FLAVA_FOR_PRETRAINED_MAPPING[CKPT_KEY], strict=False)
    return classification_model
