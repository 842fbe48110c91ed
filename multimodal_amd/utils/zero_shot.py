"""Zero-shot classification and retrieval read-outs on the towers' embeddings -- host-side mirror of the
helpers in the reference's examples: examples/flava/native/utils.py:100-160 (`_zero_shot_classifier`, `_accuracy`,
`run_imagenet_zero_shot`) and examples/flava/coco_zero_shot.py:24-31,78-97 (`compute_recall`, the normalised similarity).

Device work goes through libmmamd (mmamd_group_mean_normalize, mmamd_scale_normalize, mmamd_f32_gemm_strided, mmamd_target_rank):
a top-k hit is "fewer than k scores beat the target's", so neither torch.topk nor a sort is needed -- one pass over each score row."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from .. import ops


def _ids(tokens) -> Tensor:
    """A text transform may return the id tensor itself (CLIPTextTransform) or a mapping with 'input_ids' (the HF tokenizer of
    examples/flava/native/utils.py:105-107)."""
    if isinstance(tokens, Tensor):
        return tokens
    ids = tokens["input_ids"]
    return ids if isinstance(ids, Tensor) else torch.as_tensor(ids, dtype=torch.long)


def _f32(x: Tensor) -> Tensor:
    x = x.detach()
    if x.dtype != torch.float32:
        x = ops.convert(x.contiguous(), torch.float32)
    return x if x.is_contiguous() else x.contiguous()


def class_embedding(prompt_embeddings: Tensor, groups: int = 1) -> Tensor:
    """[groups*T, E] prompt embeddings -> [groups, E]: normalise each, average over the T prompts of a class, normalise
    (utils.py:108-111)."""
    return ops.group_mean_normalize(_f32(prompt_embeddings), groups)


def zero_shot_classifier(encode_text: Callable[[Tensor], Tensor], text_transform: Callable, classnames: Sequence[str],
                         templates: Sequence[Callable[[str], str]], device: Union[str, torch.device] = "cuda") -> Tensor:
    """The [E, C] zero-shot weight matrix (utils.py:100-114): for every class, the mean of its normalised prompt embeddings,
    normalised.  `encode_text(ids)` is e.g. `clip.encode_text` or `lambda t: flava.encode_text(t, projection=True)[1]`.
    The result is a transposed view of a [C, E] buffer (the GEMM reads it with strides; nothing is copied)."""
    rows = []
    for name in classnames:
        ids = _ids(text_transform([template(name) for template in templates])).to(device)
        rows.append(class_embedding(encode_text(ids)))
    return torch.cat(rows, dim=0).t()


def zero_shot_logits(image_features: Tensor, classifier: Tensor, scale: float = 100.0) -> Tensor:
    """(scale * normalised image features) @ classifier, fp32 [N, C] (utils.py:141-142).  `classifier` is [E, C]: either the
    transposed view zero_shot_classifier returns or any contiguous fp32 tensor."""
    f = ops.scale_normalize(_f32(image_features), scale)
    N, E = f.shape
    if classifier.dim() != 2 or classifier.shape[0] != E:
        raise ops.MmamdError(f"classifier must be [{E}, C], got {tuple(classifier.shape)}")
    w = classifier.detach()
    if w.dtype != torch.float32:
        w = ops.convert(w.contiguous(), torch.float32)
    C = w.shape[1]
    if w.t().is_contiguous():      # [C, E] storage: class n, feature k at n*E + k
        base, syn, syk = w.t(), E, 1
    else:
        base = w if w.is_contiguous() else w.contiguous()
        syn, syk = 1, C
    return ops.f32_gemm_strided(f, E, 1, base, syn, syk, N, C, E)


def accuracy(output: Tensor, target: Tensor, topk: Tuple[int, ...] = (1,)) -> List[float]:
    """Number of rows whose target is among the top-k scores, for each k (utils.py:117-123 returns these sums, not rates)."""
    t = target.detach().reshape(-1)
    t = (t if t.dtype == torch.int64 else t.to(torch.int64)).contiguous()
    rank = ops.target_rank(_f32(output), t).cpu()
    return [float((rank < k).sum().item()) for k in topk]


def compute_recall(similarity_scores: Tensor, k: int = 5) -> Tensor:
    """Recall@k of a square similarity matrix whose matches sit on the diagonal (coco_zero_shot.py:24-31): a 0-dim float tensor."""
    n = similarity_scores.size(0)
    rank = ops.target_rank(_f32(similarity_scores), None).cpu()
    return (rank < k).sum() / n


def retrieval_similarity(image_embeds: Tensor, text_embeds: Tensor) -> Tensor:
    """normalize(image_embeds) @ normalize(text_embeds).T in fp32 (coco_zero_shot.py:84-88)."""
    a = ops.scale_normalize(_f32(image_embeds), 1.0)
    b = ops.scale_normalize(_f32(text_embeds), 1.0)
    if a.shape[1] != b.shape[1]:
        raise ops.MmamdError(f"embedding widths differ: {a.shape[1]} vs {b.shape[1]}")
    return ops.f32_gemm_strided(a, a.shape[1], 1, b, b.shape[1], 1, a.shape[0], b.shape[0], a.shape[1])


def run_zero_shot(encode_image: Callable[[Tensor], Tensor], batches, classifier: Tensor, topk: Tuple[int, ...] = (1, 5),
                  max_batches: Optional[int] = None) -> dict:
    """The evaluation loop of run_imagenet_zero_shot (utils.py:126-160) over an iterable of {"image", "label"} batches:
    {"top{k}": rate}.  (The reference stops after 6 batches -- pass max_batches=6 for that.)"""
    hits = [0.0 for _ in topk]
    n = 0
    for i, sample in enumerate(batches):
        logits = zero_shot_logits(encode_image(sample["image"]), classifier)
        for j, h in enumerate(accuracy(logits, sample["label"].to(logits.device), topk)):
            hits[j] += h
        n += logits.shape[0]
        if max_batches is not None and i + 1 >= max_batches:
            break
    return {f"top{k}": (h / n if n else float("nan")) for k, h in zip(topk, hits)}
