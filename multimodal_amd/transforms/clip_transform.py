"""Host-side mirror of torchmultimodal/transforms/clip_transform.py (the input side of the hot path).

Text (clip_transform.py:82-299): CLIPBPETokenizer / CLIPBPETransform / CLIPTextTransform.  The reference merges strings of a
byte->unicode alphabet; here the vocabulary is integer ids from the start: a merge is a map (id_a, id_b) -> id_ab whose id doubles
as its rank (id = 512 + rank), a word is a list of ids and one pass picks the smallest mergeable id.  Results are identical
(tests/golden/clip_transform.npz is the reference tokenizer's output); a word cache keeps the per-text cost at a dictionary lookup
per word.  CLIPTextTransform writes the [B, L] int64 batch straight into one (optionally pinned) buffer -- one H2D copy per batch.

Images (clip_transform.py:301-352): CLIPImageTransform = Resize(bicubic, PIL) / CenterCrop (or RandomResizedCrop when training),
ToTensor, Normalize.  The reference does this per image on the host in PIL + torch; here the decoded uint8 pixels of the whole
(ragged) batch are packed into one staging buffer, copied once, and resampled on the GPU by mmamd_image_resample_* with PIL's own
fixed-point coefficients -- the result equals the reference's float tensor bit for bit (oracle/transforms_oracle.py is pinned to
PIL; tests/test_gpu_transforms.py).  `patches()` fuses crop + /255 + normalize + im2col and hands bf16 patch rows to the
patch-embedding GEMM without the fp32 image ever existing in HBM."""
from __future__ import annotations

import gzip
import math
from functools import lru_cache
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import regex
import torch
from torch import nn, Tensor

from . import text_transforms

CLIP_DEFAULT_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_DEFAULT_STD = (0.26862954, 0.26130258, 0.27577711)
CLIP_DEFAULT_VOCAB_BPE_PATH = "http://download.pytorch.org/models/text/clip_merges.bpe"

_WORD_PATTERN = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"
_END = "</w>"


def _byte_alphabet() -> List[str]:
    """alphabet[b] = the printable stand-in of byte b (clip_transform.py:31-54): the printable latin-1 bytes stand for themselves,
    the other 68 map to U+0100 onwards in byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    out, nxt = [], 0
    for b in range(256):
        if b in keep:
            out.append(chr(b))
        else:
            out.append(chr(256 + nxt))
            nxt += 1
    return out


def _vocab_order() -> List[int]:
    """The byte values in the order the reference's vocabulary lists them: the kept bytes ascending, then the remapped ones."""
    alpha = _byte_alphabet()
    kept = [b for b in range(256) if ord(alpha[b]) == b]
    return kept + [b for b in range(256) if ord(alpha[b]) != b]


def _open_text(path: str) -> str:
    if path.startswith(("http://", "https://")):
        # same behaviour as utils.common.load_module_from_url: fetch once into the torch hub cache (the reference goes through
        # iopath's HTTPURLHandler, transforms/clip_transform.py:100-104); offline this raises URLError, loudly
        import os

        import torch

        cache = os.path.join(torch.hub.get_dir(), "checkpoints")
        os.makedirs(cache, exist_ok=True)
        import hashlib

        # keyed on the FULL url (two different urls ending in merges.txt / vocab.txt must not resolve to whichever was fetched first)
        local = os.path.join(cache, hashlib.sha1(path.encode()).hexdigest()[:8] + "_" + os.path.basename(path))
        if not os.path.exists(local):
            try:
                torch.hub.download_url_to_file(path, local, progress=False)
            except Exception as e:
                raise RuntimeError(f"cannot fetch {path} ({e}): download it and pass the local file as bpe_path / text_bpe_merges_path") from e
        path = local
    if path.endswith(".gz"):
        with gzip.open(path, "rt", encoding="utf-8") as f:
            return f.read()
    with open(path, "r", encoding="utf-8") as f:
        return f.read()


class CLIPBPETokenizer:
    """Byte-level BPE tokenizer of CLIP (clip_transform.py:82-199): same constructor, `encode`, `decode`, `vocab_size`."""

    def __init__(self, bpe_path: str = CLIP_DEFAULT_VOCAB_BPE_PATH, bos_token: str = "<|startoftext|>",
                 eos_token: str = "<|endoftext|>", num_merges: Optional[int] = None):
        lines = _open_text(bpe_path).split("\n")[1:]
        num_merges = num_merges or len(lines)
        lines = lines[:num_merges]
        self.bpe_merges = lines
        self.num_merges = num_merges
        alpha = _byte_alphabet()
        order = _vocab_order()
        vocab: List[str] = [alpha[b] for b in order] + [alpha[b] + _END for b in order]
        pairs = [tuple(line.split()) for line in lines]
        vocab.extend("".join(p) for p in pairs)
        vocab.extend([bos_token, eos_token])
        self.bpe_vocab = vocab
        self.encoder: Dict[str, int] = {v: i for i, v in enumerate(vocab)}  # a repeated string keeps its last id, as in the reference
        self.decoder: Dict[int, str] = {i: v for v, i in self.encoder.items()}
        # integer form: first ids of the 256 byte values (plain / word-final), merge table keyed by id pairs
        self._plain = [self.encoder[alpha[b]] for b in range(256)]
        self._final = [self.encoder[alpha[b] + _END] for b in range(256)]
        self._merge: Dict[Tuple[int, int], Tuple[int, int]] = {}
        for rank, p in enumerate(pairs):
            if len(p) != 2 or p[0] not in self.encoder or p[1] not in self.encoder:
                continue  # blank / malformed line: can never match a symbol pair
            # a repeated pair keeps its LAST rank, like dict(zip(merges, ranks)) in the reference
            self._merge[(self.encoder[p[0]], self.encoder[p[1]])] = (rank, self.encoder[p[0] + p[1]])
        self._special = {bos_token.encode("utf-8"): self.encoder[bos_token], eos_token.encode("utf-8"): self.encoder[eos_token]}
        self._cache: Dict[bytes, Tuple[int, ...]] = {}
        self._unalpha = {c: b for b, c in enumerate(alpha)}
        self.pat = regex.compile(_WORD_PATTERN, regex.IGNORECASE)

    @property
    def vocab_size(self) -> int:
        return len(self.encoder)

    def _word_ids(self, word: bytes) -> Tuple[int, ...]:
        """BPE of one pre-token (clip_transform.py:149-187) on integer symbols."""
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        sp = self._special.get(word)
        if sp is not None:
            ids: Tuple[int, ...] = (sp,)
        else:
            sym = [self._plain[b] for b in word[:-1]] + [self._final[word[-1]]]
            merge = self._merge
            while len(sym) > 1:
                best = None
                for a, b in zip(sym, sym[1:]):
                    m = merge.get((a, b))
                    if m is not None and (best is None or m[0] < best[0]):
                        best, first, second = m, a, b
                if best is None:
                    break
                out, i, n = [], 0, len(sym)
                while i < n:
                    if i + 1 < n and sym[i] == first and sym[i + 1] == second:
                        out.append(best[1])
                        i += 2
                    else:
                        out.append(sym[i])
                        i += 1
                sym = out
            ids = tuple(sym)
        self._cache[word] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        for token in self.pat.findall(text.lower().strip()):
            out.extend(self._word_ids(token.encode("utf-8")))
        return out

    def decode(self, tokens: List[int]) -> str:
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self._unalpha[c] for c in text).decode("utf-8", errors="replace").replace(_END, " ")


class CLIPBPETransform(nn.Module):
    """nn.Module wrapper of the tokenizer (clip_transform.py:202-240): str -> List[int], List[str] -> List[List[int]]."""

    def __init__(self, bpe_path: Optional[str] = CLIP_DEFAULT_VOCAB_BPE_PATH, bos_token: Optional[str] = "<|startoftext|>",
                 eos_token: Optional[str] = "<|endoftext|>", num_merges: Optional[int] = None):
        super().__init__()
        self.bpe = CLIPBPETokenizer(bpe_path=bpe_path, bos_token=bos_token, eos_token=eos_token, num_merges=num_merges)

    def forward(self, text: Union[str, List[str]]) -> Union[List[int], List[List[int]]]:
        if isinstance(text, str):
            return self.bpe.encode(text)
        return [self.bpe.encode(t) for t in text]


class CLIPTextTransform(nn.Module):
    """BPE ids, truncated to text_max_length - 2, wrapped in start / end tokens, padded to text_max_length
    (clip_transform.py:243-299).  As in the reference a ragged batch is first padded with 0 to its longest member and only the
    remainder up to text_max_length takes text_pad_token's id.  `device` (extension): where the [B, L] batch is returned; a CUDA
    device receives it as one pinned-buffer copy."""

    def __init__(self, text_max_length: int = 77, text_start_token: str = "<|startoftext|>", text_end_token: str = "<|endoftext|>",
                 text_pad_token: str = None, text_bpe_merges_path: str = CLIP_DEFAULT_VOCAB_BPE_PATH,
                 num_merges: Optional[int] = 48894, device: Optional[Union[str, torch.device]] = None) -> None:
        super().__init__()
        self.tokenizer = CLIPBPETransform(text_bpe_merges_path, text_start_token, text_end_token, num_merges)
        self.text_start_token = self.tokenizer([text_start_token])[0][0]
        self.text_end_token = self.tokenizer([text_end_token])[0][0]
        self.text_pad_token_id = 0 if text_pad_token is None else self.tokenizer([text_pad_token])[0][0]
        self.text_max_length = text_max_length
        self.device = torch.device(device) if device is not None else None

    @property
    def text_transform(self) -> nn.Sequential:
        """The stage-by-stage pipeline the reference keeps under this name (clip_transform.py:281-294): tokenizer, Truncate, AddToken
        (start), AddToken (end), ToTensor, PadTransform.  forward() computes the same ids in one pass over a single buffer."""
        return nn.Sequential(self.tokenizer, text_transforms.Truncate(self.text_max_length - 2),
                             text_transforms.AddToken(self.text_start_token, begin=True),
                             text_transforms.AddToken(self.text_end_token, begin=False), text_transforms.ToTensor(padding_value=0),
                             text_transforms.PadTransform(max_length=self.text_max_length, pad_value=self.text_pad_token_id))

    def forward(self, text: Union[List[str], str]) -> Tensor:
        L = self.text_max_length
        single = isinstance(text, str)
        rows = [self.tokenizer.bpe.encode(t)[: max(L - 2, 0)] for t in ([text] if single else text)]
        longest = max((len(r) + 2 for r in rows), default=0)
        width = max(L, longest)
        to_gpu = self.device is not None and self.device.type == "cuda"
        out = torch.empty((len(rows), width), dtype=torch.long, pin_memory=to_gpu)
        buf = out.numpy()
        buf[:, :longest] = 0
        buf[:, longest:] = self.text_pad_token_id
        for i, r in enumerate(rows):
            buf[i, 0] = self.text_start_token
            buf[i, 1: 1 + len(r)] = r
            buf[i, 1 + len(r)] = self.text_end_token
        if single:
            out = out[0]
        if self.device is not None:
            out = out.to(self.device, non_blocking=to_gpu)
        return out


# ------------------------------------------------------------------------------------------------------------------ images
from .. import ops  # noqa: E402
from ._device_resample import DeviceResampler, as_u8_hwc, random_resized_crop_params  # noqa: E402,F401
from ._resample import center_crop_origin, normalize_lut, resize_output_size  # noqa: E402

_as_u8_hwc = as_u8_hwc


def _interp_name(mode) -> str:
    name = getattr(mode, "value", mode)
    return str(name).lower()


def convert_to_rgb(img):
    return img.convert("RGB")


class CLIPImageTransform(nn.Module):
    """CLIP image transform (clip_transform.py:301-352): RandomResizedCrop (train) or Resize + CenterCrop, RGB conversion, ToTensor,
    Normalize -- computed on the device for the whole batch by mmamd_image_resample.

    forward(image) returns what the reference returns, float32 [3, S, S] for one image / [B, 3, S, S] for a list, on `device`.
    `patches(images, patch_size, kpad)` (extension) returns the bf16 im2col rows [B*G2, kpad] the patch-embedding GEMM consumes;
    `resized(images)` (extension) the uint8 [B, S, S, 3] resized crops.  Only bicubic interpolation exists on this path."""

    def __init__(self, image_size: Union[int, Tuple[int, int]] = 224, image_interpolation="bicubic",
                 image_mean: Tuple[float, float, float] = CLIP_DEFAULT_MEAN, image_std: Tuple[float, float, float] = CLIP_DEFAULT_STD,
                 is_train: bool = True, device: Optional[Union[str, torch.device]] = None) -> None:
        super().__init__()
        if _interp_name(image_interpolation) != "bicubic":
            raise ops.MmamdError(f"interpolation {image_interpolation!r} is not implemented on the MI355X path (bicubic only)")
        self.image_size = image_size
        self.crop_hw: Tuple[int, int] = (image_size, image_size) if isinstance(image_size, int) else (int(image_size[0]), int(image_size[1]))
        self.image_mean = tuple(float(v) for v in image_mean)
        self.image_std = tuple(float(v) for v in image_std)
        if any(v == 0.0 for v in self.image_std):
            raise ValueError("image_std has a zero entry")
        self.is_train = is_train
        self.lut = normalize_lut(self.image_mean, self.image_std)
        self._eval_plans: dict = {}  # (h, w) -> geometry of the Resize + CenterCrop branch
        self.resampler = DeviceResampler(self.crop_hw, "bicubic", device)

    @property
    def device(self) -> torch.device:
        return self.resampler.device

    # -- geometry of one image: source view, resized size, crop window ------------------------------------------------------
    def _plan(self, h: int, w: int):
        ch, cw = self.crop_hw
        if self.is_train:
            i, j, vh, vw = random_resized_crop_params(h, w)
            return (i, j, vh, vw), (ch, cw), (0, 0)
        hit = self._eval_plans.get((h, w))
        if hit is None:
            oh, ow = resize_output_size(h, w, self.image_size)
            if oh < ch or ow < cw:
                raise ops.MmamdError(f"resized image {oh}x{ow} is smaller than the crop {ch}x{cw}: padding crops are not implemented")
            hit = self._eval_plans[(h, w)] = ((0, 0, h, w), (oh, ow), center_crop_origin(oh, ow, ch, cw))
        return hit

    def _plan_batch(self, items):
        return self.resampler.plan(items, [self._plan(int(a.shape[0]), int(a.shape[1])) for a, _ in items])

    def _run(self, images, want_f32: bool, patch: int = 0, kpad: int = 0, want_u8: bool = False):
        items = [as_u8_hwc(im) for im in images]
        geoms = [self._plan(int(a.shape[0]), int(a.shape[1])) for a, _ in items]
        return self.resampler.run(items, geoms, self.lut, want_f32, patch, kpad, want_u8)

    def forward(self, image) -> Tensor:
        if isinstance(image, (list, tuple)):
            return self._run(list(image), True)[0]
        return self._run([image], True)[0][0]

    def patches(self, images: Sequence, patch_size: int, kpad: int = 0) -> Tensor:
        return self._run(list(images), False, patch_size, kpad)[1]

    def resized(self, images: Sequence) -> Tensor:
        return self._run(list(images), False, 0, 0, True)[2]


class CLIPTransform(nn.Module):
    """Image and text transform for CLIP (clip_transform.py:355-416): forward(image, text) -> (image tensor, token ids)."""

    def __init__(self, image_size: Union[int, Tuple[int, int]] = 224, image_interpolation="bicubic",
                 image_mean: Tuple[float, float, float] = CLIP_DEFAULT_MEAN, image_std: Tuple[float, float, float] = CLIP_DEFAULT_STD,
                 text_max_length: int = 77, is_train: bool = True, text_start_token: str = "<|startoftext|>",
                 text_end_token: str = "<|endoftext|>", text_pad_token: str = None,
                 text_bpe_merges_path: str = CLIP_DEFAULT_VOCAB_BPE_PATH, num_merges: Optional[int] = 48894,
                 device: Optional[Union[str, torch.device]] = None) -> None:
        super().__init__()
        self.image_transform = CLIPImageTransform(image_size, image_interpolation, image_mean, image_std, is_train, device=device)
        self.text_transform = CLIPTextTransform(text_max_length, text_start_token, text_end_token, text_pad_token,
                                                text_bpe_merges_path, num_merges)

    def forward(self, image, text: Union[List[str], str]) -> Tuple[Tensor, Tensor]:
        return self.image_transform(image), self.text_transform(text)
