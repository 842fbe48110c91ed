"""Generates tests/golden/clip_transform.npz and tests/golden/clip_bpe_merges.txt.gz (the input side).

Runs in the build container only.  Imports the REFERENCE tokenizer (torchmultimodal/transforms/clip_transform.py:82-299) through
the shim -- ftfy and torchvision.transforms are absent here and are stubbed: the tokenizer's encode() path touches neither -- and
records its output on a list of texts: the reference's own KATs (tests/transforms/test_clip_transform.py:27-60), contractions,
digits, punctuation runs, accents, CJK, emoji, the special tokens, empty / whitespace-only strings, a text longer than the context.
The merges table is the public CLIP BPE vocabulary the reference's tests use (tests/assets/clip_vocab.bpe, a data file: 48894 merges);
it is stored gzip-compressed so the CPU tests and the GPU box can build the tokenizer without the network.

    python tests/golden/make_golden_clip_transform.py
"""
import gzip
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim  # noqa: E402

TEXTS = [
    "Taken with my analogue EOS 500N with black & white film.",
    "This is a shorter sentence.",
    ("Taken with my analogue EOS 500N with black & white film." + " ") * 20,
    "Hello I am using CLIP tokenizer.",
    "a photo of a cat", "A Photo Of A DOG!!!", "it's what they've said: I'm sure we'll win, he'd say, you're right, don't",
    "IT'S SHE'LL THEY'RE", "1234567890 3.14159 2nd 1,000,000", "...---... !!! ??? #hashtag @user $100 50% a+b=c",
    "café naïve façade über straße São Paulo", "日本語のテキスト と 中文 文本", "emoji 😀🎉 party 👍🏽", "tabs\tand\nnewlines\r\n  and   spaces  ",
    "", "   ", "<|startoftext|> a dog <|endoftext|>", "<|startoftext|><|endoftext|>", "x", "I", "&amp; &lt;b&gt; html &quot;entities&quot;",
    "snake_case camelCase kebab-case path/to/file.txt http://example.com/?q=1&r=2", "Ünïcödé Ǆ ǅ ǆ ß İstanbul ΑΒΓ αβγ Ω", "١٢٣ ௧௨௩ Ⅳ ½ ²",
    "supercalifragilisticexpialidocious pneumonoultramicroscopicsilicovolcanoconiosis", "a" * 300, "mixed123numbers456and789letters",
    "​ zero width   nbsp 　 ideographic", "'s 't 're 've 'm 'll 'd", "'S 'T 'RE", "rock'n'roll o'clock y'all'd've",
]


def main():
    _ref_shim.install()

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    _mod("ftfy", fix_text=lambda s: s)
    tv = sys.modules["torchvision"]
    tr = _mod("torchvision.transforms", InterpolationMode=types.SimpleNamespace(BICUBIC="bicubic"))
    tv.transforms = tr
    import torchmultimodal
    torchmultimodal._PATH_MANAGER.open = lambda p, *a, **k: open(p, *a, **k)
    torchmultimodal._PATH_MANAGER.get_local_path = lambda p: p
    from torchmultimodal.transforms.clip_transform import CLIPBPETokenizer, CLIPTextTransform

    asset = os.path.join(_ref_shim.REFERENCE_ROOT, "tests", "assets", "clip_vocab.bpe")
    with open(asset, "rb") as f:
        raw = f.read()
    with gzip.GzipFile(os.path.join(HERE, "clip_bpe_merges.txt.gz"), "wb", mtime=0) as g:
        g.write(raw)

    out = {}
    tok = CLIPBPETokenizer(asset)
    ids = [np.asarray(tok.encode(t), np.int64) for t in TEXTS]
    out["n_texts"] = np.int64(len(TEXTS))
    for i, (t, a) in enumerate(zip(TEXTS, ids)):
        out[f"text{i}"] = np.frombuffer(t.encode("utf-8"), np.uint8)
        out[f"ids{i}"] = a
        out[f"dec{i}"] = np.frombuffer(tok.decode([int(v) for v in a]).encode("utf-8"), np.uint8)
    out["vocab_size"] = np.int64(tok.vocab_size)
    tt = CLIPTextTransform(text_bpe_merges_path=asset)
    out["batch"] = tt(TEXTS).numpy()
    out["single"] = tt(TEXTS[0]).numpy()
    tt32 = CLIPTextTransform(text_max_length=32, text_bpe_merges_path=asset, text_pad_token="!")
    out["batch_len32_pad"] = tt32(TEXTS).numpy()
    small = CLIPBPETokenizer(asset, num_merges=1000)
    out["vocab_size_1000"] = np.int64(small.vocab_size)
    for i in (0, 4, 10):
        out[f"ids1000_{i}"] = np.asarray(small.encode(TEXTS[i]), np.int64)
    np.savez_compressed(os.path.join(HERE, "clip_transform.npz"), **out)
    print("wrote", len(TEXTS), "texts; batch", out["batch"].shape)


if __name__ == "__main__":
    main()
