"""Input side, CPU suite: the transforms oracle pinned to Pillow and to the reference's KATs; the product's
coefficient tables and batch geometry checked against the oracle; the CLIP tokenizer / text transforms against the reference's own
test vectors (tests/transforms/test_clip_transform.py, test_text_transforms.py) and the fixture the reference tokenizer produced
(tests/golden/make_golden_clip_transform.py)."""
import numpy as np
import pytest
import torch

from oracle import transforms_oracle as T
from multimodal_amd import ops
from multimodal_amd.transforms import text_transforms as tt
from multimodal_amd.transforms._resample import axis_tables, center_crop_origin, normalize_lut, resize_output_size
from multimodal_amd.transforms.clip_transform import (CLIP_DEFAULT_MEAN, CLIP_DEFAULT_STD, CLIPBPETokenizer, CLIPBPETransform,
                                                      CLIPImageTransform, CLIPTextTransform, _as_u8_hwc, random_resized_crop_params)

from tests.conftest import GOLDEN
from tests._image_emulator import run as emulate

MERGES = str(GOLDEN / "clip_bpe_merges.txt.gz")

SIZES = [(300, 500), (50, 100), (375, 500), (640, 427), (33, 47), (224, 224), (1000, 800), (1, 5), (7, 3), (224, 1200), (231, 224),
         (1500, 225)]


# ------------------------------------------------------------------------------------------------------------- oracle pins
@pytest.mark.parametrize("hw", SIZES)
def test_oracle_resize_is_pillow_bit_for_bit(hw):
    Image = pytest.importorskip("PIL.Image")
    h, w = hw
    a = np.random.default_rng(h * 7919 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    oh, ow = T.tv_resize_output_size(h, w, 224)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(T.pil_resize_bicubic(a, oh, ow), ref)
    # saturating content (the negative lobes overshoot 0 / 255) and a down-scale by a non-integer factor on both axes
    a2 = (np.random.default_rng(1).integers(0, 2, (h, w, 3)) * 255).astype(np.uint8)
    oh2, ow2 = max(1, h * 3 // 7), max(1, w * 5 // 9)
    ref2 = np.asarray(Image.fromarray(a2).resize((ow2, oh2), Image.BICUBIC))
    assert np.array_equal(T.pil_resize_bicubic(a2, oh2, ow2), ref2)


def test_resize_rule_matches_reference_kat():
    """tests/transforms/test_clip_transform.py:141-149: a 500x300 (WxH) image resizes to PIL size (373, 224)."""
    assert T.tv_resize_output_size(300, 500, 224) == (224, 373)
    assert resize_output_size(300, 500, 224) == (224, 373)
    for h, w, s in [(50, 100, 224), (500, 300, 224), (224, 300, 224), (300, 224, 224), (17, 17, 224), (480, 640, (224, 224)), (480, 640, [96])]:
        assert resize_output_size(h, w, s) == T.tv_resize_output_size(h, w, s)
    assert center_crop_origin(224, 373, 224, 224) == T.center_crop_box(224, 373, 224, 224) == (0, 74)
    assert center_crop_origin(229, 224, 224, 224) == (2, 0)  # 2.5 rounds half to even, like Python's round in torchvision


def test_coefficient_tables_equal_the_oracle():
    rng = np.random.default_rng(0)
    for _ in range(120):
        i, o = int(rng.integers(1, 1500)), int(rng.integers(1, 700))
        kk, bd = T.pil_resample_coeffs(i, o)
        f = int(rng.integers(0, o))
        c = int(rng.integers(1, o - f + 1))
        k2, b2 = axis_tables(i, o, f, c)
        assert np.array_equal(kk[f:f + c], k2) and np.array_equal(bd[f:f + c], b2), (i, o, f, c)
        assert int(np.abs(k2.astype(np.int64)).sum(1).max()) * 255 < 2 ** 31  # the int32 accumulator of the kernels cannot wrap


# --------------------------------------------------------------------------------------------- host geometry through the emulator
def _images(seed=0):
    rng = np.random.default_rng(seed)
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(60, 100), (90, 48), (64, 64), (33, 47), (130, 70)]]
    rgbx = rng.integers(0, 256, (70, 55, 4), dtype=np.uint8)
    return ims, rgbx


def test_eval_geometry_equals_oracle():
    ims, rgbx = _images()
    t = CLIPImageTransform(image_size=32, is_train=False)
    items = [_as_u8_hwc(a) for a in ims + [rgbx]]
    desc, tables, host_off, host_len, tmp_len, max_rows, max_seg, max_coef = t._plan_batch(items)
    assert max_rows == int(desc[:, 5].max()) and host_len >= sum(a.size for a, _ in items)
    assert 0 < max_seg <= max(a.shape[1] * px for a, px in items) and max_coef == 32 * int(desc[:, 8].max())
    f32, patches, u8 = emulate(items, desc, tables, tmp_len, 32, 32, t.lut, patch=16, kpad=768)
    for b, a in enumerate(ims + [rgbx[:, :, :3].copy()]):
        want = T.clip_image_transform_eval(a, 32)
        assert np.array_equal(f32[b], want), b
        assert np.array_equal(patches[b * 4:(b + 1) * 4], T.patchify(want, 16))


def test_eval_geometry_rectangular_size_and_pil_input():
    Image = pytest.importorskip("PIL.Image")
    ims, _ = _images(3)
    t = CLIPImageTransform(image_size=(24, 40), is_train=False)
    pil = [Image.fromarray(ims[0]), Image.fromarray(ims[1][:, :, 0]), Image.fromarray(np.dstack([ims[2], ims[2][:, :, :1]]), "RGBA")]
    items = [_as_u8_hwc(p) for p in pil]
    desc, tables, _, _, tmp_len, _, _, _ = t._plan_batch(items)
    f32, _, _ = emulate(items, desc, tables, tmp_len, 24, 40, t.lut)
    for b, p in enumerate(pil):
        assert np.array_equal(f32[b], T.clip_image_transform_eval(np.asarray(p.convert("RGB")), (24, 40)))


def test_train_geometry_equals_oracle_for_the_same_draws():
    ims, _ = _images(5)
    t = CLIPImageTransform(image_size=32, is_train=True)
    torch.manual_seed(1234)
    boxes = [random_resized_crop_params(a.shape[0], a.shape[1]) for a in ims]
    for (i, j, h, w), a in zip(boxes, ims):
        assert 0 <= i and 0 <= j and 0 < h and 0 < w and i + h <= a.shape[0] and j + w <= a.shape[1]
        assert 0.08 * a.shape[0] * a.shape[1] * 0.7 <= h * w  # rounding slack on the 8 % lower area bound
    torch.manual_seed(1234)
    items = [_as_u8_hwc(a) for a in ims]
    desc, tables, _, _, tmp_len, _, _, _ = t._plan_batch(items)
    f32, _, _ = emulate(items, desc, tables, tmp_len, 32, 32, t.lut)
    for b, (a, (i, j, h, w)) in enumerate(zip(ims, boxes)):
        assert np.array_equal(f32[b], T.resized_crop(a, i, j, h, w, 32)), b


def test_image_transform_fails_loudly_without_a_device():
    if torch.cuda.is_available():
        pytest.skip("HIP device present")
    with pytest.raises(ops.MmamdError):
        CLIPImageTransform(is_train=False)(np.zeros((40, 50, 3), np.uint8))
    with pytest.raises(ops.MmamdError):
        CLIPImageTransform(image_interpolation="bilinear")


# ------------------------------------------------------------------------------------------------------------------- text
TEXT1 = "Taken with my analogue EOS 500N with black & white film."
TEXT1_TOKENS = [49406, 2807, 593, 607, 46031, 17805, 276, 271, 271, 333, 593, 1449, 261, 1579, 1860, 269, 49407]


@pytest.fixture(scope="module")
def clip_text():
    return CLIPTextTransform(text_bpe_merges_path=MERGES)


def test_clip_single_text_kat(clip_text):
    """tests/transforms/test_clip_transform.py:66-87."""
    got = clip_text(TEXT1)
    assert got.dtype == torch.long and got.shape == (77,)
    assert got.tolist() == TEXT1_TOKENS + [0] * (77 - len(TEXT1_TOKENS))


def test_clip_multi_text_kat(clip_text):
    """tests/transforms/test_clip_transform.py:89-131: long text truncated between bos / eos, short texts zero-padded."""
    texts = [TEXT1] * 5 + ["This is a shorter sentence."] + [(TEXT1 + " ") * 20]
    got = clip_text(texts)
    assert got.shape == (7, 77)
    assert got[-1].tolist() == [TEXT1_TOKENS[0]] + (TEXT1_TOKENS[1:-1] * 20)[:75] + [TEXT1_TOKENS[-1]]
    assert int(got[:-1, len(TEXT1_TOKENS):].max()) == 0


def test_tokenizer_equals_reference_fixture(golden, clip_text):
    z = golden("clip_transform.npz")
    tok = CLIPBPETokenizer(MERGES)  # every line of the file, like the generator's tokenizer
    assert tok.vocab_size == int(z["vocab_size"]) and clip_text.tokenizer.bpe.vocab_size == 49408
    texts = []
    for i in range(int(z["n_texts"])):
        t = bytes(z[f"text{i}"]).decode("utf-8")
        texts.append(t)
        ids = tok.encode(t)
        assert ids == z[f"ids{i}"].tolist(), (i, t[:40])
        assert tok.decode(ids) == bytes(z[f"dec{i}"]).decode("utf-8"), i
    assert np.array_equal(clip_text(texts).numpy(), z["batch"])
    assert np.array_equal(clip_text(texts[0]).numpy(), z["single"])
    t32 = CLIPTextTransform(text_max_length=32, text_bpe_merges_path=MERGES, text_pad_token="!")
    assert np.array_equal(t32(texts).numpy(), z["batch_len32_pad"])
    small = CLIPBPETokenizer(MERGES, num_merges=1000)
    assert small.vocab_size == int(z["vocab_size_1000"])
    for i in (0, 4, 10):
        assert small.encode(texts[i]) == z[f"ids1000_{i}"].tolist()
    assert CLIPBPETransform(MERGES)(texts[:3]) == [z[f"ids{i}"].tolist() for i in range(3)]
    assert CLIPBPETransform(MERGES)(texts[4]) == z["ids4"].tolist()


def test_tokenizer_docstring_example(clip_text):
    """clip_transform.py:88-93."""
    tok = clip_text.tokenizer.bpe
    ids = tok.encode("Hello I am using CLIP tokenizer.")
    assert ids == [3306, 328, 687, 1996, 9289, 32634, 23895, 269]
    assert tok.decode(ids) == "hello i am using clip tokenizer . "


def test_pad_token_fills_only_beyond_the_longest_row():
    """The reference pads a ragged batch with 0 up to its longest member (ToTensor) and with the pad token's id from there on."""
    t = CLIPTextTransform(text_max_length=12, text_bpe_merges_path=MERGES, text_pad_token="!")
    got = t(["a photo", "a"])
    pad = t.text_pad_token_id
    assert pad != 0 and got.shape == (2, 12)
    assert got[0, 4:].tolist() == [pad] * 8 and got[1, 3].item() == 0 and got[1, 4:].tolist() == [pad] * 8


def test_staged_pipeline_equals_the_fused_forward(clip_text):
    """`text_transform` (the reference's nn.Sequential of text_transforms stages) and the one-pass forward agree."""
    texts = [TEXT1, "a", "", (TEXT1 + " ") * 20]
    assert torch.equal(clip_text.text_transform(texts), clip_text(texts))
    assert torch.equal(clip_text.text_transform(TEXT1), clip_text(TEXT1))
    t = CLIPTextTransform(text_max_length=12, text_bpe_merges_path=MERGES, text_pad_token="!")
    assert torch.equal(t.text_transform(["a photo", "a"]), t(["a photo", "a"]))


def test_remote_merges_path_fails_loudly():
    with pytest.raises(RuntimeError, match="cannot fetch|no network"):
        CLIPBPETokenizer()


def test_text_transform_kats():
    """tests/transforms/test_text_transforms.py:21-176."""
    assert torch.equal(tt.ToTensor(padding_value=0)([[1, 2], [1, 2, 3]]), torch.tensor([[1, 2, 0], [1, 2, 3]]))
    assert torch.equal(tt.ToTensor(padding_value=0)([1, 2]), torch.tensor([1, 2]))
    assert torch.equal(tt.to_tensor([[1, 2], [1, 2, 3]], 1), torch.tensor([[1, 2, 1], [1, 2, 3]]))
    assert tt.Truncate(2)([[1, 2], [1, 2, 3]]) == [[1, 2], [1, 2]] and tt.Truncate(2)([1, 2, 3]) == [1, 2]
    assert tt.Truncate(2)([["a", "b"], ["a", "b", "c"]]) == [["a", "b"], ["a", "b"]] and tt.truncate(["a", "b", "c"], 2) == ["a", "b"]
    assert tt.AddToken(0, begin=True)([[1, 2], [1, 2, 3]]) == [[0, 1, 2], [0, 1, 2, 3]]
    assert tt.AddToken(0, begin=False)([[1, 2], [1, 2, 3]]) == [[1, 2, 0], [1, 2, 3, 0]] and tt.AddToken(0, begin=False)([1, 2]) == [1, 2, 0]
    assert tt.AddToken("0", begin=True)([["1", "2"], ["1", "2", "3"]]) == [["0", "1", "2"], ["0", "1", "2", "3"]]
    assert tt.add_token(["1", "2"], "0", begin=False) == ["1", "2", "0"]
    with pytest.raises(TypeError):
        tt.add_token([1, 2], "0")
    with pytest.raises(TypeError):
        tt.to_tensor(["a"])
    pad = tt.PadTransform(max_length=7, pad_value=0)
    assert torch.equal(pad(torch.ones(5)), torch.cat([torch.ones(5), torch.zeros(2)]))
    assert torch.equal(pad(torch.ones(8, 5)), torch.cat([torch.ones(8, 5), torch.zeros(8, 2)], dim=-1))
    assert torch.equal(tt.PadTransform(max_length=3, pad_value=0)(torch.ones(8, 5)), torch.ones(8, 5))


# ------------------------------------------------------------------------------------------------------------------ FLAVA
@pytest.mark.parametrize("hw", [(224, 224), (300, 500), (50, 70), (375, 500), (1000, 640)])
def test_oracle_lanczos_is_pillow_bit_for_bit(hw):
    Image = pytest.importorskip("PIL.Image")
    h, w = hw
    a = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    for oh, ow in [(112, 112), (224, 224), (h // 3 + 1, w * 2)]:
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS))
        assert np.array_equal(T.pil_resize(a, oh, ow, "lanczos"), ref), (hw, oh, ow)
    kk, bd = T.pil_resample_coeffs(w, 112, "lanczos")
    k2, b2 = axis_tables(w, 112, 5, 100, "lanczos")
    assert np.array_equal(kk[5:105], k2) and np.array_equal(bd[5:105], b2)


def test_value_tables_equal_the_float_formulas():
    from multimodal_amd.transforms._resample import map_pixels_lut

    u = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    want = T.to_tensor_normalize(u, CLIP_DEFAULT_MEAN, CLIP_DEFAULT_STD).reshape(3, 256)
    assert np.array_equal(normalize_lut(CLIP_DEFAULT_MEAN, CLIP_DEFAULT_STD), want)
    assert np.array_equal(map_pixels_lut(), T.map_pixels(T.to_tensor(u)).reshape(3, 256))
    x = torch.from_numpy(T.to_tensor(u))
    from multimodal_amd.transforms.flava_transform import map_pixels

    assert np.array_equal(map_pixels(x).numpy(), T.map_pixels(x.numpy()))  # torch's fp32 kernels == the numpy restatement
    with pytest.raises(ValueError):
        map_pixels(torch.zeros(2, dtype=torch.float64))


def test_masking_generator_equals_reference_fixture(golden):
    """Draw for draw the reference ImageMaskingGenerator (flava_transform.py:31-106), including the `random` state afterwards."""
    import random

    from multimodal_amd.transforms.flava_transform import ImageMaskingGenerator
    from tests.golden.make_golden_flava_transform import CONFIGS, DRAWS, SEEDS

    z = golden("flava_transform.npz")
    for name, cfg in CONFIGS.items():
        gen = ImageMaskingGenerator(**cfg)
        assert repr(gen) == bytes(z[f"{name}.repr"]).decode()
        assert gen.get_shape() == z[f"{name}.seed0"].shape[1:]
        for seed in SEEDS:
            random.seed(seed)
            got = np.stack([gen() for _ in range(DRAWS)])
            assert got.dtype == np.int64 and np.array_equal(got, z[f"{name}.seed{seed}"]), (name, seed)
            assert random.random() == float(z[f"{name}.seed{seed}.next"])
    assert int(z["default.seed0"][0].sum()) == 75 and int(z["single.seed1234"][0].sum()) == 1


def test_flava_eval_geometry_chains_two_resamplings():
    """Encoder image = exact (S, S) bicubic; codebook image = Lanczos of THAT uint8 image: both plans through the emulator."""
    from multimodal_amd.transforms.flava_transform import FLAVAImageTransform

    ims, _ = _images(11)
    t = FLAVAImageTransform(is_train=False, encoder_input_size=32, codebook_input_size=16)
    items = [_as_u8_hwc(a) for a in ims]
    whole = [((0, 0, a.shape[0], a.shape[1]), (32, 32), (0, 0)) for a in ims]
    desc, tables, _, _, tmp_len, _, _, _ = t.encoder.plan(items, whole)
    enc, _, small = emulate(items, desc, tables, tmp_len, 32, 32, t.image_lut)
    second = [(small[b], 3) for b in range(len(ims))]
    desc, tables, _, _, tmp_len, _, _, _ = t.codebook.plan(second, [((0, 0, 32, 32), (16, 16), (0, 0))] * len(ims))
    cb, _, _ = emulate(second, desc, tables, tmp_len, 16, 16, t.codebook_lut)
    for b, a in enumerate(ims):
        want_enc, want_cb = T.flava_image_transform_eval(a, 32, 16)
        assert np.array_equal(enc[b], want_enc) and np.array_equal(cb[b], want_cb), b


# ------------------------------------------------------------------------------- live cross-checks against the reference checkout
def _reference_transforms():
    from tests.golden import _ref_shim

    if not _ref_shim.reference_available():
        pytest.skip("reference checkout not present (GPU box)")
    _ref_shim.install_transform_stubs()
    import os

    from torchmultimodal.transforms import clip_transform as rc, flava_transform as rf

    return rc, rf, os.path.join(_ref_shim.REFERENCE_ROOT, "tests", "assets", "clip_vocab.bpe")


def test_tokenizer_equals_reference_on_random_strings():
    """Build container only: our integer-id BPE against the reference tokenizer on random unicode / ascii / whitespace mixes."""
    import random

    rc, _, asset = _reference_transforms()
    ref = rc.CLIPBPETokenizer(asset)
    ours = CLIPBPETokenizer(MERGES)
    rnd = random.Random(20260924)
    pools = ["abcdefghijklmnopqrstuvwxyz", "ABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789", " \t\n  ", ".,!?'\"-_/\\()[]{}<>|&%$#@*+=~`^:;",
             "éüñçßøåæœ", "日本語中文한국어", "😀🎉👍🏽🌍", "αβγδεζηθ", "абвгдежз", "​ 　", "'s't're've'm'll'd"]
    for trial in range(400):
        n = rnd.randint(0, 40)
        text = "".join(rnd.choice(rnd.choice(pools)) for _ in range(n))
        if trial % 7 == 0:
            text = text + " <|endoftext|> " + text[::-1] + "<|startoftext|>"
        assert ours.encode(text) == ref.encode(text), repr(text)
        ids = ours.encode(text)
        assert ours.decode(ids) == ref.decode(ids), repr(text)


def test_masking_generator_equals_reference_on_random_configs():
    import random

    _, rf, _ = _reference_transforms()
    from multimodal_amd.transforms.flava_transform import ImageMaskingGenerator

    rnd = random.Random(7)
    for trial in range(60):
        h, w = rnd.randint(4, 20), rnd.randint(4, 20)
        num = rnd.randint(1, h * w - 1)
        lo = rnd.randint(1, max(1, num // 2))
        hi = rnd.choice([None, rnd.randint(lo, num)])
        cfg = dict(input_size=(h, w), num_masking_patches=num, min_num_patches=lo, max_num_patches=hi, min_aspect=rnd.choice([0.3, 0.5, 0.9]))
        a, b = rf.ImageMaskingGenerator(**cfg), ImageMaskingGenerator(**cfg)
        assert repr(a) == repr(b)
        random.seed(trial)
        ma = [a() for _ in range(3)]
        sa = random.random()
        random.seed(trial)
        mb = [b() for _ in range(3)]
        assert random.random() == sa and all(np.array_equal(x, y) for x, y in zip(ma, mb)), cfg


def test_resize_rule_agrees_with_an_independent_restatement():
    """torchvision is not installed here; Hugging Face transformers ships its own restatement of transforms.Resize(int) ("will
    replicate torchvision.transforms.Resize", image_transforms.get_resize_output_image_size(default_to_square=False)).  A second
    opinion on the size rule, not a pin: the reference KAT above is the pin.  Runs in a fresh interpreter: the reference-import shim of
    the tests above leaves stand-in torchvision modules in sys.modules that transformers' availability probe trips over."""
    import json
    import subprocess
    import sys

    rng = np.random.default_rng(5)
    cases = [(int(rng.integers(1, 3000)), int(rng.integers(1, 3000)), int(rng.integers(1, 600))) for _ in range(300)]
    script = ("import json, sys, numpy as np\n"
              "try:\n    from transformers.image_transforms import get_resize_output_image_size as f\n"
              "except Exception as e:\n    print(json.dumps(None)); sys.exit(0)\n"
              "cases = json.loads(sys.stdin.read())\n"
              "print(json.dumps([[int(v) for v in f(np.zeros((h, w, 3), np.uint8), s, default_to_square=False, "
              "input_data_format='channels_last')] for h, w, s in cases]))\n")
    res = subprocess.run([sys.executable, "-c", script], input=json.dumps(cases), capture_output=True, text=True, timeout=300)
    want = json.loads(res.stdout.strip().splitlines()[-1]) if res.returncode == 0 and res.stdout.strip() else None
    if want is None:
        pytest.skip("transformers.image_transforms not importable")
    for (h, w, sz), wv in zip(cases, want):
        assert resize_output_size(h, w, sz) == tuple(wv) == T.tv_resize_output_size(h, w, sz), (h, w, sz)


def test_batch_plan_cache_returns_equal_and_independent_plans():
    """An evaluation loader repeats the same batch geometry: the second plan comes from the cache, equals the first, and the caller's
    in-place edits of its descriptor table (the base addresses added at launch time) do not leak back."""
    ims, _ = _images(21)
    t = CLIPImageTransform(image_size=32, is_train=False)
    items = [_as_u8_hwc(a) for a in ims]
    first = t._plan_batch(items)
    first[0][:, 0] += 12345                      # what run() does to word 0
    second = t._plan_batch(items)
    assert second[0] is not first[0] and int(second[0][0, 0]) == int(first[0][0, 0]) - 12345
    assert np.array_equal(second[1], first[1]) and second[2:] == first[2:]
    other = t._plan_batch(items[:3])              # a different batch is a different plan
    assert other[0].shape == (3, 16)
    for _ in range(12):                           # the cache is bounded
        t._plan_batch([_as_u8_hwc(np.zeros((40 + _, 50, 3), np.uint8))])
    assert len(t.resampler._plans) <= 8
