"""Input side on an MI355X: mmamd_image_resample through CLIPImageTransform / CLIPTransform against the
transforms oracle (pinned to Pillow) -- BIT-EXACT: the resampling is integer work, the normalisation three IEEE fp32 operations.
Covers ragged batches, host (PIL / numpy / RGBX) and device-resident sources with a row stride, up- and down-scaling, the im2col
output for 16- and 14-pixel patches, the training crop, and the patch rows feeding the CLIP image tower."""
import numpy as np
import pytest
import torch

from oracle import transforms_oracle as T
from tests.conftest import GOLDEN, set_rng_seed

pytestmark = pytest.mark.gpu

MERGES = str(GOLDEN / "clip_bpe_merges.txt.gz")


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def _bf16(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).float().numpy()


def _ragged(seed, sizes):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]


SIZES = [(300, 500), (50, 100), (375, 500), (640, 427), (33, 47), (224, 224), (1000, 800), (3, 9), (224, 1200), (231, 224)]


def test_eval_batch_is_bit_exact():
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform

    ims = _ragged(0, SIZES)
    t = CLIPImageTransform(is_train=False)
    got = t(ims)
    assert got.shape == (len(ims), 3, 224, 224) and got.dtype == torch.float32 and got.is_cuda
    got = got.cpu().numpy()
    for b, a in enumerate(ims):
        assert np.array_equal(got[b], T.clip_image_transform_eval(a, 224)), (b, a.shape)
    one = t(ims[2])
    assert one.shape == (3, 224, 224) and np.array_equal(one.cpu().numpy(), got[2])


def test_resized_bytes_equal_pillow():
    Image = pytest.importorskip("PIL.Image")
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform

    ims = _ragged(1, SIZES[:6])
    pil = [Image.fromarray(a) for a in ims]
    u8 = CLIPImageTransform(is_train=False).resized(pil).cpu().numpy()
    for b, p in enumerate(pil):
        oh, ow = T.tv_resize_output_size(p.size[1], p.size[0], 224)
        r = np.asarray(p.resize((ow, oh), Image.BICUBIC))
        top, left = T.center_crop_box(oh, ow, 224, 224)
        assert np.array_equal(u8[b], r[top:top + 224, left:left + 224]), b


def test_sources_host_rgbx_and_device_views():
    """A PIL image, a grey PIL image (convert('RGB')), an RGBX array, a device tensor and a device VIEW with a row stride."""
    Image = pytest.importorskip("PIL.Image")
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform

    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (120, 90, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (80, 140), dtype=np.uint8)
    rgbx = rng.integers(0, 256, (75, 133, 4), dtype=np.uint8)
    big = torch.from_numpy(rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)).cuda()
    view = big[20:170, 40:260]
    batch = [Image.fromarray(a), Image.fromarray(grey), rgbx, big, view, torch.from_numpy(a)]
    want = [a, np.repeat(grey[:, :, None], 3, 2), rgbx[:, :, :3].copy(), big.cpu().numpy(), view.cpu().numpy().copy(), a]
    got = CLIPImageTransform(image_size=96, is_train=False)(batch).cpu().numpy()
    for b, w in enumerate(want):
        assert np.array_equal(got[b], T.clip_image_transform_eval(np.ascontiguousarray(w), 96)), b


@pytest.mark.parametrize("patch,size", [(16, 224), (14, 224), (32, 96)])
def test_patch_rows_equal_im2col_of_the_float_image(patch, size):
    from multimodal_amd import ops
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform

    ims = _ragged(3, SIZES[:5])
    t = CLIPImageTransform(image_size=size, is_train=False)
    k = 3 * patch * patch
    kpad = (k + 63) // 64 * 64
    rows = t.patches(ims, patch, kpad)
    g2 = (size // patch) ** 2
    assert rows.shape == (len(ims) * g2, kpad) and rows.dtype == torch.bfloat16
    got = rows.float().cpu().numpy()
    for b, a in enumerate(ims):
        want = _bf16(T.patchify(T.clip_image_transform_eval(a, size), patch))
        assert np.array_equal(got[b * g2:(b + 1) * g2, :k], want), b
    assert not got[:, k:].any()
    # and they are the rows mmamd_patchify builds from the fp32 image the reference API returns
    assert torch.equal(rows, ops.patchify(t(ims), patch, kpad))


def test_rectangular_size_and_custom_statistics():
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform

    ims = _ragged(4, [(100, 60), (61, 200), (48, 80)])
    mean, std = (0.5, 0.25, 0.125), (0.5, 0.3, 0.7)
    got = CLIPImageTransform(image_size=(48, 80), image_mean=mean, image_std=std, is_train=False)(ims).cpu().numpy()
    for b, a in enumerate(ims):
        assert np.array_equal(got[b], T.clip_image_transform_eval(a, (48, 80), mean, std)), b


def test_training_crop_is_bit_exact_for_the_same_draws():
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform, random_resized_crop_params

    ims = _ragged(5, SIZES[:6])
    set_rng_seed(1234)
    boxes = [random_resized_crop_params(a.shape[0], a.shape[1]) for a in ims]
    set_rng_seed(1234)
    got = CLIPImageTransform(is_train=True)(ims).cpu().numpy()
    for b, (a, (i, j, h, w)) in enumerate(zip(ims, boxes)):
        assert np.array_equal(got[b], T.resized_crop(a, i, j, h, w, 224)), b


def test_headline_batch_checksum_and_tower_entry():
    """B = 256 photographs of 500x375 (the shape of the headline batch): every 16th image against the oracle, plus a
    size-independent property -- a constant image stays constant through both passes (the coefficients of every output sum to
    2^22 exactly or the rounding absorbs the deficit) -- and the patch rows drive CLIPViTEncoder.forward_patches to the same
    embeddings as the fp32 image does through forward()."""
    from multimodal_amd.models.clip import CLIPViTEncoder
    from multimodal_amd.transforms.clip_transform import CLIP_DEFAULT_MEAN, CLIP_DEFAULT_STD, CLIPImageTransform

    rng = np.random.default_rng(6)
    ims = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(16)] * 16
    ims[7] = np.full((375, 500, 3), 200, np.uint8)
    t = CLIPImageTransform(is_train=False)
    x = t(ims)
    assert x.shape == (256, 3, 224, 224)
    xs = x[::16].cpu().numpy()
    for n, b in enumerate(range(0, 256, 16)):
        assert np.array_equal(xs[n], T.clip_image_transform_eval(ims[b], 224)), b
    const = x[7].cpu().numpy()
    for c in range(3):
        v = (np.float32(200) / np.float32(255) - np.float32(CLIP_DEFAULT_MEAN[c])) / np.float32(CLIP_DEFAULT_STD[c])
        assert (const[c] == v).all()
    set_rng_seed(0)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=224, width=128).cuda().eval()
    with torch.no_grad():
        e1 = vit(x[:32])
        e2 = vit.forward_patches(t.patches(ims[:32], 16, 768))
    assert torch.equal(e1, e2)


def test_clip_transform_pairs_images_with_token_ids():
    """tests/transforms/test_clip_transform.py:66-131 on the device path: sizes, token KAT, zero padding."""
    from multimodal_amd.transforms.clip_transform import CLIPTransform

    t1 = [49406, 2807, 593, 607, 46031, 17805, 276, 271, 271, 333, 593, 1449, 261, 1579, 1860, 269, 49407]
    text1 = "Taken with my analogue EOS 500N with black & white film."
    tr = CLIPTransform(text_bpe_merges_path=MERGES, is_train=False)
    im1, im2 = np.full((300, 500, 3), 255, np.uint8), np.full((50, 100, 3), 255, np.uint8)  # ToPILImage()(torch.ones(3, H, W))
    img, txt = tr(image=im1, text=text1)
    assert img.shape == (3, 224, 224) and txt.tolist() == t1 + [0] * (77 - len(t1))
    imgs, txts = tr(image=[im1] * 5 + [im2] * 2, text=[text1] * 5 + ["This is a shorter sentence."] + [(text1 + " ") * 20])
    assert imgs.shape == (7, 3, 224, 224) and txts.shape == (7, 77)
    assert txts[-1].tolist() == [t1[0]] + (t1[1:-1] * 20)[:75] + [t1[-1]]
    assert int(txts[:-1, len(t1):].max()) == 0
    assert np.array_equal(imgs[5].cpu().numpy(), T.clip_image_transform_eval(im2, 224))


def test_untiled_kernels_odd_crop_width_and_very_long_rows():
    """The direct kernels behind the tiled ones: a crop width that is not a multiple of 4 (tmp rows are not whole dwords), and a
    source row longer than the LDS tile (a 22000-pixel panorama squeezed to 8x8 reads 66 KB per row)."""
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform

    ims = _ragged(8, [(45, 70), (100, 31), (30, 30)])
    got = CLIPImageTransform(image_size=30, is_train=False)(ims).cpu().numpy()
    for b, a in enumerate(ims):
        assert np.array_equal(got[b], T.clip_image_transform_eval(a, 30)), b
    wide = _ragged(9, [(16, 22000), (12, 40)])
    got = CLIPImageTransform(image_size=(8, 8), is_train=False)(wide).cpu().numpy()
    for b, a in enumerate(wide):
        assert np.array_equal(got[b], T.clip_image_transform_eval(a, (8, 8))), b


# ------------------------------------------------------------------------------------------------------------------ FLAVA
def test_flava_eval_transform_is_bit_exact():
    """FLAVAImageTransform(is_train=False): encoder image (exact 224x224 bicubic, normalised) and codebook image (Lanczos 112x112 of
    the resized image, map_pixels) against the Pillow-pinned oracle; mask shape / count."""
    from multimodal_amd.transforms.flava_transform import FLAVAImageTransform

    ims = _ragged(10, SIZES[:6])
    t = FLAVAImageTransform(is_train=False)
    out = t(ims)
    assert set(out) == {"image", "image_for_codebook", "image_patches_mask"} and len(out["image"]) == len(ims)
    for b, a in enumerate(ims):
        want_enc, want_cb = T.flava_image_transform_eval(a)
        assert out["image"][b].shape == (3, 224, 224) and out["image_for_codebook"][b].shape == (3, 112, 112)
        assert np.array_equal(out["image"][b].cpu().numpy(), want_enc), b
        assert np.array_equal(out["image_for_codebook"][b].cpu().numpy(), want_cb), b
        m = out["image_patches_mask"][b]
        assert m.shape == (14, 14) and m.dtype == torch.int64 and 16 <= int(m.sum()) <= 75
    one = t(ims[1])
    assert np.array_equal(one["image"].cpu().numpy(), T.flava_image_transform_eval(ims[1])[0])
    bt = t.batch(ims)
    assert bt["image"].shape == (6, 3, 224, 224) and bt["image_for_codebook"].shape == (6, 3, 112, 112)
    assert bt["image_patches_mask"].shape == (6, 14, 14) and bt["image_patches_mask"].is_cuda
    assert torch.equal(bt["image"], torch.stack(out["image"]))


def test_flava_train_transform_reference_kat_and_same_draws():
    """tests/transforms/test_flava_transform.py:19-51 (2x2 white image -> 3x3: (1 - mean) / std, codebook 0.9, one masked patch),
    then random photographs against the oracle for the same crop boxes."""
    from multimodal_amd.transforms._device_resample import random_resized_crop_params
    from multimodal_amd.transforms.flava_transform import FLAVAImageTransform

    set_rng_seed(1234)
    t = FLAVAImageTransform(encoder_input_size=3, codebook_input_size=3, mask_max_patches=1, mask_min_patches=1, mask_num_patches=1)
    out = t(np.full((2, 2, 3), 255, np.uint8))
    want = torch.tensor([1.9303, 2.0749, 2.1459]).view(3, 1, 1).expand(3, 3, 3)
    assert torch.allclose(out["image"].cpu(), want, atol=1e-4, rtol=1e-4)
    assert torch.allclose(out["image_for_codebook"].cpu(), torch.full((3, 3, 3), 0.9))
    assert int(out["image_patches_mask"].sum()) == 1

    ims = _ragged(12, SIZES[:5])
    t = FLAVAImageTransform(is_train=True)
    set_rng_seed(7)
    boxes = [random_resized_crop_params(a.shape[0], a.shape[1], scale=(0.9, 1.0)) for a in ims]
    set_rng_seed(7)
    out = t(ims)
    for b, a in enumerate(ims):
        want_enc, want_cb = T.flava_image_transform_train(a, boxes[b])
        assert np.array_equal(out["image"][b].cpu().numpy(), want_enc), b
        assert np.array_equal(out["image_for_codebook"][b].cpu().numpy(), want_cb), b


def test_empty_batch_returns_empty_tensors():
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform

    t = CLIPImageTransform(is_train=False)
    assert t([]).shape == (0, 3, 224, 224)
    assert t.patches([], 16, 768).shape == (0, 768)


def test_clip_forward_patches_equals_forward_on_the_image_tensor():
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.transforms.clip_transform import CLIPTransform

    set_rng_seed(2)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=224, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=49408, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt).cuda().eval()
    tr = CLIPTransform(text_bpe_merges_path=MERGES, is_train=False)
    ims = _ragged(13, SIZES[:4])
    texts = ["a photo of a cat", "two dogs", "a bicycle in the rain", "x"]
    images, ids = tr(image=ims, text=texts)
    with torch.no_grad():
        a = clip(images, ids.cuda())
        b = clip.forward_patches(tr.image_transform.patches(ims, 16, 768), ids.cuda())
    assert torch.equal(a.embeddings_a, b.embeddings_a) and torch.equal(a.embeddings_b, b.embeddings_b)
