"""Test helper: a numpy walk-through of mmamd_image_resample's two kernels (multimodal_amd/csrc/image.hip) over the descriptor
table CLIPImageTransform._plan_batch builds -- lets the CPU suite check the HOST geometry (views, crop windows, table slices,
row ranges, tmp layout) against the oracle without a GPU.  Not product code and not a fallback: tests only."""
import numpy as np

PREC = 22


def _clip8(acc):
    acc = ((acc + 2 ** 31) % 2 ** 32) - 2 ** 31
    return np.clip(acc >> PREC, 0, 255).astype(np.uint8)


def run(items, desc, tables, tmp_len, crop_h, crop_w, lut, patch=0, kpad=0):
    """items: [(uint8 HWC array, bytes per pixel)], desc word 0 relative to each image's first byte; lut float32 [3, 256].
    Returns (f32 [B,3,ch,cw], patches f32-of-bf16-free [B*G2, kpad] as float32 (unrounded) or None, u8 [B,ch,cw,3])."""
    B = len(items)
    tmp = np.zeros(max(tmp_len, 16), np.uint8)
    out_u8 = np.zeros((B, crop_h, crop_w, 3), np.uint8)
    for b, (a, px) in enumerate(items):
        d = desc[b]
        flat = a.reshape(-1)
        assert d[13] == px
        ks = int(d[8])
        for r in range(int(d[5])):
            for x in range(crop_w):
                kk = tables[d[6] + x * ks: d[6] + (x + 1) * ks].astype(np.int64)
                x0, n = tables[d[7] + 2 * x], tables[d[7] + 2 * x + 1]
                for c in range(3):
                    base = int(d[0]) + (int(d[4]) + r) * int(d[1]) + int(x0) * px + c
                    src = flat[base: base + n * px: px].astype(np.int64)
                    assert src.size == n
                    tmp[d[12] + (r * crop_w + x) * 3 + c] = _clip8((src * kk[:n]).sum() + (1 << (PREC - 1)))
        ks = int(d[11])
        t = tmp[d[12]: d[12] + int(d[5]) * crop_w * 3].reshape(int(d[5]), crop_w, 3).astype(np.int64)
        for y in range(crop_h):
            kk = tables[d[9] + y * ks: d[9] + (y + 1) * ks].astype(np.int64)
            y0, n = int(tables[d[10] + 2 * y]), int(tables[d[10] + 2 * y + 1])
            assert 0 <= y0 and y0 + n <= int(d[5])
            out_u8[b, y] = _clip8((t[y0:y0 + n] * kk[:n, None, None]).sum(0) + (1 << (PREC - 1)))
    f32 = np.stack([lut[c][out_u8[..., c]] for c in range(3)], axis=1).astype(np.float32)  # the kernel's value-table lookup
    patches = None
    if patch:
        gh, gw = crop_h // patch, crop_w // patch
        patches = np.zeros((B * gh * gw, kpad or 3 * patch * patch), np.float32)
        p = f32.reshape(B, 3, gh, patch, gw, patch).transpose(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, 3 * patch * patch)
        patches[:, : 3 * patch * patch] = p
    return f32, patches, out_u8
