"""Generates tests/golden/flava_transform.npz: block-wise patch masks drawn by the REFERENCE ImageMaskingGenerator
(torchmultimodal/transforms/flava_transform.py:31-106) under fixed `random` seeds, for the default FLAVA configuration
(14x14 window, 75 patches, min 16), the reference test's degenerate one (1 patch; tests/transforms/test_flava_transform.py:20-27)
and a rectangular grid with a maximum block size.  Build container only: torchvision is absent here, so the module is imported
with inert stand-ins for the torchvision names it touches at import time (the mask generator uses none of them).

    python tests/golden/make_golden_flava_transform.py
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim  # noqa: E402

CONFIGS = {
    "default": dict(input_size=14, num_masking_patches=75, max_num_patches=None, min_num_patches=16),
    "single": dict(input_size=14, num_masking_patches=1, max_num_patches=1, min_num_patches=1),
    "rect": dict(input_size=(10, 24), num_masking_patches=90, max_num_patches=30, min_num_patches=6, min_aspect=0.5),
    "dense": dict(input_size=8, num_masking_patches=60, max_num_patches=None, min_num_patches=2),
}
SEEDS = (0, 1, 1234, 99)
DRAWS = 6


def main():
    _ref_shim.install()

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Base:
        def __init__(self, *a, **k):
            pass

    modes = types.SimpleNamespace(BICUBIC="bicubic", LANCZOS="lanczos")
    tr = _mod("torchvision.transforms", InterpolationMode=modes, Resize=_Base, RandomResizedCrop=_Base, Compose=_Base, Lambda=_Base,
              ToTensor=_Base, Normalize=_Base)
    fn = _mod("torchvision.transforms.functional")
    sys.modules["torchvision"].transforms = tr
    tr.functional = fn
    from torchmultimodal.transforms.flava_transform import ImageMaskingGenerator

    out = {}
    for name, cfg in CONFIGS.items():
        gen = ImageMaskingGenerator(**cfg)
        out[f"{name}.repr"] = np.frombuffer(repr(gen).encode(), np.uint8)
        for seed in SEEDS:
            random.seed(seed)
            out[f"{name}.seed{seed}"] = np.stack([gen() for _ in range(DRAWS)])
            out[f"{name}.seed{seed}.next"] = np.float64(random.random())  # the generator state after the draws
    np.savez_compressed(os.path.join(HERE, "flava_transform.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("seed0")})


if __name__ == "__main__":
    main()
