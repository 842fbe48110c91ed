"""Import shim that lets the REFERENCE (read-only checkout at /root/reference) be imported in the build
container, where torchvision and iopath are absent.  Test-harness code only: it stubs the three
third-party modules the reference imports at module scope but never calls on the CLIP / contrastive-loss path.
Used by tests/golden/make_golden.py (fixture generation) and by the optional reference cross-checks in tests/
(skipped when /root/reference does not exist, e.g. on the GPU box)."""
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("MMAMD_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "torchmultimodal"))


def install() -> None:
    if "torchmultimodal" in sys.modules:
        return

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class PathManager:  # iopath.common.file_io API touched at import time by utils/file_io.py
        def register_handler(self, h):
            pass

        def get_local_path(self, url):
            raise RuntimeError("offline")

    class HTTPURLHandler:
        pass

    class _Absent(torch.nn.Module):  # only reached by the clip_rn*_tv factories
        def __init__(self, *a, **k):
            raise NotImplementedError

    class StochasticDepth(torch.nn.Module):  # identity when p == 0 or in eval
        def __init__(self, p, mode):
            super().__init__()
            self.p, self.mode = p, mode

        def forward(self, x):
            assert (not self.training) or self.p == 0.0
            return x

    _mod("iopath"); _mod("iopath.common")
    _mod("iopath.common.file_io", PathManager=PathManager, HTTPURLHandler=HTTPURLHandler)
    _mod("torchvision"); _mod("torchvision.models")
    _mod("torchvision.models.resnet", Bottleneck=_Absent, ResNet=_Absent)
    _mod("torchvision.ops"); _mod("torchvision.ops.stochastic_depth", StochasticDepth=StochasticDepth)
    sys.path.append(REFERENCE_ROOT)  # append: the reference also has a top-level `tests` package


def install_transform_stubs() -> None:
    """Stand-ins that let torchmultimodal/transforms/{clip,flava}_transform.py be IMPORTED here (ftfy and torchvision.transforms are
    absent): the tokenizer's encode() path and the mask generator touch none of them; the image classes are never instantiated."""
    install()
    if "torchvision.transforms" in sys.modules:
        return

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Base:
        def __init__(self, *a, **k):
            pass

    _mod("ftfy", fix_text=lambda s: s)
    modes = types.SimpleNamespace(BICUBIC="bicubic", LANCZOS="lanczos")
    tr = _mod("torchvision.transforms", InterpolationMode=modes, Resize=_Base, RandomResizedCrop=_Base, Compose=_Base, Lambda=_Base,
              ToTensor=_Base, Normalize=_Base)
    fn = _mod("torchvision.transforms.functional")
    sys.modules["torchvision"].transforms = tr
    tr.functional = fn
    import torchmultimodal

    torchmultimodal._PATH_MANAGER.open = lambda p, *a, **k: open(p, *a, **k)
    torchmultimodal._PATH_MANAGER.get_local_path = lambda p: p
