"""Build libmmamd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m multimodal_amd.build [--force] [--verbose]

hipcc cross-compiles gfx950 without a GPU.  Objects are rebuilt only when the source (or a header, or the
flags) changed; the .so lands in multimodal_amd/lib/ and travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
OBJDIR = LIBDIR / "obj"
LIB = LIBDIR / "libmmamd.so"
TORCH_LIB = LIBDIR / "libmmamd_torch.so"  # TORCH_LIBRARY(mmamd, ...) shim over the C-ABI (csrc/torch_ops.cpp)
INCLUDE = PKG.parent / "include"

ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-Wno-return-type"]
if os.environ.get("MMAMD_EXPERIMENTS") == "1":  # also build the schedule experiments / ablations / traces of gemm.hip
    CXXFLAGS.append("-DMMAMD_EXPERIMENTS")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _digest(src: Path) -> str:
    h = hashlib.sha256()
    h.update(" ".join(CXXFLAGS).encode())
    for f in [src, *sorted(CSRC.glob("*.h")), *sorted(INCLUDE.glob("*.h")), *sorted(CSRC.glob("*.inc")), *sorted(CSRC.glob("experiments/*.inc"))]:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def _compile(src: Path, force: bool, verbose: bool) -> tuple[Path, bool]:
    obj = OBJDIR / (src.stem + ".o")
    stamp = OBJDIR / (src.stem + ".sha")
    dig = _digest(src)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj, False
    cmd = [_hipcc(), *CXXFLAGS, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    stamp.write_text(dig)
    return obj, True


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJDIR.mkdir(parents=True, exist_ok=True)
    sources = sorted(CSRC.glob("*.hip"))
    if not sources:
        raise RuntimeError(f"no .hip sources under {CSRC}")
    with ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, verbose), sources))
    objs = [o for o, _ in results]
    if force or not LIB.exists() or any(changed for _, changed in results):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB


def build_torch_ops(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/torch_ops.cpp (host C++: TORCH_LIBRARY registrations that call the C-ABI) against the installed torch and link it
    to libmmamd.so.  No kernels in it: hipcc is used as the C++ driver because torch's HIP headers need the HIP platform defines."""
    import torch

    build(force=False, verbose=verbose)
    src = CSRC / "torch_ops.cpp"
    tdir = Path(torch.__file__).resolve().parent
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-I{tdir / 'include'}",
             f"-I{tdir / 'include' / 'torch' / 'csrc' / 'api' / 'include'}", "-I/opt/rocm/include", "-Wno-unused-result"]
    h = hashlib.sha256()
    h.update((" ".join(flags) + torch.__version__).encode())
    for f in [src, *sorted(INCLUDE.glob("*.h"))]:
        h.update(f.read_bytes())
    dig = h.hexdigest()
    stamp = OBJDIR / "torch_ops.sha"
    if not force and TORCH_LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return TORCH_LIB
    cmd = [_hipcc(), *flags, str(src), "-o", str(TORCH_LIB), f"-L{LIBDIR}", "-lmmamd", f"-L{tdir / 'lib'}", "-ltorch", "-ltorch_cpu", "-lc10",
           "-ltorch_hip", "-lc10_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tdir / 'lib'}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"torch_ops.cpp failed to build:\n{res.stdout}\n{res.stderr}")
    stamp.write_text(dig)
    return TORCH_LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
    print(build_torch_ops(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
