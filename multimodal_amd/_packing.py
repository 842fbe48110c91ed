"""Cache of kernel-ready copies of module parameters.

The kernels want GEMM weights in bf16 and every small vector (biases, LayerNorm affine, embeddings tables'
positional rows, projections of the pooled row) in fp32, whatever dtype the nn.Parameters are kept in.
Conversions run on the HIP convert kernel (ops.convert) and are cached per parameter; a cached copy is
invalidated when the parameter's storage pointer or in-place version counter changes (load_state_dict,
optimizer.step, .to(...), any in-place op ON THE PARAMETER).  Parameters that already have the wanted dtype are used
in place (no copy).

LIMITATION: writes through `p.data` (`p.data.copy_(ckpt)`, `p.data.add_(...)` of a hand-written EMA, `module.weight.data.normal_()`)
do not bump torch's version counter, so they cannot be seen from here.  Three things cover the common cases: every top-level model
and encoder (PackedModeMixin) drops all packed copies on a train() <-> eval() transition (an EMA/averaged model is evaluated after
`.eval()`); `load_module_from_url` drops them after loading; and `invalidate_packed()` is public for everything else (call it after
editing `.data` of a module that has already run an inference forward).  The training path converts per step and is not affected.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

from . import ops


_EPOCH = 0  # bumped by invalidate_packed(): every PackedCache re-packs lazily on its next lookup


def invalidate_packed(model=None) -> None:
    """Drop every cached kernel-ready parameter copy (of all modules: the argument is accepted for readability only).  Needed after
    parameter edits that bypass torch's version counter (`.data` writes); cheap — copies are rebuilt on the next forward."""
    global _EPOCH
    _EPOCH += 1
    try:  # the scripted / compiled forwards keep their own C++ cache (csrc/torch_ops.cpp): drop it too
        from . import _torch_ops

        if _torch_ops.loaded():
            torch.ops.mmamd.clear_packed()
    except Exception:
        pass


def packed_epoch() -> int:
    """Counter bumped by invalidate_packed(): caches outside PackedCache compare it on lookup."""
    return _EPOCH


class PackedModeMixin:
    """nn.Module mixin (list it BEFORE nn.Module): a train() <-> eval() transition drops the packed parameter copies."""

    def train(self, mode: bool = True):
        if bool(mode) != self.training:
            invalidate_packed()
        return super().train(mode)


class _Ready:
    """Stream-safety of a cached copy: the conversion kernel runs on the stream that first asked for the copy; a forward that drives SEVERAL
    streams from one host thread (second tower on a side stream, the phased half-batch schedule) may look the copy up from another stream
    before that kernel has run.  The producer records an event; a lookup from a different stream waits for it (stream-side, no host sync);
    once the event has completed the guard is dropped, so the steady-state lookup costs one attribute test."""

    __slots__ = ("event", "stream")

    def __init__(self) -> None:
        self.stream = torch.cuda.current_stream()
        self.event = None
        if not torch.cuda.is_current_stream_capturing():
            self.event = torch.cuda.Event()
            self.event.record(self.stream)

    def guard(self) -> None:
        ev = self.event
        if ev is None:
            return
        cur = torch.cuda.current_stream()
        if cur == self.stream:
            return
        if torch.cuda.is_current_stream_capturing():
            cur.wait_event(ev)
        elif ev.query():
            self.event = None
        else:
            cur.wait_event(ev)


def _cacheable(p: torch.Tensor) -> bool:
    """A kernel-ready copy may be cached only for a tensor whose in-place updates bump a version counter we can see: an ordinary
    nn.Parameter.  Under FullyShardedDataParallel a module's `weight` is (use_orig_params=False) a plain Tensor VIEW of the unit's gathered
    flat parameter, recreated on every unshard, or (use_orig_params=True) a Parameter whose `.data` FSDP re-points at that buffer
    (`_fsdp_flattened`): the all-gather refills the same addresses after every optimizer step without touching any version counter, so a
    cached copy keyed on (id, data_ptr, _version) would go stale silently.  Those are converted per call (one small HIP launch each)."""
    return isinstance(p, torch.nn.Parameter) and not getattr(p, "_fsdp_flattened", False)


class PackedCache:
    def __init__(self) -> None:
        self._store: Dict[Tuple[int, torch.dtype], tuple] = {}
        self._cat: Dict[tuple, tuple] = {}
        self._epoch = _EPOCH

    # the cache is DERIVED data keyed by the identity of the owning module's parameters: a copy of the module (copy.deepcopy, pickle) has other
    # parameters and starts with an empty cache (the entries also hold HIP events, which neither copy nor pickle)
    def __deepcopy__(self, memo):
        return PackedCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def _sync_epoch(self) -> None:
        if self._epoch != _EPOCH:
            self.clear()
            self._epoch = _EPOCH

    def get(self, p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        self._sync_epoch()
        t = p.detach()
        if not t.is_cuda:
            raise ops.MmamdError(
                f"parameter lives on {t.device}: move the module to a HIP device (.to('cuda')); there is no CPU path")
        if t.dtype == dtype and t.is_contiguous():
            return t
        if not _cacheable(p):
            return ops.convert(t if t.is_contiguous() else t.contiguous(), dtype)
        key = (id(p), dtype)
        hit = self._store.get(key)
        if hit is not None and hit[0] == t.data_ptr() and hit[1] == p._version and hit[2] == t.device:
            if hit[4].event is not None:
                hit[4].guard()
            return hit[3]
        src = t if t.is_contiguous() else t.contiguous()
        conv = ops.convert(src, dtype)
        self._store[key] = (t.data_ptr(), p._version, t.device, conv, _Ready())
        return conv

    def get_cat(self, params: Sequence[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
        """One contiguous kernel-ready buffer holding `params` stacked along dim 0 (e.g. FLAVA's separate query / key /
        value Linear weights as ONE [3d, d] in-projection, so q, k and v come out of a single GEMM)."""
        self._sync_epoch()
        ts = [p.detach() for p in params]
        for t in ts:
            if not t.is_cuda:
                raise ops.MmamdError(
                    f"parameter lives on {t.device}: move the module to a HIP device (.to('cuda')); there is no CPU path")
        key = (tuple(id(p) for p in params), dtype)
        sig = tuple((t.data_ptr(), p._version) for t, p in zip(ts, params))
        hit = self._cat.get(key) if all(_cacheable(p) for p in params) else None
        if hit is not None and hit[0] == sig:
            if hit[2].event is not None:
                hit[2].guard()
            return hit[1]
        rows = sum(t.shape[0] for t in ts)
        buf = torch.empty((rows, *ts[0].shape[1:]), dtype=dtype, device=ts[0].device)
        r0 = 0
        for t in ts:
            src = t if t.is_contiguous() else t.contiguous()
            ops.convert(src, dtype, out=buf[r0:r0 + t.shape[0]])
            r0 += t.shape[0]
        self._cat[key] = (sig, buf, _Ready())
        return buf

    def get_padded_rows(self, p: torch.Tensor, dtype: torch.dtype, multiple: int) -> torch.Tensor:
        """`p` ([rows, ...]) with its row count rounded up to `multiple` (zero rows appended) — e.g. a [30522, d] vocabulary
        projection padded to the GEMM's N % 8 == 0."""
        self._sync_epoch()
        t = p.detach()
        if not t.is_cuda:
            raise ops.MmamdError(
                f"parameter lives on {t.device}: move the module to a HIP device (.to('cuda')); there is no CPU path")
        rows = t.shape[0]
        padded = (rows + multiple - 1) // multiple * multiple
        if padded == rows:
            return self.get(p, dtype)
        key = (("pad", id(p), multiple), dtype)
        sig = ((t.data_ptr(), p._version),)
        hit = self._cat.get(key) if _cacheable(p) else None
        if hit is not None and hit[0] == sig:
            if hit[2].event is not None:
                hit[2].guard()
            return hit[1]
        buf = torch.zeros((padded, *t.shape[1:]), dtype=dtype, device=t.device)  # zero fill = memset, not arithmetic
        ops.convert(t if t.is_contiguous() else t.contiguous(), dtype, out=buf[:rows])
        self._cat[key] = (sig, buf, _Ready())
        return buf

    def clear(self) -> None:
        self._cat.clear()
        self._store.clear()
