"""Device-memory plumbing between torch tensors and the C ABI: pointers, the current HIP stream,
a growable per-device scratch workspace, and parameter-blob packing with change detection."""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import torch

from . import _lib


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.GenCommHipError(
            f"{what}: tensor is on {t.device}; the GenComm hot path runs only as HIP kernels on a "
            "ROCm device (gfx950) and has no CPU fallback")


def f32c(t: torch.Tensor) -> torch.Tensor:
    """float32 + contiguous view/copy (AMP callers hand in fp16/bf16; arithmetic here is fp32)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


_PREP_FLOATS: Dict[Tuple[int, int, int, int, int], int] = {}


def conv2d_prepare(w: torch.Tensor, cin: int, cout: int, kh: int, kw: int, transposed: int, device: torch.device) -> torch.Tensor:
    """gencomm_conv2d_prepare into a buffer of gencomm_conv2d_prepared_floats floats: the fp32 k-major matrix and, for the shapes the
    f16-pipe kernel takes, the three-term operand form behind it (one launch).  `w`: contiguous fp32 in the layout `transposed` names."""
    key = (cin, cout, kh, kw, int(transposed))
    n = _PREP_FLOATS.get(key)
    l = _lib.lib()
    if n is None:
        n = _PREP_FLOATS[key] = _lib.check_size(l.gencomm_conv2d_prepared_floats(*key), "gencomm_conv2d_prepared_floats")
    prepared = torch.empty(n, dtype=torch.float32, device=device)
    _lib.check(l.gencomm_conv2d_prepare(ptr(w), ptr(prepared), cin, cout, kh, kw, int(transposed), stream_ptr(device)), "gencomm_conv2d_prepare")
    return prepared


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device) -> int:
    """The hipStream_t of torch's current stream on `device`. Called once per kernel launch (1 750 times per training-leg step):
    torch's raw accessor returns the handle without building a `torch.cuda.Stream` object (measured per call in
    tools/diag/host_call_cost.py); the object path stays as the fallback for a build without it."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


class _Workspaces:
    """One growable byte buffer per (device, HIP stream, tag). Kernels are stream-ordered on the caller's current stream, so
    reuse across calls on that stream is safe; two streams (or threads on different streams) never share scratch. A buffer
    that is outgrown is released only after the stream that used it has drained."""

    def __init__(self):
        self._bufs: Dict[Tuple[int, int, str], torch.Tensor] = {}

    def get(self, device: torch.device, nbytes: int, tag: str = "main") -> torch.Tensor:
        dev = device.index if device.index is not None else torch.cuda.current_device()
        stream = torch.cuda.current_stream(device)
        key = (dev, stream.cuda_stream, tag)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                stream.synchronize()  # kernels already enqueued on this stream may still use the old buffer
                self._bufs.pop(key, None)
                del buf
            buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf

    def clear(self) -> None:
        self._bufs.clear()


workspaces = _Workspaces()


class _ZeroPool:
    """``zeros(shape, dtype, device)`` carved out of zero-filled blocks: ONE fill launch per ``BLOCK`` bytes instead of one per tensor
    (the training leg asked for ~230 small zero-filled tensors per step -- weight-gradient blobs, statistics scratch -- each a launch of
    its own: ``profiles/r4_train_leg_kernel_stats.csv``).  A carved tensor is a view that keeps its block alive and is never handed out
    twice, so nothing aliases: a block is freed (back to torch's caching allocator) when its last view dies.  One pool per (device,
    stream): the fill is ordered on the stream the views are used on.  While a HIP graph is being captured the pool steps aside
    (plain ``torch.zeros``): a view of a block filled before the capture began would not be re-zeroed by a replay."""
    BLOCK = 64 << 20   # one training-leg step asks for ~60 MB of zeros (weight-gradient blobs up to 2.4 MB, statistics scratch): one fill per step
    ALIGN = 256

    def __init__(self):
        self._cur: Dict[Tuple[int, int], list] = {}

    def zeros(self, shape, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        numel = 1
        for d in shape:
            numel *= int(d)
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        if nbytes == 0 or nbytes > self.BLOCK // 4 or device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return torch.zeros(shape, dtype=dtype, device=device)
        dev = device.index if device.index is not None else torch.cuda.current_device()
        key = (dev, stream_ptr(device))
        cur = self._cur.get(key)
        need = (nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if cur is None or cur[1] + need > self.BLOCK:
            cur = [torch.zeros(self.BLOCK, dtype=torch.uint8, device=device), 0]
            self._cur[key] = cur
        off = cur[1]
        cur[1] = off + need
        return cur[0][off:off + nbytes].view(dtype).view(shape)

    def clear(self) -> None:
        self._cur.clear()


zero_pool = _ZeroPool()


def zeros(shape, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """Zero-filled tensor from the per-stream pool (see ``_ZeroPool``); same contract as ``torch.zeros``."""
    return zero_pool.zeros(shape, dtype, device)


class PackedParams:
    """Flat float32 copy of selected parameters in the order the C side enumerates them, rebuilt
    only when a parameter changed (``_version`` / storage pointer / device)."""

    def __init__(self, table: Sequence[Tuple[str, int, int]], total_floats: int):
        self.table = list(table)
        self.total = total_floats
        self._key = None
        self.flat: torch.Tensor = None
        self.generation = 0

    def update(self, named: Dict[str, torch.Tensor]) -> bool:
        """Returns True when the blob was (re)built."""
        tensors = []
        for name, numel, _ in self.table:
            if name not in named:
                raise KeyError(f"parameter '{name}' expected by the HIP library is missing from the module")
            t = named[name]
            if t.numel() != numel:
                raise ValueError(f"parameter '{name}' has {t.numel()} elements, the HIP library expects {numel}")
            tensors.append(t)
        if any(t.device != tensors[0].device or t.dtype != tensors[0].dtype for t in tensors):
            raise ValueError("the parameters handed to the HIP library must share one device and dtype")
        key = (tensors[0].device, tensors[0].dtype) + tuple([(t.data_ptr(), t._version) for t in tensors])   # storage + in-place version of each
        if key == self._key and self.flat is not None:
            return False
        with torch.no_grad():
            self.flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
        assert self.flat.numel() == self.total, (self.flat.numel(), self.total)
        self._key = key
        self.generation += 1
        return True


_INT_CONSTS: dict = {}


def dev_ints(values, device, dtype=torch.int32) -> torch.Tensor:
    """A small integer table (scene offsets, source rows ...) as a device tensor, cached by value: building it with
    ``torch.tensor(list, device=...)`` on every forward is a pageable host-to-device copy, i.e. a host synchronisation in the middle
    of the model (the host then issues the ~100 launches that follow at its own pace instead of running ahead of the GPU)."""
    key = (tuple(int(v) for v in values), str(device), dtype)
    t = _INT_CONSTS.get(key)
    if t is None:
        if len(_INT_CONSTS) > 4096:
            _INT_CONSTS.clear()
        t = torch.tensor(list(key[0]), dtype=dtype, device=device)
        _INT_CONSTS[key] = t
    return t


def record_len_list(record_len) -> List[int]:
    """Scene lengths as Python ints (one D2H sync when given a device tensor; the reference's
    ``regroup`` does the same ``.cpu()`` on every call, fusion_in_one.py:48-51)."""
    if record_len is None:
        return None
    if isinstance(record_len, torch.Tensor):
        return [int(v) for v in record_len.detach().cpu().tolist()]
    return [int(v) for v in record_len]


# ----------------------------------------------------------------------------------------- side stream for weight gradients
_SIDE_STREAMS: dict = {}
SIDE_MIN_PIXELS = 1 << 17      # n * H * W below which the fork / join costs the host more than the overlap returns (measured: 64 x 128 maps)


def side_stream(dev: torch.device) -> "torch.cuda.Stream":
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


class overlap:
    """``with overlap(device, pixels) as ov: dw = ov.run(lambda: wgrad(...)); dx = dgrad(...)`` -- work that nothing inside the block
    depends on (weight gradients) runs on the device's side stream beside the block's own launches and is joined when the block ends:
    from there on its results are ordered on the current stream.  Inputs of the side work must not be written inside the block.
    Disabled (plain calls) on small maps and when GENCOMM_MODE_BWD_STREAMS is 0."""

    def __init__(self, device: torch.device, pixels: int):
        from . import _lib
        mode = _lib.lib().gencomm_get_mode(_lib.MODE_BWD_STREAMS)
        self.on = device.type == "cuda" and (mode == 2 or (mode == 1 and pixels >= SIDE_MIN_PIXELS))
        self.device, self.outs, self.used = device, [], False

    def __enter__(self):
        if self.on:
            self.cur, self.side = torch.cuda.current_stream(self.device), side_stream(self.device)
        return self

    def run(self, fn):
        if not self.on:
            return fn()
        self.side.wait_stream(self.cur)
        with torch.cuda.stream(self.side):
            out = fn()
        self.used = True
        self.outs.extend(t for t in (out if isinstance(out, (tuple, list)) else (out,)) if isinstance(t, torch.Tensor))
        return out

    def join(self):
        if self.on and self.used:
            self.cur.wait_stream(self.side)
            for t in self.outs:              # allocated under the side stream, consumed on the current one from here on
                t.record_stream(self.cur)
            self.outs, self.used = [], False

    def __exit__(self, *exc):
        self.join()
        return False
