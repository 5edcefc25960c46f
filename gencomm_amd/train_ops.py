"""Training building blocks on the HIP kernels (NCHW fp32): the forward-recompute and backward primitives the Enhancer's
backward is composed of in ``autograd.py`` -- general convolution (3x3 / 1x1 = Linear) with its input and weight gradients,
LayerNorm over channels, depthwise 3x3 convolution, erf-GELU backward, and the sampling / scatter halves of the message extractor's
deformable convolution. Thin wrappers over the C ABI
(``gencomm_conv2d_{prepare,fold,fwd,wgrad}``, ``gencomm_ln_nchw_{fwd,bwd}``, ``gencomm_dwconv3x3_{act_fwd,act_wgrad}``,
``gencomm_gelu_bwd``); no torch convolution / normalisation call anywhere."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from .runtime import conv2d_prepare, ptr, stream_ptr, zeros as pool_zeros


def _c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


def conv2d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], pad: int, stride: int = 1, residual: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, out_coff: int = 0) -> torch.Tensor:
    """y = conv(x, w [Cout, Cin, K, K]) + b (+ residual), K in {1, 3} (csrc/conv_kernels.h). `out` / `out_coff`: write the result into
    channels [out_coff, out_coff + Cout) of an existing [n, ctotal, Ho, Wo] tensor (replaces a torch.cat)."""
    x, w = _c(x), _c(w)
    n, cin, H, W = x.shape
    cout, _, kh, kw = w.shape
    l, st, dev = _lib.lib(), stream_ptr(x.device), x.device
    prepared = conv2d_prepare(w, cin, cout, kh, kw, 0, dev)
    unit = _unit_scale_shift(cout, dev)                     # scale 1; shift = the bias itself (or the cached zeros): no fold launch
    ss = (unit[0], _c(b) if b is not None else unit[1])
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if residual is not None:
        assert out is None
        y = torch.empty(n, cout, Ho, Wo, dtype=torch.float32, device=dev)
        _lib.check(l.gencomm_conv2d_act_res_fwd(ptr(x), ptr(prepared), ptr(ss[0]), ptr(ss[1]), ptr(_c(residual)), ptr(y), n, cin, H, W, cout, kh, kw,
                                                stride, pad, 0, st), "gencomm_conv2d_act_res_fwd")
        return y
    y = out if out is not None else torch.empty(n, cout, Ho, Wo, dtype=torch.float32, device=dev)
    _lib.check(l.gencomm_conv2d_fwd(ptr(x), ptr(prepared), ptr(ss[0]), ptr(ss[1]), ptr(y), n, cin, H, W, cout, kh, kw, stride, pad, 0, 1,
                                    y.shape[1], int(out_coff), st), "gencomm_conv2d_fwd")
    return y


_UNIT_SS = {}
_S2_TAPS = {}


def _unit_scale_shift(c: int, device) -> torch.Tensor:
    """[2, c]: scale 1, shift 0 (a convolution without bias or folded norm) -- cached per (c, device): no fold launch per call."""
    key = (c, str(device))
    if key not in _UNIT_SS:
        ss = torch.zeros(2, c, dtype=torch.float32, device=device)
        ss[0].fill_(1.0)
        _UNIT_SS[key] = ss
    return _UNIT_SS[key]


def conv2d_dgrad(dy: torch.Tensor, w: torch.Tensor, pad: int) -> torch.Tensor:
    """Input gradient of a stride-1 convolution: the same kernel with the transposed, tap-flipped weight -- laid out by ONE prepare
    launch (gencomm_conv2d_prepare, transposed = 2) instead of flip + transpose + contiguous + prepare, and with a cached unit
    scale / zero shift instead of a fold launch."""
    dy, w = _c(dy), _c(w)
    cout, cin, kh, kw = w.shape            # forward weights: the gradient convolution maps cout -> cin channels
    n, _, H, W = dy.shape
    l, st, dev = _lib.lib(), stream_ptr(dy.device), dy.device
    prepared = conv2d_prepare(w, cout, cin, kh, kw, 2, dev)
    ss = _unit_scale_shift(cin, dev)
    p = kh - 1 - pad
    Ho, Wo = H + 2 * p - kh + 1, W + 2 * p - kw + 1
    dx = torch.empty(n, cin, Ho, Wo, dtype=torch.float32, device=dev)
    _lib.check(l.gencomm_conv2d_fwd(ptr(dy), ptr(prepared), ptr(ss[0]), ptr(ss[1]), ptr(dx), n, cout, H, W, cin, kh, kw, 1, p, 0, 1, cin, 0, st),
               "gencomm_conv2d_fwd")
    return dx


def conv2d_wgrad(dy: torch.Tensor, x: torch.Tensor, k: int, pad: int, bias: bool, stride: int = 1) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    dy, x = _c(dy), _c(x)
    n, cout = dy.shape[:2]
    cin, H, W = x.shape[1:]
    l = _lib.lib()
    # one zero-filled blob for both gradients (one fill launch instead of two)
    blob = pool_zeros(cout * cin * k * k + (cout if bias else 0), torch.float32, x.device)   # carved from a zero-filled block: no launch
    dw = blob[:cout * cin * k * k].view(cout, cin, k, k)
    db = blob[cout * cin * k * k:] if bias else None
    need = _lib.check_size(l.gencomm_conv2d_wgrad_scratch_floats(n, cin, H, W, cout, k, int(stride), pad), "gencomm_conv2d_wgrad_scratch_floats")
    scratch = torch.empty(need, dtype=torch.float32, device=x.device) if need else None   # wide 3x3 layers: per-workgroup partial sums
    _lib.check(l.gencomm_conv2d_wgrad_ws(ptr(dy), ptr(x), ptr(dw), ptr(db), n, cin, H, W, cout, k, int(stride), pad, ptr(scratch), need,
                                         stream_ptr(x.device)), "gencomm_conv2d_wgrad_ws")
    return dw, db


def conv2d_dgrad_strided(dy: torch.Tensor, w: torch.Tensor, pad: int, stride: int, in_hw: Tuple[int, int]) -> torch.Tensor:
    """Input gradient of a stride-s convolution: dy is spread onto the stride-1 output grid (zeros in between: a strided copy),
    then the stride-1 input-gradient convolution runs on the HIP kernel. `in_hw` = (H, W) of the forward input."""
    if stride == 1:
        return conv2d_dgrad(dy, w, pad)
    k = w.shape[2]
    H, W = in_hw
    if stride == 2 and k == 3 and pad == 1 and H % 2 == 0 and W % 2 == 0 and dy.shape[2] == H // 2 and dy.shape[3] == W // 2:
        return _conv3x3_s2_dgrad_subpixel(dy, w)
    up = torch.zeros(dy.shape[0], dy.shape[1], H + 2 * pad - k + 1, W + 2 * pad - k + 1, dtype=torch.float32, device=dy.device)
    up[:, :, ::stride, ::stride] = dy
    return conv2d_dgrad(up, w, pad)


def s2_subpixel_weights(w: torch.Tensor) -> torch.Tensor:
    """[ci, a, b, co, ty, tx] weights of the 2x2 sub-pixel form of a 3x3 stride-2 pad-1 convolution's input gradient from its forward weight
    w [co, ci, 3, 3]: entry = w[co][ci][a + 1 - 2 ty][b + 1 - 2 tx], zero where that tap does not exist.  One gather from the taps padded
    with a zero (index 9) instead of a zero fill + nine slice assignments -- the weights change every training step, so this runs per call
    (ten launches and 0.2 ms of host time before)."""
    cout, cin = w.shape[0], w.shape[1]
    key = str(w.device)
    if key not in _S2_TAPS:
        idx = torch.full((2, 2, 2, 2), 9, dtype=torch.long)                          # [a, b, ty, tx] -> ky * 3 + kx, or 9 = the zero tap
        for a in (0, 1):
            for ty in (0, 1):
                for b in (0, 1):
                    for tx in (0, 1):
                        ky, kx = a + 1 - 2 * ty, b + 1 - 2 * tx
                        if 0 <= ky <= 2 and 0 <= kx <= 2:
                            idx[a, b, ty, tx] = ky * 3 + kx
        _S2_TAPS[key] = idx.reshape(-1).to(w.device)
    wz = torch.nn.functional.pad(w.reshape(cout, cin, 9), (0, 1))                    # [co, ci, 10]
    return wz.index_select(2, _S2_TAPS[key]).view(cout, cin, 2, 2, 2, 2).permute(1, 2, 3, 0, 4, 5).contiguous()


def _conv3x3_s2_dgrad_subpixel(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """dx of a 3x3 stride-2 pad-1 convolution over an even-sized input as ONE 2x2 sub-pixel convolution over dy
    (gencomm_conv2d_fwd, KH = KW = 2, ups = 2): dx[ci][2u + a][2v + b] = sum_{co, ty, tx} W[co][ci][a + 1 - 2 ty][b + 1 - 2 tx] dy[co][u + ty][v + tx]
    (taps outside 0..2 are zero).  16 tap-products per output quad instead of the 36 of the zero-stuffed form below."""
    dy, w = _c(dy), _c(w)
    n, cout, Ho, Wo = dy.shape
    cin = w.shape[1]
    dev = dy.device
    wp = s2_subpixel_weights(w)
    l, st = _lib.lib(), stream_ptr(dev)
    prepared = conv2d_prepare(wp, cout, cin * 4, 2, 2, 0, dev)
    ss = _unit_scale_shift(cin, dev)
    dx = torch.empty(n, cin, 2 * Ho, 2 * Wo, dtype=torch.float32, device=dev)
    _lib.check(l.gencomm_conv2d_fwd(ptr(dy), ptr(prepared), ptr(ss[0]), ptr(ss[1]), ptr(dx), n, cout, Ho, Wo, cin, 2, 2, 1, 0, 0, 2, cin, 0, st),
               "gencomm_conv2d_fwd")
    return dx


def conv3x3_c16(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """3x3 pad-1 convolution of the first 16 channels of x [n, >= 16, H, W] with w [16, 16, 3, 3] (no bias) into the first 16 channels of
    `out` (gencomm_conv3x3_c16_fwd: the UNet's 8-channel kernel); transposed: the layer's input gradient from its forward weight."""
    n, x_ct, H, W = x.shape
    if not (x.is_contiguous() and out.is_contiguous() and x.dtype == out.dtype == torch.float32 and x_ct >= 16 and out.shape[1] >= 16
            and tuple(w.shape) == (16, 16, 3, 3) and out.shape[0] == n and out.shape[2:] == x.shape[2:]):
        raise ValueError(f"conv3x3_c16: contiguous f32 x [n, >= 16, H, W] / out [n, >= 16, H, W] and w [16, 16, 3, 3] expected, got "
                         f"{tuple(x.shape)}, {tuple(out.shape)}, {tuple(w.shape)}")
    l = _lib.lib()
    scratch = torch.empty(_lib.check_size(l.gencomm_conv3x3_c16_scratch_floats(), "gencomm_conv3x3_c16_scratch_floats"), dtype=torch.float32, device=x.device)
    _lib.check(l.gencomm_conv3x3_c16_fwd(ptr(x), x_ct, ptr(_c(w)), int(transposed), ptr(out), out.shape[1], ptr(scratch), n, H, W, stream_ptr(x.device)),
               "gencomm_conv3x3_c16_fwd")
    return out


def bn2d_train_fwd(x: torch.Tensor, bn, relu: bool):
    """(y, save): BatchNorm2d with batch statistics (+ ReLU) on the HIP kernels; updates bn.running_mean / running_var /
    num_batches_tracked exactly as nn.BatchNorm2d in training mode does."""
    x = _c(x)
    n, C, H, W = x.shape
    y = torch.empty_like(x)
    save = torch.empty(C, 2, dtype=torch.float32, device=x.device)
    scratch = pool_zeros(2 * C, torch.float64, x.device)          # arrives zeroed (flag 4): no memset launch inside the call
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = 0.0 if bn.momentum is None else float(bn.momentum)
    nbt = None
    if track:
        if bn.momentum is None:                                   # cumulative average: the factor is 1 / the counter AFTER this batch (a host read)
            bn.num_batches_tracked += 1
            momentum = 1.0 / float(bn.num_batches_tracked)
        elif bn.num_batches_tracked is not None and bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64:
            nbt = bn.num_batches_tracked                          # incremented inside the normalisation kernel (ABI v9): no launch of its own
        elif bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
    _lib.check(_lib.lib().gencomm_bn2d_train_fwd(ptr(x), ptr(_c(bn.weight)), ptr(_c(bn.bias)), ptr(bn.running_mean) if track else None,
                                                 ptr(bn.running_var) if track else None, ptr(y), ptr(save), ptr(scratch), momentum, float(bn.eps),
                                                 int(relu) | 4, n, C, H * W, ptr(nbt), stream_ptr(x.device)), "gencomm_bn2d_train_fwd")
    return y, save


def bn2d_train_bwd(x: torch.Tensor, y: torch.Tensor, dy: torch.Tensor, save: torch.Tensor, gamma: torch.Tensor, relu: bool):
    """(dx, dgamma, dbeta) of y = act(BN_batch(x))."""
    x, y, dy = _c(x), _c(y), _c(dy)
    n, C, H, W = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty(2, C, dtype=torch.float32, device=x.device).unbind(0)   # written, not accumulated (relu bit 1): no fill launch
    scratch = pool_zeros(2 * C, torch.float64, x.device)          # arrives zeroed (flag 4)
    _lib.check(_lib.lib().gencomm_bn2d_train_bwd(ptr(x), ptr(y), ptr(dy), ptr(save), ptr(_c(gamma)), ptr(dx), ptr(dg), ptr(db), ptr(scratch), int(relu) | 2 | 4,
                                                 n, C, H * W, stream_ptr(x.device)), "gencomm_bn2d_train_bwd")
    return dx, dg, db


def ln_fwd(x: torch.Tensor, gamma, beta, eps: float, residual: bool) -> torch.Tensor:
    x = _c(x)
    n, C, H, W = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.lib().gencomm_ln_nchw_fwd(ptr(x), ptr(_c(gamma)), ptr(_c(beta)), ptr(out), float(eps), int(residual), n, C, H * W,
                                              stream_ptr(x.device)), "gencomm_ln_nchw_fwd")
    return out


def ln_bwd(x: torch.Tensor, gamma, dy: torch.Tensor, eps: float, accumulate_into: Optional[torch.Tensor] = None):
    """(dx, dgamma, dbeta); `accumulate_into`: dx is ADDED to that tensor (which is returned) instead of a fresh one."""
    x, dy = _c(x), _c(dy)
    n, C, H, W = x.shape
    dx = accumulate_into if accumulate_into is not None else torch.empty_like(x)
    dg, db = pool_zeros((2, C), torch.float32, x.device).unbind(0)   # carved from a zero-filled block: no launch
    scratch = torch.empty(n * H * W * 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gencomm_ln_nchw_bwd(ptr(x), ptr(_c(gamma)), ptr(dy), ptr(dx), ptr(dg), ptr(db), ptr(scratch), float(eps),
                                              int(accumulate_into is not None), n, C, H * W, stream_ptr(x.device)), "gencomm_ln_nchw_bwd")
    return dx, dg, db


EW_COPY, EW_GELU_SPLIT, EW_GELU_GATE, EW_GATE_BWD, EW_GELU_BWD, EW_GELU2_GATE, EW_GATE_BWD2 = range(7)


def ew_slice(op: int, a, b=None, c=None, d=None, o0=None, o1=None, *, n: int, nch: int, HW: int, a_ct: int = 0, a_c0: int = 0,
             o0_ct: int = 0, o0_c0: int = 0, o1_ct: int = 0, o1_c0: int = 0) -> None:
    """gencomm_ew_slice_fwd (include/gencomm_hip.h): the elementwise pieces of the Enhancer's backward on channel slices."""
    _lib.check(_lib.lib().gencomm_ew_slice_fwd(int(op), ptr(a), ptr(b), ptr(c), ptr(d), ptr(o0), ptr(o1), n, nch, HW, a_ct, a_c0, o0_ct, o0_c0,
                                               o1_ct, o1_c0, stream_ptr(a.device)), "gencomm_ew_slice_fwd")


def copy_slice(src: torch.Tensor, c0: int, nch: int, dst: Optional[torch.Tensor] = None, dst_c0: int = 0) -> torch.Tensor:
    """dst[:, dst_c0:dst_c0 + nch] = src[:, c0:c0 + nch] (dst: a fresh [n, nch, H, W] tensor when None)."""
    n, C, H, W = src.shape
    if dst is None:
        dst = torch.empty(n, nch, H, W, dtype=torch.float32, device=src.device)
    ew_slice(EW_COPY, src, o0=dst, n=n, nch=nch, HW=H * W, a_ct=C, a_c0=c0, o0_ct=dst.shape[1], o0_c0=dst_c0)
    return dst


def nc_scale(x: torch.Tensor, a: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """out[n, c] = x[n, c] * a[n, c] + b[n, c] over the pixels."""
    n, C, H, W = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.lib().gencomm_nc_scale_fwd(ptr(x), ptr(_c(a)), ptr(_c(b)) if b is not None else None, ptr(out), n, C, H * W, stream_ptr(x.device)),
               "gencomm_nc_scale_fwd")
    return out


def nc_dot(x: torch.Tensor, y: Optional[torch.Tensor]) -> torch.Tensor:
    """[n, C]: sum over the pixels of x (* y), float64 accumulation."""
    n, C, H, W = x.shape
    out = torch.empty(n, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gencomm_nc_dot_fwd(ptr(x), ptr(y), ptr(out), n, C, H * W, stream_ptr(x.device)), "gencomm_nc_dot_fwd")
    return out


def dwconv3x3(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], flip: bool = False, gelu_in: bool = False) -> torch.Tensor:
    """Depthwise 3x3 over the FIRST C = w.shape[0] channels of x (x may carry more: Linear1's [n, 2 hidden, H, W] output), gelu_in: the
    layer's input is GELU(x), evaluated on load (gencomm_dwconv3x3_act_fwd)."""
    x = _c(x)
    n, x_ct, H, W = x.shape
    C = w.shape[0]
    y = torch.empty(n, C, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gencomm_dwconv3x3_act_fwd(ptr(x), x_ct, int(gelu_in), ptr(_c(w)), ptr(_c(b)) if b is not None else None, ptr(y), n, C, H, W,
                                                    int(flip), stream_ptr(x.device)), "gencomm_dwconv3x3_act_fwd")
    return y


def dwconv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, gelu_in: bool = False):
    """dw, db of the depthwise layer whose input was the first dy.shape[1] channels of x (GELU of them when gelu_in)."""
    x, dy = _c(x), _c(dy)
    n, C, H, W = dy.shape
    blob = pool_zeros(C * 10, torch.float32, x.device)              # both gradients, carved from a zero-filled block: no launch
    dw, db = blob[:C * 9].view(C, 1, 3, 3), blob[C * 9:]
    _lib.check(_lib.lib().gencomm_dwconv3x3_act_wgrad(ptr(x), x.shape[1], int(gelu_in), ptr(dy), ptr(dw), ptr(db), n, C, H, W, stream_ptr(x.device)),
               "gencomm_dwconv3x3_act_wgrad")
    return dw, db


def gelu_bwd(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    v, g = _c(v), _c(g)
    out = torch.empty_like(v)
    _lib.check(_lib.lib().gencomm_gelu_bwd(ptr(v), ptr(g), ptr(out), v.numel(), stream_ptr(v.device)), "gencomm_gelu_bwd")
    return out


def dcn_sample(x: torch.Tensor, offset: torch.Tensor) -> torch.Tensor:
    """col [n, 9 C, H, W]: the bilinear samples the 3x3 deformable convolution multiplies its weights with (channel c * 9 + tap)."""
    x, offset = _c(x), _c(offset)
    n, C, H, W = x.shape
    col = torch.empty(n, C * 9, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gencomm_dcn_sample_fwd(ptr(x), ptr(offset), ptr(col), n, C, H, W, stream_ptr(x.device)), "gencomm_dcn_sample_fwd")
    return col


def dcn_scatter_bwd(x: torch.Tensor, offset: torch.Tensor, dcol: torch.Tensor):
    """(dx [n, C, H, W], doffset [n, 18, H, W]) from the gradient of the sampled columns."""
    x, offset, dcol = _c(x), _c(offset), _c(dcol)
    n, C, H, W = x.shape
    dx = torch.zeros_like(x)
    doff = torch.empty(n, 18, H, W, dtype=torch.float32, device=x.device)
    l = _lib.lib()
    need = _lib.check_size(l.gencomm_dcn_scatter_scratch_floats(n, C, H, W), "gencomm_dcn_scatter_scratch_floats")
    scratch = torch.empty(need, dtype=torch.float32, device=x.device)   # per-tile regions of the input gradient (no global atomics)
    _lib.check(l.gencomm_dcn_scatter_bwd_ws(ptr(x), ptr(offset), ptr(dcol), ptr(dx), ptr(doff), n, C, H, W, ptr(scratch), need, stream_ptr(x.device)),
               "gencomm_dcn_scatter_bwd_ws")
    return dx, doff
