"""Tensor-level wrappers over the C-ABI (include/cmunet_hip.h).  PyTorch is used only for device
memory and streams; every op below runs a hand-written HIP kernel and raises if the library or a
GPU is missing (no CPU / eager fallback).

``Act`` describes an NHWC activation that may be a channel slice of a wider buffer and may carry a
*pending transform* (a training-mode BatchNorm+ReLU left for the consumer to apply while it stages
its input tile) -- see the header for the convention.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import F32, F16, BF16, call

TORCH_DT = {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16}
DT_OF = {v: k for k, v in TORCH_DT.items()}
DT_NAME = {F32: "f32", F16: "f16", BF16: "bf16"}
NAME_DT = {"f32": F32, "fp32": F32, "float32": F32, "f16": F16, "fp16": F16, "float16": F16, "bf16": BF16, "bfloat16": BF16}


def dt_code(dt):
    if isinstance(dt, str):
        return NAME_DT[dt]
    if isinstance(dt, torch.dtype):
        return DT_OF[dt]
    return int(dt)


def _stream():
    """The launch stream: resolved by ``_lib.call`` to the current stream of the device that owns the call's tensors."""
    return _lib.STREAM


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "HIP path: tensor must live on the GPU"
    return _lib.devptr(t.data_ptr(), t.device.index)



def mfma_sustained_rate(dt="f16", pattern=0, lds_fed=False, iters=60000, device="cuda"):
    """(TFLOP/s, shader clock in MHz) of back-to-back 16-bit MFMAs on every SIMD of the device (cmu_mfma_sustained_rate):
    the rate the chip's power management lets the matrix pipes hold on ``pattern`` 0 = dense ~N(0,1) operands, 1 = ReLU'd
    operands (half zeros), 2 = zeros; ``lds_fed``: operands read from LDS at the persistent conv kernel's fragment ratio instead
    of held in registers.  ~70 ms at the default ``iters``.  Measurement support for bench.py, not on the hot path."""
    scratch = torch.zeros(8, dtype=torch.int64, device=device)
    tf, clk = ctypes.c_double(0.0), ctypes.c_double(0.0)
    call("cmu_mfma_sustained_rate", dt_code(dt), int(pattern), int(bool(lds_fed)), int(iters), _p(scratch), ctypes.byref(tf),
         ctypes.byref(clk), _stream())
    return tf.value, clk.value


class dispatch_override:
    """``with ops.dispatch_override("CMU_CONV_NARROW", 0): ...`` -- force one of the library's A/B dispatch switches for the
    launches inside the block (cmu_set_dispatch_override; the switches' environment variables are read once, not per launch)."""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        call("cmu_set_dispatch_override", self.name, self.value)
        return self

    def __exit__(self, *exc):
        call("cmu_set_dispatch_override", self.name, -1)
        return False


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    return t


class Act:
    """NHWC activation view: channels [coff, coff+C) of ``buf`` (B,H,W,ld) + optional pending transform."""

    __slots__ = ("buf", "coff", "C", "scale", "shift", "relu_from")

    def __init__(self, buf, coff=0, C=None, scale=None, shift=None, relu_from=0):
        assert buf.dim() == 4 and buf.is_contiguous() and buf.is_cuda
        self.buf, self.coff = buf, coff
        self.C = buf.shape[3] - coff if C is None else C
        self.scale, self.shift, self.relu_from = scale, shift, relu_from

    @property
    def B(self):
        return self.buf.shape[0]

    @property
    def H(self):
        return self.buf.shape[1]

    @property
    def W(self):
        return self.buf.shape[2]

    @property
    def ld(self):
        return self.buf.shape[3]

    @property
    def dt(self):
        return DT_OF[self.buf.dtype]

    def ptr(self):
        return _lib.devptr(self.buf.data_ptr() + self.coff * self.buf.element_size(), self.buf.device.index)

    def with_transform(self, scale, shift, relu_from=0):
        return Act(self.buf, self.coff, self.C, scale, shift, relu_from)

    def plain(self):
        return Act(self.buf, self.coff, self.C)


def new_act(B, H, W, C, dt, device):
    return Act(torch.empty((B, H, W, C), dtype=TORCH_DT[dt_code(dt)], device=device))


# ------------------------------------------------------------------------------------------------
# packing
# ------------------------------------------------------------------------------------------------
def pack_conv3x3(w, dt, transpose_flip=False):
    dt = dt_code(dt)
    Cout, Cin = w.shape[0], w.shape[1]
    n = _lib.lib().cmu_pack_conv3x3_elems(Cin, Cout, dt, int(transpose_flip))
    out = torch.empty(n, dtype=TORCH_DT[dt], device=w.device)
    call("cmu_pack_conv3x3", _p(_f32c(w)), _p(out), Cin, Cout, dt, int(transpose_flip), _stream())
    return out


def pack_convT2x2(w, dt, mode):
    dt = dt_code(dt)
    Cin, Cout = w.shape[0], w.shape[1]
    n = _lib.lib().cmu_pack_convT2x2_elems(Cin, Cout, dt, mode)
    out = torch.empty(n, dtype=TORCH_DT[dt], device=w.device)
    call("cmu_pack_convT2x2", _p(_f32c(w)), _p(out), Cin, Cout, dt, mode, _stream())
    return out


class PackPlan:
    """Every conv / conv-transpose weight pack of a model as ONE launch (cmu_pack_batch).  ``items``: list of
    (weight fp32 tensor, kind 0 conv3x3 / 1 convT2x2, mode) -- conv3x3 mode = transpose_flip, convT mode 0 fwd / 1 dgrad.
    ``outs[i]`` is the packed tensor of item i (allocated once; the weights must keep their storage, as the flat parameter
    arena guarantees)."""

    def __init__(self, items, dt):
        import struct
        dt = dt_code(dt)
        l = _lib.lib()
        assert l.cmu_pack_desc_bytes() == 48
        self.dt, self.outs, recs, block = dt, [], [], 0
        dev = items[0][0].device
        for w, kind, mode in items:
            assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4
            if kind == 0:
                Cout, Cin = w.shape[0], w.shape[1]
                n = l.cmu_pack_conv3x3_elems(Cin, Cout, dt, int(mode))
            else:
                Cin, Cout = w.shape[0], w.shape[1]
                n = l.cmu_pack_convT2x2_elems(Cin, Cout, dt, int(mode))
            out = torch.empty(n, dtype=TORCH_DT[dt], device=dev)
            self.outs.append(out)
            recs.append(struct.pack("<QQiiiiqq", w.data_ptr(), out.data_ptr(), Cin, Cout, int(mode), kind, n, block))
            block += l.cmu_pack_desc_blocks(kind, Cin, Cout, dt, int(mode))
        self.ptrs = [w.data_ptr() for w, _, _ in items]
        self.items = items
        self.total_blocks = block
        self.descs = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)

    def run(self):
        assert all(w.data_ptr() == q for (w, _, _), q in zip(self.items, self.ptrs)), "a packed weight moved"
        call("cmu_pack_batch", _p(self.descs), len(self.items), self.total_blocks, self.dt, _stream())


# ------------------------------------------------------------------------------------------------
# forward
# ------------------------------------------------------------------------------------------------
def ntiles(B, H, W):
    return _lib.lib().cmu_conv_ntiles(B, H, W)


def new_stats(B, H, W, C, device):
    return torch.empty((ntiles(B, H, W), 2, C), dtype=torch.float32, device=device)


def conv3x3_c1_fwd(x_bhw, w, out, stats=None, mask=None, mask_per_sample=False):
    B, H, W = x_bhw.shape
    Cout = w.shape[0]
    assert out.C == Cout and (out.B, out.H, out.W) == (B, H, W)
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.is_contiguous()
    call("cmu_conv3x3_c1_fwd", _p(_f32c(x_bhw)), _p(mask), int(mask_per_sample), _p(_f32c(w)), out.ptr(), out.ld,
         _p(stats), B, H, W, Cout, out.dt, _stream())


def conv3x3_c1_fwd_tiles(x_bhw, w, out, tiles, max_tiles, mask=None, mask_per_sample=False, want_stats=True):
    """``conv3x3_c1_fwd`` over a TileList of 16 x 16 tiles (``out`` elsewhere untouched) -> slab [rows][2][Cout] of the sums over the
    listed tiles' pixels (None unless ``want_stats``).  ``max_tiles``: host-side upper bound of the list's count."""
    B, H, W = x_bhw.shape
    Cout = w.shape[0]
    assert out.C == Cout and (out.B, out.H, out.W) == (B, H, W) and (tiles.tile_h, tiles.tile_w) == (16, 16)
    stats = None
    if want_stats:
        stats = torch.empty((_lib.lib().cmu_conv3x3_c1_fwd_tiles_rows(int(max_tiles)), 2, Cout), dtype=torch.float32, device=out.buf.device)
    call("cmu_conv3x3_c1_fwd_tiles", _p(_f32c(x_bhw)), _p(mask), int(mask_per_sample), _p(_f32c(w)), out.ptr(), out.ld, _p(stats),
         _p(tiles.list), _p(tiles.count), int(max_tiles), B, H, W, Cout, out.dt, _stream())
    return stats


def conv3x3_c1_wgrad_bn_tiles(x_bhw, dA, yraw, scale, shift, save_mean, save_invstd, coef, dW, ws, tiles, max_tiles, mask=None,
                              mask_per_sample=False, w=None):
    """``conv3x3_c1_wgrad_bn`` with the contraction restricted to a TileList of 16 x 16 tiles (the gradient vanishes elsewhere)."""
    B, H, W = x_bhw.shape
    assert (tiles.tile_h, tiles.tile_w) == (16, 16)
    call("cmu_conv3x3_c1_wgrad_bn_tiles", _p(_f32c(x_bhw)), _p(mask), int(mask_per_sample), dA.ptr(), dA.ld,
         None if w is not None else yraw.ptr(), 0 if w is not None else yraw.ld, None if w is None else _p(_f32c(w)), _p(scale), _p(shift),
         _p(save_mean), _p(save_invstd), _p(coef), _p(tiles.list), _p(tiles.count), int(max_tiles), _p(_f32c(dW)), B, H, W, dW.shape[0], dA.dt,
         _p(ws), _stream())


def conv3x3_fwd(x, wpacked, out, stats=None):
    assert x.dt == out.dt and (x.B, x.H, x.W) == (out.B, out.H, out.W)
    call("cmu_conv3x3_fwd", x.ptr(), x.ld, _p(x.scale), _p(x.shift), x.relu_from, _p(wpacked), out.ptr(), out.ld,
         _p(stats), x.B, x.H, x.W, x.C, out.C, x.dt, _stream(), work=2.0 * 9 * x.C * out.C * x.B * x.H * x.W)


def bn_finalize(stats, count, conv_bias, gamma, beta, running_mean, running_var, momentum, eps, training,
                scale, shift, save_mean, save_invstd, ws):
    C = scale.numel()
    nt = 0 if stats is None else stats.shape[0]
    need = _lib.lib().cmu_bn_finalize_ws_bytes(C)
    assert ws is None or ws.numel() * ws.element_size() >= need
    call("cmu_bn_finalize", _p(stats), nt, int(count), _p(conv_bias), _p(gamma), _p(beta), _p(running_mean),
         _p(running_var), float(momentum), float(eps), int(training), _p(scale), _p(shift), _p(save_mean),
         _p(save_invstd), C, _p(ws), _stream())


def bnrelu_maxpool_fwd(y, out):
    assert y.scale is not None
    call("cmu_bnrelu_maxpool_fwd", y.ptr(), y.ld, _p(y.scale), _p(y.shift), out.ptr(), out.ld, y.B, y.H, y.W, y.C,
         y.dt, _stream())


def convT2x2_fwd(x, wpacked, bias, out):
    Cout = out.C
    assert (out.H, out.W) == (2 * x.H, 2 * x.W)
    call("cmu_convT2x2_fwd", x.ptr(), x.ld, _p(x.scale), _p(x.shift), x.relu_from, _p(wpacked), _p(_f32c(bias)),
         out.ptr(), out.ld, x.B, x.H, x.W, x.C, Cout, x.dt, _stream(), work=2.0 * 4 * x.C * Cout * x.B * x.H * x.W)


def conv1x1_head_fwd(x, w, bias, logits):
    K = w.shape[0]
    call("cmu_conv1x1_head_fwd", x.ptr(), x.ld, _p(x.scale), _p(x.shift), _p(_f32c(w)), _p(_f32c(bias)),
         _p(_f32c(logits)), x.B, x.H, x.W, x.C, K, x.dt, _stream())


def apply_to_nchw(y, out=None):
    if out is None:
        out = torch.empty((y.B, y.C, y.H, y.W), dtype=torch.float32, device=y.buf.device)
    call("cmu_apply_to_nchw", y.ptr(), y.ld, _p(y.scale), _p(y.shift), y.relu_from, _p(out), y.B, y.H, y.W, y.C,
         y.dt, _stream())
    return out


def nchw_to_nhwc(x_nchw, out):
    B, C, H, W = x_nchw.shape
    call("cmu_nchw_to_nhwc", _p(_f32c(x_nchw)), out.ptr(), out.ld, B, H, W, C, out.dt, _stream())


# ------------------------------------------------------------------------------------------------
# backward
# ------------------------------------------------------------------------------------------------
def bn_bwd_reduce(dA, y, save_mean, save_invstd, dgamma, dbeta, coef, ws):
    call("cmu_bn_bwd_reduce", dA.ptr(), dA.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(save_mean), _p(save_invstd),
         _p(dgamma), _p(dbeta), _p(coef), y.B, y.H, y.W, y.C, y.dt, _p(ws), _stream())


def bn_bwd_apply(dA, y, save_mean, save_invstd, coef, dY):
    call("cmu_bn_bwd_apply", dA.ptr(), dA.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(save_mean), _p(save_invstd),
         _p(coef), dY.ptr(), dY.ld, y.B, y.H, y.W, y.C, y.dt, _stream())


def conv3x3_wgrad(x, dY, dW, ws):
    Cout, Cin = dW.shape[0], dW.shape[1]
    assert x.C == Cin and dY.C == Cout
    call("cmu_conv3x3_wgrad", x.ptr(), x.ld, _p(x.scale), _p(x.shift), x.relu_from, dY.ptr(), dY.ld, _p(_f32c(dW)),
         x.B, x.H, x.W, Cin, Cout, x.dt, _p(ws), _stream(), work=2.0 * 9 * Cin * Cout * x.B * x.H * x.W)


def conv3x3_c1_wgrad(x_bhw, dY, dW, ws, mask=None, mask_per_sample=False):
    B, H, W = x_bhw.shape
    call("cmu_conv3x3_c1_wgrad", _p(_f32c(x_bhw)), _p(mask), int(mask_per_sample), dY.ptr(), dY.ld, _p(_f32c(dW)),
         B, H, W, dW.shape[0], dY.dt, _p(ws), _stream())


def conv3x3_c1_wgrad_bn(x_bhw, dA, yraw, scale, shift, save_mean, save_invstd, coef, dW, ws, mask=None, mask_per_sample=False, w=None):
    """First-layer weight gradient with the layer's BatchNorm+ReLU backward applied in registers (no dY in memory).  ``w``: the
    layer's forward weights -- the raw output is then recomputed from the image instead of read from ``yraw`` (same bits)."""
    B, H, W = x_bhw.shape
    assert dA.dt == yraw.dt
    if w is not None:
        call("cmu_conv3x3_c1_wgrad_bn_w", _p(_f32c(x_bhw)), _p(mask), int(mask_per_sample), dA.ptr(), dA.ld, _p(_f32c(w)), _p(scale), _p(shift),
             _p(save_mean), _p(save_invstd), _p(coef), _p(_f32c(dW)), B, H, W, dW.shape[0], dA.dt, _p(ws), _stream())
        return
    call("cmu_conv3x3_c1_wgrad_bn", _p(_f32c(x_bhw)), _p(mask), int(mask_per_sample), dA.ptr(), dA.ld, yraw.ptr(), yraw.ld,
         _p(scale), _p(shift), _p(save_mean), _p(save_invstd), _p(coef), _p(_f32c(dW)), B, H, W, dW.shape[0], dA.dt, _p(ws),
         _stream())


def maxpool_bwd(dP, dSkip, y, dA, save_mean=None, save_invstd=None, bn_ws=None, dSkip2=None):
    """``dSkip2``: a second gradient of the same skip tensor (two decoders on one encoder), summed inside the pass.
    ``dA`` None (with ``bn_ws``): only the BatchNorm-backward sums are produced -- follow with ``maxpool_bwd_apply``."""
    call("cmu_maxpool_bwd2", dP.ptr(), dP.ld, None if dSkip is None else dSkip.ptr(), 0 if dSkip is None else dSkip.ld,
         None if dSkip2 is None else dSkip2.ptr(), 0 if dSkip2 is None else dSkip2.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift),
         None if dA is None else dA.ptr(), 0 if dA is None else dA.ld, _p(save_mean), _p(save_invstd), _p(bn_ws),
         y.B, y.H, y.W, y.C, y.dt, _stream())


def maxpool_bwd_apply(dP, dSkip, y, save_mean, save_invstd, coef, dY, dSkip2=None):
    """dY of the conv+BN+ReLU layer in front of a max-pool, straight from the pooled gradient and the skip gradient(s): the second
    half of ``maxpool_bwd(..., dA=None, bn_ws=...)`` + ``bn_bwd_finalize`` (the pool's input gradient is never stored)."""
    call("cmu_maxpool_bwd_apply", dP.ptr(), dP.ld, None if dSkip is None else dSkip.ptr(), 0 if dSkip is None else dSkip.ld,
         None if dSkip2 is None else dSkip2.ptr(), 0 if dSkip2 is None else dSkip2.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift),
         _p(save_mean), _p(save_invstd), _p(_f32c(coef)), dY.ptr(), dY.ld, y.B, y.H, y.W, y.C, y.dt, _stream())


def bn_bwd_finalize(bn_ws, count, dgamma, dbeta, coef):
    call("cmu_bn_bwd_finalize", _p(bn_ws), int(count), _p(dgamma), _p(dbeta), _p(coef), coef.shape[1], _stream())


def bn_bwd_finalize_tiles(bstats, count, dgamma, dbeta, coef, ws):
    """Phase 1 of BatchNorm backward from the per-tile slab written by conv3x3_dgrad_bn / convT2x2_dgrad_bn."""
    call("cmu_bn_bwd_finalize_tiles", _p(bstats), bstats.shape[0], int(count), _p(dgamma), _p(dbeta), _p(coef), coef.shape[1], _p(ws),
         _stream())


def conv3x3_dgrad_bn(dY, wpacked_flip, dX, y, save_mean, save_invstd, bstats):
    """dX = data gradient of a 3x3 conv; ``y``: Act of the raw output (+ pending BN transform) of the layer whose activated
    output is the conv's input -- its BatchNorm+ReLU backward partial sums go to ``bstats`` (new_stats shape)."""
    call("cmu_conv3x3_dgrad_bn", dY.ptr(), dY.ld, _p(wpacked_flip), dX.ptr(), dX.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift),
         _p(save_mean), _p(save_invstd), _p(bstats), dX.B, dX.H, dX.W, dY.C, dX.C, dX.dt, _stream(),
         work=2.0 * 9 * dY.C * dX.C * dX.B * dX.H * dX.W)


def convT2x2_dgrad_bn(dOut, wpacked_dgrad, dX, y, save_mean, save_invstd, bstats):
    call("cmu_convT2x2_dgrad_bn", dOut.ptr(), dOut.ld, _p(wpacked_dgrad), dX.ptr(), dX.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift),
         _p(save_mean), _p(save_invstd), _p(bstats), dX.B, dX.H, dX.W, dX.C, dOut.C, dX.dt, _stream(),
         work=2.0 * 4 * dX.C * dOut.C * dX.B * dX.H * dX.W)


def convT2x2_dgrad(dOut, wpacked_dgrad, dX):
    call("cmu_convT2x2_dgrad", dOut.ptr(), dOut.ld, _p(wpacked_dgrad), dX.ptr(), dX.ld, dX.B, dX.H, dX.W, dX.C, dOut.C,
         dX.dt, _stream(), work=2.0 * 4 * dX.C * dOut.C * dX.B * dX.H * dX.W)


def convT2x2_wgrad(x, dOut, dW, dbias, ws):
    Cin, Cout = dW.shape[0], dW.shape[1]
    call("cmu_convT2x2_wgrad", x.ptr(), x.ld, _p(x.scale), _p(x.shift), x.relu_from, dOut.ptr(), dOut.ld, _p(_f32c(dW)),
         _p(_f32c(dbias)), x.B, x.H, x.W, Cin, Cout, x.dt, _p(ws), _stream(), work=2.0 * 4 * Cin * Cout * x.B * x.H * x.W)


def conv1x1_head_bn_apply(dlogits, x, w, save_mean, save_invstd, coef, dY):
    """dY of the conv+BN+ReLU layer that fed the 1x1 head, straight from dlogits (the head's input gradient has rank K and is never
    stored): second half of ``conv1x1_head_bwd(..., dX=None, bn_ws=...)`` + ``bn_bwd_finalize``."""
    K = w.shape[0]
    call("cmu_conv1x1_head_bn_apply", _p(_f32c(dlogits)), x.ptr(), x.ld, _p(x.scale), _p(x.shift), _p(_f32c(w)), _p(save_mean),
         _p(save_invstd), _p(_f32c(coef)), dY.ptr(), dY.ld, x.B, x.H, x.W, x.C, K, x.dt, _stream())


def conv1x1_head_bwd(dlogits, x, w, dX, dW, dbias, ws, save_mean=None, save_invstd=None, bn_ws=None):
    """``dX`` None (with ``bn_ws``): parameter gradients and BatchNorm-backward sums only (see ``conv1x1_head_bn_apply``)."""
    K = w.shape[0]
    if dX is None:
        call("cmu_conv1x1_head_bwd", _p(_f32c(dlogits)), x.ptr(), x.ld, _p(x.scale), _p(x.shift), _p(_f32c(w)), None, 0,
             _p(_f32c(dW)), _p(_f32c(dbias)), _p(save_mean), _p(save_invstd), _p(bn_ws), x.B, x.H, x.W, x.C, K, x.dt, _p(ws),
             _stream())
        return
    call("cmu_conv1x1_head_bwd", _p(_f32c(dlogits)), x.ptr(), x.ld, _p(x.scale), _p(x.shift), _p(_f32c(w)), dX.ptr(), dX.ld,
         _p(_f32c(dW)), _p(_f32c(dbias)), _p(save_mean), _p(save_invstd), _p(bn_ws), x.B, x.H, x.W, x.C, K, x.dt, _p(ws),
         _stream())


# ------------------------------------------------------------------------------------------------
# losses / heads / optimiser
# ------------------------------------------------------------------------------------------------
def masked_mse_fwd_bwd(logits, channel, img, mask, loss, dlogits, loss_scale, ws, amp=None):
    B, K, H, W = logits.shape
    call("cmu_masked_mse_fwd_bwd", _p(_f32c(logits)), K, channel, _p(_f32c(img)), _p(mask), _p(loss), _p(dlogits),
         float(loss_scale), _p(None if amp is None else amp.state), B, H, W, _p(ws), _stream())


def softmax_ce_dice_fwd_bwd(logits, y1h, out, dlogits, loss_scale, ws):
    B, K, H, W = logits.shape
    assert K == 2 and y1h.dtype == torch.float64 and y1h.is_contiguous()
    call("cmu_softmax_ce_dice_fwd_bwd", _p(_f32c(logits)), _p(y1h), _p(out), _p(dlogits), float(loss_scale), B, H, W,
         _p(ws), _stream())


def infonce_inbatch_fwd_bwd(pred, keys, loss, dpred, rank, temperature, ct_weight):
    B, D = pred.shape
    call("cmu_infonce_inbatch_fwd_bwd", _p(_f32c(pred)), _p(_f32c(keys)), _p(loss), _p(dpred), B, keys.shape[0], D,
         rank, float(temperature), float(ct_weight), _stream())


def moco_infonce_enqueue(q_raw, k_raw, keys_all, queue, queue_ptr, loss, dq, k_norm_out, temperature, ws):
    B, D = q_raw.shape
    K = queue.shape[1]
    Nk = B if keys_all is None else keys_all.shape[0]
    call("cmu_moco_infonce_enqueue", _p(_f32c(q_raw)), _p(_f32c(k_raw)), _p(keys_all), Nk, _p(_f32c(queue)),
         _p(queue_ptr), _p(loss), _p(dq), _p(k_norm_out), B, D, K, float(temperature), _p(ws), _stream())


def l2_normalize_rows(x, out):
    call("cmu_l2_normalize_rows", _p(_f32c(x)), _p(out), x.shape[0], x.shape[1], _stream())


def l2_normalize_rows_bwd(x, dy, dx):
    call("cmu_l2_normalize_rows_bwd", _p(_f32c(x)), _p(_f32c(dy)), _p(dx), x.shape[0], x.shape[1], _stream())


def moco_logits_assemble(q, k, lneg, logits, inv_t):
    B, D = q.shape
    call("cmu_moco_logits_assemble", _p(_f32c(q)), _p(_f32c(k)), _p(_f32c(lneg)), _p(logits), B, D, lneg.shape[1], float(inv_t), _stream())


def moco_logits_split(dlogits, dlneg, inv_t):
    call("cmu_moco_logits_split", _p(_f32c(dlogits)), _p(dlneg), dlneg.shape[0], dlneg.shape[1], float(inv_t), _stream())


def moco_logits_addpos(dlogits, k, dq, inv_t):
    B, D = dq.shape
    call("cmu_moco_logits_addpos", _p(_f32c(dlogits)), _p(_f32c(k)), _p(dq), B, D, dlogits.shape[1] - 1, float(inv_t), _stream())


def row_cross_entropy(logits, target, want_grad=True, want_rank=False):
    """F.cross_entropy(logits, target) (mean) on one kernel: -> (loss (1,), dlogits or None = d mean / d logits, rank (B,) int32 or None)."""
    B, N = logits.shape
    dev = logits.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    rows = torch.empty(B, dtype=torch.float32, device=dev)
    dl = torch.empty((B, N), dtype=torch.float32, device=dev) if want_grad else None
    rank = torch.empty(B, dtype=torch.int32, device=dev) if want_rank else None
    tgt = target if (target.dtype == torch.int64 and target.is_contiguous()) else target.to(torch.int64).contiguous()
    call("cmu_row_cross_entropy", _p(_f32c(logits)), _p(tgt), _p(loss), _p(rows), _p(dl), _p(rank), B, N, _stream())
    return loss, dl, rank


def patchify(x, h, w, p, inverse=False):
    """SparK.patchify (inverse False: (B, C, h*p, w*p) -> (B, h*w, p*p*C)) / unpatchify (inverse True) on cmu_patchify, fp32.
    Contract: CUDA tensors, no autograd (the reference's einsum form is differentiable; here it serves the `vis` path only).
    A shape that does not match (h, w, p) raises, as the reference's reshape does (spark.py:133-149)."""
    if not x.is_cuda:
        raise RuntimeError("ops.patchify: CUDA tensor required (no CPU fallback)")
    if not inverse:
        if x.dim() != 4 or x.shape[2] != h * p or x.shape[3] != w * p:
            raise ValueError(f"patchify: input {tuple(x.shape)} is not (B, C, {h * p}, {w * p}) for h={h}, w={w}, p={p}")
    else:
        if x.dim() != 3 or x.shape[1] != h * w or x.shape[2] % (p * p) != 0 or x.shape[2] == 0:
            raise ValueError(f"unpatchify: input {tuple(x.shape)} is not (B, {h * w}, {p * p}*C) for h={h}, w={w}, p={p}")
    x = x.float().contiguous()
    if not inverse:
        B, C = x.shape[:2]
        out = torch.empty((B, h * w, p * p * C), dtype=torch.float32, device=x.device)
    else:
        B, C = x.shape[0], x.shape[-1] // (p * p)
        out = torch.empty((B, C, h * p, w * p), dtype=torch.float32, device=x.device)
    call("cmu_patchify", _p(x), _p(out), B, C, h, w, p, int(inverse), _stream())
    return out


def scale_by_device_scalar(v, s):
    call("cmu_scale_by_device_scalar", _p(v), _p(_f32c(s.reshape(-1))), v.numel(), _stream())


# ------------------------------------------------------------------------------------------------
# SparK sparse ops
# ------------------------------------------------------------------------------------------------
def masked_channel_stats(x, active, invert=False):
    """-> slab [rows][2][C] fp32 (rows = cmu_masked_stats_rows()) of sums over the selected pixels."""
    rows = _lib.lib().cmu_masked_stats_rows()
    slab = torch.empty((rows, 2, x.C), dtype=torch.float32, device=x.buf.device)
    call("cmu_masked_channel_stats", x.ptr(), x.ld, _p(active), active.shape[-1], int(invert), _p(slab), x.B, x.H, x.W, x.C, x.dt,
         _stream())
    return slab


def rows_channel_stats(x, pixels):
    """masked_channel_stats over a PixelList: the listed pixels only."""
    rows = _lib.lib().cmu_masked_stats_rows()
    slab = torch.empty((rows, 2, x.C), dtype=torch.float32, device=x.buf.device)
    call("cmu_rows_channel_stats", x.ptr(), x.ld, _p(pixels.rows), _p(pixels.count), _p(slab), x.B, x.H, x.W, x.C, x.dt, _stream())
    return slab


def bn_bwd_reduce_rows(dA, y, save_mean, save_invstd, dgamma, dbeta, coef, pixels, count, ws):
    call("cmu_bn_bwd_reduce_rows", dA.ptr(), dA.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(save_mean), _p(save_invstd), _p(dgamma),
         _p(dbeta), _p(coef), _p(pixels.rows), _p(pixels.count), pixels.capacity, int(count), y.B, y.H, y.W, y.C, y.dt, _p(ws), _stream())


_SPARK_CELLS = os.environ.get("CMU_SPARK_CELLS", "1") != "0"      # A/B: the pixel-organised masked kernels


def cells_supported(x, active):
    """Whether the patch-organised element-wise kernels (csrc/sparse_elem.hip) serve this level."""
    if not _SPARK_CELLS:
        return False
    return bool(_lib.lib().cmu_cells_supported(x.B, x.H, x.W, active.shape[-1], x.C, x.dt))


def mask_select(x, active, out, relu=False, invert=False, fill=None, use_transform=True, ring=False, cells=True):
    """out = selected ? relu?(x * scale + shift) : fill.  Without inversion the patch-organised kernel is taken
    (``cells``); ``ring``: only the one-pixel border frame of each masked patch is zeroed -- the caller guarantees that every consumer
    of ``out`` is list-driven (reads active patches and a one-pixel halo only)."""
    sc = x.scale if use_transform else None
    sh = x.shift if use_transform else None
    if cells and not invert and cells_supported(x, active):
        call("cmu_mask_select_cells", x.ptr(), x.ld, _p(sc), _p(sh), int(relu), _p(active), active.shape[-1], int(bool(ring) and fill is None),
             _p(fill), out.ptr(), out.ld, x.B, x.H, x.W, x.C, x.dt, _stream())
        return
    call("cmu_mask_select", x.ptr(), x.ld, _p(sc), _p(sh), int(relu), _p(active), active.shape[-1], int(invert), _p(fill), out.ptr(),
         out.ld, x.B, x.H, x.W, x.C, x.dt, _stream())


def cells_channel_stats(x, active):
    """masked_channel_stats / rows_channel_stats over the active patches, patch-organised -> slab [rows][2][C] fp32."""
    slab = torch.empty((_lib.lib().cmu_cells_stats_rows(), 2, x.C), dtype=torch.float32, device=x.buf.device)
    call("cmu_cells_channel_stats", x.ptr(), x.ld, _p(active), active.shape[-1], _p(slab), x.B, x.H, x.W, x.C, x.dt, _stream())
    return slab


def bn_bwd_reduce_cells(dA, y, save_mean, save_invstd, dgamma, dbeta, coef, active, count, ws):
    """bn_bwd_reduce_masked / bn_bwd_reduce_rows, patch-organised."""
    call("cmu_bn_bwd_reduce_cells", dA.ptr(), dA.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(save_mean), _p(save_invstd), _p(dgamma),
         _p(dbeta), _p(coef), _p(active), active.shape[-1], int(count), y.B, y.H, y.W, y.C, y.dt, _p(ws), _stream())


def cells_channel_sum(x, active, out, invert=False, ws=None):
    """out[c] (fp32, C) = sum of x over the pixels of the active / masked (``invert``) patches: the mask-token gradient."""
    assert out.dtype == torch.float32 and out.numel() == x.C and out.is_contiguous()
    if ws is None:
        ws = torch.empty(_lib.lib().cmu_cells_channel_sum_ws_bytes(x.C), dtype=torch.uint8, device=x.buf.device)
    call("cmu_cells_channel_sum", x.ptr(), x.ld, _p(active), active.shape[-1], int(invert), _p(out), _p(ws), x.B, x.H, x.W, x.C, x.dt,
         _stream())


def bnrelu_maxpool_fwd_masked(y, active, out):
    """BN + ReLU + 2x2 max of the raw ``y`` with masked windows pooled to zero (no activated / masked copy of y in memory)."""
    assert y.scale is not None
    call("cmu_bnrelu_maxpool_fwd_masked", y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(active), active.shape[-1], out.ptr(), out.ld,
         y.B, y.H, y.W, y.C, y.dt, _stream())


def maxpool_bwd_masked(dP, dSkip, y, dA, active, cells=True):
    """``maxpool_bwd`` on the raw ``y`` + transform at active windows only; dA of masked windows is left unwritten."""
    name = "cmu_maxpool_bwd_cells" if (cells and y.H // active.shape[-1] >= 2 and cells_supported(y, active)) else "cmu_maxpool_bwd_masked"
    call(name, dP.ptr(), dP.ld, None if dSkip is None else dSkip.ptr(), 0 if dSkip is None else dSkip.ld, y.ptr(), y.ld,
         _p(y.scale), _p(y.shift), _p(active), active.shape[-1], dA.ptr(), dA.ld, y.B, y.H, y.W, y.C, y.dt, _stream())


def bn_bwd_reduce_masked(dA, y, save_mean, save_invstd, dgamma, dbeta, coef, active, count, ws):
    call("cmu_bn_bwd_reduce_masked", dA.ptr(), dA.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(save_mean), _p(save_invstd),
         _p(dgamma), _p(dbeta), _p(coef), _p(active), active.shape[-1], int(count), y.B, y.H, y.W, y.C, y.dt, _p(ws), _stream())


def bn_bwd_apply_masked(dA, y, save_mean, save_invstd, coef, dY, active, ring=False, cells=True):
    """dY = BatchNorm+ReLU backward at active positions, zeros at masked ones (``ring``: see mask_select)."""
    if cells and cells_supported(y, active):
        call("cmu_bn_bwd_apply_cells", dA.ptr(), dA.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(save_mean), _p(save_invstd),
             _p(coef), dY.ptr(), dY.ld, _p(active), active.shape[-1], int(bool(ring)), y.B, y.H, y.W, y.C, y.dt, _stream())
        return
    call("cmu_bn_bwd_apply_masked", dA.ptr(), dA.ld, y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(save_mean), _p(save_invstd),
         _p(coef), dY.ptr(), dY.ld, _p(active), active.shape[-1], y.B, y.H, y.W, y.C, y.dt, _stream())


class TileList:
    """Device-side list of the spatial tiles of one level that overlap an active patch (cmu_sparse_tile_list): ``list`` int32,
    ``count`` int32 (1,) -- both stay on the device.  ``n_dense`` is the dense tile count (the list's capacity).  ``defer``: only
    allocate; ``build_lists`` fills several lists in one launch."""

    def __init__(self, active, H, W, tile_h, tile_w, defer=False):
        B, f = active.shape[0], active.shape[-1]
        self.H, self.W, self.tile_h, self.tile_w = H, W, tile_h, tile_w
        self.n_dense = B * ((H + tile_h - 1) // tile_h) * ((W + tile_w - 1) // tile_w)
        self.list = torch.empty(self.n_dense, dtype=torch.int32, device=active.device)
        self.count = torch.empty(1, dtype=torch.int32, device=active.device)
        if not defer:
            call("cmu_sparse_tile_list", _p(active), f, B, H, W, tile_h, tile_w, _p(self.list), _p(self.count), _stream())


class PixelList:
    """Device-side list of the active pixels of one level (cmu_sparse_pixel_list): ``rows`` int32 dense pixel indices, patch-
    major, padded with -1 to ``capacity``; ``count`` int32 (1,).  ``max_rows``: host-side upper bound of the count (the number
    of active patches x patch area when the caller knows it, else the dense pixel count).  ``defer``: see TileList."""

    def __init__(self, active, H, W, max_rows=None, defer=False):
        B, f = active.shape[0], active.shape[-1]
        dense = B * H * W
        self.H, self.W = H, W
        self.max_rows = dense if max_rows is None else min(int(max_rows), dense)
        self.capacity = max(256, (self.max_rows + 255) // 256 * 256)
        self.rows = torch.empty(self.capacity, dtype=torch.int32, device=active.device)
        self.count = torch.empty(1, dtype=torch.int32, device=active.device)
        if not defer:
            ws = torch.empty(_lib.lib().cmu_sparse_pixel_list_ws_bytes(B, f), dtype=torch.uint8, device=active.device)
            call("cmu_sparse_pixel_list", _p(active), f, B, H, W, _p(self.rows), self.capacity, _p(self.count), _p(ws), _stream())


def build_lists(active, tile_lists=(), pixel_lists=()):
    """Fill deferred TileLists / PixelLists of ONE patch map: every tile list (plus the patch list the pixel lists expand) in one launch
    of one workgroup per list, every pixel list in a second launch -- instead of one or two single-workgroup launches per list."""
    B, f = active.shape[0], active.shape[-1]
    lib = _lib.lib()
    tls = list(tile_lists)
    cells = None
    if pixel_lists:
        assert all(pl.H == pl.W for pl in pixel_lists)
        cells = TileList(active, f, f, 1, 1, defer=True)            # tiles of one patch: entries (b*f + fy)*f + fx
        tls = tls + [cells]
    mx = lib.cmu_sparse_tile_lists_max()
    for i in range(0, len(tls), mx):
        part = tls[i:i + mx]
        n = len(part)
        IA, PA = ctypes.c_int * n, ctypes.c_void_p * n
        call("cmu_sparse_tile_lists", _p(active), f, B, n, IA(*[t.H for t in part]), IA(*[t.tile_h for t in part]),
             IA(*[t.tile_w for t in part]), PA(*[t.list.data_ptr() for t in part]), PA(*[t.count.data_ptr() for t in part]), _stream())
    pls = list(pixel_lists)
    for i in range(0, len(pls), mx):
        part = pls[i:i + mx]
        n = len(part)
        IA, PA, LA = ctypes.c_int * n, ctypes.c_void_p * n, ctypes.c_int64 * n
        call("cmu_sparse_pixel_lists", _p(cells.list), _p(cells.count), f, B, n, IA(*[pl.H for pl in part]),
             PA(*[pl.rows.data_ptr() for pl in part]), LA(*[pl.capacity for pl in part]), PA(*[pl.count.data_ptr() for pl in part]), _stream())
    return cells


def conv3x3_rows_supported(B, H, W, Cin, Cout, dt):
    return bool(_lib.lib().cmu_conv3x3_rows_supported(B, H, W, Cin, Cout, dt_code(dt)))


def conv3x3_fwd_rows(x, wpacked, out, pixels):
    """3x3 convolution at the listed pixels only (gather-GEMM, csrc/conv_gather.inc); ``out`` elsewhere untouched.  ``x`` carries no
    pending transform (the sparse encoder's inputs are already activated and masked)."""
    assert x.dt == out.dt and (x.B, x.H, x.W) == (out.B, out.H, out.W) and x.scale is None
    call("cmu_conv3x3_fwd_rows", x.ptr(), x.ld, _p(wpacked), out.ptr(), out.ld, _p(pixels.rows), _p(pixels.count), pixels.capacity,
         x.B, x.H, x.W, x.C, out.C, x.dt, _stream(), work=2.0 * 9 * x.C * out.C * pixels.max_rows)


def conv3x3_tiles_supported(B, H, W, Cin, Cout, dt):
    return bool(_lib.lib().cmu_conv3x3_tiles_supported(B, H, W, Cin, Cout, dt_code(dt)))


def conv3x3_fwd_tiles(x, wpacked, out, tiles, active_fraction=1.0):
    """conv3x3_fwd over the listed 16 x 32 tiles only (``out`` elsewhere untouched).  ``active_fraction``: share of listed
    tiles, for the profiler's algorithmic FLOP count only."""
    assert x.dt == out.dt and (x.B, x.H, x.W) == (out.B, out.H, out.W) and (tiles.tile_h, tiles.tile_w) == (16, 32)
    call("cmu_conv3x3_fwd_tiles", x.ptr(), x.ld, _p(x.scale), _p(x.shift), x.relu_from, _p(wpacked), out.ptr(), out.ld,
         _p(tiles.list), _p(tiles.count), x.B, x.H, x.W, x.C, out.C, x.dt, _stream(),
         work=2.0 * 9 * x.C * out.C * x.B * x.H * x.W * active_fraction)


def conv3x3_wgrad_tile_h(B, H, W, Cin, Cout, dt):
    """Tile height of the list ``conv3x3_wgrad_tiles`` wants for this shape: 8 (8 x 16 tiles, wide kernel) or 16 (16 x 16 tiles)."""
    return int(_lib.lib().cmu_conv3x3_wgrad_tile_h(B, H, W, Cin, Cout, dt_code(dt)))


def conv3x3_wgrad_tiles(x, dY, dW, ws, tiles, active_fraction=1.0):
    """Weight gradient with the contraction restricted to the listed tiles (``dY`` is zero elsewhere): 16 x 16 tiles on the first
    kernel, 8 x 16 tiles on the wide kernel (``conv3x3_wgrad_tile_h`` says which list a shape wants)."""
    Cout, Cin = dW.shape[0], dW.shape[1]
    assert x.C == Cin and dY.C == Cout and tiles.tile_w == 16 and tiles.tile_h in (8, 16)
    call("cmu_conv3x3_wgrad_tiles", x.ptr(), x.ld, _p(x.scale), _p(x.shift), x.relu_from, dY.ptr(), dY.ld, _p(_f32c(dW)),
         _p(tiles.list), _p(tiles.count), tiles.tile_h, x.B, x.H, x.W, Cin, Cout, x.dt, _p(ws), _stream(),
         work=2.0 * 9 * Cin * Cout * x.B * x.H * x.W * active_fraction)


def spark_loss_fwd_bwd(rec, img, active, loss, drec, loss_scale, p, ws):
    B, f = active.shape[0], active.shape[-1]
    call("cmu_spark_loss_fwd_bwd", _p(_f32c(rec)), _p(_f32c(img)), _p(active), _p(loss), _p(drec), float(loss_scale), B, f, p, _p(ws),
         _stream())


def gap_fwd(y, out):
    call("cmu_gap_fwd", y.ptr(), y.ld, _p(y.scale), _p(y.shift), _p(_f32c(out)), y.B, y.H, y.W, y.C, y.dt, _stream())


def gap_bwd(dout, dA):
    call("cmu_gap_bwd", _p(_f32c(dout)), dA.ptr(), dA.ld, dA.B, dA.H, dA.W, dA.C, dA.dt, _stream())


# bumped whenever a raw kernel rewrites parameters in place (PyTorch's version counters do not see it);
# the engine's packed-weight caches compare it together with Tensor._version
PARAM_GENERATION = 0


def bump_param_generation():
    """Call after any raw or bulk write into a parameter arena that PyTorch's version counters do not see (a collective into
    the arena, a checkpoint copied into it): the engine's packed-weight caches key on this counter."""
    global PARAM_GENERATION
    PARAM_GENERATION += 1


def ema_update(target, online, momentum):
    global PARAM_GENERATION
    PARAM_GENERATION += 1
    assert target.numel() == online.numel()
    call("cmu_ema_update", _p(_f32c(target)), _p(_f32c(online)), target.numel(), float(momentum), _stream())


def adam_step(p, g, m, v, wd_mask, lr, beta1, beta2, eps, weight_decay, decoupled, step, grad_scale=1.0, amp=None):
    global PARAM_GENERATION
    PARAM_GENERATION += 1
    call("cmu_adam_step", _p(_f32c(p)), _p(_f32c(g)), _p(_f32c(m)), _p(_f32c(v)), _p(wd_mask), p.numel(), float(lr),
         float(beta1), float(beta2), float(eps), float(weight_decay), int(decoupled), int(step), float(grad_scale),
         _p(None if amp is None else amp.state), _stream())


def adam_ema_step(p, g, m, v, wd_mask, lr, beta1, beta2, eps, weight_decay, decoupled, step, grad_scale, amp, segments, momentum):
    """``adam_step`` + the EMA of the momentum networks in one pass (cmu_adam_ema_step).  ``segments``: up to two
    (lo, hi, target fp32 tensor of hi - lo elements), ascending element ranges of the arena ``p``."""
    global PARAM_GENERATION
    PARAM_GENERATION += 1
    n = len(segments)
    assert 1 <= n <= 2
    lo = (ctypes.c_int64 * n)(*[int(s[0]) for s in segments])
    hi = (ctypes.c_int64 * n)(*[int(s[1]) for s in segments])
    for s in segments:
        assert s[2].numel() == s[1] - s[0] and s[2].device == p.device
    tg = (ctypes.c_void_p * n)(*[_f32c(s[2]).data_ptr() for s in segments])
    call("cmu_adam_ema_step", _p(_f32c(p)), _p(_f32c(g)), _p(_f32c(m)), _p(_f32c(v)), _p(wd_mask), p.numel(), float(lr),
         float(beta1), float(beta2), float(eps), float(weight_decay), int(decoupled), int(step), float(grad_scale),
         _p(None if amp is None else amp.state), n, lo, hi, tg, float(momentum), _stream())


class AmpScaler:
    """Dynamic loss scaler with its state on the device (cmu_amp_*): torch.cuda.amp.GradScaler's protocol -- what mmengine's
    AmpOptimWrapper(loss_scale='dynamic') wraps (cmunet_config.py:76-78) -- without the per-step host read of found_inf."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.state = torch.zeros(_lib.lib().cmu_amp_state_bytes(), dtype=torch.uint8, device=device)
        call("cmu_amp_init", _p(self.state), float(init_scale), _stream())

    def check(self, grad_arena):
        call("cmu_amp_check_finite", _p(_f32c(grad_arena)), grad_arena.numel(), _p(self.state), _stream())

    def update(self):
        call("cmu_amp_update", _p(self.state), float(self.growth_factor), float(self.backoff_factor), int(self.growth_interval), _stream())

    def read(self):
        """(scale, found_inf, growth_tracker, good_steps, skipped_steps) -- synchronises; for logging and tests."""
        raw = self.state.cpu()
        f = raw[:8].view(torch.float32)
        i = raw[8:20].view(torch.int32)
        return float(f[0]), float(f[1]), int(i[0]), int(i[1]), int(i[2])

    def state_dict(self):
        """GradScaler.state_dict's fields (mmengine's AmpOptimWrapper saves them as ``loss_scaler``) plus the two counters the
        optimiser kernel reads: ``good_steps`` is the step number of Adam's bias corrections, so a resume without it would
        restart them at 1 on steady-state moments (first updates ~2x too large at betas (0.9, 0.95)).  Synchronises."""
        scale, _, tracker, good, skipped = self.read()
        return {"scale": scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": tracker, "good_steps": good, "skipped_steps": skipped}

    def load_state_dict(self, sd):
        import struct
        self.growth_factor = sd.get("growth_factor", self.growth_factor)
        self.backoff_factor = sd.get("backoff_factor", self.backoff_factor)
        self.growth_interval = sd.get("growth_interval", self.growth_interval)
        raw = struct.pack("<ffiii", float(sd["scale"]), 0.0, int(sd.get("_growth_tracker", 0)), int(sd.get("good_steps", 0)),
                          int(sd.get("skipped_steps", 0)))
        raw = raw + b"\0" * (self.state.numel() - len(raw))
        self.state.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))


def sgd_step(p, g, buf, wd_mask, lr, momentum, dampening, weight_decay, nesterov, step, grad_scale=1.0):
    global PARAM_GENERATION
    PARAM_GENERATION += 1
    call("cmu_sgd_step", _p(_f32c(p)), _p(_f32c(g)), _p(buf), _p(wd_mask), p.numel(), float(lr), float(momentum), float(dampening),
         float(weight_decay), int(nesterov), int(step), float(grad_scale), _stream())


def lamb_step(p, g, m, v, u, tables, lr, beta1, beta2, eps, bias_correction, grad_averaging, max_grad_norm, trust_clip,
              always_adapt, step, grad_scale, ws):
    """``tables`` = (blk_start int64, blk_count int32, blk_tensor int32, t_blk0 int32 [T+1], t_wd float32 [T]) on the device."""
    global PARAM_GENERATION
    PARAM_GENERATION += 1
    bs, bc, bt, t0, twd = tables
    call("cmu_lamb_step", _p(_f32c(p)), _p(_f32c(g)), _p(m), _p(v), _p(u), _p(bs), _p(bc), _p(bt), bs.numel(), _p(t0), _p(twd),
         twd.numel(), float(lr), float(beta1), float(beta2), float(eps), int(bias_correction), int(grad_averaging),
         float(max_grad_norm), int(trust_clip), int(always_adapt), int(step), float(grad_scale), _p(ws), _stream())


# ---- input pipeline on the device ---------------------------------------------------------------------
def resize_bicubic(src, out_h, out_w, boxes=None, flip=None):
    """Pillow-convention bicubic resize of per-sample crop windows (+ optional horizontal flip): src (B,H,W) f32 (PIL mode 'F') or
    uint8 (mode 'L': Pillow's 22-bit fixed-point path) cuda, boxes (B,4) int32 (x0, y0, w, h) or None, flip (B,) uint8/bool or
    None -> (B,out_h,out_w) of src's dtype."""
    assert src.dtype in (torch.float32, torch.uint8) and src.dim() == 3 and src.is_contiguous()
    u8 = src.dtype == torch.uint8
    B, H, W = src.shape
    if boxes is not None:
        bh = boxes.detach().cpu().to(torch.int64)
        ok = (bh[:, 0] >= 0) & (bh[:, 1] >= 0) & (bh[:, 2] >= 1) & (bh[:, 3] >= 1) & (bh[:, 0] + bh[:, 2] <= W) & (bh[:, 1] + bh[:, 3] <= H)
        if bh.shape != (B, 4) or not bool(ok.all()):
            raise ValueError("resize_bicubic: crop windows must be (B,4) (x0, y0, w, h) inside the image")
        boxes = boxes.to(device=src.device, dtype=torch.int32).contiguous()
    if flip is not None:
        flip = flip.to(device=src.device, dtype=torch.uint8).contiguous()
    out = torch.empty(B, out_h, out_w, dtype=src.dtype, device=src.device)
    entry = "cmu_resize_bicubic_u8" if u8 else "cmu_resize_bicubic"
    ws = torch.empty(getattr(_lib.lib(), entry + "_ws_bytes")(B, H, W, out_h, out_w), dtype=torch.uint8, device=src.device)
    call(entry, _p(src), B, H, W, _p(boxes), _p(flip), _p(out), out_h, out_w, _p(ws), _stream())
    return out


def resize_nearest(src, out_h, out_w):
    """``Image.resize(size, NEAREST)`` of a batch of uint8 label masks (Finetuning/dataset.py:47): src (B,H,W) uint8 cuda ->
    (B,out_h,out_w) uint8."""
    assert src.dtype == torch.uint8 and src.dim() == 3 and src.is_contiguous()
    B, H, W = src.shape
    out = torch.empty(B, out_h, out_w, dtype=torch.uint8, device=src.device)
    ws = torch.empty(_lib.lib().cmu_resize_nearest_u8_ws_bytes(out_h, out_w), dtype=torch.uint8, device=src.device)
    call("cmu_resize_nearest_u8", _p(src), B, H, W, _p(out), out_h, out_w, _p(ws), _stream())
    return out


def two_view(src, shifts, out=224, noise=None, seed=0):
    """ShiftPixel crops + GaussNoise: src (B,S,S) f32 cuda, shifts (B,2) int (dy, dx) -> (img, img_t) (B,out,out) f32.
    ``noise``: (B,out,out) float64 standard-normal draws, or None for the in-kernel Philox generator keyed by ``seed``."""
    assert src.dtype == torch.float32 and src.dim() == 3 and src.shape[1] == src.shape[2] and src.is_contiguous()
    B, S = src.shape[0], src.shape[1]
    sh = shifts.detach().cpu().to(torch.int64)
    if sh.shape != (B, 2) or bool((sh < 0).any()) or bool((sh + out > S).any()):
        raise ValueError("two_view: shifts must be (B,2) with 0 <= d and d + out <= S")      # processing.py:112-113 asserts
    shifts = shifts.to(device=src.device, dtype=torch.int32).contiguous()
    if noise is not None:
        assert noise.dtype == torch.float64 and tuple(noise.shape) == (B, out, out) and noise.is_contiguous()
    img = torch.empty(B, out, out, dtype=torch.float32, device=src.device)
    img_t = torch.empty_like(img)
    call("cmu_two_view", _p(src), B, S, _p(shifts), _p(noise), int(seed) & (2 ** 64 - 1), _p(img), _p(img_t), out, _stream())
    return img, img_t


def philox_normal(n, offset=0, seed=0, device="cuda"):
    out = torch.empty(n, dtype=torch.float64, device=device)
    call("cmu_philox_normal", _p(out), n, int(offset), int(seed) & (2 ** 64 - 1), _stream())
    return out


# ---- skinny GEMMs of the necks ----------------------------------------------------------------------------------
SKINNY_MAX_ROWS = 256      # csrc/skinny.hip SK_MAX_M: the reference's own batch size per GPU (cmunet_config.py:55, moco2_module.py:91)


def skinny_eligible(x, weight):
    """nn.Linear on <= SKINNY_MAX_ROWS fp32 rows with K % 8 == 0: what the weight-streaming kernels take (the product path has
    no other GEMM: callers raise on anything else)."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and 1 <= x.shape[0] <= SKINNY_MAX_ROWS and x.shape[1] % 8 == 0 and x.shape[1] >= 8)


def _mm16(compute_dt, K, need):
    """16-bit-operand kernels apply for compute_dt f16 / bf16 when K allows; None / f32 -> the exact fp32 kernels."""
    if compute_dt is None:
        return None
    dt = dt_code(compute_dt)
    return dt if (dt in (F16, BF16) and K % need == 0) else None


def skinny_gemm_fwd(x, weight, bias=None, compute_dt=None):
    """``compute_dt`` 'f16' / 'bf16': operands rounded to that type in registers (the AMP arithmetic), fp32 accumulation."""
    x, weight = _f32c(x), _f32c(weight)
    M, K = x.shape
    N = weight.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    dt16 = _mm16(compute_dt, K, 16)
    b = _p(None if bias is None else _f32c(bias))
    if dt16 is not None:
        ws = torch.empty(_lib.lib().cmu_skinny16_gemm_ws_bytes(M, N, K), dtype=torch.uint8, device=x.device)
        call("cmu_skinny16_gemm_fwd", _p(x), _p(weight), b, _p(y), M, N, K, dt16, _p(ws), _stream(), work=2.0 * M * N * K)
    else:
        ws = torch.empty(_lib.lib().cmu_skinny_gemm_ws_bytes(M, N, K), dtype=torch.uint8, device=x.device)
        call("cmu_skinny_gemm_fwd", _p(x), _p(weight), b, _p(y), M, N, K, _p(ws), _stream(), work=2.0 * M * N * K)
    return y


def skinny_gemm_dgrad(dy, weight, compute_dt=None):
    dy, weight = _f32c(dy), _f32c(weight)
    M, N = dy.shape
    K = weight.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    dt16 = _mm16(compute_dt, K, 4)
    if dt16 is not None:
        call("cmu_skinny16_gemm_dgrad", _p(dy), _p(weight), _p(dx), M, N, K, dt16, _stream(), work=2.0 * M * N * K)
    else:
        ws = torch.empty(_lib.lib().cmu_skinny_gemm_bwd_ws_bytes(M, N), dtype=torch.uint8, device=dy.device)
        call("cmu_skinny_gemm_dgrad", _p(dy), _p(weight), _p(dx), M, N, K, _p(ws), _stream(), work=2.0 * M * N * K)
    return dx


def skinny_gemm_wgrad(dy, x, with_bias=False, compute_dt=None, out=None):
    """``out``: optional preallocated (N, K) fp32 tensor (a view of a trainer's gradient arena)."""
    dy, x = _f32c(dy), _f32c(x)
    M, N = dy.shape
    K = x.shape[1]
    dw = out if out is not None else torch.empty(N, K, dtype=torch.float32, device=dy.device)
    assert dw.shape == (N, K) and dw.dtype == torch.float32 and dw.is_contiguous()
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if with_bias else None
    dt16 = _mm16(compute_dt, K, 4)
    if dt16 is not None:
        call("cmu_skinny16_gemm_wgrad", _p(dy), _p(x), _p(dw), _p(db), M, N, K, dt16, _stream(), work=2.0 * M * N * K)
    else:
        call("cmu_skinny_gemm_wgrad", _p(dy), _p(x), _p(dw), _p(db), M, N, K, _stream(), work=2.0 * M * N * K)
    return dw, db


# ---- BatchNorm1d (+ ReLU) of the necks and the target latent's 1x1 reduction (csrc/necks.hip) ---------------------------------
def bn1d_colsums(x):
    M, N = x.shape
    sums = torch.empty(2, N, dtype=torch.float32, device=x.device)
    call("cmu_bn1d_colsums", _p(_f32c(x)), _p(sums), M, N, _stream())
    return sums


def bn1d_relu_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, sums=None, count=0):
    M, N = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(N, dtype=torch.float32, device=x.device)
    invstd = torch.empty(N, dtype=torch.float32, device=x.device)
    call("cmu_bn1d_relu_fwd", _p(_f32c(x)), _p(sums), int(count), _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(momentum),
         float(eps), int(training), int(relu), _p(y), _p(mean), _p(invstd), M, N, _stream())
    return y, mean, invstd


def bn1d_bwd_colsums(dy, x, y, mean, invstd, relu):
    M, N = x.shape
    sums = torch.empty(2, N, dtype=torch.float32, device=x.device)
    call("cmu_bn1d_bwd_colsums", _p(_f32c(dy)), _p(x), _p(y), _p(mean), _p(invstd), int(relu), _p(sums), M, N, _stream())
    return sums


def bn1d_relu_bwd(dy, x, y, mean, invstd, gamma, relu, sums=None, count=0, affine=True):
    M, N = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(N, dtype=torch.float32, device=x.device) if affine else None
    db = torch.empty(N, dtype=torch.float32, device=x.device) if affine else None
    call("cmu_bn1d_relu_bwd", _p(_f32c(dy)), _p(x), _p(y), _p(mean), _p(invstd), _p(gamma), int(relu), _p(sums), int(count), _p(dx), _p(dg),
         _p(db), M, N, _stream())
    return dx, dg, db


def conv1x1_nchw_fwd(x, weight, bias=None):
    """x: Act (NHWC + pending transform); weight (N, K) or (N, K, 1, 1) fp32 -> (B, N, H, W) fp32."""
    w = _f32c(weight.reshape(weight.shape[0], -1))
    N, K = w.shape
    assert K == x.C
    out = torch.empty(x.B, N, x.H, x.W, dtype=torch.float32, device=x.buf.device)
    call("cmu_conv1x1_nchw_fwd", x.ptr(), x.ld, _p(x.scale), _p(x.shift), x.relu_from, _p(w), _p(None if bias is None else _f32c(bias)),
         _p(out), x.B, x.H, x.W, K, N, x.dt, _stream(), work=2.0 * K * N * x.B * x.H * x.W)
    return out


def random_patch_mask(B, H, W, patch_size=16, mask_ratio=0.65, seed=0, offset=0, device="cuda"):
    """UNet_encoder.create_random_patch_mask (UNet_encoder.py:106-139) in one kernel: (B,H,W) uint8, 1 = masked,
    floor(ratio*H*W / patch^2) patches per sample, reproducible from (seed, offset)."""
    n_mask = int(mask_ratio * H * W) // (patch_size * patch_size)
    m = torch.empty(B, H, W, dtype=torch.uint8, device=device)
    call("cmu_random_patch_mask", _p(m), B, H, W, patch_size, n_mask, int(seed) & (2 ** 64 - 1), int(offset), _stream())
    return m
