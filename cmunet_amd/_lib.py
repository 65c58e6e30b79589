"""ctypes binding of libcmunet_hip.so (the C-ABI declared in include/cmunet_hip.h).

The product path has no CPU fallback: if the HIP library is missing (or a call fails) this module
raises.  ``build()`` compiles it in-tree with hipcc for gfx950 (works without a GPU).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("CMU_LIB_PATH") or os.path.join(CSRC, "libcmunet_hip.so")   # CMU_LIB_PATH: diagnostic builds (tools/)

F32, F16, BF16 = 0, 1, 2

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float
_U64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/cmunet_hip.h one to one.
_SIGS = {
    "cmu_last_error": (ctypes.c_char_p, []),
    "cmu_last_kernel": (ctypes.c_char_p, []),
    "cmu_softmax2_threshold": (_I, [_P, _F, _P, _I, _I, _I, _P]),
    "cmu_conv3x3_c1_wgrad_bn": (_I, [_P, _P, _I, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_conv3x3_c1_wgrad_bn_w": (_I, [_P, _P, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_soft_skeleton_ws_bytes": (_L, [_L]),
    "cmu_soft_skeleton": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "cmu_cldice_sums_ws_bytes": (_L, []),
    "cmu_cldice_sums": (_I, [_P, _P, _P, _P, _L, _P, _P, _P]),
    "cmu_sgd_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _L, _F, _P]),
    "cmu_lamb_block_elems": (_I, []),
    "cmu_lamb_ws_bytes": (_L, [_I, _I]),
    "cmu_resize_bicubic_ws_bytes": (_L, [_I, _I, _I, _I, _I]),
    "cmu_resize_bicubic": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P]),
    "cmu_resize_bicubic_u8_ws_bytes": (_L, [_I, _I, _I, _I, _I]),
    "cmu_resize_bicubic_u8": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P]),
    "cmu_resize_nearest_u8_ws_bytes": (_L, [_I, _I]),
    "cmu_resize_nearest_u8": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P]),
    "cmu_two_view": (_I, [_P, _I, _I, _P, _P, _U64, _P, _P, _I, _P]),
    "cmu_philox_normal": (_I, [_P, _L, _U64, _U64, _P]),
    "cmu_random_patch_mask": (_I, [_P, _I, _I, _I, _I, _I, _U64, _U64, _P]),
    "cmu_skinny_gemm_ws_bytes": (_L, [_I, _I, _L]),
    "cmu_skinny_gemm_fwd": (_I, [_P, _P, _P, _P, _I, _I, _L, _P, _P]),
    "cmu_skinny_gemm_bwd_ws_bytes": (_L, [_I, _I]),
    "cmu_skinny_gemm_dgrad": (_I, [_P, _P, _P, _I, _I, _L, _P, _P]),
    "cmu_skinny_gemm_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _L, _P]),
    "cmu_lamb_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _F, _F, _F, _F, _I, _I, _F, _I, _I, _L, _F, _P, _P]),
    "cmu_pack_desc_bytes": (_I, []),
    "cmu_pack_desc_blocks": (_L, [_I, _I, _I, _I, _I]),
    "cmu_pack_batch": (_I, [_P, _I, _L, _I, _P]),
    "cmu_version": (_I, []),
    "cmu_set_dispatch_override": (_I, [ctypes.c_char_p, _I]),
    "cmu_mfma_sustained_rate": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "cmu_probe_stream_reduce": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "cmu_dtype_size": (_I, [_I]),
    "cmu_pack_conv3x3_elems": (_L, [_I, _I, _I, _I]),
    "cmu_pack_conv3x3": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cmu_pack_convT2x2_elems": (_L, [_I, _I, _I, _I]),
    "cmu_pack_convT2x2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cmu_conv3x3_c1_fwd": (_I, [_P, _P, _I, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_conv3x3_fwd": (_I, [_P, _L, _P, _P, _I, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_conv_ntiles": (_I, [_I, _I, _I]),
    "cmu_bn_finalize_ws_bytes": (_L, [_I]),
    "cmu_bn_finalize": (_I, [_P, _I, _L, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _I, _P, _P]),
    "cmu_bnrelu_maxpool_fwd": (_I, [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_convT2x2_fwd": (_I, [_P, _L, _P, _P, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_conv1x1_head_fwd": (_I, [_P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_apply_to_nchw": (_I, [_P, _L, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_nchw_to_nhwc": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_nhwc_to_nchw": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_bn_bwd_ws_bytes": (_L, [_I]),
    "cmu_bn_bwd_reduce": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_bn_bwd_apply": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_conv3x3_wgrad_ws_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "cmu_conv3x3_wgrad": (_I, [_P, _L, _P, _P, _I, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_conv3x3_c1_wgrad_ws_bytes": (_L, [_I, _I, _I, _I]),
    "cmu_conv3x3_c1_wgrad": (_I, [_P, _P, _I, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_maxpool_bwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_maxpool_bwd2": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_maxpool_bwd_apply": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_bnrelu_maxpool_fwd_masked": (_I, [_P, _L, _P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_maxpool_bwd_masked": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_bn_bwd_finalize": (_I, [_P, _L, _P, _P, _P, _I, _P]),
    "cmu_convT2x2_dgrad": (_I, [_P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_convT2x2_dgrad_bn": (_I, [_P, _L, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_conv3x3_dgrad_bn": (_I, [_P, _L, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_bn_bwd_finalize_tiles": (_I, [_P, _I, _L, _P, _P, _P, _I, _P, _P]),
    "cmu_convT2x2_wgrad_ws_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "cmu_convT2x2_wgrad": (_I, [_P, _L, _P, _P, _I, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_conv1x1_head_bwd_ws_bytes": (_L, [_I, _I, _I, _I, _I]),
    "cmu_conv1x1_head_bn_apply": (_I, [_P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_conv1x1_head_bwd": (_I, [_P, _P, _L, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_masked_mse_ws_bytes": (_L, [_I, _I]),
    "cmu_masked_mse_fwd_bwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _F, _P, _I, _I, _I, _P, _P]),
    "cmu_softmax_ce_dice_ws_bytes": (_L, [_I, _I, _I]),
    "cmu_softmax_ce_dice_fwd_bwd": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P]),
    "cmu_infonce_inbatch_fwd_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "cmu_moco_ws_bytes": (_L, [_I, _I, _I]),
    "cmu_moco_infonce_enqueue": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P]),
    "cmu_l2_normalize_rows": (_I, [_P, _P, _I, _I, _P]),
    "cmu_l2_normalize_rows_bwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "cmu_moco_logits_assemble": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "cmu_moco_logits_split": (_I, [_P, _P, _I, _I, _F, _P]),
    "cmu_moco_logits_addpos": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "cmu_row_cross_entropy": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "cmu_scale_by_device_scalar": (_I, [_P, _P, _L, _P]),
    "cmu_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_masked_stats_rows": (_I, []),
    "cmu_masked_channel_stats": (_I, [_P, _L, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_mask_select": (_I, [_P, _L, _P, _P, _I, _P, _I, _I, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_bn_bwd_reduce_masked": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_bn_bwd_apply_masked": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_sparse_tile_list": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "cmu_conv3x3_tiles_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "cmu_rows_channel_stats": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_bn_bwd_reduce_rows": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_sparse_tile_lists_max": (_I, []),
    "cmu_sparse_tile_lists": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "cmu_sparse_pixel_lists": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "cmu_conv3x3_c1_fwd_tiles_rows": (_I, [_L]),
    "cmu_conv3x3_c1_fwd_tiles": (_I, [_P, _P, _I, _P, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_conv3x3_c1_wgrad_bn_tiles": (_I, [_P, _P, _I, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_cells_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "cmu_bn_bwd_apply_cells": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_mask_select_cells": (_I, [_P, _L, _P, _P, _I, _P, _I, _I, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_maxpool_bwd_cells": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_cells_stats_rows": (_I, []),
    "cmu_cells_channel_stats": (_I, [_P, _L, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_bn_bwd_reduce_cells": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_cells_channel_sum_ws_bytes": (_L, [_I]),
    "cmu_cells_channel_sum": (_I, [_P, _L, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_sparse_pixel_list_ws_bytes": (_L, [_I, _I]),
    "cmu_sparse_pixel_list": (_I, [_P, _I, _I, _I, _I, _P, _L, _P, _P, _P]),
    "cmu_conv3x3_rows_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "cmu_conv3x3_fwd_rows": (_I, [_P, _L, _P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_conv3x3_fwd_tiles": (_I, [_P, _L, _P, _P, _I, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_conv3x3_wgrad_tile_h": (_I, [_I, _I, _I, _I, _I, _I]),
    "cmu_conv3x3_wgrad_tiles": (_I, [_P, _L, _P, _P, _I, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "cmu_spark_loss_ws_bytes": (_L, [_I, _I]),
    "cmu_spark_loss_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P]),
    "cmu_gap_fwd": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cmu_gap_bwd": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "cmu_ema_update": (_I, [_P, _P, _L, _F, _P]),
    "cmu_adam_step": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _L, _F, _P, _P]),
    "cmu_adam_ema_step": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _L, _F, _P, _I, _P, _P, _P, _F, _P]),
    "cmu_skinny16_gemm_ws_bytes": (_L, [_I, _I, _L]),
    "cmu_skinny16_gemm_fwd": (_I, [_P, _P, _P, _P, _I, _I, _L, _I, _P, _P]),
    "cmu_skinny16_gemm_dgrad": (_I, [_P, _P, _P, _I, _I, _L, _I, _P]),
    "cmu_skinny16_gemm_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _L, _I, _P]),
    "cmu_bn1d_colsums": (_I, [_P, _P, _I, _I, _P]),
    "cmu_bn1d_relu_fwd": (_I, [_P, _P, _L, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P, _P, _I, _I, _P]),
    "cmu_bn1d_bwd_colsums": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "cmu_bn1d_relu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _L, _P, _P, _P, _I, _I, _P]),
    "cmu_conv1x1_nchw_fwd": (_I, [_P, _L, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cmu_amp_state_bytes": (_I, []),
    "cmu_amp_init": (_I, [_P, _F, _P]),
    "cmu_amp_check_finite": (_I, [_P, _L, _P, _P]),
    "cmu_amp_update": (_I, [_P, _F, _F, _I, _P]),
}

EXPORTS = tuple(_SIGS.keys())


class CmuError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip into libcmunet_hip.so with hipcc --offload-arch=gfx950 (in-tree)."""
    jobs = str(min(8, os.cpu_count() or 1))
    res = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise CmuError("building libcmunet_hip.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH


_lib = None


def lib():
    """The loaded library (raises if it has not been built: there is no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CmuError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). The HIP path has no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64: import it FIRST so that this library binds to the same
        # HIP runtime instance as torch's streams/allocator (loading ours first leaves two runtimes in the
        # process and every launch fails with "no ROCm-capable device is detected").
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name, None)
            if fn is None:      # reported by missing_symbols(); calling it raises below
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def missing_symbols():
    """Entry points declared in include/cmunet_hip.h that the built library does not export."""
    l = lib()
    return [n for n in _SIGS if getattr(l, n, None) is None]


class EventProfiler:
    """Optional per-entry-point timing with HIP events on the launch stream (bench.py's roofline leg).
    ``work`` is the algorithmic FLOP (or byte) count the caller attributes to the launch."""

    def __init__(self, gemm_only=False):
        self.records = []      # (name, start_event, end_event, work, kernel tag)
        self.gemm_only = gemm_only   # time only the GEMM-shaped entries (work > 0): a quarter of the events of a step

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, work, _ in self.records:
            d = out.setdefault(name, {"ms": 0.0, "calls": 0, "work": 0.0})
            d["ms"] += e0.elapsed_time(e1)
            d["calls"] += 1
            d["work"] += work
        return out

    def by_kernel(self):
        """The MFMA entries (work > 0) grouped by the compute kernel that served them (cmu_last_kernel)."""
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, work, tag in self.records:
            if work <= 0 or not tag:
                continue
            d = out.setdefault(tag, {"ms": 0.0, "calls": 0, "work": 0.0, "entries": set()})
            d["ms"] += e0.elapsed_time(e1)
            d["calls"] += 1
            d["work"] += work
            d["entries"].add(name)
        return out


PROFILER = None


class DevPtr(ctypes.c_void_p):
    """A device pointer that remembers which GPU owns it (``dev``): ``call`` binds the launch to that device."""


class _StreamOfArgs:
    """Placeholder for "the current stream of the device that owns this call's tensors" (resolved inside ``call``)."""


STREAM = _StreamOfArgs()


def devptr(addr, dev):
    p = DevPtr(addr)
    p.dev = dev
    return p


def call(name, *args, work=0.0):
    """Call an int-returning entry point; raise CmuError with cmu_last_error() on failure.

    Device binding: every ``DevPtr`` argument must live on ONE device; the launch is issued with that device current and on
    that device's current PyTorch stream (the ``STREAM`` placeholder), whatever the process's current device is -- a model on
    ``cuda:1`` (the reference's own choice, Finetuning/train.py:246,451) runs there, not on device 0's stream."""
    l = lib()
    fn = getattr(l, name, None)
    if fn is None:
        raise CmuError(f"{name} is not exported by {LIB_PATH}")
    import torch
    dev, si = None, -1
    for i, a in enumerate(args):
        if type(a) is DevPtr:
            if dev is None:
                dev = a.dev
            elif a.dev != dev:
                raise CmuError(f"{name}: tensors on different devices (cuda:{dev} and cuda:{a.dev})")
        elif a is STREAM:
            si = i
    cur = torch.cuda.current_device()
    if dev is None:
        dev = cur
    if si >= 0:
        args = list(args)
        args[si] = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if dev != cur:
        with torch.cuda.device(dev):
            return _call_bound(l, fn, name, args, work)
    return _call_bound(l, fn, name, args, work)


def _call_bound(l, fn, name, args, work):
    if PROFILER is not None and (work > 0 or not PROFILER.gemm_only):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        PROFILER.records.append((name, e0, e1, work, l.cmu_last_kernel().decode() if work > 0 else ""))
    else:
        rc = fn(*args)
    if rc != 0:
        raise CmuError(f"{name} failed ({rc}): {l.cmu_last_error().decode()}")
    return rc
