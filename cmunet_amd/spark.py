"""Drop-in for the SparK sparse masked-conv pretraining variant on the UNet (reference:
Pretraining/Spark/{spark.py:20-149, encoder.py:12-56,158-208, decoder.py:39-58, models/custom.py:113-182,
models/__init__.py:40-49}).

  build_sparse_encoder('unet_sparse', input_size, sbn=False) -> SparseEncoder   (models/__init__.py, encoder.py:158-208)
  UnetDecoder(width=768, in_chans=1)                                             (decoder.py:39-58)
  SparK(sparse_encoder, dense_decoder, mask_ratio=0.6, densify_norm='', sbn=False)
      .mask(B, device, generator)          spark.py:82-86
      .forward(inp_bchw, active_b1ff=None) spark.py:88-131  -> reconstruction loss on the non-active patches
      state_dict keys: sparse_encoder.sp_cnn.*, dense_decoder.*, mask_tokens.{i}, densify_projs.{i}.* (built but
      unused for the full UNet, kept for key parity: SURVEY A-10)

The module-global ``_cur_active`` of the reference (encoder.py:12) is an explicit argument here.  Execution model
(csrc/sparse.hip): sparse-BatchNorm statistics over the active positions, the masked BN+ReLU apply, the densify step
and the loss are HBM-bound kernels with the patch mask looked up per pixel.  The encoder's convolutions run on the
MFMA implicit-GEMM kernels; where whole kernel tiles fall on masked patches they are SKIPPED (K17): per level a
device-side list of the 16 x 32 pixel tiles (forward / data gradient, persistent kernel) and 16 x 16 pixel tiles
(weight gradient) that overlap an active patch is built once per step (``ops.TileList``) -- the reference computes
the dense op and multiplies by the mask (encoder.py:20-23).  What a dense tile can skip depends on the patch side at
the level: 16 px (level 1): 1 - 0.75^2 = 44 % of the 16 x 32 tiles and 25 % of the 16 x 16 tiles remain at mask ratio
0.75; 8 px (level 2): 90 % remain; deeper levels have patches of 4 / 2 / 1 px, every tile holds an active pixel and the
dense tile list cannot skip anything.  From level 2 down the forward and data-gradient convolutions (output channels a
multiple of 128) therefore run as a GATHER-GEMM over the list of active pixels (``ops.PixelList`` /
``ops.conv3x3_fwd_rows``, csrc/conv_gather.inc): GEMM rows = active pixels, the halo gathered per tap from the dense input,
outputs scattered to the active positions -- exactly the active fraction of the dense FLOPs; the same list drives the
sparse-BatchNorm statistics passes at every level.  ``CMU_SPARK_TILES=0`` keeps every level dense, ``CMU_SPARK_GATHER=0`` only
the gather levels (A/B switches).  ``sbn=True``: the bottleneck's two BatchNorms exchange their statistics
(forward sums and counts, backward sums) over the default process group, as nn.SyncBatchNorm does for SparseSyncBatchNorm2d.
"""
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .engine import BN_EPS, BN_MOMENTUM

_CELL_STATS = os.environ.get("CMU_SPARK_CELL_STATS", "1") != "0"    # A/B: "0" = the statistics passes walk the pixel list
_FUSE_POOL = os.environ.get("CMU_SPARK_POOL_FUSE", "1") != "0"     # A/B: "0" = the activated + masked copy of every conv output is stored
from .model import DoubleConv, DownBlock, UpBlock, _EngineOwner, _named_state, _param_args, _require_cuda
from .ops import Act


class UNET_MAE_SPARSE(nn.Module):
    """Parameter container with the reference's names (models/custom.py:113-182); executed by SparK."""

    def __init__(self, in_chans=1, base_ch=64, depth=5, dtype="f32", **kwargs):
        super().__init__()
        chans = [base_ch * 2 ** i for i in range(depth)]
        cin = in_chans
        for i in range(depth - 1):
            setattr(self, f"down_conv{i + 1}", DownBlock(cin, chans[i], dtype))
            cin = chans[i]
        self.double_conv = DoubleConv(cin, chans[-1], dtype)
        self._chans = chans

    def get_downsample_ratio(self):
        return 2 ** (len(self._chans) - 1)

    def get_feature_map_channels(self):
        return list(self._chans)


class SparseEncoder(nn.Module):
    def __init__(self, cnn, input_size, sbn=False, verbose=False):
        super().__init__()
        # sbn: dense_model_to_sparse (encoder.py:181-192) turns the plain nn.BatchNorm2d modules into SparseSyncBatchNorm2d;
        # in unet_sparse those are the two of the bottleneck DoubleConv only (custom.py:125-130 -- the down blocks are built
        # from SparseBatchNorm2d directly; SURVEY A-10).  SparK._step exchanges their statistics over the process group.
        self.sbn = bool(sbn)
        self.sp_cnn = cnn
        self.input_size, self.downsample_raito, self.enc_feat_map_chs = input_size, cnn.get_downsample_ratio(), cnn.get_feature_map_channels()


def build_sparse_encoder(name, input_size, sbn=False, drop_path_rate=0.0, verbose=False, base_ch=64, depth=5, dtype="f32"):
    if name not in ("unet_sparse", "unet"):
        raise NotImplementedError(f"only the UNet encoders of the reference's hot path are built here (got {name})")
    return SparseEncoder(UNET_MAE_SPARSE(base_ch=base_ch, depth=depth, dtype=dtype), input_size=input_size, sbn=sbn, verbose=verbose)


class UnetDecoder(nn.Module):
    def __init__(self, width=768, in_chans=1, base_ch=64, depth=5, dtype="f32"):
        super().__init__()
        self.width = width
        self.up_sample_mode = 'conv_transpose'
        chans = [base_ch * 2 ** i for i in range(depth)]
        for i in range(depth - 1, 0, -1):
            setattr(self, f"up_conv{i}", UpBlock(chans[i], chans[i - 1], self.up_sample_mode, dtype))
        self.conv_last = nn.Conv2d(chans[0], in_chans, kernel_size=1)


class _SparKFn(torch.autograd.Function):
    """Whole SparK step: the backward pass is run together with the forward (gradients for a unit loss weight are
    kept and scaled by the incoming gradient), so one Function covers encoder, densify, decoder and loss."""

    @staticmethod
    def forward(ctx, module, inp, active, need_grads, names, *params):
        loss, grads = module._step(inp, active, need_grads=need_grads)     # (grad mode is off inside Function.forward)
        ctx.names, ctx.grads, ctx.module = names, grads, module
        return loss

    @staticmethod
    def backward(ctx, g):
        grads = ctx.grads or {}
        # scaled in place by the incoming gradient in a few multi-tensor launches (one multiply per parameter was 86 launches a step);
        # the tensors are the engine's: fresh ones, or aliases of the trainer's gradient arena that autograd adopts as they are
        have = [grads[n] for n in ctx.names if grads.get(n) is not None]
        # (``_unit_backward``: set by pretrain.SparKPretrainer, whose loss.backward() passes exactly 1 -- the gradients may already be
        # in an all-reduce started from inside the forward, so they are not touched again)
        if have and not getattr(ctx.module, "_unit_backward", False):
            torch._foreach_mul_(have, g.reshape(()).to(have[0].dtype))
        out = [grads.get(n) for n in ctx.names]
        ctx.grads = None
        return (None, None, None, None, None, *out)


class SparK(_EngineOwner, nn.Module):
    def __init__(self, sparse_encoder, dense_decoder, mask_ratio=0.6, densify_norm='', sbn=False, dtype="f32"):
        super().__init__()
        if densify_norm.lower() not in ("", "identity", "none"):
            raise NotImplementedError("the UNet configuration of the reference uses densify_norm='' (arg_util.py:124-130)")
        self.dtype = dtype
        input_size, r = sparse_encoder.input_size, sparse_encoder.downsample_raito
        self.downsample_raito = r
        self.fmap_h = self.fmap_w = input_size // r
        self.mask_ratio = mask_ratio
        self.len_keep = round(self.fmap_h * self.fmap_w * (1 - mask_ratio))
        self.sparse_encoder, self.dense_decoder = sparse_encoder, dense_decoder
        self.full_unet = isinstance(dense_decoder, UnetDecoder)
        self.sbn = sbn
        self.hierarchy = len(sparse_encoder.enc_feat_map_chs)
        self.densify_norm_str = densify_norm.lower()
        # static loss scale of the fused step (set by pretrain.SparKPretrainer for f16 storage: a per-pixel loss gradient is
        # ~3e-7 at bs 32 x 512 x 512, below f16's normal range); the parameter gradients come out multiplied by it
        self.grad_scale = 1.0
        self.keep_rec, self.last_rec = False, None      # keep_rec: the step leaves its decoder output (B,1,H,W) fp32 in last_rec (vis, tests)
        self.densify_projs = nn.ModuleList()
        self.mask_tokens = nn.ParameterList()
        e_widths, d_width = list(sparse_encoder.enc_feat_map_chs), dense_decoder.width
        for i in range(self.hierarchy):
            e_width = e_widths.pop()
            p = nn.Parameter(torch.zeros(1, e_width, 1, 1))
            nn.init.trunc_normal_(p, mean=0, std=.02, a=-.02, b=.02)
            self.mask_tokens.append(p)
            if i == 0 and e_width == d_width:
                self.densify_projs.append(nn.Identity())
            else:
                k = 1 if i <= 0 else 3
                self.densify_projs.append(nn.Conv2d(e_width, d_width, kernel_size=k, stride=1, padding=k // 2, bias=True))
            d_width //= 2

    def mask(self, B, device, generator=None):
        h, w = self.fmap_h, self.fmap_w
        idx = torch.rand(B, h * w, generator=generator).argsort(dim=1)[:, :self.len_keep].to(device)
        active = torch.zeros(B, h * w, dtype=torch.bool, device=device).scatter_(dim=1, index=idx, value=True).view(B, 1, h, w)
        # known by construction (the step needs the count on the host and would read it back otherwise); stored with the tensor's version:
        # an in-place edit of the mask (forcing a patch active, ...) invalidates it (advisor, round 4: a stale count fed the sparse
        # BatchNorm counts and the list capacities -- list-driven kernels walking past their rows)
        active._cmu_n_active = (active._version, B * self.len_keep)
        return active

    def forward(self, inp_bchw, active_b1ff=None, vis=False):
        _require_cuda(inp_bchw, "SparK")
        if active_b1ff is None:
            active_b1ff = self.mask(inp_bchw.shape[0], inp_bchw.device)
        if vis:
            # spark.py:124-128: (input, masked input, reconstruction pasted into the masked patches) -- a forward of the HIP path that
            # keeps its decoder output, then the reference's own de-normalisation as element-wise glue (not a training path)
            keep, self.keep_rec = self.keep_rec, True
            try:
                with torch.no_grad():
                    self._step(inp_bchw, active_b1ff, need_grads=False)
                rec = self.last_rec
            finally:
                self.keep_rec, self.last_rec = keep, None
            r = self.downsample_raito
            active_b1hw = active_b1ff.repeat_interleave(r, 2).repeat_interleave(r, 3)
            inp = self.patchify(inp_bchw.float())
            mean = inp.mean(dim=-1, keepdim=True)
            var = (inp.var(dim=-1, keepdim=True) + 1e-6) ** .5
            rec_bchw = self.unpatchify(self.patchify(rec) * var + mean)
            return inp_bchw, inp_bchw * active_b1hw, torch.where(active_b1hw, inp_bchw.float(), rec_bchw)
        names, params = _param_args(self)
        need_grads = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _SparKFn.apply(self, inp_bchw, active_b1ff, need_grads, names, *params)

    def patchify(self, bchw):
        """spark.py:133-139: (B, C, H, W) -> (B, f*f, C*p*p)."""
        p, h, w = self.downsample_raito, self.fmap_h, self.fmap_w
        _require_cuda(bchw, "SparK.patchify")
        return ops.patchify(bchw, h, w, p).to(bchw.dtype)          # one gather pass (cmu_patchify), no einsum

    def unpatchify(self, bln):
        """spark.py:141-148: the inverse of ``patchify``."""
        p, h, w = self.downsample_raito, self.fmap_h, self.fmap_w
        _require_cuda(bln, "SparK.unpatchify")
        return ops.patchify(bln, h, w, p, inverse=True).to(bln.dtype)

    # ---------------------------------------------------------------------------------------------
    def _ident(self, eng, C):
        cache = eng.__dict__.setdefault("_ident", {})
        if C not in cache:
            cache[C] = (torch.ones(C, dtype=torch.float32, device=eng.device), torch.zeros(C, dtype=torch.float32, device=eng.device))
        return cache[C]

    @staticmethod
    def _sync_world():
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _level_tiles(self, eng, active, B, H, W, n_cells, convs=(), need_grads=True, pending=None):
        """What one level can skip: {'conv': 16 x 32 tile list, 'wgrad': tile lists by tile height, 'c1': 16 x 16 list of the one-channel
        first layer, 'cf' / 'wf': expected shares of listed tiles (profiler only), 'pix': list of active pixels for the gather kernel and
        the statistics passes} -- or None.  Patch side >= 8 px: tile lists; smaller patches (every dense tile holds an active pixel): the
        pixel list.  ``convs``: (Cin, Cout) of the level's layers -- only the lists they use are built.  ``pending`` = (tile lists, pixel
        lists): the lists are only allocated and collected there; ``ops.build_lists`` fills all of a step's lists in two launches
        (round 4: one or two single-workgroup launches per list, 0.3 ms per step, sat on the critical path)."""
        if os.environ.get("CMU_SPARK_TILES", "1") == "0":
            return None
        defer = pending is not None
        f = active.shape[-1]
        ps = H // f
        gather = os.environ.get("CMU_SPARK_GATHER", "1") != "0" and H == W
        # the list of active pixels drives the gather convolutions (small patches) and, at every level, the statistics passes
        pix = None
        if gather:
            pix = ops.PixelList(active, H, W, n_cells * ps * ps, defer=defer)
            if defer:
                pending[1].append(pix)
        if ps < 8 or H % 16 != 0 or W % 32 != 0:
            return {"conv": None, "wgrad": None, "c1": None, "cf": 1.0, "wf": 1.0, "pix": pix, "gather": True} if gather else None
        keep = 1.0 - self.mask_ratio

        def rows_ok(ci, co):
            return pix is not None and ops.conv3x3_rows_supported(B, H, W, ci, co, eng.dt)

        # the 16 x 32 list: only when a layer (or its data gradient) is served by the tile kernel and not by the gather kernel
        multi = [(ci, co) for ci, co in convs if ci != 1]
        want = [(a, b) for ci, co in multi for a, b in ((ci, co), (co, ci))] if convs else [(0, 0)]
        need_conv = (not convs) or any((not rows_ok(a, b)) and ops.conv3x3_tiles_supported(B, H, W, a, b, eng.dt) for a, b in want)
        conv = None
        if need_conv:
            conv = ops.TileList(active, H, W, 16, 32, defer=defer)
            if defer:
                pending[0].append(conv)
        cf = 1.0 - (1.0 - keep) ** max(1, (16 // ps) * (32 // ps))          # expected share of listed tiles (profiler only)
        # weight gradients: tile lists by tile height -- 16 x 16 tiles of the first kernel where a patch fills them (level 1), 8 x 16
        # tiles (the wide kernel's K tile = two 8 x 8 patches at level 2: 44 % listed at mask ratio 0.75); built on first use, or
        # with the step's other lists when the layers are known
        wg = {"active": active, "H": H, "W": W, "lists": {}, "ps": ps, "keep": keep}
        tiles = {"conv": conv, "wgrad": wg, "c1": None, "cf": cf, "wf": 1.0, "pix": pix, "gather": pix is not None}
        if need_grads:
            for ci, co in multi:
                self._wgrad_list(tiles, ops.conv3x3_wgrad_tile_h(B, H, W, ci, co, eng.dt), pending)
        # the one-channel first layer walks 16 x 16 tiles: with patches that are multiples of 16 pixels every listed tile lies inside
        # an active patch -- masked tiles are neither computed nor read, the kernel's own sums are the sparse statistics
        if any(ci == 1 for ci, _ in convs) and ps % 16 == 0 and os.environ.get("CMU_SPARK_C1_TILES", "1") != "0":
            lst = self._wgrad_list(tiles, 16, pending)
            if lst is not None:
                tiles["c1"] = (lst[0], n_cells * (ps // 16) ** 2)
        # (where both apply -- level 2: 90 % of the tiles against 25 % of the rows -- the gather kernel is taken first)
        return tiles

    @staticmethod
    def _wgrad_list(tiles, tile_h, pending=None):
        """(TileList, expected listed share) of a level for the weight-gradient kernel that walks ``tile_h`` x 16 tiles, or None when
        such tiles cannot skip anything worth a list (more than two patches per tile side)."""
        wg = tiles["wgrad"] if tiles is not None else None
        if wg is None:
            return None
        ps = wg["ps"]
        if ps < tile_h or ps < 8:
            return None
        if tile_h not in wg["lists"]:
            share = 1.0 - (1.0 - wg["keep"]) ** max(1, (tile_h // ps) * (16 // ps))
            tl = ops.TileList(wg["active"], wg["H"], wg["W"], tile_h, 16, defer=pending is not None)
            if pending is not None:
                pending[0].append(tl)
            wg["lists"][tile_h] = (tl, share)
        return wg["lists"][tile_h]

    @classmethod
    def _wgrad_inside(cls, tiles, tile_h):
        """Whether the weight-gradient kernel of a layer walks a tile list whose every tile lies inside ONE patch (patch side a multiple
        of 16 and of the tile height): it then never reads a masked patch beyond the one-pixel halo of an active one.  (An 8 x 16 tile
        over two 8 x 8 patches may pair an active patch with a masked one, whose interior it reads.)"""
        lst = cls._wgrad_list(tiles, tile_h)
        return lst is not None and tiles["wgrad"]["ps"] % 16 == 0 and tiles["wgrad"]["ps"] % tile_h == 0

    @staticmethod
    def _fwd_listed(tiles, B, H, W, Cin, Cout, dt):
        """Whether a 3x3 layer of this level runs on a list (gather rows or 16 x 32 tiles): it then reads active patches + a one-pixel
        halo of its input and nothing else."""
        if tiles is None:
            return False
        if tiles["gather"] and ops.conv3x3_rows_supported(B, H, W, Cin, Cout, dt):
            return True
        return tiles["conv"] is not None and ops.conv3x3_tiles_supported(B, H, W, Cin, Cout, dt)

    def _sp_convbn_fwd(self, eng, sd, pconv, pbn, x, x_img, inv_pix, active, count, B, H, W, training, sync=False, tiles=None, need_a=True,
                       ring=False):
        """``need_a`` False (a level's SECOND conv, round 3): the activated + masked copy of the output is not materialised -- its
        consumers (the mask-aware pool, the densify select, the pool backward) work from the raw output + transform + mask.
        ``ring``: every consumer of the activated copy is list-driven, so only the border frame of masked patches is zeroed."""
        w = sd[pconv + "weight"]
        C = w.shape[0]
        y = eng._new(B, H, W, C)
        slab = None
        c1 = tiles["c1"] if (tiles is not None and x_img is not None) else None
        if c1 is not None:
            slab = ops.conv3x3_c1_fwd_tiles(x_img, w.detach(), y, c1[0], c1[1], inv_pix, True, want_stats=training)     # masked tiles never computed
        elif x_img is not None:
            ops.conv3x3_c1_fwd(x_img, w.detach(), y, None, inv_pix, True)
        elif tiles is not None and tiles["gather"] and ops.conv3x3_rows_supported(B, H, W, x.C, C, eng.dt):
            ops.conv3x3_fwd_rows(x, eng._wp(pconv, w, False), y, tiles["pix"])                   # GEMM rows = active pixels only
        elif tiles is not None and tiles["conv"] is not None and ops.conv3x3_tiles_supported(B, H, W, x.C, C, eng.dt):
            ops.conv3x3_fwd_tiles(x, eng._wp(pconv, w, False), y, tiles["conv"], tiles["cf"])    # masked tiles never computed
        else:
            ops.conv3x3_fwd(x, eng._wp(pconv, w, False), y, None)
        scale, shift, mean, invstd = eng._f32(C), eng._f32(C), eng._f32(C), eng._f32(C)
        if not training or slab is not None:
            pass
        elif _CELL_STATS and ops.cells_supported(y, active):
            slab = ops.cells_channel_stats(y, active)                              # statistics over the active patches, row groups per workgroup
        elif tiles is not None and tiles["pix"] is not None:
            slab = ops.rows_channel_stats(y, tiles["pix"])                         # statistics over the list of active pixels
        else:
            slab = ops.masked_channel_stats(y, active)                             # ... or over all pixels with the mask looked up
        if sync and training and self._sync_world() > 1:
            # SparseSyncBatchNorm2d (encoder.py:54-55 on nn.SyncBatchNorm): sums, sums of squares and counts of all ranks
            import torch.distributed as dist
            tot = torch.cat([slab.double().sum(0).reshape(-1), torch.tensor([float(count)], dtype=torch.float64, device=slab.device)])
            dist.all_reduce(tot)
            count = int(round(tot[-1].item()))
            slab = tot[:-1].float().view(1, 2, C).contiguous()
        ws = eng.scratch.get("bnfin", eng.lib.cmu_bn_finalize_ws_bytes(C))
        ops.bn_finalize(slab, count, sd[pconv + "bias"].detach(), sd[pbn + "weight"].detach(), sd[pbn + "bias"].detach(),
                        sd[pbn + "running_mean"], sd[pbn + "running_var"], BN_MOMENTUM, BN_EPS, training, scale, shift, mean, invstd, ws)
        if training:
            eng._nbt.append(sd[pbn + "num_batches_tracked"])      # += 1 of all layers in one multi-tensor launch (engine.flush_counters)
        yt = y.with_transform(scale, shift, 0)
        a = None
        if need_a:
            a = eng._new(B, H, W, C)
            ops.mask_select(yt, active, a, relu=True, ring=ring)                   # BN + ReLU, zeros at masked positions
        return {"pconv": pconv, "pbn": pbn, "x": x, "x_img": x_img, "mask": inv_pix, "mps": True, "y": yt, "a": a,
                "mean": mean, "invstd": invstd, "sync": sync, "count_all": count, "tiles": tiles}

    def _sp_convbn_bwd(self, eng, sd, s, dA, active, count, grads, need_dx):
        y = s["y"]
        B, H, W, C = y.B, y.H, y.W, y.C
        w = sd[s["pconv"] + "weight"]
        # parameter gradients go straight into the arena of the trainer that holds the parameter, if any (engine._gbuf): no copies at
        # the end of the step, and the data-parallel trainer can start a bucket's exchange the moment its last gradient has landed
        dgamma, dbeta = eng._gbuf(s["pbn"] + "weight", sd[s["pbn"] + "weight"]), eng._gbuf(s["pbn"] + "bias", sd[s["pbn"] + "bias"])
        coef = eng._f32(2, C)
        tiles = s.get("tiles")
        if _CELL_STATS and ops.cells_supported(y, active):
            ops.bn_bwd_reduce_cells(dA, y, s["mean"], s["invstd"], dgamma, dbeta, coef, active, count, eng._bn_ws(C))
        elif tiles is not None and tiles["pix"] is not None:
            ops.bn_bwd_reduce_rows(dA, y, s["mean"], s["invstd"], dgamma, dbeta, coef, tiles["pix"], count, eng._bn_ws(C))
        else:
            ops.bn_bwd_reduce_masked(dA, y, s["mean"], s["invstd"], dgamma, dbeta, coef, active, count, eng._bn_ws(C))
        if s.get("sync") and self._sync_world() > 1:
            # SyncBatchNorm backward: the input gradient uses the sums over ALL ranks; dgamma / dbeta stay local (the gradient
            # exchange averages them like every other parameter gradient)
            import torch.distributed as dist
            tot = torch.stack([dbeta, dgamma]).double()
            dist.all_reduce(tot)
            coef.copy_((tot / float(s["count_all"])).float())
        grads[s["pbn"] + "weight"], grads[s["pbn"] + "bias"] = dgamma, dbeta
        gb = eng._gbuf(s["pconv"] + "bias", sd[s["pconv"] + "bias"])     # identically zero in front of a training-mode BatchNorm
        eng._zero_pending.append(gb)
        grads[s["pconv"] + "bias"] = gb
        dW = eng._gbuf(s["pconv"] + "weight", w)
        c1 = tiles["c1"] if (tiles is not None and s["x_img"] is not None) else None
        if c1 is not None:
            # the first layer over its tile list: the BatchNorm backward is applied on the fly inside the listed tiles -- no apply pass,
            # no dY tensor (the layer has no data gradient)
            ops.conv3x3_c1_wgrad_bn_tiles(s["x_img"], dA, y, y.scale, y.shift, s["mean"], s["invstd"], coef, dW,
                                          eng.scratch.get("wg", eng.lib.cmu_conv3x3_c1_wgrad_ws_bytes(B, H, W, C)), c1[0], c1[1], s["mask"], True)
            grads[s["pconv"] + "weight"] = dW
            return None
        Cin = w.shape[1]
        lst = None if s["x_img"] is not None else self._wgrad_list(tiles, ops.conv3x3_wgrad_tile_h(B, H, W, Cin, C, eng.dt))
        want_dx = need_dx and s["x_img"] is None
        # zeros are only needed where a consumer of dY looks: list-driven consumers read active patches + a one-pixel halo
        ring = (lst is not None and self._wgrad_inside(tiles, ops.conv3x3_wgrad_tile_h(B, H, W, Cin, C, eng.dt))
                and (not want_dx or self._fwd_listed(tiles, B, H, W, C, Cin, eng.dt)))
        dY = Act(dA.buf, dA.coff, dA.C)
        ops.bn_bwd_apply_masked(dA, y, s["mean"], s["invstd"], coef, dY, active, ring=ring)
        if s["x_img"] is not None:
            ops.conv3x3_c1_wgrad(s["x_img"], dY, dW, eng.scratch.get("wg", eng.lib.cmu_conv3x3_c1_wgrad_ws_bytes(B, H, W, C)), s["mask"], True)
        else:
            wsb = eng.scratch.get("wg", eng.lib.cmu_conv3x3_wgrad_ws_bytes(B, H, W, w.shape[1], C, eng.dt))
            if lst is not None:
                ops.conv3x3_wgrad_tiles(s["x"], dY, dW, wsb, lst[0], lst[1])      # dY is zero outside the listed tiles
            else:
                ops.conv3x3_wgrad(s["x"], dY, dW, wsb)
        grads[s["pconv"] + "weight"] = dW
        if not need_dx or s["x_img"] is not None:
            return None
        Cin = w.shape[1]
        dX = eng._new(B, H, W, Cin)
        if tiles is not None and tiles["gather"] and ops.conv3x3_rows_supported(B, H, W, C, Cin, eng.dt):
            ops.conv3x3_fwd_rows(dY, eng._wp(s["pconv"], w, True), dX, tiles["pix"])                  # dX is only needed where active
        elif tiles is not None and tiles["conv"] is not None and ops.conv3x3_tiles_supported(B, H, W, C, Cin, eng.dt):
            ops.conv3x3_fwd_tiles(dY, eng._wp(s["pconv"], w, True), dX, tiles["conv"], tiles["cf"])
        else:
            ops.conv3x3_fwd(dY, eng._wp(s["pconv"], w, True), dX, None)
        return dX

    def _step(self, inp_bchw, active_b1ff, need_grads):
        eng = self._engine(inp_bchw.device)
        sd = _named_state(self)
        eng.prepack(sd)
        training = self.training
        B, _, H, W = inp_bchw.shape
        f, r = active_b1ff.shape[-1], self.downsample_raito
        assert H == f * r and W == f * r, "input size must be fmap * downsample ratio"
        active = active_b1ff.reshape(B, f, f).to(torch.uint8).contiguous()
        # number of active patches, needed on the host (statistics counts, list capacities): carried by masks from ``mask()``,
        # remembered per mask tensor otherwise -- a read-back per step stalls the launch queue behind the previous step
        n_cells = getattr(active_b1ff, "_cmu_n_active", None)
        if n_cells is not None:
            n_cells = n_cells[1] if n_cells[0] == active_b1ff._version else None     # edited in place since mask(): count it again
        if n_cells is None:
            import weakref
            cache = self.__dict__.setdefault("_n_active_cache", {})
            hit = cache.get(id(active_b1ff))
            if hit is not None and hit[0]() is active_b1ff and hit[1] == active_b1ff._version:     # the same live tensor, unmodified
                n_cells = hit[2]
            else:
                if len(cache) > 64:
                    cache.clear()
                n_cells = int(active.sum().item())
                cache[id(active_b1ff)] = (weakref.ref(active_b1ff), active_b1ff._version, n_cells)
        x_img = inp_bchw.detach().float().reshape(B, H, W).contiguous()
        inv_pix = (1 - active).repeat_interleave(r, 1).repeat_interleave(r, 2).contiguous()
        ep, dp = "sparse_encoder.sp_cnn.", "dense_decoder."
        nd = eng.n_down(sd, ep)

        # ---- the step's lists: every level's tile / pixel lists in two launches, before the first layer ----
        pending = ([], [])
        plans = []
        for k in range(nd + 1):
            p = f"{ep}down_conv{k + 1}.double_conv.double_conv." if k < nd else f"{ep}double_conv.double_conv."
            w0, w3 = sd[p + "0.weight"], sd[p + "3.weight"]
            convs = ((w0.shape[1], w0.shape[0]), (w3.shape[1], w3.shape[0]))
            plans.append((self._level_tiles(eng, active, B, H >> k, W >> k, n_cells, convs, need_grads, pending) if (k < nd or nd > 0) else None,
                          convs))
        if pending[0] or pending[1]:
            ops.build_lists(active, pending[0], pending[1])

        def ring_ok(tl, hh, ww, conv2):
            """The activated copy of a level's first conv output only needs its masked patches' border frames zeroed when the second
            conv reads it through a list, forward and (if there is a backward) in its weight gradient."""
            if tl is None or not self._fwd_listed(tl, B, hh, ww, conv2[0], conv2[1], eng.dt):
                return False
            return (not need_grads) or self._wgrad_inside(tl, ops.conv3x3_wgrad_tile_h(B, hh, ww, conv2[0], conv2[1], eng.dt))

        # ---- sparse encoder (custom.py:152-182) ----
        levels = []
        x, ximg, h, w_ = None, x_img, H, W
        for i in range(1, nd + 1):
            p = f"{ep}down_conv{i}.double_conv.double_conv."
            cnt = n_cells * (h // f) * (w_ // f)
            tl, convs = plans[i - 1]
            s1 = self._sp_convbn_fwd(eng, sd, p + "0.", p + "1.", x, ximg, inv_pix if ximg is not None else None, active, cnt, B, h, w_, training,
                                     tiles=tl, ring=ring_ok(tl, h, w_, convs[1]))
            fuse_pool = _FUSE_POOL and (h // f) >= 2
            s2 = self._sp_convbn_fwd(eng, sd, p + "3.", p + "4.", s1["a"], None, None, active, cnt, B, h, w_, training, tiles=tl,
                                     need_a=not fuse_pool)
            C = s2["y"].C
            pooled = eng._new(B, h // 2, w_ // 2, C)
            if fuse_pool:
                ops.bnrelu_maxpool_fwd_masked(s2["y"], active, pooled)             # masked windows pool to zero; no activated copy of y
            else:
                one, zero = self._ident(eng, C)
                ops.bnrelu_maxpool_fwd(s2["a"].with_transform(one, zero, 0), pooled)   # zeros stay zeros: pool then *= active
            levels.append({"s1": s1, "s2": s2, "cnt": cnt})
            x, ximg, h, w_ = pooled, None, h // 2, w_ // 2
        p = f"{ep}double_conv.double_conv."
        cnt_b = n_cells * (h // f) * (w_ // f)
        sbn = bool(getattr(self.sparse_encoder, "sbn", False))     # (SparK's own ``sbn`` only concerns the densify norms)
        tlb, convs_b = plans[nd]
        b1 = self._sp_convbn_fwd(eng, sd, p + "0.", p + "1.", x, ximg, None, active, cnt_b, B, h, w_, training, sync=sbn, tiles=tlb,
                                 ring=ring_ok(tlb, h, w_, convs_b[1]))
        b2 = self._sp_convbn_fwd(eng, sd, p + "3.", p + "4.", b1["a"], None, None, active, cnt_b, B, h, w_, training, sync=sbn, tiles=tlb,
                                 need_a=not _FUSE_POOL)

        eng.flush_counters()

        # ---- densify (spark.py:98-111): feature maps from the smallest to the largest, mask_tokens[i] likewise ----
        # (a feature map is the activated + masked second-conv output of its level: taken from the stored copy, or -- when that copy was
        # never made -- selected from the raw output + transform in the densify pass itself)
        feats = [b2] + [lv["s2"] for lv in reversed(levels)]
        # the densified skips are written straight into the right halves of the decoder's concat buffers (no copy at the hand-over)
        # (concat buffers + affine arrays of the last shape are reused from step to step: the fused step runs forward and backward in
        # one call, so nothing of the previous step is alive -- 16 fills and four allocations per step otherwise)
        ck = (B, H, W, eng.tdt)
        cc = self.__dict__.get("_cats_cache")
        if cc is None or cc[0] != ck or cc[1] is not eng:
            cc = (ck, eng, eng.decoder_alloc(sd, B, H, W, dp))
            self.__dict__["_cats_cache"] = cc
        cats = cc[2]
        nf = len(feats)
        into_cats = len(cats) == nf - 1 and all(cats[nf - 1 - i]["Cskip"] == feats[i]["y"].C for i in range(1, nf))
        to_dec = []
        for i, s in enumerate(feats):
            a = s["a"] if s["a"] is not None else s["y"]
            tok = self.mask_tokens[i].detach().reshape(-1).contiguous()
            if i > 0 and into_cats:
                c = cats[nf - 1 - i]                             # feats[i] is the skip of up_conv{nf - i}
                d = Act(c["buf"], c["Cup"], c["Cskip"])
            else:
                d = eng._new(a.B, a.H, a.W, a.C)
            if s["a"] is not None:
                ops.mask_select(a, active, d, relu=False, fill=tok, use_transform=False)
            else:
                ops.mask_select(a, active, d, relu=True, fill=tok, use_transform=True)      # BN + ReLU at active positions, the token elsewhere
            to_dec.append(d)

        # ---- dense decoder (decoder.py:49-55) + loss (spark.py:112-123) ----
        skips = list(reversed(to_dec[1:]))                       # skips[i-1] belongs to up_conv{i}
        dctx = eng.decoder_forward(sd, to_dec[0], skips, training, dp, cats if into_cats else None, True)
        rec = dctx["logits"]                                     # (B,1,H,W)
        if self.keep_rec:
            self.last_rec = rec
        if getattr(self, "keep_ctx", False):                     # test hook (tests/test_gpu_pretrain.py: float64 backward on this forward's own gates)
            self.last_ctx = {"levels": levels, "b1": b1, "b2": b2, "dctx": dctx, "active": active}
        loss = torch.empty(1, dtype=torch.float32, device=eng.device)
        drec = torch.empty_like(rec) if need_grads else None
        ws = eng.scratch.get("sploss", eng.lib.cmu_spark_loss_ws_bytes(B, f))
        ops.spark_loss_fwd_bwd(rec.view(B, H, W), x_img, active, loss, drec, float(self.grad_scale), r, ws)
        if not need_grads:
            return loss[0], None

        # ---- backward ----
        grads = {}
        # a data-parallel trainer is told as soon as a sub-network's gradients are final (pretrain.ArenaTrainer.notify_ready)
        ready = getattr(self, "_grads_ready", None)
        d_lat, d_skips = eng.decoder_backward(sd, dctx, drec, grads, True)
        if ready is not None:
            ready("dense_decoder.", grads)
        d_feats = [d_lat] + list(reversed(d_skips))              # same order as feats / mask_tokens
        for i, d in enumerate(d_feats):
            dv = Act(d.buf, d.coff, d.C)
            if ops.cells_supported(dv, active):                                    # sum over the pixels of the MASKED patches
                g = eng._gbuf(f"mask_tokens.{i}", self.mask_tokens[i])
                ops.cells_channel_sum(dv, active, g.view(-1), invert=True, ws=eng.scratch.get("csum", eng.lib.cmu_cells_channel_sum_ws_bytes(d.C)))
                grads[f"mask_tokens.{i}"] = g
            else:
                slab = ops.masked_channel_stats(dv, active, invert=True)
                grads[f"mask_tokens.{i}"] = slab[:, 0, :].sum(0).view_as(self.mask_tokens[i])
        dA = self._sp_convbn_bwd(eng, sd, b2, d_feats[0], active, cnt_b, grads, True)
        dP = self._sp_convbn_bwd(eng, sd, b1, dA, active, cnt_b, grads, nd > 0)
        if ready is not None:
            eng.flush_zero_bias()
            ready("sparse_encoder.sp_cnn.double_conv.", grads)
        for i in range(nd, 0, -1):
            lv = levels[i - 1]
            a2 = lv["s2"]["a"]
            y2 = lv["s2"]["y"]
            dA2 = eng._new(y2.B, y2.H, y2.W, y2.C)
            if a2 is None:
                ops.maxpool_bwd_masked(dP, d_feats[nd - i + 1], y2, dA2, active)   # arg-max from the raw output + transform, active windows only
            else:
                one, zero = self._ident(eng, a2.C)
                ops.maxpool_bwd(dP, d_feats[nd - i + 1], a2.with_transform(one, zero, 0), dA2)
            dA1 = self._sp_convbn_bwd(eng, sd, lv["s2"], dA2, active, lv["cnt"], grads, True)
            dP = self._sp_convbn_bwd(eng, sd, lv["s1"], dA1, active, lv["cnt"], grads, i > 1)
        eng.flush_zero_bias()
        if ready is not None:
            ready("sparse_encoder.sp_cnn.down_conv", grads)
        return loss[0], grads
