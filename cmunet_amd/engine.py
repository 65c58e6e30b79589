"""UNet forward/backward engine: sequences the HIP kernels (ops.py) for the reference's UNet topology.

What the reference expresses as ~60 nn.Module calls per forward (Finetuning/model.py:110-131) becomes a
fixed kernel schedule over NHWC buffers:

  * every Conv3x3 writes its RAW (pre-BatchNorm) output plus per-tile channel statistics; a tiny finalize
    kernel turns them into a pending transform (scale, shift) that the CONSUMER applies while loading
    (next conv, max-pool, conv-transpose, 1x1 head) -- BatchNorm+ReLU never make their own pass over HBM;
  * torch.cat of UpBlock (model.py:80) never happens: the encoder's second conv writes its raw output
    straight into the right half of the decoder's concat buffer and the ConvTranspose pixel-shuffles into
    the left half; the concat's pending transform is (1, 0 | scale, shift) with relu_from = C;
  * backward mirrors it: BN+ReLU backward = one reduce + one apply pass, data gradients reuse the forward
    implicit-GEMM kernel with flipped packed weights, weight gradients use the transposed-read MFMA kernel.

The engine works on a dict of named tensors with the reference's ``state_dict`` key names, so the same code
serves ``UNet`` (model.py), ``UNet_encoder`` (UNet_encoder.py:51-84) and ``MUNetPretrainDecoder``
(munet_neck.py:52-82).  There is no eager/CPU fallback here: every arithmetic step is a C-ABI call.
"""
import os
import weakref

import torch

from . import _lib, ops
from .ops import Act
from .optim import claim_grad_sink

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


_FUSE_HEAD = os.environ.get("CMU_HEAD_FUSE", "1") != "0"   # A/B: "0" = the head's input gradient is stored and re-read by cmu_bn_bwd_apply
_FUSE_DGRAD_BN = os.environ.get("CMU_DGRAD_BN", "1")    # A/B: "0" = BN-backward sums as a separate pass, "64"/"128" = fused up to C
_FUSE_POOL = os.environ.get("CMU_POOL_FUSE", "1") != "0"   # A/B: "0" = the pool's input gradient is stored and re-read by cmu_bn_bwd_apply
_POISON_NEW = os.environ.get("CMU_POISON_NEW", "0") == "1"    # tests: fresh activations are filled with NaN (reads of unwritten positions show)
_FUSE_POOL2 = os.environ.get("CMU_POOL_FUSE2", "1") != "0"   # A/B: "0" = with two skip gradients (joint model) the pool's input gradient is stored
_C1W_RECOMP = os.environ.get("CMU_C1W_RECOMP", "1") != "0"  # A/B: "0" = the first layer's weight gradient reads the raw output instead of recomputing it


class _Scratch:
    """Byte workspaces cached by purpose (grown on demand, reused across steps)."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, key, nbytes):
        nbytes = max(int(nbytes), 16)
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.bufs[key] = b
        return b


class _PackCache:
    """Packed (MFMA-layout) copies of conv weights, refreshed when the parameter's version changes."""

    def __init__(self):
        self.items = {}

    def get(self, key, param, fn):
        ver = (param._version, ops.PARAM_GENERATION)
        hit = self.items.get(key)
        if hit is not None and hit[0] == ver and hit[1] == param.data_ptr():
            return hit[2]
        packed = fn()
        self.items[key] = (ver, param.data_ptr(), packed)
        return packed


class _Ctx(dict):
    """Saved state of a forward pass (a dict that can be weakly referenced: the engine watches whether the training
    forward that owns the cached concat buffers is still waiting for its backward)."""
    __slots__ = ("__weakref__",)


class UNetEngine:
    def __init__(self, dt="bf16", device="cuda"):
        self.dt = ops.dt_code(dt)
        self.tdt = ops.TORCH_DT[self.dt]
        self.device = torch.device(device)
        self.scratch = _Scratch(self.device)
        self.packs = _PackCache()
        self.lib = _lib.lib()
        self._zero_pending = []
        self._cats, self._cats_busy, self._cats_owner = None, False, None
        self._nbt, self._nbt_defer = [], 0
        self.grad_target = None      # optional dict name -> preallocated fp32 tensor (FlatParams.grad_views)
        self.grad_prefix = ""

    def _gbuf(self, name, like):
        """Where a parameter gradient is written: the caller's arena view if given, else a fresh tensor."""
        if self.grad_target is not None:
            t = self.grad_target.get(self.grad_prefix + name)
            if t is not None:
                return t
        else:
            t = claim_grad_sink(like)     # the autograd path: straight into the arena of the trainer that holds the parameter, if any
            if t is not None and t.device == self.device:
                return t
        return torch.empty(like.shape, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    def _new(self, B, H, W, C):
        if _POISON_NEW:      # tests: every fresh activation starts as NaN, so a kernel that reads a position nobody wrote shows up
            return Act(torch.full((B, H, W, C), float("nan"), dtype=self.tdt, device=self.device))
        return Act(torch.empty((B, H, W, C), dtype=self.tdt, device=self.device))

    def _f32(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def prepack(self, sd):
        """Repack every 3x3 / conv-transpose weight of ``sd`` in one launch and mark the per-layer caches fresh (a
        trainer calls this once per step, after the optimiser rewrote the parameter arena)."""
        plan = getattr(self, "_pack_plan", None)
        if plan is None or plan[0] != tuple((k, v.data_ptr()) for k, v in sd.items() if k.endswith("weight") and v.dim() == 4):
            keys, items = [], []
            for k, v in sd.items():
                if not (k.endswith("weight") and v.dim() == 4):
                    continue
                name = k[:-len("weight")]
                if v.shape[2:] == (3, 3) and v.shape[1] > 1:
                    for flip in (False, True):
                        keys.append((name, flip)); items.append((v.detach(), 0, int(flip)))
                elif v.shape[2:] == (2, 2):
                    for mode in (0, 1):
                        keys.append((name, "T", mode)); items.append((v.detach(), 1, mode))
            if not items:
                return
            sig = tuple((k, v.data_ptr()) for k, v in sd.items() if k.endswith("weight") and v.dim() == 4)
            plan = (sig, keys, ops.PackPlan(items, self.dt), [sd[k[0] + "weight"] for k in keys])
            self._pack_plan = plan
        _, keys, pp, params = plan
        stamp = (ops.PARAM_GENERATION, tuple(prm._version for prm in params))
        if getattr(self, "_pack_stamp", None) == stamp and all(k in self.packs.items for k in keys):
            return                                  # nothing was updated since the last pack (evaluation loops, frozen encoders)
        self._pack_stamp = stamp
        pp.run()
        for key, out, prm in zip(keys, pp.outs, params):
            self.packs.items[key] = ((prm._version, ops.PARAM_GENERATION), prm.data_ptr(), out)

    def _wp(self, name, w, flip):
        return self.packs.get((name, flip), w, lambda: ops.pack_conv3x3(w.detach(), self.dt, flip))

    def _wpT(self, name, w, mode):
        return self.packs.get((name, "T", mode), w, lambda: ops.pack_convT2x2(w.detach(), self.dt, mode))

    # one Conv3x3 + BatchNorm2d(+ReLU pending): returns the saved state needed by backward
    def flush_counters(self):
        """``num_batches_tracked += 1`` of every BatchNorm the forward went through, as ONE multi-tensor launch (18 one-element
        launches per UNet forward otherwise)."""
        if self._nbt_defer > 0:
            return
        pend, self._nbt = self._nbt, []
        if pend:
            torch._foreach_add_(pend, 1)

    def flush_zero_bias(self):
        """Zero the conv-bias gradients collected by ``_convbn_bwd`` in one multi-tensor launch."""
        pend, self._zero_pending = self._zero_pending, []
        if pend:
            torch._foreach_zero_(pend)

    def _convbn_fwd(self, sd, pconv, pbn, x, out, training, x_img=None, mask=None, mask_per_sample=False, affine_out=None):
        """``affine_out``: optional (scale, shift) views the pending transform is written into (a skip that lives in a concat
        buffer gets the buffer's own arrays: no copy later)."""
        w = sd[pconv + "weight"]
        Cout = w.shape[0]
        B, H, W = out.B, out.H, out.W
        stats = ops.new_stats(B, H, W, Cout, self.device) if training else None
        if x_img is not None:
            ops.conv3x3_c1_fwd(x_img, w.detach(), out, stats, mask, mask_per_sample)
        else:
            ops.conv3x3_fwd(x, self._wp(pconv, w, False), out, stats)
        scale, shift = affine_out if affine_out is not None else (self._f32(Cout), self._f32(Cout))
        mean, invstd = self._f32(Cout), self._f32(Cout)
        ws = self.scratch.get("bnfin", self.lib.cmu_bn_finalize_ws_bytes(Cout))
        ops.bn_finalize(stats, B * H * W, sd[pconv + "bias"].detach(), sd[pbn + "weight"].detach(),
                        sd[pbn + "bias"].detach(), sd[pbn + "running_mean"], sd[pbn + "running_var"], BN_MOMENTUM,
                        BN_EPS, training, scale, shift, mean, invstd, ws)
        if training and (pbn + "num_batches_tracked") in sd:
            self._nbt.append(sd[pbn + "num_batches_tracked"])
        return {"pconv": pconv, "pbn": pbn, "x": x, "x_img": x_img, "mask": mask, "mps": mask_per_sample,
                "y": out.with_transform(scale, shift, 0), "mean": mean, "invstd": invstd,
                # (ops.PARAM_GENERATION: the fused optimisers / EMA / broadcast rewrite the arena through raw pointers, no version bump)
                "w_ver": (w._version, w.data_ptr(), ops.PARAM_GENERATION) if x_img is not None else None}

    def _double_conv_fwd(self, sd, prefix, x, out2, training, x_img=None, mask=None, mask_per_sample=False, affine_out2=None):
        """DoubleConv (model.py:16-26).  ``out2``: where the second conv writes its raw output."""
        w1 = sd[prefix + "0.weight"]
        B, H, W = out2.B, out2.H, out2.W
        y1 = self._new(B, H, W, w1.shape[0])
        s1 = self._convbn_fwd(sd, prefix + "0.", prefix + "1.", x, y1, training, x_img, mask, mask_per_sample)
        s2 = self._convbn_fwd(sd, prefix + "3.", prefix + "4.", s1["y"], out2, training, affine_out=affine_out2)
        self.flush_counters()            # (a no-op inside encoder_forward / decoder_forward, which flush once at their end)
        return s1, s2

    # backward of one Conv3x3+BN+ReLU given dA (gradient w.r.t. the activated output); returns dX act or None
    def _bn_ws(self, C):
        return self.scratch.get("bnbwd", self.lib.cmu_bn_bwd_ws_bytes(C))

    def _tile_slab(self, y):
        """Shared per-tile statistics slab for the data-gradient kernels that fuse the next BN-backward reduction."""
        n = ops.ntiles(y.B, y.H, y.W)
        return self.scratch.get("bst", n * 2 * y.C * 4)[:n * 2 * y.C * 4].view(torch.float32).view(n, 2, y.C)

    def _convbn_bwd(self, sd, s, dA, grads, need_dx, dx_out=None, fused_stats=False, next_bn=None, head=None, pool=None):
        """``fused_stats``: the kernel that produced dA already wrote this layer's BN-backward partial sums into
        the shared slab (max-pool / head backward) -- only the finalisation is left of phase 1.  The same holds when the
        producer was a data-gradient kernel, which leaves a per-tile slab in ``s["bstats"]``.
        ``next_bn``: saved state of the conv+BN layer whose activated output is this conv's input: its BN-backward
        partial sums are produced by this layer's data-gradient kernel (consumed by the next ``_convbn_bwd`` call).
        ``head``: (dlogits, w) of the 1x1 head this layer feeds, with ``dA`` None: the head's rank-K input gradient was never
        stored (its BN-backward sums were: ``fused_stats``) -- dY comes straight from dlogits (``conv1x1_head_bn_apply``).
        ``pool``: (dP, dSkip, dSkip2) of the max-pool this layer feeds, with ``dA`` None: the pool's input gradient was never stored
        either (``maxpool_bwd(dA=None)`` left the sums) -- dY comes from ``maxpool_bwd_apply``."""
        y = s["y"]
        B, H, W, C = y.B, y.H, y.W, y.C
        w = sd[s["pconv"] + "weight"]
        dgamma, dbeta = self._gbuf(s["pbn"] + "weight", sd[s["pbn"] + "weight"]), self._gbuf(s["pbn"] + "bias", sd[s["pbn"] + "bias"])
        coef = self._f32(2, C)
        ws = self._bn_ws(C)
        bst = s.pop("bstats", None)
        if bst is not None:
            ops.bn_bwd_finalize_tiles(bst, B * H * W, dgamma, dbeta, coef,
                                      self.scratch.get("bnfin", self.lib.cmu_bn_finalize_ws_bytes(C)))
        elif fused_stats:
            ops.bn_bwd_finalize(ws, B * H * W, dgamma, dbeta, coef)
        else:
            ops.bn_bwd_reduce(dA, y, s["mean"], s["invstd"], dgamma, dbeta, coef, ws)
        first = s["x_img"] is not None                        # first layer: no data gradient, BN apply fused in its wgrad
        if head is not None:
            assert dA is None and fused_stats and not first
            dY = self._new(B, H, W, C)
            ops.conv1x1_head_bn_apply(head[0], y, head[1], s["mean"], s["invstd"], coef, dY)
        elif pool is not None:
            assert dA is None and fused_stats and not first
            dY = self._new(B, H, W, C)
            ops.maxpool_bwd_apply(pool[0], pool[1], y, s["mean"], s["invstd"], coef, dY, dSkip2=pool[2])
        else:
            dY = Act(dA.buf, dA.coff, dA.C)                   # in place over dA (out of place measures the same 4.3 ms)
            if not first:
                ops.bn_bwd_apply(dA, y, s["mean"], s["invstd"], coef, dY)
        grads[s["pbn"] + "weight"] = dgamma
        grads[s["pbn"] + "bias"] = dbeta
        # conv bias: followed by training-mode BN, its gradient is identically zero (sum of dY over pixels)
        # (zeroed every step -- the arena is also written by the autograd / drop-in path -- but as ONE multi-tensor fill for all
        # layers of the pass, see flush_zero_bias)
        gb = self._gbuf(s["pconv"] + "bias", sd[s["pconv"] + "bias"])
        self._zero_pending.append(gb)
        grads[s["pconv"] + "bias"] = gb
        dW = self._gbuf(s["pconv"] + "weight", w)
        if s["x_img"] is not None:
            wsb = self.scratch.get("wg", self.lib.cmu_conv3x3_c1_wgrad_ws_bytes(B, H, W, C))
            # the raw output is recomputed from the image when the weights are still the ones the forward used (same bits, half
            # the traffic of the pass); a parameter rewritten between forward and backward falls back to the stored tensor
            same_w = (_C1W_RECOMP and s.get("w_ver") == (w._version, w.data_ptr(), ops.PARAM_GENERATION)
                      and getattr(self.lib, "cmu_conv3x3_c1_wgrad_bn_w", None) is not None)    # (older builds under CMU_LIB_PATH: A/B runs)
            ops.conv3x3_c1_wgrad_bn(s["x_img"], dA, y, y.scale, y.shift, s["mean"], s["invstd"], coef, dW, wsb, s["mask"], s["mps"],
                                    w=w.detach() if same_w else None)
        else:
            Cin = w.shape[1]
            wsb = self.scratch.get("wg", self.lib.cmu_conv3x3_wgrad_ws_bytes(B, H, W, Cin, C, self.dt))
            ops.conv3x3_wgrad(s["x"], dY, dW, wsb)
        grads[s["pconv"] + "weight"] = dW
        if not need_dx or first:
            return None
        dX = dx_out if dx_out is not None else self._new(B, H, W, w.shape[1])
        if next_bn is not None and next_bn["y"].C == dX.C and (_FUSE_DGRAD_BN == "1" or (_FUSE_DGRAD_BN == "64" and dX.C <= 64)
                                                                or (_FUSE_DGRAD_BN == "128" and dX.C <= 128)):
            slab = self._tile_slab(next_bn["y"])
            ops.conv3x3_dgrad_bn(dY, self._wp(s["pconv"], w, True), dX, next_bn["y"], next_bn["mean"], next_bn["invstd"], slab)
            next_bn["bstats"] = slab
        else:
            ops.conv3x3_fwd(dY, self._wp(s["pconv"], w, True), dX, None)
        return dX

    # ------------------------------------------------------------------------------------------
    # encoder (model.py:121-125, UNet_encoder.py:79-83)
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def n_down(sd, prefix=""):
        n = 0
        while f"{prefix}down_conv{n + 1}.double_conv.double_conv.0.weight" in sd:
            n += 1
        return n

    def encoder_forward(self, sd, x_bhw, training, prefix="", mask=None, mask_per_sample=False, skip_out=None, skip_affine=None):
        """x (B,H,W) fp32 cuda.  ``skip_out[i]``: optional Act (right half of a concat buffer) where level i's
        raw skip is written.  Returns ctx with 'latent' (raw Act + pending transform) and 'skips'."""
        B, H, W = x_bhw.shape
        nd = self.n_down(sd, prefix)
        assert H % (1 << nd) == 0 and W % (1 << nd) == 0, f"H,W must be multiples of {1 << nd}"
        ctx = {"levels": [], "prefix": prefix}
        self._nbt_defer += 1
        x_act, x_img = None, x_bhw.contiguous()
        h, w_ = H, W
        for i in range(1, nd + 1):
            p = f"{prefix}down_conv{i}.double_conv.double_conv."
            C = sd[p + "0.weight"].shape[0]
            out2 = skip_out[i - 1] if skip_out is not None else self._new(B, h, w_, C)
            s1, s2 = self._double_conv_fwd(sd, p, x_act, out2, training, x_img, mask if i == 1 else None, mask_per_sample,
                                           affine_out2=skip_affine[i - 1] if skip_affine is not None else None)
            pooled = self._new(B, h // 2, w_ // 2, C)
            ops.bnrelu_maxpool_fwd(s2["y"], pooled)
            ctx["levels"].append({"s1": s1, "s2": s2, "pooled": pooled})
            x_act, x_img = pooled, None
            h, w_ = h // 2, w_ // 2
        p = f"{prefix}double_conv.double_conv."
        C = sd[p + "0.weight"].shape[0]
        s1, s2 = self._double_conv_fwd(sd, p, x_act, self._new(B, h, w_, C), training, x_img, mask if nd == 0 else None, mask_per_sample)
        ctx["bott"] = {"s1": s1, "s2": s2}
        ctx["latent"] = s2["y"]
        ctx["skips"] = [lv["s2"]["y"] for lv in ctx["levels"]]
        self._nbt_defer -= 1
        self.flush_counters()
        return ctx

    def encoder_backward(self, sd, ctx, d_latent, d_skips, grads, after_bottleneck=None):
        """d_latent: Act (gradient w.r.t. the activated latent); d_skips[i]: Act or None.  ``after_bottleneck``: called when
        the bottleneck's parameter gradients (60 % of the encoder's parameters) have been queued."""
        b = ctx["bott"]
        dA = self._convbn_bwd(sd, b["s2"], d_latent, grads, True, next_bn=b["s1"])
        dP = self._convbn_bwd(sd, b["s1"], dA, grads, len(ctx["levels"]) > 0)
        if after_bottleneck is not None:
            self.flush_zero_bias()
            after_bottleneck()
        for i in range(len(ctx["levels"]), 0, -1):
            lv = ctx["levels"][i - 1]
            y2 = lv["s2"]["y"]
            ds = d_skips[i - 1] if d_skips is not None else None
            ds, ds2 = ds if isinstance(ds, (tuple, list)) else (ds, None)     # two decoders on this encoder: both gradients of the skip
            if _FUSE_POOL and (ds2 is None or _FUSE_POOL2):
                # the pool's input gradient is never stored: sums first, dY recomputed from (dP, skip gradient) after their finalisation
                # -- one tensor pass less; with two skip gradients (the joint model's two decoders) 6.5 reads + 1 write instead of
                # 6.25 + 2: 74.91 -> 74.67 ms per joint step over three same-box pairs (round 4; CMU_POOL_FUSE2=0: the stored form)
                ops.maxpool_bwd(dP, ds, y2, None, lv["s2"]["mean"], lv["s2"]["invstd"], self._bn_ws(y2.C), dSkip2=ds2)
                dA1 = self._convbn_bwd(sd, lv["s2"], None, grads, True, fused_stats=True, next_bn=lv["s1"], pool=(dP, ds, ds2))
            else:
                dA2 = self._new(y2.B, y2.H, y2.W, y2.C)
                ops.maxpool_bwd(dP, ds, y2, dA2, lv["s2"]["mean"], lv["s2"]["invstd"], self._bn_ws(y2.C), dSkip2=ds2)
                dA1 = self._convbn_bwd(sd, lv["s2"], dA2, grads, True, fused_stats=True, next_bn=lv["s1"])
            dP = self._convbn_bwd(sd, lv["s1"], dA1, grads, i > 1)
        self.flush_zero_bias()
        return None

    # ------------------------------------------------------------------------------------------
    # decoder (model.py:126-130, munet_neck.py:75-82)
    # ------------------------------------------------------------------------------------------
    def decoder_alloc(self, sd, B, H, W, prefix=""):
        """Concat buffers + their pending-transform arrays, one per up level (index i-1 for up_conv{i})."""
        cats = []
        i = 1
        h, w_ = H, W
        while f"{prefix}up_conv{i}.double_conv.double_conv.0.weight" in sd:
            wt = sd[f"{prefix}up_conv{i}.double_conv.double_conv.0.weight"]
            Cout, C2 = wt.shape[0], wt.shape[1]
            Cs = C2 - Cout if f"{prefix}up_conv{i}.up_sample.weight" not in sd else sd[f"{prefix}up_conv{i}.up_sample.weight"].shape[1]
            buf = self._new(B, h, w_, C2).buf
            scale = torch.ones(C2, dtype=torch.float32, device=self.device)
            shift = torch.zeros(C2, dtype=torch.float32, device=self.device)
            cats.append({"buf": buf, "scale": scale, "shift": shift, "Cup": Cs, "Cskip": C2 - Cs})
            i += 1
            h, w_ = h // 2, w_ // 2
        return cats

    def decoder_alloc_shared(self, sd, B, H, W, prefix_a, prefix_b):
        """Concat buffers of TWO decoders that read the same skips (the joint model's pixel and feature decoders): per level ONE
        buffer of channels [up_a | skip | up_b] and one pair of pending-transform arrays of the same layout.  Decoder a sees the
        usual view (up, skip) at channel 0; decoder b sees (skip, up) at channel Cup_a -- ``skip_first``: its first conv runs on
        weights whose input channels are rotated accordingly (``rotated_weights``), ReLU applies to the view's FIRST Cskip channels
        (a negative ``relu_from``).  The skip is written once, by the encoder; no copy (round 4: four strided copies, 0.66 ms per step).
        Returns (cats_a, cats_b), or None when the two decoders' levels do not match."""
        ca, cb = [], []
        i = 1
        h, w_ = H, W
        while f"{prefix_a}up_conv{i}.double_conv.double_conv.0.weight" in sd:
            wa = sd[f"{prefix_a}up_conv{i}.double_conv.double_conv.0.weight"]
            wb = sd.get(f"{prefix_b}up_conv{i}.double_conv.double_conv.0.weight")
            ka, kb = f"{prefix_a}up_conv{i}.up_sample.weight", f"{prefix_b}up_conv{i}.up_sample.weight"
            if wb is None or ka not in sd or kb not in sd:
                return None
            Cup_a, Cup_b = sd[ka].shape[1], sd[kb].shape[1]
            Cs = wa.shape[1] - Cup_a
            epc = 16 // torch.empty(0, dtype=self.tdt).element_size()
            if Cs <= 0 or wb.shape[1] - Cup_b != Cs or any(c % epc for c in (Cup_a, Cup_b, Cs)):
                return None
            C3 = Cup_a + Cs + Cup_b
            buf = self._new(B, h, w_, C3).buf
            scale = torch.ones(C3, dtype=torch.float32, device=self.device)
            shift = torch.zeros(C3, dtype=torch.float32, device=self.device)
            ca.append({"buf": buf, "scale": scale[:Cup_a + Cs], "shift": shift[:Cup_a + Cs], "Cup": Cup_a, "Cskip": Cs})
            cb.append({"buf": buf, "scale": scale[Cup_a:], "shift": shift[Cup_a:], "Cup": Cup_b, "Cskip": Cs, "base": Cup_a, "skip_first": True})
            i += 1
            h, w_ = h // 2, w_ // 2
        return ca, cb

    def rotated_weights(self, sd, prefix, cats):
        """Overlay of ``sd`` for a ``skip_first`` decoder: the first conv of every level with its input channels rotated from the
        reference's (up, skip) order to the view's (skip, up) -- persistent buffers, refreshed when the parameter changes (one
        concatenating copy of the weight per level and step: 25 MB in all for the reference UNet)."""
        store = self.__dict__.setdefault("_rot_w", {})
        over = dict(sd)
        for i, cat in enumerate(cats, start=1):
            if not cat.get("skip_first"):
                continue
            k = f"{prefix}up_conv{i}.double_conv.double_conv.0.weight"
            w = sd[k]
            Cup = cat["Cup"]
            ent = store.get(k)
            if ent is None or ent[0].shape != w.shape or ent[0].device != w.device:
                ent = [torch.empty_like(w, dtype=torch.float32), None]
                store[k] = ent
            ver = (w._version, w.data_ptr(), ops.PARAM_GENERATION)
            if ent[1] != ver:
                torch.cat((w.detach()[:, Cup:], w.detach()[:, :Cup]), 1, out=ent[0])
                ent[1] = ver
            over[k] = ent[0]
        return over

    def unrotate_grads(self, sd, prefix, cats, grads):
        """The weight gradients a ``skip_first`` decoder's first convs produced (in the view's channel order) back into the
        reference's order, written where the parameter's gradient belongs (``_gbuf``)."""
        for i, cat in enumerate(cats, start=1):
            if not cat.get("skip_first"):
                continue
            k = f"{prefix}up_conv{i}.double_conv.double_conv.0.weight"
            g_rot = grads.get(k)
            if g_rot is None:
                continue
            Cs = cat["Cskip"]
            g = self._gbuf(k, sd[k])
            if g.data_ptr() == g_rot.data_ptr():          # (an explicit grad_target maps the name to the parameter's own slot)
                g_rot = g_rot.clone()
            torch.cat((g_rot[:, Cs:], g_rot[:, :Cs]), 1, out=g)
            grads[k] = g

    def decoder_forward(self, sd, latent, skips, training, prefix="", cats=None, head=True):
        """latent: Act with pending transform; skips[i-1]: Act with pending transform (level i).
        If ``cats`` is given and a skip already lives in cats[i-1]['buf'] (fused UNet), no copy is made."""
        B = latent.B
        nup = len(skips)
        if cats is None:
            cats = self.decoder_alloc(sd, B, skips[0].H, skips[0].W, prefix)
        ctx = {"levels": [None] * nup, "prefix": prefix, "cats": cats}
        self._nbt_defer += 1
        x = latent
        for i in range(nup, 0, -1):
            p = f"{prefix}up_conv{i}."
            cat = cats[i - 1]
            Cup, Cs = cat["Cup"], cat["Cskip"]
            sk = skips[i - 1]
            assert sk.C == Cs
            # the view of this decoder inside the buffer: channels [base, base + Cup + Cs), (up, skip) -- or (skip, up) for the second
            # of two decoders sharing their skips (decoder_alloc_shared)
            base, sf = cat.get("base", 0), bool(cat.get("skip_first", False))
            up_off, sk_off = base + (Cs if sf else 0), base + (0 if sf else Cup)
            s_lo = 0 if sf else Cup                                   # the skip's place in the view's pending-transform arrays
            if not (sk.buf is cat["buf"] and sk.coff == sk_off):
                cat["buf"][..., sk_off:sk_off + Cs].copy_(sk.buf[..., sk.coff:sk.coff + sk.C])   # split encoder/decoder: one strided copy
            if sk.scale is not None:
                if sk.scale.data_ptr() != cat["scale"][s_lo:].data_ptr():
                    cat["scale"][s_lo:s_lo + Cs].copy_(sk.scale)
                    cat["shift"][s_lo:s_lo + Cs].copy_(sk.shift)
                cat["ident"] = False        # (also when the producer wrote scale / shift in place: the arrays no longer hold the identity)
                relu_from = -Cs if sf else Cup
            else:                      # already-activated skip handed over at a module boundary: identity, no ReLU
                if not cat.get("ident", False):      # (buffers reused from step to step keep the identity: two fills per level saved)
                    cat["scale"][s_lo:s_lo + Cs].fill_(1.0)
                    cat["shift"][s_lo:s_lo + Cs].zero_()
                    cat["ident"] = True
                relu_from = Cup + Cs
            wt = sd[p + "up_sample.weight"]
            left = Act(cat["buf"], up_off, Cup)
            ops.convT2x2_fwd(x, self._wpT(p + "up_sample.", wt, 0), sd[p + "up_sample.bias"].detach(), left)
            cat_act = Act(cat["buf"], base, Cup + Cs, cat["scale"], cat["shift"], relu_from)
            Cout = sd[p + "double_conv.double_conv.0.weight"].shape[0]
            s1, s2 = self._double_conv_fwd(sd, p + "double_conv.double_conv.", cat_act, self._new(B, sk.H, sk.W, Cout), training)
            ctx["levels"][i - 1] = {"s1": s1, "s2": s2, "x_up": x, "cat": cat}
            x = s2["y"]
        ctx["out"] = x
        if head:
            K = sd[prefix + "conv_last.weight"].shape[0]
            logits = self._f32(B, K, x.H, x.W)
            ops.conv1x1_head_fwd(x, sd[prefix + "conv_last.weight"].detach().reshape(K, -1), sd[prefix + "conv_last.bias"].detach(), logits)
            ctx["logits"] = logits
        self._nbt_defer -= 1
        self.flush_counters()
        return ctx

    def decoder_backward(self, sd, ctx, dlogits, grads, need_input_grads=True, latent_bn=None):
        """Returns (d_latent Act, [d_skip Acts]) -- d_skip[i-1] is the right half of level i's concat gradient.
        ``latent_bn``: saved state of the conv+BN layer that produced the latent, when d_latent goes straight into its
        backward (no other gradient is added to it): the deepest ConvTranspose's data-gradient kernel then also produces
        that layer's BN-backward partial sums."""
        prefix = ctx["prefix"]
        x = ctx["out"]
        K = sd[prefix + "conv_last.weight"].shape[0]
        wl = sd[prefix + "conv_last.weight"]
        dWl, dbl = self._gbuf(prefix + "conv_last.weight", wl), self._gbuf(prefix + "conv_last.bias", sd[prefix + "conv_last.bias"])
        ws = self.scratch.get("head", self.lib.cmu_conv1x1_head_bwd_ws_bytes(x.B, x.H, x.W, x.C, K))
        last = ctx["levels"][0]["s2"] if ctx["levels"] else None      # the conv+BN layer that produced ``x``
        fused = last is not None and last["y"] is x
        # the head's input gradient has rank K: when it flows straight into ``last``'s BatchNorm backward it is never stored -- the
        # head backward leaves the BN-backward sums, the apply pass recomputes it from dlogits (CMU_HEAD_FUSE=0: stored, A/B switch)
        dl, wl2 = dlogits.contiguous(), wl.detach().reshape(K, -1)
        head = (dl, wl2) if (fused and _FUSE_HEAD) else None
        dA = None if head is not None else self._new(x.B, x.H, x.W, x.C)
        ops.conv1x1_head_bwd(dl, x, wl2, dA, dWl.view(K, -1), dbl, ws,
                             last["mean"] if fused else None, last["invstd"] if fused else None, self._bn_ws(x.C) if fused else None)
        grads[prefix + "conv_last.weight"] = dWl
        grads[prefix + "conv_last.bias"] = dbl
        nup = len(ctx["levels"])
        d_skips = [None] * nup
        for i in range(1, nup + 1):
            lv = ctx["levels"][i - 1]
            p = f"{prefix}up_conv{i}."
            cat = lv["cat"]
            Cup, Cs = cat["Cup"], cat["Cskip"]
            dA1 = self._convbn_bwd(sd, lv["s2"], dA, grads, True, fused_stats=(fused and i == 1), next_bn=lv["s1"],
                                   head=head if i == 1 else None)
            dcat = self._convbn_bwd(sd, lv["s1"], dA1, grads, True)
            sf = bool(cat.get("skip_first", False))                  # the gradient has the view's channel order
            d_skips[i - 1] = Act(dcat.buf, 0 if sf else Cup, Cs)
            dleft = Act(dcat.buf, Cs if sf else 0, Cup)
            wt = sd[p + "up_sample.weight"]
            xu = lv["x_up"]
            dWt, dbt = self._gbuf(p + "up_sample.weight", wt), self._gbuf(p + "up_sample.bias", sd[p + "up_sample.bias"])
            wsb = self.scratch.get("wg", self.lib.cmu_convT2x2_wgrad_ws_bytes(xu.B, xu.H, xu.W, xu.C, Cup, self.dt))
            ops.convT2x2_wgrad(xu, dleft, dWt, dbt, wsb)
            grads[p + "up_sample.weight"] = dWt
            grads[p + "up_sample.bias"] = dbt
            if i < nup or need_input_grads:
                dA = self._new(xu.B, xu.H, xu.W, xu.C)
                # the layer that produced this ConvTranspose's input: the next decoder level's second conv, or the latent's
                tgt = ctx["levels"][i]["s2"] if i < nup else latent_bn
                if tgt is not None and tgt["y"].buf is xu.buf and tgt["y"].coff == xu.coff and tgt["y"].C == xu.C:
                    slab = self._tile_slab(tgt["y"])
                    ops.convT2x2_dgrad_bn(dleft, self._wpT(p + "up_sample.", wt, 1), dA, tgt["y"], tgt["mean"], tgt["invstd"], slab)
                    tgt["bstats"] = slab
                else:
                    ops.convT2x2_dgrad(dleft, self._wpT(p + "up_sample.", wt, 1), dA)
            else:
                dA = None
        self.flush_zero_bias()
        return dA, d_skips

    # ------------------------------------------------------------------------------------------
    # full UNet (model.py:110-131): encoder writes its skips straight into the decoder's concat buffers
    # ------------------------------------------------------------------------------------------
    def unet_forward(self, sd, x_bhw, training, mask=None, mask_per_sample=False):
        B, H, W = x_bhw.shape
        # concat buffers + affine arrays of the last shape are reused from step to step (no per-step fills) unless a
        # training forward is still waiting for its backward: that one keeps them, this call gets fresh ones
        key = (B, H, W, self.tdt)
        if self._cats_busy and (self._cats_owner is None or self._cats_owner() is None):
            self._cats_busy = False           # the forward that held them was dropped without a backward
        owns = False
        if self._cats_busy:
            cats = self.decoder_alloc(sd, B, H, W, "")
        else:
            if self._cats is None or self._cats[0] != key:
                self._cats = (key, self.decoder_alloc(sd, B, H, W, ""))
            cats = self._cats[1]
            self._cats_busy = bool(training)
            owns = True
        skip_out = [Act(c["buf"], c["Cup"], c["Cskip"]) for c in cats]
        skip_affine = [(c["scale"][c["Cup"]:], c["shift"][c["Cup"]:]) for c in cats]
        ectx = self.encoder_forward(sd, x_bhw, training, "", mask, mask_per_sample, skip_out, skip_affine)
        dctx = self.decoder_forward(sd, ectx["latent"], ectx["skips"], training, "", cats, True)
        ctx = _Ctx(enc=ectx, dec=dctx)
        if owns and training:
            self._cats_owner = weakref.ref(ctx)
        return dctx["logits"], ctx

    def unet_backward(self, sd, ctx, dlogits, after_decoder=None, after_bottleneck=None):
        """``after_decoder`` / ``after_bottleneck``: called when every decoder / bottleneck parameter gradient has been queued
        (data-parallel trainers start those buckets' all-reduce there, under the rest of the backward pass)."""
        grads = {}
        d_latent, d_skips = self.decoder_backward(sd, ctx["dec"], dlogits, grads, True, latent_bn=ctx["enc"]["bott"]["s2"])
        if after_decoder is not None:
            after_decoder()
        self.encoder_backward(sd, ctx["enc"], d_latent, d_skips, grads, after_bottleneck)
        if self._cats is not None and ctx["dec"]["cats"] is self._cats[1]:
            self._cats_busy = False
        return grads
