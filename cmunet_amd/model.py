"""Drop-in for the reference's ``Finetuning/model.py`` -- same classes, constructor arguments,
``forward`` signatures and ``state_dict`` key names/shapes -- executing on hand-written HIP kernels.

  DoubleConv(in_channels, out_channels)                 model.py:4-26
  DownBlock(in_channels, out_channels) -> (down, skip)  model.py:29-45
  UpBlock(in_channels, out_channels, up_sample_mode)    model.py:48-81
  UNet(out_classes=2, up_sample_mode='conv_transpose')  model.py:84-131

Build extensions: ``UNet(..., base_ch=64, depth=5, dtype='f32')`` -- the defaults reproduce
the reference structure (1->64->128->256->512->1024, 31 042 434 parameters) AND its arithmetic: the reference
finetunes in fp32 (Finetuning/train.py: no autocast), so ``dtype`` -- the storage / MFMA operand type of the
activations -- defaults to 'f32' (the only type at which Dice within 1e-4 of the reference is shown);
'f16' / 'bf16' are opt-in.  Parameters, statistics and logits are fp32 in every case.

Parameters live in ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` containers (so
initialisation, ``state_dict``, ``torch.save`` and the reference's checkpoint key maps work unchanged), but
their ``forward`` is never called: the module's forward runs the fused kernel schedule of ``engine.py``
inside one ``torch.autograd.Function``.  Inputs must be CUDA (ROCm) tensors -- there is no CPU fallback.
"""
import torch
import torch.nn as nn

from .engine import UNetEngine
from . import ops


def _require_cuda(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the MI355X HIP path needs a CUDA/ROCm tensor (got {x.device}); "
                           "there is no CPU fallback in this package")


def _named_state(module):
    sd = dict(module.named_parameters())
    sd.update(dict(module.named_buffers()))
    return sd


class _EngineOwner:
    """Lazily creates one UNetEngine per (module, dtype, device)."""

    def __getstate__(self):          # torch.save(model) (train.py:212) must not pickle device scratch / the CDLL
        d = dict(super().__getstate__())     # nn.Module.__getstate__ (this mixin precedes nn.Module in the MRO)
        d.pop("_eng", None)
        d.pop("_grads_ready", None)      # a trainer's callback (pretrain.ArenaTrainer.notify_ready)
        return d

    def _engine(self, device):
        eng = getattr(self, "_eng", None)
        if eng is None or eng.device != device or eng.dt != ops.dt_code(self.dtype):
            eng = UNetEngine(self.dtype, device)
            object.__setattr__(self, "_eng", eng)
        return eng


# ---------------------------------------------------------------------------------------------------
# autograd glue: one Function per module boundary; parameters are passed as inputs so that autograd
# routes the kernel-computed gradients into .grad
# ---------------------------------------------------------------------------------------------------
class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, mask, mask_per_sample, names, *params):
        eng = module._engine(x.device)
        sd = _named_state(module)
        eng.prepack(sd)
        training = module.training
        logits, saved = eng.unet_forward(sd, x.detach().float().contiguous(), training, mask, mask_per_sample)
        ctx.module, ctx.saved, ctx.names, ctx.eng = module, saved, names, eng
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module = ctx.module
        sd = _named_state(module)
        grads = ctx.eng.unet_backward(sd, ctx.saved, dlogits.contiguous().float())
        ctx.saved = None
        out = [grads.get(n) for n in ctx.names]
        return (None, None, None, None, None, *out)


def _param_args(module):
    names, params = [], []
    for n, p in module.named_parameters():
        names.append(n)
        params.append(p)
    return tuple(names), params


class DoubleConv(_EngineOwner, nn.Module):
    """[Conv3x3 -> BatchNorm2d -> ReLU] x 2 (model.py:4-26)."""

    def __init__(self, in_channels, out_channels, dtype="f32"):
        super().__init__()
        self.dtype = dtype
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )

    def forward(self, x):
        _require_cuda(x, "DoubleConv")
        names, params = _param_args(self)
        return _BlockFn.apply(self, "double", names, x, None, *params)


class DownBlock(_EngineOwner, nn.Module):
    """DoubleConv -> MaxPool2d(2); returns (down_out, skip_out) (model.py:29-45)."""

    def __init__(self, in_channels, out_channels, dtype="f32"):
        super().__init__()
        self.dtype = dtype
        self.double_conv = DoubleConv(in_channels, out_channels, dtype)
        self.down_sample = nn.MaxPool2d(2)

    def forward(self, x):
        _require_cuda(x, "DownBlock")
        names, params = _param_args(self)
        return _BlockFn.apply(self, "down", names, x, None, *params)


class UpBlock(_EngineOwner, nn.Module):
    """ConvTranspose2d(k2,s2) -> cat([up, skip], 1) -> DoubleConv (model.py:48-81)."""

    def __init__(self, in_channels, out_channels, up_sample_mode, dtype="f32"):
        super().__init__()
        self.dtype = dtype
        if up_sample_mode == 'conv_transpose':
            self.up_sample = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        elif up_sample_mode == 'bilinear':
            # the reference builds nn.Upsample here, but its DoubleConv(in_channels, ...) then receives
            # in_channels + out_channels channels and fails (model.py:62,65,80); the HIP path does not
            # implement a mode the reference cannot run.
            raise NotImplementedError("up_sample_mode='bilinear' is not runnable in the reference either "
                                      "(channel mismatch at model.py:80); use 'conv_transpose'")
        else:
            raise ValueError("Unsupported `up_sample_mode` (can take one of `conv_transpose` or `bilinear`)")
        self.double_conv = DoubleConv(in_channels, out_channels, dtype)

    def forward(self, down_input, skip_input):
        _require_cuda(down_input, "UpBlock")
        names, params = _param_args(self)
        return _BlockFn.apply(self, "up", names, down_input, skip_input, *params)


class _BlockFn(torch.autograd.Function):
    """Module-boundary execution of a single block: NCHW fp32 in/out (the reference's tensor contract)."""

    @staticmethod
    def forward(ctx, module, kind, names, x, skip, *params):
        eng = module._engine(x.device)
        sd = _named_state(module)
        tr = module.training
        B, Cin, H, W = x.shape
        ctx.kind, ctx.module, ctx.names, ctx.eng = kind, module, names, eng
        xin = x.detach().float().contiguous()
        if kind in ("double", "down"):
            p = "double_conv." if kind == "double" else "double_conv.double_conv."
            Cout = sd[p + "0.weight"].shape[0]
            if Cin == 1:
                x_act, x_img = None, xin.view(B, H, W)
            else:
                x_act, x_img = eng._new(B, H, W, Cin), None
                ops.nchw_to_nhwc(xin, x_act)
            s1, s2 = eng._double_conv_fwd(sd, p, x_act, eng._new(B, H, W, Cout), tr, x_img)
            ctx.saved = (s1, s2)
            out = ops.apply_to_nchw(s2["y"])
            if kind == "double":
                return out
            pooled = eng._new(B, H // 2, W // 2, Cout)
            ops.bnrelu_maxpool_fwd(s2["y"], pooled)
            down = ops.apply_to_nchw(pooled)
            return down, out
        # up block
        Bs, Cs, Hs, Ws = skip.shape
        Cup = sd["up_sample.weight"].shape[1]
        cat = torch.empty((B, Hs, Ws, Cup + Cs), dtype=eng.tdt, device=x.device)
        x_act = eng._new(B, H, W, Cin)
        ops.nchw_to_nhwc(xin, x_act)
        ops.nchw_to_nhwc(skip.detach().float().contiguous(), ops.Act(cat, Cup, Cs))
        wt = sd["up_sample.weight"]
        ops.convT2x2_fwd(x_act, eng._wpT("up_sample.", wt, 0), sd["up_sample.bias"].detach(), ops.Act(cat, 0, Cup))
        Cout = sd["double_conv.double_conv.0.weight"].shape[0]
        s1, s2 = eng._double_conv_fwd(sd, "double_conv.double_conv.", ops.Act(cat), eng._new(B, Hs, Ws, Cout), tr)
        ctx.saved = (s1, s2, x_act, Cup, Cs)
        return ops.apply_to_nchw(s2["y"])

    @staticmethod
    def backward(ctx, *gouts):
        eng, module, kind = ctx.eng, ctx.module, ctx.kind
        sd = _named_state(module)
        grads = {}
        s1, s2 = ctx.saved[0], ctx.saved[1]
        y2 = s2["y"]
        B, H, W, C = y2.B, y2.H, y2.W, y2.C
        if kind == "down":
            g_down, g_skip = gouts
            dP = eng._new(B, H // 2, W // 2, C)
            dS = eng._new(B, H, W, C)
            zeros = lambda t, shape: torch.zeros(shape, device=y2.buf.device) if t is None else t.contiguous().float()
            ops.nchw_to_nhwc(zeros(g_down, (B, C, H // 2, W // 2)), dP)
            ops.nchw_to_nhwc(zeros(g_skip, (B, C, H, W)), dS)
            dA2 = eng._new(B, H, W, C)
            ops.maxpool_bwd(dP, dS, y2, dA2)
        else:
            dA2 = eng._new(B, H, W, C)
            ops.nchw_to_nhwc(gouts[0].contiguous().float(), dA2)
        dA1 = eng._convbn_bwd(sd, s2, dA2, grads, True)
        first_has_dx = s1["x_img"] is None
        dX = eng._convbn_bwd(sd, s1, dA1, grads, first_has_dx)
        eng.flush_zero_bias()
        gx, gskip = None, None
        if kind == "up":
            _, _, x_act, Cup, Cs = ctx.saved
            wt = sd["up_sample.weight"]
            dleft = ops.Act(dX.buf, 0, Cup)
            dWt, dbt = torch.empty_like(wt, dtype=torch.float32), eng._f32(Cup)
            wsb = eng.scratch.get("wg", eng.lib.cmu_convT2x2_wgrad_ws_bytes(x_act.B, x_act.H, x_act.W, x_act.C, Cup, eng.dt))
            ops.convT2x2_wgrad(x_act, dleft, dWt, dbt, wsb)
            grads["up_sample.weight"], grads["up_sample.bias"] = dWt, dbt
            dxa = eng._new(x_act.B, x_act.H, x_act.W, x_act.C)
            ops.convT2x2_dgrad(dleft, eng._wpT("up_sample.", wt, 1), dxa)
            gx = ops.apply_to_nchw(dxa)
            gskip = ops.apply_to_nchw(ops.Act(dX.buf, Cup, Cs))
        elif dX is not None:
            gx = ops.apply_to_nchw(dX)
        ctx.saved = None
        out = [grads.get(n) for n in ctx.names]
        return (None, None, None, gx, gskip, *out)


class UNet(_EngineOwner, nn.Module):
    """U-Net: 4 down blocks, 1024-channel bottleneck, 4 up blocks, 1x1 head (model.py:84-131).

    forward(x: (B,H,W)) -> logits (B,out_classes,H,W) fp32.  H and W must be multiples of 2**(depth-1).
    """

    def __init__(self, out_classes=2, up_sample_mode='conv_transpose', base_ch=64, depth=5, dtype="f32"):
        super().__init__()
        self.up_sample_mode = up_sample_mode
        self.dtype = dtype
        chans = [base_ch * 2 ** i for i in range(depth)]
        cin = 1
        for i in range(depth - 1):                       # Downsampling Path (model.py:96-99)
            setattr(self, f"down_conv{i + 1}", DownBlock(cin, chans[i], dtype))
            cin = chans[i]
        self.double_conv = DoubleConv(cin, chans[-1], dtype)   # Bottleneck (model.py:101)
        for i in range(depth - 1, 0, -1):                # Upsampling Path (model.py:103-106)
            setattr(self, f"up_conv{i}", UpBlock(chans[i], chans[i - 1], up_sample_mode, dtype))
        self.conv_last = nn.Conv2d(chans[0], out_classes, kernel_size=1)   # Final Convolution (model.py:108)

    def forward(self, x, mask=None, mask_per_sample=False):
        """``mask`` (uint8, 1 = masked) is the build's fused form of x*(1-mask) (UNet_encoder.py:156)."""
        _require_cuda(x, "UNet")
        if x.dim() != 3:
            raise ValueError(f"UNet expects (B,H,W) input like the reference (model.py:120), got {tuple(x.shape)}")
        names, params = _param_args(self)
        return _UNetFn.apply(self, x, mask, mask_per_sample, names, *params)
