"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

TEST INFRASTRUCTURE ONLY.  Imports /root/reference/Finetuning/model.py (unmodified) and
Finetuning/metrics.py (behind a 3-file scikit-image stub created in a temp dir: SURVEY Appendix C-2),
feeds them seeded inputs/weights and stores inputs + expected outputs/gradients as small fixtures.
The reference never travels: only the vectors written here do.  Every fixture is also checked
against the oracle restatement (oracle/unet.py, oracle/losses.py) before it is written, so a
fixture that exists implies "oracle == reference" on that case.

Usage (from the repo root, CPU):  python -m oracle.gen_golden
"""
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

sys.dont_write_bytecode = True


def import_reference():
    sys.path.insert(0, os.path.join(REF, "Finetuning"))
    import model as ref_model  # noqa
    stub = tempfile.mkdtemp(prefix="skimage_stub_")
    os.makedirs(os.path.join(stub, "skimage"))
    with open(os.path.join(stub, "skimage", "__init__.py"), "w") as f:
        f.write("from . import morphology, measure\n")
    with open(os.path.join(stub, "skimage", "morphology.py"), "w") as f:
        f.write("def skeletonize(*a, **k):\n    raise NotImplementedError\n"
                "def skeletonize_3d(*a, **k):\n    raise NotImplementedError\n")
    with open(os.path.join(stub, "skimage", "measure.py"), "w") as f:
        f.write("def find_contours(*a, **k):\n    raise NotImplementedError\n")
    sys.path.insert(0, stub)
    import metrics as ref_metrics  # noqa
    return ref_model, ref_metrics


def t2n(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **t2n(arrays))
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def close(a, b, tol=2e-5, what=""):
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * max(1.0, ref), f"oracle != reference on {what}: {err} (ref max {ref})"


def grads_of(mod):
    return {k: p.grad.detach().clone() for k, p in mod.named_parameters()}


def import_spark():
    """Pretraining/Spark with minimal stand-ins for timm / tensorboard and the stale `UNET` package name."""
    import importlib.util
    import types
    stub = tempfile.mkdtemp(prefix="spark_stub_")
    os.makedirs(os.path.join(stub, "timm", "models"))
    os.makedirs(os.path.join(stub, "timm", "loss"))
    os.makedirs(os.path.join(stub, "UNET"))
    with open(os.path.join(stub, "timm", "__init__.py"), "w") as f:
        f.write("_REG = {}\ndef create_model(name, **kw):\n    kw.pop('pretrained', None)\n    return _REG[name](**kw)\n")
    with open(os.path.join(stub, "timm", "models", "__init__.py"), "w") as f:
        f.write("from .. import create_model\n")
    with open(os.path.join(stub, "timm", "models", "layers.py"), "w") as f:
        f.write("import torch.nn as nn\nfrom types import SimpleNamespace\nfrom torch.nn.init import trunc_normal_\n"
                "class DropPath(nn.Identity):\n    def __init__(self, *a, **k):\n        super().__init__()\n"
                "drop = SimpleNamespace(DropPath=DropPath)\n")
    with open(os.path.join(stub, "timm", "models", "registry.py"), "w") as f:
        f.write("import timm\ndef register_model(fn):\n    timm._REG[fn.__name__] = fn\n    return fn\n")
    with open(os.path.join(stub, "timm", "loss", "__init__.py"), "w") as f:
        f.write("import torch.nn as nn\nclass SoftTargetCrossEntropy(nn.Module):\n    pass\n")
    with open(os.path.join(stub, "UNET", "__init__.py"), "w") as f:
        f.write("")
    with open(os.path.join(stub, "UNET", "model.py"), "w") as f:
        f.write("import importlib.util\n_s = importlib.util.spec_from_file_location('ref_unet_model', '%s/Finetuning/model.py')\n"
                "_m = importlib.util.module_from_spec(_s)\n_s.loader.exec_module(_m)\n"
                "DoubleConv, DownBlock, UpBlock, UNet = _m.DoubleConv, _m.DownBlock, _m.UpBlock, _m.UNet\n" % REF)
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    # Finetuning/utils.py (a plain module) would shadow Spark's namespace package `utils/`: drop it for this import
    sys.path[:] = [q for q in sys.path if not q.rstrip("/").endswith("Finetuning")]
    sys.modules.pop("utils", None)
    sys.path.insert(0, os.path.join(REF, "Pretraining", "Spark"))
    sys.path.insert(0, stub)
    import encoder as sp_encoder  # noqa
    import decoder as sp_decoder  # noqa
    import spark as sp_spark  # noqa
    from models import build_sparse_encoder  # noqa
    return sp_encoder, sp_decoder, sp_spark, build_sparse_encoder


_SPARK_MODS = None


def gen_spark(ref, S=64, ratio=0.6, name="spark_unet", seed=71, B=2):
    """``S`` = image side (f = S/16 patches per side; the reference model is size-agnostic), ``ratio`` = mask ratio.
    spark_unet: 64 px, ratio 0.6 (6 of 16 patches active).  spark_unet_m75: 128 px at BASELINE config 5's ratio 0.75
    (16 of 64 patches active per image; arg_util.py:30 / spark.py:82-86).  spark_unet_m75_b8: the same at batch 8 -- 128
    active positions per channel in the bottleneck's sparse BatchNorm instead of 32: a well-conditioned case, on which the 16-bit
    paths are held to a per-tensor gradient-norm bar without slack (round-2 review)."""
    global _SPARK_MODS
    from oracle import unet as OU, spark as OS
    if _SPARK_MODS is None:
        _SPARK_MODS = import_spark()
    enc_mod, dec_mod, spark_mod, build_sparse_encoder = _SPARK_MODS
    torch.manual_seed(0)
    f = S // 16
    senc = build_sparse_encoder("unet_sparse", input_size=S, sbn=False)
    model = spark_mod.SparK(sparse_encoder=senc, dense_decoder=dec_mod.UnetDecoder(), mask_ratio=ratio, densify_norm='', sbn=False)
    assert model.fmap_h == f and model.len_keep == round(f * f * (1 - ratio)) and model.hierarchy == 5
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=seed)
    msd = model.state_dict()
    for k, v in sd.items():
        if "up_conv" in k or "conv_last" in k:
            kk = "dense_decoder." + k
            if kk == "dense_decoder.conv_last.weight":
                v = v[:1].clone()
            if kk == "dense_decoder.conv_last.bias":
                v = v[:1].clone()
        else:
            kk = "sparse_encoder.sp_cnn." + k
        assert kk in msd and tuple(msd[kk].shape) == tuple(v.shape), (kk, tuple(v.shape))
        msd[kk] = v.clone()
    g = torch.Generator().manual_seed(seed + 1)
    tokens = [0.3 * torch.randn_like(p, generator=None) for p in model.mask_tokens]
    for i, t in enumerate(tokens):
        msd[f"mask_tokens.{i}"] = t
    model.load_state_dict(msd)
    model.train()
    x = torch.randn(B, 1, S, S, generator=g)
    active = OS.make_active(B, f, ratio, g)
    assert int(active.sum()) == B * model.len_keep
    loss = model(x, active_b1ff=active)
    loss.backward()
    # oracle restatement
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in msd.items()
           if k.startswith("sparse_encoder.sp_cnn.") or k.startswith("dense_decoder.")}
    otok = [t.clone().requires_grad_(True) for t in tokens]
    lo, rec = OS.forward(x, active, osd, otok)
    lo.backward()
    close(lo.detach(), loss.detach(), what="spark loss")
    named = dict(model.named_parameters())
    for k, v in osd.items():
        if torch.is_tensor(v) and v.requires_grad and named[k].grad is not None:
            close(v.grad, named[k].grad, tol=5e-4, what=f"spark d{k}")
    for i, t in enumerate(otok):
        close(t.grad, model.mask_tokens[i].grad, tol=5e-4, what=f"spark dtoken{i}")
    gkeys = sorted(k for k in osd if osd[k].is_floating_point() and "running" not in k)
    # the reference model once more in float64: the ground truth the f32 gradients scatter around (sparse BatchNorm over as
    # few as 6 active positions per channel is ill-conditioned, so f32-vs-f32 differences overstate the error)
    import copy
    rv = model.state_dict()["sparse_encoder.sp_cnn.double_conv.double_conv.4.running_var"].clone()
    m64 = copy.deepcopy(model).double()
    m64.zero_grad()
    m64(x.double(), active_b1ff=active).backward()
    n64 = dict(m64.named_parameters())
    spot = ("dense_decoder.conv_last.weight", "sparse_encoder.sp_cnn.down_conv1.double_conv.double_conv.0.weight",
            "sparse_encoder.sp_cnn.down_conv1.double_conv.double_conv.1.bias")
    extra = {"grad64." + k: n64[k].grad for k in spot}
    extra["token_grads64_flat"] = torch.cat([p.grad.flatten() for p in m64.mask_tokens])
    extra["grad_norms64"] = torch.stack([n64[k].grad.norm() for k in gkeys])
    save(name, **extra, x=x, active=active.to(torch.uint8), loss=loss.detach(), seed=np.array(seed), mask_ratio=np.array(ratio),
         tokens_flat=torch.cat([t.flatten() for t in tokens]),
         grad_norm_keys=np.array(gkeys), grad_norms=torch.stack([named[k].grad.norm() for k in gkeys]),
         token_grads_flat=torch.cat([p.grad.flatten() for p in model.mask_tokens]),
         **{"grad." + k: named[k].grad for k in ("dense_decoder.conv_last.weight", "sparse_encoder.sp_cnn.down_conv1.double_conv.double_conv.0.weight",
                                                "sparse_encoder.sp_cnn.down_conv1.double_conv.double_conv.1.bias")},
         bott_running_var=rv)



_CMAE = None


def import_cmae():
    """Pretraining/CM-UNet's own modules -- cmae/registry.py, models/backbones/UNet_encoder.py, models/necks/{munet_neck,
    nonlinear_neck}.py, models/heads/cmunet_head.py, models/algorithms/{base,cmunet}.py, core/hooks/momentum_update_hook.py
    -- executed unmodified behind plumbing stand-ins for the two libraries this image lacks (same policy as the timm stand-in
    of import_spark; SURVEY section 8c lists mmengine 0.10.5 / mmcv 2.2.0 as absent):
      mmengine.registry   Registry(name, ...) with register_module() / build(cfg) and the 20 root registry names cmae/registry.py
                          imports as parents;
      mmengine.model      BaseModule / BaseModel = nn.Module taking init_cfg (init_weights() is a no-op: the only init_cfg on
                          this path sets BatchNorm weight 1 / bias 0, torch's default; weights are loaded explicitly anyway);
      mmengine.dist       all_gather(t) -> [t]  (one rank);      mmengine.hooks.Hook = object; is_model_wrapper(m): true for the
                          SimpleNamespace(module=...) the generator wraps the model in (the hook addresses runner.model.module)
      mmcv.cnn            build_norm_layer(cfg, n): 'SyncBN' / 'BN1d' -> nn.BatchNorm1d(n, eps, affine) -- SyncBatchNorm over one
                          rank IS BatchNorm1d on the (B, C) input of the necks, and torch's SyncBatchNorm refuses CPU tensors;
    the package __init__ files (version checks against the absent libraries, dataset / runner imports) are not executed: the
    package objects are empty modules whose __path__ points at the reference directories.  The reference hard-codes the device
    (.cuda(), .to("cuda:0"): UNet_encoder.py:84,156, cmunet.py:128, cmunet_head.py:85); in THIS process Tensor.cuda /
    Module.cuda / a 'cuda' target of Tensor.to are redirected to the CPU.  torch.distributed runs as one gloo rank
    (cmunet_head.py:84 asks for the rank).  The arithmetic that runs is the reference's."""
    global _CMAE
    if _CMAE is not None:
        return _CMAE
    import importlib
    import socket
    import types
    import torch.distributed as dist
    import torch.nn as nn
    root = os.path.join(REF, "Pretraining", "CM-UNet")

    class Registry:
        def __init__(self, name, *a, **k):
            self.name, self._m = name, {}

        def register_module(self, name=None, force=False, module=None):
            def deco(cls):
                self._m[name or cls.__name__] = cls
                return cls
            return deco(module) if module is not None else deco

        def build(self, cfg):
            cfg = dict(cfg)
            return self._m[cfg.pop("type")](**cfg)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    roots = ("DATA_SAMPLERS DATASETS EVALUATOR HOOKS LOG_PROCESSORS LOOPS METRICS MODEL_WRAPPERS MODELS OPTIM_WRAPPER_CONSTRUCTORS "
             "OPTIM_WRAPPERS OPTIMIZERS PARAM_SCHEDULERS RUNNER_CONSTRUCTORS RUNNERS TASK_UTILS TRANSFORMS VISBACKENDS VISUALIZERS "
             "WEIGHT_INITIALIZERS").split()
    mod("mmengine", __version__="0.10.5")
    mod("mmengine.registry", Registry=Registry, **{n: Registry(n) for n in roots})

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

        def init_weights(self):
            pass

    class BaseModel(BaseModule):
        def __init__(self, data_preprocessor=None, init_cfg=None):
            super().__init__(init_cfg)

    mod("mmengine.model", BaseModule=BaseModule, BaseModel=BaseModel, is_model_wrapper=lambda m: isinstance(m, types.SimpleNamespace))
    mod("mmengine.dist", all_gather=lambda t: [t])
    mod("mmengine.hooks", Hook=object)

    def build_norm_layer(cfg, num_features):
        cfg = dict(cfg)
        kind = cfg.pop("type")
        assert kind in ("SyncBN", "BN1d", "BN"), kind
        return kind.lower(), nn.BatchNorm1d(num_features, **cfg)

    mod("mmcv", __version__="2.2.0")
    mod("mmcv.cnn", build_norm_layer=build_norm_layer)
    for pkg in ("cmae", "cmae.models", "cmae.models.backbones", "cmae.models.necks", "cmae.models.heads", "cmae.models.algorithms",
                "cmae.core", "cmae.core.hooks"):
        mod(pkg).__path__ = [os.path.join(root, *pkg.split("."))]
    # device redirects (this process only)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    _to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) or (isinstance(x, torch.device) and x.type == "cuda") else x for x in a)
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return _to(self, *a, **k)

    torch.Tensor.to = to
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    reg = importlib.import_module("cmae.registry")
    for name in ("cmae.models.backbones.UNet_encoder", "cmae.models.necks.munet_neck", "cmae.models.necks.nonlinear_neck",
                 "cmae.models.heads.cmunet_head", "cmae.models.algorithms.base", "cmae.models.algorithms.cmunet"):
        importlib.import_module(name)
    try:
        hook = importlib.import_module("cmae.core.hooks.momentum_update_hook")
    except Exception as e:      # the hook pulls more of mmengine than the stand-in has: its formula is then pinned by value only
        print("  (momentum_update_hook not importable behind the stand-in:", repr(e), ")")
        hook = None
    cfg = {}
    exec(compile(open(os.path.join(root, "configs", "cmunet_config.py")).read(), "cmunet_config.py", "exec"), cfg)
    _CMAE = (reg.MODELS, cfg["model"], hook)
    return _CMAE


def gen_cmunet(seed=4100):
    """tests/golden/cmunet_ref.npz: the reference's own CM_UNet (cmunet_config.py, 224 x 224, bs 4) run here -- the patch mask of
    UNet_encoder.create_random_patch_mask under a numpy seed, forward_train's two losses, the gradient norm of every trainable
    parameter, a few gradients in full, the EMA of momentum_update -- and CMUNetPretrainHead.forward on its own (per-row target
    normalisation, masked MSE, in-batch InfoNCE through the predictor neck).  The oracle restatement (oracle/cmunet.py) is
    asserted equal on every number before the file is written."""
    from oracle import cmunet as OC
    MODELS, model_cfg, hook = import_cmae()
    torch.manual_seed(0)
    model = MODELS.build(model_cfg)
    sd = OC.make_cmunet_sd(seed)
    msd = model.state_dict()
    assert set(msd) == set(sd), (sorted(set(msd) ^ set(sd))[:8])
    assert all(tuple(msd[k].shape) == tuple(sd[k].shape) for k in sd)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    B, S = 4, 224      # (bs 2 would do for the convolutions, but a two-sample BatchNorm1d in the necks passes gradients of order eps)
    g = torch.Generator().manual_seed(seed + 10)
    img, img_t = torch.randn(B, S, S, generator=g), torch.randn(B, S, S, generator=g)
    # the reference draws its patch mask from numpy's global RNG and its reduce_channels conv from torch's (cmunet.py:128)
    np.random.seed(seed + 11)
    torch.manual_seed(seed + 12)
    losses = model.forward_train(img, img_t=img_t)
    (losses["loss_ct"] + losses["loss_rc"]).backward()
    # the same draws again, on their own
    np.random.seed(seed + 11)
    mask = model.backbone.create_random_patch_mask(B, S)
    torch.manual_seed(seed + 12)
    rc = torch.nn.Conv2d(1024, 256, kernel_size=1)
    rw, rb = rc.weight.detach().clone(), rc.bias.detach().clone()
    assert mask.shape == (B, S, S) and int(mask[0].sum()) == (int(0.65 * S * S) // 256) * 256
    fi = OC.cmunet_fixture_inputs(seed, B, S)     # what the tests regenerate
    assert torch.equal(fi[0], img) and torch.equal(fi[1], img_t) and np.array_equal(fi[2], mask) and torch.equal(fi[3], rw) and torch.equal(fi[4], rb)
    named = dict(model.named_parameters())
    trainable = sorted(k for k, p in named.items() if p.requires_grad)
    assert all(named[k].grad is not None for k in trainable) and not any(k.startswith("target_") for k in trainable)
    # ---- oracle == reference ------------------------------------------------------------------------------------------
    osd = {k: (v.clone().requires_grad_(True) if k in trainable else v.clone()) for k, v in sd.items()}
    np.random.seed(seed + 11)
    omask = OC.create_random_patch_mask(B, S, 16, 0.65)
    assert np.array_equal(omask, mask), "oracle mask != reference mask under the same numpy seed"
    ol = OC.forward_train(img, img_t, mask, rw, rb, osd, temperature=0.07, ct_weight=1.0, rc_weight=1.0)
    (ol["loss_ct"] + ol["loss_rc"]).backward()
    close(ol["loss_rc"].detach(), losses["loss_rc"].detach(), what="cmunet loss_rc")
    close(ol["loss_ct"].detach(), losses["loss_ct"].detach(), tol=1e-4, what="cmunet loss_ct")
    for k in trainable:
        gr, go = named[k].grad, osd[k].grad
        if gr.abs().max() < 1e-7:
            assert go.abs().max() < 1e-5, k
            continue
        e = ((go - gr).norm() / gr.norm()).item()
        assert e <= 2e-3, f"oracle != reference on d{k}: {e:.2e}"
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            close(osd[k].double(), v.double(), tol=1e-4, what=k)
    after_bn = {k: v.clone() for k, v in model.state_dict().items() if k.endswith(("bn0.running_var", "double_conv.double_conv.4.running_mean"))}
    # ---- EMA (cmunet.py:78-92) ---------------------------------------------------------------------------------------------
    model.momentum = 0.9
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.momentum_update()
    OC.momentum_update(before, 0.9)
    tkeys = sorted(k for k in named if k.startswith("target_"))
    for k in tkeys:
        close(before[k], named[k].detach(), tol=1e-6, what="ema " + k)
    ema_norms = torch.stack([named[k].detach().double().norm() for k in tkeys])
    ema_sample = named["target_backbone.down_conv1.double_conv.double_conv.0.weight"].detach().clone()
    grad_norms = torch.stack([named[k].grad.double().norm() for k in trainable])
    spot = ("backbone.down_conv1.double_conv.double_conv.0.weight", "backbone.double_conv.double_conv.4.weight",
            "pixel_decoder.conv_last.weight", "feature_decoder.conv_last.weight", "projector.bn0.weight", "head.predictor.bn0.bias",
            "pixel_decoder.up_conv1.up_sample.bias")
    spot_grads = {"grad." + k: named[k].grad.detach().clone() for k in spot}
    # ---- MomentumUpdateHook (momentum_update_hook.py:29-47) driven by a stand-in runner ------------------------------------------
    hook_cases, hook_m = [], []
    if hook is not None:
        import types
        for it, mx, base, end in ((0, 100, 0.99, 1.0), (37, 100, 0.99, 1.0), (100, 100, 0.99, 1.0), (5, 300, 0.996, 0.996), (250, 300, 0.9, 0.999)):
            model.base_momentum = base
            runner = types.SimpleNamespace(model=types.SimpleNamespace(module=model), iter=it, max_iters=mx)
            hook.MomentumUpdateHook(end_momentum=end).before_train_iter(runner, 0)
            assert abs(OC.momentum_schedule(it, mx, base, end) - model.momentum) < 1e-12
            hook_cases.append((it, mx, base, end))
            hook_m.append(model.momentum)
        # after_train_iter -> momentum_update() through the wrapper branch
        model.momentum = 0.5
        b2 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        hook.MomentumUpdateHook().after_train_iter(types.SimpleNamespace(model=types.SimpleNamespace(module=model)), 0)
        OC.momentum_update(b2, 0.5)
        close(b2["target_projector.fc1.weight"], named["target_projector.fc1.weight"].detach(), tol=1e-6, what="hook ema")
        model.base_momentum = 0.996
    # ---- the head on its own (cmunet_head.py:47-91), small image ----------------------------------------------------------------
    Hh, Wh, Bh = 32, 48, 4
    x = torch.randn(Bh, Hh, Wh, generator=g) * 2 + 0.5
    pred = torch.randn(Bh, Hh, Wh, generator=g).requires_grad_(True)
    mk = (torch.rand(Bh, Hh, Wh, generator=g) > 0.4).to(torch.uint8)
    ps = torch.randn(Bh, 1, 256, generator=g).requires_grad_(True)
    pt = torch.randn(Bh, 1, 256, generator=g)
    model.zero_grad()
    hl = model.head(x, pred, mk, ps, pt)
    (hl["loss_ct"] + hl["loss_rc"]).backward()
    hsd = {k: (v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in model.state_dict().items()
           if k.startswith("head.")}
    # (the predictor's BatchNorm buffers advanced in the call above: the oracle starts from the state before it)
    for k in list(hsd):
        if "running_" in k or "num_batches" in k:
            hsd[k] = osd[k].detach().clone()
    po, pso = pred.detach().clone().requires_grad_(True), ps.detach().clone().requires_grad_(True)
    oh = OC.head(x, po, mk, pso, pt, hsd, "head.", 0.07, 1.0, 1.0)
    (oh["loss_ct"] + oh["loss_rc"]).backward()
    close(oh["loss_rc"].detach(), hl["loss_rc"].detach(), what="head loss_rc")
    close(oh["loss_ct"].detach(), hl["loss_ct"].detach(), tol=1e-4, what="head loss_ct")
    close(po.grad, pred.grad, tol=1e-4, what="head dpred")
    close(pso.grad, ps.grad, tol=2e-4, what="head dproj_s")
    out = {"seed": np.array(seed), "B": np.array(B), "S": np.array(S), "mask_patches": mask[:, ::16, ::16].copy(),
           "loss_ct": losses["loss_ct"].detach(), "loss_rc": losses["loss_rc"].detach(),
           "trainable": np.array(trainable), "grad_norms": grad_norms,
           "hook_cases": np.array(hook_cases, dtype=np.float64), "hook_momentum": np.array(hook_m, dtype=np.float64),
           "target_keys": np.array(tkeys), "ema_norms": ema_norms, "ema_sample": ema_sample,
           "head.x": x, "head.pred": pred.detach(), "head.mask": mk, "head.proj_s": ps.detach(), "head.proj_t": pt,
           "head.loss_ct": hl["loss_ct"].detach(), "head.loss_rc": hl["loss_rc"].detach(), "head.dpred": pred.grad, "head.dproj_s": ps.grad}
    out.update(spot_grads)
    out.update({"after." + k: v for k, v in after_bn.items()})
    # ---- the patch mask generator over more geometries (UNet_encoder.py:106-139; the encoder object only carries patch_size / ratio) ----
    mcases = [(3, 224, 0.65), (2, 256, 0.6), (2, 512, 0.75), (2, 512, 0.6), (1, 64, 0.0), (2, 64, 1.0), (3, 96, 0.33), (2, 128, 0.999)]
    enc = model.backbone
    for ci, (mb, ms_, mr) in enumerate(mcases):
        enc.mask_ratio = mr
        np.random.seed(seed + 300 + ci)
        mref = enc.create_random_patch_mask(mb, ms_)
        np.random.seed(seed + 300 + ci)
        assert np.array_equal(OC.create_random_patch_mask(mb, ms_, 16, mr), mref), (mb, ms_, mr)
        assert np.array_equal(np.repeat(np.repeat(mref[:, ::16, ::16], 16, 1), 16, 2), mref)
        out[f"mask{ci}"] = mref[:, ::16, ::16].copy()
    enc.mask_ratio = 0.65
    out["mask_cases"] = np.array(mcases, dtype=np.float64)
    # ---- the head with other hyper-parameters (cmunet_head.py:39-44: temperature, ct_weight, rc_weight) ------------------------------
    import copy
    head2 = copy.deepcopy(model.head)
    head2.load_state_dict({k[len("head."):]: v.clone() for k, v in sd.items() if k.startswith("head.")})
    head2.t, head2.ct_weight, head2.rc_weight = 0.2, 0.5, 2.0
    head2.train()
    pred2, ps2 = out["head.pred"].clone().requires_grad_(True), out["head.proj_s"].clone().requires_grad_(True)
    hl2 = head2(out["head.x"], pred2, out["head.mask"], ps2, out["head.proj_t"])
    (hl2["loss_ct"] + hl2["loss_rc"]).backward()
    hsd2 = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in sd.items() if k.startswith("head.")}
    po2, pso2 = out["head.pred"].clone().requires_grad_(True), out["head.proj_s"].clone().requires_grad_(True)
    oh2 = OC.head(out["head.x"], po2, out["head.mask"], pso2, out["head.proj_t"], hsd2, "head.", 0.2, 0.5, 2.0)
    (oh2["loss_ct"] + oh2["loss_rc"]).backward()
    close(oh2["loss_rc"].detach(), hl2["loss_rc"].detach(), what="head2 loss_rc")
    close(oh2["loss_ct"].detach(), hl2["loss_ct"].detach(), tol=1e-4, what="head2 loss_ct")
    close(po2.grad, pred2.grad, tol=1e-4, what="head2 dpred")
    close(pso2.grad, ps2.grad, tol=2e-4, what="head2 dproj_s")
    out.update({"head2.hyper": np.array([0.2, 0.5, 2.0]), "head2.loss_ct": hl2["loss_ct"].detach(), "head2.loss_rc": hl2["loss_rc"].detach(),
                "head2.dpred": pred2.grad, "head2.dproj_s": ps2.grad})
    save("cmunet_ref", **out)


_MOCO = None


def import_moco():
    """Pretraining/MoCo's own moco2_module.py / moco_data_module.py (Moco_v2, UNet_encoder, concat_all_gather) executed
    unmodified behind plumbing stand-ins for the libraries this image lacks (pytorch-lightning 1.6,
    lightning-bolts 0.5 -- whose vendored copy is incomplete --, torchvision, wandb):
      pytorch_lightning   LightningModule = nn.Module + save_hyperparameters() (the constructor's arguments as self.hparams),
                          log / log_dict no-ops, a plain `trainer` attribute; Trainer, LightningDataModule, ModelCheckpoint,
                          WandbLogger, DDPPlugin, DDP2Plugin as empty classes; rank_zero_only = identity;
      torchvision         empty transforms / transforms.functional / datasets modules (data pipeline only; ImageFolder = object);
      pl_bolts            utils flags (_TORCHVISION_AVAILABLE False, _PIL_AVAILABLE True), warn_missing_pkg no-op, the three
                          normalisation names transforms.py imports; pl_bolts/metrics/aggregation.py is the reference's own file.
    Only the model code runs: Moco_v2.__init__ / training_step / forward / _momentum_update_key_encoder /
    _dequeue_and_enqueue / _compute_l_s with a stand-in trainer whose strategy is not DDP (one process)."""
    global _MOCO
    if _MOCO is not None:
        return _MOCO
    import importlib
    import inspect
    import types
    import torch.nn as nn
    root = os.path.join(REF, "Pretraining", "MoCo")
    mdir = os.path.join(root, "pl_bolts", "models", "self_supervised", "moco")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, logger=True, ignore=()):
            fr = inspect.currentframe().f_back
            av = inspect.getargvalues(fr)
            self.hparams = types.SimpleNamespace(**{k: av.locals[k] for k in av.args if k != "self" and k not in ignore})

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    empty = lambda n: type(n, (), {})       # noqa: E731
    mod("pytorch_lightning", LightningModule=LightningModule, Trainer=empty("Trainer"), LightningDataModule=empty("LightningDataModule"))
    mod("pytorch_lightning.callbacks", ModelCheckpoint=empty("ModelCheckpoint"))
    mod("pytorch_lightning.loggers", WandbLogger=empty("WandbLogger"))
    mod("pytorch_lightning.plugins", DDPPlugin=empty("DDPPlugin"), DDP2Plugin=empty("DDP2Plugin"))
    mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f, _module_available=lambda n: False)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional")
    tv.datasets = mod("torchvision.datasets", ImageFolder=object, DatasetFolder=object)
    mod("pl_bolts").__path__ = [os.path.join(root, "pl_bolts")]
    mod("pl_bolts.utils", _TORCHVISION_AVAILABLE=False, _PIL_AVAILABLE=True)
    mod("pl_bolts.utils.warnings", warn_missing_pkg=lambda *a, **k: None)
    mod("pl_bolts.transforms")
    mod("pl_bolts.transforms.dataset_normalizations", imagenet_normalization=None, cifar10_normalization=None, stl10_normalization=None)
    mod("pl_bolts.metrics").__path__ = [os.path.join(root, "pl_bolts", "metrics")]
    agg = importlib.import_module("pl_bolts.metrics.aggregation")
    for n in ("accuracy", "mean", "precision_at_k"):
        setattr(sys.modules["pl_bolts.metrics"], n, getattr(agg, n))
    for clash in ("transforms", "utils", "models", "encoder", "decoder"):      # plain module names other reference trees also use
        sys.modules.pop(clash, None)
    sys.path[:] = [q for q in sys.path if "/Pretraining/Spark" not in q and not q.rstrip("/").endswith("Finetuning")]
    sys.path.insert(0, mdir)
    m2 = importlib.import_module("moco2_module")
    _MOCO = m2
    return _MOCO


def gen_moco(seed=5200):
    """tests/golden/moco_ref.npz: the reference's own Moco_v2 (UNet_encoder of moco_data_module.py:47-66; emb 1024, K = 64 negatives,
    tau 0.2, m 0.99) through one training_step at bs 4, 64 x 64: loss, logits, the enqueued keys and pointer, the gradient norm of
    every query-encoder parameter, sampled gradients, the key encoder after its EMA; a second step (pointer 4 -> 8, the queue holding
    the first step's keys); forward()'s (logits, labels, k, q) contract.  The oracle (oracle/moco.py) is asserted equal first."""
    import types
    from oracle import moco as OM, unet as OU
    m2 = import_moco()
    B, S, K, T, EM = 4, 64, 64, 0.2, 0.99
    torch.manual_seed(0)
    model = m2.Moco_v2(emb_dim=1024, num_negatives=K, encoder_momentum=EM, softmax_temperature=T)
    model.trainer = types.SimpleNamespace(datamodule=types.SimpleNamespace(name="synthetic"), strategy=None)
    sd = OM.make_moco_sd(seed, K)
    msd = model.state_dict()
    assert set(sd) <= set(msd) and all(tuple(msd[k].shape) == tuple(v.shape) for k, v in sd.items()), sorted(set(sd) - set(msd))[:5]
    assert set(msd) - set(sd) == {"val_queue", "val_queue_ptr"}
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=False)
    model.train()
    xq, xk, xq2, xk2 = OM.moco_fixture_inputs(seed, B, S)
    named = dict(model.named_parameters())
    qkeys = sorted(k for k, p in named.items() if p.requires_grad)
    assert all(k.startswith("encoder_q.") for k in qkeys)
    loss = model.training_step(((xq, xk), 0), 0)
    loss.backward()
    # oracle == reference
    osd = {k: (v.clone().requires_grad_(True) if k in qkeys else v.clone()) for k, v in sd.items()}
    queue, ptr = sd["queue"].clone(), sd["queue_ptr"].clone()
    ol, ologits, ok = OM.training_step(xq, xk, osd, queue, ptr, T, EM)
    ol.backward()
    close(ol.detach(), loss.detach(), what="moco loss")
    close(queue, model.queue, tol=1e-5, what="moco queue")
    assert int(ptr) == int(model.queue_ptr) == B
    for k in qkeys:
        gr, go = named[k].grad, osd[k].grad
        if k.endswith((".0.bias", ".3.bias")):
            continue
        e = ((go - gr).norm() / gr.norm()).item()
        assert e <= 2e-3, f"oracle != reference on d{k}: {e:.2e}"
    kkeys = sorted(k for k in named if k.startswith("encoder_k."))
    for k in kkeys:
        close(osd[k], named[k].detach(), tol=1e-6, what="ema " + k)
    gn = torch.stack([named[k].grad.double().norm() for k in qkeys])
    spot = ("encoder_q.down_conv1.double_conv.double_conv.0.weight", "encoder_q.double_conv.double_conv.3.weight",
            "encoder_q.down_conv2.double_conv.double_conv.4.bias")
    out = {"seed": np.array(seed), "B": np.array(B), "S": np.array(S), "K": np.array(K), "T": np.array(T), "EM": np.array(EM),
           "loss": loss.detach(), "keys": model.queue[:, :B].t().clone(), "queue_ptr": model.queue_ptr.clone(),
           "qkeys": np.array(qkeys), "grad_norms": gn, "kkeys": np.array(kkeys),
           "ema_norms": torch.stack([named[k].detach().double().norm() for k in kkeys]),
           "ema_sample": named["encoder_k.double_conv.double_conv.0.weight"].detach()[:8].clone()}
    out.update({"grad." + k: named[k].grad.detach().clone() for k in spot[:1]})
    out.update({"gradsum." + k: named[k].grad.detach().double().sum() for k in spot})
    # forward() contract on the updated state (moco2_module.py:224-270), before the second step changes it
    with torch.no_grad():
        lg, lb, kk, qq = model(xq2, xk2, model.queue)
    assert lg.shape == (B, 1 + K) and lb.shape == (B,) and kk.shape == (B, 1024) and qq.shape == (B, 1024)
    # second step: the oracle continues from its own state
    model.zero_grad()
    loss2 = model.training_step(((xq2, xk2), 0), 1)
    for k in qkeys:
        osd[k].grad = None
    with torch.no_grad():   # (forward() above advanced the BatchNorm buffers of both encoders once more: so does the oracle)
        OM.encoder_gap(xq2, osd, "encoder_q.", True)
        OM.encoder_gap(xk2, osd, "encoder_k.", True)
    ol2, _, _ = OM.training_step(xq2, xk2, osd, queue, ptr, T, EM)
    close(ol2.detach(), loss2.detach(), what="moco loss step 2")
    close(queue, model.queue, tol=1e-5, what="moco queue step 2")
    assert int(ptr) == int(model.queue_ptr) == 2 * B
    out.update({"fwd.logits": lg, "fwd.labels": lb, "loss2": loss2.detach(), "keys2": model.queue[:, B:2 * B].t().clone(),
                "queue_ptr2": model.queue_ptr.clone()})
    save("moco_ref", **out)


def gen_moco_val(seed=5400):
    """tests/golden/moco_val_ref.npz: the reference's own Moco_v2.validation_step (moco2_module.py:311-329) with the module in eval
    mode (what Lightning's validation loop sets), bs 8, 64 x 64, K = 64: val_loss, val_acc1, val_acc5, the keys enqueued into
    val_queue and its pointer; a second batch against the updated val_queue.  The oracle is asserted equal first."""
    import types
    from oracle import moco as OM
    m2 = import_moco()
    B, S, K, T = 8, 64, 64, 0.2
    torch.manual_seed(0)
    model = m2.Moco_v2(emb_dim=1024, num_negatives=K, encoder_momentum=0.99, softmax_temperature=T)
    model.trainer = types.SimpleNamespace(datamodule=types.SimpleNamespace(name="synthetic"), strategy=None)
    sd = OM.make_moco_sd(seed, K)
    sd["val_queue"], sd["val_queue_ptr"] = OM.init_queue(1024, K, seed + 1), torch.zeros(1, dtype=torch.long)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.eval()
    a = OM.moco_fixture_inputs(seed, B, S)
    b = OM.moco_fixture_inputs(seed + 1, B, S)
    vq, vp = sd["val_queue"].clone(), sd["val_queue_ptr"].clone()
    out = {"seed": np.array(seed), "B": np.array(B), "S": np.array(S), "K": np.array(K), "T": np.array(T)}
    for i, (x1, x2) in enumerate(((a[0], a[1]), (b[0], b[1]))):
        with torch.no_grad():
            r = model.validation_step(((x1, x2), torch.zeros(B)), i)
        ol, o1, o5 = OM.validation_step(x1, x2, sd, vq, vp, T)
        close(ol, r["val_loss"], what=f"moco val loss {i}")
        assert torch.equal(o1, r["val_acc1"]) and torch.equal(o5, r["val_acc5"]), (o1, r["val_acc1"], o5, r["val_acc5"])
        close(vq, model.val_queue, tol=1e-5, what="moco val queue")
        assert int(vp) == int(model.val_queue_ptr) == (i + 1) * B
        out.update({f"val_loss{i}": r["val_loss"].detach(), f"val_acc1_{i}": r["val_acc1"], f"val_acc5_{i}": r["val_acc5"],
                    f"keys{i}": model.val_queue[:, i * B:(i + 1) * B].t().clone()})
    assert torch.equal(model.queue, sd["queue"]) and int(model.queue_ptr) == 0          # the training queue is untouched
    out["val_queue_ptr"] = model.val_queue_ptr.clone()
    save("moco_val_ref", **out)


def _moco_2rank_worker(rank, port, outdir, seed):
    """One of two gloo ranks running the reference's own Moco_v2 with a DDP strategy: shuffle-BN, gathered keys, enqueue."""
    import types
    import torch.distributed as dist
    from oracle import moco as OM
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    m2 = import_moco()
    torch.Tensor.cuda = lambda self, *a, **k: self       # (_batch_shuffle_ddp: torch.randperm(n).cuda(), moco2_module.py:191)
    B, S, K, T, EM = 4, 64, 64, 0.2, 0.99
    torch.manual_seed(0)
    model = m2.Moco_v2(emb_dim=1024, num_negatives=K, encoder_momentum=EM, softmax_temperature=T)
    ddp = sys.modules["pytorch_lightning.plugins"].DDPPlugin()
    model.trainer = types.SimpleNamespace(datamodule=types.SimpleNamespace(name="synthetic"), strategy=ddp)
    model.load_state_dict({k: v.clone() for k, v in OM.make_moco_sd(seed, K).items()}, strict=False)
    model.train()
    xq, xk, _, _ = OM.moco_fixture_inputs(seed + 50 * (rank + 1), B, S)        # every rank its own images
    torch.manual_seed(seed + 7)          # rank 0's permutation (torch.randperm on the CPU generator) is the one that is broadcast
    loss = model.training_step(((xq, xk), 0), 0)
    loss.backward()
    named = dict(model.named_parameters())
    qkeys = sorted(k for k, p in named.items() if p.requires_grad)
    torch.save({"loss": loss.detach(), "queue": model.queue.clone(), "ptr": model.queue_ptr.clone(), "qkeys": qkeys,
                "grad_norms": torch.stack([named[k].grad.double().norm() for k in qkeys]),
                "grad0": named["encoder_q.down_conv1.double_conv.double_conv.0.weight"].grad.clone(),
                "bn_k": model.state_dict()["encoder_k.down_conv1.double_conv.double_conv.1.running_mean"].clone()},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def gen_moco_2rank(seed=5300):
    """tests/golden/moco_ref_2rank.npz: the reference's own Moco_v2 on TWO gloo ranks (strategy = DDPPlugin, so _use_ddp_or_ddp2
    is true): _batch_shuffle_ddp / _batch_unshuffle_ddp around the key encoder (rank 0's torch.randperm broadcast), concat_all_gather
    of the keys, 2B rows enqueued on both ranks.  Per rank: loss, local (not yet averaged) gradient norms of the query encoder, the
    queue and pointer after the step, a key-encoder BatchNorm buffer (its batch = the shuffled mixture of both ranks' images).  The
    oracle emulation (the same permutation redrawn from the seed, the key encoder per shuffled group) is asserted equal first."""
    import socket
    import subprocess
    import torch.nn.functional as F
    from oracle import moco as OM
    B, S, K, T, EM = 4, 64, 64, 0.2, 0.99
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    outdir = tempfile.mkdtemp(prefix="moco2rank_")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.gen_golden", "--moco-worker", str(r), str(port), outdir, str(seed)],
                              cwd=os.path.dirname(os.path.dirname(OUT))) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(2)]
    assert torch.equal(res[0]["queue"], res[1]["queue"]) and int(res[0]["ptr"]) == int(res[1]["ptr"]) == 2 * B
    # ---- oracle emulation of the two ranks in one process ------------------------------------------------------------------
    sd0 = OM.make_moco_sd(seed, K)
    ins = [OM.moco_fixture_inputs(seed + 50 * (r + 1), B, S) for r in range(2)]
    torch.manual_seed(seed + 7)
    idx = torch.randperm(2 * B)
    un = torch.argsort(idx)
    xk_all = torch.cat([ins[0][1], ins[1][1]])
    losses, gnorms, bn_k, queue_after = [], [], [], None
    ksh = []
    sds = []
    for r in range(2):
        sd = {k: v.clone() for k, v in sd0.items()}
        OM.momentum_update(sd, EM)
        with torch.no_grad():
            ksh.append(OM.encoder_gap(xk_all[idx.view(2, -1)[r]], sd, "encoder_k.", True))
        sds.append(sd)
    k_all = F.normalize(torch.cat(ksh)[un], dim=1)                                   # keys back with their owners, rank-major
    for r in range(2):
        sd = sds[r]
        qkeys = res[r]["qkeys"]
        for k in qkeys:
            sd[k] = sd[k].clone().requires_grad_(True)
        q = F.normalize(OM.encoder_gap(ins[r][0], sd, "encoder_q.", True), dim=1)
        k_own = k_all[r * B:(r + 1) * B]
        logits = torch.cat([(q * k_own).sum(1, keepdim=True), q @ sd0["queue"]], 1) / T
        loss = F.cross_entropy(logits, torch.zeros(B, dtype=torch.long))
        loss.backward()
        close(loss.detach(), res[r]["loss"], what=f"2-rank moco loss (rank {r})")
        gn = torch.stack([sd[k].grad.double().norm() for k in qkeys])
        live = torch.tensor([not k.endswith((".0.bias", ".3.bias")) for k in qkeys])
        assert ((gn - res[r]["grad_norms"]).abs() / res[r]["grad_norms"].clamp_min(1e-30))[live].max().item() <= 2e-3
        close(sd["encoder_k.down_conv1.double_conv.double_conv.1.running_mean"], res[r]["bn_k"], tol=1e-5, what="key-encoder BN buffer")
    queue = sd0["queue"].clone()
    queue[:, :2 * B] = k_all.t()
    close(queue, res[0]["queue"], tol=1e-5, what="2-rank queue")
    save("moco_ref_2rank", seed=np.array(seed), B=np.array(B), S=np.array(S), K=np.array(K), T=np.array(T), EM=np.array(EM),
         perm=idx, keys=res[0]["queue"][:, :2 * B].t().clone(), queue_ptr=res[0]["ptr"],
         loss=torch.stack([res[0]["loss"], res[1]["loss"]]), qkeys=np.array(res[0]["qkeys"]),
         grad_norms=torch.stack([res[0]["grad_norms"], res[1]["grad_norms"]]), grad0=torch.stack([res[0]["grad0"], res[1]["grad0"]]),
         bn_k=torch.stack([res[0]["bn_k"], res[1]["bn_k"]]))


def gen_finetune_sensitivity(ref, M, seed=6100, runs=4, eps=2.0 ** -22):
    """tests/golden/finetune_sensitivity.npz: how far the REFERENCE's own two-epoch trajectory of finetune_ref.npz moves when its
    initial weights are perturbed by rounding-size noise (w * (1 + eps * u), u uniform in [-1, 1], eps = 2^-22: four float32 ulps
    peak) -- the reference's loop run ``runs`` times on CPU, every log of every epoch stored.  A second implementation of the same
    arithmetic in another summation order cannot track the fixture closer than this spread: tests/test_gpu_finetune.py takes its
    trajectory bar from it."""
    import types
    from oracle import unet as OU
    for name in ("cv2", "albumentations"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if not any(q.rstrip("/").endswith("Finetuning") for q in sys.path):
        sys.path.insert(0, os.path.join(REF, "Finetuning"))
    for clash in ("utils", "config", "dataset", "train"):
        sys.modules.pop(clash, None)
    import train as ref_train  # noqa
    base = np.load(os.path.join(OUT, "finetune_ref.npz"))
    keys = [str(k) for k in base["log_keys"]]
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=seed)
    train_loader, valid_loader = OU.finetune_fixture_data(seed + 1)
    mk = dict(activation="softmax", threshold=0.5, ignore_channels=[0])
    tls, vls = [], []
    for r in range(runs):
        g = torch.Generator().manual_seed(9000 + r)
        torch.manual_seed(0)
        model = ref.UNet()
        model.load_state_dict({k: (v * (1 + eps * (2 * torch.rand(v.shape, generator=g) - 1))).to(v.dtype) if v.is_floating_point() else v.clone()
                               for k, v in sd.items()})
        loss = M.DiceLoss(**mk) + M.CrossEntropyLoss()
        metrics = [M.DiceLoss(**mk), M.CrossEntropyLoss(), M.IoU(**mk), M.soft_cldice(**mk)]
        opt = torch.optim.Adam([dict(params=model.parameters(), lr=1e-3)])
        tr = ref_train.TrainEpoch(model, loss=loss, metrics=metrics, optimizer=opt, device="cpu", verbose=False)
        va = ref_train.ValidEpoch(model, loss=loss, metrics=metrics, device="cpu", verbose=False)
        tmp = tempfile.mkdtemp(prefix="ft_sens_")
        tl, vl = ref_train.train(model, train_loader, valid_loader, tr, va, True, 2, name=os.path.join(tmp, "best_model.pth"))
        tls.append([[float(tl[ep][k]) for k in keys] for ep in range(2)])
        vls.append([[float(vl[ep][k]) for k in keys] for ep in range(2)])
    tls, vls = np.array(tls, dtype=np.float64), np.array(vls, dtype=np.float64)
    dev_t = np.abs(tls - base["train_logs"][None]).max(0)
    dev_v = np.abs(vls - base["valid_logs"][None]).max(0)
    for ep in range(2):
        for i, k in enumerate(keys):
            print(f"  sensitivity epoch {ep} {k}: train {dev_t[ep, i]:.2e}  valid {dev_v[ep, i]:.2e}")
    save("finetune_sensitivity", seed=np.array(seed), eps=np.array(eps), runs=np.array(runs), log_keys=np.array(keys),
         train_logs=tls, valid_logs=vls, train_dev=dev_t, valid_dev=dev_v)


def gen_finetune(ref, M, seed=6100):
    """tests/golden/finetune_ref.npz: the reference's OWN training loop -- Finetuning/train.py's TrainEpoch / ValidEpoch / train()
    imported behind empty cv2 / albumentations stand-ins (dataset.py imports them at the top; the Dataset class itself is not used)
    -- driving the reference UNet (31 M parameters) with the reference's DiceLoss + CrossEntropyLoss criterion (train.py:455), the
    tensor metrics of train.py:458-465 and torch.optim.Adam([dict(params=model.parameters(), lr=1e-3)]) (train.py:341) for two
    epochs over a synthetic split (6 + 2 images of 64 x 64, bs 2).  Stored: the per-epoch log dictionaries train() returns (their
    keys are the log-key contract), parameter norms after training.  The oracle loop is asserted equal first."""
    import types
    from oracle import losses as OL, unet as OU
    for name in ("cv2", "albumentations"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if not any(q.rstrip("/").endswith("Finetuning") for q in sys.path):
        sys.path.insert(0, os.path.join(REF, "Finetuning"))
    for clash in ("utils", "config", "dataset", "train"):
        sys.modules.pop(clash, None)
    import train as ref_train  # noqa
    torch.manual_seed(0)
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=seed)
    train_loader, valid_loader = OU.finetune_fixture_data(seed + 1)
    model = ref.UNet()
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    mk = dict(activation="softmax", threshold=0.5, ignore_channels=[0])
    loss = M.DiceLoss(**mk) + M.CrossEntropyLoss()
    metrics = [M.DiceLoss(**mk), M.CrossEntropyLoss(), M.IoU(**mk), M.soft_cldice(**mk)]
    opt = torch.optim.Adam([dict(params=model.parameters(), lr=1e-3)])
    tr = ref_train.TrainEpoch(model, loss=loss, metrics=metrics, optimizer=opt, device="cpu", verbose=False)
    va = ref_train.ValidEpoch(model, loss=loss, metrics=metrics, device="cpu", verbose=False)
    tmp = tempfile.mkdtemp(prefix="ft_ref_")
    tl, vl = ref_train.train(model, train_loader, valid_loader, tr, va, True, 2, name=os.path.join(tmp, "best_model.pth"))
    assert os.path.exists(os.path.join(tmp, "best_model.pth"))
    keys = sorted(tl[0])
    assert keys == sorted(vl[0]) and "dice_loss + cross_entropy_loss" in keys
    # ---- oracle loop == reference loop -------------------------------------------------------------------------------------
    osd = OU.clone_sd(sd, requires_grad=True)
    oopt = torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3)

    def run(loader, training):
        acc = {k: [] for k in ("dice_loss + cross_entropy_loss", "dice_loss", "cross_entropy_loss", "iou_loss")}
        for x, y in loader:
            if training:
                oopt.zero_grad()
                lo = OU.unet_forward(x, osd, training=True)
                l = OL.dice_ce_loss(lo, y)
                l.backward()
                oopt.step()
            else:
                with torch.no_grad():
                    lo = OU.unet_forward(x, osd, training=False)
                    l = OL.dice_ce_loss(lo, y)
            lo = lo.detach()
            acc["dice_loss + cross_entropy_loss"].append(float(l)); acc["dice_loss"].append(float(OL.dice_loss(lo, y)))
            acc["cross_entropy_loss"].append(float(OL.cross_entropy_prob(lo, y))); acc["iou_loss"].append(float(OL.iou_loss(lo, y)))
        return {k: float(np.mean(v)) for k, v in acc.items()}
    for ep in range(2):
        ot, ov = run(train_loader, True), run(valid_loader, False)
        for got, refl, what in ((ot, tl[ep], "train"), (ov, vl[ep], "valid")):
            for k in got:
                assert abs(got[k] - float(refl[k])) <= 2e-4 * max(1.0, abs(float(refl[k]))), (ep, what, k, got[k], float(refl[k]))
    named = dict(model.named_parameters())
    pk = sorted(named)
    for k in pk:
        d = (osd[k].detach() - named[k].detach()).norm().item()
        assert d <= 2e-3 * max(named[k].detach().norm().item(), 1e-6) + 1e-6, (k, d)
    save("finetune_ref", seed=np.array(seed), log_keys=np.array(keys),
         train_logs=np.array([[float(tl[ep][k]) for k in keys] for ep in range(2)], dtype=np.float64),
         valid_logs=np.array([[float(vl[ep][k]) for k in keys] for ep in range(2)], dtype=np.float64),
         param_keys=np.array(pk), param_norms=torch.stack([named[k].detach().double().norm() for k in pk]),
         conv_last_weight=named["conv_last.weight"].detach().clone())


def gen_load_model(seed=1):
    """tests/golden/load_model_ref.npz: the reference's own load_model (Finetuning/train.py:240-308) run on synthetic checkpoints in
    each of the five layouts it dispatches on -> for every key of UNet().state_dict(), whether the loaded model carries the
    checkpoint's tensor.  (train.py imported as in gen_finetune; torch.load's hard-coded map_location="cuda:1" redirected to the CPU.)"""
    import types
    from oracle import unet as OU
    for name in ("cv2", "albumentations"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if not any(q.rstrip("/").endswith("Finetuning") for q in sys.path):
        sys.path.insert(0, os.path.join(REF, "Finetuning"))
    for clash in ("utils", "config", "dataset", "train"):
        sys.modules.pop(clash, None)
    import train as ref_train  # noqa
    orig = torch.load
    torch.load = lambda f, map_location=None, **k: orig(f, map_location="cpu", weights_only=False)
    try:
        sd = OU.make_state_dict(base_ch=64, depth=5, seed=seed)
        cases = OU.checkpoint_layout_cases(sd)
        tmp = tempfile.mkdtemp(prefix="lm_ref_")
        keys = None
        table = []
        for name, ck in cases.items():
            path = os.path.join(tmp, name)
            torch.save(ck, path)
            torch.manual_seed(123)
            m = ref_train.load_model(types.SimpleNamespace(pretrained=path))
            got = m.state_dict()
            if keys is None:
                keys = list(got.keys())
            table.append([bool(torch.equal(got[k], sd[k])) if not k.endswith("num_batches_tracked") else True for k in keys])
        m0 = ref_train.load_model(types.SimpleNamespace(pretrained=None))
        assert sum(p.numel() for p in m0.parameters()) == 31042434
    finally:
        torch.load = orig
    save("load_model_ref", seed=np.array(seed), layouts=np.array(list(cases.keys())), keys=np.array(keys), loaded=np.array(table, dtype=np.uint8))
    for name, row in zip(cases, table):
        print(f"    {name}: {sum(row)} of {len(row)} keys carry the checkpoint's tensor")


def _cmunet_head_2rank_worker(rank, port, outdir, seed):
    """One of two gloo ranks running the reference's own CMUNetPretrainHead.forward (cmunet_head.py:47-91): concat_all_gather of the
    normalised target projections, labels arange(B) + B * rank.  The predictor's BatchNorm runs in eval mode (running statistics:
    SyncBatchNorm and BatchNorm1d are the same function there, so the stand-in's norm layer changes nothing) -- what is pinned is
    the head's own multi-rank arithmetic."""
    import torch.distributed as dist
    from oracle import cmunet as OC
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    MODELS, model_cfg, _ = import_cmae()
    sys.modules["mmengine.dist"].all_gather = lambda t: (lambda out: (dist.all_gather(out, t), out)[1])([torch.zeros_like(t) for _ in range(2)])
    import importlib
    importlib.import_module("cmae.models.heads.cmunet_head").all_gather = sys.modules["mmengine.dist"].all_gather
    head = MODELS.build(model_cfg["head"])
    hsd = OC.make_neck_sd("predictor.", 256, 1536, 256, seed)
    g0 = torch.Generator().manual_seed(seed + 1)
    hsd["predictor.bn0.running_mean"] = 0.1 * torch.randn(1536, generator=g0)
    hsd["predictor.bn0.running_var"] = 0.5 + torch.rand(1536, generator=g0)
    head.load_state_dict({k: v.clone() for k, v in hsd.items()}, strict=True)
    head.train()
    head.predictor.bn0.eval()
    x, pred, mk, ps, pt = OC.head_fixture_inputs(seed + 10 * (rank + 1))
    pred.requires_grad_(True); ps.requires_grad_(True)
    out = head(x, pred, mk, ps, pt)
    (out["loss_ct"] + out["loss_rc"]).backward()
    torch.save({"loss_ct": out["loss_ct"].detach(), "loss_rc": out["loss_rc"].detach(), "dpred": pred.grad.clone(), "dproj_s": ps.grad.clone(),
                "dfc1": head.predictor.fc1.weight.grad.clone()}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def gen_cmunet_head_2rank(seed=4300):
    """tests/golden/cmunet_head_2rank.npz: the reference's own CMUNetPretrainHead on TWO gloo ranks, every rank its own tensors.
    The oracle's head with the gather emulated (rank r sees the keys of both ranks, labels i + B*r) is asserted equal first."""
    import socket
    import subprocess
    import torch.nn.functional as F
    from oracle import cmunet as OC
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    outdir = tempfile.mkdtemp(prefix="head2rank_")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.gen_golden", "--head-worker", str(r), str(port), outdir, str(seed)],
                              cwd=os.path.dirname(os.path.dirname(OUT))) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(2)]
    hsd0 = OC.make_neck_sd("head.predictor.", 256, 1536, 256, seed)
    g0 = torch.Generator().manual_seed(seed + 1)
    hsd0["head.predictor.bn0.running_mean"] = 0.1 * torch.randn(1536, generator=g0)
    hsd0["head.predictor.bn0.running_var"] = 0.5 + torch.rand(1536, generator=g0)
    ins = [OC.head_fixture_inputs(seed + 10 * (r + 1)) for r in range(2)]
    keys_all = torch.cat([F.normalize(ins[r][4].squeeze(1), dim=1, p=2) for r in range(2)])
    for r in range(2):
        x, pred, mk, ps, pt = ins[r]
        hsd = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in hsd0.items()}
        po, pso = pred.clone().requires_grad_(True), ps.clone().requires_grad_(True)
        # eval-mode BatchNorm in the predictor: training=False for the neck only
        loss_rc = OC.masked_mse(po, x, mk)
        pred_s = OC.nonlinear_neck(pso, hsd, "head.predictor.", training=False)
        loss_ct = OC.infonce_inbatch(pred_s.squeeze(1), keys_all, 0.07, rank=r, ct_weight=1.0)
        (loss_ct + loss_rc).backward()
        close(loss_rc.detach(), res[r]["loss_rc"], what=f"2-rank head loss_rc (rank {r})")
        close(loss_ct.detach(), res[r]["loss_ct"], tol=1e-4, what=f"2-rank head loss_ct (rank {r})")
        close(po.grad, res[r]["dpred"], tol=1e-4, what="2-rank head dpred")
        close(pso.grad, res[r]["dproj_s"], tol=2e-4, what="2-rank head dproj_s")
        close(hsd["head.predictor.fc1.weight"].grad, res[r]["dfc1"], tol=2e-4, what="2-rank head dfc1")
    save("cmunet_head_2rank", seed=np.array(seed), loss_ct=torch.stack([res[r]["loss_ct"] for r in range(2)]),
         loss_rc=torch.stack([res[r]["loss_rc"] for r in range(2)]), dpred=torch.stack([res[r]["dpred"] for r in range(2)]),
         dproj_s=torch.stack([res[r]["dproj_s"] for r in range(2)]), dfc1_norm=torch.stack([res[r]["dfc1"].double().norm() for r in range(2)]))

def gen_neck_syncbn_2rank(seed=4400):
    """tests/golden/neck_syncbn_2rank.npz (round-2 review): the necks' TRAINING-mode SyncBatchNorm across two ranks.  torch's
    SyncBatchNorm refuses CPU tensors, but what it computes is fixed: batch statistics over the rows of ALL ranks, and a backward
    whose sums run over all ranks too -- i.e. the reference's own NonLinearNeck (cmae/models/necks/nonlinear_neck.py:35-102, norm
    layer = the BatchNorm1d that SyncBN is on one process) applied to the CONCATENATED 2B rows, with the loss the sum of both ranks'
    losses.  Stored: per-rank inputs and upstream gradients, the outputs, the input gradients (rows of the concatenated run), the
    parameter gradients of the concatenated run (= the SUM of the two ranks' local parameter gradients) and the running
    statistics.  Geometry: the shipped projector (hid 1536, out 256, with_bias, no last BN, no average pool) on 32 x 32 inputs."""
    from oracle import cmunet as OC
    MODELS, _, _ = import_cmae()
    B, S = 4, 32
    cfg = dict(type='NonLinearNeck', in_channels=S * S, hid_channels=1536, out_channels=256, num_layers=2, with_bias=True,
               with_last_bn=False, with_avg_pool=False)
    neck = MODELS.build(cfg)
    sd = OC.make_neck_sd("", S * S, 1536, 256, seed)
    assert set(neck.state_dict()) == set(sd)
    neck.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    neck.train()
    g = torch.Generator().manual_seed(seed + 1)
    xs = [torch.randn(B, 1, S, S, generator=g) for _ in range(2)]
    gos = [torch.randn(B, 1, 256, generator=g) for _ in range(2)]
    x_all = torch.cat(xs).requires_grad_(True)
    y_all = neck(x_all)
    (y_all * torch.cat(gos)).sum().backward()
    # oracle restatement on the same concatenated batch
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    xo = torch.cat(xs).requires_grad_(True)
    yo = OC.nonlinear_neck(xo, osd, "", training=True)
    (yo * torch.cat(gos)).sum().backward()
    close(yo.detach(), y_all.detach(), what="2-rank neck output")
    close(xo.grad, x_all.grad, tol=2e-4, what="2-rank neck dx")
    named = dict(neck.named_parameters())
    for k in ("fc0.weight", "fc0.bias", "bn0.weight", "bn0.bias", "fc1.weight"):
        close(osd[k].grad, named[k].grad, tol=2e-4, what=f"2-rank neck d{k}")
    save("neck_syncbn_2rank", seed=np.array(seed), B=np.array(B), S=np.array(S), x=torch.stack(xs), go=torch.stack(gos),
         y=y_all.detach().view(2, B, 1, 256), dx=x_all.grad.view(2, B, 1, S, S),
         dfc0_bias=named["fc0.bias"].grad, dbn0_weight=named["bn0.weight"].grad, dbn0_bias=named["bn0.bias"].grad,
         dfc1_weight_rows=named["fc1.weight"].grad[:16].clone(), dfc1_weight_norm=named["fc1.weight"].grad.double().norm(),
         dfc0_weight_norm=named["fc0.weight"].grad.double().norm(),
         dfc0_weight_rows=named["fc0.weight"].grad[:8].clone(),
         running_mean=neck.bn0.running_mean.clone(), running_var=neck.bn0.running_var.clone())


def gen_cldice(M):
    """soft_cldice of the reference (metrics.py:401-431, the driver's configuration train.py:464) on vessel-like masks."""
    from oracle import losses as OL
    g = torch.Generator().manual_seed(77)
    B, H, W = 2, 48, 56
    # curvilinear foreground: thresholded smooth noise + a few thin lines
    base = torch.nn.functional.avg_pool2d(torch.randn(B, 1, H, W, generator=g), 5, 1, 2)
    fg = (base[:, 0] > base.flatten(1).quantile(0.88, dim=1).view(B, 1, 1))
    fg[:, 10, :] = True
    fg[:, :, 23] = True
    y1h = torch.stack([~fg, fg], 1).double()
    logits = torch.randn(B, 2, H, W, generator=g) * 0.8 + (y1h.float() * 2 - 1) * torch.tensor([1.0, 1.6]).view(1, 2, 1, 1)
    ref = M.soft_cldice(threshold=0.5, activation="softmax", ignore_channels=[0])(logits, y1h)
    mine = OL.soft_cldice(logits, y1h)
    close(mine.float(), ref.float(), what="soft clDice")
    yp = (torch.softmax(logits, 1) > 0.5).float()[:, 1:2]
    skel = M.SoftSkeletonize(num_iter=10)(yp)
    close(OL.soft_skel(yp, 10), skel, what="soft skeleton")
    save("cldice", logits=logits, y1h=y1h, cldice=ref.detach(), skel_pred=skel, metric_name=np.array(M.soft_cldice().__name__))


def gen_optim():
    """LAMB: the reference's own class (Pretraining/Spark/utils/lamb.py, pure torch); SGD: torch.optim.SGD as MoCo configures it."""
    import importlib.util
    from oracle import optim as OO
    spec = importlib.util.spec_from_file_location("ref_lamb", os.path.join(REF, "Pretraining", "Spark", "utils", "lamb.py"))
    ref_lamb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_lamb)
    g = torch.Generator().manual_seed(91)
    shapes = [(24, 8, 3, 3), (24,), (5000,), (16, 24)]
    p0 = [torch.randn(s, generator=g) * 0.3 for s in shapes]
    grads = [[torch.randn(s, generator=g) * (3.0 if k == 0 else 0.2) for s in shapes] for k in range(3)]   # step 1 gets clipped
    wds = [0.05, 0.0, 0.05, 0.0]
    out = {"shapes": np.array([list(s) + [0] * (4 - len(s)) for s in shapes]), "wds": np.array(wds)}
    for i, t in enumerate(p0):
        out[f"p0.{i}"] = t
    for k in range(3):
        for i, t in enumerate(grads[k]):
            out[f"g{k}.{i}"] = t
    for tag, kw in (("a", dict(trust_clip=False, always_adapt=False)), ("b", dict(trust_clip=True, always_adapt=True))):
        ps = [torch.nn.Parameter(t.clone()) for t in p0]
        opt = ref_lamb.TheSameAsTimmLAMB([{"params": [ps[0], ps[2]], "weight_decay": 0.05}, {"params": [ps[1], ps[3]], "weight_decay": 0.0}],
                                        lr=2e-2, betas=(0.9, 0.98), eps=1e-6, max_grad_norm=2.0, **kw)
        om = [t.clone() for t in p0]
        ms, vs = [torch.zeros_like(t) for t in p0], [torch.zeros_like(t) for t in p0]
        for k in range(3):
            for prm, gr in zip(ps, grads[k]):
                prm.grad = gr.clone()
            opt.step()
            OO.lamb_step(om, grads[k], ms, vs, 2e-2, wds, betas=(0.9, 0.98), eps=1e-6, max_grad_norm=2.0, step=k + 1, **kw)
            for i in range(len(ps)):
                close(om[i], ps[i].detach(), tol=2e-6, what=f"lamb {tag} step {k} tensor {i}")
        for i, prm in enumerate(ps):
            out[f"lamb_{tag}.{i}"] = prm.detach().clone()
    for tag, kw in (("a", dict(momentum=0.9, weight_decay=1e-4)), ("b", dict(momentum=0.9, weight_decay=1e-2, nesterov=True)),
                    ("c", dict(momentum=0.0, weight_decay=0.0))):
        ps = [torch.nn.Parameter(t.clone()) for t in p0]
        opt = torch.optim.SGD(ps, lr=0.03, **kw)
        om, bufs = [t.clone() for t in p0], [torch.zeros_like(t) for t in p0]
        for k in range(3):
            for prm, gr in zip(ps, grads[k]):
                prm.grad = gr.clone()
            opt.step()
            OO.sgd_step(om, grads[k], bufs, 0.03, step=k + 1, **kw)
            for i in range(len(ps)):
                close(om[i], ps[i].detach(), tol=2e-6, what=f"sgd {tag} step {k} tensor {i}")
        for i, prm in enumerate(ps):
            out[f"sgd_{tag}.{i}"] = prm.detach().clone()
    # ---- SparK's per-iteration lr / weight-decay annealing: the reference's own lr_wd_annealing + get_param_groups
    # (Spark/utils/lr_control.py, pure Python) driving the reference's LAMB class over four steps (main.py:107-140,192) -----------
    spec2 = importlib.util.spec_from_file_location("ref_lr_control", os.path.join(REF, "Pretraining", "Spark", "utils", "lr_control.py"))
    ref_lrc = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(ref_lrc)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w0 = torch.nn.Parameter(p0[0].clone()); self.b0 = torch.nn.Parameter(p0[1].clone())
            self.w2 = torch.nn.Parameter(p0[2].clone().view(50, 100)); self.mask_token = torch.nn.Parameter(p0[3].clone())
    hold = Holder()
    import contextlib
    import io as _io
    with contextlib.redirect_stdout(_io.StringIO()):
        groups = ref_lrc.get_param_groups(hold, nowd_keys={"cls_token", "pos_embed", "mask_token", "gamma"})
    wd_scales = {id(q): gr["weight_decay_scale"] for gr in groups for q in gr["params"]}
    sched_scales = [wd_scales[id(q)] for q in (hold.w0, hold.b0, hold.w2, hold.mask_token)]
    assert sched_scales == [1.0, 0.0, 1.0, 0.0]           # 1-D tensors, biases and nowd keys: no decay (lr_control.py:39)
    opt = ref_lamb.TheSameAsTimmLAMB(groups, lr=2e-2, weight_decay=0.0, betas=(0.9, 0.95), max_grad_norm=5.0)
    peak, wd, wde, wp_it, max_it = 2e-2, 0.04, 0.2, 2, 9
    its = (0, 1, 2, 5)
    sched = []
    om = [t.clone() for t in p0]
    ms, vs = [torch.zeros_like(t) for t in p0], [torch.zeros_like(t) for t in p0]
    from cmunet_amd.pretrain import spark_lr_wd            # (host arithmetic of the product: asserted equal to the reference's here)
    for k, it in enumerate(its):
        mn_lr, mx_lr, mn_wd, mx_wd = ref_lrc.lr_wd_annealing(opt, peak, wd, wde, it, wp_it, max_it)
        sched.append((it, mx_lr, mx_wd))
        lr_p, wd_p = spark_lr_wd(peak, wd, wde, it, wp_it, max_it)
        assert abs(lr_p - mx_lr) < 1e-15 and abs(wd_p - mx_wd) < 1e-15 and mn_wd == 0.0
        gk = grads[k % 3]
        for prm, gr in zip((hold.w0, hold.b0, hold.w2, hold.mask_token), gk):
            prm.grad = gr.clone().view_as(prm)
        opt.step()
        OO.lamb_step(om, gk, ms, vs, mx_lr, [mx_wd * sc for sc in sched_scales], betas=(0.9, 0.95), eps=1e-6, max_grad_norm=5.0, step=k + 1)
        for i, prm in enumerate((hold.w0, hold.b0, hold.w2, hold.mask_token)):
            close(om[i], prm.detach().view_as(om[i]), tol=2e-6, what=f"lamb annealed step {k} tensor {i}")
    out["sched_args"] = np.array([peak, wd, wde, wp_it, max_it], dtype=np.float64)
    out["sched"] = np.array(sched, dtype=np.float64)
    out["sched_wd_scale"] = np.array(sched_scales)
    for i, prm in enumerate((hold.w0, hold.b0, hold.w2, hold.mask_token)):
        out[f"lamb_sched.{i}"] = prm.detach().clone().view_as(p0[i])
    # a table of the schedule itself over a whole run (lr_control.py:11-22)
    tab = []
    for (pk, w0_, w1_, wp, mx) in ((2e-4, 0.04, 0.2, 40.0, 1600), (1.0, 0.05, 0.05, 0.6, 11), (3e-3, 0.1, 0.0, 0, 5)):
        dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
        for it in sorted({0, 1, round(wp), round(wp) + 1, mx // 2, mx - 2, mx - 1}):
            if 0 <= it < mx:
                _, lr_, _, wd_ = ref_lrc.lr_wd_annealing(dummy, pk, w0_, w1_, it, wp, mx)
                tab.append((pk, w0_, w1_, wp, mx, it, lr_, wd_))
    out["sched_table"] = np.array(tab, dtype=np.float64)
    save("optim_traces", **out)


def gen_augment():
    """Input pipeline: expected outputs of Pillow itself (``Image.fromarray(f32).resize(size, BICUBIC)``, the
    call of cmunet_dataset.py:74-75) on small seeded arrays -- the pin of oracle/augment.py's restatement -- and of the
    reference's ShiftPixel / GaussNoise arithmetic (processing.py:118, auto_augment.py:1149-1153) written out with numpy."""
    from PIL import Image
    import PIL
    from oracle import augment as OA
    rng = np.random.RandomState(123)
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, oh, ow) in enumerate([(40, 56, 32, 24), (64, 64, 32, 32), (20, 30, 48, 40), (33, 33, 33, 20), (97, 15, 16, 16)]):
        a = (rng.standard_normal((h, w)) * 2 + 0.3).astype(np.float32)
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=Image.BICUBIC))
        assert np.array_equal(OA.resize_bicubic(a, oh, ow).view(np.uint32), ref.view(np.uint32)), (h, w, oh, ow)
        out[f"resize{i}.in"], out[f"resize{i}.out"] = a, ref
    # the same call on uint8 arrays (mode 'L': Finetuning/dataset.py:44-46) and the NEAREST resize of the masks (:47)
    rng8 = np.random.RandomState(321)
    for i, (h, w, oh, ow) in enumerate([(40, 56, 32, 24), (64, 64, 32, 32), (20, 30, 48, 40), (33, 33, 33, 20), (97, 15, 16, 16), (300, 200, 256, 256)]):
        a = rng8.randint(0, 256, (h, w)).astype(np.uint8)
        if i == 2:
            a = np.where(rng8.rand(h, w) < 0.5, 0, 255).astype(np.uint8)          # overshoot on both sides of the 0..255 clip
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=Image.BICUBIC))
        assert ref.dtype == np.uint8 and np.array_equal(OA.resize_bicubic_u8(a, oh, ow), ref), (h, w, oh, ow)
        m = rng8.randint(0, 2, (h, w)).astype(np.uint8)
        refm = np.asarray(Image.fromarray(m).resize((ow, oh), resample=Image.NEAREST))
        assert np.array_equal(OA.resize_nearest(m, oh, ow), refm), (h, w, oh, ow)
        out[f"resize_u8_{i}.in"], out[f"resize_u8_{i}.out"] = a, ref
        out[f"nearest{i}.in"], out[f"nearest{i}.out"] = m, refm
    # ShiftPixel + GaussNoise exactly as the reference's lines evaluate them
    img = rng.standard_normal((2, 40, 40)).astype(np.float32)
    shifts = np.array([[3, 7], [8, 0]], np.int32)
    noise = rng.standard_normal((2, 32, 32))
    exp_t = []
    for b in range(2):
        v = img[b][0 + shifts[b, 0]:shifts[b, 0] + 32, 0 + shifts[b, 1]:shifts[b, 1] + 32]       # processing.py:118
        sigma = np.max(v) / 10                                                                  # auto_augment.py:1149
        exp_t.append(np.array(v + sigma * noise[b], dtype=v.dtype))                             # :1151-1152
    o_img, o_t = OA.two_view(img, shifts, noise, 32)
    assert np.array_equal(o_t, np.stack(exp_t)) and np.array_equal(o_img, img[:, :32, :32])
    out.update(tv_in=img, tv_shifts=shifts, tv_noise=noise, tv_img_t=np.stack(exp_t))
    save("augment", **out)


def main():
    from oracle import unet as OU, losses as OL
    if "--only-augment" in sys.argv:    # needs Pillow only, not the reference
        gen_augment()
        return
    ref, M = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(4)
    if "--only-spark75b8" in sys.argv:  # tests/golden/spark_unet_m75_b8.npz alone (config 5's mask ratio at batch 8)
        gen_spark(None, S=128, ratio=0.75, name="spark_unet_m75_b8", seed=271, B=8)
        sys.exit(0)
    if "--only-spark75" in sys.argv:    # tests/golden/spark_unet_m75.npz alone (BASELINE config 5's mask ratio)
        gen_spark(None, S=128, ratio=0.75, name="spark_unet_m75", seed=171)
        return
    if "--only-spark" in sys.argv:      # regenerate tests/golden/spark_unet.npz alone
        gen_spark(ref)
        return
    if "--only-optim" in sys.argv:      # regenerate tests/golden/optim_traces.npz alone
        gen_optim()
        return
    if "--only-cldice" in sys.argv:
        gen_cldice(M)
        return
    if "--only-loadmodel" in sys.argv:  # tests/golden/load_model_ref.npz alone (the reference's own load_model on the five layouts)
        gen_load_model()
        return
    if "--only-finetune-sensitivity" in sys.argv:   # tests/golden/finetune_sensitivity.npz (needs finetune_ref.npz)
        gen_finetune_sensitivity(ref, M)
        return
    if "--only-finetune" in sys.argv:   # tests/golden/finetune_ref.npz alone (the reference's own TrainEpoch / ValidEpoch / train())
        gen_finetune(ref, M)
        return
    if "--head-worker" in sys.argv:     # (child of gen_cmunet_head_2rank)
        i = sys.argv.index("--head-worker")
        _cmunet_head_2rank_worker(int(sys.argv[i + 1]), int(sys.argv[i + 2]), sys.argv[i + 3], int(sys.argv[i + 4]))
        return
    if "--only-neck2" in sys.argv:      # tests/golden/neck_syncbn_2rank.npz alone
        gen_neck_syncbn_2rank()
        return
    if "--only-head2" in sys.argv:      # tests/golden/cmunet_head_2rank.npz alone
        gen_cmunet_head_2rank()
        return
    if "--moco-worker" in sys.argv:     # (child of gen_moco_2rank)
        i = sys.argv.index("--moco-worker")
        _moco_2rank_worker(int(sys.argv[i + 1]), int(sys.argv[i + 2]), sys.argv[i + 3], int(sys.argv[i + 4]))
        return
    if "--only-moco2" in sys.argv:      # tests/golden/moco_ref_2rank.npz alone
        gen_moco_2rank()
        return
    if "--only-mocoval" in sys.argv:    # tests/golden/moco_val_ref.npz alone (the reference's validation_step)
        gen_moco_val()
        return
    if "--only-moco" in sys.argv:       # tests/golden/moco_ref.npz alone (the reference's Moco_v2 behind the lightning stand-in)
        gen_moco()
        return
    if "--only-cmunet" in sys.argv:     # tests/golden/cmunet_ref.npz alone (the reference's cmae modules behind the mmengine stand-in)
        gen_cmunet()
        return

    def load(mod, sd):
        missing = mod.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        return missing

    # ---- 1. DoubleConv ------------------------------------------------------------------
    for tag, cin, cout, shape in (("a", 1, 16, (2, 20, 24)), ("b", 16, 32, (3, 16, 16)), ("c", 8, 8, (2, 7, 9))):
        g = torch.Generator().manual_seed(100 + cin)
        full = OU.make_state_dict(base_ch=cout, depth=1, in_ch=cin, seed=11 + cin, decoder=False)
        sd = {k[len("double_conv."):]: v for k, v in full.items()}        # keys 'double_conv.0.weight' ...
        m = ref.DoubleConv(cin, cout)
        load(m, sd)
        m.train()
        x = torch.randn(shape[0], cin, shape[1], shape[2], generator=g, requires_grad=True)
        go = torch.randn(shape[0], cout, shape[1], shape[2], generator=g)
        y = m(x)
        (y * go).sum().backward()
        osd = OU.clone_sd(sd, requires_grad=True)
        xo = x.detach().clone().requires_grad_(True)
        yo = OU.double_conv(xo, osd, "double_conv.", True)
        (yo * go).sum().backward()
        close(yo, y, what=f"double_conv_{tag} y")
        close(xo.grad, x.grad, what=f"double_conv_{tag} dx")
        for k, p in m.named_parameters():
            close(osd[k].grad, p.grad, tol=1e-4, what=f"double_conv_{tag} d{k}")
        close(osd["double_conv.1.running_var"], m.double_conv[1].running_var, what="running_var")
        out = {"x": x, "go": go, "y": y, "dx": x.grad}
        out.update({"sd." + k: v for k, v in sd.items()})
        out.update({"grad." + k: v for k, v in grads_of(m).items()})
        out.update({"after." + k: v for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
        save(f"double_conv_{tag}", **out)

    # ---- 2. DownBlock -------------------------------------------------------------------
    g = torch.Generator().manual_seed(200)
    full = OU.make_state_dict(base_ch=32, depth=1, in_ch=16, seed=21, decoder=False)
    sd = dict(full)                                                       # 'double_conv.double_conv.0.weight'
    m = ref.DownBlock(16, 32)
    load(m, sd)
    m.train()
    x = torch.randn(2, 16, 16, 16, generator=g, requires_grad=True)
    gd = torch.randn(2, 32, 8, 8, generator=g)
    gs = torch.randn(2, 32, 16, 16, generator=g)
    down, skip = m(x)
    ((down * gd).sum() + (skip * gs).sum()).backward()
    osd = OU.clone_sd(sd, requires_grad=True)
    xo = x.detach().clone().requires_grad_(True)
    d2, s2 = OU.down_block(xo, osd, "", True)
    ((d2 * gd).sum() + (s2 * gs).sum()).backward()
    close(d2, down, what="down_block down"); close(s2, skip, what="down_block skip"); close(xo.grad, x.grad, what="down_block dx")
    out = {"x": x, "gd": gd, "gs": gs, "down": down, "skip": skip, "dx": x.grad}
    out.update({"sd." + k: v for k, v in sd.items()})
    out.update({"grad." + k: v for k, v in grads_of(m).items()})
    save("down_block", **out)

    # ---- 3. UpBlock (conv_transpose) ----------------------------------------------------
    g = torch.Generator().manual_seed(300)
    full = OU.make_state_dict(base_ch=16, depth=2, in_ch=1, seed=31, encoder=False)
    sd = {k[len("up_conv1."):]: v for k, v in full.items() if k.startswith("up_conv1.")}
    m = ref.UpBlock(32, 16, "conv_transpose")
    load(m, sd)
    m.train()
    xd = torch.randn(2, 32, 8, 8, generator=g, requires_grad=True)
    xs = torch.randn(2, 16, 16, 16, generator=g, requires_grad=True)
    go = torch.randn(2, 16, 16, 16, generator=g)
    y = m(xd, xs)
    (y * go).sum().backward()
    osd = OU.clone_sd(sd, requires_grad=True)
    xdo, xso = xd.detach().clone().requires_grad_(True), xs.detach().clone().requires_grad_(True)
    yo = OU.up_block(xdo, xso, osd, "", "conv_transpose", True)
    (yo * go).sum().backward()
    close(yo, y, what="up_block y"); close(xdo.grad, xd.grad, what="up_block dxd"); close(xso.grad, xs.grad, what="up_block dxs")
    out = {"xd": xd, "xs": xs, "go": go, "y": y, "dxd": xd.grad, "dxs": xs.grad}
    out.update({"sd." + k: v for k, v in sd.items()})
    out.update({"grad." + k: v for k, v in grads_of(m).items()})
    save("up_block", **out)

    # bad up_sample_mode -> ValueError (model.py:64)
    try:
        ref.UpBlock(4, 2, "nearest")
        raise AssertionError("reference accepted a bad up_sample_mode")
    except ValueError as e:
        bad_mode_msg = str(e)

    # ---- 4. UNet-small: the reference's own blocks composed in the reference's order -----
    class SmallUNet(torch.nn.Module):          # base 16, depth 3 (BASELINE config 1)
        def __init__(self):
            super().__init__()
            self.down_conv1 = ref.DownBlock(1, 16)
            self.down_conv2 = ref.DownBlock(16, 32)
            self.double_conv = ref.DoubleConv(32, 64)
            self.up_conv2 = ref.UpBlock(64, 32, "conv_transpose")
            self.up_conv1 = ref.UpBlock(32, 16, "conv_transpose")
            self.conv_last = torch.nn.Conv2d(16, 2, kernel_size=1)

        def forward(self, x):
            x = x.unsqueeze(1)
            x, s1 = self.down_conv1(x)
            x, s2 = self.down_conv2(x)
            x = self.double_conv(x)
            x = self.up_conv2(x, s2)
            x = self.up_conv1(x, s1)
            return self.conv_last(x)

    g = torch.Generator().manual_seed(400)
    sd = OU.make_state_dict(base_ch=16, depth=3, seed=41)
    m = SmallUNet()
    load(m, sd)
    m.train()
    x = torch.randn(2, 32, 48, generator=g)
    fg = (torch.rand(2, 32, 48, generator=g) > 0.8)
    y1h = torch.stack([~fg, fg], 1).to(torch.float64)                      # dataset.py:48 one-hot float64
    crit = M.DiceLoss(activation="softmax", threshold=0.5, ignore_channels=[0]) + M.CrossEntropyLoss()
    logits = m(x)
    loss = crit(logits, y1h)
    loss.backward()
    osd = OU.clone_sd(sd, requires_grad=True)
    lo = OU.unet_forward(x, osd, training=True)
    losso = OL.dice_ce_loss(lo, y1h)
    losso.backward()
    close(lo, logits, what="unet_small logits")
    close(losso.float(), loss.float(), what="unet_small loss")
    for k, p in m.named_parameters():
        close(osd[k].grad, p.grad, tol=2e-4, what=f"unet_small d{k}")
    out = {"x": x, "y1h": y1h, "logits": logits, "loss": loss.detach(), "crit_name": np.array(crit.__name__)}
    out.update({"sd." + k: v for k, v in sd.items()})
    out.update({"grad." + k: v for k, v in grads_of(m).items()})
    out.update({"after." + k: v for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    m.eval()
    with torch.no_grad():
        out["logits_eval"] = m(x)
    osd_e = {k: v.detach().clone() for k, v in m.state_dict().items()}
    close(OU.unet_forward(x, osd_e, training=False), out["logits_eval"], what="unet_small eval logits")
    save("unet_small", **out)

    # ---- 4b. two-step Adam trace on UNet-small (restated batch_update, train.py:163-169) ---
    m = SmallUNet()
    load(m, sd)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for step in range(2):
        opt.zero_grad()
        pred = m(x)
        l = crit(pred, y1h)
        l.backward()
        opt.step()
        losses.append(l.detach().clone())
    save("unet_small_adam_trace", x=x, y1h=y1h, losses=torch.stack(losses),
         conv_last_weight=m.conv_last.weight.detach(), first_conv_weight=m.down_conv1.double_conv.double_conv[0].weight.detach(),
         bott_bn_running_var=m.double_conv.double_conv[4].running_var)

    # ---- 5. full reference UNet(), weights by seed (regenerated on the test side) ----------
    g = torch.Generator().manual_seed(500)
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=51)
    m = ref.UNet()
    assert sum(p.numel() for p in m.parameters()) == 31042434
    assert set(m.state_dict().keys()) == set(sd.keys()), "state_dict names differ from the reference"
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape) and v.dtype == sd[k].dtype, k
    load(m, sd)
    m.train()
    x = torch.randn(2, 32, 32, generator=g)
    go = torch.randn(2, 2, 32, 32, generator=g)
    logits = m(x)
    (logits * go).sum().backward()
    osd = OU.clone_sd(sd, requires_grad=True)
    lo = OU.unet_forward(x, osd, training=True)
    (lo * go).sum().backward()
    close(lo, logits, what="unet_full logits")
    gn = {}
    for k, p in m.named_parameters():
        close(osd[k].grad, p.grad, tol=5e-4, what=f"unet_full d{k}")
        gn[k] = p.grad.norm()
    keys = sorted(gn.keys())
    save("unet_full", x=x, go=go, logits=logits, seed=np.array(51), grad_norm_keys=np.array(keys),
         grad_norms=torch.stack([gn[k] for k in keys]),
         **{"grad." + k: m.get_parameter(k).grad for k in
            ("conv_last.weight", "conv_last.bias", "down_conv1.double_conv.double_conv.0.weight",
             "down_conv1.double_conv.double_conv.1.weight", "up_conv1.up_sample.bias")},
         state_keys=np.array(list(m.state_dict().keys())),
         state_shapes=np.array([str(tuple(v.shape)) for v in m.state_dict().values()]))

    # ---- 6. losses -------------------------------------------------------------------------
    g = torch.Generator().manual_seed(600)
    logits = torch.randn(2, 2, 40, 40, generator=g, requires_grad=True)
    fg = torch.rand(2, 40, 40, generator=g) > 0.9
    y1h = torch.stack([~fg, fg], 1).to(torch.float64)
    dice = M.DiceLoss(activation="softmax", threshold=0.5, ignore_channels=[0])(logits, y1h)
    ce = M.CrossEntropyLoss()(logits, y1h)
    iou = M.IoU(threshold=0.5, activation="softmax", ignore_channels=[0])(logits, y1h)
    tot = (M.DiceLoss(activation="softmax", threshold=0.5, ignore_channels=[0]) + M.CrossEntropyLoss())(logits, y1h)
    tot.backward()
    lo = logits.detach().clone().requires_grad_(True)
    close(OL.dice_loss(lo, y1h).float(), dice.float(), what="dice")
    close(OL.cross_entropy_prob(lo, y1h).float(), ce.float(), what="ce")
    close(OL.iou_loss(lo, y1h).float(), iou.float(), what="iou")
    OL.dice_ce_loss(lo, y1h).backward()
    close(lo.grad, logits.grad, what="dlogits")
    save("losses", logits=logits, y1h=y1h, dice=dice.detach(), ce=ce.detach(), iou=iou.detach(), total=tot.detach(),
         dlogits=logits.grad, bad_mode_msg=np.array(bad_mode_msg))
    # ---- 7. SparK (sparse masked conv) -- reference imported behind the stubs of SURVEY Appendix C-3 ---------
    gen_spark(ref)
    gen_spark(ref, S=128, ratio=0.75, name="spark_unet_m75", seed=171)
    gen_spark(ref, S=128, ratio=0.75, name="spark_unet_m75_b8", seed=271, B=8)
    gen_optim()
    gen_cldice(M)
    print("all fixtures written; oracle == reference on every case")


if __name__ == "__main__":
    main()
