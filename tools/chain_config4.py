#!/usr/bin/env python3
"""BASELINE config 4 as a CHAIN, measured: joint CM-UNet pretraining (contrastive + masked reconstruction) -> the mmengine-layout
checkpoint -> ``load_model`` -> 3-fold finetuning on an 18-image split -> Dice and images/s of both legs.

    python tools/chain_config4.py [--pre-steps 20] [--pre-size 512] [--pre-batch 32] [--ft-images 18] [--ft-size 256]
                                  [--ft-epochs 8] [--ft-batch 6] [--lr 1e-3] [--base-ch 64] [--depth 5] [--pre-dtype f16] [--ft-dtype f32]

What the reference spreads over Pretraining/CM-UNet/training/train.py (mmengine Runner + CheckpointHook) and
Finetuning/train.py:311-378 (``main_finetuning``: load_model once, KFold(3, shuffle, random_state 42), a fresh Adam per fold,
best-validation checkpoint per fold; README.md:14's "18 images" setting) runs here on synthetic data (seeded random images for the
pretraining leg, synthetic vessel masks for the finetuning leg: the FAME2 data are private).  Rank 0 / one GPU; prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_chain(pre_steps=20, pre_size=512, pre_batch=32, ft_images=18, ft_size=256, ft_epochs=8, ft_batch=6, lr=1e-3, base_ch=64,
              depth=5, pre_dtype="f16", ft_dtype="f32", device="cuda:0", seed=0, work_dir=None, verbose=False):
    import contextlib
    import io
    import numpy as np
    import torch
    from cmunet_amd import cmunet as C, metrics as M, pretrain as P, train as T
    from cmunet_amd.dataset import SyntheticSegmentationDataset
    dev = torch.device(device)
    torch.manual_seed(seed)
    out = {}
    # ---- leg 1: joint pretraining (cmunet.py:108-135 under cmunet_config.py:76-114) -------------------------------------------
    model = C.build_model(C.cmunet_config(img_size=pre_size, dtype=pre_dtype, mask_ratio=0.6, base_ch=base_ch, depth=depth)).to(dev)
    model.init_weights()
    tr = P.JointPretrainer(model, lr=1.5e-4 * pre_batch / 256.0, amp=(pre_dtype == "f16"))
    g = torch.Generator(device=dev).manual_seed(1234 + seed)
    nb = 2
    xs = [(torch.randn(pre_batch, pre_size, pre_size, generator=g, device=dev), torch.randn(pre_batch, pre_size, pre_size, generator=g, device=dev))
          for _ in range(nb)]
    masks = [P.random_patch_mask_device(pre_batch, pre_size, pre_size, 16, 0.6, g, dev) for _ in range(nb)]
    warm = min(3, pre_steps)
    for i in range(warm):
        tr.step(xs[i % nb][0], xs[i % nb][1], masks[i % nb], cur_iter=i, max_iter=pre_steps + warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(pre_steps):
        l = tr.step(xs[i % nb][0], xs[i % nb][1], masks[i % nb], cur_iter=warm + i, max_iter=pre_steps + warm)
    torch.cuda.synchronize()
    dt_pre = time.perf_counter() - t0
    out["pretrain"] = {"steps": pre_steps, "images_per_s": round(pre_batch * pre_steps / dt_pre, 2), "ms_per_step": round(1e3 * dt_pre / max(1, pre_steps), 2),
                       "loss_ct": float(l["loss_ct"]), "loss_rc": float(l["loss_rc"]), "dtype": pre_dtype, "size": pre_size, "batch": pre_batch}
    # ---- the checkpoint the reference's finetuning driver reads (mmengine CheckpointHook layout, train.py:262-273) -----------------
    msd = model.state_dict()
    unet_sd = {k[len("backbone."):]: v for k, v in msd.items() if k.startswith("backbone.")}
    unet_sd.update({k[len("pixel_decoder."):]: v for k, v in msd.items() if k.startswith("pixel_decoder.")})
    tmp = work_dir or tempfile.mkdtemp(prefix="chain4_")
    os.makedirs(tmp, exist_ok=True)
    ckpt = os.path.join(tmp, "cmunet_pretrained.pth")
    T.export_checkpoint(unet_sd, ckpt, "cmunet", epoch=1, extra={"iter": pre_steps + warm})
    pre_enc = {k: v.detach().cpu().clone() for k, v in unet_sd.items()}
    del tr, model, xs, masks
    torch.cuda.empty_cache()
    # ---- leg 2: load_model + 3-fold finetuning (train.py:240-308, 311-378) -----------------------------------------------------------
    args = T.get_args(["-p", ckpt, "-e", str(ft_epochs), "-b", str(ft_batch), "-l", str(lr), "-n", "chain4", "--dtype", ft_dtype])
    args.base_ch, args.depth = base_ch, depth
    ds = SyntheticSegmentationDataset(n=ft_images, size=ft_size, seed=42 + seed)
    items = [ds[i] for i in range(ft_images)]

    def make_loaders(train_idx, val_idx, BATCH):
        def batches(idx, shuffle_seed=None):
            idx = list(idx)
            if shuffle_seed is not None:
                np.random.RandomState(shuffle_seed).shuffle(idx)
            return [(torch.from_numpy(np.stack([items[i][0] for i in idx[j:j + BATCH]])), torch.from_numpy(np.stack([items[i][1] for i in idx[j:j + BATCH]])))
                    for j in range(0, len(idx), BATCH)]
        return batches(train_idx, 7), batches(val_idx)

    mk = dict(activation="softmax", threshold=0.5, ignore_channels=[0])
    crit = M.DiceLoss(**mk) + M.CrossEntropyLoss()
    mets = [M.DiceLoss(**mk), M.CrossEntropyLoss(), M.IoU(**mk)]
    loaded = T.load_model(args)                         # (what main_finetuning does first; here also to report the loaded keys)
    lsd = loaded.state_dict()
    same = [k for k in pre_enc if k in lsd and not k.startswith("conv_last") and torch.equal(lsd[k].cpu(), pre_enc[k])]
    out["checkpoint"] = {"layout": "cmunet (mmengine)", "keys_loaded": len(same), "keys_of_unet": len(lsd),
                         "head_reinitialised": not torch.equal(lsd["conv_last.weight"].cpu(), pre_enc["conv_last.weight"])}
    sink = io.StringIO()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with (contextlib.nullcontext() if verbose else contextlib.redirect_stdout(sink)):
        best, result = T.main_finetuning(args, crit, mets, str(dev), None, list(range(ft_images)), list(range(ft_images)),
                                         make_loaders=make_loaders, work_dir=tmp, save_best=False, keep_models=True)
    torch.cuda.synchronize()
    dt_ft = time.perf_counter() - t0
    n_train, n_val = 2 * ft_images // 3, ft_images // 3
    imgs = 3 * ft_epochs * (n_train + n_val)            # images through the network (training + validation passes, three folds)
    folds = [min(v["dice_loss"] for v in r["valid_logs_list"]) for r in result]
    out["finetune"] = {"folds": 3, "epochs": ft_epochs, "batch": ft_batch, "lr": lr, "images": ft_images, "size": ft_size, "dtype": ft_dtype,
                       "images_per_s": round(imgs / dt_ft, 2), "seconds": round(dt_ft, 2),
                       "best_valid_dice_per_fold": [round(1.0 - d, 4) for d in folds], "mean_best_valid_dice": round(1.0 - float(np.mean(folds)), 4),
                       "first_epoch_valid_dice_fold1": round(1.0 - result[0]["valid_logs_list"][0]["dice_loss"], 4)}
    out["_model"] = result[-1]["model"]
    out["_val_batches"] = make_loaders(*T.kfold_indices(ft_images, 3, 42)[-1], ft_batch)[1]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pre-steps", type=int, default=20)
    ap.add_argument("--pre-size", type=int, default=512)
    ap.add_argument("--pre-batch", type=int, default=32)
    ap.add_argument("--ft-images", type=int, default=18)
    ap.add_argument("--ft-size", type=int, default=256)
    ap.add_argument("--ft-epochs", type=int, default=8)
    ap.add_argument("--ft-batch", type=int, default=6)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--base-ch", type=int, default=64)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--pre-dtype", default="f16")
    ap.add_argument("--ft-dtype", default="f32")
    a = ap.parse_args()
    r = run_chain(a.pre_steps, a.pre_size, a.pre_batch, a.ft_images, a.ft_size, a.ft_epochs, a.ft_batch, a.lr, a.base_ch, a.depth,
                  a.pre_dtype, a.ft_dtype)
    r = {k: v for k, v in r.items() if not k.startswith("_")}
    r["workload"] = "BASELINE config 4 chain: joint CM-UNet pretrain -> cmunet checkpoint -> load_model -> 3-fold finetune (synthetic data)"
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
