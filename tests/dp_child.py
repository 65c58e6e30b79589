"""Child process of tests/test_gpu_dataparallel.py (not a test module): one data-parallel rank of a pretraining trainer.

    python dp_child.py <mode> <out_prefix> [key=value ...]

Runs under gloo with every rank on cuda:0 (the one-GPU box) -- or RCCL with one device per rank when CMU_DIST_BACKEND=nccl
and enough GPUs exist.  Every rank is fed the SAME seeded batch unless data=rank; the result (losses, a selection of updated
parameters, buffers) goes to <out_prefix>.<rank>.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402
import torch.distributed as dist   # noqa: E402


def main():
    mode, out = sys.argv[1], sys.argv[2]
    opts = dict(kv.split("=", 1) for kv in sys.argv[3:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = os.environ.get("CMU_DIST_BACKEND", "gloo")
    dev = torch.device("cuda", rank if (backend == "nccl" and torch.cuda.device_count() > rank) else 0)
    torch.cuda.set_device(dev)
    rehearse = world == 1 and os.environ.get("CMU_DP_REHEARSE") == "1"       # a one-rank group that runs every collective
    if world > 1 or rehearse:
        if rehearse:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", opts.get("port", "29531"))
        dist.init_process_group(backend, rank=rank, world_size=world)
    from cmunet_amd import pretrain as P
    if opts.get("late") == "1":
        # a gradient written into a bucket BEHIND its all-reduce (what CMU_DP_CHECK_LATE_WRITES=1 must catch): the first early exchange
        # of the run is followed by a write into its range
        fired = []

        def late(flat, lo, hi):
            if not fired:
                fired.append((lo, hi))
                flat.grad[lo:lo + 1] += 1.0
        P._TEST_AFTER_LAUNCH = late
    try:
        _run(mode, out, opts, world, rank, dev, P)
    except RuntimeError as e:
        if opts.get("late") != "1":
            raise
        torch.save({"raised": str(e)}, f"{out}.{rank}")
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _run(mode, out, opts, world, rank, dev, P):
    from oracle import unet as OU
    steps = int(opts.get("steps", "2"))
    dseed = 100 + (rank if opts.get("data") == "rank" else 0)
    g = torch.Generator().manual_seed(dseed)
    res = {}
    if mode == "recon":
        from cmunet_amd import model as M
        net = M.UNet(base_ch=16, depth=3, dtype=opts.get("dtype", "f32"))
        # init=rank: every rank starts from its own weights (what DDP's construction-time broadcast is for)
        net.load_state_dict(OU.make_state_dict(base_ch=16, depth=3, seed=9 + (rank if opts.get("init") == "rank" else 0)))
        tr = P.MaskedReconPretrainer(net.to(dev), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, amp=opts.get("amp") == "1")
        if opts.get("prefwd") == "1":
            # a forward BEFORE the broadcast fills the engine's packed-weight caches with this rank's own weights
            with torch.no_grad():
                net.eval()
                net(torch.randn(2, 32, 64, generator=torch.Generator().manual_seed(1)).to(dev))
                net.train()
        tr.broadcast_parameters()
        losses = []
        for it in range(steps):
            x = torch.randn(2, 32, 64, generator=g).to(dev)
            mask = torch.from_numpy(P.create_random_patch_mask(2, 32, 16, 0.5, np.random.RandomState(dseed + it))).to(dev)
            mask = torch.cat([mask, mask], 2).contiguous()
            losses.append(float(tr.step(x, mask)))
        res = {"losses": losses, "arena": tr.flat.arena.cpu(), "bufs": {n: b.cpu() for n, b in net.named_buffers()}}
    elif mode == "joint":
        from cmunet_amd import cmunet as C
        torch.manual_seed(0)
        B, S = 4, 32
        model = C.build_model(C.cmunet_config(img_size=S, dtype=opts.get("dtype", "f32"), base_ch=16, depth=3)).to(dev).train()
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() == 1 and ("bn" in n or ".1." in n or ".4." in n):
                    p.add_(0.2 * torch.randn_like(p))
            for pb, pm in zip(model.backbone.parameters(), model.target_backbone.parameters()):
                pm.copy_(pb * 0.9)
            for pb, pm in zip(model.projector.parameters(), model.target_projector.parameters()):
                pm.copy_(pb * 1.1)
        res["init"] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        tr = P.JointPretrainer(model, lr=1e-3)
        tr.broadcast_parameters()
        img, img_t = torch.randn(B, S, S, generator=g), torch.randn(B, S, S, generator=g)
        mask = torch.from_numpy(P.create_random_patch_mask(B, S, 16, 0.65, np.random.RandomState(2)))
        Cr = model.reduced_channels()
        gw = torch.Generator().manual_seed(7)
        rw, rb = torch.randn(Cr, 64, 1, 1, generator=gw) * 0.1, torch.randn(Cr, generator=gw) * 0.1
        model.momentum = 0.9
        for _ in range(int(opts.get("steps", 1))):        # (steps > 1: the sharded exchange's parameter all-gather completes inside the next forward)
            l = tr.step(img.to(dev), img_t.to(dev), mask.to(dev), reduce_w=rw.to(dev), reduce_b=rb.to(dev))
        tr.finish_pending()
        res["sharded"] = tr._shard is not None
        res["opt"] = {k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in tr.state_dict()["optimizer"].items() if k in ("m", "v", "step")}
        res.update({"loss_ct": float(l["loss_ct"]), "loss_rc": float(l["loss_rc"]), "img": img, "img_t": img_t, "mask": mask, "rw": rw, "rb": rb,
                    "final": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                    "exchange": getattr(tr, "last_exchange", None)})
    elif mode == "moco":
        from cmunet_amd import moco as MO
        torch.manual_seed(0)
        B, S, K, T = 4, 32, 64, 0.2
        # (shuffle_bn off: with the same batch on both ranks the shuffled shares would hold duplicates -- the shuffle has its own
        # test, mode moco_shuffle)
        m = MO.Moco_v2(emb_dim=64, num_negatives=K, softmax_temperature=T, encoder_momentum=0.99, learning_rate=0.05, dtype="f32",
                       base_ch=16, depth=3, shuffle_bn=False).to(dev).train()
        with torch.no_grad():
            for pk in m.encoder_k.parameters():
                pk.mul_(0.95)
        res["init"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        tr = P.MocoPretrainer(m)
        tr.broadcast_parameters()
        xq, xk = torch.randn(B, 1, S, S, generator=g), torch.randn(B, 1, S, S, generator=g)
        loss = tr.step(xq.to(dev), xk.to(dev))
        res.update({"loss": float(loss), "xq": xq, "xk": xk, "final": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    "exchange": getattr(tr, "last_exchange", None)})
    elif mode == "moco_shuffle":
        # shuffle-BN (moco2_module.py:177-222): every rank has its own key images; rank 0's permutation comes from the global
        # CPU generator, seeded here so that the test can redraw it
        from cmunet_amd import moco as MO
        torch.manual_seed(0)
        B, S = 4, 32
        m = MO.Moco_v2(emb_dim=64, num_negatives=64, softmax_temperature=0.2, dtype="f32", base_ch=16, depth=3).to(dev).train()
        res["init"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        gk = torch.Generator().manual_seed(500 + rank)
        xq, xk = torch.randn(B, 1, S, S, generator=gk), torch.randn(B, 1, S, S, generator=gk)
        torch.manual_seed(1234)
        _, _, k, _ = m(xq.to(dev), xk.to(dev), m.queue)
        res.update({"k": k.cpu(), "xk": xk})
    elif mode == "head_ref2":
        # the reference's own two-rank head run (tests/golden/cmunet_head_2rank.npz): predictor BatchNorm in eval mode
        from cmunet_amd import cmunet as C
        from oracle import cmunet as OC
        seed = int(opts["seed"])
        cfg = C.cmunet_config(img_size=32, dtype="f32")["head"]
        head = C.build_model(cfg).to(dev)
        hsd = OC.make_neck_sd("predictor.", 256, 1536, 256, seed)
        g0 = torch.Generator().manual_seed(seed + 1)
        hsd["predictor.bn0.running_mean"] = 0.1 * torch.randn(1536, generator=g0)
        hsd["predictor.bn0.running_var"] = 0.5 + torch.rand(1536, generator=g0)
        head.load_state_dict({k: v.clone() for k, v in hsd.items()}, strict=True)
        head.train()
        head.predictor.bn0.eval()
        x, pred, mk, ps, pt = (t.to(dev) for t in OC.head_fixture_inputs(seed + 10 * (rank + 1)))
        logits = torch.stack([torch.zeros_like(pred), pred], 1).contiguous().requires_grad_(True)
        ps.requires_grad_(True)
        hl = head(x, logits, mk, ps, pt)
        (hl["loss_ct"] + hl["loss_rc"]).backward()
        res = {"loss_ct": float(hl["loss_ct"].detach()), "loss_rc": float(hl["loss_rc"].detach()), "dpred": logits.grad[:, 1].cpu(), "dproj_s": ps.grad.cpu(),
               "dfc1_norm": float(head.predictor.fc1.weight.grad.double().norm())}
    elif mode == "neck_ref2":
        # tests/golden/neck_syncbn_2rank.npz: the reference's NonLinearNeck over the concatenated rows of both ranks = SyncBN semantics
        from cmunet_amd import cmunet as C
        from oracle import cmunet as OC
        f = np.load(os.path.join(ROOT, "tests", "golden", "neck_syncbn_2rank.npz"))
        S = int(f["S"])
        neck = C.NonLinearNeck(in_channels=S * S, hid_channels=1536, out_channels=256, num_layers=2, with_bias=True, with_last_bn=False,
                               with_avg_pool=False).to(dev).train()
        neck.load_state_dict({k: v.clone() for k, v in OC.make_neck_sd("", S * S, 1536, 256, int(f["seed"])).items()}, strict=True)
        x = torch.from_numpy(f["x"][rank]).to(dev).requires_grad_(True)
        y = neck(x)
        (y * torch.from_numpy(f["go"][rank]).to(dev)).sum().backward()
        res = {"y": y.detach().cpu(), "dx": x.grad.cpu(), "grads": {n: p.grad.cpu() for n, p in neck.named_parameters()},
               "running_mean": neck.bn0.running_mean.cpu(), "running_var": neck.bn0.running_var.cpu()}
    elif mode == "moco_ref2":
        # the reference's own two-rank run (tests/golden/moco_ref_2rank.npz): the same seeded state, every rank its own images,
        # rank 0's permutation from the global CPU generator seeded as the generator seeded it
        from cmunet_amd import moco as MO
        from oracle import moco as OM
        seed, B, S, K = int(opts["seed"]), 4, 64, 64
        m = MO.Moco_v2(emb_dim=1024, num_negatives=K, softmax_temperature=0.2, encoder_momentum=0.99, dtype="f32").to(dev).train()
        m.load_state_dict({k: v.clone() for k, v in OM.make_moco_sd(seed, K).items()}, strict=False)
        xq, xk, _, _ = OM.moco_fixture_inputs(seed + 50 * (rank + 1), B, S)
        torch.manual_seed(seed + 7)
        loss = m.training_step(((xq.to(dev), xk.to(dev)), 0))
        loss.backward()
        named = dict(m.named_parameters())
        qkeys = sorted(k for k, p in named.items() if p.requires_grad)
        res = {"loss": float(loss), "queue": m.queue.cpu(), "ptr": int(m.queue_ptr), "qkeys": qkeys,
               "grad_norms": torch.stack([named[k].grad.double().norm().cpu() for k in qkeys]),
               "grad0": named["encoder_q.down_conv1.double_conv.double_conv.0.weight"].grad.cpu(),
               "bn_k": m.state_dict()["encoder_k.down_conv1.double_conv.double_conv.1.running_mean"].cpu()}
    elif mode == "spark":
        from cmunet_amd import spark as S
        torch.manual_seed(5)
        enc = S.build_sparse_encoder("unet_sparse", input_size=64, sbn=False, base_ch=16, depth=3, dtype="f32")
        model = S.SparK(enc, S.UnetDecoder(base_ch=16, depth=3, dtype="f32"), mask_ratio=0.6, densify_norm="", dtype="f32").to(dev).train()
        tr = P.SparKPretrainer(model, lr=1e-2)
        tr.broadcast_parameters()
        losses = []
        for it in range(steps):
            x = torch.randn(4, 1, 64, 64, generator=g)
            f = model.fmap_h
            active = torch.zeros(4, 1, f, f, dtype=torch.bool)
            for b in range(4):
                active[b, 0].view(-1)[torch.randperm(f * f, generator=g)[:model.len_keep]] = True
            losses.append(float(tr.step(x.to(dev), active.to(dev), loss_scale=float(opts.get("loss_scale", "1")))))
        res = {"losses": losses, "arena": tr.flat.arena.cpu(), "names": list(tr.flat.names), "exchange": getattr(tr, "last_exchange", None)}
    else:
        raise SystemExit(f"unknown mode {mode}")
    torch.save(res, f"{out}.{rank}")


if __name__ == "__main__":
    main()
