#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CM-UNet hot path on MI355X.

Default workload (BASELINE.json configs[1]): reference UNet (base 64, depth 5, 31.04 M parameters)
as CM-UNet masked-reconstruction pretraining (rc_weight=1, ct_weight=0, mask_ratio 0.6, patch 16) on synthetic
512x512 grayscale batches, bs 32 per GPU, in the reference's arithmetic for this configuration: fp16 operands with
fp32 accumulation and dynamic loss scaling (AmpOptimWrapper, cmunet_config.py:76-78).  One step = patch mask fused into
the first conv + encoder + pixel decoder forward + masked MSE + full backward + gradient all-reduce (N>1) + inf/nan
check + fused AdamW + loss-scale update, all on the hand-written HIP path.  Inputs (images and masks) are resident in
HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--workload recon|moco|joint|spark] [--dtype f16|bf16|f32]

N > 1: one rank per GPU over RCCL.  Either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or directly -- then this process only spawns its N ranks (it never touches the GPU itself) and exits with
the first non-zero child status.

Other workloads (same bs 32 / GPU, 512x512; BASELINE configs 3-5):
    moco   Moco_v2 (K = 4096, tau = 0.2, emb 1024, m = 0.999), key all-gather, fused InfoNCE + enqueue, SGD-momentum
    joint  CM_UNet contrastive + masked reconstruction (projector in = 262,144), AdamW, EMA of the target networks
    spark  SparK sparse masked-conv encoder (mask 0.75: 256 of 1,024 patches active) + UNet decoder, LAMB

Rank 0 prints ONE JSON line: metric images/sec (whole job), roofline of the dominant kernel (algorithmic FLOPs of its
launches / their HIP-event time, measured inside the timed region) and the CPU baseline (the oracle's torch-CPU
restatement of the same step on a bounded sample, N=1 only).  By default the HIP events bracket the GEMM-shaped entry
points only (~80 launches of a step: what the roofline needs); --all-kernel-events brackets every entry point and prints
the full kernel_ms_per_step table at ~0.5 ms per step (1.2 % of it).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


# ---------------------------------------------------------------------------------------------------------------------
# host description for the CPU baseline
# ---------------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    """(model string, physical cores of the machine, CPUs this process may use)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    try:                                           # cgroup v2 CPU quota of the container
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            avail = max(1, min(avail, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    physical = len(cores) if cores else (os.cpu_count() or 1)
    return model, physical, avail


def cpu_baseline(workload, H, W, seconds_budget=25.0):
    """Oracle (CPU restatement pinned to the reference, oracle/) timed on the host cores: same step, bs 2."""
    import numpy as np
    import torch
    from oracle import cmunet as OC, unet as OU
    from cmunet_amd.pretrain import create_random_patch_mask
    model, physical, avail = host_cpu_info()
    threads = max(1, min(physical, avail))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    bs = 2
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(bs, H, W, generator=g)

    def grouped_adamw(params, lr=1.5e-4):
        decay = [v for k, v in params.items() if not (k.endswith(".bias") or v.dim() <= 1)]
        no_decay = [v for k, v in params.items() if k.endswith(".bias") or v.dim() <= 1]
        return torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}], lr=lr, betas=(0.9, 0.95))

    if workload == "recon":
        sd = OU.make_state_dict(base_ch=64, depth=5, seed=0)
        params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
        opt = grouped_adamw(params)
        mask = torch.from_numpy(create_random_patch_mask(bs, H, 16, 0.6, np.random.RandomState(0)))

        def step():
            opt.zero_grad()
            logits = OU.unet_forward(x * (1 - mask[0]).float(), sd, training=True)
            OC.masked_mse(logits[:, 1], x, mask).backward()
            opt.step()
        what = "masked-reconstruction step"
    elif workload == "moco":
        from oracle import moco as OM
        enc = {k: v for k, v in OU.make_state_dict(base_ch=64, depth=5, seed=0).items() if k.startswith(("down_conv", "double_conv"))}
        sd = {"encoder_q." + k: v.clone() for k, v in enc.items()}
        sd.update({"encoder_k." + k: v.clone() for k, v in enc.items()})
        params = [v.requires_grad_(True) for k, v in sd.items() if k.startswith("encoder_q.") and v.is_floating_point() and "running" not in k]
        opt = torch.optim.SGD(params, lr=0.03, momentum=0.9, weight_decay=1e-4)
        queue, ptr = OM.init_queue(1024, 4096), torch.zeros(1, dtype=torch.long)
        xk = torch.randn(bs, H, W, generator=g)

        def step():
            opt.zero_grad()
            loss, _, _ = OM.training_step(x.unsqueeze(1), xk.unsqueeze(1), sd, queue, ptr, 0.2, 0.999)
            loss.backward()
            opt.step()
        what = "MoCo-v2 step (K 4096)"
    elif workload == "spark":
        from oracle import spark as OS
        full = OU.make_state_dict(base_ch=64, depth=5, seed=0)
        sd = {}
        for k, v in full.items():
            if "up_conv" in k or "conv_last" in k:
                sd["dense_decoder." + k] = v[:1].clone() if "conv_last" in k else v
            else:
                sd["sparse_encoder.sp_cnn." + k] = v
        toks = [torch.zeros(1, c, 1, 1).normal_(0, 0.02, generator=g).requires_grad_(True) for c in (1024, 512, 256, 128, 64)]
        params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k] + toks
        opt = torch.optim.AdamW(params, lr=2e-4, betas=(0.9, 0.95), weight_decay=0.04)      # (stand-in for LAMB: same passes)
        active = OS.make_active(bs, H // 16, 0.75, g)

        def step():
            opt.zero_grad()
            loss, _ = OS.forward(x.unsqueeze(1), active, sd, toks)
            loss.backward()
            opt.step()
        what = "SparK step (mask 0.75; AdamW standing in for LAMB)"
    elif workload == "joint":
        # the joint step in the REFERENCE's own geometry (cmunet_config.py:5-42: 224 x 224, projector 50,176 -> 1,536: 217.8 M
        # parameters) -- at 512 x 512 the 403 M-parameter projector step takes minutes per image on the host
        H = W = 224
        x, xt = torch.randn(bs, H, W, generator=g), torch.randn(bs, H, W, generator=g)
        sd = OC.make_cmunet_sd(0, img_size=H)
        params = {k: v.requires_grad_(True) for k, v in sd.items()
                  if v.is_floating_point() and "running" not in k and not k.startswith(("target_backbone.", "target_projector."))}
        opt = grouped_adamw(params)
        mask = create_random_patch_mask(bs, H, 16, 0.6, np.random.RandomState(0))
        rw, rb = torch.randn(256, 1024, 1, 1, generator=g) * 0.03, torch.zeros(256)

        def step():
            opt.zero_grad()
            l = OC.forward_train(x, xt, mask, rw, rb, sd)
            (l["loss_ct"] + l["loss_rc"]).backward()
            opt.step()
            with torch.no_grad():
                OC.momentum_update(sd, 0.996)
        what = "joint CM-UNet step at the reference's own 224x224 geometry (projector 50,176 x 1,536; the bench runs 512x512)"
    else:
        return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port", "sample": "not timed for this workload"}

    step()                                   # warm-up
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < seconds_budget and n < 4):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(bs / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port", "cpu_model": model,
            "physical_cores": physical, "cpus_available": avail,
            "sample": f"oracle (torch CPU fp32 restatement of the reference's {what}) bs={bs} {H}x{W}, {n} timed step(s) after 1 warm-up, "
                      f"torch.set_num_threads({threads})"}


# ---------------------------------------------------------------------------------------------------------------------
# whole-step rooflines
# ---------------------------------------------------------------------------------------------------------------------
def unet_layers(H, es):
    """(flops, bytes) per image and forward pass of every layer of the reference UNet, split into
    encoder and decoder lists."""
    enc, dec = [], []

    def conv(lst, cin, cout, s, k):
        lst.append((2.0 * k * k * cin * cout * s * s, (cin * s * s + cout * s * s + k * k * cin * cout) * es))

    chans, s = [64, 128, 256, 512], H
    cin = 1
    for c in chans:
        conv(enc, cin, c, s, 3); conv(enc, c, c, s, 3)
        enc.append((3.0 * c * (s // 2) ** 2, (c * s * s + c * (s // 2) ** 2) * es))      # max-pool
        cin, s = c, s // 2
    conv(enc, 512, 1024, s, 3); conv(enc, 1024, 1024, s, 3)
    cin = 1024
    for c in reversed(chans):
        s *= 2
        dec.append((2.0 * cin * c * s * s, (cin * (s // 2) ** 2 + c * s * s + 4 * cin * c) * es))   # ConvTranspose 2x2 s2
        conv(dec, 2 * c, c, s, 3); conv(dec, c, c, s, 3)
        cin = c
    conv(dec, 64, 2, s, 1)
    return enc, dec


def step_roofline(workload, H, dtype, img_s):
    """Whole-step rooflines of SURVEY 8(d): per layer F = algorithmic FLOPs, bytes = (in + out + weights) * sizeof(storage
    dtype); a trained pass counts x3 (forward + data gradient + weight gradient), a forward-only pass x1; composite = sum over
    layers of max(F / MFMA peak, bytes / HBM peak).  recon: encoder + decoder trained.  moco: query encoder trained + key
    encoder forward.  joint: online encoder + two decoders trained, target encoder forward, projector / target projector /
    predictor as 3 + 1 passes over 2*H*W*1536 FLOP and H*W*1536 fp32 weights.  spark: as recon (dense-equivalent work; the
    sparse encoder's useful share is its active fraction)."""
    es = 4 if dtype == "f32" else 2
    peak = PEAK_TFLOPS[dtype] * 1e12
    enc, dec = unet_layers(H, es)
    parts = {"recon": [(enc, 3), (dec, 3)], "spark": [(enc, 3), (dec, 3)], "moco": [(enc, 3), (enc, 1)],
             "joint": [(enc, 3), (dec, 3), (dec, 3), (enc, 1)]}[workload]
    f = sum(m * sum(l[0] for l in lst) for lst, m in parts)
    by = sum(m * sum(l[1] for l in lst) for lst, m in parts)
    t_comp = sum(m * sum(max(l[0] / peak, l[1] / HBM_PEAK) for l in lst) for lst, m in parts)
    if workload == "joint":        # projector (trained: 3 weight passes) + target projector (1 pass), fp32 weights, per BATCH of 32
        wbytes = H * H * 1536 * 4.0
        t_comp += 4 * wbytes / HBM_PEAK / 32.0
        by += 4 * wbytes / 32.0
        f += 4 * 2.0 * H * H * 1536
    return {"gflop_per_image": round(f / 1e9, 1), "mb_per_image": round(by / 1e6, 1),
            "mfma_only_img_s": round(peak / f, 1), "hbm_only_img_s": round(HBM_PEAK / by, 1), "composite_img_s": round(1.0 / t_comp, 1),
            "frac_of_mfma_only": round(img_s * f / peak, 4), "frac_of_hbm_only": round(img_s * by / HBM_PEAK, 4),
            "frac_of_composite": round(img_s * t_comp, 4),
            "note": "SURVEY 8(d): algorithmic FLOPs / bytes per layer (x3 for a trained pass); peaks 2.5 PFLOP/s (16-bit) and 8 TB/s"}


# ---------------------------------------------------------------------------------------------------------------------
# self-launch (python bench.py --gpus N without a launcher)
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start one child per GPU with the torch.distributed environment and wait.  This parent never initialises the GPU (no
    torch import, no HIP call) and never replaces itself with another program; a failing rank ends the others."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rk in range(n):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for rk in sorted(pending):
                r = procs[rk].poll()
                if r is None:
                    continue
                pending.discard(rk)
                if r != 0 and rc == 0:
                    rc = r
                    for other in pending:          # a rank failed: its peers would wait in a collective for ever
                        procs[other].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# workloads: each returns (step callable -> loss tensor, description dict)
# ---------------------------------------------------------------------------------------------------------------------
def build_workload(name, args, dev, rank, world):
    import torch
    from cmunet_amd import model as M
    from cmunet_amd import pretrain as P
    B, H, W = args.batch, args.size, args.size
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    nb = 2                                                          # distinct pre-staged synthetic batches
    torch.manual_seed(0)                                            # same initial weights on every rank (+ broadcast below)
    lr_rule = B * world / 256.0                                     # cmunet_config.py:70-73 / arg_util.py:133: base_lr * batch * gpus / 256
    if name == "recon":
        net = M.UNet(out_classes=2, dtype=args.dtype).to(dev)
        tr = P.MaskedReconPretrainer(net, lr=1.5e-4 * lr_rule, betas=(0.9, 0.95), weight_decay=0.05, amp=(args.dtype == "f16"))
        tr.broadcast_parameters()
        imgs = [torch.randn(B, H, W, generator=g, device=dev) for _ in range(nb)]
        masks = [P.random_patch_mask_device(B, H, W, 16, 0.6, g, dev) for _ in range(nb)]
        return (lambda i: tr.step(imgs[i % nb], masks[i % nb])), {
            "workload": f"cmunet_masked_recon_unet64x5_{H}x{W}_bs{B}_mask0.6", "optimizer": "AdamW(fused)",
            "amp": "dynamic loss scale (device-side GradScaler protocol)" if tr.amp is not None else "off"}, tr
    if name == "moco":
        from cmunet_amd import moco as MO
        m = MO.Moco_v2(emb_dim=1024, num_negatives=4096, softmax_temperature=0.2, encoder_momentum=0.999, dtype=args.dtype).to(dev)
        tr = P.MocoPretrainer(m, lr=0.03 * lr_rule)
        tr.broadcast_parameters()
        xs = [(torch.randn(B, 1, H, W, generator=g, device=dev), torch.randn(B, 1, H, W, generator=g, device=dev)) for _ in range(nb)]
        ls = 1024.0 if args.dtype == "f16" else 1.0     # static loss scale (f16 stores the activation gradients)
        return (lambda i: tr.step(*xs[i % nb], loss_scale=ls)), {
            "workload": f"moco_v2_unet64x5_encoder_{H}x{W}_bs{B}_K4096_tau0.2", "optimizer": "SGD-momentum(fused)", "loss_scale": ls}, tr
    if name == "joint":
        from cmunet_amd import cmunet as C
        m = C.build_model(C.cmunet_config(img_size=H, dtype=args.dtype, mask_ratio=0.6)).to(dev)
        m.init_weights()
        tr = P.JointPretrainer(m, lr=1.5e-4 * lr_rule, amp=(args.dtype == "f16"))
        tr.broadcast_parameters()
        xs = [(torch.randn(B, H, W, generator=g, device=dev), torch.randn(B, H, W, generator=g, device=dev)) for _ in range(nb)]
        masks = [P.random_patch_mask_device(B, H, W, 16, 0.6, g, dev) for _ in range(nb)]

        def step(i):
            l = tr.step(xs[i % nb][0], xs[i % nb][1], masks[i % nb])
            return l["loss_ct"] + l["loss_rc"]
        return step, {"workload": f"cmunet_joint_ct+rc_unet64x5_{H}x{W}_bs{B}_mask0.6_projector{H * W}x1536", "optimizer": "AdamW(fused) + EMA",
                      "amp": "dynamic loss scale (device-side GradScaler protocol)" if tr.amp is not None else "off"}, tr
    if name == "spark":
        from cmunet_amd import spark as S
        enc = S.build_sparse_encoder("unet_sparse", input_size=H, dtype=args.dtype)
        m = S.SparK(enc, S.UnetDecoder(dtype=args.dtype), mask_ratio=0.75, densify_norm="", dtype=args.dtype).to(dev)
        tr = P.SparKPretrainer(m, lr=2e-4 * lr_rule)
        tr.broadcast_parameters()
        xs = [torch.randn(B, 1, H, W, generator=g, device=dev) for _ in range(nb)]
        acts = [m.mask(B, dev, torch.Generator().manual_seed(77 + rank + 1000 * i)) for i in range(nb)]
        ls = 1024.0 if args.dtype == "f16" else 1.0     # static loss scale: the patch-normalised loss has O(1) gradients per patch
        return (lambda i: tr.step(xs[i % nb], acts[i % nb], loss_scale=ls)), {
            "workload": f"spark_sparse_unet64x5_{H}x{W}_bs{B}_mask0.75_keep{m.len_keep}of{m.fmap_h * m.fmap_w}", "optimizer": "LAMB(fused)"}, tr
    raise SystemExit(f"unknown workload {name}")


def rank_plan(args, env):
    """What this rank will do, derived from the launcher's environment alone (no GPU, no process group): the torch.distributed
    coordinates, the device it binds, and the workload's data-parallel invariants -- per-GPU batch fixed (weak scaling), the
    reference's learning-rate rule lr * batch * gpus / 256 (cmunet_config.py:70-73, arg_util.py:133), and for MoCo the queue length a
    multiple of the gathered key batch (moco2_module.py:169: ``assert self.hparams.num_negatives % batch_size == 0`` on B * world keys)."""
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local_rank = int(env.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): CMU_DIST_BACKEND=gloo CMU_SINGLE_DEVICE=1 run all ranks on cuda:0 over gloo
    backend = env.get("CMU_DIST_BACKEND", "nccl")
    single_dev = env.get("CMU_SINGLE_DEVICE", "0") == "1"
    dev_index = 0 if (world == 1 or single_dev) else local_rank
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not (0 <= rank < world and 0 <= local_rank < world):
        raise SystemExit(f"RANK={rank} / LOCAL_RANK={local_rank} outside WORLD_SIZE={world}")
    plan = {"world": world, "rank": rank, "local_rank": local_rank, "backend": backend, "dev_index": dev_index,
            "master": f"{env.get('MASTER_ADDR', '')}:{env.get('MASTER_PORT', '')}", "workload": args.workload,
            "batch_per_gpu": args.batch, "global_batch": args.batch * world, "lr_rule": args.batch * world / 256.0,
            "data_seed": 1234 + rank, "ipc_mode_legacy": env.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    if args.workload == "moco":
        K = 4096
        plan["queue"] = {"K": K, "gathered_keys": args.batch * world, "divides": K % (args.batch * world) == 0}
        if not plan["queue"]["divides"]:
            raise SystemExit(f"MoCo queue of {K} is not a multiple of the gathered key batch {args.batch} x {world} (moco2_module.py:169)")
    return plan


def dry_run_env(plan, dist, torch):
    """--dry-run-env: the ranks meet in a gloo group on the CPU (the launcher's own MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE),
    prove the rendezvous with an all-reduce of ones, gather their plans on rank 0 and check them against each other: one device per
    rank (LOCAL_RANK -> device index, all distinct), one rendezvous address, one workload and batch, distinct data seeds.  Returns
    the exit status; rank 0 prints one JSON line."""
    world, rank = plan["world"], plan["rank"]
    if world > 1 or "MASTER_PORT" in os.environ:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        ones = torch.ones(1)
        dist.all_reduce(ones)
        plans = [None] * world
        dist.all_gather_object(plans, plan)
        seen = int(ones.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        plans, seen = [plan], 1
    if rank != 0:
        return 0
    ok = (seen == world and sorted(p["rank"] for p in plans) == list(range(world))
          and len({p["master"] for p in plans}) == 1 and len({(p["workload"], p["batch_per_gpu"], p["global_batch"], p["lr_rule"]) for p in plans}) == 1
          and len({p["data_seed"] for p in plans}) == world
          and (world == 1 or os.environ.get("CMU_SINGLE_DEVICE", "0") == "1" or sorted(p["dev_index"] for p in plans) == list(range(world))))
    print(json.dumps({"dry_run_env": True, "ok": bool(ok), "ranks_seen": seen, "plans": plans}), flush=True)
    return 0 if ok else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "f32"],
                    help="activation / MFMA operand type (f16 = the reference's AMP arithmetic, cmunet_config.py:76-78)")
    ap.add_argument("--workload", default="recon", choices=["recon", "moco", "joint", "spark"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (pure throughput run)")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="HIP events around EVERY entry point (the full kernel_ms_per_step table: ~460 events per step cost ~0.5 ms of "
                         "the 41 ms) instead of the GEMM-shaped entries only (the default: what the roofline needs, ~0.15 ms)")
    ap.add_argument("--gemm-events-only", action="store_true", help="(default behaviour; kept for scripts)")
    ap.add_argument("--graph", action="store_true", help="capture the step of each pre-staged batch in a hipGraph and replay it "
                    "(experiment: removes the host's ~230 launches per step; implies --no-kernel-events)")
    ap.add_argument("--dry-run-env", action="store_true",
                    help="rehearse the launch up to the first GPU call: every rank derives its plan from the environment (device index, "
                         "batch, learning-rate rule, queue divisibility), the ranks meet in a CPU (gloo) group, compare their plans and "
                         "rank 0 prints them as one JSON line -- no GPU is touched (tests/test_cpu_distributed.py runs --gpus 8 this way)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    plan = rank_plan(args, os.environ)
    world, rank, local_rank, backend, dev_index = plan["world"], plan["rank"], plan["local_rank"], plan["backend"], plan["dev_index"]
    if args.dry_run_env:
        sys.exit(dry_run_env(plan, dist, torch))
    from cmunet_amd import _lib
    # An older library under a newer host package (CMU_LIB_PATH in an A/B script) lacks entry points: say so in ONE line instead of a
    # traceback from the middle of a step -- round 4's g21 A/B printed nothing for exactly that variant (the round-3 library under the
    # round-4 package, `2>&1 | tail -1` in front of a JSON parser; profiles/HISTORY.md section E)
    missing = _lib.missing_symbols()
    if missing and os.environ.get("CMU_LIB_PATH"):
        # (a warning, not an exit: an A/B against last round's library is legitimate as long as the workload does not reach the new entries;
        # a call that does raises CmuError naming the symbol, and tools/ab_bench.sh shows the last stderr line of a run without a JSON line)
        print(f"bench.py: warning: {_lib.LIB_PATH} does not export {len(missing)} entry point(s) of this package (first: {missing[0]}): "
              "an older build under CMU_LIB_PATH -- a workload that needs them will stop with CmuError", file=sys.stderr, flush=True)
    # CMU_DP_REHEARSE=1 under a launcher with one rank: the RCCL group is built and every collective of the step runs on it
    use_dist = world > 1 or (os.environ.get("CMU_DP_REHEARSE", "0") == "1" and "WORLD_SIZE" in os.environ)
    if use_dist:
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))   # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    B, H, W = args.batch, args.size, args.size
    step, desc, tr = build_workload(args.workload, args, dev, rank, world)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if use_dist and tr is not None:
        tr.time_exchange = True          # the trainers then bracket their exchange waits with a host clock and two stream events
    for i in range(args.warmup):
        step(i)
    if args.graph:
        # one graph per pre-staged batch (the step reads its batch by index); the loss tensor of the trainer is static
        args.no_kernel_events = True
        torch.cuda.synchronize()
        graphs = []
        side = torch.cuda.Stream()
        for b in range(2):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                step(b)                                   # warm the allocator's pool on the capture stream
                torch.cuda.synchronize()
                with torch.cuda.graph(gph, stream=side):
                    out_b = step(b)
            graphs.append((gph, out_b))
        torch.cuda.synchronize()
        eager_step = step

        def step(i):                                      # noqa: F811
            gph, out_b = graphs[i % 2]
            gph.replay()
            return out_b
    prof = None
    if not args.no_kernel_events:
        prof = _lib.EventProfiler(gemm_only=not args.all_kernel_events)
        _lib.PROFILER = prof
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    sync()
    elapsed = time.perf_counter() - t0
    _lib.PROFILER = None
    rccl = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # proof that the collective library saw N ranks (not just WORLD_SIZE in the environment): an all-reduce of ones, the group's
        # own world size and backend, and what the last step's gradient exchange looked like (buckets started inside the backward)
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        rccl = {"world": dist.get_world_size(), "ranks_seen": int(round(float(ones.item()))), "backend": str(dist.get_backend()),
                # (exchange_report(): + wait_ms = host clock around the waits of the LAST timed step, exposed_ms = how long the compute
                # stream stood still for the collectives -- events on that stream around the waits; time, not launch-order bookkeeping)
                "exchange": tr.exchange_report() if hasattr(tr, "exchange_report") else getattr(tr, "last_exchange", None)}
        if rccl["world"] != world or rccl["ranks_seen"] != world:
            raise SystemExit(f"process group saw {rccl['ranks_seen']} of {rccl['world']} ranks, WORLD_SIZE={world}")
    loss_val = float(loss.reshape(-1)[0].item())

    metric = {"recon": "pretrain images/sec/node at 512x512 (CM-UNet masked-reconstruction step)",
              "moco": "pretrain images/sec/node at 512x512 (MoCo-v2 step on the UNet encoder)",
              "joint": "pretrain images/sec/node at 512x512 (CM-UNet joint contrastive + masked-reconstruction step)",
              "spark": "pretrain images/sec/node at 512x512 (SparK sparse masked-conv step)"}[args.workload]
    cfg = dict(desc)
    cfg.update({"global_batch": B * world, "image": [H, W], "parallelism": f"dp{world}", "loss": float(f"{loss_val:.6g}")})
    amp = getattr(tr, "amp", None)
    if amp is not None and rank == 0:
        sc, _, _, good, skipped = amp.read()
        cfg["amp_state"] = {"scale": sc, "updates": good, "skipped": skipped}
    out = {
        "metric": metric,
        "value": round(B * world * args.steps / elapsed, 3),
        "unit": "images/sec",
        "n_gpus": rccl["world"] if rccl is not None else world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (seeded randn images, random 16x16 patch masks; random-init weights)",
        "config": cfg,
    }
    if rccl is not None:
        out["rccl"] = rccl
    if rank == 0 and prof is not None:
        summ = prof.summary()
        tot_ms = sum(v["ms"] for v in summ.values())
        # the MFMA entries grouped by the kernel that served them (rocprofv3 lists kernels, not entry points)
        mf = prof.by_kernel()
        # The dominant kernel is the persistent 3x3 implicit-GEMM conv kernel in its two instruction shapes: conv_igemm5_kernel (16x16x32
        # MFMA, round 5: the whole-tile 128-channel layers) and conv_igemm3p_kernel (32x32x16: everything else).  They serve the SAME 34
        # launches per step as before round 5 (forward + data gradient of the 17 multi-channel 3x3 convs); taken together, so that
        # `frac` stays comparable across rounds -- the faster shape alone would flatter it (0.55 against 0.50 over all 34).
        FAMILY = ("conv_igemm5_kernel", "conv_igemm3p_kernel")
        fam = [k for k in FAMILY if k in mf]
        if len(fam) == 2:
            name = "+".join(fam)
            mf_all = dict(mf)
            d = {"ms": sum(mf[k]["ms"] for k in fam), "calls": sum(mf[k]["calls"] for k in fam), "work": sum(mf[k]["work"] for k in fam),
                 "entries": set().union(*(mf[k]["entries"] for k in fam))}
        else:
            name = max(mf, key=lambda k: mf[k]["ms"])
            d = mf[name]
        ach = d["work"] / (d["ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch of that kernel come from a separate rocprofv3 --pmc run of this same command
        # (tools/profile_round.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes), committed under profiles/
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)
            ent = tj.get(name)
            if ent is None and "+" in name:      # two kernels: launch-weighted mean of their per-launch traffic
                parts = [(tj.get(k), mf[k]["calls"]) for k in name.split("+")]
                if all(e for e, _ in parts):
                    ent = {"hbm_bytes_per_launch": sum(e["hbm_bytes_per_launch"] * c for e, c in parts) / sum(c for _, c in parts)}
            if ent and tj.get("_command", {}).get("dtype", "bf16") == args.dtype and args.workload == "recon" and B == 32 and H == 512:
                traffic = int(round(ent["hbm_bytes_per_launch"]))
        except (OSError, ValueError, KeyError, AttributeError):
            traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "entries": sorted(d["entries"]), "frac": round(ach / peak, 4), "traffic": traffic,
                           "traffic_unit": "HBM bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE from a separate rocprofv3 --pmc run, profiles/traffic.json)",
                           "algorithmic_gflop_per_launch": round(d["work"] / d["calls"] / 1e9, 1),
                           "launches_per_step": d["calls"] // args.steps,
                           "avg_launch_ms": round(d["ms"] / d["calls"], 4),
                           "share_of_timed_entries": round(d["ms"] / tot_ms, 3)}
        if args.dtype in ("f16", "bf16"):
            # what the matrix pipes of THIS chip sustain: back-to-back MFMAs from registers on every SIMD (csrc/probe.hip).  The
            # power management lowers the shader clock under that load by an amount that depends on the operand data, so the
            # nominal 2.5 PFLOP/s (2.4 GHz) is out of reach of any kernel fed with dense data; measured here, after the timed region
            try:
                from cmunet_amd import ops as _ops
                sus = {}
                for fed, fl in ((False, "registers"), (True, "lds_fed")):
                    for pat, label in ((0, "dense_normal_operands"), (1, "relu_operands_half_zero"), (2, "zero_operands")):
                        tf, clk = _ops.mfma_sustained_rate(args.dtype, pat, lds_fed=fed, device=dev)
                        sus[f"{fl}.{label}"] = {"tflops": round(tf, 1), "clock_mhz": round(clk)}
                ref = sus["lds_fed.dense_normal_operands"]["tflops"]
                out["roofline"]["sustained"] = {
                    "measured": sus, "frac_of_sustained_lds_fed_dense": round(ach / ref, 4),
                    "note": "32x32x16 MFMA loops on all SIMDs, operands held in registers or read from LDS at the conv kernel's fragment ratio "
                            "(cmu_mfma_sustained_rate, csrc/probe.hip): the shader clock under a matrix load depends on the operand data; "
                            "`peak` stays the nominal dense figure at 2.4 GHz and `frac` = achieved / peak"}
            except Exception as e:      # a diagnostic: never takes the line down
                out["roofline"]["sustained"] = {"error": repr(e)}
        out["mfma_kernels"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["calls"] // args.steps,
                                   "tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in mf.items()}
        out["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_ms_scope"] = ("every entry point (HIP events around all ~230 launches of a step: ~0.5 ms of it)" if args.all_kernel_events
                                  else "GEMM-shaped entries only (default; --all-kernel-events for every entry point)")
        flops_step = sum(v["work"] for v in summ.values()) / args.steps
        out["step_mfma_tflops"] = round(flops_step / (elapsed / args.steps) / 1e12, 2)
        if H == W:
            out["step_roofline"] = step_roofline(args.workload, H, args.dtype, out["value"] / world)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.workload, H, W)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
