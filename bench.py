#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CM-UNet hot path on MI355X.

Workload (BASELINE.json configs[1]): reference UNet (base 64, depth 5, 31.04 M parameters)
as CM-UNet masked-reconstruction pretraining (rc_weight=1, ct_weight=0, mask_ratio 0.6, patch 16) on synthetic
512x512 grayscale batches, bs 32 per GPU.  One step = patch mask fused into the first conv + encoder +
pixel decoder forward + masked MSE + full backward + gradient all-reduce (N>1) + fused AdamW, all on the
hand-written HIP path.  Inputs (images and masks) are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Rank 0 prints ONE JSON line: metric images/sec (whole job), roofline of the dominant kernel (algorithmic FLOPs
of its launches / their HIP-event time, measured inside the timed region) and the CPU baseline (the oracle's
torch-CPU restatement of the same step on a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md


def cpu_baseline(H, W, seconds_budget=25.0):
    """Oracle (CPU restatement pinned to the reference, oracle/) timed on the host cores: same step, bs 2."""
    import torch
    from oracle import cmunet as OC, unet as OU
    from cmunet_amd.pretrain import create_random_patch_mask
    import numpy as np
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    bs = 2
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=0)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    decay = [v for k, v in params.items() if not (k.endswith(".bias") or v.dim() <= 1)]
    no_decay = [v for k, v in params.items() if k.endswith(".bias") or v.dim() <= 1]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}],
                            lr=1.5e-4, betas=(0.9, 0.95))
    x = torch.randn(bs, H, W, generator=torch.Generator().manual_seed(1234))
    mask = torch.from_numpy(create_random_patch_mask(bs, H, 16, 0.6, np.random.RandomState(0)))

    def step():
        opt.zero_grad()
        xm = x * (1 - mask[0]).float()
        logits = OU.unet_forward(xm, sd, training=True)
        loss = OC.masked_mse(logits[:, 1], x, mask)
        loss.backward()
        opt.step()

    step()                                   # warm-up
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < seconds_budget and n < 4):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(bs / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (torch CPU fp32 restatement of the reference step) bs={bs} {H}x{W}, {n} timed step(s) after 1 warm-up"}


def step_roofline(B, H, W, dtype, img_s):
    """Whole-step rooflines of SURVEY 8(d) for the reference UNet (base 64, depth 5): per layer F = algorithmic FLOPs,
    bytes = (in + out + weights) * sizeof(storage dtype), training convention x3 (forward + data gradient + weight
    gradient); composite = sum over layers of max(F / MFMA peak, bytes / HBM peak)."""
    es = 4 if dtype == "f32" else 2
    peak = PEAK_TFLOPS[dtype] * 1e12
    bw = 8.0e12
    layers = []   # (flops, bytes) per image, forward

    def conv(cin, cout, s, k):
        layers.append((2.0 * k * k * cin * cout * s * s, (cin * s * s + cout * s * s + k * k * cin * cout) * es))

    chans, s = [64, 128, 256, 512], H
    cin = 1
    for c in chans:
        conv(cin, c, s, 3); conv(c, c, s, 3)
        layers.append((3.0 * c * (s // 2) ** 2, (c * s * s + c * (s // 2) ** 2) * es))      # max-pool
        cin, s = c, s // 2
    conv(512, 1024, s, 3); conv(1024, 1024, s, 3)
    cin = 1024
    for c in reversed(chans):
        s *= 2
        layers.append((2.0 * cin * c * s * s, (cin * (s // 2) ** 2 + c * s * s + 4 * cin * c) * es))   # ConvTranspose 2x2 s2
        conv(2 * c, c, s, 3); conv(c, c, s, 3)
        cin = c
    conv(64, 2, s, 1)
    f = 3.0 * sum(l[0] for l in layers)
    by = 3.0 * sum(l[1] for l in layers)
    t_comp = 3.0 * sum(max(l[0] / peak, l[1] / bw) for l in layers)
    return {"gflop_per_image": round(f / 1e9, 1), "mb_per_image": round(by / 1e6, 1),
            "mfma_only_img_s": round(peak / f, 1), "hbm_only_img_s": round(bw / by, 1), "composite_img_s": round(1.0 / t_comp, 1),
            "frac_of_mfma_only": round(img_s * f / peak, 4), "frac_of_hbm_only": round(img_s * by / bw, 4),
            "frac_of_composite": round(img_s * t_comp, 4),
            "note": "SURVEY 8(d): algorithmic FLOPs / bytes per layer, x3 for training; peaks 2.5 PFLOP/s (16-bit) and 8 TB/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (pure throughput run)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cmunet_amd import _lib, model as M
    from cmunet_amd.pretrain import MaskedReconPretrainer, random_patch_mask_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): CMU_DIST_BACKEND=gloo CMU_SINGLE_DEVICE=1 run all ranks on cuda:0 over gloo
    backend = os.environ.get("CMU_DIST_BACKEND", "nccl")
    single_dev = os.environ.get("CMU_SINGLE_DEVICE", "0") == "1"
    dev_index = 0 if (world == 1 or single_dev) else local_rank
    if world > 1:
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))   # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    B, H, W = args.batch, args.size, args.size
    torch.manual_seed(0)
    net = M.UNet(out_classes=2, dtype=args.dtype).to(dev)           # reference structure, random init (same seed on all ranks)
    # lr rule of cmunet_config.py:70-73: base_lr * batch * gpus / 256
    tr = MaskedReconPretrainer(net, lr=1.5e-4 * B * world / 256.0, betas=(0.9, 0.95), weight_decay=0.05)
    tr.broadcast_parameters()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    nb = 2                                                          # distinct pre-staged synthetic batches
    imgs = [torch.randn(B, H, W, generator=g, device=dev) for _ in range(nb)]
    masks = [random_patch_mask_device(B, H, W, 16, 0.6, g, dev) for _ in range(nb)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        tr.step(imgs[i % nb], masks[i % nb])
    prof = None
    if not args.no_kernel_events:
        prof = _lib.EventProfiler()
        _lib.PROFILER = prof
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = tr.step(imgs[i % nb], masks[i % nb])
    sync()
    elapsed = time.perf_counter() - t0
    _lib.PROFILER = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())

    out = {
        "metric": "pretrain images/sec/node at 512x512 (CM-UNet masked-reconstruction step)",
        "value": round(B * world * args.steps / elapsed, 3),
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (seeded randn images, random 16x16 patch masks; random-init weights)",
        "config": {"workload": f"cmunet_masked_recon_unet64x5_{H}x{W}_bs{B}_mask0.6", "global_batch": B * world,
                   "image": [H, W], "parallelism": f"dp{world}", "optimizer": "AdamW(fused)", "loss": float(f"{loss_val:.6g}")},
    }
    if rank == 0 and prof is not None:
        summ = prof.summary()
        tot_ms = sum(v["ms"] for v in summ.values())
        # the MFMA entries grouped by the kernel that served them (rocprofv3 lists kernels, not entry points)
        mf = prof.by_kernel()
        name = max(mf, key=lambda k: mf[k]["ms"])
        d = mf[name]
        ach = d["work"] / (d["ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch of that kernel come from a separate rocprofv3 --pmc run of this same command
        # (tools/profile_round.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes), committed under profiles/
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f).get(name)
            if tj and args.dtype == "bf16" and B == 32 and H == 512:
                traffic = int(round(tj["hbm_bytes_per_launch"]))
        except (OSError, ValueError, KeyError):
            traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "entries": sorted(d["entries"]), "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json)",
                           "algorithmic_gflop_per_launch": round(d["work"] / d["calls"] / 1e9, 1),
                           "launches_per_step": d["calls"] // args.steps,
                           "avg_launch_ms": round(d["ms"] / d["calls"], 4),
                           "share_of_kernel_time": round(d["ms"] / tot_ms, 3)}
        out["mfma_kernels"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["calls"] // args.steps,
                                   "tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in mf.items()}
        out["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        flops_step = sum(v["work"] for v in summ.values()) / args.steps
        out["step_mfma_tflops"] = round(flops_step / (elapsed / args.steps) / 1e12, 2)
        out["step_roofline"] = step_roofline(B, H, W, args.dtype, out["value"] / world)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(H, W)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
