"""World-size-2 gloo tests (CPU) of the data-parallel logic around the hot path: embedding
all-gather with rank-offset labels, gathered-key queue update staying replicated, gradient mean via one
all-reduce with the 1/world scale folded into the optimiser.  The arithmetic checker is the oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cmunet_amd.cmunet import concat_all_gather
        from cmunet_amd.optim import all_reduce_sum_scale
        from oracle import cmunet as OC, moco as OM
        B, D = 4, 16
        g = torch.Generator().manual_seed(0)                      # same data on every rank, sliced by rank
        pred_all = torch.randn(world * B, D, generator=g)
        proj_all = F.normalize(torch.randn(world * B, D, generator=g), dim=1)
        pred, proj = pred_all[rank * B:(rank + 1) * B], proj_all[rank * B:(rank + 1) * B]
        keys = concat_all_gather(proj)                             # cmunet_head.py:77
        assert torch.equal(keys, proj_all)
        loss_rank = OC.infonce_inbatch(pred, keys, 0.07, rank=rank)            # labels i + B*rank
        t = loss_rank.detach().clone()
        dist.all_reduce(t)
        single = OC.infonce_inbatch(pred_all, proj_all, 0.07, rank=0)          # one process, concatenated batch
        assert abs(float(t) / world - float(single)) < 1e-5
        # MoCo: every rank enqueues the SAME gathered keys -> queue and pointer stay replicated
        queue, ptr = OM.init_queue(D, 32, seed=1), torch.zeros(1, dtype=torch.long)
        OM.dequeue_and_enqueue(concat_all_gather(proj), queue, ptr, 32)
        ref = queue.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, queue) and int(ptr) == world * B
        # gradient mean: sum-all-reduce + 1/world scale == gradient of the big batch
        w = torch.randn(D, 3, generator=g, requires_grad=True)
        x_all, y_all = torch.randn(world * B, D, generator=g), torch.randn(world * B, 3, generator=g)
        loss = ((x_all[rank * B:(rank + 1) * B] @ w - y_all[rank * B:(rank + 1) * B]) ** 2).mean()
        loss.backward()
        grad = w.grad.clone()
        scale = all_reduce_sum_scale(grad)
        w2 = w.detach().clone().requires_grad_(True)
        ((x_all @ w2 - y_all) ** 2).mean().backward()
        assert torch.allclose(grad * scale, w2.grad, atol=1e-6)
        # two-bucket exchange (decoder tail first, asynchronously) == one all-reduce of the whole arena
        from cmunet_amd.optim import FlatParams
        flat = FlatParams.__new__(FlatParams)      # the arena itself is GPU-only: exercise the bucket logic on a CPU stand-in
        flat.names = ["down_conv1.w", "down_conv1.b", "double_conv.w", "up_conv1.w", "up_conv1.b", "conv_last.w"]
        sizes, flat.offsets, o = [36, 8, 24, 12, 4, 8], {}, 0
        for n, sz in zip(flat.names, sizes):
            flat.offsets[n] = (o, sz)
            o += sz
        flat.grad = torch.zeros(o)
        off = flat.tail_offset(("up_conv", "conv_last"))
        assert off == 36 + 8 + 24 and flat.tail_offset(("double_conv",)) is None and flat.tail_offset(("nope",)) is None
        flat.grad.copy_(torch.arange(flat.grad.numel(), dtype=torch.float32) * (rank + 1))
        whole = flat.grad.clone()
        dist.all_reduce(whole)
        h1 = flat.all_reduce_range_async(off, flat.grad.numel())
        h0 = flat.all_reduce_range_async(0, off)
        h1.wait(); h0.wait()
        assert torch.equal(flat.grad, whole)
        # round 6 -- sharded exchange of one big parameter (pretrain.reduce_scatter_sum_async / all_gather_shares_async: the joint step's
        # projector.fc0.weight, cmunet_config.py:18-26): reduce-scatter of the gradient, an element-wise update on the own share, all-gather
        # of the updated parameter == all-reduce + the same update everywhere, bit for bit on two ranks (a + b commutes)
        from cmunet_amd.pretrain import all_gather_shares_async, reduce_scatter_sum_async, shard_bounds
        n = 4096
        gsh = torch.Generator().manual_seed(10 + rank)
        grad_local = torch.randn(n, generator=gsh)
        p0 = torch.randn(n, generator=torch.Generator().manual_seed(3))

        def update(p, gsum):                        # stand-in for the optimiser kernel: element-wise, the same on every path
            return p - 0.1 * (gsum * (1.0 / world)) - 0.01 * p
        ga = grad_local.clone()
        dist.all_reduce(ga)
        pa = update(p0.clone(), ga)
        gb, pb = grad_local.clone(), p0.clone()
        reduce_scatter_sum_async(gb).wait()
        lo, hi = shard_bounds(n, world, rank)
        assert (lo, hi) == (rank * n // world, (rank + 1) * n // world)
        assert torch.equal(gb[lo:hi], ga[lo:hi])    # the own share holds the SUM; the rest of gb is unspecified
        pb[lo:hi] = update(pb[lo:hi], gb[lo:hi])
        all_gather_shares_async(pb).wait()
        assert torch.equal(pb, pa)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _one_rank_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("CMU_DP_REHEARSE", None)
    try:
        from cmunet_amd.cmunet import concat_all_gather
        from cmunet_amd.optim import all_reduce_sum_scale, dp_exchanges, dp_world
        assert dp_world() == 1 and not dp_exchanges()               # no group at all
        dist.init_process_group("gloo", rank=0, world_size=1)
        t = torch.arange(6.0)
        assert dp_world() == 1 and not dp_exchanges()               # a one-rank group exchanges nothing ...
        assert all_reduce_sum_scale(t.clone()) == 1.0 and concat_all_gather(t) is t
        os.environ["CMU_DP_REHEARSE"] = "1"                         # ... unless the rehearsal knob asks for the call sequence
        assert dp_exchanges() and dp_world() == 1
        u = t.clone()
        assert all_reduce_sum_scale(u) == 1.0 and torch.equal(u, t)         # SUM over one rank, scale 1 / 1
        g = concat_all_gather(t.view(2, 3))
        assert g is not t and torch.equal(g, t.view(2, 3))
        q.put("ok")
    except Exception as e:  # noqa: BLE001
        q.put(repr(e))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_one_rank_group_and_the_rehearsal_knob():
    """optim.dp_world / dp_exchanges: the single gate of every collective of the trainers.  One rank exchanges nothing; with
    CMU_DP_REHEARSE=1 the collectives run on the one-rank group (how tests/test_gpu_dataparallel.py drives RCCL on a one-GPU box)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=180)
    p.join(timeout=60)
    assert res == "ok", res


@pytest.mark.parametrize("workload", ["recon", "moco", "joint", "spark"])
def test_bench_self_launcher_eight_rank_environment(workload):
    """``python bench.py --gpus 8 --workload X`` up to the first GPU call (the driver's 8-GPU run is the first time RCCL
    sees more than one rank): the parent spawns eight children with the torch.distributed environment, every child derives its plan
    (LOCAL_RANK -> device index, per-GPU batch, lr rule batch * gpus / 256, MoCo's K % (B * world)), the children rendezvous on
    127.0.0.1 over gloo, all-reduce ones and compare their plans; no GPU and no HIP library is touched (``--dry-run-env``)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", workload, "--dry-run-env"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ok"] and d["ranks_seen"] == 8 and len(d["plans"]) == 8
    plans = sorted(d["plans"], key=lambda p: p["rank"])
    assert [p["dev_index"] for p in plans] == list(range(8)) and [p["local_rank"] for p in plans] == list(range(8))
    assert all(p["world"] == 8 and p["backend"] == "nccl" and p["global_batch"] == 256 and p["lr_rule"] == 1.0 for p in plans)
    assert len({p["master"] for p in plans}) == 1 and plans[0]["master"].startswith("127.0.0.1:")
    assert all(p["ipc_mode_legacy"] == "0" for p in plans)          # dmabuf IPC: RCCL across processes needs it on this image
    if workload == "moco":
        assert all(p["queue"] == {"K": 4096, "gathered_keys": 256, "divides": True} for p in plans)


def test_bench_refuses_a_queue_the_gathered_keys_do_not_divide():
    """MoCo's ``assert num_negatives % batch_size == 0`` (moco2_module.py:169) on the gathered key batch: 48 x 8 keys into K = 4096."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="8", RANK="3", LOCAL_RANK="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "moco", "--batch", "48", "--dry-run-env"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and "not a multiple of the gathered key batch" in (res.stdout + res.stderr)
