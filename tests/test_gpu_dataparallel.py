"""Data-parallel trainers on the GPU: two ranks of every pretraining trainer against one rank and against the
oracle.  Ranks run as child processes (tests/dp_child.py) over gloo with both ranks on cuda:0 -- the collectives, the arena
exchange, the embedding all-gather with B*rank label offsets and the SyncBN exchange are the code paths RCCL takes on a
multi-GPU node; with >= 2 GPUs and CMU_DIST_BACKEND=nccl the same tests run over RCCL.

Reference semantics: DistributedDataParallel gradient MEAN (Spark/main.py:102; dist_train.sh:9-17 + cmunet_config.py:120;
Lightning DDP for MoCo), concat_all_gather of the keys (cmunet_head.py:77-85, moco2_module.py:160-175), per-GPU encoder
BatchNorm (UNet_encoder.py:22,25), SyncBN in the necks (nonlinear_neck.py:45,58)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Ranks:
    """One group of dp_child.py ranks in flight (``launch_ranks``); ``collect()`` waits, reaps and loads the result dicts."""

    def __init__(self, td, out, procs, world):
        self.td, self.out, self.procs, self.world = td, out, procs, world

    def collect(self):
        try:
            for p in self.procs:
                rc = p.wait(timeout=420)
                assert rc == 0, f"rank exited with {rc}"
            return [torch.load(f"{self.out}.{rk}", weights_only=False) for rk in range(self.world)]
        finally:
            self.kill()

    def kill(self):
        for p in self.procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        self.td.cleanup()


def launch_ranks(mode, world, extra_env=None, **opts):
    """Start ``world`` ranks of dp_child.py and return at once.  Several groups may run side by side (round 6: the A/B tests below start both of
    their groups before waiting for either -- every child spends most of its few seconds importing torch; at most four of them touch the GPU at
    a time, within the box's limit of six)."""
    td = tempfile.TemporaryDirectory()
    out = os.path.join(td.name, "r")
    port = _free_port()
    procs = []
    try:
        for rk in range(world):
            env = dict(os.environ)
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            if world > 1:
                env.update(WORLD_SIZE=str(world), RANK=str(rk), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.update(extra_env or {})
            argv = [sys.executable, os.path.join(HERE, "dp_child.py"), mode, out] + [f"{k}={v}" for k, v in opts.items()]
            procs.append(subprocess.Popen(argv, env=env))
    except BaseException:
        _Ranks(td, out, procs, world).kill()
        raise
    return _Ranks(td, out, procs, world)


def run_ranks(mode, world, extra_env=None, **opts):
    """Start ``world`` ranks of dp_child.py, wait, return their result dicts.  Every child is reaped (and killed if a peer
    failed or the wait times out), so a failing rank cannot leave a process holding the GPU."""
    return launch_ranks(mode, world, extra_env, **opts).collect()


def run_groups(*groups):
    """``groups``: (mode, world, extra_env, opts) tuples, all started before any is waited for; returns their result lists in order."""
    live = []
    try:
        for mode, world, env, opts in groups:
            live.append(launch_ranks(mode, world, env, **opts))
        return [g.collect() for g in live]
    finally:
        for g in live:
            g.kill()


def rel(got, ref):
    return (got.double() - ref.double()).abs().max().item() / max(ref.double().abs().max().item(), 1e-9)


def test_masked_recon_two_ranks_same_batch_equals_one_rank(cuda):
    """Both ranks fed the same batches: SUM over ranks x 1/world == the local gradient exactly (2g * 0.5), so parameters after
    two steps are bit-identical to the single-process trainer -- with the overlapped two-bucket exchange and without."""
    (one,), two_ov = run_groups(("recon", 1, None, {}), ("recon", 2, {"CMU_DDP_OVERLAP": "1"}, {}))     # (at most five processes on the card, runner included)
    two_plain = run_ranks("recon", 2, extra_env={"CMU_DDP_OVERLAP": "0"})
    for overlap, two in (("1", two_ov), ("0", two_plain)):
        for r in two:
            assert r["losses"] == one["losses"]
            assert torch.equal(r["arena"], one["arena"]), f"overlap={overlap}: max diff {(r['arena'] - one['arena']).abs().max().item():.3e}"
            for k, v in one["bufs"].items():
                assert torch.equal(r["bufs"][k], v), k


def test_broadcast_after_a_forward_invalidates_packed_weights(cuda):
    """Advisor (round 2): ``broadcast_parameters`` rewrites the arena with a collective that neither the Parameters' version
    counters nor (before the fix) ``ops.PARAM_GENERATION`` see.  Ranks that start from DIFFERENT weights and run a forward before the
    broadcast must still train on rank 0's weights afterwards: both ranks end bit-identical to a single process started from rank
    0's weights (stale packed conv weights on rank 1 would change its loss and, through the all-reduce, everybody's update)."""
    (one,), two = run_groups(("recon", 1, None, {}), ("recon", 2, None, {"init": "rank", "prefwd": "1"}))
    for r in two:
        assert r["losses"] == one["losses"], (r["losses"], one["losses"])
        assert torch.equal(r["arena"], one["arena"])


def test_masked_recon_two_ranks_amp_and_f16(cuda):
    """The same with f16 storage and the dynamic loss scaler: the inf / nan check runs on the exchanged gradients, so both
    ranks take the same decision and stay bit-identical to each other and to one rank."""
    (one,), two = run_groups(("recon", 1, None, {"dtype": "f16", "amp": "1"}), ("recon", 2, None, {"dtype": "f16", "amp": "1"}))
    assert torch.equal(two[0]["arena"], two[1]["arena"])
    assert torch.equal(two[0]["arena"], one["arena"])
    assert all(np.isfinite(one["losses"]))


def test_masked_recon_two_ranks_different_batches_vs_oracle(cuda):
    """Different data per rank: the updated parameters follow the MEAN of the two ranks' gradients, each computed with its own
    per-GPU BatchNorm statistics (oracle: one CPU forward/backward per rank, then torch AdamW on the averaged gradient)."""
    from cmunet_amd.pretrain import create_random_patch_mask
    from oracle import cmunet as OC, unet as OU
    two = run_ranks("recon", 2, data="rank", steps=1)
    assert torch.equal(two[0]["arena"], two[1]["arena"])
    sd = OU.make_state_dict(base_ch=16, depth=3, seed=9)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    grads, losses = {k: 0.0 for k in names}, []
    for rk in range(2):
        osd = OU.clone_sd(sd, requires_grad=True)
        g = torch.Generator().manual_seed(100 + rk)
        x = torch.randn(2, 32, 64, generator=g)
        m = torch.from_numpy(create_random_patch_mask(2, 32, 16, 0.5, np.random.RandomState(100 + rk)))
        m = torch.cat([m, m], 2)
        logits = OU.unet_forward(x * (1 - m[0]).float(), osd, training=True)
        loss = OC.masked_mse(logits[:, 1], x, m)
        loss.backward()
        losses.append(float(loss))
        for k in names:
            grads[k] = grads[k] + osd[k].grad / 2
    for rk in range(2):
        assert abs(two[rk]["losses"][0] - losses[rk]) <= 1e-5 * max(1.0, abs(losses[rk]))
    # first AdamW step from the averaged gradient: p*(1 - lr*wd) - lr*g/(|g| + eps); compare where the sign of g is not in doubt
    off, checked = 0, 0
    for k in names:
        p0, g = sd[k], grads[k]
        n = p0.numel()
        got = two[0]["arena"][off:off + n].view_as(p0)
        off += ((n + 3) // 4) * 4
        wd = 0.0 if any(key in k for key in ("ln", "bias", "pos_embed", "mask_token", "cls_token")) else 0.05   # cmunet_config.py:84-91 (BatchNorm weights decay)
        exp = p0 * (1 - 1e-3 * wd) - 1e-3 * g / (g.abs() + 1e-8)
        sure = g.abs() > 2e-3 * g.abs().max()
        if ".0.bias" in k or ".3.bias" in k or not bool(sure.any()):
            continue
        assert (got - exp)[sure].abs().max().item() <= 2e-6, k
        checked += int(sure.sum())
    assert checked > 10000


def _joint_oracle(r, gather_dup, rank=0):
    from oracle import cmunet as OC
    sd = r["init"]
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.startswith("target_") else v.clone())
           for k, v in sd.items()}
    gather = (lambda t: torch.cat([t, t], 0)) if gather_dup else None
    ref = OC.forward_train(r["img"], r["img_t"], r["mask"].numpy(), r["rw"], r["rb"], osd, temperature=0.07, ct_weight=1.0, rc_weight=1.0,
                           gather=gather, rank=rank)
    (ref["loss_ct"] + ref["loss_rc"]).backward()
    return ref, osd


def _check_joint(r, ref, osd, lr=1e-3, momentum=0.9):
    assert abs(r["loss_rc"] - float(ref["loss_rc"])) <= 2e-4 * max(1, abs(float(ref["loss_rc"])))
    assert abs(r["loss_ct"] - float(ref["loss_ct"])) <= 2e-3 * max(1, abs(float(ref["loss_ct"])))
    init, final = r["init"], r["final"]
    checked = 0
    for k, v in osd.items():
        if not (torch.is_tensor(v) and v.requires_grad) or v.grad is None or ".0.bias" in k or ".3.bias" in k:
            continue
        g = v.grad
        if g.abs().max() < 1e-6:
            continue
        wd = 0.0 if any(key in k for key in ("ln", "bias", "pos_embed", "mask_token", "cls_token")) else 0.05   # cmunet_config.py:84-91 (BatchNorm weights decay)
        exp = init[k] * (1 - lr * wd) - lr * g / (g.abs() + 1e-8)
        sure = g.abs() > 2e-2 * g.abs().max()            # (gradient parity is 5e-3 of the max norm: test_gpu_pretrain)
        assert (final[k] - exp)[sure].abs().max().item() <= 5e-6, k
        checked += int(sure.sum())
    assert checked > 5000
    # EMA after the optimiser step (MomentumUpdateHook.after_train_iter): target = m*target + (1-m)*online_new
    for src, dst in (("backbone.", "target_backbone."), ("projector.", "target_projector.")):
        for k in final:
            if k.startswith(src) and final[k].is_floating_point() and "running" not in k:
                kt = dst + k[len(src):]
                assert rel(final[kt], init[kt] * momentum + final[k] * (1 - momentum)) <= 1e-6, kt


def test_joint_trainer_one_and_two_ranks_vs_oracle(cuda):
    """CM-UNet joint step through JointPretrainer (AdamW arena + EMA arenas).  One rank against the oracle; two ranks fed the
    same batch against the oracle with the key all-gather emulated (2B keys, labels i + B*rank) -- SyncBN over duplicated rows
    has the single-rank statistics, so everything but the contrastive loss's denominator is unchanged."""
    (one,), two = run_groups(("joint", 1, None, {}), ("joint", 2, None, {}))
    ref, osd = _joint_oracle(one, False)
    _check_joint(one, ref, osd)
    for rk, r in enumerate(two):
        ref2, osd2 = _joint_oracle(r, True, rank=rk)
        _check_joint(r, ref2, osd2)
    for k, v in two[0]["final"].items():
        if v.is_floating_point():
            assert torch.equal(v, two[1]["final"][k]), k                      # replicas stay identical
    assert abs(two[0]["loss_ct"] - one["loss_ct"]) > 1e-3                  # 2B keys: a different denominator than one rank


def test_moco_trainer_two_ranks_vs_oracle(cuda):
    """MocoPretrainer on two ranks with the same batch: keys gathered (2B rows enqueued, pointer += 2B), gradient mean, fused
    SGD-momentum, one-launch EMA of the key encoder -- against the oracle step with the gather emulated."""
    from oracle import moco as OM
    two = run_ranks("moco", 2)
    r = two[0]
    sd = r["init"]
    osd = {k: (v.clone().requires_grad_(True) if k.startswith("encoder_q.") and v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    queue, ptr = sd["queue"].clone(), sd["queue_ptr"].clone()
    ref, _, _ = OM.training_step(r["xq"], r["xk"], osd, queue, ptr, 0.2, 0.99, gather=lambda t: torch.cat([t, t], 0))
    ref.backward()
    assert abs(r["loss"] - float(ref)) <= 1e-3 * max(1.0, abs(float(ref)))
    fin = r["final"]
    assert int(fin["queue_ptr"]) == int(ptr) == 8 and rel(fin["queue"], queue) <= 1e-4
    lr, wd = 0.05, 1e-4
    for k in ("encoder_q.down_conv1.double_conv.double_conv.0.weight", "encoder_q.double_conv.double_conv.3.weight",
              "encoder_q.down_conv2.double_conv.double_conv.4.bias"):
        exp = sd[k] - lr * (osd[k].grad + wd * sd[k])                      # first SGD-momentum step: buf = g + wd*p
        assert rel(fin[k] - sd[k], exp - sd[k]) <= 5e-3, k
    kk = "encoder_k.double_conv.double_conv.0.weight"
    assert rel(fin[kk], osd[kk]) <= 1e-6                                    # EMA before the forward (A-8), one launch
    for k, v in fin.items():
        if v.is_floating_point():
            assert torch.equal(v, two[1]["final"][k]), k


def test_moco_shuffle_bn_two_ranks_vs_oracle(cuda):
    """Shuffle-BN (moco2_module.py:177-222): the key images of both ranks are gathered, permuted with rank 0's permutation,
    each rank encodes its share (its BatchNorm sees a mixture of both ranks' images), and the keys travel back to their owners.
    Oracle: the same permutation redrawn from the seed, the key encoder per shuffled group on the CPU."""
    import torch.nn.functional as F
    from oracle import moco as OM
    two = run_ranks("moco_shuffle", 2)
    B = 4
    xk_all = torch.cat([two[0]["xk"], two[1]["xk"]])
    torch.manual_seed(1234)
    idx = torch.randperm(2 * B)                                   # what rank 0 drew and broadcast
    un = torch.argsort(idx)
    ks = []
    for rk in range(2):
        sd = {k: v.clone() for k, v in two[rk]["init"].items()}
        ks.append(OM.encoder_gap(xk_all[idx.view(2, -1)[rk]], sd, "encoder_k.", True))
    k_ref = F.normalize(torch.cat(ks)[un], dim=1)
    for rk in range(2):
        assert rel(two[rk]["k"], k_ref[rk * B:(rk + 1) * B]) <= 1e-4, rk
    # and it matters: without the shuffle each rank's BatchNorm would see only its own images
    plain = F.normalize(OM.encoder_gap(two[0]["xk"], {k: v.clone() for k, v in two[0]["init"].items()}, "encoder_k.", True), dim=1)
    assert rel(two[0]["k"], plain) > 1e-3


def test_cmunet_head_two_ranks_vs_reference_two_rank_fixture(cuda):
    """Two ranks of the HIP CMUNetPretrainHead against what the REFERENCE's own head produced on two gloo ranks
    (tests/golden/cmunet_head_2rank.npz): gathered target projections, labels arange(B) + B * rank, per-rank losses and gradients."""
    f = np.load(os.path.join(HERE, "golden", "cmunet_head_2rank.npz"))
    two = run_ranks("head_ref2", 2, seed=int(f["seed"]))
    for rk in range(2):
        r = two[rk]
        assert abs(r["loss_rc"] - float(f["loss_rc"][rk])) <= 2e-5 * max(1.0, abs(float(f["loss_rc"][rk])))
        assert abs(r["loss_ct"] - float(f["loss_ct"][rk])) <= 2e-4 * max(1.0, abs(float(f["loss_ct"][rk]))), (rk, r["loss_ct"], float(f["loss_ct"][rk]))
        assert rel(r["dpred"], torch.from_numpy(f["dpred"][rk])) <= 1e-4
        assert rel(r["dproj_s"], torch.from_numpy(f["dproj_s"][rk])) <= 1e-3
        assert abs(r["dfc1_norm"] - float(f["dfc1_norm"][rk])) <= 1e-3 * float(f["dfc1_norm"][rk])
    assert abs(two[0]["loss_ct"] - two[1]["loss_ct"]) > 1e-4


def test_neck_syncbn_two_ranks_vs_reference_fixture(cuda):
    """The necks' TRAINING-mode SyncBatchNorm on two ranks (nonlinear_neck.py:45,58 under norm_cfg SyncBN) against the reference's own
    NonLinearNeck run over the CONCATENATED rows of both ranks (tests/golden/neck_syncbn_2rank.npz, oracle/gen_golden.py::
    gen_neck_syncbn_2rank) -- what SyncBN computes: statistics and backward sums over all ranks.  Per rank: outputs and input gradients
    are the fixture's rows; the two ranks' local parameter gradients SUM to the concatenated run's; running statistics agree on both."""
    f = np.load(os.path.join(HERE, "golden", "neck_syncbn_2rank.npz"))
    two = run_ranks("neck_ref2", 2)
    for rk in range(2):
        assert rel(two[rk]["y"], torch.from_numpy(f["y"][rk])) <= 2e-5, rk
        assert rel(two[rk]["dx"], torch.from_numpy(f["dx"][rk])) <= 2e-4, rk
        assert rel(two[rk]["running_mean"], torch.from_numpy(f["running_mean"])) <= 1e-5
        assert rel(two[rk]["running_var"], torch.from_numpy(f["running_var"])) <= 1e-5
    tot = {n: two[0]["grads"][n] + two[1]["grads"][n] for n in two[0]["grads"]}
    # (fc0's bias is followed by a training-mode BatchNorm: its gradient is a rounding remainder of sums that cancel, ~1e-8 in the
    # reference's run as well -- held to an absolute bar on the scale of the BatchNorm bias gradient)
    assert (tot["fc0.bias"] - torch.from_numpy(f["dfc0_bias"])).abs().max().item() <= 1e-5 * float(np.abs(f["dbn0_bias"]).max())
    assert rel(tot["bn0.weight"], torch.from_numpy(f["dbn0_weight"])) <= 2e-4
    assert rel(tot["bn0.bias"], torch.from_numpy(f["dbn0_bias"])) <= 2e-4
    assert rel(tot["fc1.weight"][:16], torch.from_numpy(f["dfc1_weight_rows"])) <= 2e-4
    assert abs(float(tot["fc1.weight"].double().norm()) - float(f["dfc1_weight_norm"])) <= 2e-4 * float(f["dfc1_weight_norm"])
    assert rel(tot["fc0.weight"][:8], torch.from_numpy(f["dfc0_weight_rows"])) <= 2e-4
    assert abs(float(tot["fc0.weight"].double().norm()) - float(f["dfc0_weight_norm"])) <= 2e-4 * float(f["dfc0_weight_norm"])
    # and it is not the per-rank BatchNorm: one rank alone normalises with its own four rows
    one = run_ranks("neck_ref2", 1)[0]
    assert rel(one["y"], torch.from_numpy(f["y"][0])) > 1e-2


def test_moco_two_ranks_vs_reference_two_rank_fixture(cuda):
    """Two ranks of the HIP Moco_v2 against what the REFERENCE's own Moco_v2 produced on two gloo ranks
    (tests/golden/moco_ref_2rank.npz, oracle/gen_golden.py::gen_moco_2rank): shuffle-BN with rank 0's broadcast permutation, keys
    gathered from both ranks and enqueued on both (pointer += 2B), per-rank losses and local gradient norms."""
    f = np.load(os.path.join(HERE, "golden", "moco_ref_2rank.npz"))
    two = run_ranks("moco_ref2", 2, seed=int(f["seed"]))
    B = int(f["B"])
    qkeys = [str(k) for k in f["qkeys"]]
    live = torch.tensor([not k.endswith((".0.bias", ".3.bias")) for k in qkeys])
    for rk in range(2):
        r = two[rk]
        assert r["qkeys"] == qkeys
        assert abs(r["loss"] - float(f["loss"][rk])) <= 1e-3 * max(1.0, abs(float(f["loss"][rk]))), (rk, r["loss"], float(f["loss"][rk]))
        assert r["ptr"] == int(f["queue_ptr"].reshape(-1)[0]) == 2 * B
        assert rel(r["queue"][:, :2 * B].t(), torch.from_numpy(f["keys"])) <= 1e-4
        ref = torch.from_numpy(f["grad_norms"][rk]).double()
        e = ((r["grad_norms"] - ref).abs() / ref.clamp_min(1e-30))[live]
        assert float(e.max()) <= 5e-3, (rk, float(e.max()))
        assert rel(r["grad0"], torch.from_numpy(f["grad0"][rk])) <= 5e-3
        assert rel(r["bn_k"], torch.from_numpy(f["bn_k"][rk])) <= 1e-4            # the key encoder's BatchNorm saw the shuffled mixture
    assert torch.equal(two[0]["queue"], two[1]["queue"])


def test_spark_trainer_two_ranks_equals_one_rank(cuda):
    """SparKPretrainer (LAMB) with grad-less ``densify_projs`` parameters in the arena (SURVEY A-10): two ranks fed the same
    batches match one rank; a static loss scale inside the fused step leaves the update unchanged."""
    (one,), two, (scaled,) = run_groups(("spark", 1, None, {}), ("spark", 2, None, {}), ("spark", 1, None, {"loss_scale": 256}))
    assert any(n.startswith("densify_projs") for n in one["names"])
    for r in two:
        assert np.allclose(r["losses"], one["losses"], rtol=1e-6)
        assert rel(r["arena"], one["arena"]) <= 1e-6
    assert np.allclose(scaled["losses"], one["losses"], rtol=1e-6) and rel(scaled["arena"], one["arena"]) <= 1e-5


@pytest.mark.parametrize("mode", ["joint", "moco", "spark"])
def test_arena_trainers_overlapped_exchange_is_bit_identical(cuda, mode):
    """ArenaTrainer's backward-overlapped bucket exchange (what DDP's bucketed all-reduce does for the reference: Spark/main.py:102,
    cmunet_config.py:120) against ONE all-reduce of the whole arena behind loss.backward() (CMU_DDP_OVERLAP=0), two ranks each: a SUM
    over two ranks is the same number whatever the cut of the arena, so parameters, momentum networks and buffers must agree bit for
    bit -- and the overlapped run must really have started buckets before the backward pass was over (joint / MoCo: the decoders
    and the bottleneck are announced from inside the fused node; the projector / predictor by autograd's hooks)."""
    ov, plain = run_groups((mode, 2, {"CMU_DDP_OVERLAP": "1"}, {}), (mode, 2, {"CMU_DDP_OVERLAP": "0"}, {}))
    for rk in range(2):
        a, b = ov[rk], plain[rk]
        if mode == "spark":
            assert a["losses"] == b["losses"]
            assert torch.equal(a["arena"], b["arena"]), float((a["arena"] - b["arena"]).abs().max())
        else:
            for k, v in a["final"].items():
                assert torch.equal(v, b["final"][k]), k
    ex = ov[0]["exchange"]
    assert ex is not None and ex["buckets"] >= 2 and ex["early"] + ex["in_backward"] + ex["flushed"] == ex["buckets"]
    if mode == "joint":
        assert ex["early"] >= 3 and ex["in_backward"] >= 2, ex        # pixel / feature decoder + bottleneck; projector + predictor
    elif mode == "moco":
        assert ex["early"] >= 1, ex                                      # the query encoder's bottleneck
    assert plain[0]["exchange"] is None or plain[0]["exchange"].get("early", 0) == 0


def test_joint_sharded_projector_exchange_is_bit_identical(cuda):
    """Round 6 (review item 6): the joint step's projector.fc0.weight -- 1.61 GB of gradient at the bench's geometry (cmunet_config.py:18-26 under
    the DDP of dist_train.sh:9-17) -- exchanged as reduce-scatter + AdamW / EMA on the own share + all-gather of the updated parameter inside the
    NEXT forward (JointPretrainer._setup_shard) against the all-reduce path (CMU_DP_SHARD_PROJECTOR=0), two ranks, two steps each (the second
    forward is where the first step's all-gather completes): parameters, momentum networks, buffers AND the optimiser's moments (gathered for
    the checkpoint) agree bit for bit -- a SUM over two ranks is the same number either way, and the optimiser kernel runs on the same operands
    element for element.  With the overlapped exchange (f32) and with one exchange behind the backward pass (f16 under the dynamic loss scaler:
    the ranks' inf / nan flags are combined)."""
    # (f32 with the backward-overlapped exchange; f16 + dynamic loss scaler with ONE exchange behind the backward pass, CMU_DDP_OVERLAP=0: the
    # same split of the arena either way)
    slow = os.environ.get("CMU_TEST_SLOW") == "1"      # (the f16 + loss-scaler / one-exchange twin: run by the builder once per round, profiles/r06_parity.txt)
    for opts, env in [({"steps": 2}, {})] + ([({"steps": 2, "dtype": "f16"}, {"CMU_DDP_OVERLAP": "0"})] if slow else []):
        sh, plain = run_groups(("joint", 2, dict(env, CMU_DP_SHARD_MIN="0"), opts), ("joint", 2, dict(env, CMU_DP_SHARD_PROJECTOR="0"), opts))
        assert all(r["sharded"] for r in sh) and not any(r["sharded"] for r in plain)
        if not env:
            ex = sh[0]["exchange"]
            assert ex is not None and ex.get("reduce_scatter_bytes", 0) > 0, ex
        for rk in range(2):
            a, b = sh[rk], plain[rk]
            assert np.isfinite(a["loss_ct"]) and a["loss_ct"] == b["loss_ct"] and a["loss_rc"] == b["loss_rc"]
            for k, v in b["final"].items():
                assert torch.equal(a["final"][k], v), (opts, k)
            assert a["opt"]["step"] == b["opt"]["step"] == 2
            for k in ("m", "v"):
                assert torch.equal(a["opt"][k], b["opt"][k]), (opts, k)
        for k, v in sh[0]["final"].items():                  # the replicas agree with each other
            assert torch.equal(sh[1]["final"][k], v), k


@pytest.mark.parametrize("mode", ["recon", "joint", "moco", "spark"])
def test_no_gradient_is_written_behind_its_bucket_exchange(cuda, mode):
    """The in-forward / in-backward bucket announcements are the only guarantee that no gradient of a bucket is written after that
    bucket's all-reduce started (advisor, round 4): on a one-rank group (CMU_DP_REHEARSE=1: the SUM is an identity, a late write would
    pass every equal-loss test) CMU_DP_CHECK_LATE_WRITES=1 copies each bucket when its exchange starts and compares it bit for bit
    after the waits.  The trainers' own steps pass; a deliberate write behind the first exchange (late=1) is caught."""
    env = {"CMU_DP_REHEARSE": "1", "CMU_DP_CHECK_LATE_WRITES": "1"}
    keys = {"recon": ("losses",), "spark": ("losses",), "joint": ("loss_ct", "loss_rc"), "moco": ("loss",)}[mode]
    (ok,), (plain,), (bad,) = run_groups((mode, 1, env, {"port": _free_port()}), (mode, 1, None, {}), (mode, 1, env, {"port": _free_port(), "late": 1}))
    assert "raised" not in ok
    for k in keys:
        assert np.all(np.isfinite(ok[k])) and ok[k] == plain[k], (k, ok[k], plain[k])
    if mode != "recon":
        ex = ok["exchange"]
        assert ex is not None and ex["early"] + ex["in_backward"] >= 1, ex       # exchanges really started inside the step
    assert "raised" in bad and "after its all-reduce had started" in bad["raised"], bad


@pytest.mark.parametrize("workload", ["recon", "joint", "moco", "spark"])
def test_bench_step_on_a_one_rank_rccl_group(cuda, workload):
    """The collectives of every trainer on RCCL itself (backend "nccl"): bench.py as ONE rank of an initialised group with
    CMU_DP_REHEARSE=1 -- the overlapped bucket all-reduces behind the backward, the arena all-reduce, the embedding all-gathers,
    the barrier and the MAX over ranks all run on the process group's stream -- must give the loss of the same steps without a group
    (a SUM over one rank is the identity).  Two RCCL ranks cannot share the one card of this box; the two-rank tests above use gloo.
    CMU_DP_CHECK_LATE_WRITES=1: every bucket is copied when its exchange starts and compared after the waits -- a gradient written into
    a bucket behind its (identity) all-reduce, which equal losses would not show, raises."""
    import json
    root = os.path.dirname(HERE)
    argv = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4", "--size", "128",
            "--workload", workload, "--no-cpu-baseline", "--no-kernel-events"]
    base = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CMU_DP_REHEARSE", "CMU_DIST_BACKEND"):
        base.pop(k, None)
    lines, procs = [], []
    try:
        # (both runs side by side: most of either is interpreter start-up)
        for env in (base, dict(base, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                               CMU_DP_REHEARSE="1", CMU_DP_CHECK_LATE_WRITES="1")):
            procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        for pr in procs:
            so, se = pr.communicate(timeout=420)
            assert pr.returncode == 0, se[-2000:]
            lines.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
            pr.wait()
    plain, rccl = lines
    if workload == "joint":
        # round 6: the sharded projector exchange on RCCL itself -- reduce_scatter_tensor / all_gather_into_tensor of a one-rank group are
        # identities, so the losses must not move (the late-write check above takes the all-reduce path: it snapshots whole buckets)
        env = dict(base, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), CMU_DP_REHEARSE="1")
        r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=420)
        assert r.returncode == 0, r.stderr[-2000:]
        shard = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert shard["rccl"]["exchange"].get("reduce_scatter_bytes", 0) == 4 * 128 * 128 * 1536, shard["rccl"]["exchange"]
        assert abs(shard["config"]["loss"] - plain["config"]["loss"]) <= 1e-5 * max(1.0, abs(plain["config"]["loss"]))
        assert "reduce_scatter_bytes" not in rccl["rccl"]["exchange"]
    assert rccl["n_gpus"] == 1 and rccl["config"]["parallelism"] == "dp1"
    assert np.isfinite(rccl["config"]["loss"])
    assert abs(rccl["config"]["loss"] - plain["config"]["loss"]) <= 1e-5 * max(1.0, abs(plain["config"]["loss"])), (plain["config"]["loss"], rccl["config"]["loss"])
