"""Tile skipping of the SparK sparse encoder (reference semantics Spark/encoder.py:20-36: dense op, then
`*= active`): the device-side tile list, the persistent conv kernel and the weight-gradient kernel over listed tiles against
their dense forms, and the whole SparK step with and without skipping."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import ops as o
    return o


def _active(B, f, keep, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.zeros(B, f * f, dtype=torch.uint8)
    for b in range(B):
        a[b, torch.randperm(f * f, generator=g)[:keep]] = 1
    return a.view(B, f, f)


@pytest.mark.parametrize("case", [(3, 4, 64, 16, 32, 4), (2, 8, 128, 16, 32, 16), (2, 8, 128, 16, 16, 16), (2, 8, 64, 16, 32, 16),
                                  (1, 32, 512, 16, 32, 256), (4, 4, 64, 16, 16, 0), (2, 4, 64, 16, 32, 16), (5, 8, 256, 16, 32, 3)])
def test_tile_list_matches_numpy(ops, case):
    B, f, H, th, tw, keep = case
    act = _active(B, f, keep, seed=sum(case))
    tl = ops.TileList(act.cuda(), H, H, th, tw)
    n = int(tl.count.item())
    got = tl.list[:n].cpu().numpy()
    px = np.repeat(np.repeat(act.numpy(), H // f, 1), H // f, 2)            # (B, H, H) pixel map
    ty, tx = H // th, H // tw
    on = px.reshape(B, ty, th, tx, tw).max(axis=(2, 4)).reshape(-1)
    ref = np.nonzero(on)[0]
    assert n == len(ref) and np.array_equal(got, ref), (n, len(ref))
    assert tl.n_dense == B * ty * tx


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("shape", [(2, 128, 64, 64, 8), (2, 64, 64, 128, 4), (1, 128, 128, 128, 8), (3, 64, 256, 128, 4)])
def test_conv_tiles_bit_identical_on_listed_tiles_and_untouched_elsewhere(ops, dt, shape):
    """The persistent kernel over a tile list: on the listed 16 x 32 tiles the output equals the dense launch bit for bit
    (same kernel, same per-tile arithmetic); the other tiles keep what was there before (a sentinel)."""
    B, S, Cin, Cout, f = shape
    if not ops.conv3x3_tiles_supported(B, S, S, Cin, Cout, dt):
        pytest.skip("shape not served by the persistent kernel for this dtype")
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator(device="cuda").manual_seed(1)
    act = _active(B, f, max(1, f * f // 4), seed=S + Cin).cuda()
    x = torch.randn(B, S, S, Cin, generator=g, device="cuda").to(tdt)
    sc = (1 + 0.2 * torch.randn(Cin, generator=g, device="cuda")).contiguous()
    sh = (0.1 * torch.randn(Cin, generator=g, device="cuda")).contiguous()
    w = torch.randn(Cout, Cin, 3, 3, generator=g, device="cuda") / (3 * Cin ** 0.5)
    wp = ops.pack_conv3x3(w, dt)
    xa = ops.Act(x, 0, Cin, sc, sh, 0)
    dense = ops.new_act(B, S, S, Cout, dt, "cuda")
    ops.conv3x3_fwd(xa, wp, dense, None)
    tl = ops.TileList(act, S, S, 16, 32)
    out = ops.Act(torch.full((B, S, S, Cout), 7.0, dtype=tdt, device="cuda"))
    ops.conv3x3_fwd_tiles(xa, wp, out, tl)
    n = int(tl.count.item())
    listed = torch.zeros(tl.n_dense, dtype=torch.bool, device="cuda")
    listed[tl.list[:n].long()] = True
    m = listed.view(B, S // 16, S // 32).repeat_interleave(16, 1).repeat_interleave(32, 2)     # (B, S, S) pixels of listed tiles
    assert 0 < n < tl.n_dense
    assert torch.equal(out.buf[m], dense.buf[m]), "listed tiles differ from the dense launch"
    assert bool((out.buf[~m] == 7.0).all()), "a skipped tile was written"
    # every active pixel lies in a listed tile
    pix = act.bool().repeat_interleave(S // f, 1).repeat_interleave(S // f, 2)
    assert bool((m | ~pix).all())


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("shape", [(2, 128, 64, 64, 8), (2, 64, 32, 64, 4), (1, 128, 128, 64, 8),
                                   (2, 512, 64, 64, 32)])      # level 1 of a 512 x 512 input, two images (config 5's geometry)
def test_wgrad_tiles_equals_dense_when_dy_is_masked(ops, dt, shape):
    """Weight gradient over the listed 16 x 16 tiles == the dense weight gradient when dY vanishes outside the active patches
    (the sparse BatchNorm backward writes zeros there); only the split-K summation order differs."""
    from cmunet_amd import _lib
    B, S, Cin, Cout, f = shape
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator(device="cuda").manual_seed(2)
    act = _active(B, f, max(1, f * f // 4), seed=S + Cout).cuda()
    pix = act.bool().repeat_interleave(S // f, 1).repeat_interleave(S // f, 2).unsqueeze(-1)
    x = torch.randn(B, S, S, Cin, generator=g, device="cuda").to(tdt)
    dy = (torch.randn(B, S, S, Cout, generator=g, device="cuda") * pix).to(tdt)
    ws = torch.empty(_lib.lib().cmu_conv3x3_wgrad_ws_bytes(B, S, S, Cin, Cout, ops.dt_code(dt)), dtype=torch.uint8, device="cuda")
    dW = torch.empty(Cout, Cin, 3, 3, device="cuda")
    ops.conv3x3_wgrad(ops.Act(x), ops.Act(dy), dW, ws)
    tl = ops.TileList(act, S, S, 16, 16)
    dWt = torch.empty_like(dW)
    ops.conv3x3_wgrad_tiles(ops.Act(x), ops.Act(dy), dWt, ws, tl)
    assert 0 < int(tl.count.item()) < tl.n_dense
    e = (dW - dWt).abs().max().item() / dW.abs().max().item()
    assert e <= (1e-5 if dt != "f32" else 2e-5), e
    # against float64 as well
    rdev = "cuda" if S >= 256 else "cpu"           # (the full-size geometry: float64 on the GPU, 39 GFLOP)
    ref = torch.nn.grad.conv2d_weight(x.double().to(rdev).permute(0, 3, 1, 2), (Cout, Cin, 3, 3), dy.double().to(rdev).permute(0, 3, 1, 2), padding=1).cpu()
    assert (dWt.double().cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 128, 8), (3, 96, 128, 128, 12), (2, 128, 64, 256, 16), (1, 32, 128, 128, 2),
                                   # 64 -> 64 (SparK level 1): the 64 n x 64 c form (conv_wgrad2s.inc) walks the same 8 x 16 lists
                                   (2, 128, 64, 64, 8), (3, 64, 64, 64, 4),
                                   # config 5's geometry, two-image sub-batch of a 512 x 512 input: level 1 (64 -> 64 @ 512) and level 2
                                   # (64 -> 128 and 128 -> 128 @ 256), 32 x 32 patch map
                                   (2, 512, 64, 64, 32), (2, 256, 64, 128, 32), (2, 256, 128, 128, 32)])
def test_wgrad_tiles_wide_kernel_8x16_list(ops, dt, shape):
    """The wide weight-gradient kernel over a list of 8 x 16 pixel tiles (its K tile; SparK level 2: one tile = two 8 x 8 patches):
    equal to the dense launch when dY vanishes outside the active patches, and to float64.  Lists longer and shorter than the
    split count, a batch of three, 8-pixel and larger patches; ``conv3x3_wgrad_tile_h`` names the list a shape wants."""
    from cmunet_amd import _lib
    B, S, Cin, Cout, f = shape
    assert ops.conv3x3_wgrad_tile_h(B, S, S, Cin, Cout, dt) == 8 and ops.conv3x3_wgrad_tile_h(B, S, S, 32, 64, dt) == 16
    assert ops.conv3x3_wgrad_tile_h(B, S, S, Cin, Cout, "f32") == 16
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator(device="cuda").manual_seed(4)
    act = _active(B, f, max(1, f * f // 4), seed=S + Cout + f).cuda()
    pix = act.bool().repeat_interleave(S // f, 1).repeat_interleave(S // f, 2).unsqueeze(-1)
    x = torch.randn(B, S, S, Cin, generator=g, device="cuda").to(tdt)
    sc, sh = (1 + 0.2 * torch.randn(Cin, generator=g, device="cuda")).contiguous(), (0.1 * torch.randn(Cin, generator=g, device="cuda")).contiguous()
    dy = (torch.randn(B, S, S, Cout, generator=g, device="cuda") * pix).to(tdt)
    ws = torch.empty(_lib.lib().cmu_conv3x3_wgrad_ws_bytes(B, S, S, Cin, Cout, ops.dt_code(dt)), dtype=torch.uint8, device="cuda")
    xa = ops.Act(x, 0, Cin, sc, sh, 0)                       # pending BatchNorm + ReLU on X, as in the encoder
    dW = torch.empty(Cout, Cin, 3, 3, device="cuda")
    ops.conv3x3_wgrad(xa, ops.Act(dy), dW, ws)
    tl = ops.TileList(act, S, S, 8, 16)
    n = int(tl.count.item())
    assert 0 < n < tl.n_dense
    dWt = torch.full_like(dW, 7.0)
    ops.conv3x3_wgrad_tiles(xa, ops.Act(dy), dWt, ws, tl)
    e = (dW - dWt).abs().max().item() / dW.abs().max().item()
    assert e <= 1e-5, e
    rdev = "cuda" if S >= 256 else "cpu"           # (the full-size geometry: float64 on the GPU)
    xt = torch.relu(x.double().to(rdev) * sc.double().to(rdev) + sh.double().to(rdev))
    ref = torch.nn.grad.conv2d_weight(xt.permute(0, 3, 1, 2), (Cout, Cin, 3, 3), dy.double().to(rdev).permute(0, 3, 1, 2), padding=1).cpu()
    assert (dWt.double().cpu() - ref).abs().max().item() <= (2e-3 if dt == "f16" else 1.6e-2) * ref.abs().max().item()
    # a 16 x 16 list is refused for nothing: it runs the first kernel; an 8 x 16 list on a shape of the first kernel is refused
    with pytest.raises(Exception, match="8 x 16 tile list needs"):
        ops.conv3x3_wgrad_tiles(ops.Act(x[..., :32].contiguous()), ops.Act(dy[..., :64].contiguous()), torch.empty(64, 32, 3, 3, device="cuda"), ws, tl)


@pytest.mark.parametrize("case", [(2, 4, 32, 4), (3, 8, 32, 16), (2, 8, 8, 64), (1, 4, 64, 5), (2, 16, 16, 0)])
def test_pixel_list_matches_numpy(ops, case):
    """cmu_sparse_pixel_list: active pixels in patch-major order (patches ascending, pixels row-major inside a patch), -1 padding."""
    B, f, H, keep = case
    act = _active(B, f, keep, seed=sum(case))
    pl = ops.PixelList(act.cuda(), H, H, max_rows=B * keep * (H // f) ** 2 if keep else None)
    n = int(pl.count.item())
    s = H // f
    ref = []
    for b in range(B):
        for fy in range(f):
            for fx in range(f):
                if act[b, fy, fx]:
                    for py in range(s):
                        for px in range(s):
                            ref.append((b * H + fy * s + py) * H + fx * s + px)
    got = pl.rows.cpu().numpy()
    assert n == len(ref) and np.array_equal(got[:n], np.array(ref, dtype=np.int32)) and (got[n:] == -1).all()
    assert pl.capacity % 256 == 0 and pl.capacity >= n


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("shape", [(2, 32, 128, 256, 8), (2, 16, 256, 256, 8), (3, 8, 512, 512, 4), (1, 32, 64, 256, 16), (2, 4, 1024, 1024, 4),
                                   (2, 64, 64, 128, 8), (2, 32, 128, 128, 8), (1, 16, 128, 384, 4)])
def test_conv_rows_gather_vs_dense_at_active_pixels(ops, dt, shape):
    """The gather-GEMM over the list of active pixels against the dense launch on the same (masked) input: equal at every
    active pixel up to the summation order (tap-major K here, slice-major there), untouched elsewhere; with the flipped pack
    (data gradient) as well."""
    B, S, Cin, Cout, f = shape
    if not ops.conv3x3_rows_supported(B, S, S, Cin, Cout, dt):
        pytest.skip("shape not served by the gather kernel for this dtype")
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator(device="cuda").manual_seed(3)
    act = _active(B, f, max(1, f * f // 4), seed=S + Cin).cuda()
    pix = act.bool().repeat_interleave(S // f, 1).repeat_interleave(S // f, 2)
    x = (torch.randn(B, S, S, Cin, generator=g, device="cuda") * pix.unsqueeze(-1)).to(tdt)           # masked input, as in SparK
    w = torch.randn(Cout, Cin, 3, 3, generator=g, device="cuda") / (3 * Cin ** 0.5)
    pl = ops.PixelList(act, S, S, max_rows=int(act.sum()) * (S // f) ** 2)
    tol = {"f32": 2e-5, "f16": 2e-3, "bf16": 1.6e-2}[dt]
    for flip in (False, True):
        if flip and not ops.conv3x3_rows_supported(B, S, S, Cout, Cin, dt):
            continue
        ci, co = (Cout, Cin) if flip else (Cin, Cout)
        xin = x if not flip else (torch.randn(B, S, S, ci, generator=g, device="cuda") * pix.unsqueeze(-1)).to(tdt)
        wp = ops.pack_conv3x3(w, dt, transpose_flip=flip)
        dense = ops.new_act(B, S, S, co, dt, "cuda")
        ops.conv3x3_fwd(ops.Act(xin), wp, dense, None)
        out = ops.Act(torch.full((B, S, S, co), 7.0, dtype=tdt, device="cuda"))
        ops.conv3x3_fwd_rows(ops.Act(xin), wp, out, pl)
        a, b = out.buf[pix].float(), dense.buf[pix].float()
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)
        assert err <= tol, (flip, err)
        assert bool((out.buf[~pix] == 7.0).all()), "a masked pixel was written"


@pytest.mark.parametrize("dt", ["f16", "f32"])
def test_rows_statistics_equal_masked_statistics(ops, dt):
    """The sparse-BatchNorm statistics (forward sums, backward sums) driven by the list of active pixels == the masked passes
    over all pixels (same pixels, different visiting order: fp32 summation-order tolerance)."""
    from cmunet_amd import _lib
    B, S, C, f = 3, 64, 128, 8
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator(device="cuda").manual_seed(4)
    act = _active(B, f, 17, seed=9).cuda()
    y = ops.Act(torch.randn(B, S, S, C, generator=g, device="cuda").to(tdt))
    pl = ops.PixelList(act, S, S)
    a = ops.masked_channel_stats(y, act).double().sum(0)
    b = ops.rows_channel_stats(y, pl).double().sum(0)
    assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    mean, invstd = torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5
    yt = y.with_transform(sc, sh, 0)
    dA = ops.Act(torch.randn(B, S, S, C, generator=g, device="cuda").to(tdt))
    count = int(act.sum()) * (S // f) ** 2
    outs = []
    for use_rows in (False, True):
        dg, db, coef = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(2, C, device="cuda")
        ws = torch.empty(_lib.lib().cmu_bn_bwd_ws_bytes(C), dtype=torch.uint8, device="cuda")
        if use_rows:
            ops.bn_bwd_reduce_rows(dA, yt, mean, invstd, dg, db, coef, pl, count, ws)
        else:
            ops.bn_bwd_reduce_masked(dA, yt, mean, invstd, dg, db, coef, act, count, ws)
        outs.append((dg.clone(), db.clone(), coef.clone()))
    for u, v in zip(*outs):
        assert (u - v).abs().max().item() <= 2e-5 * max(v.abs().max().item(), 1e-6)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("case", [(2, 4, 16, 32), (3, 8, 16, 64), (1, 8, 32, 16)])     # (B, f, H, C): patch sides 4, 2, 4
def test_masked_pools_equal_the_materialised_form(ops, dt, case):
    """Round 3 (SparK): the mask-aware pools work from the raw conv output + transform + patch mask, so the activated + masked copy of a
    level's second conv output (cmu_mask_select) is never written.  Forward: identical bits to select -> plain pool (rounding is
    monotonic, so max and rounding commute).  Backward: at f32 identical bits on every active window; at 16 bits the arg-max is taken
    on the un-rounded activations, so a window whose two largest values ROUND to the same number may route its pooled gradient to the
    other one -- the window's gradient SUM is identical, and such windows are rare.  Masked windows: zero forward, untouched backward."""
    B, f, H, C = case
    W = H
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator().manual_seed(B * 100 + f)
    act = _active(B, f, max(1, f * f // 3), 7).cuda()
    y = ops.Act(torch.randn(B, H, W, C, generator=g).to(tdt).cuda(), 0, C, (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda(), 0)
    a = ops.new_act(B, H, W, C, dt, "cuda")
    ops.mask_select(y, act, a, relu=True)
    one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ref, got = ops.new_act(B, H // 2, W // 2, C, dt, "cuda"), ops.new_act(B, H // 2, W // 2, C, dt, "cuda")
    ops.bnrelu_maxpool_fwd(a.with_transform(one, zero, 0), ref)
    ops.bnrelu_maxpool_fwd_masked(y, act, got)
    assert torch.equal(ref.buf.view(torch.uint8), got.buf.view(torch.uint8))
    up = act.repeat_interleave(H // f, 1).repeat_interleave(W // f, 2).bool()            # (B,H,W) pixel mask
    assert bool((got.buf.float()[~up[:, ::2, ::2]] == 0).all())
    dP = ops.Act(torch.randn(B, H // 2, W // 2, C, generator=g).to(tdt).cuda())
    dS = ops.Act(torch.randn(B, H, W, 2 * C, generator=g).to(tdt).cuda(), C, C)         # strided skip gradient
    dA_ref, dA = ops.new_act(B, H, W, C, dt, "cuda"), ops.Act(torch.full((B, H, W, C), 7.0, device="cuda").to(tdt))
    ops.maxpool_bwd(dP, dS, a.with_transform(one, zero, 0), dA_ref)
    ops.maxpool_bwd_masked(dP, dS, y, dA, act)
    assert bool((dA.buf.float()[~up] == 7.0).all())                                      # masked windows are left alone
    r, q = dA_ref.buf.float()[up], dA.buf.float()[up]
    if dt == "f32":
        assert torch.equal(r, q)
    else:
        # per 2x2 window and channel the gradient sum agrees; elementwise differences only where the arg-max moved inside a window
        win = lambda t: t.float().view(B, H // 2, 2, W // 2, 2, C).sum((2, 4))
        mw = up[:, ::2, ::2]
        tol = 2e-2 if dt == "f16" else 1e-1
        assert (win(dA.buf)[mw] - win(dA_ref.buf)[mw]).abs().max().item() <= tol
        assert (r != q).float().mean().item() <= 2e-2


_STEP = r'''
import sys, torch
sys.path.insert(0, %r)
from cmunet_amd import spark as S
torch.manual_seed(5)
dt = sys.argv[1]
enc = S.build_sparse_encoder("unet_sparse", input_size=128, base_ch=64, depth=5, dtype=dt)
model = S.SparK(enc, S.UnetDecoder(base_ch=64, depth=5, dtype=dt), mask_ratio=0.75, densify_norm="", dtype=dt).cuda().train()
g = torch.Generator().manual_seed(11)
x = torch.randn(6, 1, 128, 128, generator=g).cuda()
active = model.mask(6, "cuda", g)
loss = model(x, active_b1ff=active)
loss.backward()
torch.save({"loss": loss.detach().cpu(), "grads": {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}}, sys.argv[2])
'''


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_spark_step_with_and_without_tile_skipping(ops, dt, tmp_path):
    """The whole SparK step (reference geometry, 128 px, mask ratio 0.75) with tile lists + gather levels (default), tile lists
    only (CMU_SPARK_GATHER=0) and fully dense (CMU_SPARK_TILES=0): tile lists give the identical loss and equal gradients up to
    the weight gradients' summation order; the gather levels agree to rounding."""
    outs = {}
    # ("10": tile lists only, the one-channel first layer on its dense kernels -- the same arithmetic as the dense step at every active
    # pixel, patch-organised element-wise passes and border-frame zeroing included; "10c": the first layer over its tile list as well --
    # its statistics are then summed in another order, which sparse BatchNorm amplifies like the gather levels' differences)
    # (round 6: the child processes run three and two at a time -- each is mostly interpreter start-up)
    runs = [(key, dict(CMU_SPARK_TILES=flag, CMU_SPARK_GATHER=gather, CMU_SPARK_C1_TILES=c1, CMU_POISON_NEW="1"))
            for key, flag, gather, c1 in (("11", "1", "1", "1"), ("10", "1", "0", "0"), ("10c", "1", "0", "1"), ("00", "0", "0", "1"))]
    # (CMU_POISON_NEW: every fresh activation starts as NaN -- the list-driven layers leave masked patches unwritten, or zero only
    # their border frames; a kernel that read such a position would poison the loss and the gradients)
    # round 3: the mask-aware pools (no activated copy of a level's second conv output) against the materialised form, everything else
    # at its default: the forward is the same arithmetic (identical loss at f32), the backward differs by arg-max ties only
    runs.append(("nofuse", dict(CMU_SPARK_POOL_FUSE="0", CMU_POISON_NEW="1")))
    for batch in (runs[:3], runs[3:]):          # (with the test runner itself at most four processes on the card at a time)
        procs = []
        try:
            for key, env in batch:
                procs.append((key, subprocess.Popen([sys.executable, "-c", _STEP % ROOT, dt, str(tmp_path / f"r{key}.pt")], env=dict(os.environ, **env))))
            for key, pr in procs:
                assert pr.wait(timeout=420) == 0, key
        finally:
            for _, pr in procs:
                if pr.poll() is None:
                    pr.kill()
                pr.wait()
    for key, _ in runs:
        outs[key] = torch.load(str(tmp_path / f"r{key}.pt"))
    o = None
    nofuse = outs.pop("nofuse")
    for o_ in list(outs.values()) + [nofuse]:
        assert bool(torch.isfinite(o_["loss"]).all()) and all(bool(torch.isfinite(g_).all()) for g_ in o_["grads"].values()), "an unwritten position was read"
    if dt == "f32":
        assert float(nofuse["loss"]) == float(outs["11"]["loss"])
    num = sum((outs["11"]["grads"][k] - g0).double().pow(2).sum().item() for k, g0 in nofuse["grads"].items())
    den = sum(g0.double().pow(2).sum().item() for g0 in nofuse["grads"].values())
    assert (num / den) ** 0.5 <= (1e-5 if dt == "f32" else 5e-2), (num / den) ** 0.5
    # tile lists alone: the forward is the same arithmetic on every active pixel
    assert float(outs["10"]["loss"]) == float(outs["00"]["loss"])
    # The gather kernel sums K tap-major, the statistics passes visit the pixels in list order: rounding-level differences, which
    # sparse BatchNorm over a few dozen positions amplifies by a condition number of ~5e4 (the reference's own f32 run sits ~3e-3
    # from its f64 run, tests/test_gpu_pretrain.py).  So this whole-step A/B is a sanity bound on relative L2 errors per tensor;
    # the strict comparisons are the op-level tests above and the reference fixtures.
    # (per tensor: a loose bound -- a 64-element bias can move by a third in f16; over all gradients together: a tighter one)
    ltol, gtol, atol = (1e-4, 5e-2, 2e-2) if dt == "f32" else (1e-2, 0.5, 0.15)
    assert abs(float(outs["11"]["loss"]) - float(outs["00"]["loss"])) <= ltol * abs(float(outs["00"]["loss"]))
    assert abs(float(outs["10c"]["loss"]) - float(outs["00"]["loss"])) <= ltol * abs(float(outs["00"]["loss"]))
    for key, tol, all_tol in (("10", 1e-3 if dt == "f32" else 5e-2, 1e-3 if dt == "f32" else 5e-2), ("10c", gtol, atol), ("11", gtol, atol)):
        num = den = 0.0
        for k, g0 in outs["00"]["grads"].items():
            g1 = outs[key]["grads"][k]
            d2, n2 = (g1 - g0).double().pow(2).sum().item(), g0.double().pow(2).sum().item()
            # each tensor relative to its own norm, and all of them relative to the norm of the whole gradient
            assert d2 ** 0.5 <= tol * max(n2 ** 0.5, 1e-12), (key, k, (d2 / max(n2, 1e-24)) ** 0.5)
            num, den = num + d2, den + n2
        print(f"{dt} {key}: all gradients rel L2 {(num / den) ** 0.5:.3e}")
        assert (num / den) ** 0.5 <= all_tol, (key, (num / den) ** 0.5)


# ---- round 4: patch-organised element-wise passes (csrc/sparse_elem.hip) and the batched list builders ------------------------------
# (B, f, H, C): patches of 4, 2, 4, 16, 8, 2, 1, 32, 4, 2, 1 pixels; patch maps whose side is not a power of two; several small
# patches side by side per work item
_CELL_CASES = [(2, 4, 16, 32), (3, 8, 16, 64), (1, 8, 32, 16), (2, 4, 64, 64), (2, 8, 64, 128), (1, 16, 32, 1024), (2, 4, 4, 256), (3, 2, 64, 8),
               (2, 6, 24, 64), (1, 14, 28, 128), (2, 32, 32, 1024), (3, 12, 12, 512),
               # round 5: the geometry BASELINE config 5 is quoted on -- a two-image sub-batch of a 512 x 512
               # input (32 x 32 patch map, mask 0.75) at levels 1 and 2 of the sparse encoder
               (2, 32, 512, 64), (2, 32, 256, 128)]


def _frame(act, H):
    """(B, H, H) bool maps: pixels of active patches, and the one-pixel border frame of masked patches."""
    B, f = act.shape[0], act.shape[-1]
    s = H // f
    up = act.bool().repeat_interleave(s, 1).repeat_interleave(s, 2)
    yy = torch.arange(H, device=act.device) % s
    edge = (yy == 0) | (yy == s - 1)
    border = edge.view(1, H, 1) | edge.view(1, 1, H)
    return up, (~up) & border


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("case", _CELL_CASES)
def test_cells_bn_bwd_apply_and_select_bit_identical_to_pixel_form(ops, dt, case):
    """cmu_bn_bwd_apply_cells / cmu_mask_select_cells against the pixel-organised masked kernels: the same bits everywhere (full-zero
    form); ring form: the same bits in active patches, zeros on the border frame of masked patches, their interior untouched.
    Strided operands (a channel slice of a wider buffer) included."""
    B, f, H, C = case
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    if C % (16 // torch.empty(0, dtype=tdt).element_size()) != 0:
        pytest.skip("C below one 16-byte chunk")
    g = torch.Generator().manual_seed(sum(case))
    act = _active(B, f, max(1, f * f // 4), seed=sum(case)).cuda()
    y = ops.Act(torch.randn(B, H, H, 2 * C, generator=g).to(tdt).cuda(), C, C, (torch.rand(C, generator=g) + 0.5).cuda(),
                (torch.randn(C, generator=g) * 0.3).cuda(), 0)
    assert ops.cells_supported(y, act)
    dA = ops.Act(torch.randn(B, H, H, C, generator=g).to(tdt).cuda())
    mean, invstd = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    coef = (torch.randn(2, C, generator=g) * 0.01).cuda()
    up, frame = _frame(act, H)
    ref = ops.Act(torch.full((B, H, H, C), 7.0, dtype=tdt, device="cuda"))
    ops.bn_bwd_apply_masked(dA, y, mean, invstd, coef, ref, act, cells=False)
    got = ops.Act(torch.full((B, H, H, C), 7.0, dtype=tdt, device="cuda"))
    ops.bn_bwd_apply_masked(dA, y, mean, invstd, coef, got, act)
    assert torch.equal(ref.buf.view(torch.uint8), got.buf.view(torch.uint8))
    ring = ops.Act(torch.full((B, H, H, C), 7.0, dtype=tdt, device="cuda"))
    ops.bn_bwd_apply_masked(dA, y, mean, invstd, coef, ring, act, ring=True)
    assert torch.equal(ring.buf[up], ref.buf[up]) and bool((ring.buf[frame] == 0).all()) and bool((ring.buf[~up & ~frame] == 7.0).all())
    # select: BN + ReLU in active patches, zeros elsewhere; written into the right half of a wider buffer
    for relu, tr in ((True, True), (False, True), (False, False)):
        ref = ops.Act(torch.full((B, H, H, 2 * C), 7.0, dtype=tdt, device="cuda"), C, C)
        got = ops.Act(torch.full((B, H, H, 2 * C), 7.0, dtype=tdt, device="cuda"), C, C)
        ring = ops.Act(torch.full((B, H, H, 2 * C), 7.0, dtype=tdt, device="cuda"), C, C)
        ops.mask_select(y, act, ref, relu=relu, use_transform=tr, cells=False)
        ops.mask_select(y, act, got, relu=relu, use_transform=tr)
        ops.mask_select(y, act, ring, relu=relu, use_transform=tr, ring=True)
        assert torch.equal(ref.buf.view(torch.uint8), got.buf.view(torch.uint8))
        r = ring.buf[..., C:]
        assert torch.equal(r[up], ref.buf[..., C:][up]) and bool((r[frame] == 0).all()) and bool((r[~up & ~frame] == 7.0).all())
        assert bool((ring.buf[..., :C] == 7.0).all())
    # densify: the per-channel fill vector (mask tokens) at masked positions
    tok = torch.randn(C, generator=g).cuda()
    ref = ops.Act(torch.full((B, H, H, 2 * C), 7.0, dtype=tdt, device="cuda"), C, C)
    got = ops.Act(torch.full((B, H, H, 2 * C), 7.0, dtype=tdt, device="cuda"), C, C)
    ops.mask_select(y, act, ref, relu=True, fill=tok, cells=False)
    ops.mask_select(y, act, got, relu=True, fill=tok)
    assert torch.equal(ref.buf.view(torch.uint8), got.buf.view(torch.uint8))


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("case", [c for c in _CELL_CASES if c[2] // c[1] >= 2])
def test_cells_maxpool_bwd_bit_identical_to_pixel_form(ops, dt, case):
    B, f, H, C = case
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    if C % (16 // torch.empty(0, dtype=tdt).element_size()) != 0:
        pytest.skip("C below one 16-byte chunk")
    g = torch.Generator().manual_seed(sum(case) + 1)
    act = _active(B, f, max(1, f * f // 4), seed=sum(case)).cuda()
    y = ops.Act(torch.randn(B, H, H, C, generator=g).to(tdt).cuda(), 0, C, (torch.rand(C, generator=g) + 0.5).cuda(),
                (torch.randn(C, generator=g) * 0.3).cuda(), 0)
    dP = ops.Act(torch.randn(B, H // 2, H // 2, C, generator=g).to(tdt).cuda())
    dS = ops.Act(torch.randn(B, H, H, 2 * C, generator=g).to(tdt).cuda(), C, C)
    for skip in (dS, None):
        ref = ops.Act(torch.full((B, H, H, C), 7.0, dtype=tdt, device="cuda"))
        got = ops.Act(torch.full((B, H, H, C), 7.0, dtype=tdt, device="cuda"))
        ops.maxpool_bwd_masked(dP, skip, y, ref, act, cells=False)
        ops.maxpool_bwd_masked(dP, skip, y, got, act)
        assert torch.equal(ref.buf.view(torch.uint8), got.buf.view(torch.uint8))


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("case", _CELL_CASES)
def test_cells_channel_sum_vs_float64(ops, dt, case):
    """The mask-token gradient sum (masked patches) and its complement against float64; bitwise reproducible."""
    B, f, H, C = case
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    if C % (16 // torch.empty(0, dtype=tdt).element_size()) != 0:
        pytest.skip("C below one 16-byte chunk")
    g = torch.Generator().manual_seed(sum(case) + 2)
    act = _active(B, f, max(1, f * f // 4), seed=sum(case)).cuda()
    x = ops.Act(torch.randn(B, H, H, 2 * C, generator=g).to(tdt).cuda(), C, C)
    up, _ = _frame(act, H)
    xs = x.buf[..., C:].double()
    for invert in (False, True):
        sel = ~up if invert else up
        ref = (xs * sel.unsqueeze(-1)).sum((0, 1, 2))
        out, out2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        ops.cells_channel_sum(x, act, out, invert=invert)
        ops.cells_channel_sum(x, act, out2, invert=invert)
        assert torch.equal(out, out2)
        scale = float((xs.abs() * sel.unsqueeze(-1)).sum((0, 1, 2)).max()) + 1e-6
        assert (out.double() - ref).abs().max().item() <= 2e-6 * scale


def test_build_lists_equals_one_list_per_launch(ops):
    """ops.build_lists: every tile list in one launch and every pixel list in a second one -- the same lists, element for element, as
    cmu_sparse_tile_list / cmu_sparse_pixel_list built one by one; more lists than one call holds included."""
    B, f = 3, 8
    act = _active(B, f, 19, seed=3).cuda()
    specs = [(256, 16, 32), (256, 16, 16), (128, 8, 16), (64, 16, 32), (32, 4, 4), (256, 8, 16), (128, 16, 32), (64, 8, 16), (16, 2, 2),
             (8, 1, 1), (512, 16, 32), (512, 16, 16), (512, 8, 16), (1024, 16, 32)]
    tls = [ops.TileList(act, H, H, th, tw, defer=True) for H, th, tw in specs]
    pls = [ops.PixelList(act, H, H, max_rows=19 * B * (H // f) ** 2, defer=True) for H in (256, 128, 64, 32, 16)]
    cells = ops.build_lists(act, tls, pls)
    assert int(cells.count.item()) == 19 * B
    for t, (H, th, tw) in zip(tls, specs):
        one = ops.TileList(act, H, H, th, tw)
        n = int(one.count.item())
        assert int(t.count.item()) == n and torch.equal(t.list[:n], one.list[:n])
    for p, H in zip(pls, (256, 128, 64, 32, 16)):
        one = ops.PixelList(act, H, H, max_rows=19 * B * (H // f) ** 2)
        assert int(p.count.item()) == int(one.count.item()) and torch.equal(p.rows, one.rows)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("case", [(2, 4, 64, 64), (3, 2, 64, 16), (1, 8, 128, 64), (2, 4, 128, 32),       # (B, f, H, Cout): patches of 16 / 32 px
                                  (2, 32, 512, 64)])      # the first layer of config 5: 512 x 512, 32 x 32 patch map, two images
def test_first_layer_over_a_tile_list(ops, dt, case):
    """cmu_conv3x3_c1_fwd_tiles / cmu_conv3x3_c1_wgrad_bn_tiles (the sparse encoder's one-channel first layer over its 16 x 16 tile list):
    listed tiles carry the dense launch's bits, the others stay untouched; the slab sums are the statistics over the active pixels
    (against float64); the weight gradient equals the dense fused pass on a gradient that vanishes outside the active patches, in
    the reading and the recomputing form."""
    from cmunet_amd import _lib
    B, f, H, C = case
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator().manual_seed(sum(case))
    act = _active(B, f, max(1, f * f // 4), seed=sum(case)).cuda()
    n_cells = int(act.sum())
    ps = H // f
    up, _ = _frame(act, H)
    inv_pix = (~up).to(torch.uint8).contiguous()
    x = torch.randn(B, H, H, generator=g).cuda()
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).cuda()
    dense = ops.new_act(B, H, H, C, dt, "cuda")
    ops.conv3x3_c1_fwd(x, w, dense, None, inv_pix, True)
    tl = ops.TileList(act, H, H, 16, 16)
    mx = n_cells * (ps // 16) ** 2
    assert int(tl.count.item()) == mx
    out = ops.Act(torch.full((B, H, H, C), 7.0, dtype=tdt, device="cuda"))
    slab = ops.conv3x3_c1_fwd_tiles(x, w, out, tl, mx, inv_pix, True)
    assert torch.equal(out.buf[up], dense.buf[up]) and bool((out.buf[~up] == 7.0).all())
    ref = torch.nn.functional.conv2d((x * up).double().cpu().unsqueeze(1), w.double().cpu(), padding=1).permute(0, 2, 3, 1)[up.cpu()]
    s = slab.double().sum(0).cpu()
    assert (s[0] - ref.sum(0)).abs().max().item() <= 1e-5 * ref.abs().sum(0).max().item()
    assert (s[1] - (ref * ref).sum(0)).abs().max().item() <= 1e-5 * (ref * ref).sum(0).max().item()
    # weight gradient with the BatchNorm backward applied on the fly
    sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    mean, invstd = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    coef = (torch.randn(2, C, generator=g) * 0.01).cuda()
    dA = ops.Act((torch.randn(B, H, H, C, generator=g).cuda() * up.unsqueeze(-1)).to(tdt))
    ws = torch.empty(_lib.lib().cmu_conv3x3_c1_wgrad_ws_bytes(B, H, H, C), dtype=torch.uint8, device="cuda")
    yt = out.with_transform(sc, sh, 0)
    # reference: the masked apply pass (zeros outside the active patches) followed by the plain first-layer weight gradient
    dY = ops.new_act(B, H, H, C, dt, "cuda")
    ops.bn_bwd_apply_masked(dA, dense.with_transform(sc, sh, 0), mean, invstd, coef, dY, act)
    dW0 = torch.empty(C, 1, 3, 3, device="cuda")
    ops.conv3x3_c1_wgrad(x, dY, dW0, ws, inv_pix, True)
    for wf in (None, w):
        dW = torch.full((C, 1, 3, 3), 7.0, device="cuda")
        ops.conv3x3_c1_wgrad_bn_tiles(x, dA, yt, sc, sh, mean, invstd, coef, dW, ws, tl, mx, inv_pix, True, w=wf)
        # (the fused pass keeps dY in fp32 where the two-pass form rounded it to the storage type)
        tol = {"f32": 1e-5, "f16": 2e-3, "bf16": 1.6e-2}[dt]
        assert (dW - dW0).abs().max().item() <= tol * dW0.abs().max().item(), (wf is None, (dW - dW0).abs().max().item(), dW0.abs().max().item())


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("case", _CELL_CASES)
def test_cells_statistics_vs_float64_and_pixel_form(ops, dt, case):
    """cmu_cells_channel_stats / cmu_bn_bwd_reduce_cells (sparse BatchNorm statistics and backward sums over the active patches, walked
    as groups of patch rows) against float64 on the stored values and against the pixel-organised passes; bitwise reproducible."""
    from cmunet_amd import _lib
    B, f, H, C = case
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    if C % (16 // torch.empty(0, dtype=tdt).element_size()) != 0:
        pytest.skip("C below one 16-byte chunk")
    g = torch.Generator().manual_seed(sum(case) + 5)
    act = _active(B, f, max(1, f * f // 4), seed=sum(case)).cuda()
    up, _ = _frame(act, H)
    y = ops.Act(torch.randn(B, H, H, 2 * C, generator=g).to(tdt).cuda(), C, C)
    ys = y.buf[..., C:].double()[up]
    slab = ops.cells_channel_stats(y, act)
    assert torch.equal(slab, ops.cells_channel_stats(y, act))
    s = slab.double().sum(0)
    n = max(1, ys.shape[0])
    assert (s[0] - ys.sum(0)).abs().max().item() <= 2e-6 * n ** 0.5 * max(1.0, float(ys.abs().max()))
    assert (s[1] - (ys * ys).sum(0)).abs().max().item() <= 1e-5 * float((ys * ys).sum(0).max())
    old = ops.masked_channel_stats(y, act).double().sum(0)
    assert (s - old).abs().max().item() <= 1e-5 * old.abs().max().item()
    sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    mean, invstd = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    yt = y.with_transform(sc, sh, 0)
    dA = ops.Act(torch.randn(B, H, H, C, generator=g).to(tdt).cuda())
    count = int(act.sum()) * (H // f) ** 2
    outs = []
    for cells in (True, True, False):
        dg, db, coef = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(2, C, device="cuda")
        ws = torch.empty(_lib.lib().cmu_bn_bwd_ws_bytes(C), dtype=torch.uint8, device="cuda")
        if cells:
            ops.bn_bwd_reduce_cells(dA, yt, mean, invstd, dg, db, coef, act, count, ws)
        else:
            ops.bn_bwd_reduce_masked(dA, yt, mean, invstd, dg, db, coef, act, count, ws)
        outs.append((dg.clone(), db.clone(), coef.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    for u, v in zip(outs[0], outs[2]):
        assert (u - v).abs().max().item() <= 2e-5 * max(v.abs().max().item(), 1e-6)
    # float64 on the stored values
    dz = torch.where(ys * sc.double().cpu().cuda() + sh.double().cuda() > 0, dA.buf.double()[up], torch.zeros((), dtype=torch.float64, device="cuda"))
    xh = (ys - mean.double()) * invstd.double()
    assert (outs[0][1].double() - dz.sum(0)).abs().max().item() <= 1e-5 * max(1.0, float(dz.abs().sum(0).max()))
    assert (outs[0][0].double() - (dz * xh).sum(0)).abs().max().item() <= 1e-5 * max(1.0, float((dz * xh).abs().sum(0).max()))


def test_spark_mask_edited_in_place_is_recounted(ops):
    """Advisor (round 4): ``SparK.mask()`` hands the active-patch count to the step on the mask tensor; an in-place edit of that mask used
    to leave the count stale (sparse BatchNorm counts, list capacities).  The count now carries the tensor's version: a mask from
    ``mask()`` with one more patch forced active gives bit for bit the loss and gradients of a fresh tensor with the same content."""
    from cmunet_amd import spark as S

    def run(edit_in_place):
        torch.manual_seed(5)
        enc = S.build_sparse_encoder("unet_sparse", input_size=64, base_ch=16, depth=5, dtype="f32")
        model = S.SparK(enc, S.UnetDecoder(base_ch=16, depth=5, dtype="f32"), mask_ratio=0.75, densify_norm="", dtype="f32").cuda().train()
        g = torch.Generator().manual_seed(11)
        x = torch.randn(4, 1, 64, 64, generator=g).cuda()
        active = model.mask(4, "cuda", g)
        first_off = (~active[0, 0]).nonzero()[0]
        if edit_in_place:
            active[0, 0, first_off[0], first_off[1]] = True                  # same tensor object: the attribute survives, its version does not
            assert hasattr(active, "_cmu_n_active") and active._cmu_n_active[0] != active._version
        else:
            fresh = active.clone()
            fresh[0, 0, first_off[0], first_off[1]] = True
            active = fresh                                                   # no attribute: counted from the tensor
            assert not hasattr(active, "_cmu_n_active")
        loss = model(x, active_b1ff=active)
        loss.backward()
        return loss.detach().cpu(), {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}
    l0, g0 = run(False)
    l1, g1 = run(True)
    assert bool(torch.isfinite(l1).all()) and float(l0) == float(l1)
    assert g0.keys() == g1.keys() and all(torch.equal(g0[k], g1[k]) for k in g0)
