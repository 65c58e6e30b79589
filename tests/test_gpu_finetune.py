"""BASELINE config 1 on the HIP path: UNet-small (base 16, depth 3) supervised finetune on 8 synthetic 256x256
images + masks, bs 2, Adam lr 1e-3, 2 epochs, seed 42 -- driven through the drop-in Epoch classes and fused
losses, compared with the oracle running the restated reference loop (train.py:163-169) on the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_finetune_unet_small_vs_oracle(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import metrics as M, model as Mod, train as T
    from cmunet_amd.dataset import SyntheticSegmentationDataset
    from oracle import losses as OL, unet as OU
    torch.manual_seed(42)
    ds = SyntheticSegmentationDataset(n=8, size=256, seed=42)
    train_idx, valid_idx = list(range(6)), [6, 7]
    mk = lambda idx: [(torch.from_numpy(np.stack([ds[i][0] for i in idx[j:j + 2]])), torch.from_numpy(np.stack([ds[i][1] for i in idx[j:j + 2]])))
                      for j in range(0, len(idx), 2)]
    train_loader, valid_loader = mk(train_idx), mk(valid_idx)
    sd = OU.make_state_dict(base_ch=16, depth=3, seed=42)

    net = Mod.UNet(base_ch=16, depth=3, dtype="f32")
    net.load_state_dict(sd)
    crit = M.DiceLoss(activation="softmax", threshold=0.5, ignore_channels=[0]) + M.CrossEntropyLoss()
    mets = [M.DiceMetric(), M.IoU(threshold=0.5, activation="softmax", ignore_channels=[0])]
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    tr = T.TrainEpoch(net, loss=crit, metrics=mets, optimizer=opt, device="cuda", verbose=False)
    va = T.ValidEpoch(net, loss=crit, metrics=mets, device="cuda", verbose=False)
    got = []
    for ep in range(2):
        got.append((tr.run(train_loader), va.run(valid_loader)))

    # oracle loop
    osd = OU.clone_sd(sd, requires_grad=True)
    oopt = torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3)

    def run(loader, training):
        ls, ds_, io = [], [], []
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
            ls.append(float(l)); ds_.append(float(OL.dice_loss(lo.detach(), y))); io.append(float(OL.iou_loss(lo.detach(), y)))
        return {"dice_loss + cross_entropy_loss": np.mean(ls), "dice_loss": np.mean(ds_), "iou_loss": np.mean(io)}
    for ep in range(2):
        rt, rv = run(train_loader, True), run(valid_loader, False)
        for logs, ref in ((got[ep][0], rt), (got[ep][1], rv)):
            assert set(logs) == set(ref)
            for k in ref:
                # Dice / IoU are ratios of thresholded pixel counts: a handful of flipped pixels out of 131072
                assert abs(logs[k] - ref[k]) <= 2e-3, (ep, k, logs[k], ref[k])
    assert got[1][0]["dice_loss + cross_entropy_loss"] < got[0][0]["dice_loss + cross_entropy_loss"]    # it trains
    # torch.save(model) of the reference (train.py:212) works on the drop-in module
    import io
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    net2 = torch.load(buf, weights_only=False)
    x = train_loader[0][0].cuda()
    net.eval(); net2.eval()
    with torch.no_grad():
        assert torch.equal(net(x), net2.cuda()(x))


def test_soft_cldice_vs_reference_fixture(golden_dir):
    """Device soft-clDice (skeleton kernels + sums) against the value and the skeleton the reference's classes produced."""
    import numpy as np
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import _lib, metrics as PM, ops
    d = np.load(f"{golden_dir}/cldice.npz")
    logits, y1h = torch.from_numpy(d["logits"]).cuda(), torch.from_numpy(d["y1h"]).cuda()
    m = PM.soft_cldice(threshold=0.5, activation="softmax", ignore_channels=[0])
    assert m.__name__ == str(d["metric_name"])
    got = float(m(logits, y1h))
    assert abs(got - float(d["cldice"])) <= 1e-5, (got, float(d["cldice"]))
    # the skeleton itself
    B, _, H, W = logits.shape
    yp = torch.empty(B, H, W, device="cuda")
    _lib.call("cmu_softmax2_threshold", ops._p(logits.float().contiguous()), 0.5, ops._p(yp), B, H, W, ops._stream())
    ws = torch.empty(_lib.lib().cmu_soft_skeleton_ws_bytes(B * H * W), dtype=torch.uint8, device="cuda")
    sk = torch.empty_like(yp)
    _lib.call("cmu_soft_skeleton", ops._p(yp), ops._p(sk), B, H, W, 10, ops._p(ws), ops._stream())
    assert torch.allclose(sk.cpu(), torch.from_numpy(d["skel_pred"])[:, 0], atol=1e-6)
    with pytest.raises(NotImplementedError):
        PM.soft_cldice(activation=None)
