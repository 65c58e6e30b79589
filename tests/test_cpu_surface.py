"""CPU suite: host-side surface kept from the reference -- dataset item contracts, checkpoint key maps of
load_model (train.py:240-308), loss naming algebra (metrics.py:9-82), CLI flags."""
import numpy as np
import pytest
import torch

from oracle import unet as OU


def test_segmentation_dataset_contract(tmp_path):
    from cmunet_amd.dataset import SegmentationDataset, SyntheticSegmentationDataset, two_view_item
    rng = np.random.RandomState(0)
    imgs, masks = [], []
    for i in range(3):
        np.save(tmp_path / f"i{i}.npy", rng.standard_normal((300, 280)).astype(np.float32))
        np.save(tmp_path / f"m{i}.npy", (rng.rand(300, 280) > 0.9).astype(np.uint8))
        imgs.append(str(tmp_path / f"i{i}.npy")); masks.append(str(tmp_path / f"m{i}.npy"))
    calls = []

    def aug(image, mask):                                   # albumentations-style protocol
        calls.append(1)
        return {"image": image[:, ::-1].copy(), "mask": mask[:, ::-1].copy()}
    ds = SegmentationDataset(imgs, masks, augmentation=aug, class_values=[0, 1])
    x, y = ds[1]
    assert len(ds) == 3 and calls
    assert x.shape == (256, 256) and x.dtype == np.float32                 # dataset.py:46,53
    assert y.shape == (2, 256, 256) and y.dtype == np.float64              # dataset.py:47-48 (one-hot, float64: A-5)
    assert np.array_equal(y.sum(0), np.ones((256, 256)))
    x3, _ = SegmentationDataset(imgs, masks, last_axis=True)[0]
    assert x3.shape == (1, 256, 256)
    s = SyntheticSegmentationDataset(n=2, size=64)
    xs, ys = s[0]
    assert xs.shape == (64, 64) and xs.dtype == np.float32 and ys.shape == (2, 64, 64) and ys.dtype == np.float64
    it = two_view_item(rng.standard_normal((256, 256)).astype(np.float32), rng)
    assert it["img"].shape == it["img_t"].shape == (224, 224) and it["img"].dtype == np.float32   # cmunet_dataset.py:60-88


def test_load_model_checkpoint_layouts(tmp_path, golden_dir):
    """Synthetic checkpoints in each of the five foreign layouts -> the keys that must land in UNet(): exactly the keys the
    REFERENCE's own load_model loaded from the same files (tests/golden/load_model_ref.npz, oracle/gen_golden.py::gen_load_model)."""
    from cmunet_amd import train as T
    ref = np.load(f"{golden_dir}/load_model_ref.npz")
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=int(ref["seed"]))
    cases = OU.checkpoint_layout_cases(sd)
    assert list(cases) == [str(n) for n in ref["layouts"]]
    keys = [str(k) for k in ref["keys"]]
    expect_decoder = {"spark.pth": True, "cmunet.pth": True, "encoder.pth": False, "moco.ckpt": False, "genesis.pt": True}
    for row, (name, ck) in zip(ref["loaded"], cases.items()):
        path = str(tmp_path / name)
        torch.save(ck, path)
        args = T.get_args(["-p", path])
        m = T.load_model(args)
        got = m.state_dict()
        assert list(got.keys()) == keys
        mine = [bool(torch.equal(got[k], sd[k])) if not k.endswith("num_batches_tracked") else True for k in keys]
        diff = [k for k, a, b in zip(keys, mine, row) if a != bool(b)]
        assert not diff, (name, diff[:6])
        w_enc = "down_conv3.double_conv.double_conv.3.weight"
        w_dec = "up_conv2.double_conv.double_conv.0.weight"
        assert torch.equal(got[w_enc], sd[w_enc]), name
        assert torch.equal(got[w_dec], sd[w_dec]) == expect_decoder[name], name
        assert not torch.equal(got["conv_last.weight"], sd["conv_last.weight"]), name      # head always dropped
        if name == "cmunet.pth":                                                            # target copy must not win
            assert got["down_conv1.double_conv.double_conv.0.weight"].abs().sum() > 0
    # deviation, on purpose: a plain encoder state dict WITHOUT a "meta" entry makes the reference raise KeyError('meta')
    # (train.py:262 indexes checkpoint["meta"] before its "encoder only" branch); here it loads as the encoder layout
    plain = {"module." + k: v for k, v in sd.items() if "down_conv" in k or k.startswith("double_conv")}
    torch.save(plain, str(tmp_path / "plain.pth"))
    got = T.load_model(T.get_args(["-p", str(tmp_path / "plain.pth")])).state_dict()
    assert torch.equal(got["down_conv3.double_conv.double_conv.3.weight"], sd["down_conv3.double_conv.double_conv.3.weight"])
    m = T.load_model(T.get_args([]))
    assert sum(p.numel() for p in m.parameters()) == 31042434


def test_export_checkpoint_round_trips(tmp_path):
    """export_checkpoint writes each foreign layout so that the reader (the reference's load_model dispatch, restated in
    remap_checkpoint) recovers exactly the tensors that layout carries."""
    from cmunet_amd import train as T
    sd = OU.make_state_dict(base_ch=16, depth=3, seed=4)
    enc_keys = [k for k in sd if k.startswith(("down_conv", "double_conv"))]
    dec_keys = [k for k in sd if k.startswith("up_conv")]
    for layout, name, has_dec in (("spark", "a.pth", True), ("cmunet", "b.pth", True), ("encoder", "c.pth", False),
                                  ("moco", "d.ckpt", False), ("genesis", "e.pt", True)):
        path = T.export_checkpoint(sd, str(tmp_path / name), layout, epoch=7, optimizer_state={"lr": 0.1})
        ck = torch.load(path, map_location="cpu", weights_only=False)
        got, label = T.remap_checkpoint(ck, path)
        assert label == layout
        for k in enc_keys:
            assert torch.equal(got[k], sd[k]), (layout, k)
        for k in dec_keys:
            assert (k in got and torch.equal(got[k], sd[k])) == has_dec, (layout, k)
        assert "conv_last.weight" not in got
    ck = torch.load(str(tmp_path / "a.pth"), weights_only=False)
    assert {"args", "input_size", "arch", "epoch", "performance_desc", "module", "optimizer", "is_pretrain"} <= set(ck) and ck["epoch"] == 7
    assert torch.load(str(tmp_path / "b.pth"), weights_only=False)["meta"]["mmengine_version"]
    pre = T.export_checkpoint(sd, str(tmp_path / "f.pth"), "encoder", extra={"prefix": "module."})
    assert all(k.startswith("module.") for k in torch.load(pre, weights_only=False))
    with pytest.raises(ValueError):
        T.export_checkpoint(sd, str(tmp_path / "x.pt"), "spark")
    with pytest.raises(ValueError):
        T.export_checkpoint(sd, str(tmp_path / "x.pth"), "nope")


def test_cli_flags_and_meter():
    from cmunet_amd import train as T
    a = T.get_args(["-e", "3", "-b", "2,4", "-l", "0.001", "-n", "x", "-r", "0.5"])
    assert a.epochs == [3] and a.batch_size == [2, 4] and a.lr == [0.001] and a.name == "x" and a.ratio == 0.5
    d = T.get_args([])
    assert d.epochs == [2] and d.batch_size == [16, 32] and d.lr == [0.1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6] and d.pretrained is None
    # the logs contract of Epoch.run (train.py:129-140): per key, the mean over the epoch's batches
    means = T._epoch_means([[1.0, 10.0], [2.0, 20.0], [4.0, 60.0]])
    assert abs(means[0] - 7.0 / 3) < 1e-12 and abs(means[1] - 30.0) < 1e-12


def test_loss_algebra_names(golden_dir):
    from cmunet_amd import metrics as M
    crit = M.DiceLoss(activation="softmax", threshold=0.5, ignore_channels=[0]) + M.CrossEntropyLoss()
    fx = np.load(f"{golden_dir}/unet_small.npz")
    assert crit.__name__ == str(fx["crit_name"]) == "dice_loss + cross_entropy_loss"
    assert M.IoU(threshold=0.5, activation="softmax", ignore_channels=[0]).__name__ == "iou_loss"
    assert (2 * M.CrossEntropyLoss()).__name__ == "2 * cross_entropy_loss"
    with pytest.raises(NotImplementedError):
        M.DiceLoss(activation="sigmoid", threshold=0.5, ignore_channels=[0])
    with pytest.raises(ValueError):
        M.CrossEntropyLoss() + 3
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        crit(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4, dtype=torch.float64))


def test_lr_schedules_vs_reference(golden_dir):
    """SparK's per-iteration learning-rate / weight-decay annealing against a table the REFERENCE's own lr_wd_annealing produced
    (Spark/utils/lr_control.py:11-29, tests/golden/optim_traces.npz), and MoCo's cosine learning rate against
    torch.optim.lr_scheduler.CosineAnnealingLR as moco2_module.py:345-348 configures it."""
    from cmunet_amd.pretrain import moco_cosine_lr, spark_lr_wd
    d = np.load(f"{golden_dir}/optim_traces.npz")
    for pk, wd, wde, wp, mx, it, lr, cur_wd in d["sched_table"]:
        got = spark_lr_wd(float(pk), float(wd), float(wde), int(it), float(wp), int(mx))
        assert abs(got[0] - lr) <= 1e-15 + 1e-12 * abs(lr) and abs(got[1] - cur_wd) <= 1e-15 + 1e-12 * abs(cur_wd), (pk, wp, mx, it)
    assert len(d["sched_table"]) >= 15
    prm = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([prm], lr=0.03, momentum=0.9, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 7)
    for ep in range(8):
        assert abs(moco_cosine_lr(0.03, ep, 7) - opt.param_groups[0]["lr"]) < 1e-12, ep
        opt.step()
        sch.step()


def test_kfold_indices_are_sklearns_split():
    """train.kfold_indices == sklearn.model_selection.KFold(n_splits=3, shuffle=True, random_state=42).split (Finetuning/train.py:326,330:
    the folds of main_finetuning), for the 18-image split of BASELINE config 4 and for sizes that do not divide by three."""
    sk = pytest.importorskip("sklearn.model_selection")
    from cmunet_amd import train as T
    for n in (18, 20, 7, 3):
        ours = T.kfold_indices(n, 3, 42)
        ref = list(sk.KFold(n_splits=3, shuffle=True, random_state=42).split(list(range(n))))
        assert len(ours) == 3
        for (a_tr, a_te), (b_tr, b_te) in zip(ours, ref):
            assert np.array_equal(a_tr, b_tr) and np.array_equal(a_te, b_te), n
    assert sorted(int(i) for _, te in T.kfold_indices(18) for i in te) == list(range(18))


def test_find_best_epochs_contract():
    """utils.py:4-60: the best epoch by the training criterion, hyper-parameters attached; the first epoch may be the best one (the
    reference leaves ``best_result`` unbound there and raises UnboundLocalError)."""
    from cmunet_amd import train as T
    logs = [{"dice_loss + cross_entropy_loss": 1.0, "dice_loss": 0.6, "iou_loss": 0.7},
            {"dice_loss + cross_entropy_loss": 0.8, "dice_loss": 0.7, "iou_loss": 0.6},
            {"dice_loss + cross_entropy_loss": 0.9, "dice_loss": 0.5, "iou_loss": 0.5}]
    r = T.find_best_epochs(logs, 3, 1e-3, 6, 12.5)
    assert (r["epochs"], r["lr"], r["batch_size"], r["runtime"]) == (3, 1e-3, 6, 12.5)
    assert r["dice_loss"] == 0.7 and r["iou_loss"] == 0.6                       # epoch 1: smallest criterion
    assert T.find_best_epochs(logs[:1], 1, 1e-3, 6, 1.0)["dice_loss"] == 0.6    # a single / first epoch is fine
    assert T.find_best_epochs([{"dice_loss": 0.4}, {"dice_loss": 0.3}], 2, 1e-2, 2, 0.0)["dice_loss"] == 0.3   # falls back to dice_loss


def test_train_test_split_indices_are_sklearns():
    """dataset.train_test_split_indices == sklearn.model_selection.train_test_split (what cmunet_dataset.py:30-31 calls)."""
    from sklearn.model_selection import train_test_split
    from cmunet_amd.dataset import train_test_split_indices
    for n, ts, rs in ((50, 0.2, 42), (40, 0.0125, 42), (7, 0.2, 1), (1001, 0.33, 7)):
        tr, te = train_test_split_indices(n, ts, rs)
        a, b = train_test_split(list(range(n)), test_size=ts, random_state=rs)
        assert list(tr) == a and list(te) == b, (n, ts, rs)


def test_cmunet_dataset_surface(tmp_path):
    """CMUNetDataset(data_root, data_ann, pipeline, pixel, test) on a directory of .npy images: the reference's two
    train_test_split calls (cmunet_dataset.py:28-35), the {'img','img_t'} item contract (:60-88), GaussNoise always applied (A-11),
    and a custom pipeline of callables; a config-dict pipeline (mmcv / mmengine) raises."""
    from sklearn.model_selection import train_test_split
    from cmunet_amd.dataset import CMUNetDataset
    rng = np.random.RandomState(0)
    for i in range(30):
        np.save(tmp_path / f"im_{i:03d}.npy", rng.standard_normal((300, 280)).astype(np.float32))
    ds = CMUNetDataset(str(tmp_path), "unused.json", None, pixel=31, seed=5)
    paths = [str(tmp_path / f) for f in sorted(p.name for p in tmp_path.iterdir())]
    X_train, X_test, y_train, _ = train_test_split(paths, paths.copy(), test_size=0.2, random_state=42)
    inner, _, _, _ = train_test_split(X_train, y_train, test_size=0.0125, random_state=42)
    assert ds.image_paths == inner and ds.test == X_test and len(ds) == len(inner) == 23
    it = ds[0]
    assert set(it) == {"img", "img_t"} and it["img"].shape == (224, 224) and it["img_t"].shape == (224, 224)
    assert it["img"].dtype == torch.float32 and it["img_t"].dtype == torch.float32
    assert not torch.equal(it["img"], it["img_t"]) and bool(torch.isfinite(it["img_t"]).all())
    # the base pipeline is the first two entries, the final one the rest -- both views go through the final one
    seen = []
    base = [lambda r: {"img": np.asarray(r["img"], dtype=np.float32) * 0 + 2.0}, lambda r: r]
    final = [lambda r: (seen.append(r["img"].shape), {"img": r["img"] + 1})[1]]
    ds2 = CMUNetDataset(str(tmp_path), None, base + final, pixel=0, seed=1)
    it2 = ds2[3]
    assert seen == [(224, 224), (224, 224)] and float(it2["img"].min()) == 3.0 and it2["img"].max() == 3.0
    assert abs(float(it2["img_t"].mean()) - 3.0) < 0.05 and float(it2["img_t"].std()) > 0.1      # 2 + N(0, (2 / 10)^2) + 1
    with pytest.raises(TypeError):
        CMUNetDataset(str(tmp_path), None, [dict(type="RandomResizedCrop", scale=256)])


def test_moco_dataset_surface(tmp_path):
    """MoCoDataset(data_path, tau_g): list of .npy paths, two global transforms on a (1, 256, 256) tensor, item ((c0, c1), 0)
    (moco_data_set.py:11-37)."""
    from PIL import Image
    from cmunet_amd.dataset import MoCoDataset
    rng = np.random.RandomState(1)
    paths = []
    for i in range(3):
        paths.append(str(tmp_path / f"m{i}.npy"))
        np.save(paths[-1], rng.standard_normal((180, 200)).astype(np.float32))
    ds = MoCoDataset(paths, [lambda t: t[:, :224, :224], lambda t: t.flip(-1)[:, 16:240, 16:240]])
    assert len(ds) == 3 and str(ds) == "LoGoDataset with 3 images"
    (c0, c1), label = ds[1]
    ref = torch.from_numpy(np.array(Image.fromarray(np.load(paths[1])).resize((256, 256), resample=Image.BICUBIC)))[None]
    assert label == 0 and torch.equal(c0, ref[:, :224, :224]) and torch.equal(c1, ref.flip(-1)[:, 16:240, 16:240])


def test_cmunet_adamw_decay_groups_are_the_reference_config():
    """cmunet_config.py:84-91 under mmengine's DefaultOptimWrapperConstructor: decay_mult 0 iff the qualified parameter name CONTAINS
    'ln' / 'bias' / 'pos_embed' / 'mask_token' / 'cls_token'; no norm_decay_mult -- the BatchNorm weights decay.  Restated on the names of
    the 217.8 M-parameter model (built on the meta device: names and shapes only)."""
    import torch
    from cmunet_amd import cmunet as C
    from cmunet_amd.optim import CMUNET_NO_DECAY_KEYS, cmunet_paramwise_decay
    with torch.device("meta"):
        m = C.CM_UNet(**C.cmunet_config(img_size=224))
    named = list(m.named_parameters())
    assert len(named) == 179 and sum(p.numel() for _, p in named) == 217_812_228
    assert CMUNET_NO_DECAY_KEYS == ("ln", "bias", "pos_embed", "mask_token", "cls_token")

    def mmengine_rule(name):          # the constructor's loop: first key (longest first) that is a substring of the name wins
        for key in sorted(sorted(CMUNET_NO_DECAY_KEYS), key=len, reverse=True):
            if key in name:
                return 0.0            # decay_mult of every key in the shipped config
        return 1.0
    decays = [n for n, p in named if cmunet_paramwise_decay(n, p)]
    assert decays == [n for n, _ in named if mmengine_rule(n) == 1.0]
    exempt = [n for n, _ in named if n not in decays]
    assert all(n.endswith(".bias") or n.endswith("_bias") or "bias" in n for n in exempt)
    assert not any("ln" in n for n, _ in named)                        # no name of this model hits the 'ln' key
    bn_weights = [n for n, p in named if p.dim() == 1 and "bias" not in n]
    assert len(bn_weights) == 39 and all(n in decays for n in bn_weights)   # 36 BatchNorm2d + 3 neck BatchNorm1d weights: all decay
    assert len(decays) == 91 and len(exempt) == 88
    # what the trainers use
    import inspect
    from cmunet_amd import pretrain as P
    assert "decay_filter=cmunet_paramwise_decay" in inspect.getsource(P.MaskedReconPretrainer.__init__)
    assert "decay_filter=cmunet_paramwise_decay" in inspect.getsource(P.JointPretrainer.__init__)
