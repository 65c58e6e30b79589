"""Device input pipeline against the oracle (oracle/augment.py, itself pinned bit-for-bit to Pillow in
tests/test_cpu_oracle.py): the HIP kernels must reproduce Pillow's bicubic resize and the reference's ShiftPixel +
GaussNoise arithmetic EXACTLY (float32 outputs compared as bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import ops as o
    return o


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("shape,out", [((3, 512, 512), (256, 256)), ((2, 300, 200), (256, 256)), ((2, 51, 77), (256, 256)),
                                       ((2, 256, 256), (256, 256)), ((1, 100, 100), (64, 48)), ((2, 37, 512), (256, 256)),
                                       ((1, 1024, 640), (256, 256)),
                                       # degenerate sources: one pixel, a 2 x 3 image, one row, output of one pixel
                                       ((2, 1, 1), (8, 8)), ((1, 2, 3), (5, 7)), ((1, 1, 40), (16, 16)), ((1, 37, 29), (1, 1))])
def test_resize_bicubic_whole_image_matches_pillow_convention(ops, shape, out):
    from oracle.augment import resize_bicubic
    rng = np.random.RandomState(1)
    a = (rng.standard_normal(shape) * 3 + 0.5).astype(np.float32)
    got = ops.resize_bicubic(torch.from_numpy(a).cuda(), out[0], out[1]).cpu().numpy()
    ref = np.stack([resize_bicubic(x, out[0], out[1]) for x in a])
    assert np.array_equal(_bits(got), _bits(ref)), float(np.abs(got - ref).max())


def test_resize_bicubic_crop_windows_and_flip(ops):
    """RandomResizedCrop + RandomFlip: ragged windows (1-pixel, full-size, thin, equal-to-output) per sample."""
    from oracle.augment import resize_bicubic_crop
    rng = np.random.RandomState(2)
    B, S = 8, 256
    a = rng.standard_normal((B, S, S)).astype(np.float32)
    boxes = np.array([[0, 0, 256, 256], [10, 20, 100, 57], [255, 255, 1, 1], [0, 100, 256, 3], [17, 0, 51, 256], [3, 5, 200, 250],
                      [0, 0, 256, 128], [128, 0, 128, 256]], np.int32)
    flips = np.array([0, 1, 0, 1, 1, 0, 1, 0], np.uint8)
    got = ops.resize_bicubic(torch.from_numpy(a).cuda(), 256, 256, torch.from_numpy(boxes), torch.from_numpy(flips)).cpu().numpy()
    ref = resize_bicubic_crop(a, boxes, flips, 256, 256)
    assert np.array_equal(_bits(got), _bits(ref)), float(np.abs(got - ref).max())
    with pytest.raises(ValueError):
        ops.resize_bicubic(torch.from_numpy(a).cuda(), 256, 256, torch.tensor([[0, 0, 257, 10]] * B))


def test_two_view_with_explicit_noise_is_exact(ops):
    from oracle.augment import two_view
    rng = np.random.RandomState(3)
    B, S, out = 5, 256, 224
    a = (rng.standard_normal((B, S, S)) * 2).astype(np.float32)
    a[3] = -np.abs(a[3]) - 1.0                              # all-negative sample: sigma = max/10 is negative, as numpy has it
    shifts = np.array([[0, 0], [31, 31], [7, 19], [32, 0], [1, 32]], np.int32)
    noise = rng.standard_normal((B, out, out))
    img, img_t = ops.two_view(torch.from_numpy(a).cuda(), torch.from_numpy(shifts), out, noise=torch.from_numpy(noise).cuda())
    r_img, r_img_t = two_view(a, shifts, noise, out)
    assert np.array_equal(_bits(img.cpu().numpy()), _bits(r_img))
    assert np.array_equal(_bits(img_t.cpu().numpy()), _bits(r_img_t))
    with pytest.raises(ValueError):
        ops.two_view(torch.from_numpy(a).cuda(), torch.tensor([[33, 0]] * B), out)      # dy + 224 > 256 (processing.py:112)


def test_philox_generator_matches_oracle(ops):
    """Counter-based draws: the integer stream is Philox4x32-10 (known answers in the CPU suite); the normals agree with the
    numpy restatement to rounding of log / cos / sqrt (1e-12), at an offset that crosses the 32-bit counter word."""
    from oracle.augment import philox_normal
    for off, seed in ((0, 0), (2 ** 32 - 100, 0x1234_5678_9ABC_DEF0)):
        got = ops.philox_normal(5000, off, seed).cpu().numpy()
        ref = philox_normal(5000, off, seed)
        assert np.abs(got - ref).max() < 1e-12
    z = ops.philox_normal(1 << 20, 0, 7).cpu().numpy()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3


def test_two_view_in_kernel_noise_equals_explicit_draws(ops):
    rng = np.random.RandomState(4)
    B, S, out = 3, 256, 224
    a = rng.standard_normal((B, S, S)).astype(np.float32)
    shifts = torch.tensor([[3, 4], [0, 31], [31, 0]])
    x = torch.from_numpy(a).cuda()
    z = ops.philox_normal(B * out * out, 0, 99).view(B, out, out)
    _, t_explicit = ops.two_view(x, shifts, out, noise=z)
    _, t_kernel = ops.two_view(x, shifts, out, seed=99)
    assert torch.equal(t_explicit, t_kernel)


def test_device_pipeline_equals_host_pipeline_on_the_same_draws(ops):
    """The whole device pipeline against Pillow-convention resize + numpy crops on the host, same random scalars."""
    from cmunet_amd.dataset import DeviceTwoViewPipeline
    from oracle.augment import resize_bicubic, resize_bicubic_crop, two_view
    rng = np.random.RandomState(5)
    raw = rng.standard_normal((4, 384, 384)).astype(np.float32)
    pipe = DeviceTwoViewPipeline(seed=11)
    boxes, flips, shifts = pipe.draw(4)
    noise = rng.standard_normal((4, 224, 224))
    got = pipe(torch.from_numpy(raw).cuda(), params=(boxes, flips, shifts), noise=torch.from_numpy(noise).cuda())
    base = np.stack([resize_bicubic(x, 256, 256) for x in raw])
    crop = resize_bicubic_crop(base, boxes, flips, 256, 256)
    r_img, r_img_t = two_view(crop, shifts, noise, 224)
    assert np.array_equal(_bits(got['img'].cpu().numpy()), _bits(r_img))
    assert np.array_equal(_bits(got['img_t'].cpu().numpy()), _bits(r_img_t))
    out = pipe(torch.from_numpy(raw).cuda())                  # fully random call: shapes / dtype / finiteness
    assert out['img'].shape == (4, 224, 224) and out['img_t'].dtype == torch.float32 and bool(torch.isfinite(out['img_t']).all())


@pytest.mark.parametrize("shape,patch,ratio", [((4, 512, 512), 16, 0.6), ((3, 224, 224), 16, 0.65), ((2, 64, 96), 16, 0.75), ((2, 32, 32), 16, 0.0)])
def test_random_patch_mask_kernel(ops, shape, patch, ratio):
    """Device patch-mask generator: identical to the numpy restatement (same Philox keys, same ranking), the reference's count
    of masked patches per sample (UNet_encoder.py:113), whole patches only, fresh draws at another offset."""
    from oracle.augment import random_patch_mask
    B, H, W = shape
    n_mask = int(ratio * H * W) // (patch * patch)
    got = ops.random_patch_mask(B, H, W, patch, ratio, seed=77, offset=12345).cpu().numpy()
    ref = random_patch_mask(B, H, W, patch, n_mask, 77, 12345)
    assert np.array_equal(got, ref)
    blocks = got.reshape(B, H // patch, patch, W // patch, patch)
    assert (blocks.min(axis=(2, 4)) == blocks.max(axis=(2, 4))).all()
    assert (blocks[:, :, 0, :, 0].reshape(B, -1).sum(1) == n_mask).all()
    if n_mask:
        other = ops.random_patch_mask(B, H, W, patch, ratio, seed=77, offset=12345 + B * (H // patch) * (W // patch)).cpu().numpy()
        assert not np.array_equal(other, got)


@pytest.mark.parametrize("shape,out", [((3, 512, 512), (256, 256)), ((2, 300, 200), (256, 256)), ((2, 51, 77), (256, 256)),
                                       ((2, 256, 256), (256, 256)), ((1, 100, 100), (64, 48)), ((1, 1024, 640), (256, 256)),
                                       ((2, 1, 1), (8, 8)), ((1, 2, 3), (5, 7)), ((1, 37, 29), (1, 1))])
def test_resize_bicubic_u8_and_nearest_match_pillow_arithmetic(ops, shape, out):
    """8-bit images (PIL mode 'L': Finetuning/dataset.py:44-46 on a uint8 .npy) and the NEAREST resize of the label masks (:47):
    byte-identical to the oracle (pinned to Pillow in tests/test_cpu_oracle.py), incl. black / white checker noise whose cubic
    overshoot clips on both sides."""
    from oracle.augment import resize_bicubic_u8, resize_nearest
    rng = np.random.RandomState(hash(shape + out) % 2 ** 31)
    a = rng.randint(0, 256, shape).astype(np.uint8)
    a[0] = np.where(rng.rand(*shape[1:]) < 0.5, 0, 255)
    got = ops.resize_bicubic(torch.from_numpy(a).cuda(), out[0], out[1])
    assert got.dtype == torch.uint8
    assert np.array_equal(got.cpu().numpy(), np.stack([resize_bicubic_u8(x, out[0], out[1]) for x in a]))
    m = rng.randint(0, 3, shape).astype(np.uint8)
    gm = ops.resize_nearest(torch.from_numpy(m).cuda(), out[0], out[1]).cpu().numpy()
    assert np.array_equal(gm, np.stack([resize_nearest(x, out[0], out[1]) for x in m]))


def test_resize_u8_golden_fixture_and_crop_windows(ops):
    """The committed Pillow outputs (tests/golden/augment.npz) straight through the kernels; crop windows + flip on 8-bit images."""
    import os
    from oracle.augment import resize_bicubic_u8
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))
    i = 0
    while f"resize_u8_{i}.in" in z:
        a, ref = z[f"resize_u8_{i}.in"], z[f"resize_u8_{i}.out"]
        assert np.array_equal(ops.resize_bicubic(torch.from_numpy(a[None]).cuda(), *ref.shape)[0].cpu().numpy(), ref), i
        m, refm = z[f"nearest{i}.in"], z[f"nearest{i}.out"]
        assert np.array_equal(ops.resize_nearest(torch.from_numpy(m[None]).cuda(), *refm.shape)[0].cpu().numpy(), refm), i
        i += 1
    assert i == 6
    rng = np.random.RandomState(4)
    a = rng.randint(0, 256, (3, 256, 256)).astype(np.uint8)
    boxes = np.array([[0, 0, 256, 256], [17, 40, 100, 131], [200, 3, 56, 250]], np.int32)
    flips = np.array([1, 0, 1], np.uint8)
    got = ops.resize_bicubic(torch.from_numpy(a).cuda(), 256, 256, torch.from_numpy(boxes), torch.from_numpy(flips)).cpu().numpy()
    for b in range(3):
        x0, y0, w, h = boxes[b]
        r = resize_bicubic_u8(a[b, y0:y0 + h, x0:x0 + w], 256, 256)
        assert np.array_equal(got[b], r[:, ::-1] if flips[b] else r), b


def test_device_segmentation_batch(ops):
    """DeviceSegmentationBatch against the host item path of SegmentationDataset (Pillow resize + one_hot_encode), float32 and uint8."""
    from cmunet_amd.dataset import DeviceSegmentationBatch, one_hot_encode
    from oracle.augment import resize_bicubic, resize_bicubic_u8, resize_nearest
    rng = np.random.RandomState(8)
    masks = (rng.rand(3, 300, 280) < 0.2).astype(np.uint8)
    for imgs, host in ((rng.standard_normal((3, 300, 280)).astype(np.float32), resize_bicubic),
                       (rng.randint(0, 256, (3, 300, 280)).astype(np.uint8), resize_bicubic_u8)):
        x, y = DeviceSegmentationBatch(256, class_values=np.array([[0], [1]]), last_axis=True)(torch.from_numpy(imgs).cuda(), torch.from_numpy(masks).cuda())
        assert x.shape == (3, 1, 256, 256) and y.shape == (3, 2, 256, 256) and y.dtype == torch.float64
        assert np.array_equal(x[:, 0].cpu().numpy(), np.stack([host(v, 256, 256) for v in imgs]))
        ref = np.stack([one_hot_encode(resize_nearest(m, 256, 256), [0, 1]).astype('float') for m in masks])
        assert np.array_equal(y.cpu().numpy(), ref)
