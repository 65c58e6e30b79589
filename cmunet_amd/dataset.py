"""Dataset surface of the hot path's callers (SURVEY rows a5 / 2.1 "surface to keep"): same item contracts as
the reference, plus the synthetic generators used by bench.py and the tests (the FAME2 data are private).

  SegmentationDataset(images_dir, masks_dir, augmentation=None, class_values=None, last_axis=False)
      Finetuning/dataset.py:12-55: ``.npy`` float32 image + ``.npy`` {0,1} mask -> optional augmentation
      (any callable with albumentations' ``aug(image=, mask=) -> {'image','mask'}`` protocol; the library itself
      is out of scope) -> PIL resize to 256x256 (bicubic / nearest) -> one-hot float64 mask.
      item = (image (256,256) float32, mask (n_cls,256,256) float64).
  CMUNetDataset-style two-view items {'img','img_t'} (cmunet_dataset.py:60-88): ``two_view_item``.
"""
import numpy as np
import torch
from torch.utils.data import Dataset


def one_hot_encode(label, label_values):
    """Finetuning/dataset.py:79-97 semantics: stack (label == v) over the class values -> (n_cls,H,W)."""
    label = np.asarray(label)
    return np.stack([np.equal(label, v) for v in label_values], axis=0)


class SegmentationDataset(Dataset):
    def __init__(self, images_dir, masks_dir, augmentation=None, class_values=None, last_axis=False, size=256):
        self.image_paths = images_dir
        self.mask_paths = masks_dir
        self.class_values = class_values if class_values is not None else [0, 1]
        self.augmentation = augmentation
        self.last_axis = last_axis
        self.size = size

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        from PIL import Image
        image = np.load(self.image_paths[idx])
        mask = np.load(self.mask_paths[idx])
        if self.augmentation is not None:
            sample = self.augmentation(image=image, mask=mask)
            image, mask = sample['image'], sample['mask']
        image = Image.fromarray(image).resize((self.size, self.size), resample=Image.BICUBIC)
        mask = Image.fromarray(mask).resize((self.size, self.size), resample=Image.NEAREST)
        mask = one_hot_encode(mask, self.class_values).astype('float')
        if self.last_axis:
            image = np.transpose(np.asarray(image)[..., np.newaxis], (2, 0, 1))
        else:
            image = np.asarray(image)
        return (image, np.asarray(mask))


def synthetic_vessel_mask(h, w, rng, density=0.1):
    """Smooth random field thresholded at its (1-density) quantile: vessel-like blobs."""
    f = rng.standard_normal((h // 8 + 2, w // 8 + 2)).astype(np.float32)
    f = np.kron(f, np.ones((8, 8), np.float32))[:h + 8, :w + 8]
    k = np.ones(9, np.float32) / 9
    for ax in (0, 1):
        f = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), ax, f)
    f = f[4:h + 4, 4:w + 4]
    return (f > np.quantile(f, 1 - density)).astype(np.uint8)


class SyntheticSegmentationDataset(Dataset):
    """Same item contract as SegmentationDataset, data generated from a seed (z-scored float32 images whose
    intensity correlates with the mask, so a few epochs of training visibly reduce the loss)."""

    def __init__(self, n=8, size=256, seed=42, density=0.1):
        rng = np.random.RandomState(seed)
        self.items = []
        for _ in range(n):
            m = synthetic_vessel_mask(size, size, rng, density)
            img = rng.standard_normal((size, size)).astype(np.float32) + 1.5 * m
            img = (img - img.mean()) / (img.std() + 1e-6)
            self.items.append((img.astype(np.float32), one_hot_encode(m, [0, 1]).astype('float')))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        return self.items[idx]


def shift_pixel_crop(img, pixel, out, rng):
    """ShiftPixel (processing.py:97-127): crop ``out`` x ``out`` at a random offset in [0, pixel]."""
    dy, dx = (rng.randint(0, pixel + 1), rng.randint(0, pixel + 1)) if pixel > 0 else (0, 0)
    return img[dy:dy + out, dx:dx + out]


def two_view_item(img256, rng, pixel=31, out=224):
    """cmunet_dataset.py:60-88: 'img' = ShiftPixel(0) crop, 'img_t' = ShiftPixel(<=pixel) crop + Gaussian noise
    with sigma = max(img)/10 (always applied: auto_augment.py:1149-1155, SURVEY A-11)."""
    a = shift_pixel_crop(img256, 0, out, rng).astype(np.float32)
    b = shift_pixel_crop(img256, pixel, out, rng).astype(np.float32)
    b = b + rng.standard_normal(b.shape).astype(np.float32) * (float(b.max()) / 10.0)
    return {'img': a, 'img_t': b.astype(np.float32)}


class SyntheticTwoViewDataset(Dataset):
    def __init__(self, n=64, seed=60, pixel=31, out=224):
        self.rng = np.random.RandomState(seed)
        self.base = [self.rng.standard_normal((256, 256)).astype(np.float32) for _ in range(n)]
        self.pixel, self.out = pixel, out

    def __len__(self):
        return len(self.base)

    def __getitem__(self, idx):
        it = two_view_item(self.base[idx], self.rng, self.pixel, self.out)
        return {k: torch.from_numpy(v) for k, v in it.items()}


# ---------------------------------------------------------------------------------------------------------------------
# the same pipeline on the device: batches of raw images resident in HBM -> the two views of a step
# ---------------------------------------------------------------------------------------------------------------------
def random_resized_crop_params(h, w, rng, crop_ratio_range=(0.2, 1.0), aspect_ratio_range=(3. / 4., 4. / 3.), max_attempts=10):
    """Crop window of mmcv's RandomResizedCrop (the transform configs/cmunet_config.py:49 names; mmcv itself is a third-party
    dependency absent from /root/reference): up to ``max_attempts`` draws of (area ratio uniform, aspect ratio log-uniform),
    the first that fits wins, else the central crop with the aspect ratio clamped.  Returns (x0, y0, cw, ch)."""
    area = h * w
    lo, hi = np.log(aspect_ratio_range[0]), np.log(aspect_ratio_range[1])
    for _ in range(max_attempts):
        target_area = rng.uniform(*crop_ratio_range) * area
        aspect = np.exp(rng.uniform(lo, hi))
        cw, ch = int(round(np.sqrt(target_area * aspect))), int(round(np.sqrt(target_area / aspect)))
        if 0 < cw <= w and 0 < ch <= h:
            return int(rng.randint(0, w - cw + 1)), int(rng.randint(0, h - ch + 1)), cw, ch
    in_ratio = w / h
    if in_ratio < aspect_ratio_range[0]:
        cw, ch = w, int(round(w / aspect_ratio_range[0]))
    elif in_ratio > aspect_ratio_range[1]:
        cw, ch = int(round(h * aspect_ratio_range[1])), h
    else:
        cw, ch = w, h
    return (w - cw) // 2, (h - ch) // 2, cw, ch


class DeviceSegmentationBatch:
    """SegmentationDataset.__getitem__'s resize and encoding (Finetuning/dataset.py:44-55) for a whole batch on the GPU: images
    (B,H,W) float32 or uint8 -> bicubic resize to ``size`` in Pillow's arithmetic for that dtype (mode 'F' / mode 'L'); masks (B,H,W)
    uint8 -> NEAREST resize -> one-hot over ``class_values`` as float64 (SURVEY A-5).  Returns (image (B,size,size) of the input
    dtype -- (B,1,size,size) with ``last_axis`` --, mask (B,n_cls,size,size) float64).  Requires the HIP library (no CPU fallback)."""

    def __init__(self, size=256, class_values=None, last_axis=False):
        self.size, self.last_axis = size, last_axis
        self.class_values = [int(np.asarray(v).reshape(-1)[0]) for v in (class_values if class_values is not None else [0, 1])]

    def __call__(self, images, masks):
        from . import ops
        assert images.is_cuda and images.dim() == 3 and masks.is_cuda and masks.dtype == torch.uint8 and masks.shape[0] == images.shape[0]
        S = self.size
        img = ops.resize_bicubic(images.contiguous(), S, S)
        lab = ops.resize_nearest(masks.contiguous(), S, S)
        vals = torch.tensor(self.class_values, dtype=torch.uint8, device=lab.device).view(1, -1, 1, 1)
        onehot = (lab[:, None] == vals).to(torch.float64)
        return (img[:, None] if self.last_axis else img), onehot


class DeviceTwoViewPipeline:
    """CMUNetDataset.__getitem__ (cmunet_dataset.py:60-88) for a whole batch on the GPU: bicubic resize to ``size`` ->
    RandomResizedCrop(size, crop_ratio (0.2, 1), bicubic) + RandomFlip(0.5) (cmunet_config.py:48-51) -> 'img' =
    ShiftPixel(0) crop, 'img_t' = ShiftPixel(<= pixel) crop + GaussNoise.  The few random scalars per sample (window, flip,
    shifts) are drawn on the host from a seeded numpy generator; the pixels never leave HBM (the noise comes from the
    kernel's counter-based generator, keyed by (seed, call count)).  Requires the HIP library (no CPU fallback)."""

    def __init__(self, size=256, out=224, pixel=31, crop_ratio_range=(0.2, 1.0), flip_prob=0.5, seed=0):
        self.size, self.out, self.pixel = size, out, pixel
        self.crop_ratio_range, self.flip_prob = crop_ratio_range, flip_prob
        self.rng = np.random.RandomState(seed)
        self.seed, self.calls = int(seed), 0

    def draw(self, B):
        """The random scalars of one batch: boxes (B,4) int32 (x0, y0, w, h), flips (B,) uint8, shifts (B,2) int32."""
        boxes = np.array([random_resized_crop_params(self.size, self.size, self.rng, self.crop_ratio_range) for _ in range(B)], np.int32)
        flips = (self.rng.uniform(size=B) < self.flip_prob).astype(np.uint8)
        shifts = self.rng.randint(0, self.pixel + 1, size=(B, 2)).astype(np.int32) if self.pixel > 0 else np.zeros((B, 2), np.int32)
        return boxes, flips, shifts

    def __call__(self, raw, params=None, noise=None):
        from . import ops
        assert raw.is_cuda and raw.dtype in (torch.float32, torch.uint8) and raw.dim() == 3
        raw = raw.contiguous()
        B = raw.shape[0]
        boxes, flips, shifts = params if params is not None else self.draw(B)
        base = raw if tuple(raw.shape[1:]) == (self.size, self.size) else ops.resize_bicubic(raw, self.size, self.size)
        crop = ops.resize_bicubic(base, self.size, self.size, torch.from_numpy(np.asarray(boxes)), torch.from_numpy(np.asarray(flips)))
        if crop.dtype == torch.uint8:      # 8-bit sources: resized and cropped in Pillow's mode-'L' arithmetic, the views are float32
            crop = crop.float()
        self.calls += 1
        img, img_t = ops.two_view(crop, torch.from_numpy(np.asarray(shifts)), self.out, noise=noise,
                                  seed=(self.seed << 32) | (self.calls & 0xFFFFFFFF))
        return {'img': img, 'img_t': img_t}
