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
