"""Dataset surface of the hot path's callers (SURVEY rows a5 / 2.1 "surface to keep"): same item contracts as
the reference, plus the synthetic generators used by bench.py and the tests (the FAME2 data are private).

  SegmentationDataset(images_dir, masks_dir, augmentation=None, class_values=None, last_axis=False)
      Finetuning/dataset.py:12-55: ``.npy`` float32 image + ``.npy`` {0,1} mask -> optional augmentation
      (any callable with albumentations' ``aug(image=, mask=) -> {'image','mask'}`` protocol; the library itself
      is out of scope) -> PIL resize to 256x256 (bicubic / nearest) -> one-hot float64 mask.
      item = (image (256,256) float32, mask (n_cls,256,256) float64).
  CMUNetDataset(data_root, data_ann, pipeline, pixel=31, test=False)   cmae/datasets/cmunet_dataset.py:16-88: directory of ``.npy``
      images, the reference's two train_test_split calls, items {'img','img_t'} built by ``two_view_item``.
  MoCoDataset(data_path, tau_g)                                         moco/moco_data_set.py:11-37: item ((crop_0, crop_1), 0).
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


def train_test_split_indices(n, test_size, random_state):
    """Index form of sklearn.model_selection.train_test_split(shuffle=True) for a float ``test_size`` (what
    cmunet_dataset.py:30-31 calls twice): n_test = ceil(test_size * n), one permutation of a RandomState(random_state), the first
    n_test indices are the test set, the rest the training set.  -> (train indices, test indices); checked against sklearn in
    tests/test_cpu_surface.py."""
    n_test = int(np.ceil(test_size * n))
    n_train = n - n_test
    if n_train <= 0:
        raise ValueError(f"With n_samples={n} and test_size={test_size} the resulting train set will be empty")
    perm = np.random.RandomState(random_state).permutation(n)
    return perm[n_test:n_test + n_train], perm[:n_test]


class _RandomResizedCropFlip:
    """The two base transforms of configs/cmunet_config.py:48-51 on a 'img' results dict: RandomResizedCrop(scale 256,
    crop_ratio_range (0.2, 1), pillow bicubic) + RandomFlip(0.5, horizontal).  (mmcv's transform classes are a third-party
    dependency absent here; the crop law is ``random_resized_crop_params``, the resampling is Pillow's, as the config asks.)"""

    def __init__(self, scale=256, crop_ratio_range=(0.2, 1.0), flip_prob=0.5, rng=None):
        self.scale, self.crop_ratio_range, self.flip_prob = scale, crop_ratio_range, flip_prob
        self.rng = rng if rng is not None else np.random

    def __call__(self, results):
        from PIL import Image
        img = np.asarray(results['img'])
        h, w = img.shape[:2]
        x0, y0, cw, ch = random_resized_crop_params(h, w, self.rng, self.crop_ratio_range)
        out = Image.fromarray(img).resize((self.scale, self.scale), resample=Image.BICUBIC, box=(x0, y0, x0 + cw, y0 + ch))
        out = np.asarray(out)
        if self.rng.uniform() < self.flip_prob:
            out = out[:, ::-1]
        return {'img': np.ascontiguousarray(out)}


class CMUNetDataset(Dataset):
    """cmae/datasets/cmunet_dataset.py:16-88, same constructor and item contract: ``data_root`` is a directory of ``.npy`` float
    images; the file list is sorted, split 80 / 20 (random_state 42) and the training part once more 98.75 / 1.25 -- ``image_paths``
    is that inner training part, ``test`` the 20 % (``test=`` itself is accepted and unused, as in the reference).
    ``__getitem__`` -> {'img', 'img_t'}: np.load -> PIL bicubic resize to 256 x 256 -> base pipeline (first two entries:
    RandomResizedCrop + RandomFlip) -> 'img' = ShiftPixel(0) crop, 'img_t' = ShiftPixel(<= pixel) crop + GaussNoise (always applied,
    sigma = max / 10: SURVEY A-11) -> final pipeline (the entries after the first two).

    ``pipeline``: None = the shipped configuration (cmunet_config.py:48-53) with tensors out; or a list whose entries are callables on
    a results dict (mmengine-style config dicts need mmcv / mmengine, which are not part of this build: a dict entry raises).  The
    whole-batch form that keeps the pixels in HBM is ``DeviceTwoViewPipeline``."""

    def __init__(self, data_root, data_ann=None, pipeline=None, pixel=31, test=False, out=224, seed=None):
        import os
        self.data_root = data_root
        paths = [os.path.join(data_root, f) for f in sorted(os.listdir(data_root))]
        tr, te = train_test_split_indices(len(paths), 0.2, 42)
        inner, _ = train_test_split_indices(len(tr), 0.0125, 42)
        self.image_paths = [paths[tr[i]] for i in inner]
        self.test = [paths[i] for i in te]
        self.pixel, self.out = pixel, out
        self.rng = np.random.RandomState(seed) if seed is not None else np.random
        if pipeline is None:
            base, final = [_RandomResizedCropFlip(256, (0.2, 1.0), 0.5, self.rng)], [lambda r: {'img': torch.from_numpy(np.ascontiguousarray(r['img']))}]
        else:
            for t in pipeline:
                if not callable(t):
                    raise TypeError("CMUNetDataset: pipeline entries must be callables on a results dict (config dicts need mmcv / mmengine)")
            base, final = list(pipeline[:2]), list(pipeline[2:])
        self.pipeline_base, self.pipeline_final = base, final

    def __len__(self):
        return len(self.image_paths)

    @staticmethod
    def _run(transforms, results):
        for t in transforms:
            results = t(results)
        return results

    def __getitem__(self, idx):
        from PIL import Image
        image = np.load(self.image_paths[idx])
        image = np.asarray(Image.fromarray(image).resize((256, 256), resample=Image.BICUBIC))
        src = self._run(self.pipeline_base, {'img': image})['img']
        views = two_view_item(np.asarray(src, dtype=np.float32), self.rng, self.pixel, self.out)
        patch = self._run(self.pipeline_final, {'img': views['img']})
        img_t = self._run(self.pipeline_final, {'img': views['img_t']})
        return {'img': patch['img'], 'img_t': img_t['img']}


class MoCoDataset(Dataset):
    """pl_bolts/models/self_supervised/moco/moco_data_set.py:11-37, same constructor and item contract: ``data_path`` is a LIST of
    ``.npy`` paths, ``tau_g`` a list of (two) global transforms on a (1, 256, 256) tensor; item = ((crop_0, crop_1), 0) after
    np.load -> PIL bicubic resize to 256 x 256."""

    def __init__(self, data_path, tau_g):
        self.tau_g = tau_g
        self.data_path = data_path

    def __len__(self):
        return len(self.data_path)

    def __str__(self):
        return f"LoGoDataset with {self.__len__()} images"

    def __getitem__(self, idx):
        from PIL import Image
        image = np.load(self.data_path[idx])
        image = np.array(Image.fromarray(image).resize((256, 256), resample=Image.BICUBIC))
        image = torch.from_numpy(image[np.newaxis, :])
        crops = [t(image) for t in self.tau_g]
        return (crops[0], crops[1]), 0


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
