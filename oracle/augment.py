"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the CM-UNet input pipeline's arithmetic.

  resize_bicubic_crop   cmunet_dataset.py:74-75 (``Image.resize((256, 256), resample=Image.BICUBIC)`` on a float32 array ->
                        PIL mode 'F') and the RandomResizedCrop of cmunet_config.py:49 (integer crop window, then a
                        bicubic resize to 256 x 256 with the pillow backend) + RandomFlip (cmunet_config.py:50).
                        The resampling itself lives in Pillow (third party, pinned ``pillow`` of environment.yml; 12.2.0 in
                        this container), not under /root/reference: this restates Pillow's published separable
                        convolution resize (Resample.c: support = 2 * max(scale, 1), Keys cubic a = -0.5, window
                        [int(c - s + .5), int(c + s + .5)) clipped to the image, coefficients normalised by their sum,
                        double accumulation, float32 store after each pass, horizontal pass first) and is pinned
                        bit-for-bit against Pillow itself in tests/test_cpu_oracle.py.
  resize_bicubic_u8     the same call on a uint8 array (PIL mode 'L'; Finetuning/dataset.py:44-46, Spark/utils/dataset.py:25-27 take whatever
                        dtype the .npy holds): Pillow's 8-bit path -- the same double coefficients turned into 22-bit fixed point
                        (round half away from zero), int32 accumulation from 2^21, arithmetic shift, clip to 0..255, a uint8 image
                        between the two passes.  Pinned bit-for-bit against Pillow (fixture + live).
  resize_nearest        ``mask.resize((256, 256), resample=Image.NEAREST)`` of Finetuning/dataset.py:47: Pillow's affine nearest path
                        (Geometry.c ImagingScaleAffine): source index int(x) of a running double x = scale/2, += scale per step.
  two_view              cmunet_dataset.py:76-88 + processing.py:97-127 (ShiftPixel: crop 224 x 224 at (dy, dx)) +
                        auto_augment.py:1136-1153 (GaussNoise: ``img + (max(img)/10) * randn`` in float64, cast back to
                        float32; applied whatever ``prob`` says, SURVEY A-11).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np


def _cubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def _coeffs(in_size, out_size):
    """Pillow precompute_coeffs for a whole-image box: per output index (xmin, normalised double weights)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        k = np.array([_cubic((x + xmin - center + 0.5) * ss) for x in range(xmax)], dtype=np.float64)
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = k / ww
        out.append((xmin, k))
    return out


def resize_bicubic(img, out_h, out_w):
    """img (H, W) float32 -> (out_h, out_w) float32, Pillow mode-'F' BICUBIC semantics."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape
    tmp = img
    if W != out_w:
        cx = _coeffs(W, out_w)
        tmp = np.empty((H, out_w), np.float32)
        for xx, (xmin, k) in enumerate(cx):
            acc = np.zeros(H, np.float64)
            for x in range(len(k)):
                acc = acc + img[:, xmin + x].astype(np.float64) * k[x]
            tmp[:, xx] = acc.astype(np.float32)
    out = tmp
    if H != out_h:
        cy = _coeffs(H, out_h)
        out = np.empty((out_h, tmp.shape[1]), np.float32)
        for yy, (ymin, k) in enumerate(cy):
            acc = np.zeros(tmp.shape[1], np.float64)
            for y in range(len(k)):
                acc = acc + tmp[ymin + y, :].astype(np.float64) * k[y]
            out[yy, :] = acc.astype(np.float32)
    return out


_PREC = 32 - 8 - 2          # Pillow PRECISION_BITS of the 8-bit resampler


def _coeffs_u8(in_size, out_size):
    """normalize_coeffs_8bpc: the double coefficients as 22-bit fixed point, rounded half away from zero (C cast truncation)."""
    out = []
    for xmin, k in _coeffs(in_size, out_size):
        ki = [int(-0.5 + v * (1 << _PREC)) if v < 0 else int(0.5 + v * (1 << _PREC)) for v in k]
        out.append((xmin, np.array(ki, dtype=np.int64)))
    return out


def _clip8(acc):
    return np.clip(acc >> _PREC, 0, 255).astype(np.uint8)       # arithmetic shift on signed values, as C does on int32


def resize_bicubic_u8(img, out_h, out_w):
    """img (H, W) uint8 -> (out_h, out_w) uint8, Pillow mode-'L' BICUBIC semantics (Resample.c ImagingResampleHorizontal_8bpc /
    Vertical_8bpc: horizontal pass first into a uint8 image, a pass whose sizes agree is skipped)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    H, W = img.shape
    tmp = img
    if W != out_w:
        tmp = np.empty((H, out_w), np.uint8)
        for xx, (xmin, k) in enumerate(_coeffs_u8(W, out_w)):
            acc = np.full(H, 1 << (_PREC - 1), np.int64)
            for x in range(len(k)):
                acc = acc + img[:, xmin + x].astype(np.int64) * k[x]
            assert np.all(np.abs(acc) < 2 ** 31)                  # Pillow accumulates in int32: no wrap to restate
            tmp[:, xx] = _clip8(acc)
    out = tmp
    if H != out_h:
        out = np.empty((out_h, tmp.shape[1]), np.uint8)
        for yy, (ymin, k) in enumerate(_coeffs_u8(H, out_h)):
            acc = np.full(tmp.shape[1], 1 << (_PREC - 1), np.int64)
            for y in range(len(k)):
                acc = acc + tmp[ymin + y, :].astype(np.int64) * k[y]
            out[yy, :] = _clip8(acc)
    return out


def _nearest_index(in_size, out_size):
    scale = in_size / out_size
    idx = np.empty(out_size, np.int64)
    xo = scale * 0.5                                   # a[2] + a[0] * 0.5 with a zero box origin
    for x in range(out_size):
        idx[x] = -1 if xo < 0.0 else int(xo)           # COORD()
        xo += scale                                    # running sum in double, as Pillow accumulates it
    return idx


def resize_nearest(img, out_h, out_w):
    """img (H, W) any dtype -> (out_h, out_w), Pillow NEAREST resize (source pixels outside the image do not occur for a whole-image
    box; Pillow would leave them unwritten)."""
    img = np.asarray(img)
    H, W = img.shape
    iy, ix = _nearest_index(H, out_h), _nearest_index(W, out_w)
    assert iy.min() >= 0 and iy.max() < H and ix.min() >= 0 and ix.max() < W
    return img[iy][:, ix]


def resize_bicubic_crop(batch, boxes, flips, out_h, out_w):
    """batch (B, H, W) float32; boxes (B, 4) int (x0, y0, w, h) crop windows; flips (B,) bool (horizontal flip after the
    resize: mmcv RandomFlip default direction) -> (B, out_h, out_w) float32."""
    res = np.empty((len(batch), out_h, out_w), np.float32)
    for b, img in enumerate(batch):
        x0, y0, w, h = [int(v) for v in boxes[b]]
        r = resize_bicubic(img[y0:y0 + h, x0:x0 + w], out_h, out_w)
        res[b] = r[:, ::-1] if flips[b] else r
    return res


def two_view(batch, shifts, noise, out=224):
    """batch (B, S, S) float32; shifts (B, 2) int (dy, dx) of the augmented view; noise (B, out, out) float64 standard
    normal draws -> (img (B, out, out) float32, img_t (B, out, out) float32)."""
    B = len(batch)
    img = np.empty((B, out, out), np.float32)
    img_t = np.empty((B, out, out), np.float32)
    for b in range(B):
        dy, dx = int(shifts[b][0]), int(shifts[b][1])
        img[b] = batch[b, :out, :out]
        v = batch[b, dy:dy + out, dx:dx + out]
        sigma = np.max(v) / 10                      # float32 scalar
        img_t[b] = np.array(v + sigma * noise[b], dtype=np.float32)   # float64 arithmetic, cast back
    return img, img_t


def philox4x32_10(c, k):
    """Philox4x32-10 (Salmon et al., SC'11; the Random123 constants): c (n, 4) uint32 counters, k (2,) uint32 key."""
    c = np.asarray(c, dtype=np.uint64).copy()
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def philox_normal(n, offset, seed):
    """The in-kernel generator of cmu_two_view / cmu_philox_normal: element e = offset + i -> counter (e_lo, e_hi, 0, 0),
    key (seed_lo, seed_hi); Box-Muller on the first two output words, uniforms (r + 0.5) / 2^32."""
    e = np.arange(n, dtype=np.uint64) + np.uint64(offset)
    c = np.stack([e & np.uint64(0xFFFFFFFF), e >> np.uint64(32), np.zeros_like(e), np.zeros_like(e)], axis=1)
    r = philox4x32_10(c, (seed & 0xFFFFFFFF, seed >> 32)).astype(np.float64)
    u1, u2 = (r[:, 0] + 0.5) / 4294967296.0, (r[:, 1] + 0.5) / 4294967296.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)


def random_patch_mask(B, H, W, patch, n_mask, seed, offset=0):
    """cmu_random_patch_mask restated: patch p of sample b draws the first Philox word of counter offset + b*P + p; the n_mask
    smallest (key, p) pairs of a sample are masked.  Same distribution as UNet_encoder.py:106-139 (shuffle, take the first)."""
    ph, pw = H // patch, W // patch
    P = ph * pw
    e = np.arange(B * P, dtype=np.uint64) + np.uint64(offset)
    c = np.stack([e & np.uint64(0xFFFFFFFF), e >> np.uint64(32), np.zeros_like(e), np.zeros_like(e)], axis=1)
    key = philox4x32_10(c, (seed & 0xFFFFFFFF, seed >> 32))[:, 0].astype(np.uint64).reshape(B, P)
    full = (key << np.uint64(32)) | np.arange(P, dtype=np.uint64)[None, :]
    rank = full.argsort(axis=1).argsort(axis=1)
    m = (rank < n_mask).astype(np.uint8).reshape(B, ph, pw)
    return np.repeat(np.repeat(m, patch, axis=1), patch, axis=2)
