// augment.hip -- the CM-UNet input pipeline on the device: what the reference does per sample on the host
// with Pillow / numpy (Pretraining/CM-UNet/cmae/datasets/cmunet_dataset.py:60-88) as three batch kernels, so that a batch of
// raw images resident in HBM becomes the two views of a pretraining step without host loops or H2D copies.
//   resize_h / resize_v   Image.resize((256, 256), BICUBIC) of cmunet_dataset.py:74-75 and the RandomResizedCrop + RandomFlip
//                         of configs/cmunet_config.py:49-50 (integer crop window, bicubic resize with the pillow backend,
//                         horizontal flip).  Pillow's separable resampling, restated: per output index the window
//                         [int(c - s + .5), int(c + s + .5)) around c = (i + .5) * scale with s = 2 * max(scale, 1), clipped
//                         to the crop; Keys cubic (a = -0.5) weights normalised by their sum; double accumulation in tap
//                         order; float32 store after each pass; horizontal pass first; a pass whose sizes agree is a copy.
//                         Every double operation is an explicit round-to-nearest mul / add / div (no FMA contraction), which
//                         is what makes the result bit-identical to Pillow's (tests/test_gpu_augment.py via oracle/augment.py,
//                         itself pinned to Pillow bit for bit).
//   two_view              ShiftPixel (pipelines/processing.py:97-127: 224 x 224 crop at (0, 0) for 'img', at (dy, dx) for
//                         'img_t') + GaussNoise (pipelines/auto_augment.py:1136-1153: img + (max(img) / 10) * randn evaluated
//                         in float64 and cast back; applied whatever `prob` says, SURVEY A-11).  The normal draws are an
//                         explicit float64 input (bit-exact parity) or come from a counter-based Philox4x32-10 + Box-Muller
//                         generator in the kernel (no 8-byte-per-pixel read; one workgroup per sample).
// All three are HBM-bound single passes over a few MB; nothing here is on the bench's timed path.
#include "common.h"
#include <math.h>
// hipcc contracts a*b + c into an FMA by default (-ffp-contract=fast), through the __d*_rn intrinsics too (they are plain
// operators in the HIP headers) and past this pragma: one rounding instead of two, i.e. a handful of results per image 1 ulp
// off Pillow's.  The Makefile compiles this file with -ffp-contract=off; tests/test_gpu_augment.py would catch a build without.
#pragma clang fp contract(off)

__device__ static inline double aug_cubic(double x) {
    // Pillow bicubic_filter with a = -0.5: ((a+2)x - (a+3))x^2 + 1 on [0,1), (((x-5)x + 8)x - 4)a on [1,2)
    x = fabs(x);
    if (x < 1.0) return __dadd_rn(__dmul_rn(__dmul_rn(__dadd_rn(__dmul_rn(1.5, x), -2.5), x), x), 1.0);
    if (x < 2.0) return __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(x, -5.0), x), 8.0), x), -4.0), -0.5);
    return 0.0;
}

struct AugWin {
    int xmin, count;
    double center, ss, ww;
};
// the window and the normalisation sum of output index xx for a resize in_size -> out_size
__device__ static inline AugWin aug_window(int xx, int in_size, int out_size) {
    AugWin w;
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = __dmul_rn(2.0, filterscale);
    w.ss = __ddiv_rn(1.0, filterscale);
    w.center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
    int xmin = (int)__dadd_rn(__dadd_rn(w.center, -support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(w.center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    w.xmin = xmin;
    w.count = xmax - xmin;
    double ww = 0.0;
    for (int x = 0; x < w.count; ++x) ww = __dadd_rn(ww, aug_cubic(__dmul_rn(__dadd_rn(__dadd_rn((double)(x + xmin), -w.center), 0.5), w.ss)));
    w.ww = ww;
    return w;
}
__device__ static inline double aug_coeff(const AugWin& w, int x) {
    const double k = aug_cubic(__dmul_rn(__dadd_rn(__dadd_rn((double)(x + w.xmin), -w.center), 0.5), w.ss));
    return w.ww != 0.0 ? __ddiv_rn(k, w.ww) : k;
}

// horizontal pass: tmp[b][y][xx] for the rows of the crop window; y indexes rows of the window
__global__ void resize_h_kernel(const float* __restrict__ src, int Hs, int Ws, const int* __restrict__ boxes, float* __restrict__ tmp,
                                int Wo, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo);
        const int y = (int)((i / Wo) % Hs);
        const int b = (int)(i / ((int64_t)Wo * Hs));
        const int x0 = boxes ? boxes[4 * b + 0] : 0, y0 = boxes ? boxes[4 * b + 1] : 0;
        const int w = boxes ? boxes[4 * b + 2] : Ws, h = boxes ? boxes[4 * b + 3] : Hs;
        if (y >= h) continue;
        const float* row = src + ((int64_t)b * Hs + y0 + y) * Ws + x0;
        float r;
        if (w == Wo) {
            r = row[xx];
        } else {
            const AugWin win = aug_window(xx, w, Wo);
            double ss = 0.0;
            for (int x = 0; x < win.count; ++x) ss = __dadd_rn(ss, __dmul_rn((double)row[win.xmin + x], aug_coeff(win, x)));
            r = (float)ss;
        }
        tmp[i] = r;
    }
}
// vertical pass + optional horizontal flip of the result
__global__ void resize_v_kernel(const float* __restrict__ tmp, int Hs, const int* __restrict__ boxes, const uint8_t* __restrict__ flip,
                                float* __restrict__ out, int Ho, int Wo, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo);
        const int yy = (int)((i / Wo) % Ho);
        const int b = (int)(i / ((int64_t)Wo * Ho));
        const int h = boxes ? boxes[4 * b + 3] : Hs;
        const float* col = tmp + (int64_t)b * Hs * Wo + xx;
        float r;
        if (h == Ho) {
            r = col[(int64_t)yy * Wo];
        } else {
            const AugWin win = aug_window(yy, h, Ho);
            double ss = 0.0;
            for (int y = 0; y < win.count; ++y) ss = __dadd_rn(ss, __dmul_rn((double)col[(int64_t)(win.xmin + y) * Wo], aug_coeff(win, y)));
            r = (float)ss;
        }
        const int xo = (flip && flip[b]) ? Wo - 1 - xx : xx;
        out[((int64_t)b * Ho + yy) * Wo + xo] = r;
    }
}

extern "C" int64_t cmu_resize_bicubic_ws_bytes(int B, int Hs, int Ws, int Ho, int Wo) {
    (void)Ws; (void)Ho;
    return (int64_t)B * Hs * Wo * (int64_t)sizeof(float);
}
extern "C" int cmu_resize_bicubic(const float* src, int B, int Hs, int Ws, const int* boxes, const uint8_t* flip, float* out, int Ho, int Wo,
                                  void* ws, void* stream) {
    CMU_CHECK_ARG(src && out && ws && B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "cmu_resize_bicubic: bad args");
    // (the crop windows are device data: the host wrapper checks 0 <= x0, x0 + w <= Ws, w >= 1 and the same for rows)
    const int64_t t1 = (int64_t)B * Hs * Wo, t2 = (int64_t)B * Ho * Wo;
    const int g1 = (int)(cmu_div_up64(t1, 256) < 16384 ? cmu_div_up64(t1, 256) : 16384);
    const int g2 = (int)(cmu_div_up64(t2, 256) < 16384 ? cmu_div_up64(t2, 256) : 16384);
    hipLaunchKernelGGL(resize_h_kernel, dim3(g1), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, boxes, (float*)ws, Wo, t1);
    CMU_CHECK_LAUNCH("cmu_resize_bicubic(horizontal)");
    hipLaunchKernelGGL(resize_v_kernel, dim3(g2), dim3(256), 0, (hipStream_t)stream, (const float*)ws, Hs, boxes, flip, out, Ho, Wo, t2);
    CMU_CHECK_LAUNCH("cmu_resize_bicubic(vertical)");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// 8-bit images (PIL mode 'L': what Image.fromarray gives for a uint8 .npy; Finetuning/dataset.py:44-46, Spark/utils/dataset.py:25-27)
// Pillow's 8-bit resampler: the double coefficients above as 22-bit fixed point (rounded half away from zero by the C cast),
// int32 accumulation from 2^21, arithmetic shift by 22, clip to 0..255, a uint8 image between the passes.
// ---------------------------------------------------------------------------------------------
#define AUG_PREC 22
__device__ static inline int aug_coeff_u8(const AugWin& w, int x) {
    const double k = aug_coeff(w, x);
    return k < 0.0 ? (int)__dadd_rn(-0.5, __dmul_rn(k, (double)(1 << AUG_PREC))) : (int)__dadd_rn(0.5, __dmul_rn(k, (double)(1 << AUG_PREC)));
}
__device__ static inline uint8_t aug_clip8(int acc) {
    const int v = acc >> AUG_PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
__global__ void resize_h_u8_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, const int* __restrict__ boxes, uint8_t* __restrict__ tmp,
                                   int Wo, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo);
        const int y = (int)((i / Wo) % Hs);
        const int b = (int)(i / ((int64_t)Wo * Hs));
        const int x0 = boxes ? boxes[4 * b + 0] : 0, y0 = boxes ? boxes[4 * b + 1] : 0;
        const int w = boxes ? boxes[4 * b + 2] : Ws, h = boxes ? boxes[4 * b + 3] : Hs;
        if (y >= h) continue;
        const uint8_t* row = src + ((int64_t)b * Hs + y0 + y) * Ws + x0;
        uint8_t r;
        if (w == Wo) {
            r = row[xx];
        } else {
            const AugWin win = aug_window(xx, w, Wo);
            int ss = 1 << (AUG_PREC - 1);
            for (int x = 0; x < win.count; ++x) ss += (int)row[win.xmin + x] * aug_coeff_u8(win, x);
            r = aug_clip8(ss);
        }
        tmp[i] = r;
    }
}
__global__ void resize_v_u8_kernel(const uint8_t* __restrict__ tmp, int Hs, const int* __restrict__ boxes, const uint8_t* __restrict__ flip,
                                   uint8_t* __restrict__ out, int Ho, int Wo, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo);
        const int yy = (int)((i / Wo) % Ho);
        const int b = (int)(i / ((int64_t)Wo * Ho));
        const int h = boxes ? boxes[4 * b + 3] : Hs;
        const uint8_t* col = tmp + (int64_t)b * Hs * Wo + xx;
        uint8_t r;
        if (h == Ho) {
            r = col[(int64_t)yy * Wo];
        } else {
            const AugWin win = aug_window(yy, h, Ho);
            int ss = 1 << (AUG_PREC - 1);
            for (int y = 0; y < win.count; ++y) ss += (int)col[(int64_t)(win.xmin + y) * Wo] * aug_coeff_u8(win, y);
            r = aug_clip8(ss);
        }
        const int xo = (flip && flip[b]) ? Wo - 1 - xx : xx;
        out[((int64_t)b * Ho + yy) * Wo + xo] = r;
    }
}
extern "C" int64_t cmu_resize_bicubic_u8_ws_bytes(int B, int Hs, int Ws, int Ho, int Wo) {
    (void)Ws; (void)Ho;
    return (int64_t)B * Hs * Wo;
}
extern "C" int cmu_resize_bicubic_u8(const uint8_t* src, int B, int Hs, int Ws, const int* boxes, const uint8_t* flip, uint8_t* out, int Ho,
                                     int Wo, void* ws, void* stream) {
    CMU_CHECK_ARG(src && out && ws && B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "cmu_resize_bicubic_u8: bad args");
    const int64_t t1 = (int64_t)B * Hs * Wo, t2 = (int64_t)B * Ho * Wo;
    const int g1 = (int)(cmu_div_up64(t1, 256) < 16384 ? cmu_div_up64(t1, 256) : 16384);
    const int g2 = (int)(cmu_div_up64(t2, 256) < 16384 ? cmu_div_up64(t2, 256) : 16384);
    hipLaunchKernelGGL(resize_h_u8_kernel, dim3(g1), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, boxes, (uint8_t*)ws, Wo, t1);
    CMU_CHECK_LAUNCH("cmu_resize_bicubic_u8(horizontal)");
    hipLaunchKernelGGL(resize_v_u8_kernel, dim3(g2), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)ws, Hs, boxes, flip, out, Ho, Wo, t2);
    CMU_CHECK_LAUNCH("cmu_resize_bicubic_u8(vertical)");
    return CMU_OK;
}

// Image.resize(size, NEAREST) of the label masks (Finetuning/dataset.py:47): Pillow's affine nearest path takes source index
// int(x) of a running double x = scale / 2, += scale per output index (the running sum, not (i + .5) * scale: its roundings are
// Pillow's).  One thread per axis builds the index table, a second launch gathers.
__global__ void nearest_table_kernel(int Hs, int Ws, int Ho, int Wo, int* __restrict__ tab) {
    const int axis = threadIdx.x;                       // 0: columns -> tab[0 .. Wo), 1: rows -> tab[Wo .. Wo + Ho)
    if (axis > 1) return;
    const int in_size = axis ? Hs : Ws, out_size = axis ? Ho : Wo;
    int* t = tab + (axis ? Wo : 0);
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    double xo = __dmul_rn(scale, 0.5);
    for (int x = 0; x < out_size; ++x) {
        int v = xo < 0.0 ? -1 : (int)xo;
        t[x] = v < in_size ? v : in_size - 1;          // (never taken for a whole-image box; keeps the gather in bounds)
        xo = __dadd_rn(xo, scale);
    }
}
__global__ void resize_nearest_u8_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, const int* __restrict__ tab, uint8_t* __restrict__ out,
                                         int Ho, int Wo, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo);
        const int yy = (int)((i / Wo) % Ho);
        const int64_t b = i / ((int64_t)Wo * Ho);
        out[i] = src[(b * Hs + tab[Wo + yy]) * Ws + tab[xx]];
    }
}
extern "C" int64_t cmu_resize_nearest_u8_ws_bytes(int Ho, int Wo) { return (int64_t)(Ho + Wo) * (int64_t)sizeof(int); }
extern "C" int cmu_resize_nearest_u8(const uint8_t* src, int B, int Hs, int Ws, uint8_t* out, int Ho, int Wo, void* ws, void* stream) {
    CMU_CHECK_ARG(src && out && ws && B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "cmu_resize_nearest_u8: bad args");
    hipLaunchKernelGGL(nearest_table_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, Hs, Ws, Ho, Wo, (int*)ws);
    CMU_CHECK_LAUNCH("cmu_resize_nearest_u8(table)");
    const int64_t t = (int64_t)B * Ho * Wo;
    const int g = (int)(cmu_div_up64(t, 256) < 16384 ? cmu_div_up64(t, 256) : 16384);
    hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, (const int*)ws, out, Ho, Wo, t);
    CMU_CHECK_LAUNCH("cmu_resize_nearest_u8");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// two views of a batch: ShiftPixel crops + GaussNoise on the shifted one
// ---------------------------------------------------------------------------------------------
__device__ static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// standard normal of element e: Box-Muller on the first two words of Philox(counter = (e_lo, e_hi, 0, 0), key = seed)
__device__ static inline double aug_normal(uint64_t e, uint64_t seed) {
    uint32_t r[4];
    philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u1 = ((double)r[0] + 0.5) * (1.0 / 4294967296.0), u2 = ((double)r[1] + 0.5) * (1.0 / 4294967296.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

__global__ __launch_bounds__(256) void two_view_kernel(const float* __restrict__ src, int S, const int* __restrict__ shifts,
                                                      const double* __restrict__ noise, uint64_t seed, float* __restrict__ img,
                                                      float* __restrict__ img_t, int out) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int dy = shifts[2 * b], dx = shifts[2 * b + 1];
    const float* s = src + (int64_t)b * S * S;
    const int n = out * out;
    float mx = -__builtin_inff();
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, s[(int64_t)(dy + i / out) * S + dx + i % out]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const double sigma = (double)__fdiv_rn(mx, 10.f);   // float32 max / 10 as numpy evaluates it, then promoted
    for (int i = threadIdx.x; i < n; i += 256) {
        const int y = i / out, x = i % out;
        const int64_t o = (int64_t)b * n + i;
        img[o] = s[(int64_t)y * S + x];
        const double z = noise ? noise[o] : aug_normal((uint64_t)o, seed);
        img_t[o] = (float)__dadd_rn((double)s[(int64_t)(dy + y) * S + dx + x], __dmul_rn(sigma, z));
    }
}
extern "C" int cmu_two_view(const float* src, int B, int S, const int* shifts, const double* noise, uint64_t seed, float* img, float* img_t,
                            int out, void* stream) {
    CMU_CHECK_ARG(src && shifts && img && img_t && B > 0 && S > 0 && out > 0 && out <= S, "cmu_two_view: bad args");
    // (shifts are device data: the host wrapper checks 0 <= dy, dx and dy + out <= S, dx + out <= S)
    hipLaunchKernelGGL(two_view_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, src, S, shifts, noise, seed, img, img_t, out);
    CMU_CHECK_LAUNCH("cmu_two_view");
    return CMU_OK;
}
// the generator alone (tests / callers that want the draws): out[i] = normal(element offset + i)
__global__ void philox_normal_kernel(double* __restrict__ out, int64_t n, uint64_t offset, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = aug_normal(offset + (uint64_t)i, seed);
}
extern "C" int cmu_philox_normal(double* out, int64_t n, uint64_t offset, uint64_t seed, void* stream) {
    CMU_CHECK_ARG(out && n > 0, "cmu_philox_normal: bad args");
    const int g = (int)(cmu_div_up64(n, 256) < 4096 ? cmu_div_up64(n, 256) : 4096);
    hipLaunchKernelGGL(philox_normal_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, out, n, offset, seed);
    CMU_CHECK_LAUNCH("cmu_philox_normal");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// random patch mask (backbones/UNet_encoder.py:106-139: per sample a random permutation of the patches, the first
// floor(ratio*H*W / patch^2) of it masked).  One workgroup per sample: patch p draws the 32-bit key Philox(offset + b*P + p)
// (first output word); it is masked iff (key, p) ranks among the n_mask smallest of the sample -- a uniformly random
// n_mask-subset, reproducible from (seed, offset), with no sort and no host loop.  mask (B,H,W) u8, 1 = masked.
// ---------------------------------------------------------------------------------------------
constexpr int PM_MAX_PATCHES = 4096;
__global__ __launch_bounds__(256) void patch_mask_kernel(uint8_t* __restrict__ mask, int H, int W, int patch, int n_mask, uint64_t seed,
                                                        uint64_t offset) {
    __shared__ unsigned long long key[PM_MAX_PATCHES];
    __shared__ uint8_t flag[PM_MAX_PATCHES];
    const int b = blockIdx.x, ph = H / patch, pw = W / patch, P = ph * pw;
    for (int q = threadIdx.x; q < P; q += 256) {
        uint32_t r[4];
        const uint64_t e = offset + (uint64_t)b * P + q;
        philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        key[q] = ((unsigned long long)r[0] << 32) | (unsigned)q;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += 256) {
        const unsigned long long k = key[q];
        int rank = 0;
        for (int j = 0; j < P; ++j) rank += key[j] < k ? 1 : 0;
        flag[q] = rank < n_mask ? 1 : 0;
    }
    __syncthreads();
    uint8_t* m = mask + (int64_t)b * H * W;
    for (int i = threadIdx.x; i < H * W; i += 256) m[i] = flag[(i / W / patch) * pw + (i % W) / patch];
}
extern "C" int cmu_random_patch_mask(uint8_t* mask, int B, int H, int W, int patch, int n_mask, uint64_t seed, uint64_t offset, void* stream) {
    CMU_CHECK_ARG(mask && B > 0 && H > 0 && W > 0 && patch > 0 && H % patch == 0 && W % patch == 0, "cmu_random_patch_mask: bad shape");
    const int P = (H / patch) * (W / patch);
    CMU_CHECK_ARG(P <= PM_MAX_PATCHES && n_mask >= 0 && n_mask <= P, "cmu_random_patch_mask: %d patches (max %d), n_mask %d", P, PM_MAX_PATCHES, n_mask);
    hipLaunchKernelGGL(patch_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mask, H, W, patch, n_mask, seed, offset);
    CMU_CHECK_LAUNCH("cmu_random_patch_mask");
    return CMU_OK;
}
