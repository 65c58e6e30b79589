// sparse.hip -- SparK-style sparse (masked) convolution support (reference: Pretraining/Spark/encoder.py:12-56,
// spark.py:88-131).  The reference runs every op densely and then multiplies by the up-sampled active mask, and
// normalises with BatchNorm statistics taken over ACTIVE positions only (gather -> BN1d -> scatter into zeros).
// Round-1 form on MI355X: the dense implicit-GEMM kernels are reused unchanged and the mask semantics live in
// three HBM-bound kernels (16-byte chunks per lane, fixed channel chunk per thread, slab reductions):
//   cmu_masked_channel_stats  sum / sum-of-squares per channel over selected pixels (sparse BN statistics,
//                             mask-token gradient)
//   cmu_mask_select           out = selected ? relu?(x*scale+shift) : fill[c]   (sparse BN apply + ReLU, densify with
//                             mask tokens, gradient masking)
//   cmu_spark_loss_fwd_bwd    per-patch normalised L2 on the NON-active patches (spark.py:115-123)
// `active` is the (B, f, f) uint8 patch map; a pixel (y, x) of a (H, W) level looks up active[b][y >> s][x >> s]
// with s = log2(H / f).  The tile-skipping MFMA variant is the next step.
#include "common.h"

constexpr int SP_ROWS = 1024;  // slab rows written by cmu_masked_channel_stats (one per workgroup, unused rows zero)


template <class TR>
__global__ __launch_bounds__(256) void masked_stats_kernel(const unsigned char* __restrict__ x, int64_t ldx,
                                                          const uint8_t* __restrict__ active, int f, int sbits, int invert,
                                                          float* __restrict__ slab, int B, int H, int W, int C, int cpb, int ppb,
                                                          const int* __restrict__ rows, const int* __restrict__ n_rows) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int nchunk = C / EPC;
    const int ch = blockIdx.y * cpb + tid % cpb;
    const int prow = tid / cpb;
    const bool on = prow < ppb && ch < nchunk;
    const int64_t npix = rows != nullptr ? (int64_t)*n_rows : (int64_t)B * H * W;     // sparse form: the list of active pixels
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    if (on)
        for (int64_t pp = (int64_t)blockIdx.x * ppb + prow; pp < npix; pp += (int64_t)gridDim.x * ppb) {
            int64_t p = pp;
            if (rows != nullptr) {
                p = rows[pp];
                if (p < 0) continue;
            } else {
                int xx, yy, b;
                cmu_pixel_coords(p, W, H, npix <= 0x7fffffffll, b, yy, xx);
                if (!sp_active(active, f, sbits, b, yy, xx, invert)) continue;
            }
            float v[EPC];
            TR::unpack(ld_global16(x + (p * ldx + ch * EPC) * ES), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s1[e] += v[e];
                s2[e] = fmaf(v[e], v[e], s2[e]);
            }
        }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        __syncthreads();
        red[tid] = s1[e];
        __syncthreads();
        float a = 0.f;
        if (tid < cpb)
            for (int k = 0; k < ppb; ++k) a += red[k * cpb + tid];
        __syncthreads();
        red[tid] = s2[e];
        __syncthreads();
        if (tid < cpb && blockIdx.y * cpb + tid < nchunk) {
            float q = 0.f;
            for (int k = 0; k < ppb; ++k) q += red[k * cpb + tid];
            const int c = (blockIdx.y * cpb + tid) * EPC + e;
            slab[((int64_t)blockIdx.x * 2 + 0) * C + c] = a;
            slab[((int64_t)blockIdx.x * 2 + 1) * C + c] = q;
        }
    }
}

static void sp_geometry(int nchunk, int* cpb, int* ppb, int* gy) {
    *cpb = nchunk < 256 ? nchunk : 256;
    *ppb = 256 / *cpb;
    *gy = cmu_div_up(nchunk, *cpb);
}

template <class TR>
static int masked_stats_t(const void* x, int64_t ldx, const uint8_t* active, int f, int sbits, int invert, float* slab, int B, int H, int W,
                          int C, hipStream_t st, const int* rows = nullptr, const int* n_rows = nullptr) {
    int cpb, ppb, gy;
    sp_geometry(C / TR::EPC, &cpb, &ppb, &gy);
    hipLaunchKernelGGL((masked_stats_kernel<TR>), dim3(SP_ROWS, gy), dim3(256), 0, st, (const unsigned char*)x, ldx, active, f, sbits, invert,
                       slab, B, H, W, C, cpb, ppb, rows, n_rows);
    CMU_CHECK_LAUNCH("cmu_masked_channel_stats");
    return CMU_OK;
}
extern "C" int cmu_masked_stats_rows(void) { return SP_ROWS; }
extern "C" int cmu_masked_channel_stats(const void* x, int64_t ldx, const uint8_t* active, int f, int invert, float* slab, int B, int H,
                                        int W, int C, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && active && slab && B > 0 && H > 0 && W > 0 && f > 0, "cmu_masked_channel_stats: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(C > 0 && C % epc == 0 && ldx % epc == 0 && ldx >= C && cmu_aligned16(x), "cmu_masked_channel_stats: C / ld alignment");
    const int sbits = sp_shift_bits(H, f);
    CMU_CHECK_ARG(sbits >= 0 && (f << sbits) == W, "cmu_masked_channel_stats: H=%d, W=%d must be f=%d times a power of two", H, W, f);
    CMU_DISPATCH_DT(dt, masked_stats_t, x, ldx, active, f, sbits, invert, slab, B, H, W, C, (hipStream_t)stream);
}

// the same statistics over a list of active pixels (cmu_sparse_pixel_list): only the listed pixels are visited
extern "C" int cmu_rows_channel_stats(const void* x, int64_t ldx, const int* rows, const int* n_rows, float* slab, int B, int H, int W, int C,
                                      int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && rows && n_rows && slab && B > 0 && H > 0 && W > 0, "cmu_rows_channel_stats: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(C > 0 && C % epc == 0 && ldx % epc == 0 && ldx >= C && cmu_aligned16(x), "cmu_rows_channel_stats: C / ld alignment");
    CMU_DISPATCH_DT(dt, masked_stats_t, x, ldx, (const uint8_t*)nullptr, 1, 0, 0, slab, B, H, W, C, (hipStream_t)stream, rows, n_rows);
}

template <class TR>
__global__ void mask_select_kernel(const unsigned char* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int relu, const uint8_t* __restrict__ active, int f, int sbits,
                                   int invert, const float* __restrict__ fill, unsigned char* __restrict__ out, int64_t ldo, int B, int H,
                                   int W, int C, int cpb, int ppb) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    const int tid = threadIdx.x;
    const int nchunk = C / EPC;
    const int ch = blockIdx.y * cpb + tid % cpb;
    const int prow = tid / cpb;
    if (!(prow < ppb && ch < nchunk)) return;
    const int64_t npix = (int64_t)B * H * W;
    float sc[EPC], sh[EPC], fl[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = scale ? scale[ch * EPC + e] : 1.f;
        sh[e] = scale ? shift[ch * EPC + e] : 0.f;
        fl[e] = fill ? fill[ch * EPC + e] : 0.f;
    }
    const u32x4 fillv = TR::pack(fl);
    const bool small = npix <= 0x7fffffffll;
    for (int64_t p = (int64_t)blockIdx.x * ppb + prow; p < npix; p += (int64_t)gridDim.x * ppb) {
        int xx, yy, b;
        cmu_pixel_coords(p, W, H, small, b, yy, xx);
        u32x4 o = fillv;
        if (sp_active(active, f, sbits, b, yy, xx, invert)) {
            o = ld_global16(x + (p * ldx + ch * EPC) * ES);
            if (scale || relu) {
                float v[EPC];
                TR::unpack(o, v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float t = fmaf(v[e], sc[e], sh[e]);
                    v[e] = relu ? fmaxf(t, 0.f) : t;
                }
                o = TR::pack(v);
            }
        }
        st_global16(out + (p * ldo + ch * EPC) * ES, o);
    }
}
template <class TR>
static int mask_select_t(const void* x, int64_t ldx, const float* scale, const float* shift, int relu, const uint8_t* active, int f, int sbits,
                         int invert, const float* fill, void* out, int64_t ldo, int B, int H, int W, int C, hipStream_t st) {
    int cpb, ppb, gy;
    sp_geometry(C / TR::EPC, &cpb, &ppb, &gy);
    const int64_t npix = (int64_t)B * H * W;
    // 2,048 workgroups looping, four pixel chunks apart (same-box sweep at bs 32, 512^2, ten launches per step: 4,096 x 2 -> 1.83 ms,
    // 2,048 x 4 -> 1.73, 8,192 x 2 -> 2.02, no cap 2.6 ... 6.8: unlike the BatchNorm-backward apply this pass is mostly stores)
    int gx = (int)(cmu_div_up64(npix, ppb * 4) < 2048 ? cmu_div_up64(npix, ppb * 4) : 2048);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((mask_select_kernel<TR>), dim3(gx, gy), dim3(256), 0, st, (const unsigned char*)x, ldx, scale, shift, relu, active, f,
                       sbits, invert, fill, (unsigned char*)out, ldo, B, H, W, C, cpb, ppb);
    CMU_CHECK_LAUNCH("cmu_mask_select");
    return CMU_OK;
}
extern "C" int cmu_mask_select(const void* x, int64_t ldx, const float* scale, const float* shift, int relu, const uint8_t* active, int f,
                               int invert, const float* fill, void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && out && active && B > 0 && H > 0 && W > 0 && f > 0, "cmu_mask_select: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(C > 0 && C % epc == 0 && ldx % epc == 0 && ldo % epc == 0 && ldx >= C && ldo >= C && cmu_aligned16(x) && cmu_aligned16(out),
                  "cmu_mask_select: C / ld alignment");
    CMU_CHECK_ARG((scale == nullptr) == (shift == nullptr), "cmu_mask_select: scale/shift must both be set");
    const int sbits = sp_shift_bits(H, f);
    CMU_CHECK_ARG(sbits >= 0 && (f << sbits) == W, "cmu_mask_select: H=%d, W=%d must be f=%d times a power of two", H, W, f);
    CMU_DISPATCH_DT(dt, mask_select_t, x, ldx, scale, shift, relu, active, f, sbits, invert, fill, out, ldo, B, H, W, C, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// SparK reconstruction loss (spark.py:112-123): patches of p x p pixels; target patch normalised with its own
// mean and UNBIASED variance ((x-mean)/sqrt(var+1e-6)); l2 = mean over the patch of (rec-target)^2; the loss is
// the sum of l2 over NON-active patches divided by (their number + 1e-8).  One workgroup per patch.
// ws: [B*f*f] l2 values + [B*f*f][2] (mean, rstd) + 2 floats.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spark_patch_kernel(const float* __restrict__ rec, const float* __restrict__ img,
                                                         const uint8_t* __restrict__ active, float* __restrict__ ws, int B, int f, int p,
                                                         int W) {
    __shared__ float red[4];
    const int patch = blockIdx.x;
    const int px = patch % f, py = (patch / f) % f, b = patch / (f * f);
    const int H = f * p;
    const int n = p * p;
    auto at = [&](const float* t, int i) { return t[((int64_t)b * H + py * p + i / p) * W + px * p + i % p]; };
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += at(img, i);
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n;
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = at(img, i) - mean;
        v = fmaf(d, d, v);
    }
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)(n - 1);
    const float rstd = 1.f / sqrtf(var + 1e-6f);
    float l = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = at(rec, i) - (at(img, i) - mean) * rstd;
        l = fmaf(d, d, l);
    }
    l = wave_sum(l);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int np = B * f * f;
        ws[patch] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
        ws[np + 2 * patch + 0] = mean;
        ws[np + 2 * patch + 1] = rstd;
    }
}
__global__ __launch_bounds__(256) void spark_final_kernel(float* __restrict__ ws, const uint8_t* __restrict__ active, int np, float* loss) {
    __shared__ double red[2][4];
    double num = 0.0, den = 0.0;
    for (int i = threadIdx.x; i < np; i += 256)
        if (!active[i]) {
            num += (double)ws[i];
            den += 1.0;
        }
    num = wave_sum_d(num);
    den = wave_sum_d(den);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = num;
        red[1][threadIdx.x >> 6] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        num = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        den = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        loss[0] = (float)(num / (den + 1e-8));
        ws[3 * np + 0] = (float)(den + 1e-8);
    }
}
__global__ void spark_grad_kernel(const float* __restrict__ rec, const float* __restrict__ img, const uint8_t* __restrict__ active,
                                  const float* __restrict__ ws, float* __restrict__ drec, float loss_scale, int B, int f, int p, int W,
                                  int64_t total) {
    const int np = B * f * f;
    const int H = f * p;
    const float k = 2.f * loss_scale / (ws[3 * np] * (float)(p * p));
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(o % W), y = (int)((o / W) % H), b = (int)(o / ((int64_t)W * H));
        const int patch = (b * f + y / p) * f + x / p;
        float g = 0.f;
        if (!active[patch]) g = k * (rec[o] - (img[o] - ws[np + 2 * patch]) * ws[np + 2 * patch + 1]);
        drec[o] = g;
    }
}
extern "C" int64_t cmu_spark_loss_ws_bytes(int B, int f) { return ((int64_t)3 * B * f * f + 4) * (int64_t)sizeof(float); }
extern "C" int cmu_spark_loss_fwd_bwd(const float* rec, const float* img, const uint8_t* active, float* loss, float* drec, float loss_scale,
                                      int B, int f, int p, void* ws, void* stream) {
    CMU_CHECK_ARG(rec && img && active && loss && ws && B > 0 && f > 0 && p > 1, "cmu_spark_loss_fwd_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int W = f * p, np = B * f * f;
    hipLaunchKernelGGL(spark_patch_kernel, dim3(np), dim3(256), 0, st, rec, img, active, (float*)ws, B, f, p, W);
    CMU_CHECK_LAUNCH("cmu_spark_loss(patch)");
    hipLaunchKernelGGL(spark_final_kernel, dim3(1), dim3(256), 0, st, (float*)ws, active, np, loss);
    CMU_CHECK_LAUNCH("cmu_spark_loss(final)");
    if (drec) {
        const int64_t total = (int64_t)B * W * W;
        const int grid = (int)(cmu_div_up64(total, 256) < 4096 ? cmu_div_up64(total, 256) : 4096);
        hipLaunchKernelGGL(spark_grad_kernel, dim3(grid), dim3(256), 0, st, rec, img, active, (const float*)ws, drec, loss_scale, B, f, p, W,
                           total);
        CMU_CHECK_LAUNCH("cmu_spark_loss(grad)");
    }
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// SparK.patchify / unpatchify (Spark/spark.py:133-148): (B, C, h*p, w*p) <-> (B, h*w, p*p*C), element (b, c, hy*p + py, wx*p + px) <->
// (b, hy*w + wx, (py*p + px)*C + c) -- the reference's einsum('bchpwq->bhwpqc') + reshape and its inverse as one gather pass (fp32).
// ---------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int h, int w, int p, int64_t total, int inverse) {
    const int H = h * p, W = w * p;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the (B, C, H, W) tensor (coalesced on that side; the patch side is strided by C)
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t b = i / ((int64_t)W * H * C);
        const int64_t j = ((b * h + y / p) * w + x / p) * (int64_t)(p * p * C) + ((y % p) * p + (x % p)) * C + c;
        if (inverse) dst[i] = src[j];
        else dst[j] = src[i];
    }
}
extern "C" int cmu_patchify(const float* src, float* dst, int B, int C, int h, int w, int p, int inverse, void* stream) {
    CMU_CHECK_ARG(src && dst && B > 0 && C > 0 && h > 0 && w > 0 && p > 0, "cmu_patchify: bad args");
    const int64_t total = (int64_t)B * C * h * p * w * p;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, h, w, p, total, inverse);
    CMU_CHECK_LAUNCH("cmu_patchify");
    return CMU_OK;
}
