// backward_elem.hip -- HBM-bound backward pieces of the UNet hot path (gfx950):
//   BatchNorm+ReLU backward (reduce + apply), MaxPool backward fused with the skip add,
//   1x1 head backward, first-layer (Cin=1) weight gradient.
// Pattern: each lane owns one 16-byte channel chunk (fixed per thread, so per-channel parameters and
// partial sums live in registers), pixels are walked grid-stride; per-block partial sums go to a slab
// that a second tiny kernel adds in a fixed order (bitwise reproducible, no float atomics).
#include "common.h"

constexpr int RED_MAX_BLOCKS = 2048;     // rows of the partial-sum slabs (workspace sizes)
// measured on the bench step: the max-pool backward is fastest with 2048 workgroups (1.22 ms; 1.34 at 1024, 1.38 at 8192), the
// head backward with 1024 (0.47 ms; 0.54 at 2048, 0.76 at 8192) -- each workgroup ends with a cross-wave fold of its sums
// (round 4: with the per-workgroup fold inside the waves -- fold_rows -- the A/B was repeated, see the macros' defaults)
#ifndef CMU_POOLB_BLOCKS
#define CMU_POOLB_BLOCKS 2048
#endif
#ifndef CMU_HEADB_BLOCKS
#define CMU_HEADB_BLOCKS 768
#endif
#ifndef CMU_POOLA_PPT
#define CMU_POOLA_PPT 0      // > 0: the apply form (MODE 2, no sums) on an uncapped grid with that many pooled pixels per thread -- measured SLOWER than the capped grid-stride form (4 levels of the bench step: 1.15 ms capped, 1.24 at 4, 1.48 at 2, 2.12 at 1)
#endif
constexpr int POOLB_MAX_BLOCKS = CMU_POOLB_BLOCKS, HEADB_MAX_BLOCKS = CMU_HEADB_BLOCKS;
// BN-backward partial-sum slab ("bn_ws"): int32 header word 0 = number of rows written by the producer kernel
// (device side, no host sync), then float rows [row][2][C] from byte 16 on.  Producers: cmu_bn_bwd_reduce,
// cmu_maxpool_bwd, cmu_conv1x1_head_bwd (fused); consumer: bn_bwd_final_kernel.
constexpr int BNWS_HDR = 16;

// block-level sum of `v` over the threads that share `key = tid % cpb` (prow = tid / cpb < ppb).
// red: LDS float[256].  Result valid for tid < cpb.  All 256 threads must call.
__device__ static inline float sum_over_rows(float v, float* red, int tid, int cpb, int ppb) {
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    float a = 0.f;
    if (tid < cpb)
        for (int k = 0; k < ppb; ++k) a += red[k * cpb + tid];
    return a;
}

// The same for NV values at once (round 4): threads that share a channel chunk sit cpb lanes apart, so for cpb <= 64 (a power of
// two) the fold runs inside the wave first (xor shuffles in a fixed order) and only the four waves' partial sums cross LDS --
// one barrier pair per group of FOLD_G values instead of one pair per value and a serial ppb-term sum by cpb threads (the head
// backward folded 34 values per workgroup through 68 barriers: a tenth of its time).  cpb > 64 (cpb = 128 / 256: ppb = 2 / 1):
// the rows meet in LDS only.  red: LDS float[FOLD_G * 256].  Results valid for tid < cpb.  All 256 threads must call; needs
// prow < ppb for every thread (256 % cpb == 0), which the callers' chunk geometry gives for every power-of-two chunk count.
constexpr int FOLD_G = 8;
template <int NV>
__device__ static inline void fold_rows(float (&v)[NV], float* red, int tid, int cpb, int ppb) {
    const int lane = tid & 63, wv = tid >> 6;
    if (cpb <= 64) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            for (int o = cpb; o < 64; o <<= 1) v[i] += __shfl_xor(v[i], o, 64);
    }
#pragma unroll
    for (int g0 = 0; g0 < NV; g0 += FOLD_G) {
        __syncthreads();
        if (cpb <= 64) {
            if (lane < cpb) {
#pragma unroll
                for (int i = g0; i < g0 + FOLD_G && i < NV; ++i) red[((i - g0) * 4 + wv) * 64 + lane] = v[i];
            }
        } else {
#pragma unroll
            for (int i = g0; i < g0 + FOLD_G && i < NV; ++i) red[(i - g0) * 256 + tid] = v[i];
        }
        __syncthreads();
        if (tid < cpb) {
#pragma unroll
            for (int i = g0; i < g0 + FOLD_G && i < NV; ++i) {
                float a = 0.f;
                if (cpb <= 64) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) a += red[((i - g0) * 4 + w) * 64 + tid];
                } else {
                    for (int k = 0; k < ppb; ++k) a += red[(i - g0) * 256 + k * cpb + tid];
                }
                v[i] = a;
            }
        }
    }
}
__device__ static inline bool fold_ok(int cpb) { return (cpb & (cpb - 1)) == 0; }

// ---------------------------------------------------------------------------------------------
// BatchNorm2d + ReLU backward (autograd of model.py:18-19, 21-22)
// ---------------------------------------------------------------------------------------------
template <class TR>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const unsigned char* __restrict__ dA, int64_t ldd,
                                                           const unsigned char* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ ws, int64_t npix, int C, int cpb, int ppb,
                                                           const uint8_t* __restrict__ amask, int f, int sbits, int H, int W,
                                                           const int* __restrict__ rows, const int* __restrict__ n_rows) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    __shared__ float red[FOLD_G * 256];
    const int tid = threadIdx.x;
    if (rows != nullptr) npix = *n_rows;      // sparse form: the loop runs over the list of active pixels only
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *reinterpret_cast<int*>(ws) = (int)gridDim.x;
    ws += BNWS_HDR / 4;
    const int nchunk = C / EPC;
    const int ch = blockIdx.y * cpb + tid % cpb;
    const int prow = tid / cpb;
    const bool active = prow < ppb && ch < nchunk;
    float sc[EPC], sh[EPC], mu[EPC], is[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = active ? ch * EPC + e : 0;
        sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
        s1[e] = s2[e] = 0.f;
    }
    if (active)
        for (int64_t pp = (int64_t)blockIdx.x * ppb + prow; pp < npix; pp += (int64_t)gridDim.x * ppb) {
            int64_t p = pp;
            if (rows != nullptr) {
                p = rows[pp];
                if (p < 0) continue;
            } else if (amask) {
                int b_, y_, x_;
                cmu_pixel_coords(p, W, H, npix <= 0x7fffffffll, b_, y_, x_);
                if (!sp_active(amask, f, sbits, b_, y_, x_, 0)) continue;
            }
            float g[EPC], v[EPC];
            TR::unpack(ld_global16(dA + (p * ldd + ch * EPC) * ES), g);
            TR::unpack(ld_global16(y + (p * ldy + ch * EPC) * ES), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float dz = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
                s1[e] += dz;
                s2[e] = fmaf(dz, (v[e] - mu[e]) * is[e], s2[e]);
            }
        }
    if (fold_ok(cpb)) {
        float t[2 * EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { t[e] = s1[e]; t[EPC + e] = s2[e]; }
        fold_rows<2 * EPC>(t, red, tid, cpb, ppb);
        if (tid < cpb && blockIdx.y * cpb + tid < nchunk) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int c = (blockIdx.y * cpb + tid) * EPC + e;
                ws[((int64_t)blockIdx.x * 2 + 0) * C + c] = t[e];
                ws[((int64_t)blockIdx.x * 2 + 1) * C + c] = t[EPC + e];
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float a = sum_over_rows(s1[e], red, tid, cpb, ppb);
        const float b = sum_over_rows(s2[e], red, tid, cpb, ppb);
        if (tid < cpb && blockIdx.y * cpb + tid < nchunk) {
            const int c = (blockIdx.y * cpb + tid) * EPC + e;
            ws[((int64_t)blockIdx.x * 2 + 0) * C + c] = a;
            ws[((int64_t)blockIdx.x * 2 + 1) * C + c] = b;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ ws, double count, float* dgamma,
                                                          float* dbeta, float* coef, int C) {
    __shared__ double red[2][16][16];
    const int nblocks = *reinterpret_cast<const int*>(ws);
    ws += BNWS_HDR / 4;
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        // eight rows per trip, loaded before they are added: the row loop is a chain of ~1 us round trips otherwise (54 us per
        // launch for 2,048 rows); the order of the additions stays fixed
        constexpr int U = 8;
        int b = part;
        for (; b + 16 * (U - 1) < nblocks; b += 16 * U) {
            float a[U], q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a[u] = ws[((int64_t)(b + 16 * u) * 2 + 0) * C + c];
                q[u] = ws[((int64_t)(b + 16 * u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s1 += (double)a[u];
                s2 += (double)q[u];
            }
        }
        for (; b < nblocks; b += 16) {
            s1 += (double)ws[((int64_t)b * 2 + 0) * C + c];
            s2 += (double)ws[((int64_t)b * 2 + 1) * C + c];
        }
    }
    red[0][part][cl] = s1;
    red[1][part][cl] = s2;
    __syncthreads();
    if (part != 0 || c >= C) return;
    s1 = 0.0; s2 = 0.0;
    for (int q = 0; q < 16; ++q) { s1 += red[0][q][cl]; s2 += red[1][q][cl]; }
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / count);
    coef[C + c] = (float)(s2 / count);
}

static void chunk_geometry(int nchunk, int* cpb, int* ppb, int* gy) {
    *cpb = nchunk < 256 ? nchunk : 256;
    *ppb = 256 / *cpb;
    *gy = cmu_div_up(nchunk, *cpb);
}

template <class TR>
static int bn_bwd_reduce_t(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                           const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef, int B, int H, int W,
                           int C, void* ws, const uint8_t* active, int f, int64_t count, hipStream_t st, const int* rows = nullptr,
                           const int* n_rows = nullptr, int64_t max_rows = 0) {
    int cpb, ppb, gy;
    chunk_geometry(C / TR::EPC, &cpb, &ppb, &gy);
    const int64_t npix = (int64_t)B * H * W;
    const int sbits = active ? sp_shift_bits(H, f) : 0;
    const int64_t nloop = rows ? max_rows : npix;
    int gx = (int)(cmu_div_up64(nloop, ppb * 4) < RED_MAX_BLOCKS ? cmu_div_up64(nloop, ppb * 4) : RED_MAX_BLOCKS);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<TR>), dim3(gx, gy), dim3(256), 0, st, (const unsigned char*)dA, ldd,
                       (const unsigned char*)y, ldy, scale, shift, mean, invstd, (float*)ws, npix, C, cpb, ppb, active, f, sbits, H, W, rows,
                       n_rows);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cmu_div_up(C, 16)), dim3(256), 0, st, (const float*)ws, (double)((active || rows) ? count : npix), dgamma,
                       dbeta, coef, C);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_reduce(final)");
    return CMU_OK;
}

static int check_pair(const char* name, const void* a, int64_t lda, const void* b, int64_t ldb, int C, int dt) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0, "%s: bad dtype %d", name, dt);
    const int epc = 16 / es;
    CMU_CHECK_ARG(a && b && cmu_aligned16(a) && cmu_aligned16(b), "%s: null / unaligned tensor", name);
    CMU_CHECK_ARG(C > 0 && C % epc == 0 && lda % epc == 0 && ldb % epc == 0 && lda >= C && ldb >= C, "%s: C=%d / ld alignment (multiple of %d)", name, C, epc);
    return CMU_OK;
}

extern "C" int64_t cmu_bn_bwd_ws_bytes(int C) { return BNWS_HDR + (int64_t)RED_MAX_BLOCKS * 2 * C * (int64_t)sizeof(float); }

extern "C" int cmu_bn_bwd_finalize(const void* bn_ws, int64_t count, float* dgamma, float* dbeta, float* coef, int C, void* stream) {
    CMU_CHECK_ARG(bn_ws && coef && C > 0 && count > 0, "cmu_bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cmu_div_up(C, 16)), dim3(256), 0, (hipStream_t)stream, (const float*)bn_ws, (double)count,
                       dgamma, dbeta, coef, C);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_finalize");
    return CMU_OK;
}

extern "C" int cmu_bn_bwd_reduce(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                                 const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef, int B,
                                 int H, int W, int C, int dt, void* ws, void* stream) {
    int rc;
    if ((rc = check_pair("cmu_bn_bwd_reduce", dA, ldd, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && ws && B > 0 && H > 0 && W > 0, "cmu_bn_bwd_reduce: null argument");
    CMU_DISPATCH_DT(dt, bn_bwd_reduce_t, dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, dgamma, dbeta, coef, B, H, W, C, ws,
                    (const uint8_t*)nullptr, 0, (int64_t)0, (hipStream_t)stream);
}
extern "C" int cmu_bn_bwd_reduce_masked(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                                        const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                                        const uint8_t* active, int f, int64_t count, int B, int H, int W, int C, int dt, void* ws,
                                        void* stream) {
    int rc;
    if ((rc = check_pair("cmu_bn_bwd_reduce_masked", dA, ldd, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && ws && active && count > 0 && B > 0, "cmu_bn_bwd_reduce_masked: null argument");
    CMU_CHECK_ARG(sp_shift_bits(H, f) >= 0 && (f << sp_shift_bits(H, f)) == W, "cmu_bn_bwd_reduce_masked: H,W must be f times a power of two");
    CMU_DISPATCH_DT(dt, bn_bwd_reduce_t, dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, dgamma, dbeta, coef, B, H, W, C, ws, active, f,
                    count, (hipStream_t)stream);
}
// the same over a list of active pixels (cmu_sparse_pixel_list): visits the listed pixels only
extern "C" int cmu_bn_bwd_reduce_rows(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                                      const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                                      const int* rows, const int* n_rows, int64_t max_rows, int64_t count, int B, int H, int W, int C, int dt,
                                      void* ws, void* stream) {
    int rc;
    if ((rc = check_pair("cmu_bn_bwd_reduce_rows", dA, ldd, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && ws && rows && n_rows && max_rows > 0 && count > 0 && B > 0,
                  "cmu_bn_bwd_reduce_rows: null argument");
    CMU_DISPATCH_DT(dt, bn_bwd_reduce_t, dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, dgamma, dbeta, coef, B, H, W, C, ws,
                    (const uint8_t*)nullptr, 0, count, (hipStream_t)stream, rows, n_rows, max_rows);
}

#ifndef CMU_APPLY_PPT
#define CMU_APPLY_PPT 4
#endif
template <class TR>
__global__ void bn_bwd_apply_kernel(const unsigned char* __restrict__ dA, int64_t ldd, const unsigned char* __restrict__ y,
                                    int64_t ldy, const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ coef,
                                    unsigned char* __restrict__ dY, int64_t ldo, int64_t npix, int C, int cpb, int ppb,
                                    const uint8_t* __restrict__ amask, int f, int sbits, int H, int W,
                                    const float* __restrict__ hd_dlogits = nullptr, const float* __restrict__ hd_w = nullptr, int hd_K = 2) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    const int tid = threadIdx.x;
    const int nchunk = C / EPC;
    const int ch = blockIdx.y * cpb + tid % cpb;
    const int prow = tid / cpb;
    if (!(prow < ppb && ch < nchunk)) return;
    float sc[EPC], sh[EPC], mu[EPC], is[EPC], c1[EPC], c2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = ch * EPC + e;
        sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
        c1[e] = coef[c]; c2[e] = coef[C + c];
    }
    // head mode (hd_dlogits != NULL, one or two classes): dA is not in memory -- it is the rank-K product dlogits[p][0..K) * w[0..K)[c]
    // of the 1x1 head this layer feeds, recomputed here and rounded as the head backward would have stored it (K = 1: SparK's
    // one-channel reconstruction head, decoder.py:47 -- it ran on the generic kernel at half this one's rate until round 4)
    float w0[EPC], w1[EPC];
    if (hd_dlogits != nullptr) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { w0[e] = hd_w[ch * EPC + e]; w1[e] = hd_K > 1 ? hd_w[C + ch * EPC + e] : 0.f; }
    }
    const unsigned HWu = (unsigned)H * (unsigned)W;
    // a workgroup's CMU_APPLY_PPT chunks per thread are ADJACENT pixel groups (one contiguous range per workgroup) rather than one group
    // in each quarter of the tensor: 3.03 -> 2.95 ms for the 17 launches of a bench step
    // (the launcher caps the grid at 2^20 workgroups: ranges beyond it are taken grid-stride)
    for (int64_t rng = (int64_t)blockIdx.x * (ppb * CMU_APPLY_PPT); rng < npix; rng += (int64_t)gridDim.x * (ppb * CMU_APPLY_PPT))
    for (int64_t p = rng + prow; p < npix && p < rng + (int64_t)ppb * CMU_APPLY_PPT; p += ppb) {
        if (amask) {
            int b_, y_, x_;
            cmu_pixel_coords(p, W, H, npix <= 0x7fffffffll, b_, y_, x_);
            if (!sp_active(amask, f, sbits, b_, y_, x_, 0)) {
                st_global16(dY + (p * ldo + ch * EPC) * ES, u32x4{0u, 0u, 0u, 0u});   // sparse BN: no gradient at masked positions
                continue;
            }
        }
        float g[EPC], v[EPC], o[EPC];
        if (hd_dlogits != nullptr) {
            const unsigned bq = (unsigned)p / HWu, r = (unsigned)p - bq * HWu;      // (the launcher keeps npix below 2^31 here)
            const float d0 = hd_dlogits[(int64_t)(hd_K * bq) * HWu + r], d1 = hd_K > 1 ? hd_dlogits[(int64_t)(2 * bq + 1) * HWu + r] : 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = fmaf(d1, w1[e], fmaf(d0, w0[e], 0.f));
            TR::unpack(TR::pack(o), g);
        } else {
            TR::unpack(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dA + (p * ldd + ch * EPC) * ES)), g);
        }
        TR::unpack(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + (p * ldy + ch * EPC) * ES)), v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float dz = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
            const float xh = (v[e] - mu[e]) * is[e];
            o[e] = sc[e] * (dz - c1[e] - xh * c2[e]);
        }
        __builtin_nontemporal_store(TR::pack(o), reinterpret_cast<u32x4*>(dY + (p * ldo + ch * EPC) * ES));
    }
}
template <class TR>
static int bn_bwd_apply_t(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* coef, void* dY, int64_t ldo, int B, int H, int W,
                          int C, const uint8_t* active, int f, hipStream_t st, const float* hd_dlogits = nullptr,
                          const float* hd_w = nullptr, int hd_K = 2) {
    int cpb, ppb, gy;
    chunk_geometry(C / TR::EPC, &cpb, &ppb, &gy);
    const int64_t npix = (int64_t)B * H * W;
    const int sbits = active ? sp_shift_bits(H, f) : 0;
    // four pixel chunks per thread, no grid cap: 3.4 ms per bench step against 4.1 with 4,096 workgroups looping 64 times (A/B: 1 chunk
    // per thread 4.6 ms, 2: 3.5, 8: 3.6)
    int gx = (int)(cmu_div_up64(npix, ppb * CMU_APPLY_PPT) < (1 << 20) ? cmu_div_up64(npix, ppb * CMU_APPLY_PPT) : (1 << 20));
#ifndef CMU_HEADA_CAP
#define CMU_HEADA_CAP 2048
#endif
    // head mode writes as much as it reads (dA is recomputed from dlogits): like the select pass of sparse.hip it runs best on a capped
    // grid looping -- 0.650 ms uncapped, 0.483 / 0.460 / 0.466 with 2,048 / 1,024 / 3,072 workgroups (0.75 with 512), tools/elem_bench.py
    if (hd_dlogits != nullptr && gx > CMU_HEADA_CAP) gx = CMU_HEADA_CAP;
#ifdef CMU_APPLY_CAP
    if (hd_dlogits == nullptr && gx > CMU_APPLY_CAP) gx = CMU_APPLY_CAP;
#endif
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TR>), dim3(gx, gy), dim3(256), 0, st, (const unsigned char*)dA, ldd,
                       (const unsigned char*)y, ldy, scale, shift, mean, invstd, coef, (unsigned char*)dY, ldo, npix, C, cpb, ppb, active, f,
                       sbits, H, W, hd_dlogits, hd_w, hd_K);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_apply");
    return CMU_OK;
}
extern "C" int cmu_bn_bwd_apply(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                                const float* save_mean, const float* save_invstd, const float* coef, void* dY, int64_t ldo, int B,
                                int H, int W, int C, int dt, void* stream) {
    int rc;
    if ((rc = check_pair("cmu_bn_bwd_apply", dA, ldd, y, ldy, C, dt))) return rc;
    if ((rc = check_pair("cmu_bn_bwd_apply(dY)", dY, ldo, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && B > 0 && H > 0 && W > 0, "cmu_bn_bwd_apply: null argument");
    CMU_DISPATCH_DT(dt, bn_bwd_apply_t, dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, coef, dY, ldo, B, H, W, C,
                    (const uint8_t*)nullptr, 0, (hipStream_t)stream);
}
extern "C" int cmu_bn_bwd_apply_masked(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                                       const float* save_mean, const float* save_invstd, const float* coef, void* dY, int64_t ldo,
                                       const uint8_t* active, int f, int B, int H, int W, int C, int dt, void* stream) {
    int rc;
    if ((rc = check_pair("cmu_bn_bwd_apply_masked", dA, ldd, y, ldy, C, dt))) return rc;
    if ((rc = check_pair("cmu_bn_bwd_apply_masked(dY)", dY, ldo, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && active && B > 0, "cmu_bn_bwd_apply_masked: null argument");
    CMU_CHECK_ARG(sp_shift_bits(H, f) >= 0 && (f << sp_shift_bits(H, f)) == W, "cmu_bn_bwd_apply_masked: H,W must be f times a power of two");
    CMU_DISPATCH_DT(dt, bn_bwd_apply_t, dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, coef, dY, ldo, B, H, W, C, active, f,
                    (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(2) backward + skip-gradient add (autograd of model.py:42-45 where both outputs are used)
// ---------------------------------------------------------------------------------------------
// MODE 0: dA is written (+ the BatchNorm-backward sums when bn_ws is given).  Round 3: when the gradient flows straight into this
// layer's BatchNorm backward (the UNet encoder: always), dA is never stored -- MODE 1 leaves only the sums (same values, rounded to the
// storage type as they would have been stored), and after their finalisation MODE 2 recomputes the pooled / skip sum and writes
// dY = scale * (gate * dA - c1 - xhat * c2) (cmu_bn_bwd_apply's arithmetic: bit-identical to MODE 0 + cmu_bn_bwd_apply).  Per element
// of the layer: 2.25 reads + 2.25 reads + 1 write instead of (2.25 reads + 1 write) + (2 reads + 1 write).
template <class TR, int MODE>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ dP, int64_t ldp,
                                                         const unsigned char* __restrict__ dS, int64_t lds,
                                                         const unsigned char* __restrict__ dS2, int64_t lds2,
                                                         const unsigned char* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         unsigned char* __restrict__ dA, int64_t lda, int B, int H, int W, int C,
                                                         int cpb, int ppb, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float* __restrict__ bn_ws,
                                                         const float* __restrict__ coef, const uint8_t* __restrict__ amask = nullptr,
                                                         int mf = 0, int msbits = 0) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    __shared__ float red[MODE == 2 ? 1 : FOLD_G * 256];
    const int tid = threadIdx.x;
    const int nchunk = C / EPC;
    const int ch = blockIdx.y * cpb + tid % cpb;
    const int prow = tid / cpb;
    const bool active = prow < ppb && ch < nchunk;
    const int Ho = H / 2, Wo = W / 2;
    const int64_t npool = (int64_t)B * Ho * Wo;
    float sc[EPC], sh[EPC], mu[EPC], is[EPC], s1[EPC], s2[EPC], c1[EPC], c2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = active ? ch * EPC + e : 0;
        sc[e] = scale[c];
        sh[e] = shift[c];
        mu[e] = (bn_ws || MODE == 2) ? mean[c] : 0.f;
        is[e] = (bn_ws || MODE == 2) ? invstd[c] : 0.f;
        c1[e] = MODE == 2 ? coef[c] : 0.f;
        c2[e] = MODE == 2 ? coef[C + c] : 0.f;
        s1[e] = s2[e] = 0.f;
    }
    // MODE 2 (no sums, uncapped grid): a workgroup's pooled pixels are one contiguous range (cmu_bn_bwd_apply's finding)
    constexpr int PPT = (MODE == 2 && CMU_POOLA_PPT > 0) ? CMU_POOLA_PPT : 0;
    const int64_t span = PPT > 0 ? (int64_t)ppb * PPT : (int64_t)gridDim.x * ppb;          // distance between a thread's ranges
    if (active)
        for (int64_t rng = (int64_t)blockIdx.x * (PPT > 0 ? span : ppb); rng < npool; rng += (PPT > 0 ? (int64_t)gridDim.x * span : npool))
        for (int64_t pp = rng + prow; pp < npool && (PPT == 0 || pp < rng + span); pp += (PPT > 0 ? ppb : span)) {
            int xo, yo, b;
            cmu_pixel_coords(pp, Wo, Ho, npool <= 0x7fffffffll, b, yo, xo);
            // SparK's sparse encoder: a masked window (see bnrelu_maxpool_kernel) has no gradient -- its dA is left unwritten (the
            // masked BatchNorm-backward passes that follow never read masked positions and write zeros there)
            if (amask != nullptr && !sp_active(amask, mf, msbits, b, 2 * yo, 2 * xo, 0)) continue;
            float best[EPC], g[EPC], f[4][EPC];
            int arg[EPC];
            TR::unpack(ld_global16_nt(dP + (pp * ldp + ch * EPC) * ES), g);
            int64_t src[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                src[q] = ((int64_t)b * H + 2 * yo + (q >> 1)) * W + 2 * xo + (q & 1);
                TR::unpack(ld_global16(y + (src[q] * ldy + ch * EPC) * ES), f[q]);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float a = fmaxf(fmaf(f[q][e], sc[e], sh[e]), 0.f);
                    if (q == 0 || a > best[e]) { best[e] = a; arg[e] = q; }   // first maximum wins (ATen max_pool2d)
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float d[EPC];
                if (dS) TR::unpack(ld_global16_nt(dS + (src[q] * lds + ch * EPC) * ES), d);
                else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) d[e] = 0.f;
                }
                if (dS2) {   // a second consumer of the skip (the joint model's two decoders): summed here in fp32, no add pass
                    float d2[EPC];
                    TR::unpack(ld_global16_nt(dS2 + (src[q] * lds2 + ch * EPC) * ES), d2);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) d[e] += d2[e];
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) d[e] += (arg[e] == q) ? g[e] : 0.f;
                const u32x4 packed = TR::pack(d);
                if (MODE == 0) st_global16(dA + (src[q] * lda + ch * EPC) * ES, packed);
                if (MODE == 2) {   // dY straight from the recomputed dA (rounded as MODE 0 would have stored it)
                    float dr[EPC], o[EPC];
                    TR::unpack(packed, dr);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float dz = fmaf(f[q][e], sc[e], sh[e]) > 0.f ? dr[e] : 0.f;
                        const float xh = (f[q][e] - mu[e]) * is[e];
                        o[e] = sc[e] * (dz - c1[e] - xh * c2[e]);
                    }
                    __builtin_nontemporal_store(TR::pack(o), reinterpret_cast<u32x4*>(dA + (src[q] * lda + ch * EPC) * ES));
                } else if (bn_ws) {   // BatchNorm+ReLU backward statistics of this layer, on the values as stored
                    float dr[EPC];
                    TR::unpack(packed, dr);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float dz = fmaf(f[q][e], sc[e], sh[e]) > 0.f ? dr[e] : 0.f;
                        s1[e] += dz;
                        s2[e] = fmaf(dz, (f[q][e] - mu[e]) * is[e], s2[e]);
                    }
                }
            }
        }
    if (MODE == 2 || bn_ws == nullptr) return;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *reinterpret_cast<int*>(bn_ws) = (int)gridDim.x;
    float* ws = bn_ws + BNWS_HDR / 4;
    if (fold_ok(cpb)) {
        float t[2 * EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { t[e] = s1[e]; t[EPC + e] = s2[e]; }
        fold_rows<2 * EPC>(t, red, tid, cpb, ppb);
        if (tid < cpb && blockIdx.y * cpb + tid < nchunk) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int c = (blockIdx.y * cpb + tid) * EPC + e;
                ws[((int64_t)blockIdx.x * 2 + 0) * C + c] = t[e];
                ws[((int64_t)blockIdx.x * 2 + 1) * C + c] = t[EPC + e];
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float a = sum_over_rows(s1[e], red, tid, cpb, ppb);
        const float bsum = sum_over_rows(s2[e], red, tid, cpb, ppb);
        if (tid < cpb && blockIdx.y * cpb + tid < nchunk) {
            const int c = (blockIdx.y * cpb + tid) * EPC + e;
            ws[((int64_t)blockIdx.x * 2 + 0) * C + c] = a;
            ws[((int64_t)blockIdx.x * 2 + 1) * C + c] = bsum;
        }
    }
}
template <class TR>
static int maxpool_bwd_t(const void* dP, int64_t ldp, const void* dS, int64_t lds, const void* dS2, int64_t lds2, const void* y, int64_t ldy, const float* scale,
                         const float* shift, void* dA, int64_t lda, int B, int H, int W, int C, const float* mean, const float* invstd,
                         void* bn_ws, hipStream_t st, const float* coef = nullptr, const uint8_t* amask = nullptr, int mf = 0) {
    int cpb, ppb, gy;
    chunk_geometry(C / TR::EPC, &cpb, &ppb, &gy);
    const int64_t npool = (int64_t)B * (H / 2) * (W / 2);
    int gx = (int)(cmu_div_up64(npool, ppb * 2) < POOLB_MAX_BLOCKS ? cmu_div_up64(npool, ppb * 2) : POOLB_MAX_BLOCKS);
    if (coef != nullptr && CMU_POOLA_PPT > 0) {
        const int64_t nb = cmu_div_up64(npool, (int64_t)ppb * (CMU_POOLA_PPT > 0 ? CMU_POOLA_PPT : 1));
        gx = (int)(nb < (1 << 20) ? nb : (1 << 20));
    }
    if (gx < 1) gx = 1;
#define CMU_POOLB_LAUNCH(MODE_)                                                                                                            \
    hipLaunchKernelGGL((maxpool_bwd_kernel<TR, MODE_>), dim3(gx, gy), dim3(256), 0, st, (const unsigned char*)dP, ldp, (const unsigned char*)dS, lds, \
                       (const unsigned char*)dS2, lds2, (const unsigned char*)y, ldy, scale, shift, (unsigned char*)dA, lda, B, H, W, C, cpb, ppb, \
                       mean, invstd, (float*)bn_ws, coef, amask, mf, amask ? sp_shift_bits(H, mf) : 0)
    if (coef != nullptr) CMU_POOLB_LAUNCH(2);
    else if (dA == nullptr) CMU_POOLB_LAUNCH(1);
    else CMU_POOLB_LAUNCH(0);
#undef CMU_POOLB_LAUNCH
    CMU_CHECK_LAUNCH("cmu_maxpool_bwd");
    return CMU_OK;
}
extern "C" int cmu_maxpool_bwd2(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* dSkip2, int64_t lds2, const void* y,
                                int64_t ldy, const float* scale, const float* shift, void* dA, int64_t lda, const float* save_mean,
                                const float* save_invstd, void* bn_ws, int B, int H, int W, int C, int dt, void* stream) {
    CMU_CHECK_ARG(bn_ws == nullptr || (save_mean && save_invstd), "cmu_maxpool_bwd: fused BN statistics need save_mean / save_invstd");
    CMU_CHECK_ARG(dSkip2 == nullptr || dSkip != nullptr, "cmu_maxpool_bwd2: dSkip2 without dSkip");
    int rc;
    if ((rc = check_pair("cmu_maxpool_bwd(dP,y)", dP, ldp, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(dA != nullptr || bn_ws != nullptr, "cmu_maxpool_bwd: dA may only be NULL in the statistics-only form (bn_ws given)");
    if (dA && (rc = check_pair("cmu_maxpool_bwd(dA,y)", dA, lda, y, ldy, C, dt))) return rc;
    if (dSkip && (rc = check_pair("cmu_maxpool_bwd(dSkip,y)", dSkip, lds, y, ldy, C, dt))) return rc;
    if (dSkip2 && (rc = check_pair("cmu_maxpool_bwd2(dSkip2,y)", dSkip2, lds2, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "cmu_maxpool_bwd: bad dims (%d,%d)", H, W);
    CMU_DISPATCH_DT(dt, maxpool_bwd_t, dP, ldp, dSkip, lds, dSkip2, lds2, y, ldy, scale, shift, dA, lda, B, H, W, C, save_mean, save_invstd,
                    bn_ws, (hipStream_t)stream);
}
extern "C" int cmu_maxpool_bwd_masked(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* y, int64_t ldy, const float* scale,
                                      const float* shift, const uint8_t* active, int f, void* dA, int64_t lda, int B, int H, int W, int C, int dt,
                                      void* stream) {
    CMU_CHECK_ARG(active && f > 0 && dA && scale && shift, "cmu_maxpool_bwd_masked: null argument");
    int rc;
    if ((rc = check_pair("cmu_maxpool_bwd_masked(dP,y)", dP, ldp, y, ldy, C, dt))) return rc;
    if ((rc = check_pair("cmu_maxpool_bwd_masked(dA,y)", dA, lda, y, ldy, C, dt))) return rc;
    if (dSkip && (rc = check_pair("cmu_maxpool_bwd_masked(dSkip,y)", dSkip, lds, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "cmu_maxpool_bwd_masked: bad dims (%d,%d)", H, W);
    const int sbits = sp_shift_bits(H, f);
    CMU_CHECK_ARG(sbits >= 1 && (f << sbits) == W, "cmu_maxpool_bwd_masked: H=%d, W=%d must be f=%d times a power of two >= 2", H, W, f);
    CMU_DISPATCH_DT(dt, maxpool_bwd_t, dP, ldp, dSkip, lds, nullptr, 0, y, ldy, scale, shift, dA, lda, B, H, W, C, nullptr, nullptr, nullptr,
                    (hipStream_t)stream, nullptr, active, f);
}
extern "C" int cmu_maxpool_bwd_apply(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* dSkip2, int64_t lds2, const void* y,
                                     int64_t ldy, const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                                     const float* coef, void* dY, int64_t ldo, int B, int H, int W, int C, int dt, void* stream) {
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && dY, "cmu_maxpool_bwd_apply: null argument");
    CMU_CHECK_ARG(dSkip2 == nullptr || dSkip != nullptr, "cmu_maxpool_bwd_apply: dSkip2 without dSkip");
    int rc;
    if ((rc = check_pair("cmu_maxpool_bwd_apply(dP,y)", dP, ldp, y, ldy, C, dt))) return rc;
    if ((rc = check_pair("cmu_maxpool_bwd_apply(dY,y)", dY, ldo, y, ldy, C, dt))) return rc;
    if (dSkip && (rc = check_pair("cmu_maxpool_bwd_apply(dSkip,y)", dSkip, lds, y, ldy, C, dt))) return rc;
    if (dSkip2 && (rc = check_pair("cmu_maxpool_bwd_apply(dSkip2,y)", dSkip2, lds2, y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "cmu_maxpool_bwd_apply: bad dims (%d,%d)", H, W);
    CMU_DISPATCH_DT(dt, maxpool_bwd_t, dP, ldp, dSkip, lds, dSkip2, lds2, y, ldy, scale, shift, dY, ldo, B, H, W, C, save_mean, save_invstd,
                    nullptr, (hipStream_t)stream, coef);
}
extern "C" int cmu_maxpool_bwd(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* y, int64_t ldy,
                               const float* scale, const float* shift, void* dA, int64_t lda, const float* save_mean,
                               const float* save_invstd, void* bn_ws, int B, int H, int W, int C, int dt, void* stream) {
    return cmu_maxpool_bwd2(dP, ldp, dSkip, lds, nullptr, 0, y, ldy, scale, shift, dA, lda, save_mean, save_invstd, bn_ws, B, H, W, C, dt,
                            stream);
}

// ---------------------------------------------------------------------------------------------
// 1x1 head backward (autograd of model.py:130)
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_MAX_K = 8;
template <class TR, int KT>   // KT = compiled class count (2 for the reference's heads, KT otherwise)
__global__ __launch_bounds__(256) void conv1x1_head_bwd_kernel(const float* __restrict__ dlogits, const unsigned char* __restrict__ x,
                                                              int64_t ldx, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ w,
                                                              unsigned char* __restrict__ dX, int64_t ldo, float* __restrict__ ws,
                                                              int B, int H, int W, int C, int K, int64_t npix,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ bn_ws) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    __shared__ float red[FOLD_G * 256];
    const int tid = threadIdx.x;
    const int nchunk = C / EPC;  // power of two <= 64
    const int ch = tid % nchunk;
    const int ppb = 256 / nchunk;
    const int prow = tid / nchunk;
    float sc[EPC], sh[EPC], wk[KT][EPC], dw[KT][EPC], db[KT], mu[EPC], is[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = scale ? scale[ch * EPC + e] : 1.f;
        sh[e] = scale ? shift[ch * EPC + e] : 0.f;
        mu[e] = bn_ws ? mean[ch * EPC + e] : 0.f;
        is[e] = bn_ws ? invstd[ch * EPC + e] : 0.f;
        s1[e] = s2[e] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        db[k] = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            wk[k][e] = (k < K) ? w[k * C + ch * EPC + e] : 0.f;
            dw[k][e] = 0.f;
        }
    }
    const int64_t HW = (int64_t)H * W;
    // Round 4: four pixels per thread and trip, their 16-byte loads and dlogits reads issued together (one load in flight per
    // thread left this pass latency-bound at 2.6 TB/s), and the workgroup's sums folded inside the waves (fold_rows).
    // (image, pixel-in-image) of the trip's first pixel is carried along the grid-stride loop instead of a 64-bit division per
    // pixel and thread; the other three follow by increments.
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * ppb * U;
    const int64_t sb = stride / HW, sr = stride % HW;
    int64_t pix0 = (int64_t)blockIdx.x * ppb * U + prow;
    int64_t b0 = pix0 / HW, r0 = pix0 % HW;
    for (; pix0 < npix; pix0 += stride, b0 += sb, r0 += sr) {
        if (r0 >= HW) { r0 -= HW; ++b0; }
        u32x4 q[U];
        float dlv[U][KT];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pix = pix0 + (int64_t)u * ppb;
            ok[u] = pix < npix;
            int64_t b = b0, r = r0 + (int64_t)u * ppb;
            while (r >= HW) { r -= HW; ++b; }
            q[u] = ok[u] ? ld_global16(x + (pix * ldx + ch * EPC) * ES) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < KT; ++k) dlv[u][k] = (ok[u] && k < K) ? dlogits[(b * K + k) * HW + r] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int64_t pix = pix0 + (int64_t)u * ppb;
            float f[EPC], o[EPC];
            TR::unpack(q[u], f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float a = fmaf(f[e], sc[e], sh[e]);
                if (scale) a = fmaxf(a, 0.f);
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    d = fmaf(dlv[u][k], wk[k][e], d);
                    dw[k][e] = fmaf(dlv[u][k], a, dw[k][e]);
                }
                o[e] = d;
            }
            if (ch == 0) {
#pragma unroll
                for (int k = 0; k < KT; ++k) db[k] += dlv[u][k];
            }
            const u32x4 packed = TR::pack(o);
            if (dX) st_global16(dX + (pix * ldo + ch * EPC) * ES, packed);
            if (bn_ws) {   // BN+ReLU backward statistics of the producing layer, on dX as stored
                float dr[EPC];
                TR::unpack(packed, dr);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float dz = fmaf(f[e], sc[e], sh[e]) > 0.f ? dr[e] : 0.f;
                    s1[e] += dz;
                    s2[e] = fmaf(dz, (f[e] - mu[e]) * is[e], s2[e]);
                }
            }
        }
    }
    // nchunk is a power of two <= 64 (host check): every sum of the workgroup goes through fold_rows
    if (bn_ws) {
        if (blockIdx.x == 0 && tid == 0) *reinterpret_cast<int*>(bn_ws) = (int)gridDim.x;
        float* bws = bn_ws + BNWS_HDR / 4;
        float t[2 * EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { t[e] = s1[e]; t[EPC + e] = s2[e]; }
        fold_rows<2 * EPC>(t, red, tid, nchunk, ppb);
        if (tid < nchunk) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                bws[((int64_t)blockIdx.x * 2 + 0) * C + tid * EPC + e] = t[e];
                bws[((int64_t)blockIdx.x * 2 + 1) * C + tid * EPC + e] = t[EPC + e];
            }
        }
    }
    float* out = ws + (int64_t)blockIdx.x * (K * C + K);
    {
        float t[KT * EPC + KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) t[k * EPC + e] = dw[k][e];
            t[KT * EPC + k] = db[k];     // bias: only ch == 0 threads carry data (tid % nchunk == 0)
        }
        fold_rows<KT * EPC + KT>(t, red, tid, nchunk, ppb);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            if (k >= K) break;   // K is uniform; indices stay compile-time constants after unrolling
            if (tid < nchunk) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) out[k * C + tid * EPC + e] = t[k * EPC + e];
            }
            if (tid == 0) out[K * C + k] = t[KT * EPC + k];
        }
    }
}
__global__ __launch_bounds__(256) void sum_slab_kernel(const float* __restrict__ ws, int nblocks, int64_t n, float* __restrict__ out0,
                                                      int64_t n0, float* __restrict__ out1) {
    // out0 gets elements [0,n0), out1 the rest; fixed-order sum over the blocks' partials (16 outputs x 16 row-parts)
    __shared__ double red[16][16];
    const int il = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + il;
    double s = 0.0;
    if (i < n)
        for (int b = part; b < nblocks; b += 16) s += (double)ws[(int64_t)b * n + i];
    red[part][il] = s;
    __syncthreads();
    if (part != 0 || i >= n) return;
    s = 0.0;
    for (int q = 0; q < 16; ++q) s += red[q][il];
    if (i < n0) out0[i] = (float)s;
    else if (out1) out1[i - n0] = (float)s;
}
template <class TR>
static int conv1x1_head_bwd_t(const float* dlogits, const void* x, int64_t ldx, const float* scale, const float* shift, const float* w,
                              void* dX, int64_t ldo, float* dW, float* dbias, int B, int H, int W, int C, int K, void* ws,
                              const float* mean, const float* invstd, void* bn_ws, hipStream_t st) {
    const int nchunk = C / TR::EPC;
    const int ppb = 256 / nchunk;
    const int64_t npix = (int64_t)B * H * W;
    int gx = (int)(cmu_div_up64(npix, ppb * 4) < HEADB_MAX_BLOCKS ? cmu_div_up64(npix, ppb * 4) : HEADB_MAX_BLOCKS);
    if (gx < 1) gx = 1;
    if (K <= 2)
        hipLaunchKernelGGL((conv1x1_head_bwd_kernel<TR, 2>), dim3(gx), dim3(256), 0, st, dlogits, (const unsigned char*)x, ldx, scale, shift, w,
                           (unsigned char*)dX, ldo, (float*)ws, B, H, W, C, K, npix, mean, invstd, (float*)bn_ws);
    else
        hipLaunchKernelGGL((conv1x1_head_bwd_kernel<TR, HEAD_MAX_K>), dim3(gx), dim3(256), 0, st, dlogits, (const unsigned char*)x, ldx, scale,
                           shift, w, (unsigned char*)dX, ldo, (float*)ws, B, H, W, C, K, npix, mean, invstd, (float*)bn_ws);
    CMU_CHECK_LAUNCH("cmu_conv1x1_head_bwd");
    const int64_t n = (int64_t)K * C + K;
    hipLaunchKernelGGL(sum_slab_kernel, dim3((unsigned)cmu_div_up64(n, 16)), dim3(256), 0, st, (const float*)ws, gx, n, dW, (int64_t)K * C,
                       dbias);
    CMU_CHECK_LAUNCH("cmu_conv1x1_head_bwd(sum)");
    return CMU_OK;
}
extern "C" int64_t cmu_conv1x1_head_bwd_ws_bytes(int B, int H, int W, int C, int K) {
    return (int64_t)RED_MAX_BLOCKS * ((int64_t)K * C + K) * (int64_t)sizeof(float);
}
extern "C" int cmu_conv1x1_head_bwd(const float* dlogits, const void* x, int64_t ldx, const float* in_scale, const float* in_shift,
                                    const float* w, void* dX, int64_t ldo, float* dW, float* dbias, const float* save_mean,
                                    const float* save_invstd, void* bn_ws, int B, int H, int W, int C, int K, int dt, void* ws,
                                    void* stream) {
    CMU_CHECK_ARG(bn_ws == nullptr || (save_mean && save_invstd && in_scale), "cmu_conv1x1_head_bwd: fused BN statistics need the transform and save_mean / save_invstd");
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && dlogits && x && w && dW && dbias && ws && B > 0 && H > 0 && W > 0, "cmu_conv1x1_head_bwd: bad args");
    const int epc = 16 / es;
    const int nchunk = C / epc;
    CMU_CHECK_ARG(K >= 1 && K <= HEAD_MAX_K, "cmu_conv1x1_head_bwd: K=%d must be in 1..%d", K, HEAD_MAX_K);
    CMU_CHECK_ARG(C % epc == 0 && nchunk > 0 && (nchunk & (nchunk - 1)) == 0 && nchunk <= 64, "cmu_conv1x1_head_bwd: C=%d unsupported", C);
    CMU_CHECK_ARG(cmu_aligned16(x) && ldx % epc == 0 && ldx >= C, "cmu_conv1x1_head_bwd: x alignment / stride");
    CMU_CHECK_ARG(!dX || (cmu_aligned16(dX) && ldo % epc == 0 && ldo >= C), "cmu_conv1x1_head_bwd: dX alignment / stride");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_conv1x1_head_bwd: scale/shift must both be set");
    CMU_DISPATCH_DT(dt, conv1x1_head_bwd_t, dlogits, x, ldx, in_scale, in_shift, w, dX, ldo, dW, dbias, B, H, W, C, K, ws, save_mean,
                    save_invstd, bn_ws, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// 1x1 head backward + BatchNorm+ReLU backward of the layer that fed the head, second half: dY written straight from dlogits.
// The head's input gradient has rank K (dA[p][c] = sum_k dlogits[p][k] * w[k][c]): it never needs to exist in memory.
// cmu_conv1x1_head_bwd with dX = NULL leaves the BN-backward partial sums (taken on dA rounded to the storage type, as the
// two-pass form stores it) and the head's parameter gradients; after the finalisation this pass recomputes dA per pixel chunk
// and applies  dY = scale * (gate * dA - c1 - xhat * c2)  -- x is read twice and dY written once (3.2 GB at 32 x 512 x 512 x 64
// f16) where head backward + cmu_bn_bwd_apply read x twice, wrote and re-read dA and wrote dY (5.4 GB).  Same bits as the two passes.
// ---------------------------------------------------------------------------------------------
template <class TR, int KT>
__global__ __launch_bounds__(256) void conv1x1_head_bn_apply_kernel(const float* __restrict__ dlogits, const unsigned char* __restrict__ x,
                                                                   int64_t ldx, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ w,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ coef, unsigned char* __restrict__ dY,
                                                                   int64_t ldo, int H, int W, int C, int K, int64_t npix) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    const int tid = threadIdx.x;
    const int nchunk = C / EPC;  // power of two <= 64
    const int ch = tid % nchunk;
    const int ppb = 256 / nchunk;
    const int prow = tid / nchunk;
    float sc[EPC], sh[EPC], wk[KT][EPC], mu[EPC], is[EPC], c1[EPC], c2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = ch * EPC + e;
        sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
        c1[e] = coef[c]; c2[e] = coef[C + c];
#pragma unroll
        for (int k = 0; k < KT; ++k) wk[k][e] = (k < K) ? w[k * C + c] : 0.f;
    }
    const int64_t HW = (int64_t)H * W;
    // (image, pixel-in-image) of the thread's pixel, carried along the grid-stride loop: a 64-bit division per pixel and thread
    // was what bound this pass (1.07 ms for 2.2 GB)
    const int64_t stride = (int64_t)gridDim.x * ppb;
    const int64_t sb = stride / HW, sr = stride % HW;
    int64_t pix = (int64_t)blockIdx.x * ppb + prow;
    int64_t b = pix / HW, r = pix % HW;
    for (; pix < npix; pix += stride, b += sb, r += sr) {
        if (r >= HW) { r -= HW; ++b; }
        float f[EPC], o[EPC], dl[KT];
        TR::unpack(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + (pix * ldx + ch * EPC) * ES)), f);
#pragma unroll
        for (int k = 0; k < KT; ++k) dl[k] = (k < K) ? dlogits[(b * K + k) * HW + r] : 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) d = fmaf(dl[k], wk[k][e], d);
            o[e] = d;
        }
        float g[EPC];
        TR::unpack(TR::pack(o), g);                    // dA as the two-pass form stores it
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float dz = fmaf(f[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
            const float xh = (f[e] - mu[e]) * is[e];
            o[e] = sc[e] * (dz - c1[e] - xh * c2[e]);
        }
        __builtin_nontemporal_store(TR::pack(o), reinterpret_cast<u32x4*>(dY + (pix * ldo + ch * EPC) * ES));
    }
}
template <class TR>
static int conv1x1_head_bn_apply_t(const float* dlogits, const void* x, int64_t ldx, const float* scale, const float* shift, const float* w,
                                   const float* mean, const float* invstd, const float* coef, void* dY, int64_t ldo, int B, int H, int W,
                                   int C, int K, hipStream_t st) {
    const int nchunk = C / TR::EPC;
    const int ppb = 256 / nchunk;
    const int64_t npix = (int64_t)B * H * W;
    int gx = (int)(cmu_div_up64(npix, ppb * 4) < (1 << 20) ? cmu_div_up64(npix, ppb * 4) : (1 << 20));   // four pixel chunks per thread (cmu_bn_bwd_apply)
    if (gx < 1) gx = 1;
    if (K <= 2)
        hipLaunchKernelGGL((conv1x1_head_bn_apply_kernel<TR, 2>), dim3(gx), dim3(256), 0, st, dlogits, (const unsigned char*)x, ldx, scale, shift, w,
                           mean, invstd, coef, (unsigned char*)dY, ldo, H, W, C, K, npix);
    else
        hipLaunchKernelGGL((conv1x1_head_bn_apply_kernel<TR, HEAD_MAX_K>), dim3(gx), dim3(256), 0, st, dlogits, (const unsigned char*)x, ldx, scale,
                           shift, w, mean, invstd, coef, (unsigned char*)dY, ldo, H, W, C, K, npix);
    CMU_CHECK_LAUNCH("cmu_conv1x1_head_bn_apply");
    return CMU_OK;
}
extern "C" int cmu_conv1x1_head_bn_apply(const float* dlogits, const void* x, int64_t ldx, const float* in_scale, const float* in_shift,
                                         const float* w, const float* save_mean, const float* save_invstd, const float* coef, void* dY,
                                         int64_t ldo, int B, int H, int W, int C, int K, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && dlogits && x && w && in_scale && in_shift && save_mean && save_invstd && coef && dY && B > 0 && H > 0 && W > 0,
                  "cmu_conv1x1_head_bn_apply: bad args");
    const int epc = 16 / es;
    const int nchunk = C / epc;
    CMU_CHECK_ARG(K >= 1 && K <= HEAD_MAX_K, "cmu_conv1x1_head_bn_apply: K=%d must be in 1..%d", K, HEAD_MAX_K);
    CMU_CHECK_ARG(C % epc == 0 && nchunk > 0 && (nchunk & (nchunk - 1)) == 0 && nchunk <= 64, "cmu_conv1x1_head_bn_apply: C=%d unsupported", C);
    CMU_CHECK_ARG(cmu_aligned16(x) && ldx % epc == 0 && ldx >= C && cmu_aligned16(dY) && ldo % epc == 0 && ldo >= C,
                  "cmu_conv1x1_head_bn_apply: alignment / stride");
    if (K <= 2 && (int64_t)B * H * W < (1ll << 31)) {
        // one or two classes (every head of the reference: two for the UNet / CM-UNet decoders, one for SparK's): the BatchNorm-backward
        // apply kernel itself, with dA recomputed from dlogits in its load slot (same grid, same loop: it runs at the HBM rate; the
        // generic kernel below measured 1.07 ms against 0.56)
        CMU_DISPATCH_DT(dt, bn_bwd_apply_t, nullptr, ldx, x, ldx, in_scale, in_shift, save_mean, save_invstd, coef, dY, ldo, B, H, W, C,
                        (const uint8_t*)nullptr, 0, (hipStream_t)stream, dlogits, w, K);
    }
    CMU_DISPATCH_DT(dt, conv1x1_head_bn_apply_t, dlogits, x, ldx, in_scale, in_shift, w, save_mean, save_invstd, coef, dY, ldo, B, H, W, C, K,
                    (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// first layer weight gradient: dW (Cout,1,3,3) = sum_p dY[p][n] * xm[p + tap]
// ---------------------------------------------------------------------------------------------
#ifndef CMU_C1W_BLOCKS
#define CMU_C1W_BLOCKS 512
#endif
#ifndef CMU_C1W_RECOMP
#define CMU_C1W_RECOMP 1     // 1: the fused-BN form recomputes the layer's raw output from the image instead of reading it (below)
#endif
constexpr int C1W_MAX_BLOCKS = CMU_C1W_BLOCKS;
// Round 4.  (i) The next tile's halo is loaded into registers before the current tile's pixels are walked and stored to the other
// LDS buffer behind them: one barrier per tile and no exposed round trip (was: barrier, load, barrier per 16 x 16 tile).
// (ii) The workgroup's 72 sums per thread are folded inside the waves (fold_rows): 9 barrier pairs instead of 144, which is what
// had capped the grid at 512 workgroups.  (iii) RECOMP (the fused-BN form, DESIGN section 8 item 0 on its cheapest consumer): the
// raw output y1 of this layer is 9 FMAs per element away from the halo the pass holds in LDS anyway, so it is recomputed -- in the
// forward kernel's order (conv3x3_c1_fwd_kernel: a = fmaf(in[t], w[t], a), t = 0..8, from 0) and rounded through the storage type, i.e.
// the very bits the forward stored -- instead of read: 1.07 GB of the pass's 2.28 GB at 32 x 512 x 512 never leave HBM.
template <class TR, bool RECOMP>
__global__ __launch_bounds__(256, 2) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                              int mask_per_sample, const unsigned char* __restrict__ dY, int64_t ldd,
                                                              float* __restrict__ ws, int B, int H, int W, int Cout, int tilesX,
                                                              int tilesY, int ntiles, const unsigned char* __restrict__ yraw, int64_t ldy,
                                                              const float* __restrict__ bsc, const float* __restrict__ bsh,
                                                              const float* __restrict__ bmu, const float* __restrict__ bis,
                                                              const float* __restrict__ coef, const float* __restrict__ wfwd,
                                                              const int* __restrict__ tlist = nullptr, const int* __restrict__ tcount = nullptr) {
    // tlist (SparK's sparse encoder, cmu_conv3x3_c1_wgrad_bn_tiles): the contraction runs over the listed 16 x 16 tiles only (the
    // gradient is zero elsewhere), round-robin over the workgroups
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    __shared__ float halo[2][18 * 18];
    __shared__ float red[FOLD_G * 256];
    const int tid = threadIdx.x;
    const int nchunk = Cout / EPC;
    const int ppi = 256 / nchunk;
    const int chunk = tid % nchunk, prow = tid / nchunk;
    const bool active = prow < ppi;
    const bool bn = RECOMP || yraw != nullptr;
    float acc[EPC][9];
#pragma unroll
    for (int e = 0; e < EPC; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[e][t] = 0.f;
    // bn: ``dY`` holds dA (gradient w.r.t. the activated output) and the BatchNorm+ReLU backward of this layer is
    // applied on the fly (cmu_conv3x3_c1_wgrad_bn) -- the first layer has no data gradient, so dY is never materialised
    float sc[EPC], sh[EPC], mu[EPC], is[EPC], c1[EPC], c2[EPC];
    if (bn) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = active ? chunk * EPC + e : 0;
            sc[e] = bsc[c]; sh[e] = bsh[c]; mu[e] = bmu[c]; is[e] = bis[c]; c1[e] = coef[c]; c2[e] = coef[Cout + c];
        }
    }
    float wr[RECOMP ? EPC : 1][9];
    if (RECOMP) {
#pragma unroll
        for (int e = 0; e < EPC; ++e)
#pragma unroll
            for (int t = 0; t < 9; ++t) wr[e][t] = active ? wfwd[(chunk * EPC + e) * 9 + t] : 0.f;
    }
    const int tpi = tilesX * tilesY;
    float hv[2];
    auto halo_load = [&](int tile) {
        const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY, b = tile / tpi;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * 256;
            const int gy = ty * 16 - 1 + i / 18, gx = tx * 16 - 1 + i % 18;
            float v = 0.f;
            if (i < 18 * 18 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                v = x[((int64_t)b * H + gy) * W + gx];
                if (mask) v *= (float)(1 - (int)mask[((int64_t)(mask_per_sample ? b : 0) * H + gy) * W + gx]);
            }
            hv[k] = v;
        }
    };
    auto halo_store = [&](float* buf) {
        buf[tid] = hv[0];
        if (tid + 256 < 18 * 18) buf[tid + 256] = hv[1];
    };
    const int nwork = tlist != nullptr ? tcount[0] : ntiles;
    int wi = blockIdx.x;
    if (wi < nwork) { halo_load(tlist != nullptr ? tlist[wi] : wi); halo_store(halo[0]); }
    __syncthreads();
    for (int it = 0; wi < nwork; wi += gridDim.x, ++it) {
        const int tile = tlist != nullptr ? tlist[wi] : wi;
        const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY, b = tile / tpi;
        const int ty0 = ty * 16, tx0 = tx * 16;
        const float* hb = halo[it & 1];
        const bool more = wi + (int)gridDim.x < nwork;
        if (more) halo_load(tlist != nullptr ? tlist[wi + gridDim.x] : wi + (int)gridDim.x);
        // four pixels per trip: their 16-byte loads (dY, and the raw output when it is read) go out together --
        // one or two loads in flight per wave left this pass latency-bound at ~2.6 TB/s
        constexpr int U = 4;
        if (active)
            for (int pix0 = prow; pix0 < 256; pix0 += ppi * U) {
                u32x4 gq[U], vq[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pix = pix0 + u * ppi;
                    const int gy = ty0 + (pix >> 4), gx = tx0 + (pix & 15);
                    ok[u] = pix < 256 && gy < H && gx < W;
                    gq[u] = vq[u] = u32x4{0u, 0u, 0u, 0u};
                    if (ok[u]) {
                        gq[u] = ld_global16_nt(dY + ((((int64_t)b * H + gy) * W + gx) * ldd + chunk * EPC) * ES);
                        if (!RECOMP && yraw != nullptr) vq[u] = ld_global16_nt(yraw + ((((int64_t)b * H + gy) * W + gx) * ldy + chunk * EPC) * ES);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!ok[u]) continue;
                    const int pix = pix0 + u * ppi;
                    const int py = pix >> 4, px = pix & 15;
                    float g[EPC], in[9];
                    TR::unpack(gq[u], g);
#pragma unroll
                    for (int t = 0; t < 9; ++t) in[t] = hb[(py + t / 3) * 18 + px + t % 3];
                    if (bn) {
                        float v[EPC];
                        if (RECOMP) {
                            float o[EPC];
#pragma unroll
                            for (int e = 0; e < EPC; ++e) {
                                float a = 0.f;
#pragma unroll
                                for (int t = 0; t < 9; ++t) a = fmaf(in[t], wr[e][t], a);
                                o[e] = a;
                            }
                            TR::unpack(TR::pack(o), v);
                        } else {
                            TR::unpack(vq[u], v);
                        }
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            const float dz = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
                            g[e] = sc[e] * (dz - c1[e] - (v[e] - mu[e]) * is[e] * c2[e]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[e][t] = fmaf(g[e], in[t], acc[e][t]);
                    }
                }
            }
        if (more) halo_store(halo[(it + 1) & 1]);
        __syncthreads();
    }
    float* out = ws + (int64_t)blockIdx.x * Cout * 9;
    if (fold_ok(nchunk) && nchunk <= 256) {
        float t9[EPC * 9];
#pragma unroll
        for (int e = 0; e < EPC; ++e)
#pragma unroll
            for (int t = 0; t < 9; ++t) t9[e * 9 + t] = active ? acc[e][t] : 0.f;
        fold_rows<EPC * 9>(t9, red, tid, nchunk, ppi);
        if (tid < nchunk) {
#pragma unroll
            for (int q = 0; q < EPC * 9; ++q) out[tid * EPC * 9 + q] = t9[q];
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float a = sum_over_rows(active ? acc[e][t] : 0.f, red, tid, nchunk, ppi);
            if (tid < nchunk) out[(tid * EPC + e) * 9 + t] = a;
        }
}
template <class TR>
static int conv3x3_c1_wgrad_t(const float* x, const uint8_t* mask, int mps, const void* dY, int64_t ldd, float* dW, int B, int H, int W,
                              int Cout, void* ws, hipStream_t st, const void* yraw = nullptr, int64_t ldy = 0, const float* bsc = nullptr,
                              const float* bsh = nullptr, const float* bmu = nullptr, const float* bis = nullptr,
                              const float* coef = nullptr, const float* wfwd = nullptr, const int* tlist = nullptr,
                              const int* tcount = nullptr, int64_t max_tiles = 0) {
    const int tilesX = cmu_div_up(W, 16), tilesY = cmu_div_up(H, 16);
    const int ntiles = B * tilesX * tilesY;
    const int64_t nwork = tlist != nullptr ? (max_tiles < 1 ? 1 : max_tiles) : ntiles;
    const int grid = (int)(nwork < C1W_MAX_BLOCKS ? nwork : C1W_MAX_BLOCKS);
    if (wfwd != nullptr)
        hipLaunchKernelGGL((conv3x3_c1_wgrad_kernel<TR, true>), dim3(grid), dim3(256), 0, st, x, mask, mps, (const unsigned char*)dY, ldd, (float*)ws,
                           B, H, W, Cout, tilesX, tilesY, ntiles, (const unsigned char*)yraw, ldy, bsc, bsh, bmu, bis, coef, wfwd, tlist, tcount);
    else
        hipLaunchKernelGGL((conv3x3_c1_wgrad_kernel<TR, false>), dim3(grid), dim3(256), 0, st, x, mask, mps, (const unsigned char*)dY, ldd, (float*)ws,
                           B, H, W, Cout, tilesX, tilesY, ntiles, (const unsigned char*)yraw, ldy, bsc, bsh, bmu, bis, coef, wfwd, tlist, tcount);
    CMU_CHECK_LAUNCH("cmu_conv3x3_c1_wgrad");
    const int64_t n = (int64_t)Cout * 9;
    hipLaunchKernelGGL(sum_slab_kernel, dim3((unsigned)cmu_div_up64(n, 16)), dim3(256), 0, st, (const float*)ws, grid, n, dW, n, (float*)nullptr);
    CMU_CHECK_LAUNCH("cmu_conv3x3_c1_wgrad(sum)");
    return CMU_OK;
}
extern "C" int64_t cmu_conv3x3_c1_wgrad_ws_bytes(int B, int H, int W, int Cout) {
    return (int64_t)C1W_MAX_BLOCKS * Cout * 9 * (int64_t)sizeof(float);
}
extern "C" int cmu_conv3x3_c1_wgrad(const float* x, const uint8_t* mask, int mask_per_sample, const void* dY, int64_t ldd, float* dW,
                                    int B, int H, int W, int Cout, int dt, void* ws, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && dY && dW && ws && B > 0 && H > 0 && W > 0, "cmu_conv3x3_c1_wgrad: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(Cout > 0 && Cout % epc == 0 && Cout / epc <= 256, "cmu_conv3x3_c1_wgrad: Cout=%d unsupported", Cout);
    CMU_CHECK_ARG(cmu_aligned16(dY) && ldd % epc == 0 && ldd >= Cout, "cmu_conv3x3_c1_wgrad: dY alignment / stride");
    CMU_DISPATCH_DT(dt, conv3x3_c1_wgrad_t, x, mask, mask_per_sample, dY, ldd, dW, B, H, W, Cout, ws, (hipStream_t)stream);
}
extern "C" int cmu_conv3x3_c1_wgrad_bn(const float* x, const uint8_t* mask, int mask_per_sample, const void* dA, int64_t ldd,
                                       const void* yraw, int64_t ldy, const float* scale, const float* shift, const float* save_mean,
                                       const float* save_invstd, const float* coef, float* dW, int B, int H, int W, int Cout, int dt,
                                       void* ws, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && dA && yraw && scale && shift && save_mean && save_invstd && coef && dW && ws && B > 0 && H > 0 && W > 0,
                  "cmu_conv3x3_c1_wgrad_bn: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(Cout > 0 && Cout % epc == 0 && Cout / epc <= 256, "cmu_conv3x3_c1_wgrad_bn: Cout=%d unsupported", Cout);
    CMU_CHECK_ARG(cmu_aligned16(dA) && ldd % epc == 0 && ldd >= Cout && cmu_aligned16(yraw) && ldy % epc == 0 && ldy >= Cout,
                  "cmu_conv3x3_c1_wgrad_bn: alignment / stride");
    CMU_DISPATCH_DT(dt, conv3x3_c1_wgrad_t, x, mask, mask_per_sample, dA, ldd, dW, B, H, W, Cout, ws, (hipStream_t)stream, yraw, ldy, scale,
                    shift, save_mean, save_invstd, coef);
}
// cmu_conv3x3_c1_wgrad_bn / cmu_conv3x3_c1_wgrad_bn_w over a list of 16 x 16 tiles (cmu_sparse_tile_list numbering): SparK's sparse first
// layer (Spark/encoder.py:20-36) -- dA, and the BatchNorm backward applied on the fly, only exist inside active patches, so neither the
// masked apply pass nor its dY tensor is needed.  Exactly one of yraw (read) / w (recomputed from the image) is given.
extern "C" int cmu_conv3x3_c1_wgrad_bn_tiles(const float* x, const uint8_t* mask, int mask_per_sample, const void* dA, int64_t ldd,
                                             const void* yraw, int64_t ldy, const float* w, const float* scale, const float* shift,
                                             const float* save_mean, const float* save_invstd, const float* coef, const int* tile_list,
                                             const int* tile_count, int64_t max_tiles, float* dW, int B, int H, int W, int Cout, int dt,
                                             void* ws, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && dA && scale && shift && save_mean && save_invstd && coef && dW && ws && tile_list && tile_count && B > 0 && H > 0 &&
                      W > 0 && ((yraw != nullptr) != (w != nullptr)),
                  "cmu_conv3x3_c1_wgrad_bn_tiles: bad args (exactly one of yraw / w)");
    const int epc = 16 / es;
    CMU_CHECK_ARG(Cout > 0 && Cout % epc == 0 && Cout / epc <= 256, "cmu_conv3x3_c1_wgrad_bn_tiles: Cout=%d unsupported", Cout);
    CMU_CHECK_ARG(cmu_aligned16(dA) && ldd % epc == 0 && ldd >= Cout && (yraw == nullptr || (cmu_aligned16(yraw) && ldy % epc == 0 && ldy >= Cout)),
                  "cmu_conv3x3_c1_wgrad_bn_tiles: alignment / stride");
    CMU_DISPATCH_DT(dt, conv3x3_c1_wgrad_t, x, mask, mask_per_sample, dA, ldd, dW, B, H, W, Cout, ws, (hipStream_t)stream, yraw, ldy, scale, shift,
                    save_mean, save_invstd, coef, w, tile_list, tile_count, max_tiles);
}
// The same with the raw output RECOMPUTED from the image and the layer's forward weights ``w`` (Cout,1,3,3) instead of read
// (conv3x3_c1_wgrad_kernel<., true>): bit-identical to cmu_conv3x3_c1_wgrad_bn on the tensor cmu_conv3x3_c1_fwd stored, at half
// the HBM traffic.  ``w`` must be the weights (and x / mask the inputs) that forward pass used.
extern "C" int cmu_conv3x3_c1_wgrad_bn_w(const float* x, const uint8_t* mask, int mask_per_sample, const void* dA, int64_t ldd,
                                         const float* w, const float* scale, const float* shift, const float* save_mean,
                                         const float* save_invstd, const float* coef, float* dW, int B, int H, int W, int Cout, int dt,
                                         void* ws, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && dA && w && scale && shift && save_mean && save_invstd && coef && dW && ws && B > 0 && H > 0 && W > 0,
                  "cmu_conv3x3_c1_wgrad_bn_w: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(Cout > 0 && Cout % epc == 0 && Cout / epc <= 256, "cmu_conv3x3_c1_wgrad_bn_w: Cout=%d unsupported", Cout);
    CMU_CHECK_ARG(cmu_aligned16(dA) && ldd % epc == 0 && ldd >= Cout, "cmu_conv3x3_c1_wgrad_bn_w: alignment / stride");
    CMU_DISPATCH_DT(dt, conv3x3_c1_wgrad_t, x, mask, mask_per_sample, dA, ldd, dW, B, H, W, Cout, ws, (hipStream_t)stream, nullptr, 0, scale,
                    shift, save_mean, save_invstd, coef, w);
}
