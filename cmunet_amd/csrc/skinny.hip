// skinny.hip -- weight-streaming GEMMs of the CM-UNet projector / predictor necks (SURVEY row a9, 8(b) `cmu_skinny_gemm_*`):
// NonLinearNeck's first Linear maps a whole feature image to 1,536 units (Pretraining/CM-UNet/cmae/models/necks/
// nonlinear_neck.py:63-66 with configs/cmunet_config.py:18-26: in_channels = H*W = 50,176 at 224^2, 262,144 at 512^2), i.e.
// 77-403 M fp32 weights for a batch of 32 rows per GPU.  All three products of a training step are bound by ONE pass over
// those 0.3-1.6 GB (reading W forward and for the input gradient, writing dW), the arithmetic is 64 FLOP per weight:
//   fwd    y  (M,N) = x (M,K) . w (N,K)^T (+ bias)     [nn.Linear]
//   dgrad  dx (M,K) = dy (M,N) . w (N,K)
//   wgrad  dw (N,K) = dy^T (N,M) . x (M,K),  dbias (N) = sum_m dy
// for M <= 32 rows in one 32-row MFMA tile; up to SK_MAX_M = 256 rows (the reference's own batch size per GPU, cmunet_config.py:55,
// moco2_module.py:91) as row groups of 32 (round 4: nothing on the neck path dispatches into a library any more) -- forward and
// input gradient take one pass over the weights per group, the weight gradient contracts over all groups inside one launch.  At
// M = 256, K = 50,176, N = 1,536 that is 8 x 0.3 GB per product, ~0.5 ms: 1 % of the joint step at that batch size.
// fp32 in, fp32 out on `v_mfma_f32_32x32x2_f32` (exact f32 products, f32 accumulation -- the reference computes these layers
// in fp32): per wave 16-byte loads straight from HBM into the MFMA operand registers (lane l: row or column l & 31, four
// consecutive k of half l >> 5, so MFMA j of a step contracts k = kb + j and kb + 4 + j), no LDS.  Each wave streams its own
// 32 weight rows (fwd) or 128 weight columns (dgrad, wgrad); the other operand (x, dy: <= 32 rows) comes from L2.
// Split-K partial sums (fwd) go through a slab and a fixed-order second kernel: no atomics, bitwise reproducible.
#include "common.h"
#include <stdlib.h>

typedef float f32x4s __attribute__((ext_vector_type(4)));
constexpr int SK_MAX_M = 256;      // rows per call (groups of 32)

__device__ static inline f32x4s sk_ld4(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const f32x4s*>(p) : f32x4s{0.f, 0.f, 0.f, 0.f};
}
__device__ static inline f32x16 sk_zero() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

// ---- forward: block = 4 waves = 4 x 32 rows of w over one K range; grid (ceil(N/128), splits) -------------------------------------
// The MFMA wants lane = weight row (four consecutive k per lane), but read that way a wave touches 32 rows a megabyte apart with
// 32 bytes each per load -- 2.2 TB/s of weights at K = 262,144 whatever the number of streams per workgroup (measured both
// ways).  So the operands go through LDS: per 64-k chunk the workgroup loads its 128 weight rows and the 32 rows of x as 128-byte
// runs (lane = 16-byte piece of a run, 8 rows per wave instruction), stores them row-major with a 144-byte pitch and the waves
// read their operands from there (conflict-free 16-byte reads: bank = 4 * row mod 64 inside a lane group); two buffers, one
// barrier per chunk, the next chunk's loads in flight across the MFMAs.
constexpr int SKF_KC = 32;                 // k per staged chunk (46 KB of LDS per workgroup: three per CU keep ~60 KB of loads in flight)
constexpr int SKF_PITCH = SKF_KC + 4;      // floats per LDS row
constexpr int SKF_ROWS = 128 + 32;         // weight rows + x rows per buffer
constexpr int SKF_LDS_BYTES = 2 * SKF_ROWS * SKF_PITCH * 4;   // 46,080
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ slab,
                                                        int M, int N, int64_t K, int64_t kchunk) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 128;
    const int n = n0 + wave * 32 + c;                         // w row of this lane (B operand column)
    const int64_t k0 = (int64_t)blockIdx.y * kchunk, k1 = k0 + kchunk < K ? k0 + kchunk : K;
    const bool nok = n < N;
    // staging role: piece = 16 bytes of a row's run; with P = SKF_KC / 4 pieces per row a pass of the 256 threads covers 256 / P rows
    constexpr int P = SKF_KC / 4, RPP = 256 / P, WIT = 128 / RPP, XIT = 32 / RPP;
    const int piece = tid % P, srow = tid / P;
    f32x4s wreg[WIT], xreg[XIT];
    auto load_chunk = [&](int64_t kb) {
        const int64_t k = kb + 4 * piece;
        const bool kok = k < k1;
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int row = n0 + srow + RPP * it;
            wreg[it] = sk_ld4(w + (int64_t)(row < N ? row : 0) * K + k, kok && row < N);
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int row = srow + RPP * it;
            xreg[it] = sk_ld4(x + (int64_t)(row < M ? row : 0) * K + k, kok && row < M);
        }
    };
    auto store_chunk = [&](int buf) {
        float* base = sk_lds + buf * (SKF_ROWS * SKF_PITCH) + 4 * piece;
#pragma unroll
        for (int it = 0; it < WIT; ++it) *reinterpret_cast<f32x4s*>(base + (srow + RPP * it) * SKF_PITCH) = wreg[it];
#pragma unroll
        for (int it = 0; it < XIT; ++it) *reinterpret_cast<f32x4s*>(base + (128 + srow + RPP * it) * SKF_PITCH) = xreg[it];
    };
    f32x16 acc = sk_zero();
    const int nch = (int)((k1 - k0 + SKF_KC - 1) / SKF_KC);
    if (nch > 0) {
        load_chunk(k0);
        store_chunk(0);
    }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) load_chunk(k0 + (int64_t)(ch + 1) * SKF_KC);
        const float* wl = sk_lds + (ch & 1) * (SKF_ROWS * SKF_PITCH) + (wave * 32 + c) * SKF_PITCH + 4 * h;
        const float* xl = sk_lds + (ch & 1) * (SKF_ROWS * SKF_PITCH) + (128 + c) * SKF_PITCH + 4 * h;
#pragma unroll
        for (int u = 0; u < SKF_KC / 8; ++u) {
            const f32x4s wv = *reinterpret_cast<const f32x4s*>(wl + 8 * u), xv = *reinterpret_cast<const f32x4s*>(xl + 8 * u);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j], wv[j], acc, 0, 0, 0);
        }
        if (ch + 1 < nch) store_chunk((ch + 1) & 1);
        __syncthreads();
    }
    // D[m][n]: lane holds column n (its w row), rows m = (e & 3) + 8 (e >> 2) + 4 h
    float* out = slab + (int64_t)blockIdx.y * M * N;
    if (nok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < M) out[(int64_t)m * N + n] = acc[e];
        }
    }
}
__global__ void skinny_fwd_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                         int splits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    float s = bias ? bias[i % N] : 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(int64_t)k * M * N + i];
    y[i] = s;
}
static int skf_splits(int N, int64_t K, int64_t* kchunk) {
    const int nblk = cmu_div_up(N, 128);
    // three workgroups fit a CU (LDS): one round of 768 measured best at both projector shapes (512: 0.49 / 0.083 ms, 768: 0.413 /
    // 0.081, 1,024: 0.436 / 0.095, 1,536: 0.434 / 0.098 at K = 262,144 / 50,176); CMU_SKF_WGS overrides (A/B)
    static const int target = []() { const char* e = getenv("CMU_SKF_WGS"); return e ? atoi(e) : 768; }();
    int splits = (int)cmu_div_up64(target, nblk);
    int64_t kc = cmu_div_up64(K, splits);
    kc = cmu_div_up64(kc, SKF_KC) * SKF_KC;
    if (kc < SKF_KC) kc = SKF_KC;
    *kchunk = kc;
    return (int)cmu_div_up64(K, kc);
}
extern "C" int64_t cmu_skinny_gemm_ws_bytes(int M, int N, int64_t K) {
    int64_t kc;
    return (int64_t)skf_splits(N, K, &kc) * (M < 32 ? M : 32) * N * (int64_t)sizeof(float);     // one slab, reused by every 32-row group
}
extern "C" int cmu_skinny_gemm_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int64_t K, void* ws, void* stream) {
    CMU_CHECK_ARG(x && w && y && ws && M >= 1 && M <= SK_MAX_M && N >= 1 && K >= 8 && K % 8 == 0, "cmu_skinny_gemm_fwd: needs 1 <= M <= %d, K %% 8 == 0 (M=%d, K=%lld)",
                  SK_MAX_M, M, (long long)K);
    CMU_CHECK_ARG(cmu_aligned16(x) && cmu_aligned16(w), "cmu_skinny_gemm_fwd: x / w must be 16-byte aligned");
    int64_t kc;
    const int splits = skf_splits(N, K, &kc);
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SKF_LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("cmu_skinny_gemm_fwd: hipFuncSetAttribute(%d B LDS): %s", SKF_LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    for (int m0 = 0; m0 < M; m0 += 32) {      // row groups of 32: the slab is reused (same stream: the group's reduce precedes the next kernel)
        const int Mg = M - m0 < 32 ? M - m0 : 32;
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(cmu_div_up(N, 128), splits), dim3(256), SKF_LDS_BYTES, (hipStream_t)stream, x + (int64_t)m0 * K, w,
                           (float*)ws, Mg, N, K, kc);
        CMU_CHECK_LAUNCH("cmu_skinny_gemm_fwd");
        hipLaunchKernelGGL(skinny_fwd_reduce_kernel, dim3((unsigned)cmu_div_up64((int64_t)Mg * N, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)ws, bias, y + (int64_t)m0 * N, Mg, N, splits);
        CMU_CHECK_LAUNCH("cmu_skinny_gemm_fwd(reduce)");
    }
    return CMU_OK;
}

// ---- input gradient: block = 128 columns of w (four 32-column MFMA tiles: lane c holds columns 4c .. 4c+3); its four waves take
// a quarter of the N rows each and combine through LDS in wave order (one 128-column tile per wave alone leaves K / 512
// workgroups: 98 at K = 50,176 for 256 CUs).
// dyt: dy transposed (N, M) so that the A operand (lane r: dy[m = r][n + h]) is a contiguous 128-byte read per n
constexpr int SKD_UNROLL = 8;    // n pairs in flight per wave
__global__ __launch_bounds__(256) void skinny_dgrad_kernel(const float* __restrict__ dyt, const float* __restrict__ w, float* __restrict__ dx,
                                                          int M, int N, int64_t K) {
    __shared__ float red[4][4][16][64];                                       // [wave][tile j][e][lane]: 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t kcol = (int64_t)blockIdx.x * 128 + 4 * c;                   // first of this lane's four columns
    const bool kok = kcol < K, mok = c < M;
    const int nq = ((N + 3) / 4 + 1) & ~1;                                    // rows per wave (even: n pairs)
    const int nbeg = wave * nq, nend = nbeg + nq < N ? nbeg + nq : N;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = sk_zero();
    for (int n0 = nbeg; n0 < nend; n0 += 2 * SKD_UNROLL) {
        f32x4s wv[SKD_UNROLL];
        float av[SKD_UNROLL];
#pragma unroll
        for (int u = 0; u < SKD_UNROLL; ++u) {
            const int n = n0 + 2 * u + h;
            wv[u] = sk_ld4(w + (int64_t)(n < nend ? n : 0) * K + kcol, kok && n < nend);
            av[u] = (mok && n < nend) ? dyt[(int64_t)n * M + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SKD_UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wv[u][j], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][j][e][lane] = acc[j][e];
    __syncthreads();
    // wave q sums accumulator elements e = 4q .. 4q+3 of all four tiles over the waves, in wave order
    if (kok) {
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 4 * wave + ee;
            const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
            f32x4s v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((red[0][j][e][lane] + red[1][j][e][lane]) + red[2][j][e][lane]) + red[3][j][e][lane];
            if (m < M) *reinterpret_cast<f32x4s*>(dx + (int64_t)m * K + kcol) = v;
        }
    }
}
__global__ void skinny_transpose_kernel(const float* __restrict__ a, float* __restrict__ at, int M, int N) {   // (M,N) -> (N,M)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)M * N) at[(i % N) * M + i / N] = a[i];
}
extern "C" int64_t cmu_skinny_gemm_bwd_ws_bytes(int M, int N) { return (int64_t)(M < 32 ? M : 32) * N * (int64_t)sizeof(float); }
extern "C" int cmu_skinny_gemm_dgrad(const float* dy, const float* w, float* dx, int M, int N, int64_t K, void* ws, void* stream) {
    CMU_CHECK_ARG(dy && w && dx && ws && M >= 1 && M <= SK_MAX_M && N >= 1 && K >= 4 && K % 4 == 0, "cmu_skinny_gemm_dgrad: needs 1 <= M <= %d, K %% 4 == 0 (M=%d, K=%lld)",
                  SK_MAX_M, M, (long long)K);
    CMU_CHECK_ARG(cmu_aligned16(w) && cmu_aligned16(dx), "cmu_skinny_gemm_dgrad: w / dx must be 16-byte aligned");
    for (int m0 = 0; m0 < M; m0 += 32) {
        const int Mg = M - m0 < 32 ? M - m0 : 32;
        hipLaunchKernelGGL(skinny_transpose_kernel, dim3((unsigned)cmu_div_up64((int64_t)Mg * N, 256)), dim3(256), 0, (hipStream_t)stream, dy + (int64_t)m0 * N,
                           (float*)ws, Mg, N);
        CMU_CHECK_LAUNCH("cmu_skinny_gemm_dgrad(transpose)");
        hipLaunchKernelGGL(skinny_dgrad_kernel, dim3((unsigned)cmu_div_up64(K, 128)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, w,
                           dx + (int64_t)m0 * K, Mg, N, K);
        CMU_CHECK_LAUNCH("cmu_skinny_gemm_dgrad");
    }
    return CMU_OK;
}

// ---- weight gradient: wave = 32 rows n x 128 columns k; contraction over the M <= 32 rows (16 MFMA k-pairs) -----------------------
// A operand: lane r holds dy[m = 2p + h][n0 + r] for the 16 pairs p (contiguous 128-byte reads of dy rows); B: x[m][4c .. 4c+3]
constexpr int SKW_NT = 2;        // 32-row n tiles per wave and x fragment (halves the passes over x)
// MULTI (round 4, M > 32): the contraction runs over row groups of 32 with both n tiles' accumulators live; the one-group form
// keeps one tile's accumulators at a time (the shape the kernel was tuned for: the looped form measured 0.54 -> 1.00 ms on it)
template <bool MULTI>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                                          int M, int N, int64_t K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t kcol = ((int64_t)blockIdx.x * 4 + wave) * 128 + 4 * c;
    const bool kok = kcol < K;
    const int nbase = blockIdx.y * 32 * SKW_NT;
    if (!MULTI) {
        f32x4s xv[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int m = 2 * p + h;
            xv[p] = sk_ld4(x + (int64_t)(m < M ? m : 0) * K + kcol, kok && m < M);
        }
#pragma unroll
        for (int t = 0; t < SKW_NT; ++t) {
            const int n = nbase + 32 * t + c;       // A operand row of this lane
            f32x16 acc1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[j] = sk_zero();
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int m = 2 * p + h;
                const float a = (n < N && m < M) ? dy[(int64_t)m * N + n] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv[p][j], acc1[j], 0, 0, 0);
            }
            if (kok) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int nr = nbase + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;      // output row (D row = A row)
                    if (nr < N) *reinterpret_cast<f32x4s*>(dw + (int64_t)nr * K + kcol) = f32x4s{acc1[0][e], acc1[1][e], acc1[2][e], acc1[3][e]};
                }
            }
        }
        return;
    }
    f32x16 acc[SKW_NT][4];
#pragma unroll
    for (int t = 0; t < SKW_NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = sk_zero();
    for (int m0 = 0; m0 < M; m0 += 32) {      // row groups of 32 (one trip for the <= 32 rows the kernel was written for)
        f32x4s xv[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int m = m0 + 2 * p + h;
            xv[p] = sk_ld4(x + (int64_t)(m < M ? m : 0) * K + kcol, kok && m < M);
        }
#pragma unroll
        for (int t = 0; t < SKW_NT; ++t) {
            const int n = nbase + 32 * t + c;       // A operand row of this lane
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int m = m0 + 2 * p + h;
                const float a = (n < N && m < M) ? dy[(int64_t)m * N + n] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv[p][j], acc[t][j], 0, 0, 0);
            }
        }
    }
    if (kok) {
#pragma unroll
        for (int t = 0; t < SKW_NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int nr = nbase + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;      // output row (D row = A row)
                if (nr < N) *reinterpret_cast<f32x4s*>(dw + (int64_t)nr * K + kcol) = f32x4s{acc[t][0][e], acc[t][1][e], acc[t][2][e], acc[t][3][e]};
            }
    }
}
__global__ void skinny_colsum_kernel(const float* __restrict__ dy, float* __restrict__ dbias, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[(int64_t)m * N + n];
    dbias[n] = s;
}
extern "C" int cmu_skinny_gemm_wgrad(const float* dy, const float* x, float* dw, float* dbias, int M, int N, int64_t K, void* stream) {
    CMU_CHECK_ARG(dy && x && dw && M >= 1 && M <= SK_MAX_M && N >= 1 && K >= 4 && K % 4 == 0, "cmu_skinny_gemm_wgrad: needs 1 <= M <= %d, K %% 4 == 0 (M=%d, K=%lld)",
                  SK_MAX_M, M, (long long)K);
    CMU_CHECK_ARG(cmu_aligned16(x) && cmu_aligned16(dw), "cmu_skinny_gemm_wgrad: x / dw must be 16-byte aligned");
    const dim3 grid((unsigned)cmu_div_up64(K, 512), cmu_div_up(N, 32 * SKW_NT));
    if (M <= 32) hipLaunchKernelGGL(skinny_wgrad_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dy, x, dw, M, N, K);
    else hipLaunchKernelGGL(skinny_wgrad_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dy, x, dw, M, N, K);
    CMU_CHECK_LAUNCH("cmu_skinny_gemm_wgrad");
    if (dbias != nullptr) {
        hipLaunchKernelGGL(skinny_colsum_kernel, dim3(cmu_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, dy, dbias, M, N);
        CMU_CHECK_LAUNCH("cmu_skinny_gemm_wgrad(bias)");
    }
    return CMU_OK;
}
