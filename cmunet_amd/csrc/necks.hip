// necks.hip -- the small dense ops around the CM-UNet necks (kernel K12), so that nothing on that path
// dispatches into a library:
//   * BatchNorm1d (+ ReLU) over the rows of the necks' hidden layer, forward and backward, with the column sums exposed for
//     the SyncBN exchange (Pretraining/CM-UNet/cmae/models/necks/nonlinear_neck.py:88-102 with norm_cfg SyncBN eps 1e-6,
//     configs/cmunet_config.py:18-38) -- M = 32 rows per GPU, N = 1,536 columns: one thread per column, rows in order;
//   * the per-call Conv2d(C, C/4, 1) that reduces the target latent (cmae/models/algorithms/cmunet.py:128-131): a 1x1
//     convolution from the raw NHWC latent (+ its pending BatchNorm+ReLU) to an NCHW fp32 tensor whose memory the reference
//     re-views as a (B,1,H,W) image -- MFMA from global memory, no LDS (17 GFLOP at 512^2, weights L2-resident);
//   * 16-bit-operand variants of the weight-streaming skinny GEMMs (skinny.hip) for the AMP configuration
//     (cmunet_config.py:76-78: nn.Linear under autocast multiplies fp16 operands into fp32): x, w, dy stay fp32 in HBM, are
//     rounded to f16 / bf16 in registers and go through v_mfma_f32_32x32x16 -- 1/16 of the matrix instructions of the exact
//     fp32 path (whose 157 TFLOP/s pipe co-limits it), so the products are bound by the one pass over the weights alone.
#include "common.h"

typedef float f32x4s __attribute__((ext_vector_type(4)));

// =====================================================================================================================
// BatchNorm1d (+ ReLU)
// =====================================================================================================================
// sums[0][n] = sum_m x[m][n], sums[1][n] = sum_m x[m][n]^2 (the SyncBN exchange adds these over the ranks)
__global__ void bn1d_colsums_kernel(const float* __restrict__ x, float* __restrict__ sums, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f, q = 0.f;
    for (int m = 0; m < M; ++m) {
        const float v = x[(int64_t)m * N + n];
        s += v;
        q = fmaf(v, v, q);
    }
    sums[n] = s;
    sums[N + n] = q;
}
// training: statistics from the rows (two-pass variance) or from exchanged sums; eval: running statistics
__global__ void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ sums, float count, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                float momentum, float eps, int training, int relu, float* __restrict__ y, float* __restrict__ save_mean,
                                float* __restrict__ save_invstd, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float mean, var;
    if (!training) {
        mean = running_mean[n];
        var = running_var[n];
    } else if (sums != nullptr) {
        mean = sums[n] / count;
        var = fmaxf(sums[N + n] / count - mean * mean, 0.f);
    } else {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += x[(int64_t)m * N + n];
        mean = s / (float)M;
        float q = 0.f;
        for (int m = 0; m < M; ++m) {
            const float d = x[(int64_t)m * N + n] - mean;
            q = fmaf(d, d, q);
        }
        var = q / (float)M;
        count = (float)M;
    }
    const float invstd = 1.f / sqrtf(var + eps);
    // (eval mode too: the backward of the fixed affine map needs the mean and 1/sqrt(running_var + eps) it was taken with)
    if (save_mean) save_mean[n] = mean;
    if (save_invstd) save_invstd[n] = invstd;
    if (training) {
        if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean;
        if (running_var) running_var[n] = (1.f - momentum) * running_var[n] + momentum * var * (count > 1.f ? count / (count - 1.f) : 1.f);
    }
    const float g = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
    for (int m = 0; m < M; ++m) {
        float v = (x[(int64_t)m * N + n] - mean) * invstd * g + b;
        if (relu) v = fmaxf(v, 0.f);
        y[(int64_t)m * N + n] = v;
    }
}
// local backward sums: sums[0][n] = sum dz, sums[1][n] = sum dz * xhat, dz = dy * [y > 0] when relu
__global__ void bn1d_bwd_colsums_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                        const float* __restrict__ save_mean, const float* __restrict__ save_invstd, int relu,
                                        float* __restrict__ sums, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float mean = save_mean[n], invstd = save_invstd[n];
    float s = 0.f, q = 0.f;
    for (int m = 0; m < M; ++m) {
        const int64_t i = (int64_t)m * N + n;
        const float dz = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
        s += dz;
        q = fmaf(dz, (x[i] - mean) * invstd, q);
    }
    sums[n] = s;
    sums[N + n] = q;
}
// dx = gamma * invstd * (dz - S0 / count - xhat * S1 / count); dgamma = local sum dz * xhat, dbeta = local sum dz.
// sums (nullable): the totals over all ranks (count = total rows); NULL: the local sums are the totals.
__global__ void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                const float* __restrict__ save_mean, const float* __restrict__ save_invstd, const float* __restrict__ gamma,
                                int relu, const float* __restrict__ sums, float count, float* __restrict__ dx, float* __restrict__ dgamma,
                                float* __restrict__ dbeta, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float mean = save_mean[n], invstd = save_invstd[n];
    float s = 0.f, q = 0.f;
    for (int m = 0; m < M; ++m) {
        const int64_t i = (int64_t)m * N + n;
        const float dz = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
        s += dz;
        q = fmaf(dz, (x[i] - mean) * invstd, q);
    }
    if (dgamma) dgamma[n] = q;
    if (dbeta) dbeta[n] = s;
    float S0 = s, S1 = q;
    if (sums != nullptr) {
        S0 = sums[n];
        S1 = sums[N + n];
    } else {
        count = (float)M;
    }
    const float k = (gamma ? gamma[n] : 1.f) * invstd, c0 = S0 / count, c1 = S1 / count;
    for (int m = 0; m < M; ++m) {
        const int64_t i = (int64_t)m * N + n;
        const float dz = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
        dx[i] = k * (dz - c0 - (x[i] - mean) * invstd * c1);
    }
}

extern "C" int cmu_bn1d_colsums(const float* x, float* sums, int M, int N, void* stream) {
    CMU_CHECK_ARG(x && sums && M >= 1 && N >= 1, "cmu_bn1d_colsums: bad args");
    hipLaunchKernelGGL(bn1d_colsums_kernel, dim3(cmu_div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, x, sums, M, N);
    CMU_CHECK_LAUNCH("cmu_bn1d_colsums");
    return CMU_OK;
}
extern "C" int cmu_bn1d_relu_fwd(const float* x, const float* sums, int64_t count, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float momentum, float eps, int training, int relu, float* y, float* save_mean,
                                 float* save_invstd, int M, int N, void* stream) {
    CMU_CHECK_ARG(x && y && M >= 1 && N >= 1 && (training || (running_mean && running_var)) && (sums == nullptr || count >= 1),
                  "cmu_bn1d_relu_fwd: bad args");
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(cmu_div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, x, sums, (float)count, gamma, beta,
                       running_mean, running_var, momentum, eps, training, relu, y, save_mean, save_invstd, M, N);
    CMU_CHECK_LAUNCH("cmu_bn1d_relu_fwd");
    return CMU_OK;
}
extern "C" int cmu_bn1d_bwd_colsums(const float* dy, const float* x, const float* y, const float* save_mean, const float* save_invstd,
                                    int relu, float* sums, int M, int N, void* stream) {
    CMU_CHECK_ARG(dy && x && save_mean && save_invstd && sums && (!relu || y) && M >= 1 && N >= 1, "cmu_bn1d_bwd_colsums: bad args");
    hipLaunchKernelGGL(bn1d_bwd_colsums_kernel, dim3(cmu_div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, dy, x, y, save_mean, save_invstd,
                       relu, sums, M, N);
    CMU_CHECK_LAUNCH("cmu_bn1d_bwd_colsums");
    return CMU_OK;
}
extern "C" int cmu_bn1d_relu_bwd(const float* dy, const float* x, const float* y, const float* save_mean, const float* save_invstd,
                                 const float* gamma, int relu, const float* sums, int64_t count, float* dx, float* dgamma, float* dbeta,
                                 int M, int N, void* stream) {
    CMU_CHECK_ARG(dy && x && save_mean && save_invstd && dx && (!relu || y) && M >= 1 && N >= 1 && (sums == nullptr || count >= 1),
                  "cmu_bn1d_relu_bwd: bad args");
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(cmu_div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, dy, x, y, save_mean, save_invstd, gamma,
                       relu, sums, (float)count, dx, dgamma, dbeta, M, N);
    CMU_CHECK_LAUNCH("cmu_bn1d_relu_bwd");
    return CMU_OK;
}

// =====================================================================================================================
// 1x1 convolution NHWC (dt, pending transform) -> NCHW fp32
// =====================================================================================================================
// wave = 32 pixels x NT*32 output channels; A operand: one 16-byte chunk of a pixel's channel vector per lane and k-step,
// straight from global memory (NHWC: the MFMA's k-contiguous operand layout), transform applied in registers; B operand:
// fp32 weight rows converted to dt in registers (rows re-read from L2 by every wave: the matrix is <= 1 MB).
template <class TR, int NT>
__global__ __launch_bounds__(256) void conv1x1_nchw_kernel(const typename TR::elem_t* __restrict__ x, int64_t ldx, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, int relu_from, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int64_t npix, int HW,
                                                          int K, int N) {
    constexpr int EPC = TR::EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int n0 = blockIdx.y * (NT * 32);
    const int64_t pix = p0 + r;
    const bool pok = pix < npix;
    const typename TR::elem_t* xp = x + (pok ? pix : 0) * ldx + h * EPC;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2 * EPC) {
        const int kk = k0 + h * EPC;
        u32x4 a = u32x4{0u, 0u, 0u, 0u};
        if (pok && kk < K) {
            a = *reinterpret_cast<const u32x4*>(xp + k0);
            if (sc != nullptr) {
                float f[EPC];
                TR::unpack(a, f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    f[e] = fmaf(f[e], sc[kk + e], sh[kk + e]);
                    if (kk + e >= relu_from) f[e] = fmaxf(f[e], 0.f);
                }
                a = TR::pack(f);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + 32 * t + r;
            float f[EPC];
#pragma unroll
            for (int e = 0; e < EPC; e += 4) {
                const f32x4s v = (n < N && kk < K) ? *reinterpret_cast<const f32x4s*>(w + (int64_t)n * K + kk + e) : f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) f[e + q] = v[q];
            }
            TR::mma16(a, TR::pack(f), acc[t]);
        }
    }
    // D[m = pixel][n]: lane holds column n = n0 + 32 t + r, rows m = (e & 3) + 8 (e >> 2) + 4 h: four consecutive pixels per store
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 32 * t + r;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t pm = p0 + 8 * g + 4 * h;         // first of four pixels (HW % 4 == 0: they share an image)
            if (pm >= npix) continue;
            const int64_t b = pm / HW, yx = pm % HW;
            f32x4s v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[t][4 * g + q] + bv;
            *reinterpret_cast<f32x4s*>(out + ((int64_t)b * N + n) * HW + yx) = v;
        }
    }
}
template <class TR>
static int conv1x1_nchw_t(const void* x, int64_t ldx, const float* sc, const float* sh, int relu_from, const float* w, const float* bias,
                          float* out, int B, int H, int W, int K, int N, hipStream_t st) {
    const int64_t npix = (int64_t)B * H * W;
    constexpr int NT = 4;
    hipLaunchKernelGGL((conv1x1_nchw_kernel<TR, NT>), dim3((unsigned)cmu_div_up64(npix, 128), cmu_div_up(N, NT * 32)), dim3(256), 0, st,
                       (const typename TR::elem_t*)x, ldx, sc, sh, relu_from, w, bias, out, npix, H * W, K, N);
    CMU_CHECK_LAUNCH("cmu_conv1x1_nchw_fwd");
    return CMU_OK;
}
extern "C" int cmu_conv1x1_nchw_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from, const float* w,
                                    const float* bias, float* out, int B, int H, int W, int K, int N, int dt, void* stream) {
    CMU_CHECK_ARG(x && w && out && B > 0 && H > 0 && W > 0 && K > 0 && N > 0, "cmu_conv1x1_nchw_fwd: bad args");
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && K % (32 / es) == 0 && (ldx * es) % 16 == 0 && cmu_aligned16(x) && cmu_aligned16(w) && cmu_aligned16(out) &&
                      (H * W) % 4 == 0 && K % 4 == 0,
                  "cmu_conv1x1_nchw_fwd: K must be a whole number of 32-byte slices, H*W %% 4 == 0, 16-byte aligned rows");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_conv1x1_nchw_fwd: scale and shift come together");
    CMU_DISPATCH_DT(dt, conv1x1_nchw_t, x, ldx, in_scale, in_shift, relu_from, w, bias, out, B, H, W, K, N, (hipStream_t)stream);
}

// =====================================================================================================================
// skinny GEMMs with 16-bit operands (fp32 in memory, rounded in registers, fp32 accumulation)
// =====================================================================================================================
template <class TR>
__device__ static inline u32x4 sk16_pack8(const f32x4s& a, const f32x4s& b) {
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return TR::pack(f);
}
constexpr int SK16_MAX_M = 256;    // rows per call, in groups of 32 (skinny.hip: SK_MAX_M)
__device__ static inline f32x4s sk16_ld4(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const f32x4s*>(p) : f32x4s{0.f, 0.f, 0.f, 0.f};
}

// ---- forward: y (M,N) = x (M,K) . w (N,K)^T: wave = 32 weight rows over one K range, 16 k per MFMA -----------------------
constexpr int SK16_UNROLL = 4;    // 16-k steps in flight per wave (8 x 16-byte weight loads per lane)
template <class TR>
__global__ __launch_bounds__(256) void skinny16_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ slab,
                                                          int M, int N, int64_t K, int64_t kchunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int n = (blockIdx.x * 4 + wave) * 32 + c;
    const int64_t k0 = (int64_t)blockIdx.y * kchunk, k1 = k0 + kchunk < K ? k0 + kchunk : K;
    const bool nok = n < N, mok = c < M;
    const float* wp = w + (int64_t)(nok ? n : 0) * K + 8 * h;
    const float* xp = x + (int64_t)(mok ? c : 0) * K + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int64_t kb = k0; kb < k1; kb += 16 * SK16_UNROLL) {
        f32x4s wv[SK16_UNROLL][2], xv[SK16_UNROLL][2];
#pragma unroll
        for (int u = 0; u < SK16_UNROLL; ++u) {
            const int64_t k = kb + u * 16;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wv[u][q] = sk16_ld4(wp + k + 4 * q, nok && k + 8 * h + 4 * q < k1);
                xv[u][q] = sk16_ld4(xp + k + 4 * q, mok && k + 8 * h + 4 * q < k1);
            }
        }
#pragma unroll
        for (int u = 0; u < SK16_UNROLL; ++u) TR::mma16(sk16_pack8<TR>(xv[u][0], xv[u][1]), sk16_pack8<TR>(wv[u][0], wv[u][1]), acc);
    }
    float* out = slab + (int64_t)blockIdx.y * M * N;
    if (nok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < M) out[(int64_t)m * N + n] = acc[e];
        }
    }
}
__global__ void skinny16_fwd_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                           int splits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    float s = bias ? bias[i % N] : 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(int64_t)k * M * N + i];
    y[i] = s;
}
static int sk16_splits(int N, int64_t K, int64_t* kchunk) {
    const int nblk = cmu_div_up(N, 128);
    int splits = (int)cmu_div_up64(2048, nblk);                       // ~8 workgroups per CU: the loads are the only work
    int64_t kc = cmu_div_up64(K, splits);
    kc = cmu_div_up64(kc, 16 * SK16_UNROLL) * (16 * SK16_UNROLL);
    if (kc < 16 * SK16_UNROLL) kc = 16 * SK16_UNROLL;
    *kchunk = kc;
    return (int)cmu_div_up64(K, kc);
}
extern "C" int64_t cmu_skinny16_gemm_ws_bytes(int M, int N, int64_t K) {
    int64_t kc;
    return (int64_t)sk16_splits(N, K, &kc) * (M < 32 ? M : 32) * N * (int64_t)sizeof(float);     // one slab, reused by every 32-row group
}
template <class TR>
static int skinny16_fwd_t(const float* x, const float* w, const float* bias, float* y, int M, int N, int64_t K, void* ws, hipStream_t st) {
    int64_t kc;
    const int splits = sk16_splits(N, K, &kc);
    for (int m0 = 0; m0 < M; m0 += 32) {      // row groups of 32 over one slab (same stream)
        const int Mg = M - m0 < 32 ? M - m0 : 32;
        hipLaunchKernelGGL((skinny16_fwd_kernel<TR>), dim3(cmu_div_up(N, 128), splits), dim3(256), 0, st, x + (int64_t)m0 * K, w, (float*)ws, Mg, N, K, kc);
        CMU_CHECK_LAUNCH("cmu_skinny16_gemm_fwd");
        hipLaunchKernelGGL(skinny16_fwd_reduce_kernel, dim3((unsigned)cmu_div_up64((int64_t)Mg * N, 256)), dim3(256), 0, st, (const float*)ws, bias,
                           y + (int64_t)m0 * N, Mg, N, splits);
        CMU_CHECK_LAUNCH("cmu_skinny16_gemm_fwd(reduce)");
    }
    return CMU_OK;
}
extern "C" int cmu_skinny16_gemm_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int64_t K, int dt, void* ws,
                                     void* stream) {
    CMU_CHECK_ARG(x && w && y && ws && M >= 1 && M <= SK16_MAX_M && N >= 1 && K >= 16 && K % 16 == 0 && (dt == CMU_F16 || dt == CMU_BF16),
                  "cmu_skinny16_gemm_fwd: needs 1 <= M <= 256, K %% 16 == 0, dt f16 / bf16 (M=%d, K=%lld, dt=%d)", M, (long long)K, dt);
    CMU_CHECK_ARG(cmu_aligned16(x) && cmu_aligned16(w), "cmu_skinny16_gemm_fwd: x / w must be 16-byte aligned");
    if (dt == CMU_F16) return skinny16_fwd_t<F16Traits>(x, w, bias, y, M, N, K, ws, (hipStream_t)stream);
    return skinny16_fwd_t<BF16Traits>(x, w, bias, y, M, N, K, ws, (hipStream_t)stream);
}

// ---- input gradient: dx (M,K) = dy (M,N) . w (N,K); block = 128 columns of w, its four waves take a quarter of the N rows each
// and combine through LDS in wave order.  B operand of tile j: w[n0 + 8h + i][4c + j], i = 0..7 (eight 16-byte row reads per lane
// and 16 rows); A operand: dy[m = c][n0 + 8h .. + 7] read in place (no transposed copy).
template <class TR>
__global__ __launch_bounds__(256) void skinny16_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                            int M, int N, int64_t K) {
    __shared__ float red[4][4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t kcol = (int64_t)blockIdx.x * 128 + 4 * c;
    const bool kok = kcol < K, mok = c < M;
    const int nq = (((N + 3) / 4 + 15) / 16) * 16;                            // rows per wave (whole 16-row steps)
    const int nbeg = wave * nq, nend = nbeg + nq < N ? nbeg + nq : N;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int n0 = nbeg; n0 < nend; n0 += 16) {
        f32x4s wv[8];
        float av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + 8 * h + i;
            wv[i] = sk16_ld4(w + (int64_t)(n < nend ? n : 0) * K + kcol, kok && n < nend);
            av[i] = (mok && n < nend) ? dy[(int64_t)c * N + n] : 0.f;
        }
        const u32x4 a = TR::pack(av);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float bf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) bf[i] = wv[i][j];
            TR::mma16(a, TR::pack(bf), acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][j][e][lane] = acc[j][e];
    __syncthreads();
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
extern "C" int cmu_skinny16_gemm_dgrad(const float* dy, const float* w, float* dx, int M, int N, int64_t K, int dt, void* stream) {
    CMU_CHECK_ARG(dy && w && dx && M >= 1 && M <= SK16_MAX_M && N >= 1 && K >= 4 && K % 4 == 0 && (dt == CMU_F16 || dt == CMU_BF16),
                  "cmu_skinny16_gemm_dgrad: needs 1 <= M <= 256, K %% 4 == 0, dt f16 / bf16 (M=%d, K=%lld, dt=%d)", M, (long long)K, dt);
    CMU_CHECK_ARG(cmu_aligned16(w) && cmu_aligned16(dx), "cmu_skinny16_gemm_dgrad: w / dx must be 16-byte aligned");
    const dim3 grid((unsigned)cmu_div_up64(K, 128));
    for (int m0 = 0; m0 < M; m0 += 32) {      // row groups of 32: one pass over the weights each
        const int Mg = M - m0 < 32 ? M - m0 : 32;
        const float* dyg = dy + (int64_t)m0 * N;
        float* dxg = dx + (int64_t)m0 * K;
        if (dt == CMU_F16) hipLaunchKernelGGL((skinny16_dgrad_kernel<F16Traits>), grid, dim3(256), 0, (hipStream_t)stream, dyg, w, dxg, Mg, N, K);
        else hipLaunchKernelGGL((skinny16_dgrad_kernel<BF16Traits>), grid, dim3(256), 0, (hipStream_t)stream, dyg, w, dxg, Mg, N, K);
        CMU_CHECK_LAUNCH("cmu_skinny16_gemm_dgrad");
    }
    return CMU_OK;
}

// ---- weight gradient: dw (N,K) = dy^T (N,M) . x (M,K): wave = 32 * NT rows n x 128 columns k, contraction over the M <= 32 rows in
// two 16-row MFMAs.  B operand of tile j, MFMA s: x[m = 16 s + 8 h + i][4c + j]; A operand: dy[m = 16 s + 8 h + i][n0 + r].
constexpr int SK16W_NT = 2;
// MULTI (round 4, M > 32): row groups of 32 with both n tiles' accumulators live; the one-group form keeps one tile's at a time
template <class TR, bool MULTI>
__global__ __launch_bounds__(256) void skinny16_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                                            int M, int N, int64_t K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t kcol = ((int64_t)blockIdx.x * 4 + wave) * 128 + 4 * c;
    const bool kok = kcol < K;
    const int nbase = blockIdx.y * 32 * SK16W_NT;
    if (!MULTI) {
        u32x4 xb[2][4];     // [MFMA s][tile j]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4s xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 16 * s + 8 * h + i;
                xv[i] = sk16_ld4(x + (int64_t)(m < M ? m : 0) * K + kcol, kok && m < M);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bf[i] = xv[i][j];
                xb[s][j] = TR::pack(bf);
            }
        }
#pragma unroll
        for (int t = 0; t < SK16W_NT; ++t) {
            const int n = nbase + 32 * t + c;
            f32x16 acc1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float af[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = 16 * s + 8 * h + i;
                    af[i] = (n < N && m < M) ? dy[(int64_t)m * N + n] : 0.f;
                }
                const u32x4 a = TR::pack(af);
#pragma unroll
                for (int j = 0; j < 4; ++j) TR::mma16(a, xb[s][j], acc1[j]);
            }
            if (kok) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int nr = nbase + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (nr < N) __builtin_nontemporal_store(f32x4s{acc1[0][e], acc1[1][e], acc1[2][e], acc1[3][e]}, reinterpret_cast<f32x4s*>(dw + (int64_t)nr * K + kcol));
                }
            }
        }
        return;
    }
    f32x16 acc[SK16W_NT][4];
#pragma unroll
    for (int t = 0; t < SK16W_NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;
    for (int m0 = 0; m0 < M; m0 += 32) {      // row groups of 32 (one trip for the <= 32 rows the kernel was written for)
        u32x4 xb[2][4];     // [MFMA s][tile j]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4s xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + 16 * s + 8 * h + i;
                xv[i] = sk16_ld4(x + (int64_t)(m < M ? m : 0) * K + kcol, kok && m < M);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bf[i] = xv[i][j];
                xb[s][j] = TR::pack(bf);
            }
        }
#pragma unroll
        for (int t = 0; t < SK16W_NT; ++t) {
            const int n = nbase + 32 * t + c;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float af[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = m0 + 16 * s + 8 * h + i;
                    af[i] = (n < N && m < M) ? dy[(int64_t)m * N + n] : 0.f;
                }
                const u32x4 a = TR::pack(af);
#pragma unroll
                for (int j = 0; j < 4; ++j) TR::mma16(a, xb[s][j], acc[t][j]);
            }
        }
    }
    if (kok) {
#pragma unroll
        for (int t = 0; t < SK16W_NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int nr = nbase + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (nr < N) __builtin_nontemporal_store(f32x4s{acc[t][0][e], acc[t][1][e], acc[t][2][e], acc[t][3][e]}, reinterpret_cast<f32x4s*>(dw + (int64_t)nr * K + kcol));
            }
    }
}
__global__ void skinny16_colsum_kernel(const float* __restrict__ dy, float* __restrict__ dbias, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[(int64_t)m * N + n];
    dbias[n] = s;
}
extern "C" int cmu_skinny16_gemm_wgrad(const float* dy, const float* x, float* dw, float* dbias, int M, int N, int64_t K, int dt, void* stream) {
    CMU_CHECK_ARG(dy && x && dw && M >= 1 && M <= SK16_MAX_M && N >= 1 && K >= 4 && K % 4 == 0 && (dt == CMU_F16 || dt == CMU_BF16),
                  "cmu_skinny16_gemm_wgrad: needs 1 <= M <= 256, K %% 4 == 0, dt f16 / bf16 (M=%d, K=%lld, dt=%d)", M, (long long)K, dt);
    CMU_CHECK_ARG(cmu_aligned16(x) && cmu_aligned16(dw), "cmu_skinny16_gemm_wgrad: x / dw must be 16-byte aligned");
    const dim3 grid((unsigned)cmu_div_up64(K, 512), cmu_div_up(N, 32 * SK16W_NT));
    if (dt == CMU_F16) {
        if (M <= 32) hipLaunchKernelGGL((skinny16_wgrad_kernel<F16Traits, false>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, dw, M, N, K);
        else hipLaunchKernelGGL((skinny16_wgrad_kernel<F16Traits, true>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, dw, M, N, K);
    } else {
        if (M <= 32) hipLaunchKernelGGL((skinny16_wgrad_kernel<BF16Traits, false>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, dw, M, N, K);
        else hipLaunchKernelGGL((skinny16_wgrad_kernel<BF16Traits, true>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, dw, M, N, K);
    }
    CMU_CHECK_LAUNCH("cmu_skinny16_gemm_wgrad");
    if (dbias != nullptr) {
        hipLaunchKernelGGL(skinny16_colsum_kernel, dim3(cmu_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, dy, dbias, M, N);
        CMU_CHECK_LAUNCH("cmu_skinny16_gemm_wgrad(bias)");
    }
    return CMU_OK;
}
