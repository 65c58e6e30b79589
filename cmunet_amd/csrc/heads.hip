// heads.hip -- losses and pretraining heads of the CM-UNet hot path (gfx950), all fp32/fp64 arithmetic:
//   masked reconstruction loss (cmunet_head.py:62-70), finetune softmax-CE + Dice/IoU counters
//   (metrics.py:135-180,503), in-batch InfoNCE (cmunet_head.py:72-88), MoCo InfoNCE + queue update in
//   one launch (moco2_module.py:160-175,256-285), L2 row normalisation, EMA and Adam over flat arenas.
// These are latency/HBM-bound: one pass per tensor, wave64 shuffle reductions, fixed-order final sums.
#include "common.h"

__device__ static inline float block_sum(float v, float* red /*[4]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float a = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) a += red[i];
    return a;
}
__device__ static inline double block_sum_d(double v, double* red /*[4]*/) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double a = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) a += red[i];
    return a;
}
__device__ static inline float block_max(float v, float* red /*[4]*/) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float a = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) a = fmaxf(a, red[i]);
    return a;
}

// ---------------------------------------------------------------------------------------------
// masked MSE: ws = [rows][4] floats (mean, rstd, num, den) + [2] (loss, den_total)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mmse_rows_kernel(const float* __restrict__ logits, int K, int channel,
                                                       const float* __restrict__ img, const uint8_t* __restrict__ mask,
                                                       float* __restrict__ ws, int B, int H, int W) {
    __shared__ float red[4];
    const int row = blockIdx.x;  // b*H + y
    const int b = row / H, yy = row % H;
    const float* im = img + (int64_t)row * W;
    const float* pr = logits + (((int64_t)b * K + channel) * H + yy) * W;
    const uint8_t* mk = mask + (int64_t)row * W;
    float s = 0.f;
    for (int x = threadIdx.x; x < W; x += 256) s += im[x];
    const float mean = block_sum(s, red) / (float)W;
    float v = 0.f;
    for (int x = threadIdx.x; x < W; x += 256) {
        const float d = im[x] - mean;
        v = fmaf(d, d, v);
    }
    const float var = block_sum(v, red) / (float)(W > 1 ? W - 1 : 1);  // unbiased (torch.var default)
    const float rstd = 1.f / sqrtf(var + 1.e-6f);
    float num = 0.f, den = 0.f;
    for (int x = threadIdx.x; x < W; x += 256) {
        const float t = (im[x] - mean) * rstd;
        const float d = pr[x] - t;
        const float m = (float)mk[x];
        num = fmaf(d * d, m, num);
        den += m;
    }
    num = block_sum(num, red);
    den = block_sum(den, red);
    if (threadIdx.x == 0) {
        ws[row * 4 + 0] = mean;
        ws[row * 4 + 1] = rstd;
        ws[row * 4 + 2] = num;
        ws[row * 4 + 3] = den;
    }
}
// the same per-row quantities with one WAVE per image row (rows of whole 16-byte groups, at most 1,024 pixels): the row is read
// once into registers (the block form walks it three times between four block-wide sums with two barriers each: 56 us for the
// 16,384 rows of a bench batch)
__global__ __launch_bounds__(256) void mmse_rows_wave_kernel(const float* __restrict__ logits, int K, int channel,
                                                            const float* __restrict__ img, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ ws, int B, int H, int W, int rows) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / H, yy = row % H;
    const float* im = img + (int64_t)row * W;
    const float* pr = logits + (((int64_t)b * K + channel) * H + yy) * W;
    const uint8_t* mk = mask + (int64_t)row * W;
    f32x4 iv[4], pv[4];
    uint32_t mv[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = (i * 64 + lane) * 4;
        const bool ok = x < W;
        iv[i] = ok ? *reinterpret_cast<const f32x4*>(im + x) : f32x4{0.f, 0.f, 0.f, 0.f};
        pv[i] = ok ? *reinterpret_cast<const f32x4*>(pr + x) : f32x4{0.f, 0.f, 0.f, 0.f};
        mv[i] = ok ? *reinterpret_cast<const uint32_t*>(mk + x) : 0u;
        s += (iv[i][0] + iv[i][1]) + (iv[i][2] + iv[i][3]);
    }
    const float mean = wave_sum(s) / (float)W;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((i * 64 + lane) * 4 < W)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = iv[i][j] - mean;
                v = fmaf(d, d, v);
            }
    const float var = wave_sum(v) / (float)(W > 1 ? W - 1 : 1);  // unbiased (torch.var default)
    const float rstd = 1.f / sqrtf(var + 1.e-6f);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((i * 64 + lane) * 4 < W)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = (iv[i][j] - mean) * rstd;
                const float d = pv[i][j] - t;
                const float m = (float)((mv[i] >> (8 * j)) & 0xffu);
                num = fmaf(d * d, m, num);
                den += m;
            }
    num = wave_sum(num);
    den = wave_sum(den);
    if (lane == 0) {
        ws[row * 4 + 0] = mean;
        ws[row * 4 + 1] = rstd;
        ws[row * 4 + 2] = num;
        ws[row * 4 + 3] = den;
    }
}
__global__ __launch_bounds__(256) void mmse_final_kernel(float* __restrict__ ws, int rows, float* loss) {
    __shared__ double red[4];
    double num = 0.0, den = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        num += (double)ws[r * 4 + 2];
        den += (double)ws[r * 4 + 3];
    }
    num = block_sum_d(num, red);
    den = block_sum_d(den, red);
    if (threadIdx.x == 0) {
        loss[0] = (float)(num / den);
        ws[rows * 4 + 0] = (float)(num / den);
        ws[rows * 4 + 1] = (float)den;
    }
}
__global__ void mmse_grad_kernel(const float* __restrict__ logits, int K, int channel, const float* __restrict__ img,
                                 const uint8_t* __restrict__ mask, const float* __restrict__ ws, float* __restrict__ dlogits,
                                 float loss_scale, const CmuAmpState* __restrict__ amp, int B, int H, int W, int64_t total) {
    const int rows = B * H;
    if (amp != nullptr) loss_scale *= amp->scale;      // dynamic loss scale (power of two: exact)
    const float k = 2.f * loss_scale / ws[rows * 4 + 1];
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(o % W), yy = (int)((o / W) % H), c = (int)((o / ((int64_t)W * H)) % K), b = (int)(o / ((int64_t)W * H * K));
        float g = 0.f;
        if (c == channel) {
            const int row = b * H + yy;
            const float t = (img[(int64_t)row * W + x] - ws[row * 4 + 0]) * ws[row * 4 + 1];
            g = k * (logits[o] - t) * (float)mask[(int64_t)row * W + x];
        }
        dlogits[o] = g;
    }
}
// four pixels per thread (W a multiple of 4: a 16-byte group never leaves its row)
__global__ void mmse_grad4_kernel(const float* __restrict__ logits, int K, int channel, const float* __restrict__ img,
                                  const uint8_t* __restrict__ mask, const float* __restrict__ ws, float* __restrict__ dlogits,
                                  float loss_scale, const CmuAmpState* __restrict__ amp, int B, int H, int W, int64_t total4) {
    const int rows = B * H, W4 = W / 4;
    if (amp != nullptr) loss_scale *= amp->scale;      // dynamic loss scale (power of two: exact)
    const float k = 2.f * loss_scale / ws[rows * 4 + 1];
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total4; o += (int64_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(o % W4), yy = (int)((o / W4) % H), c = (int)((o / ((int64_t)W4 * H)) % K), b = (int)(o / ((int64_t)W4 * H * K));
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (c == channel) {
            const int row = b * H + yy;
            const f32x4 iv = *reinterpret_cast<const f32x4*>(img + (int64_t)row * W + 4 * x4);
            const f32x4 lv = *reinterpret_cast<const f32x4*>(logits + 4 * o);
            const uint32_t mv = *reinterpret_cast<const uint32_t*>(mask + (int64_t)row * W + 4 * x4);
            const float mean = ws[row * 4 + 0], rstd = ws[row * 4 + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = (iv[j] - mean) * rstd;
                g[j] = k * (lv[j] - t) * (float)((mv >> (8 * j)) & 0xffu);
            }
        }
        *reinterpret_cast<f32x4*>(dlogits + 4 * o) = g;
    }
}
extern "C" int64_t cmu_masked_mse_ws_bytes(int B, int H) { return ((int64_t)B * H * 4 + 4) * (int64_t)sizeof(float); }
extern "C" int cmu_masked_mse_fwd_bwd(const float* logits, int K, int channel, const float* img, const uint8_t* mask, float* loss,
                                      float* dlogits, float loss_scale, const void* amp_state, int B, int H, int W, void* ws,
                                      void* stream) {
    CMU_CHECK_ARG(logits && img && mask && loss && ws && B > 0 && H > 0 && W > 0 && K > 0 && channel >= 0 && channel < K,
                  "cmu_masked_mse_fwd_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (W % 4 == 0 && W <= 1024 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(img)) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(mask) & 3) == 0)
        hipLaunchKernelGGL(mmse_rows_wave_kernel, dim3(cmu_div_up(B * H, 4)), dim3(256), 0, st, logits, K, channel, img, mask, (float*)ws, B, H, W,
                           B * H);
    else
        hipLaunchKernelGGL(mmse_rows_kernel, dim3(B * H), dim3(256), 0, st, logits, K, channel, img, mask, (float*)ws, B, H, W);
    CMU_CHECK_LAUNCH("cmu_masked_mse(rows)");
    hipLaunchKernelGGL(mmse_final_kernel, dim3(1), dim3(256), 0, st, (float*)ws, B * H, loss);
    CMU_CHECK_LAUNCH("cmu_masked_mse(final)");
    if (dlogits) {
        const int64_t total = (int64_t)B * K * H * W;
        if (W % 4 == 0 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(mask) & 3) == 0) {
            const int grid = (int)(cmu_div_up64(total / 4, 256) < 8192 ? cmu_div_up64(total / 4, 256) : 8192);
            hipLaunchKernelGGL(mmse_grad4_kernel, dim3(grid), dim3(256), 0, st, logits, K, channel, img, mask, (const float*)ws, dlogits,
                               loss_scale, (const CmuAmpState*)amp_state, B, H, W, total / 4);
            CMU_CHECK_LAUNCH("cmu_masked_mse(grad)");
            return CMU_OK;
        }
        const int grid = (int)(cmu_div_up64(total, 256) < 8192 ? cmu_div_up64(total, 256) : 8192);
        hipLaunchKernelGGL(mmse_grad_kernel, dim3(grid), dim3(256), 0, st, logits, K, channel, img, mask, (const float*)ws, dlogits,
                           loss_scale, (const CmuAmpState*)amp_state, B, H, W, total);
        CMU_CHECK_LAUNCH("cmu_masked_mse(grad)");
    }
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// softmax CE (probability targets) + Dice / IoU counters, 2 classes
// ws: [blocks][4] doubles (ce, tp, sum_pr, sum_gt)
// ---------------------------------------------------------------------------------------------
constexpr int CE_MAX_BLOCKS = 1024;
__global__ __launch_bounds__(256) void ce_dice_kernel(const float* __restrict__ logits, const double* __restrict__ y1h,
                                                     double* __restrict__ ws, float* __restrict__ dlogits, float gscale, int64_t HW,
                                                     int64_t npix) {
    __shared__ double red[4];
    double ce = 0.0, tp = 0.0, spr = 0.0, sgt = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / HW, r = p % HW;
        const int64_t i0 = (b * 2) * HW + r, i1 = i0 + HW;
        const float l0 = logits[i0], l1 = logits[i1];
        const double y0 = y1h[i0], y1 = y1h[i1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float se = e0 + e1;
        const float lse = m + logf(se);
        const float p0 = e0 / se, p1 = e1 / se;
        ce -= y0 * (double)(l0 - lse) + y1 * (double)(l1 - lse);
        const double pr = p1 > 0.5f ? 1.0 : 0.0;
        tp += y1 * pr;
        spr += pr;
        sgt += y1;
        if (dlogits) {
            const float ys = (float)(y0 + y1);
            dlogits[i0] = gscale * (p0 * ys - (float)y0);
            dlogits[i1] = gscale * (p1 * ys - (float)y1);
        }
    }
    ce = block_sum_d(ce, red);
    tp = block_sum_d(tp, red);
    spr = block_sum_d(spr, red);
    sgt = block_sum_d(sgt, red);
    if (threadIdx.x == 0) {
        ws[blockIdx.x * 4 + 0] = ce;
        ws[blockIdx.x * 4 + 1] = tp;
        ws[blockIdx.x * 4 + 2] = spr;
        ws[blockIdx.x * 4 + 3] = sgt;
    }
}
__global__ void ce_dice_final_kernel(const double* __restrict__ ws, int nblocks, double npix, float* out) {
    // one wave, fixed order (lane l takes rows l, l + 64, ...; then the wave's butterfly): the one-thread loop over up to 1,024 rows
    // of dependent loads took 140 us -- 2 % of a finetuning step at 256 x 256
    double ce = 0.0, tp = 0.0, spr = 0.0, sgt = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        ce += ws[b * 4 + 0];
        tp += ws[b * 4 + 1];
        spr += ws[b * 4 + 2];
        sgt += ws[b * 4 + 3];
    }
    ce = wave_sum_d(ce);
    tp = wave_sum_d(tp);
    spr = wave_sum_d(spr);
    sgt = wave_sum_d(sgt);
    if (threadIdx.x != 0) return;
    const double fp = spr - tp, fn = sgt - tp;
    out[0] = (float)(ce / npix);
    out[1] = (float)(1.0 - (2.0 * tp + 1e-5) / (2.0 * tp + fn + fp + 1e-5));   // metrics.py:135-157, beta=1, eps 1e-5
    out[2] = (float)(1.0 - (tp + 1e-7) / (sgt + spr - tp + 1e-7));              // metrics.py:182-198, eps 1e-7
    out[3] = (float)tp;
    out[4] = (float)spr;
    out[5] = (float)sgt;
}
extern "C" int64_t cmu_softmax_ce_dice_ws_bytes(int B, int H, int W) { return (int64_t)CE_MAX_BLOCKS * 4 * (int64_t)sizeof(double); }
extern "C" int cmu_softmax_ce_dice_fwd_bwd(const float* logits, const double* y1h, float* out, float* dlogits, float loss_scale, int B,
                                           int H, int W, void* ws, void* stream) {
    CMU_CHECK_ARG(logits && y1h && out && ws && B > 0 && H > 0 && W > 0, "cmu_softmax_ce_dice_fwd_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int64_t npix = (int64_t)B * H * W;
    const int grid = (int)(cmu_div_up64(npix, 256) < CE_MAX_BLOCKS ? cmu_div_up64(npix, 256) : CE_MAX_BLOCKS);
    hipLaunchKernelGGL(ce_dice_kernel, dim3(grid), dim3(256), 0, st, logits, y1h, (double*)ws, dlogits, loss_scale / (float)npix,
                       (int64_t)H * W, npix);
    CMU_CHECK_LAUNCH("cmu_softmax_ce_dice");
    hipLaunchKernelGGL(ce_dice_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, grid, (double)npix, out);
    CMU_CHECK_LAUNCH("cmu_softmax_ce_dice(final)");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// L2 row normalisation (F.normalize(dim=1), eps 1e-12)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int D) {
    __shared__ float red[4];
    const float* r = x + (int64_t)blockIdx.x * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s = fmaf(r[d], r[d], s);
    const float inv = 1.f / fmaxf(sqrtf(block_sum(s, red)), 1e-12f);
    for (int d = threadIdx.x; d < D; d += 256) out[(int64_t)blockIdx.x * D + d] = r[d] * inv;
}
extern "C" int cmu_l2_normalize_rows(const float* x, float* out, int B, int D, void* stream) {
    CMU_CHECK_ARG(x && out && B > 0 && D > 0, "cmu_l2_normalize_rows: bad args");
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, out, D);
    CMU_CHECK_LAUNCH("cmu_l2_normalize_rows");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// in-batch InfoNCE (CM-UNet): one block per query row; loss[0] = total, loss[1+b] = per-row terms
// ---------------------------------------------------------------------------------------------
constexpr int NCE_MAX_N = 8192;
__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ pred, const float* __restrict__ keys,
                                                          float* __restrict__ loss, float* __restrict__ dpred, int B, int N, int D,
                                                          int rank, float temp, float ct_w) {
    extern __shared__ float sm[];  // [D] normalised query, [N] scores / probabilities
    __shared__ float red[4];
    float* qn = sm;
    float* sc = sm + D;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = pred + (int64_t)b * D;
    float s = 0.f;
    for (int d = tid; d < D; d += 256) s = fmaf(p[d], p[d], s);
    const float nrm = fmaxf(sqrtf(block_sum(s, red)), 1e-12f);
    for (int d = tid; d < D; d += 256) qn[d] = p[d] / nrm;
    __syncthreads();
    // scores: one wave per key, lanes over D (coalesced key rows)
    const int lane = tid & 63, wave = tid >> 6;
    for (int n = wave; n < N; n += 4) {
        const float* k = keys + (int64_t)n * D;
        float a = 0.f;
        for (int d = lane; d < D; d += 64) a = fmaf(qn[d], k[d], a);
        a = wave_sum(a);
        if (lane == 0) sc[n] = a / temp;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int n = tid; n < N; n += 256) m = fmaxf(m, sc[n]);
    m = block_max(m, red);
    float se = 0.f;
    for (int n = tid; n < N; n += 256) se += expf(sc[n] - m);
    se = block_sum(se, red);
    const int label = b + B * rank;
    const float lse = m + logf(se);
    const float coef = ct_w * 2.f * temp;
    if (tid == 0) loss[1 + b] = coef * (lse - sc[label]) / (float)B;
    if (dpred == nullptr) return;
    // d loss / d score = coef/B * (softmax - onehot); d score / d qn = key / temp
    __syncthreads();
    for (int n = tid; n < N; n += 256) sc[n] = (coef / (float)B) * (expf(sc[n] - m) / se - (n == label ? 1.f : 0.f)) / temp;
    __syncthreads();
    // dqn[d] = sum_n g[n] * key[n][d]; then through the normalisation: (dqn - qn*(qn.dqn)) / nrm
    float dot = 0.f;
    for (int d = tid; d < D; d += 256) {
        float a = 0.f;
        for (int n = 0; n < N; ++n) a = fmaf(sc[n], keys[(int64_t)n * D + d], a);
        dpred[(int64_t)b * D + d] = a;  // temporarily dqn
        dot = fmaf(a, qn[d], dot);
    }
    dot = block_sum(dot, red);
    for (int d = tid; d < D; d += 256) dpred[(int64_t)b * D + d] = (dpred[(int64_t)b * D + d] - qn[d] * dot) / nrm;
}
__global__ void sum_small_kernel(float* loss, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += (double)loss[1 + b];
        loss[0] = (float)s;
    }
}
extern "C" int cmu_infonce_inbatch_fwd_bwd(const float* pred, const float* keys, float* loss, float* dpred, int B, int N, int D,
                                           int rank, float temperature, float ct_weight, void* stream) {
    CMU_CHECK_ARG(pred && keys && loss && B > 0 && N > 0 && D > 0 && rank >= 0 && temperature > 0.f, "cmu_infonce_inbatch_fwd_bwd: bad args");
    CMU_CHECK_ARG(N <= NCE_MAX_N && D <= 4096, "cmu_infonce_inbatch_fwd_bwd: N=%d (max %d) / D=%d (max 4096) too large", N, NCE_MAX_N, D);
    CMU_CHECK_ARG(B * (rank + 1) <= N, "cmu_infonce_inbatch_fwd_bwd: labels i + B*rank exceed N=%d", N);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(D + N) * sizeof(float);
    hipLaunchKernelGGL(infonce_rows_kernel, dim3(B), dim3(256), lds, st, pred, keys, loss, dpred, B, N, D, rank, temperature, ct_weight);
    CMU_CHECK_LAUNCH("cmu_infonce_inbatch");
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, st, loss, B);
    CMU_CHECK_LAUNCH("cmu_infonce_inbatch(sum)");
    return CMU_OK;
}

// (MoCo's InfoNCE + queue update: moco.hip)

// ---------------------------------------------------------------------------------------------
// EMA and Adam over flat fp32 arenas (float4 per lane)
// ---------------------------------------------------------------------------------------------
// cmunet.py:85-86 order: p_t*m + p_o*(1-m); the contraction is spelled out so that the stand-alone kernel and the form fused into the
// AdamW kernel (cmu_adam_ema_step) round identically
__device__ __forceinline__ float cmu_ema1(float t, float o, float m, float om) { return fmaf(t, m, o * om); }
__global__ void ema_kernel(float* __restrict__ t, const float* __restrict__ o, int64_t n, float m) {
    const int64_t n4 = n >> 2;
    const float om = 1.f - m;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 a = reinterpret_cast<f32x4*>(t)[i];
        const f32x4 b = reinterpret_cast<const f32x4*>(o)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = cmu_ema1(a[e], b[e], m, om);
        reinterpret_cast<f32x4*>(t)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        t[i] = cmu_ema1(t[i], o[i], m, om);
    }
}
extern "C" int cmu_ema_update(float* target, const float* online, int64_t n, float momentum, void* stream) {
    CMU_CHECK_ARG(target && online && n > 0 && cmu_aligned16(target) && cmu_aligned16(online), "cmu_ema_update: bad args / alignment");
    const int64_t nb = cmu_div_up64(n >> 2, 256);
    const int grid = (int)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096);
    hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, target, online, n, momentum);
    CMU_CHECK_LAUNCH("cmu_ema_update");
    return CMU_OK;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const uint8_t* __restrict__ wd_mask, int64_t n, float lr, float b1, float b2, float eps, float wd,
                            int decoupled, float bc1, float bc2_sqrt, float gscale, const CmuAmpState* __restrict__ amp) {
    if (amp != nullptr) {
        // GradScaler.step: skip the whole update when the gradients hold an inf / nan; else unscale.  The step number of the
        // bias corrections is the count of updates actually taken (a skipped step never reaches optimizer.step()).
        if (amp->found_inf != 0.f) return;
        gscale /= amp->scale;
        const double t = (double)(amp->good_steps + 1);
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i], gi = g[i] * gscale;
        const float w = (wd_mask == nullptr || wd_mask[i]) ? wd : 0.f;
        if (decoupled) pi *= (1.f - lr * w);
        else gi = fmaf(w, pi, gi);
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);            // exp_avg.lerp_(grad, 1-beta1)
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}
extern "C" int cmu_adam_step(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int decoupled, int64_t step, float grad_scale,
                             const void* amp_state, void* stream) {
    CMU_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "cmu_adam_step: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t nb = cmu_div_up64(n, 256);
    const int grid = (int)(nb < 8192 ? nb : 8192);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, wd_mask, n, lr, beta1, beta2, eps,
                       weight_decay, decoupled, (float)bc1, (float)sqrt(bc2), grad_scale, (const CmuAmpState*)amp_state);
    CMU_CHECK_LAUNCH("cmu_adam_step");
    return CMU_OK;
}

// AdamW + the EMA of the momentum networks in ONE pass over the arena (round 3; the joint CM-UNet step: 447 M parameters, of which
// 422 M -- backbone and projector -- have a target copy): the EMA as its own launch re-reads the 1.7 GB of parameters the optimiser
// has just written.  Up to two segments [lo, hi) of the online arena map onto target arrays; elements outside them are a plain
// AdamW update.  Float4 per lane (every tensor of a FlatParams arena starts on a 16-byte boundary and the arena is a multiple of four
// elements).  A skipped step (amp found_inf) still runs the EMA with the unchanged parameters, as MomentumUpdateHook.after_train_iter
// does behind a skipped optimiser step.  Element arithmetic = adam_kernel's and ema_kernel's: bit-identical to the two launches.
struct AdamEmaSeg {
    int64_t lo4[2], hi4[2];   // segment bounds in float4 units
    float* target[2];
    int nseg;
    float momentum;
};
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, const uint8_t* __restrict__ wd_mask, int64_t n4, float lr,
                                                       float b1, float b2, float eps, float wd, int decoupled, float bc1, float bc2_sqrt,
                                                       float gscale, const CmuAmpState* __restrict__ amp, const AdamEmaSeg seg) {
    bool skip = false;
    if (amp != nullptr) {
        skip = amp->found_inf != 0.f;
        gscale /= amp->scale;
        const double t = (double)(amp->good_steps + 1);
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    const float em = seg.momentum, eom = 1.f - seg.momentum;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float* tp = nullptr;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < seg.nseg && i >= seg.lo4[k] && i < seg.hi4[k]) tp = seg.target[k] + ((i - seg.lo4[k]) << 2);
        if (skip && tp == nullptr) continue;
        f32x4 pv = reinterpret_cast<const f32x4*>(p)[i];
        if (!skip) {
            const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
            f32x4 mv = reinterpret_cast<const f32x4*>(m)[i], vv = reinterpret_cast<const f32x4*>(v)[i];
            const uint32_t wm = wd_mask != nullptr ? reinterpret_cast<const uint32_t*>(wd_mask)[i] : 0x01010101u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pi = pv[e], gi = gv[e] * gscale;
                const float w = ((wm >> (8 * e)) & 0xffu) ? wd : 0.f;
                if (decoupled) pi *= (1.f - lr * w);
                else gi = fmaf(w, pi, gi);
                const float mi = fmaf(b1, mv[e], (1.f - b1) * gi);
                const float vi = fmaf(b2, vv[e], (1.f - b2) * gi * gi);
                mv[e] = mi;
                vv[e] = vi;
                const float denom = sqrtf(vi) / bc2_sqrt + eps;
                pv[e] = pi - step_size * (mi / denom);
            }
            reinterpret_cast<f32x4*>(m)[i] = mv;
            reinterpret_cast<f32x4*>(v)[i] = vv;
            reinterpret_cast<f32x4*>(p)[i] = pv;
        }
        if (tp != nullptr) {
            f32x4 tv = *reinterpret_cast<const f32x4*>(tp);
#pragma unroll
            for (int e = 0; e < 4; ++e) tv[e] = cmu_ema1(tv[e], pv[e], em, eom);
            *reinterpret_cast<f32x4*>(tp) = tv;
        }
    }
}
extern "C" int cmu_adam_ema_step(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int decoupled, int64_t step, float grad_scale,
                                 const void* amp_state, int nseg, const int64_t* seg_lo, const int64_t* seg_hi, float* const* seg_target,
                                 float ema_momentum, void* stream) {
    CMU_CHECK_ARG(p && g && m && v && n > 0 && step >= 1 && (n & 3) == 0, "cmu_adam_ema_step: bad args (n must be a multiple of 4)");
    CMU_CHECK_ARG(cmu_aligned16(p) && cmu_aligned16(g) && cmu_aligned16(m) && cmu_aligned16(v) && (wd_mask == nullptr || ((uintptr_t)wd_mask & 3) == 0),
                  "cmu_adam_ema_step: arenas must be 16-byte aligned");
    CMU_CHECK_ARG(nseg >= 0 && nseg <= 2 && (nseg == 0 || (seg_lo && seg_hi && seg_target)), "cmu_adam_ema_step: at most two EMA segments");
    AdamEmaSeg seg;
    seg.nseg = nseg;
    seg.momentum = ema_momentum;
    for (int k = 0; k < 2; ++k) { seg.lo4[k] = seg.hi4[k] = 0; seg.target[k] = nullptr; }
    for (int k = 0; k < nseg; ++k) {
        CMU_CHECK_ARG(seg_lo[k] >= 0 && seg_hi[k] > seg_lo[k] && seg_hi[k] <= n && (seg_lo[k] & 3) == 0 && (seg_hi[k] & 3) == 0 && seg_target[k] &&
                      cmu_aligned16(seg_target[k]), "cmu_adam_ema_step: segment bounds must be multiples of 4 inside [0, n], targets 16-byte aligned");
        CMU_CHECK_ARG(k == 0 || seg_lo[k] >= seg_hi[k - 1], "cmu_adam_ema_step: segments must be ascending and disjoint");
        seg.lo4[k] = seg_lo[k] >> 2;
        seg.hi4[k] = seg_hi[k] >> 2;
        seg.target[k] = seg_target[k];
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t nb = cmu_div_up64(n >> 2, 256);
    const int grid = (int)(nb < 16384 ? nb : 16384);
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, wd_mask, n >> 2, lr, beta1, beta2, eps,
                       weight_decay, decoupled, (float)bc1, (float)sqrt(bc2), grad_scale, (const CmuAmpState*)amp_state, seg);
    CMU_CHECK_LAUNCH("cmu_adam_ema_step");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// Dynamic loss scaling (AmpOptimWrapper(loss_scale='dynamic') of cmunet_config.py:76-78 = torch.cuda.amp.GradScaler):
// the state lives on the device, nothing here synchronises with the host.
// ---------------------------------------------------------------------------------------------
__global__ void amp_init_kernel(CmuAmpState* s, float init_scale) {
    s->scale = init_scale;
    s->found_inf = 0.f;
    s->growth_tracker = 0;
    s->good_steps = 0;
    s->skipped_steps = 0;
    s->pad[0] = s->pad[1] = s->pad[2] = 0;
}
__global__ __launch_bounds__(256) void amp_check_kernel(const float* __restrict__ g, int64_t n, CmuAmpState* __restrict__ s) {
    const int64_t n4 = n >> 2;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= !(fabsf(a[e]) <= 3.4028234664e38f);    // inf or nan
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= !(fabsf(g[(n4 << 2) + threadIdx.x]) <= 3.4028234664e38f);
    if (__any(bad) && (threadIdx.x & 63) == 0) s->found_inf = 1.f;     // every writer stores the same value
}
__global__ void amp_update_kernel(CmuAmpState* s, float growth, float backoff, int interval) {
    if (s->found_inf != 0.f) {
        s->scale *= backoff;
        s->growth_tracker = 0;
        s->skipped_steps += 1;
    } else {
        s->good_steps += 1;
        if (++s->growth_tracker == interval) {
            s->scale *= growth;
            s->growth_tracker = 0;
        }
    }
    s->found_inf = 0.f;
}
extern "C" int cmu_amp_state_bytes(void) { return (int)sizeof(CmuAmpState); }
extern "C" int cmu_amp_init(void* state, float init_scale, void* stream) {
    CMU_CHECK_ARG(state && init_scale > 0.f, "cmu_amp_init: bad args");
    hipLaunchKernelGGL(amp_init_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (CmuAmpState*)state, init_scale);
    CMU_CHECK_LAUNCH("cmu_amp_init");
    return CMU_OK;
}
extern "C" int cmu_amp_check_finite(const float* g, int64_t n, void* state, void* stream) {
    CMU_CHECK_ARG(g && state && n > 0 && cmu_aligned16(g), "cmu_amp_check_finite: bad args / alignment");
    const int64_t nb = cmu_div_up64(n >> 2, 256);
    const int grid = (int)(nb < 8192 ? (nb < 1 ? 1 : nb) : 8192);
    hipLaunchKernelGGL(amp_check_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, (CmuAmpState*)state);
    CMU_CHECK_LAUNCH("cmu_amp_check_finite");
    return CMU_OK;
}
extern "C" int cmu_amp_update(void* state, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    CMU_CHECK_ARG(state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1, "cmu_amp_update: bad args");
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (CmuAmpState*)state, growth_factor, backoff_factor,
                       growth_interval);
    CMU_CHECK_LAUNCH("cmu_amp_update");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// Global average pool of the activated latent (moco_data_module.py:65: x.mean([2,3])) and its backward
// ---------------------------------------------------------------------------------------------
template <class TR>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const typename TR::elem_t* __restrict__ y, int64_t ldy,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ out, int HW, int C) {
    // grid (ceil(C/64), B); thread = (channel lane, pixel part); fixed-order combine of the 4 parts
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6, b = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
        for (int p = part; p < HW; p += 4) {
            float v = fmaf(TR::to_float(y[((int64_t)b * HW + p) * ldy + c]), sc, sh);
            if (scale) v = fmaxf(v, 0.f);
            s += v;
        }
    }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < C) {
        const int l = threadIdx.x;
        out[(int64_t)b * C + c] = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) / (float)HW;
    }
}
// 16-byte form (C a multiple of the chunk): thread = (8-channel chunk, one of 16 pixel parts), four loads in flight per thread;
// the 2-byte form above walks 256 dependent loads per thread at the bottleneck of the 512^2 UNet (0.11 ms per call for 67 MB)
template <class TR>
__global__ __launch_bounds__(256) void gap_fwd16_kernel(const unsigned char* __restrict__ y, int64_t ldy, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ out, int HW, int C) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    __shared__ float red[16][16][EPC];
    const int lc = threadIdx.x & 15, part = threadIdx.x >> 4, b = blockIdx.y;
    const int chunk = blockIdx.x * 16 + lc, nchunk = C / EPC;
    float s[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        s[e] = 0.f;
        sc[e] = (scale && chunk < nchunk) ? scale[chunk * EPC + e] : 1.f;
        sh[e] = (scale && chunk < nchunk) ? shift[chunk * EPC + e] : 0.f;
    }
    if (chunk < nchunk) {
        const unsigned char* base = y + ((int64_t)b * HW * ldy + chunk * EPC) * ES;
        for (int p0 = part; p0 < HW; p0 += 64) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + 16 * u;
                v[u] = p < HW ? ld_global16(base + (int64_t)p * ldy * ES) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + 16 * u >= HW) continue;
                float f[EPC];
                TR::unpack(v[u], f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float t = fmaf(f[e], sc[e], sh[e]);
                    if (scale) t = fmaxf(t, 0.f);
                    s[e] += t;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[part][lc][e] = s[e];
    __syncthreads();
    if (threadIdx.x < 16 * EPC) {
        const int c_l = threadIdx.x / EPC, e = threadIdx.x % EPC, ch = blockIdx.x * 16 + c_l;
        if (ch < nchunk) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) a += red[q][c_l][e];      // fixed order
            out[(int64_t)b * C + ch * EPC + e] = a / (float)HW;
        }
    }
}
template <class TR>
__global__ void gap_bwd_kernel(const float* __restrict__ dout, typename TR::elem_t* __restrict__ dA, int64_t ldd, int HW, int C,
                               int64_t total) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        const int64_t pix = o / C;
        const int64_t b = pix / HW;
        dA[pix * ldd + c] = TR::from_float(dout[b * C + c] / (float)HW);
    }
}
template <class TR>
__global__ void gap_bwd16_kernel(const float* __restrict__ dout, unsigned char* __restrict__ dA, int64_t ldd, int HW, int nchunk, int64_t total) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(o % nchunk);
        const int64_t pix = o / nchunk;
        const int64_t b = pix / HW;
        float f[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = dout[(b * nchunk + ch) * EPC + e] / (float)HW;   // (the element kernel's expression)
        st_global16(dA + (pix * ldd + ch * EPC) * ES, TR::pack(f));
    }
}
template <class TR>
static int gap_fwd_t(const void* y, int64_t ldy, const float* scale, const float* shift, float* out, int B, int HW, int C, hipStream_t st) {
    if (C % TR::EPC == 0 && ldy % TR::EPC == 0 && cmu_aligned16(y)) {
        hipLaunchKernelGGL((gap_fwd16_kernel<TR>), dim3(cmu_div_up(C / TR::EPC, 16), B), dim3(256), 0, st, (const unsigned char*)y, ldy, scale,
                           shift, out, HW, C);
        CMU_CHECK_LAUNCH("cmu_gap_fwd");
        return CMU_OK;
    }
    hipLaunchKernelGGL((gap_fwd_kernel<TR>), dim3(cmu_div_up(C, 64), B), dim3(256), 0, st, (const typename TR::elem_t*)y, ldy, scale, shift,
                       out, HW, C);
    CMU_CHECK_LAUNCH("cmu_gap_fwd");
    return CMU_OK;
}
template <class TR>
static int gap_bwd_t(const float* dout, void* dA, int64_t ldd, int B, int HW, int C, hipStream_t st) {
    if (C % TR::EPC == 0 && ldd % TR::EPC == 0 && cmu_aligned16(dA)) {   // 16-byte stores
        const int64_t total = (int64_t)B * HW * (C / TR::EPC);
        const int grid = (int)(cmu_div_up64(total, 256) < 4096 ? cmu_div_up64(total, 256) : 4096);
        hipLaunchKernelGGL((gap_bwd16_kernel<TR>), dim3(grid), dim3(256), 0, st, dout, (unsigned char*)dA, ldd, HW, C / TR::EPC, total);
        CMU_CHECK_LAUNCH("cmu_gap_bwd");
        return CMU_OK;
    }
    const int64_t total = (int64_t)B * HW * C;
    const int grid = (int)(cmu_div_up64(total, 256) < 4096 ? cmu_div_up64(total, 256) : 4096);
    hipLaunchKernelGGL((gap_bwd_kernel<TR>), dim3(grid), dim3(256), 0, st, dout, (typename TR::elem_t*)dA, ldd, HW, C, total);
    CMU_CHECK_LAUNCH("cmu_gap_bwd");
    return CMU_OK;
}
extern "C" int cmu_gap_fwd(const void* y, int64_t ldy, const float* in_scale, const float* in_shift, float* out, int B, int H, int W,
                           int C, int dt, void* stream) {
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0 && y && out && B > 0 && H > 0 && W > 0 && C > 0 && ldy >= C, "cmu_gap_fwd: bad args");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_gap_fwd: scale/shift must both be set");
    CMU_DISPATCH_DT(dt, gap_fwd_t, y, ldy, in_scale, in_shift, out, B, H * W, C, (hipStream_t)stream);
}
extern "C" int cmu_gap_bwd(const float* dout, void* dA, int64_t ldd, int B, int H, int W, int C, int dt, void* stream) {
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0 && dout && dA && B > 0 && H > 0 && W > 0 && C > 0 && ldd >= C, "cmu_gap_bwd: bad args");
    CMU_DISPATCH_DT(dt, gap_bwd_t, dout, dA, ldd, B, H * W, C, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// soft-clDice (Finetuning/metrics.py:401-492): soft skeletonisation by iterated min/max pooling on fp32
// (planes, H, W) stacks.  With img_{j+1} = erode(img_j) the reference's soft_open(img_j) is dilate(img_{j+1}), so one erode
// and one fused dilate+update per level (it runs two erodes); pooling pads with the identity of min / max.
//   erode:  out = min( min over the 3x1 column window, min over the 1x3 row window )      (metrics.py:456-459)
//   update: delta = relu(img - dilate3x3(next)); skel = first ? delta : skel + relu(delta - skel*delta)   (:476-486)
// ---------------------------------------------------------------------------------------------
__global__ void soft_erode_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int64_t total) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(o % W), y = (int)((o / W) % H);
        const float c = in[o];
        float m = c;
        if (y > 0) m = fminf(m, in[o - W]);
        if (y + 1 < H) m = fminf(m, in[o + W]);
        if (x > 0) m = fminf(m, in[o - 1]);
        if (x + 1 < W) m = fminf(m, in[o + 1]);
        out[o] = m;
    }
}
__global__ void soft_skel_update_kernel(const float* __restrict__ img, const float* __restrict__ next, float* __restrict__ skel, int H,
                                        int W, int64_t total, int first) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(o % W), y = (int)((o / W) % H);
        float d = next[o];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) d = fmaxf(d, next[o + (int64_t)dy * W + dx]);
            }
        const float delta = fmaxf(img[o] - d, 0.f);
        skel[o] = first ? delta : skel[o] + fmaxf(delta - skel[o] * delta, 0.f);
    }
}
// four sums for clDice: sum(skel_pred*y_true), sum(skel_pred), sum(skel_true*y_pred), sum(skel_true) -> out[4] (fp64 combine)
__global__ __launch_bounds__(256) void cldice_sums_kernel(const float* __restrict__ sp, const float* __restrict__ yt,
                                                         const float* __restrict__ stt, const float* __restrict__ yp, int64_t n,
                                                         float* __restrict__ part) {
    __shared__ float red[4][4];
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        a[0] = fmaf(sp[i], yt[i], a[0]);
        a[1] += sp[i];
        a[2] = fmaf(stt[i], yp[i], a[2]);
        a[3] += stt[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = wave_sum(a[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) part[(int64_t)blockIdx.x * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ void cldice_final_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= 4) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += (double)part[(int64_t)b * 4 + k];
    out[k] = (float)s;
}
constexpr int CLD_BLOCKS = 512;
extern "C" int64_t cmu_soft_skeleton_ws_bytes(int64_t n) { return 2 * n * (int64_t)sizeof(float); }
extern "C" int cmu_soft_skeleton(const float* img, float* skel, int planes, int H, int W, int num_iter, void* ws, void* stream) {
    CMU_CHECK_ARG(img && skel && ws && planes > 0 && H > 0 && W > 0 && num_iter >= 0, "cmu_soft_skeleton: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)planes * H * W;
    const int grid = (int)(cmu_div_up64(n, 256) < 4096 ? cmu_div_up64(n, 256) : 4096);
    float* a = (float*)ws;
    float* b = a + n;
    // level 0: cur = img, nxt = erode(img)
    const float* cur = img;
    float* nxt = a;
    hipLaunchKernelGGL(soft_erode_kernel, dim3(grid), dim3(256), 0, st, cur, nxt, H, W, n);
    CMU_CHECK_LAUNCH("cmu_soft_skeleton(erode)");
    hipLaunchKernelGGL(soft_skel_update_kernel, dim3(grid), dim3(256), 0, st, cur, (const float*)nxt, skel, H, W, n, 1);
    CMU_CHECK_LAUNCH("cmu_soft_skeleton(update)");
    for (int j = 0; j < num_iter; ++j) {
        cur = nxt;                       // img_{j+1}
        nxt = (cur == a) ? b : a;        // img_{j+2} = erode(img_{j+1})
        hipLaunchKernelGGL(soft_erode_kernel, dim3(grid), dim3(256), 0, st, cur, nxt, H, W, n);
        CMU_CHECK_LAUNCH("cmu_soft_skeleton(erode)");
        hipLaunchKernelGGL(soft_skel_update_kernel, dim3(grid), dim3(256), 0, st, cur, (const float*)nxt, skel, H, W, n, 0);
        CMU_CHECK_LAUNCH("cmu_soft_skeleton(update)");
    }
    return CMU_OK;
}
extern "C" int64_t cmu_cldice_sums_ws_bytes(void) { return (int64_t)CLD_BLOCKS * 4 * (int64_t)sizeof(float); }
extern "C" int cmu_cldice_sums(const float* skel_pred, const float* y_true, const float* skel_true, const float* y_pred, int64_t n,
                               float* out4, void* ws, void* stream) {
    CMU_CHECK_ARG(skel_pred && y_true && skel_true && y_pred && out4 && ws && n > 0, "cmu_cldice_sums: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(cmu_div_up64(n, 1024) < CLD_BLOCKS ? cmu_div_up64(n, 1024) : CLD_BLOCKS);
    hipLaunchKernelGGL(cldice_sums_kernel, dim3(grid), dim3(256), 0, st, skel_pred, y_true, skel_true, y_pred, n, (float*)ws);
    CMU_CHECK_LAUNCH("cmu_cldice_sums");
    hipLaunchKernelGGL(cldice_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, grid, out4);
    CMU_CHECK_LAUNCH("cmu_cldice_sums(final)");
    return CMU_OK;
}
// (softmax(logits, dim=1)[:, 1] > threshold) as fp32 for 2-class logits (B,2,H,W): the binarised foreground the reference's
// metrics take after Activation('softmax') + _threshold + _take_channels(ignore_channels=[0]) (metrics.py:84-133)
__global__ void softmax2_threshold_kernel(const float* __restrict__ logits, float thr, float* __restrict__ out, int64_t HW, int64_t total) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = o / HW, r = o % HW;
        const float l0 = logits[(b * 2 + 0) * HW + r], l1 = logits[(b * 2 + 1) * HW + r];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        out[o] = (e1 / (e0 + e1) > thr) ? 1.f : 0.f;
    }
}
extern "C" int cmu_softmax2_threshold(const float* logits, float threshold, float* out, int B, int H, int W, void* stream) {
    CMU_CHECK_ARG(logits && out && B > 0 && H > 0 && W > 0, "cmu_softmax2_threshold: bad args");
    const int64_t HW = (int64_t)H * W, total = HW * B;
    const int grid = (int)(cmu_div_up64(total, 256) < 4096 ? cmu_div_up64(total, 256) : 4096);
    hipLaunchKernelGGL(softmax2_threshold_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, threshold, out, HW, total);
    CMU_CHECK_LAUNCH("cmu_softmax2_threshold");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// Round 5: the pieces of the NON-fused MoCo API (moco2_module.py:224-285, 311-329: ``forward`` / ``_compute_l_s`` / ``validation_step``) that
// ran on ATen ops until round 4 (F.normalize, torch.cat, F.cross_entropy, topk): row normalisation backward, the logits row
// [q.k | q @ queue] / T assembled in place, its backward split, the row-wise cross entropy (+ the rank of the target for precision@k).
// Not the training hot path (the fused step is cmu_moco_infonce_enqueue): simple one-block-per-row kernels, fixed-order sums.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int D) {
    __shared__ float red[4];
    const float* r = x + (int64_t)blockIdx.x * D;
    const float* g = dy + (int64_t)blockIdx.x * D;
    float s = 0.f, t = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { s = fmaf(r[d], r[d], s); t = fmaf(r[d], g[d], t); }
    const float n2 = block_sum(s, red);
    __syncthreads();
    const float xg = block_sum(t, red);
    const float n = sqrtf(n2);
    // y = x / max(n, eps): dx = dy / n - x (x . dy) / n^3 above the clamp, dy / eps below it (F.normalize's own backward)
    if (n > 1e-12f) {
        const float inv = 1.f / n, c = xg * inv * inv * inv;
        for (int d = threadIdx.x; d < D; d += 256) dx[(int64_t)blockIdx.x * D + d] = fmaf(g[d], inv, -r[d] * c);
    } else {
        for (int d = threadIdx.x; d < D; d += 256) dx[(int64_t)blockIdx.x * D + d] = g[d] * 1e12f;
    }
}
extern "C" int cmu_l2_normalize_rows_bwd(const float* x, const float* dy, float* dx, int B, int D, void* stream) {
    CMU_CHECK_ARG(x && dy && dx && B > 0 && D > 0, "cmu_l2_normalize_rows_bwd: bad args");
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, dy, dx, D);
    CMU_CHECK_LAUNCH("cmu_l2_normalize_rows_bwd");
    return CMU_OK;
}

// logits[b] = [ q[b].k[b] | lneg[b][0..K) ] * inv_t   (torch.cat([l_pos, l_neg], 1) / T of moco2_module.py:264-267)
__global__ __launch_bounds__(256) void moco_logits_assemble_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ lneg,
                                                                  float* __restrict__ logits, int D, int K, float inv_t) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s = fmaf(q[(int64_t)b * D + d], k[(int64_t)b * D + d], s);
    const float pos = block_sum(s, red);
    float* out = logits + (int64_t)b * (K + 1);
    if (threadIdx.x == 0) out[0] = pos * inv_t;
    for (int j = threadIdx.x; j < K; j += 256) out[1 + j] = lneg[(int64_t)b * K + j] * inv_t;
}
extern "C" int cmu_moco_logits_assemble(const float* q, const float* k, const float* lneg, float* logits, int B, int D, int K, float inv_t, void* stream) {
    CMU_CHECK_ARG(q && k && lneg && logits && B > 0 && D > 0 && K > 0, "cmu_moco_logits_assemble: bad args");
    hipLaunchKernelGGL(moco_logits_assemble_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, q, k, lneg, logits, D, K, inv_t);
    CMU_CHECK_LAUNCH("cmu_moco_logits_assemble");
    return CMU_OK;
}
// backward, first half: dlneg[b][j] = dlogits[b][1 + j] * inv_t (the operand of dq_neg = dlneg @ queue^T on the skinny kernel)
__global__ __launch_bounds__(256) void moco_logits_split_kernel(const float* __restrict__ dlogits, float* __restrict__ dlneg, int K, float inv_t) {
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < K; j += 256) dlneg[(int64_t)b * K + j] = dlogits[(int64_t)b * (K + 1) + 1 + j] * inv_t;
}
// second half: dq[b] += dlogits[b][0] * inv_t * k[b]
__global__ __launch_bounds__(256) void moco_logits_addpos_kernel(const float* __restrict__ dlogits, const float* __restrict__ k, float* __restrict__ dq, int D, int K,
                                                                float inv_t) {
    const int b = blockIdx.x;
    const float c = dlogits[(int64_t)b * (K + 1)] * inv_t;
    for (int d = threadIdx.x; d < D; d += 256) dq[(int64_t)b * D + d] = fmaf(c, k[(int64_t)b * D + d], dq[(int64_t)b * D + d]);
}
extern "C" int cmu_moco_logits_split(const float* dlogits, float* dlneg, int B, int K, float inv_t, void* stream) {
    CMU_CHECK_ARG(dlogits && dlneg && B > 0 && K > 0, "cmu_moco_logits_split: bad args");
    hipLaunchKernelGGL(moco_logits_split_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dlogits, dlneg, K, inv_t);
    CMU_CHECK_LAUNCH("cmu_moco_logits_split");
    return CMU_OK;
}
extern "C" int cmu_moco_logits_addpos(const float* dlogits, const float* k, float* dq, int B, int D, int K, float inv_t, void* stream) {
    CMU_CHECK_ARG(dlogits && k && dq && B > 0 && D > 0 && K > 0, "cmu_moco_logits_addpos: bad args");
    hipLaunchKernelGGL(moco_logits_addpos_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dlogits, k, dq, D, K, inv_t);
    CMU_CHECK_LAUNCH("cmu_moco_logits_addpos");
    return CMU_OK;
}

// F.cross_entropy(logits, target) with mean reduction over the rows (moco2_module.py:283, 324): per row logsumexp - x[target]; dlogits = (softmax -
// onehot) / B (the gradient of the MEAN; the caller multiplies by the incoming scalar with cmu_scale_by_device_scalar); rank[b] = number of
// logits strictly above the target's (precision@k: hit iff rank < k).  One block per row, two passes over the row; the mean in a
// second one-block launch (fixed order).
__global__ __launch_bounds__(256) void row_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, float* __restrict__ row_loss,
                                                    float* __restrict__ dlogits, int* __restrict__ rank, int N, float inv_b) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (int64_t)b * N;
    // a target outside [0, N) (F.cross_entropy asserts on it): no read outside the row -- the row's loss, and with it the mean, becomes NaN
    const int64_t t64 = target[b];
    const bool t_ok = t64 >= 0 && t64 < (int64_t)N;
    const int t = t_ok ? (int)t64 : 0;
    float m = -__builtin_inff();
    for (int j = tid; j < N; j += 256) m = fmaxf(m, x[j]);
    m = block_max(m, red);
    __syncthreads();
    const float xt = t_ok ? x[t] : __builtin_nanf("");
    float s = 0.f, above = 0.f;
    for (int j = tid; j < N; j += 256) {
        s += __expf(x[j] - m);
        above += x[j] > xt ? 1.f : 0.f;
    }
    s = block_sum(s, red);
    __syncthreads();
    above = block_sum(above, red);
    const float lse = m + __logf(s);
    if (tid == 0) {
        row_loss[b] = lse - xt;
        if (rank != nullptr) rank[b] = (int)above;
    }
    if (dlogits != nullptr) {
        float* d = dlogits + (int64_t)b * N;
        for (int j = tid; j < N; j += 256) d[j] = (__expf(x[j] - lse) - (j == t ? 1.f : 0.f)) * inv_b;
    }
}
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ v, float* __restrict__ out, int B) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += v[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s / (float)B;
}
extern "C" int cmu_row_cross_entropy(const float* logits, const int64_t* target, float* loss, float* row_loss, float* dlogits, int* rank, int B, int N,
                                     void* stream) {
    CMU_CHECK_ARG(logits && target && loss && row_loss && B > 0 && N > 0, "cmu_row_cross_entropy: bad args");
    hipLaunchKernelGGL(row_ce_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, target, row_loss, dlogits, rank, N, 1.f / (float)B);
    hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_loss, loss, B);
    CMU_CHECK_LAUNCH("cmu_row_cross_entropy");
    return CMU_OK;
}
// v[i] *= s[0] (s on the device: the incoming gradient of a scalar loss, never read back)
__global__ void scale_by_device_scalar_kernel(float* __restrict__ v, const float* __restrict__ s, int64_t n) {
    const float c = s[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] *= c;
}
extern "C" int cmu_scale_by_device_scalar(float* v, const float* s, int64_t n, void* stream) {
    CMU_CHECK_ARG(v && s && n > 0, "cmu_scale_by_device_scalar: bad args");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, s, n);
    CMU_CHECK_LAUNCH("cmu_scale_by_device_scalar");
    return CMU_OK;
}
