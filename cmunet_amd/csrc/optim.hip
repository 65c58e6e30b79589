// optim.hip -- fused optimiser steps over the flat fp32 parameter arena beyond Adam/AdamW (heads.hip): SGD with momentum
// (torch.optim.SGD as MoCo uses it, Pretraining/MoCo/moco2_module.py:339-344) and LAMB (Pretraining/Spark/utils/lamb.py:
// 67-159: global gradient-norm clip, Adam moments, per-tensor trust ratio).  HBM-bound single passes; every reduction goes
// through a slab and a fixed-order second kernel (bitwise reproducible, no float atomics, no host synchronisation).
#include "common.h"
#include <math.h>

// ---------------------------------------------------------------------------------------------
// SGD (+ momentum, dampening, Nesterov, L2 weight decay with an optional per-element mask)
// ---------------------------------------------------------------------------------------------
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, const uint8_t* __restrict__ wd_mask,
                           int64_t n, float lr, float mom, float damp, float wd, int nesterov, int first, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        float gi = g[i] * gscale;
        const float w = (wd_mask == nullptr || wd_mask[i]) ? wd : 0.f;
        gi = fmaf(w, pi, gi);
        float d = gi;
        if (mom != 0.f) {
            const float b = first ? gi : fmaf(mom, buf[i], (1.f - damp) * gi);   // first step: buf = clone(grad)
            buf[i] = b;
            d = nesterov ? fmaf(mom, b, gi) : b;
        }
        p[i] = fmaf(-lr, d, pi);
    }
}
extern "C" int cmu_sgd_step(float* p, const float* g, float* buf, const uint8_t* wd_mask, int64_t n, float lr, float momentum,
                            float dampening, float weight_decay, int nesterov, int64_t step, float grad_scale, void* stream) {
    CMU_CHECK_ARG(p && g && n > 0 && step >= 1 && (momentum == 0.f || buf), "cmu_sgd_step: bad args");
    const int64_t nb = cmu_div_up64(n, 256);
    const int grid = (int)(nb < 8192 ? nb : 8192);
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, buf, wd_mask, n, lr, momentum, dampening, weight_decay,
                       nesterov, step == 1 ? 1 : 0, grad_scale);
    CMU_CHECK_LAUNCH("cmu_sgd_step");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// LAMB.  The arena is cut into blocks of <= LAMB_BLK elements that never straddle a parameter tensor (table built by the
// host once): blk_start[b], blk_count[b], blk_tensor[b]; tensor t owns blocks [t_blk0[t], t_blk0[t+1]).
//   1. sum g^2 per block -> global gradient norm -> clip factor (device scalar)
//   2. moments + update u (kept in a scratch arena) + per-block sum p^2, sum u^2
//   3. per-tensor trust ratio from its blocks' partial sums (fixed order)
//   4. p -= lr * ratio[t] * u
// ---------------------------------------------------------------------------------------------
constexpr int LAMB_BLK = 4096;

__device__ static inline float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void lamb_gradsq_kernel(const float* __restrict__ g, const int64_t* __restrict__ blk_start,
                                                         const int* __restrict__ blk_count, float gscale, float* __restrict__ part) {
    __shared__ float red[4];
    const int64_t s = blk_start[blockIdx.x];
    const int n = blk_count[blockIdx.x];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = g[s + i] * gscale;
        a = fmaf(gi, gi, a);
    }
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void lamb_gnorm_kernel(const float* __restrict__ part, int nblocks, float max_norm, float* __restrict__ scal) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) a += (double)part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float gn = (float)sqrt(red[0]);
        scal[0] = gn;
        scal[1] = (max_norm > 0.f && gn > max_norm) ? 1.f / (gn / max_norm) : 1.f;   // lamb.py:93-96
    }
}
__global__ __launch_bounds__(256) void lamb_update_kernel(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, float* __restrict__ u, const int64_t* __restrict__ blk_start,
                                                         const int* __restrict__ blk_count, const int* __restrict__ blk_tensor,
                                                         const float* __restrict__ t_wd, const float* __restrict__ scal, float gscale,
                                                         float b1, float b2, float b3, float eps, float bc1, float bc2_sqrt,
                                                         float* __restrict__ part2) {
    __shared__ float red[4];
    const int64_t s = blk_start[blockIdx.x];
    const int n = blk_count[blockIdx.x];
    const float wd = t_wd[blk_tensor[blockIdx.x]];
    const float gs = gscale * scal[1];
    float sp = 0.f, su = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int64_t k = s + i;
        const float pi = p[k], gi = g[k] * gs;
        const float mi = fmaf(b1, m[k], b3 * gi);                 // exp_avg.mul_(beta1).add_(grad, alpha=beta3)
        const float vi = fmaf(b2, v[k], (1.f - b2) * gi * gi);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
        m[k] = mi;
        v[k] = vi;
        float up = (mi / bc1) / (sqrtf(vi) / bc2_sqrt + eps);
        up = fmaf(wd, pi, up);
        u[k] = up;
        sp = fmaf(pi, pi, sp);
        su = fmaf(up, up, su);
    }
    sp = block_sum_256(sp, red);
    su = block_sum_256(su, red);
    if (threadIdx.x == 0) {
        part2[2 * (int64_t)blockIdx.x + 0] = sp;
        part2[2 * (int64_t)blockIdx.x + 1] = su;
    }
}
// one wave per tensor: the lanes stride over the tensor's blocks, fixed-order xor fold in double (one THREAD per tensor walked up to
// 2,304 partials through dependent loads: 0.29 ms per step on the SparK workload's 2 x 64-thread grid)
__global__ __launch_bounds__(64) void lamb_ratio_kernel(const float* __restrict__ part2, const int* __restrict__ t_blk0, const float* __restrict__ t_wd,
                                                       int ntensors, int always_adapt, int trust_clip, float* __restrict__ ratio) {
    const int t = blockIdx.x;
    if (t >= ntensors) return;
    double sp = 0.0, su = 0.0;
    const int b1 = t_blk0[t + 1];
    for (int b = t_blk0[t] + (int)threadIdx.x; b < b1; b += 64) {
        sp += (double)part2[2 * (int64_t)b + 0];
        su += (double)part2[2 * (int64_t)b + 1];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sp += __shfl_xor(sp, o, 64);
        su += __shfl_xor(su, o, 64);
    }
    if (threadIdx.x != 0) return;
    float r = 1.f;
    if (t_wd[t] != 0.f || always_adapt) {
        const float wn = (float)sqrt(sp), gn = (float)sqrt(su);
        r = (wn > 0.f && gn > 0.f) ? wn / gn : 1.f;
        if (trust_clip) r = fminf(r, 1.f);
    }
    ratio[t] = r;
}
__global__ __launch_bounds__(256) void lamb_apply_kernel(float* __restrict__ p, const float* __restrict__ u, const int64_t* __restrict__ blk_start,
                                                        const int* __restrict__ blk_count, const int* __restrict__ blk_tensor,
                                                        const float* __restrict__ ratio, float lr) {
    const int64_t s = blk_start[blockIdx.x];
    const int n = blk_count[blockIdx.x];
    const float a = -lr * ratio[blk_tensor[blockIdx.x]];
    for (int i = threadIdx.x; i < n; i += 256) p[s + i] = fmaf(a, u[s + i], p[s + i]);
}

extern "C" int cmu_lamb_block_elems(void) { return LAMB_BLK; }
// workspace: [nblocks] grad^2 partials | [2*nblocks] (p^2, u^2) partials | [ntensors] ratios | 2 scalars (grad norm, clip)
extern "C" int64_t cmu_lamb_ws_bytes(int nblocks, int ntensors) { return ((int64_t)3 * nblocks + ntensors + 4) * (int64_t)sizeof(float); }
extern "C" int cmu_lamb_step(float* p, const float* g, float* m, float* v, float* u, const int64_t* blk_start, const int* blk_count,
                             const int* blk_tensor, int nblocks, const int* t_blk0, const float* t_wd, int ntensors, float lr, float beta1,
                             float beta2, float eps, int bias_correction, int grad_averaging, float max_grad_norm, int trust_clip,
                             int always_adapt, int64_t step, float grad_scale, void* ws, void* stream) {
    CMU_CHECK_ARG(p && g && m && v && u && blk_start && blk_count && blk_tensor && t_blk0 && t_wd && ws && nblocks > 0 && ntensors > 0 &&
                      step >= 1,
                  "cmu_lamb_step: bad args");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    float* part2 = part + nblocks;
    float* ratio = part2 + 2 * (int64_t)nblocks;
    float* scal = ratio + ntensors;
    const double bc1 = bias_correction ? 1.0 - pow((double)beta1, (double)step) : 1.0;
    const double bc2 = bias_correction ? 1.0 - pow((double)beta2, (double)step) : 1.0;
    const float b3 = grad_averaging ? 1.f - beta1 : 1.f;
    hipLaunchKernelGGL(lamb_gradsq_kernel, dim3(nblocks), dim3(256), 0, st, g, blk_start, blk_count, grad_scale, part);
    CMU_CHECK_LAUNCH("cmu_lamb_step(grad norm)");
    hipLaunchKernelGGL(lamb_gnorm_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nblocks, max_grad_norm, scal);
    CMU_CHECK_LAUNCH("cmu_lamb_step(clip)");
    hipLaunchKernelGGL(lamb_update_kernel, dim3(nblocks), dim3(256), 0, st, (const float*)p, g, m, v, u, blk_start, blk_count, blk_tensor, t_wd,
                       (const float*)scal, grad_scale, beta1, beta2, b3, eps, (float)bc1, (float)sqrt(bc2), part2);
    CMU_CHECK_LAUNCH("cmu_lamb_step(update)");
    hipLaunchKernelGGL(lamb_ratio_kernel, dim3(ntensors), dim3(64), 0, st, (const float*)part2, t_blk0, t_wd, ntensors,
                       always_adapt, trust_clip, ratio);
    CMU_CHECK_LAUNCH("cmu_lamb_step(trust ratio)");
    hipLaunchKernelGGL(lamb_apply_kernel, dim3(nblocks), dim3(256), 0, st, p, (const float*)u, blk_start, blk_count, blk_tensor,
                       (const float*)ratio, lr);
    CMU_CHECK_LAUNCH("cmu_lamb_step(apply)");
    return CMU_OK;
}
