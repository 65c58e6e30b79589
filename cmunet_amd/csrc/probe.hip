// probe.hip -- what MFMA rate does THIS chip sustain?  (measurement support for bench.py's roofline block)
//
// The conv kernels of this library are priced against the 2.5 PFLOP/s dense 16-bit peak (2.4 GHz).  Under a matrix-pipe
// load the MI355X does not hold 2.4 GHz: its power management lowers the shader clock, and by how much depends on the
// OPERAND DATA (bit toggling in the multipliers).  Measured with this probe (tools/mfma_clock.hip is the stand-alone
// form; profiles/r02_mfma_clock.txt): back-to-back 32x32x16 f16 MFMAs from registers -- no LDS, no memory, every SIMD
// busy -- run at 2.45 PFLOP/s / 2.38 GHz on all-zero operands, 2.0 PFLOP/s / 1.97 GHz on ReLU'd N(0,1) operands (half
// zeros) and 1.63 PFLOP/s / 1.61 GHz on dense N(0,1) operands.  That last figure is the ceiling a kernel fed with dense
// random data can reach on this chip whatever its schedule; bench.py reports it beside the nominal peak.
//
// The same MFMAs fed from LDS at the persistent conv kernel's fragment ratio (lds_fed = 1) sustain 1.41 PFLOP/s at 1.44 GHz
// on dense operands, 1.70 on ReLU'd ones: the ceiling of an LDS-tiled kernel on this chip.  (The 16x16x32 shape draws less:
// 1.83 / 1.58 PFLOP/s from registers / LDS on the same dense data -- tools/mfma_clock.hip.)
//
// cmu_mfma_sustained_rate runs that loop for about `iters` x 8 MFMAs per wave on the given stream and returns the rate
// and the shader clock (s_memtime / s_memrealtime of one wave).  Operands are generated in the kernel (a hash of the lane
// id; sum-of-uniforms ~ N(0,1)), so the probe needs no memory but a 64-byte scratch for its two counters.
#include "common.h"

typedef _Float16 pr_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pr_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pr_f32x16 __attribute__((ext_vector_type(16)));

__device__ static inline unsigned pr_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// ~N(0,1): sum of four uniform bytes, centred and scaled (variance 4 * (256^2 - 1) / 12 -> / 147.8)
__device__ static inline float pr_normal(unsigned h) {
    const float s = (float)(h & 255u) + (float)((h >> 8) & 255u) + (float)((h >> 16) & 255u) + (float)(h >> 24);
    return (s - 510.f) * (1.f / 147.8f);
}

// PATTERN 0: dense ~N(0,1) operands; 1: ReLU'd (half zeros); 2: all zeros.   BF = bf16 instead of f16
template <bool BF, int PATTERN>
__global__ __launch_bounds__(512) void mfma_probe_kernel(int iters, unsigned long long* out) {
    const unsigned tid = threadIdx.x + blockIdx.x * 512u;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f0 = pr_normal(pr_hash(tid * 64u + i * 8u + q * 2u)), f1 = pr_normal(pr_hash(tid * 64u + i * 8u + q * 2u + 1u));
            if (PATTERN == 1) { f0 = fmaxf(f0, 0.f); f1 = fmaxf(f1, 0.f); }
            if (PATTERN == 2) { f0 = 0.f; f1 = 0.f; }
            if (BF) {
                w[q] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u);
            } else {
                const _Float16 h0 = (_Float16)f0, h1 = (_Float16)f1;
                w[q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            }
        }
        (i < 4 ? a[i & 3] : b[i & 3]) = u32x4{w[0], w[1], w[2], w[3]};
    }
    pr_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (BF)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pr_bf16x8, a[i & 3]), __builtin_bit_cast(pr_bf16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pr_f16x8, a[i & 3]), __builtin_bit_cast(pr_f16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[7] = 1ull;            // keeps the accumulators alive
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

// The same MFMAs FED FROM LDS at the persistent conv kernel's ratio (wave tile 128 x 64: per step 4 A + 2 B fragment reads of
// 1 KB for 8 MFMAs), software-pipelined through a register double buffer: what a perfectly scheduled LDS-tiled kernel could
// sustain.  64 KB of LDS filled with the pattern; `iters` steps of 8 MFMAs per wave.
template <bool BF, int PATTERN>
__global__ __launch_bounds__(512) void mfma_probe_lds_kernel(int iters, unsigned long long* out) {
    __shared__ u32x4 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f0 = pr_normal(pr_hash((blockIdx.x * 4096u + i) * 8u + q * 2u)), f1 = pr_normal(pr_hash((blockIdx.x * 4096u + i) * 8u + q * 2u + 1u));
            if (PATTERN == 1) { f0 = fmaxf(f0, 0.f); f1 = fmaxf(f1, 0.f); }
            if (PATTERN == 2) { f0 = 0.f; f1 = 0.f; }
            if (BF) {
                w[q] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u);
            } else {
                const _Float16 h0 = (_Float16)f0, h1 = (_Float16)f1;
                w[q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            }
        }
        lds[i] = u32x4{w[0], w[1], w[2], w[3]};
    }
    __syncthreads();
    u32x4 fa[2][4], fb[2][2];
    pr_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    auto rd = [&](int set, int it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = lds[(tid + 64 * i + 331 * it) & 4095];
#pragma unroll
        for (int i = 0; i < 2; ++i) fb[set][i] = lds[(tid + 64 * (i + 4) + 173 * it) & 4095];
    };
    auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (BF)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pr_bf16x8, fa[set][i]), __builtin_bit_cast(pr_bf16x8, fb[set][j]), acc[i * 2 + j], 0, 0, 0);
                else
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pr_f16x8, fa[set][i]), __builtin_bit_cast(pr_f16x8, fb[set][j]), acc[i * 2 + j], 0, 0, 0);
            }
    };
    rd(0, 0);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        rd(1, it + 1);
        mm(0);
        rd(0, it + 2);
        mm(1);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[7] = 1ull;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

extern "C" int cmu_mfma_sustained_rate(int dt, int pattern, int lds_fed, int iters, void* scratch64, double* tflops, double* clock_mhz, void* stream) {
    CMU_CHECK_ARG(dt == CMU_F16 || dt == CMU_BF16, "cmu_mfma_sustained_rate: dt must be f16 or bf16");
    CMU_CHECK_ARG(pattern >= 0 && pattern <= 2 && iters > 0 && iters <= (1 << 24) && scratch64 && tflops && clock_mhz,
                  "cmu_mfma_sustained_rate: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cmu_current_device()) != hipSuccess || cus <= 0) cus = 256;
    unsigned long long* out = reinterpret_cast<unsigned long long*>(scratch64);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        cmu_set_error("cmu_mfma_sustained_rate: hipEventCreate failed");
        return CMU_ERR_LAUNCH;
    }
    auto launch = [&](int n) {
        const bool bf = dt == CMU_BF16;
#define PR_GO(BF_, P_)                                                                                                       \
    do {                                                                                                                    \
        if (lds_fed) hipLaunchKernelGGL((mfma_probe_lds_kernel<BF_, P_>), dim3((unsigned)cus), dim3(512), 0, st, (n + 1) & ~1, out); \
        else hipLaunchKernelGGL((mfma_probe_kernel<BF_, P_>), dim3((unsigned)cus), dim3(512), 0, st, n, out);               \
    } while (0)
        if (pattern == 0) { if (bf) PR_GO(true, 0); else PR_GO(false, 0); }
        else if (pattern == 1) { if (bf) PR_GO(true, 1); else PR_GO(false, 1); }
        else { if (bf) PR_GO(true, 2); else PR_GO(false, 2); }
#undef PR_GO
    };
    launch(iters / 16 + 1);                      // the clock settles under the load before the timed launch
    hipError_t e = hipEventRecord(e0, st);
    launch(iters);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess || ms <= 0.f) {
        cmu_set_error("cmu_mfma_sustained_rate: %s", hipGetErrorString(e));
        return CMU_ERR_LAUNCH;
    }
    const double steps = lds_fed ? (double)((iters + 1) & ~1) : (double)iters;
    *tflops = (double)cus * 8.0 * steps * 8.0 * 32768.0 / ((double)ms * 1e-3) * 1e-12;
    *clock_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// cmu_probe_stream_reduce -- a stand-in for a collective's reduction kernel on ONE GPU (round 5, tools/exchange_probe.py): `grid`
// workgroups of 256 threads sweep out[i] = a[i] + b[i] over n floats `passes` times (RCCL's ring all-reduce of a gradient bucket runs a
// few dozen such workgroups for bytes / link-rate seconds).  What the probe answers: launched on a side stream where a bucket is announced
// inside the backward pass, does it make progress beside the persistent conv kernels (one 512-thread workgroup with ~160 KB of LDS and
// 2 x 248 registers per SIMD on every CU), and what does the backward pass lose?  The kernel keeps 64 floats per lane in registers on purpose
// (a collective kernel's register footprint: it cannot slip into the 16 registers per SIMD lane the conv workgroups leave free).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stream_reduce_probe_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n4,
                                                                 int passes) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (int pass = 0; pass < passes; ++pass) {
        for (int64_t i0 = (int64_t)blockIdx.x * 256 * 16 + threadIdx.x; i0 < n4; i0 += (int64_t)gridDim.x * 256 * 16) {
            f32x4 va[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) va[u] = i0 + u * 256 < n4 ? a4[i0 + u * 256] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (i0 + u * 256 < n4) o4[i0 + u * 256] = va[u] + b4[i0 + u * 256];
        }
    }
}
extern "C" int cmu_probe_stream_reduce(const float* a, const float* b, float* out, int64_t n, int grid, int passes, void* stream) {
    CMU_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0 && grid > 0 && grid <= 1024 && passes > 0, "cmu_probe_stream_reduce: bad args (n % 4 == 0, grid <= 1024)");
    hipLaunchKernelGGL(stream_reduce_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4, passes);
    CMU_CHECK_LAUNCH("cmu_probe_stream_reduce");
    return CMU_OK;
}
