// tools/mfma_clock.hip -- diagnostic: what shader clock and MFMA rate does the chip SUSTAIN under a pure matrix-pipe load?
// The conv kernels run at 1.2-1.45 PFLOP/s with the clock at 1.2-1.45 GHz (tools/igemm3p_stamps.py): this program separates
// the power limit from the kernels' own stalls.  Every wave issues back-to-back independent MFMAs from registers (no LDS, no
// memory) for ~0.3 s per variant; wave 0 of a few workgroups reads s_memtime / s_memrealtime around its loop.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/_diag/mfma_clock tools/mfma_clock.hip
//   run:   tools/_diag/mfma_clock
// Variants: instruction shape (32x32x16 / 16x16x32), dtype (f16 / bf16), operand data (random / zeros / small-range),
// waves per SIMD (1 / 2), optional LDS fragment reads beside the MFMAs (6 x 1 KB per 8 MFMAs, the conv kernel's ratio).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: 32x32x16 f16   1: 16x16x32 f16   2: 32x32x16 bf16
template <int MODE, bool LDS>
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ src, int iters, unsigned long long* out, float* sink) {
    __shared__ u32x4 lds[LDS ? 2048 : 1];
    const int tid = threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(tid * 8 + i) & 4095];
        b[i] = src[(tid * 8 + 4 + i) & 4095];
    }
    if (LDS) {
        for (int i = tid; i < 2048; i += blockDim.x) lds[i] = src[i & 4095];
        __syncthreads();
    }
    f32x16 acc[8];
    f32x4 acc4[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f;
    }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            // six 1 KB fragment reads per eight MFMAs (the persistent conv kernel's ratio); rotating addresses
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a[i] = lds[(tid + 64 * i + 7 * it) & 2047];
                b[i] = lds[(tid + 64 * (i + 3) + 5 * it) & 2047];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 3]), __builtin_bit_cast(f16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
            else if (MODE == 2)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
            else {
                // two 16x16x32 = the FLOPs of one 32x32x16 ... no: 16*16*32*2 = 16,384 = half; issue two per slot
                acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i & 3]), __builtin_bit_cast(f16x8, b[(i >> 1) & 3]), acc4[i], 0, 0, 0);
                acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(i + 1) & 3]), __builtin_bit_cast(f16x8, b[(i >> 1) & 3]), acc4[i], 0, 0, 0);
            }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc4[i][e];
    }
    if (s == 123.456f) sink[0] = s;
    if (tid == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 8) {
        out[(blockIdx.x / 37) * 2 + 0] = c1 - c0;
        out[(blockIdx.x / 37) * 2 + 1] = r1 - r0;
    }
}


// MFMAs fed from LDS at the conv kernels' ratio, software-pipelined (register double buffer): wave tile 128 x 64 of fp32
// accumulators (128 registers) in both shapes.  SHAPE 0: 32x32x16 -- per step 4 A + 2 B fragment reads (1 KB each), 8 MFMAs;
// SHAPE 1: 16x16x32 -- per step 8 A + 4 B fragment reads, 32 MFMAs (twice the K: the same bytes per FLOP).
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_lds_loop(const u32x4* __restrict__ src, int iters, unsigned long long* out, float* sink) {
    __shared__ u32x4 lds[4096];   // 64 KB
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = src[i & 4095];
    __syncthreads();
    constexpr int NA = SHAPE == 0 ? 4 : 8, NBF = SHAPE == 0 ? 2 : 4;
    u32x4 fa[2][NA], fb[2][NBF];
    f32x16 acc[8];
    f32x4 acc4[32];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f;
    auto rd = [&](int set, int it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) fa[set][i] = lds[(tid + 64 * i + 331 * it) & 4095];
#pragma unroll
        for (int i = 0; i < NBF; ++i) fb[set][i] = lds[(tid + 64 * (i + NA) + 173 * it) & 4095];
    };
    auto mm = [&](int set) {
        if (SHAPE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][i]), __builtin_bit_cast(f16x8, fb[set][j]), acc[i * 2 + j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[set][i]), __builtin_bit_cast(f16x8, fb[set][j]), acc4[i * 4 + j], 0, 0, 0);
        }
    };
    rd(0, 0);
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        rd(1, it + 1);
        mm(0);
        rd(0, it + 2);
        mm(1);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc4[i][e];
    if (s == 123.456f) sink[0] = s;
    if (tid == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 8) {
        out[(blockIdx.x / 37) * 2 + 0] = c1 - c0;
        out[(blockIdx.x / 37) * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE>
static void run_lds(const char* name, const u32x4* src, int cus, unsigned long long* d_out, float* d_sink) {
    const int iters = SHAPE == 0 ? 600000 : 150000;          // the same FLOPs per wave in both shapes
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(d_out, 0, 16 * 8));
    hipLaunchKernelGGL((mfma_lds_loop<SHAPE>), dim3(cus), dim3(512), 0, 0, src, iters / 20, d_out, d_sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma_lds_loop<SHAPE>), dim3(cus), dim3(512), 0, 0, src, iters, d_out, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    double clk = 0; int n = 0;
    for (int i = 0; i < 8; ++i) if (h[2 * i + 1] > 0) { clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; ++n; }
    const double flop = (double)cus * 8.0 * (double)iters * (SHAPE == 0 ? 8.0 * 32768.0 : 32.0 * 16384.0);
    printf("%-60s %8.2f ms  %7.1f TFLOP/s  clock %5.0f MHz\n", name, ms, flop / ms * 1e-9, n ? clk / n : 0.0);
    fflush(stdout);
}

// ONE wave per SIMD with a 128 x 128 wave tile (16 accumulators of 32 x 32 = 256 registers, which a lone wave may hold: the
// unified file gives it 512): per step 4 A + 4 B fragment reads and 16 MFMAs -- 8 reads per 16 MFMAs against the conv kernel's 6
// per 8, a third less LDS traffic per FLOP.  What would such a kernel's ceiling be?
__global__ __launch_bounds__(256) void mfma_lds_wide_loop(const u32x4* __restrict__ src, int iters, unsigned long long* out, float* sink) {
    __shared__ u32x4 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = src[i & 4095];
    __syncthreads();
    u32x4 fa[2][4], fb[2][4];
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    auto rd = [&](int set, int it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = lds[(tid + 64 * i + 331 * it) & 4095];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[set][i] = lds[(tid + 64 * (i + 4) + 173 * it) & 4095];
    };
    auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][i]), __builtin_bit_cast(f16x8, fb[set][j]), acc[i * 4 + j], 0, 0, 0);
    };
    rd(0, 0);
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        rd(1, it + 1);
        mm(0);
        rd(0, it + 2);
        mm(1);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) sink[0] = s;
    if (tid == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 8) {
        out[(blockIdx.x / 37) * 2 + 0] = c1 - c0;
        out[(blockIdx.x / 37) * 2 + 1] = r1 - r0;
    }
}
static void run_lds_wide(const char* name, const u32x4* src, int cus, unsigned long long* d_out, float* d_sink) {
    const int iters = 600000;                                   // 16 MFMAs per step and wave, 4 waves per CU: the FLOPs of run_lds<0>
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(d_out, 0, 16 * 8));
    hipLaunchKernelGGL(mfma_lds_wide_loop, dim3(cus), dim3(256), 0, 0, src, iters / 20, d_out, d_sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_lds_wide_loop, dim3(cus), dim3(256), 0, 0, src, iters, d_out, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    double clk = 0; int n = 0;
    for (int i = 0; i < 8; ++i) if (h[2 * i + 1] > 0) { clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; ++n; }
    const double flop = (double)cus * 4.0 * (double)iters * 16.0 * 32768.0;
    printf("%-60s %8.2f ms  %7.1f TFLOP/s  clock %5.0f MHz\n", name, ms, flop / ms * 1e-9, n ? clk / n : 0.0);
    fflush(stdout);
}

// LDS-fed 32x32x16 loop with DIFFERENT data for the two operands (A = first MFMA source, rows; B = second, columns): is the power
// limit symmetric in the operands?
__global__ __launch_bounds__(512) void mfma_lds_ab_loop(const u32x4* __restrict__ srcA, const u32x4* __restrict__ srcB, int iters, unsigned long long* out, float* sink) {
    __shared__ u32x4 ldsA[2048], ldsB[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += blockDim.x) { ldsA[i] = srcA[i]; ldsB[i] = srcB[i]; }
    __syncthreads();
    u32x4 fa[2][4], fb[2][2];
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    auto rd = [&](int set, int it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = ldsA[(tid + 64 * i + 331 * it) & 2047];
#pragma unroll
        for (int i = 0; i < 2; ++i) fb[set][i] = ldsB[(tid + 64 * i + 173 * it) & 2047];
    };
    auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][i]), __builtin_bit_cast(f16x8, fb[set][j]), acc[i * 2 + j], 0, 0, 0);
    };
    rd(0, 0);
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        rd(1, it + 1);
        mm(0);
        rd(0, it + 2);
        mm(1);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) sink[0] = s;
    if (tid == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 8) {
        out[(blockIdx.x / 37) * 2 + 0] = c1 - c0;
        out[(blockIdx.x / 37) * 2 + 1] = r1 - r0;
    }
}
static void run_ab(const char* name, const u32x4* a, const u32x4* b, int cus, unsigned long long* d_out, float* d_sink) {
    const int iters = 600000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(d_out, 0, 16 * 8));
    hipLaunchKernelGGL(mfma_lds_ab_loop, dim3(cus), dim3(512), 0, 0, a, b, iters / 20, d_out, d_sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_lds_ab_loop, dim3(cus), dim3(512), 0, 0, a, b, iters, d_out, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    double clk = 0; int n = 0;
    for (int i = 0; i < 8; ++i) if (h[2 * i + 1] > 0) { clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; ++n; }
    const double flop = (double)cus * 8.0 * (double)iters * 8.0 * 32768.0;
    printf("%-60s %8.2f ms  %7.1f TFLOP/s  clock %5.0f MHz\n", name, ms, flop / ms * 1e-9, n ? clk / n : 0.0);
    fflush(stdout);
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

template <int MODE, bool LDS>
static void run(const char* name, const u32x4* src, int threads, int wg_per_cu, int cus, unsigned long long* d_out, float* d_sink) {
    const int iters = 1200000 / (threads / 256);
    const int grid = cus * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(d_out, 0, 16 * 8));
    hipLaunchKernelGGL((mfma_loop<MODE, LDS>), dim3(grid), dim3(threads), 0, 0, src, iters / 20, d_out, d_sink);   // warm-up
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma_loop<MODE, LDS>), dim3(grid), dim3(threads), 0, 0, src, iters, d_out, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    double clk = 0; int n = 0;
    for (int i = 0; i < 8; ++i) if (h[2 * i + 1] > 0) { clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; ++n; }
    const double flop = (double)grid * (threads / 64) * (double)iters * 8.0 * 32768.0;
    const double cyc_per_mfma = n ? (double)h[0] / ((double)iters * 8.0) * (threads / 256 >= 2 ? 0.5 : 1.0) : 0;
    printf("%-44s %4d thr x %d WG/CU: %8.2f ms  %7.1f TFLOP/s  clock %5.0f MHz  %.1f cycles per 32x32x16-equivalent per SIMD\n", name, threads, wg_per_cu, ms,
           flop / ms * 1e-9, n ? clk / n : 0.0, cyc_per_mfma);
    fflush(stdout);
}

int main() {
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d\n", cus);
    std::vector<unsigned short> h_rand(4096 * 8), h_zero(4096 * 8, 0), h_small(4096 * 8), h_bf(4096 * 8), h_relu(4096 * 8);
    srand(1);
    for (size_t i = 0; i < h_rand.size(); ++i) {
        // approx. normal(0, 1): sum of uniforms
        float u = 0.f;
        for (int k = 0; k < 12; ++k) u += (float)rand() / RAND_MAX;
        u -= 6.f;
        h_rand[i] = f2h(u);
        h_bf[i] = f2bf(u);
        h_small[i] = f2h(((i * 2654435761u) >> 28 & 1) ? 1.0f : 0.5f);   // two values: few toggling bits
        h_relu[i] = f2h(u > 0.f ? u : 0.f);                               // post-ReLU activations: half zeros
    }
    u32x4 *d_rand, *d_zero, *d_small, *d_bf, *d_relu;
    unsigned long long* d_out; float* d_sink;
    CK(hipMalloc(&d_rand, 65536)); CK(hipMalloc(&d_zero, 65536)); CK(hipMalloc(&d_small, 65536)); CK(hipMalloc(&d_bf, 65536)); CK(hipMalloc(&d_relu, 65536));
    CK(hipMalloc(&d_out, 16 * 8)); CK(hipMalloc(&d_sink, 4));
    CK(hipMemcpy(d_rand, h_rand.data(), 65536, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_zero, h_zero.data(), 65536, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_small, h_small.data(), 65536, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bf, h_bf.data(), 65536, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_relu, h_relu.data(), 65536, hipMemcpyHostToDevice));
    run<0, false>("f16 32x32x16, N(0,1) operands", d_rand, 256, 1, cus, d_out, d_sink);
    run<0, false>("f16 32x32x16, N(0,1) operands", d_rand, 512, 1, cus, d_out, d_sink);
    run<0, false>("f16 32x32x16, zero operands", d_zero, 512, 1, cus, d_out, d_sink);
    run<0, false>("f16 32x32x16, two-valued operands", d_small, 512, 1, cus, d_out, d_sink);
    run<0, false>("f16 32x32x16, ReLU'd N(0,1) operands", d_relu, 512, 1, cus, d_out, d_sink);
    run<1, false>("f16 16x16x32 (x2), N(0,1) operands", d_rand, 512, 1, cus, d_out, d_sink);
    run<2, false>("bf16 32x32x16, N(0,1) operands", d_bf, 512, 1, cus, d_out, d_sink);
    run<0, true>("f16 32x32x16 + 6 LDS frag reads / 8 MFMA", d_rand, 512, 1, cus, d_out, d_sink);
    run<1, true>("f16 16x16x32 (x2) + 6 LDS frag reads / 8", d_rand, 512, 1, cus, d_out, d_sink);
    run<0, false>("f16 32x32x16, N(0,1) operands (again)", d_rand, 512, 1, cus, d_out, d_sink);
    run_lds<0>("f16 32x32x16 fed from LDS (6 frag reads / 8 MFMA), N(0,1)", d_rand, cus, d_out, d_sink);
    run_lds<1>("f16 16x16x32 fed from LDS (12 frag reads / 32 MFMA), N(0,1)", d_rand, cus, d_out, d_sink);
    run_lds<0>("f16 32x32x16 fed from LDS, ReLU'd N(0,1)", d_relu, cus, d_out, d_sink);
    run_lds<1>("f16 16x16x32 fed from LDS, ReLU'd N(0,1)", d_relu, cus, d_out, d_sink);
    run_lds<0>("f16 32x32x16 fed from LDS (again), N(0,1)", d_rand, cus, d_out, d_sink);
    run_lds_wide("f16 32x32x16, ONE wave / SIMD, 128x128 wave tile (8 reads / 16 MFMA), N(0,1)", d_rand, cus, d_out, d_sink);
    run_lds_wide("same, ReLU'd N(0,1)", d_relu, cus, d_out, d_sink);
    run_lds<0>("f16 32x32x16 fed from LDS (6 / 8, 2 waves: once more), N(0,1)", d_rand, cus, d_out, d_sink);
    run_ab("LDS-fed 32x32x16: A dense, B dense", d_rand, d_rand, cus, d_out, d_sink);
    run_ab("LDS-fed 32x32x16: A ReLU'd, B dense", d_relu, d_rand, cus, d_out, d_sink);
    run_ab("LDS-fed 32x32x16: A dense, B ReLU'd", d_rand, d_relu, cus, d_out, d_sink);
    run_ab("LDS-fed 32x32x16: A zero, B dense", d_zero, d_rand, cus, d_out, d_sink);
    run_ab("LDS-fed 32x32x16: A dense, B zero", d_rand, d_zero, cus, d_out, d_sink);
    run_ab("LDS-fed 32x32x16: A dense, B dense (again)", d_rand, d_rand, cus, d_out, d_sink);
    return 0;
}
