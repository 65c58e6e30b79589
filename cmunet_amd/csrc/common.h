// common.h -- shared host/device helpers for libcmunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cmunet_hip.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void cmu_set_error(const char* fmt, ...);
void cmu_set_kernel_tag(const char* tag);

#define CMU_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            cmu_set_error(__VA_ARGS__);          \
            return CMU_ERR_ARG;                  \
        }                                        \
    } while (0)

#define CMU_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            cmu_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return CMU_ERR_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

static inline bool cmu_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Dispatch switches that tests A/B inside ONE process (elementwise.hip): the environment variable of the same name is read ONCE
// (no getenv on the launch path: an environment scan per conv launch, racing with setenv from loader threads -- advisor, round 3);
// cmu_set_dispatch_override (test entry of the C-ABI) forces a value afterwards.  Default of every switch: on.
enum CmuSwitch { CMU_SW_CONV_NARROW = 0, CMU_SW_CONV_SLIM, CMU_SW_CONV_PERSIST_PART, CMU_SW_WGRAD_SQUARE, CMU_SW_WGRAD_WIDE_F32, CMU_SW_CONV_V5, CMU_SW_CONV_V6, CMU_SW_COUNT };
bool cmu_switch_on(int id);
bool cmu_switch_forced(int id);

// the calling thread's current HIP device (the one its launches go to)
static inline int cmu_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}
// one flag per device for work that has to be done once on each device the process launches on (function attributes
// such as the dynamic LDS limit are per device, not per process)
struct CmuPerDevice {
    unsigned long long mask[4] = {0ull, 0ull, 0ull, 0ull};   // up to 256 devices
    bool done() const {
        const int d = cmu_current_device() & 255;
        return (__atomic_load_n(&mask[d >> 6], __ATOMIC_ACQUIRE) >> (d & 63)) & 1ull;
    }
    void mark() {
        const int d = cmu_current_device() & 255;
        __atomic_fetch_or(&mask[d >> 6], 1ull << (d & 63), __ATOMIC_RELEASE);
    }
};

// device-resident state of the dynamic loss scaler (cmu_amp_*; 32 bytes, see include/cmunet_hip.h)
struct CmuAmpState {
    float scale;          // current loss scale
    float found_inf;      // 1 when the gradients of the step in flight hold an inf / nan
    int growth_tracker;   // consecutive clean steps since the last change of the scale
    int good_steps;       // optimiser updates actually taken (the step number of Adam's bias corrections)
    int skipped_steps;
    int pad[3];
};

// ---------------------------------------------------------------------------------------------
// vector types
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---------------------------------------------------------------------------------------------
// dtype traits: T is the storage type of activations / packed weights.
//   EPC  = elements per 16-byte chunk
//   unpack(chunk, float[EPC]) / pack(float[EPC]) -> chunk
//   mma16(a_chunk, b_chunk, acc): acc(32x32 f32) += A(32 x Kc) * B(Kc x 32), Kc = 2*EPC channels split
//          over the two lane halves (lane half h holds channels [h*EPC, h*EPC+EPC) of the 32-byte k-step)
// ---------------------------------------------------------------------------------------------
struct F32Traits {
    typedef float elem_t;
    static constexpr int EPC = 4;
    static constexpr int DT = CMU_F32;
    __device__ static inline void unpack(const u32x4& c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
    }
    __device__ static inline u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
        return c;
    }
    __device__ static inline void mma16(const u32x4& a, const u32x4& b, f32x16& acc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i]), __uint_as_float(b[i]), acc, 0, 0, 0);
    }
    __device__ static inline float to_float(float v) { return v; }
    __device__ static inline float from_float(float v) { return v; }
    __device__ static inline uint32_t pack2(float a, float) { return __float_as_uint(a); }   // (16-bit paths only; never called)
    __device__ static inline void unpack2(uint32_t c, float& lo, float& hi) { lo = __uint_as_float(c); hi = 0.f; }   // (same)
};

struct F16Traits {
    typedef _Float16 elem_t;
    static constexpr int EPC = 8;
    static constexpr int DT = CMU_F16;
    __device__ static inline void unpack(const u32x4& c, float* f) {
        f16x8 h = __builtin_bit_cast(f16x8, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
    }
    // (pairs: the vector conversion compiles to gfx950's v_cvt_pk_f16_f32 -- one instruction per two elements, round to
    // nearest even like the scalar cast)
    __device__ static inline u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = pack2(f[2 * i], f[2 * i + 1]);
        return c;
    }
    __device__ static inline void mma16(const u32x4& a, const u32x4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    // acc(16x16 f32) += A(16 x 32) * B(32 x 16): lane l holds A[l & 15][8 (l >> 4) ..+8), B[8 (l >> 4) ..+8)[l & 15]; D[4 (l >> 4) + e][l & 15]
    __device__ static inline void mma32(const u32x4& a, const u32x4& b, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    __device__ static inline float to_float(_Float16 v) { return (float)v; }
    __device__ static inline _Float16 from_float(float v) { return (_Float16)v; }
    __device__ static inline uint32_t pack2(float a, float b) {   // a in the low half, b in the high half
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
        return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, f16x2_t));
    }
    __device__ static inline void unpack2(uint32_t c, float& lo, float& hi) {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
        const f16x2_t h = __builtin_bit_cast(f16x2_t, c);
        lo = (float)h[0];
        hi = (float)h[1];
    }
};

struct BF16Traits {
    typedef __bf16 elem_t;
    static constexpr int EPC = 8;
    static constexpr int DT = CMU_BF16;
    __device__ static inline void unpack(const u32x4& c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(c[i] << 16);
            f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
        }
    }
    __device__ static inline u32x4 pack(const float* f) {
        bf16x8 h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (__bf16)f[i];
        return __builtin_bit_cast(u32x4, h);
    }
    __device__ static inline void mma16(const u32x4& a, const u32x4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    __device__ static inline void mma32(const u32x4& a, const u32x4& b, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    __device__ static inline float to_float(__bf16 v) { return (float)v; }
    __device__ static inline __bf16 from_float(float v) { return (__bf16)v; }
    // two conversions in one instruction (round to nearest even, same as the cast): a in the low half, b in the high half
    __device__ static inline uint32_t pack2(float a, float b) {
        uint32_t r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    __device__ static inline void unpack2(uint32_t c, float& lo, float& hi) {
        lo = __uint_as_float(c << 16);
        hi = __uint_as_float(c & 0xffff0000u);
    }
};

// dispatch a templated launcher on the runtime dtype
#define CMU_DISPATCH_DT(dt, FN, ...)                                        \
    do {                                                                    \
        switch (dt) {                                                       \
            case CMU_F32: return FN<F32Traits>(__VA_ARGS__);                \
            case CMU_F16: return FN<F16Traits>(__VA_ARGS__);                \
            case CMU_BF16: return FN<BF16Traits>(__VA_ARGS__);              \
            default: cmu_set_error("unknown dtype %d", (int)(dt)); return CMU_ERR_ARG; \
        }                                                                   \
    } while (0)

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ static inline u32x4 ld_global16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ static inline void st_global16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
// streaming variants: data that is dead after this read / not read again soon (no reason to keep it in L2)
__device__ static inline u32x4 ld_global16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ static inline void st_global16_nt(void* p, const u32x4& v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }

__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ static inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ static inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// SparK patch mask: pixel (y, x) of a level whose side is f << sbits looks up active[b][y >> sbits][x >> sbits]
// (b, y, x) of pixel index p of a (B, H, W) grid.  ``small`` (uniform: the grid has fewer than 2^31 pixels -- every shape of the
// benches) takes 32-bit divisions: a 64-bit division is ~10x the instructions, and the element-wise passes paid three of them per
// 16-byte chunk (mask-select 1.8 ms per SparK step for 4.2 GB, max-pool backward, ...).
__device__ static inline void cmu_pixel_coords(int64_t p, int W, int H, bool small, int& b, int& y, int& x) {
    if (small) {
        const unsigned pu = (unsigned)p, t = pu / (unsigned)W;
        x = (int)(pu - t * (unsigned)W);
        const unsigned bb = t / (unsigned)H;
        y = (int)(t - bb * (unsigned)H);
        b = (int)bb;
    } else {
        x = (int)(p % W);
        y = (int)((p / W) % H);
        b = (int)(p / ((int64_t)W * H));
    }
}

__device__ static inline bool sp_active(const uint8_t* __restrict__ active, int f, int sbits, int b, int y, int x, int invert) {
    const bool a = active[((int64_t)b * f + (y >> sbits)) * f + (x >> sbits)] != 0;
    return invert ? !a : a;
}
// `relu_from` of a pending transform: channels c >= relu_from of the view are activated (a concat input whose LEFT part -- a
// ConvTranspose output -- carries no activation); a NEGATIVE value -n activates the channels c < n instead (round 4: the concat view
// (skip, up) of a decoder that shares its skip with another one -- see engine.decoder_forward, `skip_first`).
__host__ __device__ static inline bool cmu_relu_on(int c0, int relu_from) { return relu_from >= 0 ? c0 >= relu_from : c0 < -relu_from; }
static inline int sp_shift_bits(int H, int f) {
    int s = 0;
    while ((f << s) < H) ++s;
    return ((f << s) == H) ? s : -1;
}

// spatial tile of the implicit-GEMM kernels
constexpr int CMU_TH = 16;
constexpr int CMU_TW = 16;
__host__ __device__ static inline int cmu_div_up(int a, int b) { return (a + b - 1) / b; }
// rows per (slice, tap) of the packed 3x3 weights: [K/32B][9][npad][32 B]; 64 for narrow layers, else a multiple of 128
__host__ __device__ static inline int cmu_conv3x3_npad(int N) { return N <= 64 ? 64 : ((N + 127) / 128) * 128; }
static inline int64_t cmu_div_up64(int64_t a, int64_t b) { return (a + b - 1) / b; }
