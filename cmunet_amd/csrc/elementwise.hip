// elementwise.hip -- HBM-bound forward pieces of the UNet hot path and library plumbing (gfx950).
//   weight packing, BatchNorm statistics finalisation, first-layer (Cin=1) direct conv,
//   BN+ReLU+MaxPool, 1x1 head, layout converters at the module boundary.
// All kernels move 16-byte chunks per lane (coalesced NHWC rows) and keep per-channel parameters in
// registers; reductions use wave64 shuffles + one LDS combine and write per-block partial slabs that a
// second tiny kernel sums in a fixed order (deterministic, no float atomics).
#include "common.h"
#include <string.h>

// ---------------------------------------------------------------------------------------------
// plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void cmu_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* cmu_last_error(void) { return g_err; }
// name of the compute kernel the calling thread launched last through a GEMM-shaped entry point (several kernels serve
// cmu_conv3x3_fwd / _wgrad): lets a profiler attribute the entry's time to the kernel rocprofv3 will list
static thread_local const char* g_kernel_tag = "";
void cmu_set_kernel_tag(const char* tag) { g_kernel_tag = tag; }
extern "C" const char* cmu_last_kernel(void) { return g_kernel_tag; }
extern "C" int cmu_version(void) { return 100; }

// ---- dispatch switches (common.h: CmuSwitch) -----------------------------------------------------------------------------------
static const char* const g_switch_names[CMU_SW_COUNT] = {"CMU_CONV_NARROW", "CMU_CONV_SLIM", "CMU_CONV_PERSIST_PART", "CMU_WGRAD_SQUARE",
                                                         "CMU_WGRAD_WIDE_F32", "CMU_CONV_V5", "CMU_CONV_V6"};
static int g_switch_env[CMU_SW_COUNT] = {-1, -1, -1, -1, -1, -1, -1};        // -1: environment not read yet; 0 / 1 afterwards
static int g_switch_override[CMU_SW_COUNT] = {-1, -1, -1, -1, -1, -1, -1};   // -1: no override
bool cmu_switch_on(int id) {
    const int ov = __atomic_load_n(&g_switch_override[id], __ATOMIC_RELAXED);
    if (ov >= 0) return ov != 0;
    int v = __atomic_load_n(&g_switch_env[id], __ATOMIC_RELAXED);
    if (v < 0) {
        const char* e = getenv(g_switch_names[id]);
        v = (e && e[0] == '0') ? 0 : 1;
        __atomic_store_n(&g_switch_env[id], v, __ATOMIC_RELAXED);
    }
    return v != 0;
}
// 1 while a test forces the switch ON through cmu_set_dispatch_override (conv_igemm6: a forced switch also lifts the launch-size gates)
bool cmu_switch_forced(int id) { return __atomic_load_n(&g_switch_override[id], __ATOMIC_RELAXED) == 1; }
#include <mutex>
static std::mutex g_switch_mutex;   // (test hook: concurrent setters are serialised; readers on the launch path take relaxed atomic loads)
extern "C" int cmu_set_dispatch_override(const char* name, int value) {
    CMU_CHECK_ARG(name != nullptr && value >= -1 && value <= 1, "cmu_set_dispatch_override: value must be -1 (environment), 0 or 1");
    std::lock_guard<std::mutex> lock(g_switch_mutex);
    for (int i = 0; i < CMU_SW_COUNT; ++i)
        if (strcmp(name, g_switch_names[i]) == 0) {
            __atomic_store_n(&g_switch_override[i], value, __ATOMIC_RELAXED);
            return CMU_OK;
        }
    cmu_set_error("cmu_set_dispatch_override: unknown switch '%s'", name);
    return CMU_ERR_ARG;
}
extern "C" int cmu_dtype_size(int dt) { return dt == CMU_F32 ? 4 : (dt == CMU_F16 || dt == CMU_BF16) ? 2 : 0; }

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
template <class TR>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, typename TR::elem_t* __restrict__ out, int Cin, int Cout,
                                    int K, int N, int npad, int64_t total, int tflip) {
    constexpr int KC = 32 / (int)sizeof(typename TR::elem_t);   // one 32-byte K slice per row
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(o % KC);
        const int n = (int)((o / KC) % npad);
        const int t = (int)((o / ((int64_t)KC * npad)) % 9);
        const int s = (int)(o / ((int64_t)KC * npad * 9));
        const int c = s * KC + k;
        float v = 0.f;
        if (n < N && c < K) {
            const int kh = t / 3, kw = t % 3;
            if (!tflip) v = w[(((int64_t)n * Cin + c) * 3 + kh) * 3 + kw];               // n = co, c = ci
            else v = w[(((int64_t)c * Cin + n) * 3 + (2 - kh)) * 3 + (2 - kw)];          // n = ci, c = co
        }
        out[o] = TR::from_float(v);
    }
}

template <class TR>
static int pack_conv3x3_t(const float* w, void* out, int Cin, int Cout, int tflip, hipStream_t st) {
    constexpr int KC = 32 / (int)sizeof(typename TR::elem_t);
    const int K = tflip ? Cout : Cin, N = tflip ? Cin : Cout;
    const int npad = cmu_conv3x3_npad(N);
    const int64_t total = (int64_t)cmu_div_up(K, KC) * 9 * npad * KC;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((pack_conv3x3_kernel<TR>), dim3(grid), dim3(256), 0, st, w, (typename TR::elem_t*)out, Cin, Cout, K, N,
                       npad, total, tflip);
    CMU_CHECK_LAUNCH("cmu_pack_conv3x3");
    return CMU_OK;
}

extern "C" int64_t cmu_pack_conv3x3_elems(int Cin, int Cout, int dt, int tflip) {
    const int es = cmu_dtype_size(dt);
    if (es == 0) return -1;
    const int KC = 32 / es;
    const int K = tflip ? Cout : Cin, N = tflip ? Cin : Cout;
    return (int64_t)cmu_div_up(K, KC) * 9 * cmu_conv3x3_npad(N) * KC;
}
extern "C" int cmu_pack_conv3x3(const float* w, void* out, int Cin, int Cout, int dt, int tflip, void* stream) {
    CMU_CHECK_ARG(w && out && Cin > 0 && Cout > 0, "cmu_pack_conv3x3: bad args");
    CMU_DISPATCH_DT(dt, pack_conv3x3_t, w, out, Cin, Cout, tflip, (hipStream_t)stream);
}

template <class TR>
__global__ void pack_convT_kernel(const float* __restrict__ w, typename TR::elem_t* __restrict__ out, int Cin, int Cout,
                                  int npad, int cpi, int64_t total, int mode) {
    constexpr int KC = 64 / (int)sizeof(typename TR::elem_t);
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(o % KC);
        const int n = (int)((o / KC) % npad);
        const int s = (int)(o / ((int64_t)KC * npad));
        float v = 0.f;
        if (mode == 0) {  // rows n = ij*Cout + co, k -> ci
            const int ci = s * KC + k;
            if (n < 4 * Cout && ci < Cin) {
                const int ij = n / Cout, co = n % Cout;
                v = w[(((int64_t)ci * Cout + co) * 2 + (ij >> 1)) * 2 + (ij & 1)];
            }
        } else {  // slices (ij, cc), rows n = ci, k -> co
            const int ij = s / cpi, cc = s % cpi;
            const int co = cc * KC + k;
            if (n < Cin && co < Cout) v = w[(((int64_t)n * Cout + co) * 2 + (ij >> 1)) * 2 + (ij & 1)];
        }
        out[o] = TR::from_float(v);
    }
}
__host__ __device__ static int64_t convT_pack_elems(int Cin, int Cout, int es, int mode, int* npad_out, int* cpi_out) {
    const int KC = 64 / es;
    int64_t slices;
    int npad, cpi = cmu_div_up(Cout, KC);
    if (mode == 0) { slices = cmu_div_up(Cin, KC); npad = cmu_div_up(4 * Cout, 64) * 64; }
    else { slices = 4 * (int64_t)cpi; npad = cmu_div_up(Cin, 64) * 64; }
    if (npad_out) *npad_out = npad;
    if (cpi_out) *cpi_out = cpi;
    return slices * npad * KC;
}
template <class TR>
static int pack_convT_t(const float* w, void* out, int Cin, int Cout, int mode, hipStream_t st) {
    int npad, cpi;
    const int64_t total = convT_pack_elems(Cin, Cout, (int)sizeof(typename TR::elem_t), mode, &npad, &cpi);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((pack_convT_kernel<TR>), dim3(grid), dim3(256), 0, st, w, (typename TR::elem_t*)out, Cin, Cout, npad, cpi,
                       total, mode);
    CMU_CHECK_LAUNCH("cmu_pack_convT2x2");
    return CMU_OK;
}
extern "C" int64_t cmu_pack_convT2x2_elems(int Cin, int Cout, int dt, int mode) {
    const int es = cmu_dtype_size(dt);
    if (es == 0) return -1;
    return convT_pack_elems(Cin, Cout, es, mode, nullptr, nullptr);
}
extern "C" int cmu_pack_convT2x2(const float* w, void* out, int Cin, int Cout, int dt, int mode, void* stream) {
    CMU_CHECK_ARG(w && out && Cin > 0 && Cout > 0 && (mode == 0 || mode == 1), "cmu_pack_convT2x2: bad args");
    CMU_DISPATCH_DT(dt, pack_convT_t, w, out, Cin, Cout, mode, (hipStream_t)stream);
}

// All packs of a step in ONE launch (a trainer repacks every weight after each optimiser step: 42 launches of 5-40 us
// otherwise).  Descriptor table in device memory; workgroup -> descriptor by its first-block prefix.
struct CmuPackDescDev {
    const float* w;
    void* out;
    int Cin, Cout, mode, kind;   // kind 0: conv3x3 (mode = transpose_flip), 1: convT2x2 (mode 0 / 1)
    int64_t total;               // elements of the packed array
    int64_t block0;              // first workgroup of this descriptor (cmu_pack_desc_blocks of them)
};
// workgroups of one descriptor.  conv3x3: one per (32-byte K slice, 256 / KC weight rows): it reads runs of 9 * KC (or 9 * 256 / KC)
// contiguous floats of the weight tensor, turns them in LDS and writes its nine taps as 256-element runs; convT2x2: 4096 elements each
__host__ __device__ static int64_t pack_desc_blocks(int kind, int Cin, int Cout, int es, int mode, int64_t total) {
    if (kind == 0) {
        const int KC = 32 / es, K = mode ? Cout : Cin, N = mode ? Cin : Cout;
        return (int64_t)cmu_div_up(K, KC) * (cmu_conv3x3_npad(N) / (256 / KC));
    }
    return (total + 4095) / 4096;
}
template <class TR>
__global__ __launch_bounds__(256) void pack_batch_kernel(const CmuPackDescDev* __restrict__ descs, int ndesc) {
    typedef typename TR::elem_t elem_t;
    __shared__ int sd;
    if (threadIdx.x == 0) {
        int lo = 0, hi = ndesc - 1;
        while (lo < hi) {   // last descriptor with block0 <= blockIdx.x
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sd = lo;
    }
    __syncthreads();
    const CmuPackDescDev d = descs[sd];
    const int Cin = d.Cin, Cout = d.Cout;
    elem_t* out = reinterpret_cast<elem_t*>(d.out);
    const int64_t o0 = ((int64_t)blockIdx.x - d.block0) * 4096;
    if (d.kind == 0) {
        // One workgroup = one K slice x NR weight rows x nine taps (2,304 outputs).  Read side: the 9 * KC floats of a row's slice are
        // contiguous in (Cout,Cin,3,3) (forward pack: row n = co, columns ci*9 + tap), and so are the 9 * NR floats of NR consecutive
        // input channels of one output channel (data-gradient pack: row = co of the slice, columns ci*9 + tap) -- nine coalesced loads
        // per thread instead of gathers 36 bytes apart; the tile is turned in LDS and each tap goes out as one 256-element run.
        constexpr int KC = 32 / (int)sizeof(elem_t), NR = 256 / KC;
        __shared__ float tile[2304 + 64];     // rows x (cols + 1): 16 x 145 (16-bit), 32 x 73 / 8 x 289 (f32)
        const int K = d.mode ? Cout : Cin, N = d.mode ? Cin : Cout;
        const int npad = cmu_conv3x3_npad(N), nbr = npad / NR;
        const int64_t bi = (int64_t)blockIdx.x - d.block0;
        const int s = (int)(bi / nbr), n0 = (int)(bi % nbr) * NR;
        const int cols = 9 * (d.mode ? NR : KC);     // rows x cols = 2,304 (rows = NR weight rows, or the KC output channels of the slice)
        for (int e = threadIdx.x; e < 2304; e += 256) {
            const int row = e / cols, col = e % cols;
            float v = 0.f;
            if (!d.mode) {   // row = weight row n0 + row, col = k * 9 + tap
                if (n0 + row < N && s * KC + col / 9 < K) v = d.w[((int64_t)(n0 + row) * Cin + s * KC) * 9 + col];
            } else {         // row = output channel s * KC + row (the K axis), col = (input channel - n0) * 9 + tap
                if (s * KC + row < K && n0 + col / 9 < N) v = d.w[((int64_t)(s * KC + row) * Cin + n0) * 9 + col];
            }
            tile[row * (cols + 1) + col] = v;
        }
        __syncthreads();
        const int nl = threadIdx.x / KC, k = threadIdx.x % KC;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = !d.mode ? tile[nl * (cols + 1) + k * 9 + t] : tile[k * (cols + 1) + nl * 9 + (8 - t)];   // data gradient: flipped taps
            out[(((int64_t)s * 9 + t) * npad + n0 + nl) * KC + k] = TR::from_float(v);
        }
    } else {
        constexpr int KC = 64 / (int)sizeof(elem_t);
        int npad, cpi;
        convT_pack_elems(Cin, Cout, (int)sizeof(elem_t), d.mode, &npad, &cpi);
        for (int i = 0; i < 16; ++i) {
            const int64_t o = o0 + i * 256 + threadIdx.x;
            if (o >= d.total) break;
            const int k = (int)(o % KC);
            const int n = (int)((o / KC) % npad);
            const int s = (int)(o / ((int64_t)KC * npad));
            float v = 0.f;
            if (d.mode == 0) {
                const int ci = s * KC + k;
                if (n < 4 * Cout && ci < Cin) {
                    const int ij = n / Cout, co = n % Cout;
                    v = d.w[(((int64_t)ci * Cout + co) * 2 + (ij >> 1)) * 2 + (ij & 1)];
                }
            } else {
                const int ij = s / cpi, cc = s % cpi;
                const int co = cc * KC + k;
                if (n < Cin && co < Cout) v = d.w[(((int64_t)n * Cout + co) * 2 + (ij >> 1)) * 2 + (ij & 1)];
            }
            out[o] = TR::from_float(v);
        }
    }
}
template <class TR>
static int pack_batch_t(const void* descs, int ndesc, int64_t total_blocks, hipStream_t st) {
    hipLaunchKernelGGL((pack_batch_kernel<TR>), dim3((unsigned)total_blocks), dim3(256), 0, st, (const CmuPackDescDev*)descs, ndesc);
    CMU_CHECK_LAUNCH("cmu_pack_batch");
    return CMU_OK;
}
extern "C" int cmu_pack_desc_bytes(void) { return (int)sizeof(CmuPackDescDev); }
extern "C" int64_t cmu_pack_desc_blocks(int kind, int Cin, int Cout, int dt, int mode) {
    const int es = cmu_dtype_size(dt);
    if (es == 0 || (kind != 0 && kind != 1) || Cin <= 0 || Cout <= 0) return -1;
    const int64_t total = kind == 0 ? cmu_pack_conv3x3_elems(Cin, Cout, dt, mode) : cmu_pack_convT2x2_elems(Cin, Cout, dt, mode);
    return pack_desc_blocks(kind, Cin, Cout, es, mode, total);
}
extern "C" int cmu_pack_batch(const void* descs_dev, int ndesc, int64_t total_blocks, int dt, void* stream) {
    CMU_CHECK_ARG(descs_dev && ndesc > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "cmu_pack_batch: bad args");
    CMU_DISPATCH_DT(dt, pack_batch_t, descs_dev, ndesc, total_blocks, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics: slab [ntiles][2][C] -> mean/var -> pending transform + running stats
// ---------------------------------------------------------------------------------------------
constexpr int BN_MAX_SPLITS = 256;

__global__ void bn_reduce_slab_kernel(const float* __restrict__ stats, int ntiles, int C, int nsplit, double* __restrict__ ws) {
    __shared__ double red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int split = blockIdx.y;
    const int per = (ntiles + nsplit - 1) / nsplit;
    const int t0 = split * per, t1 = min(ntiles, t0 + per);
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int t = t0 + part; t < t1; t += 4) {
            s1 += (double)stats[((int64_t)t * 2 + 0) * C + c];
            s2 += (double)stats[((int64_t)t * 2 + 1) * C + c];
        }
    red[0][part][threadIdx.x & 63] = s1;
    red[1][part][threadIdx.x & 63] = s2;
    __syncthreads();
    if (part == 0 && c < C) {
        const int l = threadIdx.x;
        ws[((int64_t)split * 2 + 0) * C + c] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        ws[((int64_t)split * 2 + 1) * C + c] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ ws, int nsplit, double count, const float* __restrict__ conv_bias,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                   float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                                   float* save_mean, float* save_invstd, int C) {
    // 16 channels x 16 split-parts per 256-thread block; fixed-order combine
    __shared__ double red[2][16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (training && c < C)
        for (int s = part; s < nsplit; s += 16) {
            s1 += ws[((int64_t)s * 2 + 0) * C + c];
            s2 += ws[((int64_t)s * 2 + 1) * C + c];
        }
    red[0][part][cl] = s1;
    red[1][part][cl] = s2;
    __syncthreads();
    if (part != 0 || c >= C) return;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float cb = conv_bias ? conv_bias[c] : 0.f;
    if (training) {
        s1 = 0.0; s2 = 0.0;
        for (int q = 0; q < 16; ++q) { s1 += red[0][q][cl]; s2 += red[1][q][cl]; }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const float sc = (float)((double)g * invstd);
        scale[c] = sc;
        shift[c] = (float)((double)b - mean * (double)g * invstd);
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = (float)invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)(mean + (double)cb);
        if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    } else {
        const float invstd = 1.f / sqrtf(running_var[c] + eps);
        const float sc = g * invstd;
        scale[c] = sc;
        shift[c] = b + (cb - running_mean[c]) * sc;
        if (save_mean) save_mean[c] = running_mean[c] - cb;
        if (save_invstd) save_invstd[c] = invstd;
    }
}

extern "C" int64_t cmu_bn_finalize_ws_bytes(int C) { return (int64_t)BN_MAX_SPLITS * 2 * C * (int64_t)sizeof(double); }

extern "C" int cmu_bn_finalize(const float* stats, int ntiles, int64_t count, const float* conv_bias, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               int training, float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* ws,
                               void* stream) {
    CMU_CHECK_ARG(C > 0 && scale && shift, "cmu_bn_finalize: bad args");
    hipStream_t st = (hipStream_t)stream;
    int nsplit = 1;
    if (training) {
        CMU_CHECK_ARG(stats && ws && ntiles > 0 && count > 0, "cmu_bn_finalize: training needs stats/ws");
        nsplit = ntiles / 16;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > BN_MAX_SPLITS) nsplit = BN_MAX_SPLITS;
        hipLaunchKernelGGL(bn_reduce_slab_kernel, dim3(cmu_div_up(C, 64), nsplit), dim3(256), 0, st, stats, ntiles, C, nsplit,
                           (double*)ws);
        CMU_CHECK_LAUNCH("cmu_bn_finalize(reduce)");
    } else {
        CMU_CHECK_ARG(running_mean && running_var, "cmu_bn_finalize: eval needs running statistics");
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cmu_div_up(C, 16)), dim3(256), 0, st, (const double*)ws, nsplit, (double)count,
                       conv_bias, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift, save_mean,
                       save_invstd, C);
    CMU_CHECK_LAUNCH("cmu_bn_finalize");
    return CMU_OK;
}

// BatchNorm backward phase 1 from a per-tile slab written by the data-gradient kernels (cmu_conv3x3_dgrad_bn,
// cmu_convT2x2_dgrad_bn): same two-level fixed-order reduction as the forward statistics
__global__ void bn_bwd_final_tiles_kernel(const double* __restrict__ ws, int nsplit, double count, float* dgamma, float* dbeta,
                                          float* coef, int C) {
    __shared__ double red[2][16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int s = part; s < nsplit; s += 16) {
            s1 += ws[((int64_t)s * 2 + 0) * C + c];
            s2 += ws[((int64_t)s * 2 + 1) * C + c];
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
extern "C" int cmu_bn_bwd_finalize_tiles(const float* bstats, int ntiles, int64_t count, float* dgamma, float* dbeta, float* coef, int C,
                                         void* ws, void* stream) {
    CMU_CHECK_ARG(bstats && coef && ws && C > 0 && ntiles > 0 && count > 0, "cmu_bn_bwd_finalize_tiles: bad args");
    hipStream_t st = (hipStream_t)stream;
    int nsplit = ntiles / 16;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > BN_MAX_SPLITS) nsplit = BN_MAX_SPLITS;
    hipLaunchKernelGGL(bn_reduce_slab_kernel, dim3(cmu_div_up(C, 64), nsplit), dim3(256), 0, st, bstats, ntiles, C, nsplit, (double*)ws);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_finalize_tiles(reduce)");
    hipLaunchKernelGGL(bn_bwd_final_tiles_kernel, dim3(cmu_div_up(C, 16)), dim3(256), 0, st, (const double*)ws, nsplit, (double)count,
                       dgamma, dbeta, coef, C);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_finalize_tiles");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------
// first layer: Conv2d(1, Cout, 3, p=1) direct, optional patch-mask multiply fused into the load
// ---------------------------------------------------------------------------------------------
#ifndef CMU_C1F_WAVES
#define CMU_C1F_WAVES 4      // waves per SIMD the register allocation is held to (132 registers unconstrained: one wave per SIMD less)
#endif
// Round 4: (i) the next tile's halo is loaded into registers before the current tile's pixels are walked and stored to the other
// LDS buffer behind them (one barrier per tile, no exposed round trip); (ii) the batch statistics are summed per THREAD over all
// tiles of the workgroup and folded once at its end (inside the waves, then across them) -- the per-tile fold (48 shuffles, a
// barrier, a 4-way LDS sum) was a sixth of the pass.  The slab keeps one row per tile: the workgroup's sums go to the row of its
// first tile, zeros to the rows of its other tiles (consumers sum the rows).
template <class TR>
__global__ __launch_bounds__(256, CMU_C1F_WAVES) void conv3x3_c1_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                            int mask_per_sample, const float* __restrict__ w,
                                                            typename TR::elem_t* __restrict__ y, int64_t ldy, float* stats, int B,
                                                            int H, int W, int Cout, int tilesX, int tilesY,
                                                            const int* __restrict__ tlist = nullptr, const int* __restrict__ tcount = nullptr) {
    // tlist (SparK's sparse encoder, cmu_conv3x3_c1_fwd_tiles): only the listed 16 x 16 tiles are computed, round-robin over the
    // workgroups; the slab then has one row per WORKGROUP (all written), and when every listed tile lies inside active patches its
    // sums are the sparse BatchNorm statistics themselves
    constexpr int EPC = TR::EPC;
    __shared__ float halo[2][18 * 18];
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    const int nchunk = Cout / EPC;
    const int ppi = 256 / nchunk;  // pixels per iteration
    const int chunk = tid % nchunk, prow = tid / nchunk;
    const bool active = prow < ppi;
    float wr[EPC][9];   // this thread's 8 (4) filters: loaded once, the workgroup walks several tiles
#pragma unroll
    for (int e = 0; e < EPC; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[e][t] = active ? w[(chunk * EPC + e) * 9 + t] : 0.f;
    const int ntile = B * tilesX * tilesY;
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
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    const int nwork = tlist != nullptr ? tcount[0] : ntile;
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
        if (active)
            for (int pix = prow; pix < 256; pix += ppi) {
                const int py = pix >> 4, px = pix & 15;
                const int gy = ty0 + py, gx = tx0 + px;
                if (gy >= H || gx >= W) continue;
                float in[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) in[t] = hb[(py + t / 3) * 18 + px + t % 3];
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float a = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) a = fmaf(in[t], wr[e][t], a);
                    o[e] = a;
                    s1[e] += a;
                    s2[e] = fmaf(a, a, s2[e]);
                }
                st_global16_nt(reinterpret_cast<unsigned char*>(y) + ((((int64_t)b * H + gy) * W + gx) * ldy + chunk * EPC) * sizeof(typename TR::elem_t),
                            TR::pack(o));
            }
        if (stats != nullptr && it > 0 && tlist == nullptr)      // rows of the workgroup's later tiles: zero (its sums go to its first tile's row)
            for (int c = tid; c < 2 * Cout; c += 256) stats[(int64_t)tile * 2 * Cout + c] = 0.f;
        if (more) halo_store(halo[(it + 1) & 1]);
        __syncthreads();
    }
    if (stats == nullptr || (tlist == nullptr && (int)blockIdx.x >= ntile)) return;
    const int tile = blockIdx.x;                 // slab row: the workgroup's first tile (dense form) / the workgroup (list form)
    if ((nchunk & (nchunk - 1)) == 0 && nchunk <= 32 && nchunk * EPC <= 64) {
        // threads sharing a channel chunk sit nchunk lanes apart: fold inside the wave (fixed xor order), then the four
        // waves through LDS
        float* wred = &red[0][0];   // [4 waves][Cout <= 64][2]
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float a = s1[e], q = s2[e];
            for (int o = nchunk; o < 64; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if ((tid & 63) < nchunk) {
                wred[((tid >> 6) * 64 + (tid & 63) * EPC + e) * 2 + 0] = a;
                wred[((tid >> 6) * 64 + (tid & 63) * EPC + e) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < Cout) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                a += wred[(wv * 64 + tid) * 2 + 0];
                q += wred[(wv * 64 + tid) * 2 + 1];
            }
            stats[((int64_t)tile * 2 + 0) * Cout + tid] = a;
            stats[((int64_t)tile * 2 + 1) * Cout + tid] = q;
        }
        return;
    }
    // general shapes: combine the threads that share a channel chunk (same tid % nchunk): fixed-order tree over prow
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        __syncthreads();
        red[0][tid] = active ? s1[e] : 0.f;
        red[1][tid] = active ? s2[e] : 0.f;
        __syncthreads();
        if (tid < nchunk) {
            float a = 0.f, q = 0.f;
            for (int k = 0; k < ppi; ++k) {
                a += red[0][k * nchunk + tid];
                q += red[1][k * nchunk + tid];
            }
            stats[((int64_t)tile * 2 + 0) * Cout + tid * EPC + e] = a;
            stats[((int64_t)tile * 2 + 1) * Cout + tid * EPC + e] = q;
        }
    }
}

#ifndef CMU_C1F_CAP
#define CMU_C1F_CAP 2048
#endif
template <class TR>
static int conv3x3_c1_fwd_t(const float* x, const uint8_t* mask, int mps, const float* w, void* y, int64_t ldy, float* stats,
                            int B, int H, int W, int Cout, hipStream_t st, const int* tlist = nullptr, const int* tcount = nullptr,
                            int64_t max_tiles = 0) {
    const int tilesX = cmu_div_up(W, 16), tilesY = cmu_div_up(H, 16);
    const int64_t ntile = tlist != nullptr ? max_tiles : (int64_t)B * tilesX * tilesY;
    hipLaunchKernelGGL((conv3x3_c1_fwd_kernel<TR>), dim3((unsigned)(ntile < CMU_C1F_CAP ? ntile : CMU_C1F_CAP)), dim3(256), 0, st, x, mask, mps, w,
                       (typename TR::elem_t*)y, ldy, stats, B, H, W, Cout, tilesX, tilesY, tlist, tcount);
    CMU_CHECK_LAUNCH("cmu_conv3x3_c1_fwd");
    return CMU_OK;
}

extern "C" int cmu_conv3x3_c1_fwd(const float* x, const uint8_t* mask, int mask_per_sample, const float* w, void* y, int64_t ldy,
                                  float* stats, int B, int H, int W, int Cout, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0, "cmu_conv3x3_c1_fwd: bad dtype");
    const int epc = 16 / es;
    CMU_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0, "cmu_conv3x3_c1_fwd: bad args");
    CMU_CHECK_ARG(Cout % epc == 0 && Cout / epc <= 256 && Cout > 0, "cmu_conv3x3_c1_fwd: Cout=%d must be a multiple of %d (<= %d)", Cout, epc, 256 * epc);
    CMU_CHECK_ARG(cmu_aligned16(y) && ldy % epc == 0 && ldy >= Cout, "cmu_conv3x3_c1_fwd: y alignment / stride");
    CMU_DISPATCH_DT(dt, conv3x3_c1_fwd_t, x, mask, mask_per_sample, w, y, ldy, stats, B, H, W, Cout, (hipStream_t)stream);
}

// The same over a list of 16 x 16 tiles (cmu_sparse_tile_list numbering; SparK's sparse encoder, Spark/encoder.py:20-23: the masked tiles
// are never computed, y there is left untouched).  stats: [cmu_conv3x3_c1_fwd_tiles_rows(max_tiles)][2][Cout], fully written -- the sums
// over the listed tiles' pixels.  max_tiles: host-side upper bound of tile_count[0] (sizes the grid).
extern "C" int cmu_conv3x3_c1_fwd_tiles_rows(int64_t max_tiles) { return (int)(max_tiles < CMU_C1F_CAP ? (max_tiles < 1 ? 1 : max_tiles) : CMU_C1F_CAP); }
extern "C" int cmu_conv3x3_c1_fwd_tiles(const float* x, const uint8_t* mask, int mask_per_sample, const float* w, void* y, int64_t ldy,
                                        float* stats, const int* tile_list, const int* tile_count, int64_t max_tiles, int B, int H, int W,
                                        int Cout, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0, "cmu_conv3x3_c1_fwd_tiles: bad dtype");
    const int epc = 16 / es;
    CMU_CHECK_ARG(x && w && y && tile_list && tile_count && B > 0 && H > 0 && W > 0, "cmu_conv3x3_c1_fwd_tiles: bad args");
    CMU_CHECK_ARG(Cout % epc == 0 && Cout / epc <= 256 && Cout > 0, "cmu_conv3x3_c1_fwd_tiles: Cout=%d must be a multiple of %d (<= %d)", Cout, epc, 256 * epc);
    CMU_CHECK_ARG(cmu_aligned16(y) && ldy % epc == 0 && ldy >= Cout, "cmu_conv3x3_c1_fwd_tiles: y alignment / stride");
    if (max_tiles < 1) max_tiles = 1;
    CMU_DISPATCH_DT(dt, conv3x3_c1_fwd_t, x, mask, mask_per_sample, w, y, ldy, stats, B, H, W, Cout, (hipStream_t)stream, tile_list, tile_count,
                    max_tiles);
}

// ---------------------------------------------------------------------------------------------
// BN + ReLU + MaxPool2d(2)
// ---------------------------------------------------------------------------------------------
template <class TR>
__global__ void bnrelu_maxpool_kernel(const unsigned char* __restrict__ y, int64_t ldy, const float* __restrict__ scale,
                                      const float* __restrict__ shift, unsigned char* __restrict__ out, int64_t ldo, int B, int H,
                                      int W, int C, int64_t total, const uint8_t* __restrict__ amask = nullptr, int f = 0, int sbits = 0) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    const int nchunk = C / EPC;
    const int Ho = H / 2, Wo = W / 2;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        int ch, xo, yo, b;
        int64_t pix;
        if (total <= 0x7fffffffll) {      // 32-bit index arithmetic where the tensor allows it (cmu_pixel_coords)
            const unsigned ou = (unsigned)o, pq = ou / (unsigned)nchunk;
            ch = (int)(ou - pq * (unsigned)nchunk);
            pix = pq;
        } else {
            ch = (int)(o % nchunk);
            pix = o / nchunk;
        }
        cmu_pixel_coords(pix, Wo, Ho, total <= 0x7fffffffll, b, yo, xo);
        // SparK's sparse encoder (amask: the patch mask, patches at least 2 px wide at every pooled level, so a window is active or
        // masked as a whole): a masked window pools to zero without reading anything -- the activated, masked copy of y that the
        // plain pool would read (cmu_mask_select) is never materialised
        if (amask != nullptr && !sp_active(amask, f, sbits, b, 2 * yo, 2 * xo, 0)) {
            st_global16(out + ((((int64_t)b * Ho + yo) * Wo + xo) * ldo + ch * EPC) * ES, u32x4{0u, 0u, 0u, 0u});
            continue;
        }
        float sc[EPC], sh[EPC], m[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = scale[ch * EPC + e];
            sh[e] = shift[ch * EPC + e];
            m[e] = 0.f;  // ReLU floor
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t src = (((int64_t)b * H + 2 * yo + (q >> 1)) * W + 2 * xo + (q & 1)) * ldy + ch * EPC;
            float f[EPC];
            TR::unpack(ld_global16_nt(y + src * ES), f);   // the skip is not read again before the decoder
#pragma unroll
            for (int e = 0; e < EPC; ++e) m[e] = fmaxf(m[e], fmaf(f[e], sc[e], sh[e]));
        }
        st_global16(out + ((((int64_t)b * Ho + yo) * Wo + xo) * ldo + ch * EPC) * ES, TR::pack(m));
    }
}
template <class TR>
static int bnrelu_maxpool_t(const void* y, int64_t ldy, const float* scale, const float* shift, void* out, int64_t ldo, int B, int H,
                            int W, int C, hipStream_t st, const uint8_t* amask = nullptr, int f = 0) {
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / TR::EPC);
#ifndef CMU_POOLF_CAP
#define CMU_POOLF_CAP 65536
#endif
    const int grid = (int)(cmu_div_up64(total, 256) < CMU_POOLF_CAP ? cmu_div_up64(total, 256) : CMU_POOLF_CAP);
    hipLaunchKernelGGL((bnrelu_maxpool_kernel<TR>), dim3(grid), dim3(256), 0, st, (const unsigned char*)y, ldy, scale, shift,
                       (unsigned char*)out, ldo, B, H, W, C, total, amask, f, amask ? sp_shift_bits(H, f) : 0);
    CMU_CHECK_LAUNCH("cmu_bnrelu_maxpool_fwd");
    return CMU_OK;
}
extern "C" int cmu_bnrelu_maxpool_fwd(const void* y, int64_t ldy, const float* scale, const float* shift, void* out, int64_t ldo,
                                      int B, int H, int W, int C, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && y && out && scale && shift, "cmu_bnrelu_maxpool_fwd: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0 && B > 0, "cmu_bnrelu_maxpool_fwd: H,W must be even (got %d,%d)", H, W);
    CMU_CHECK_ARG(C % epc == 0 && ldy % epc == 0 && ldo % epc == 0 && ldy >= C && ldo >= C, "cmu_bnrelu_maxpool_fwd: C/ld alignment");
    CMU_CHECK_ARG(cmu_aligned16(y) && cmu_aligned16(out), "cmu_bnrelu_maxpool_fwd: pointer alignment");
    CMU_DISPATCH_DT(dt, bnrelu_maxpool_t, y, ldy, scale, shift, out, ldo, B, H, W, C, (hipStream_t)stream);
}
extern "C" int cmu_bnrelu_maxpool_fwd_masked(const void* y, int64_t ldy, const float* scale, const float* shift, const uint8_t* active, int f,
                                             void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && y && out && scale && shift && active && f > 0, "cmu_bnrelu_maxpool_fwd_masked: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0 && B > 0, "cmu_bnrelu_maxpool_fwd_masked: H,W must be even (got %d,%d)", H, W);
    CMU_CHECK_ARG(C % epc == 0 && ldy % epc == 0 && ldo % epc == 0 && ldy >= C && ldo >= C, "cmu_bnrelu_maxpool_fwd_masked: C/ld alignment");
    CMU_CHECK_ARG(cmu_aligned16(y) && cmu_aligned16(out), "cmu_bnrelu_maxpool_fwd_masked: pointer alignment");
    const int sbits = sp_shift_bits(H, f);
    CMU_CHECK_ARG(sbits >= 1 && (f << sbits) == W, "cmu_bnrelu_maxpool_fwd_masked: H=%d, W=%d must be f=%d times a power of two >= 2 (a pool window lies in one patch)", H, W, f);
    CMU_DISPATCH_DT(dt, bnrelu_maxpool_t, y, ldy, scale, shift, out, ldo, B, H, W, C, (hipStream_t)stream, active, f);
}

// ---------------------------------------------------------------------------------------------
// 1x1 head: logits (B,K,H,W) fp32 NCHW from the raw NHWC tensor + pending transform
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_MAX_K = 8;
template <class TR, int KT>   // KT = compiled class count (2 for the reference's heads, HEAD_MAX_K otherwise)
__global__ void conv1x1_head_fwd_kernel(const unsigned char* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                        const float* __restrict__ shift, const float* __restrict__ w, const float* __restrict__ bias,
                                        float* __restrict__ logits, int B, int H, int W, int C, int K, int64_t npix) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    constexpr int U = 4;         // pixels per thread and trip: four 16-byte loads in flight (this pass is HBM-bound)
    const int nchunk = C / EPC;  // power of two <= 64 (checked on the host)
    const int ch = threadIdx.x % nchunk;
    const int ppb = blockDim.x / nchunk;
    float sc[EPC], sh[EPC], wk[KT][EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = scale ? scale[ch * EPC + e] : 1.f;
        sh[e] = scale ? shift[ch * EPC + e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) wk[k][e] = (k < K) ? w[k * C + ch * EPC + e] : 0.f;
    const float lo = scale ? 0.f : -__builtin_inff();
    const int64_t HW = (int64_t)H * W;
    for (int64_t p0 = (int64_t)blockIdx.x * ppb * U; p0 < npix; p0 += (int64_t)gridDim.x * ppb * U) {
        u32x4 q[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pix = p0 + u * ppb + threadIdx.x / nchunk;
            ok[u] = pix < npix;
            q[u] = ok[u] ? ld_global16(x + (pix * ldx + ch * EPC) * ES) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pix = p0 + u * ppb + threadIdx.x / nchunk;
            float f[EPC];
            TR::unpack(q[u], f);
            float acc[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[k] = 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float a = fmaxf(fmaf(f[e], sc[e], sh[e]), lo);
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] = fmaf(a, wk[k][e], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < KT; ++k)
                if (k < K)
                    for (int o = 1; o < nchunk; o <<= 1) acc[k] += __shfl_xor(acc[k], o, 64);
            if (ok[u] && ch == 0) {
                // (32-bit division where the pixel index allows it -- every shape of the benches: a 64-bit division is ~10x the
                // instructions, executed by the whole wave for its eight storing lanes)
                int64_t b, r;
                if (npix <= 0x7fffffffll) {
                    const unsigned bq = (unsigned)pix / (unsigned)HW;
                    b = bq; r = (unsigned)pix - bq * (unsigned)HW;
                } else {
                    b = pix / HW; r = pix % HW;
                }
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < K) logits[(b * K + k) * HW + r] = acc[k] + bias[k];
            }
        }
    }
}
template <class TR>
static int conv1x1_head_fwd_t(const void* x, int64_t ldx, const float* scale, const float* shift, const float* w, const float* bias,
                              float* logits, int B, int H, int W, int C, int K, hipStream_t st) {
    const int nchunk = C / TR::EPC;
    const int64_t npix = (int64_t)B * H * W;
    const int ppb = 256 / nchunk;
    const int64_t nb = cmu_div_up64(npix, ppb * 4);
#ifndef CMU_HEADF_CAP
#define CMU_HEADF_CAP 8192
#endif
    const int grid = (int)(nb < CMU_HEADF_CAP ? nb : CMU_HEADF_CAP);
    if (K <= 2)
        hipLaunchKernelGGL((conv1x1_head_fwd_kernel<TR, 2>), dim3(grid), dim3(256), 0, st, (const unsigned char*)x, ldx, scale, shift, w,
                           bias, logits, B, H, W, C, K, npix);
    else
        hipLaunchKernelGGL((conv1x1_head_fwd_kernel<TR, HEAD_MAX_K>), dim3(grid), dim3(256), 0, st, (const unsigned char*)x, ldx, scale,
                           shift, w, bias, logits, B, H, W, C, K, npix);
    CMU_CHECK_LAUNCH("cmu_conv1x1_head_fwd");
    return CMU_OK;
}
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
extern "C" int cmu_conv1x1_head_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, const float* w,
                                    const float* bias, float* logits, int B, int H, int W, int C, int K, int dt, void* stream) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0 && x && w && bias && logits && B > 0 && H > 0 && W > 0, "cmu_conv1x1_head_fwd: bad args");
    const int epc = 16 / es;
    CMU_CHECK_ARG(K >= 1 && K <= HEAD_MAX_K, "cmu_conv1x1_head_fwd: K=%d must be in 1..%d", K, HEAD_MAX_K);
    CMU_CHECK_ARG(C % epc == 0 && is_pow2(C / epc) && C / epc <= 64, "cmu_conv1x1_head_fwd: C=%d: C/%d must be a power of two <= 64", C, epc);
    CMU_CHECK_ARG(cmu_aligned16(x) && ldx % epc == 0 && ldx >= C, "cmu_conv1x1_head_fwd: x alignment / stride");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_conv1x1_head_fwd: scale/shift must both be set");
    CMU_DISPATCH_DT(dt, conv1x1_head_fwd_t, x, ldx, in_scale, in_shift, w, bias, logits, B, H, W, C, K, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// module-boundary layout converters (not on the fused hot path)
// ---------------------------------------------------------------------------------------------
template <class TR>
__global__ void apply_to_nchw_kernel(const typename TR::elem_t* __restrict__ y, int64_t ldy, const float* __restrict__ scale,
                                     const float* __restrict__ shift, int relu_from, float* __restrict__ out, int B, int H, int W,
                                     int C, int64_t total) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(o % W), yy = (int)((o / W) % H), c = (int)((o / ((int64_t)W * H)) % C), b = (int)(o / ((int64_t)W * H * C));
        float v = TR::to_float(y[(((int64_t)b * H + yy) * W + x) * ldy + c]);
        if (scale) {
            v = fmaf(v, scale[c], shift[c]);
            if (cmu_relu_on(c, relu_from)) v = fmaxf(v, 0.f);
        }
        out[o] = v;
    }
}
template <class TR>
static int apply_to_nchw_t(const void* y, int64_t ldy, const float* scale, const float* shift, int relu_from, float* out, int B,
                           int H, int W, int C, hipStream_t st) {
    const int64_t total = (int64_t)B * C * H * W;
    const int grid = (int)(cmu_div_up64(total, 256) < 8192 ? cmu_div_up64(total, 256) : 8192);
    hipLaunchKernelGGL((apply_to_nchw_kernel<TR>), dim3(grid), dim3(256), 0, st, (const typename TR::elem_t*)y, ldy, scale, shift,
                       relu_from, out, B, H, W, C, total);
    CMU_CHECK_LAUNCH("cmu_apply_to_nchw");
    return CMU_OK;
}
extern "C" int cmu_apply_to_nchw(const void* y, int64_t ldy, const float* scale, const float* shift, int relu_from, float* out,
                                 int B, int H, int W, int C, int dt, void* stream) {
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0 && y && out && B > 0 && H > 0 && W > 0 && C > 0 && ldy >= C, "cmu_apply_to_nchw: bad args");
    CMU_CHECK_ARG((scale == nullptr) == (shift == nullptr), "cmu_apply_to_nchw: scale/shift must both be set");
    CMU_DISPATCH_DT(dt, apply_to_nchw_t, y, ldy, scale, shift, relu_from, out, B, H, W, C, (hipStream_t)stream);
}
extern "C" int cmu_nhwc_to_nchw(const void* x, int64_t ldx, float* out, int B, int H, int W, int C, int dt, void* stream) {
    return cmu_apply_to_nchw(x, ldx, nullptr, nullptr, 0, out, B, H, W, C, dt, stream);
}

template <class TR>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, typename TR::elem_t* __restrict__ out, int64_t ldo, int B, int H,
                                    int W, int C, int64_t total) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        const int64_t pix = o / C;
        const int xx = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        out[pix * ldo + c] = TR::from_float(x[(((int64_t)b * C + c) * H + yy) * W + xx]);
    }
}
template <class TR>
static int nchw_to_nhwc_t(const float* x, void* out, int64_t ldo, int B, int H, int W, int C, hipStream_t st) {
    const int64_t total = (int64_t)B * C * H * W;
    const int grid = (int)(cmu_div_up64(total, 256) < 8192 ? cmu_div_up64(total, 256) : 8192);
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<TR>), dim3(grid), dim3(256), 0, st, x, (typename TR::elem_t*)out, ldo, B, H, W, C, total);
    CMU_CHECK_LAUNCH("cmu_nchw_to_nhwc");
    return CMU_OK;
}
extern "C" int cmu_nchw_to_nhwc(const float* x, void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream) {
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0 && x && out && B > 0 && H > 0 && W > 0 && C > 0 && ldo >= C, "cmu_nchw_to_nhwc: bad args");
    CMU_DISPATCH_DT(dt, nchw_to_nhwc_t, x, out, ldo, B, H, W, C, (hipStream_t)stream);
}
