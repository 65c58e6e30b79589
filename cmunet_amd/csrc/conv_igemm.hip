// conv_igemm.hip -- implicit-GEMM convolution on MFMA for gfx950 (CDNA4).
//
// One kernel template serves three GEMM-shaped ops of the UNet hot path:
//   MODE_CONV3       Conv2d 3x3 p1 forward (Finetuning/model.py:17,20) and, with flipped/transposed
//                    packed weights, its data gradient.
//   MODE_CONVT_FWD   ConvTranspose2d 2x2 s2 forward (model.py:60,78): GEMM M=pixels, N=4*Cout, K=Cin,
//                    pixel-shuffle store + bias.
//   MODE_CONVT_DGRAD ConvTranspose2d data gradient: M=low-res pixels, N=Cin, K=4*Cout gathered stride-2.
//
// Design (MI355X-first, not a translation of a cuDNN/CUTLASS tiling):
//   * workgroup = 256 threads = 4 wave64; output tile = 16x16 pixels (M=256) x 64 channels (N=64);
//     wave w owns tile rows 4w..4w+3 (M=64) x N=64 as 2x2 MFMA 32x32 accumulators (64 VGPRs).
//   * K is consumed in 64-BYTE channel slices per pixel (32 x f16/bf16 or 16 x f32).  For the 3x3 conv
//     one stage = one slice of the 18x18 input halo (staged ONCE in LDS and re-read at 9 shifted
//     addresses: no im2col buffer, no 9x re-read of the input) + the 9 tap slices of the weights.
//   * global -> registers -> LDS staging (not LDS-DMA) because the producer's BatchNorm+ReLU is applied
//     on the way (x*scale[c]+shift[c], max 0) and conv zero-padding must stay zero AFTER that transform;
//     the loads for stage s+1 are issued before the MFMAs of stage s (register prefetch) and two
//     workgroups per CU overlap one's staging with the other's MFMAs.
//   * LDS rows are padded to 80 B (64+16): 16-byte fragment reads (ds_read_b128) of consecutive
//     pixels / output channels then spread over all 16 slots of the 256-B bank row.
//   * every MFMA operand fragment is one 16-byte LDS read per lane: 8 x 16-bit -> one
//     v_mfma_f32_32x32x16_{f16,bf16}; 4 x f32 -> four v_mfma_f32_32x32x2_f32 (exact fp32 chain),
//     so the three dtypes share one kernel body (traits in common.h).
//   * epilogue: per-channel sum / sum-of-squares of the raw output (BatchNorm batch statistics) from the
//     accumulators (in-lane over pixels, one cross-half shuffle, 4-wave LDS combine) into a per-tile slab
//     (deterministic, no atomics); the tile is transposed through LDS and written as whole 16-byte
//     channel chunks per pixel (coalesced NHWC rows).
#include "common.h"
#include <stdlib.h>

enum { MODE_CONV3 = 0, MODE_CONVT_FWD = 1, MODE_CONVT_DGRAD = 2 };

struct IGParams {
    const void* x;
    int64_t ldx;
    const float* in_scale;
    const float* in_shift;
    int relu_from;
    const void* w;
    int npad;  // padded rows per (slice, tap) in the packed weights
    void* y;
    int64_t ldy;
    float* stats;
    const float* bias;
    int B, H, W;  // pixels of the GEMM's M dimension
    int K;        // contraction channels per tap (CONV3: Cin; CONVT_FWD: Cin; CONVT_DGRAD: Cout of the convT)
    int N;        // GEMM N (CONV3: Cout; CONVT_FWD: 4*Cout; CONVT_DGRAD: Cin)
    int Cq;       // CONVT_FWD: Cout (n -> (ij, co))
    int nslices;  // total number of 64-byte K slices
    int nslices32;  // CONV3: number of 32-byte slices in the packed weights
    int tilesX, tilesY, nblk;
    int64_t total_blocks;
    // optional BatchNorm+ReLU backward statistics of the layer the output gradient flows into (dgrad entries)
    const void* bx;       // that layer's raw conv output (same pixel grid / channels as y)
    int64_t ldbx;
    const float *b_scale, *b_shift, *b_mean, *b_invstd;
    float* bstats;        // slab [cmu_conv_ntiles][2][N]
    int buf_ok;           // first kernel, 3x3 mode: one image and the weight pack fit buffer descriptors
    // persistent wide kernel: division by nblk / tilesX / tilesY as (mulhi(n, M) + n) >> s (n < 2^31)
    unsigned fd_nblk[2], fd_tx[2], fd_ty[2];
    // sparse (SparK) form of the persistent kernel: only the spatial tiles tile_list[0 .. *tile_count) are computed (device
    // arrays written by cmu_sparse_tile_list; dense tile numbering (b * tilesY + ty) * tilesX + tx); the rest of y is untouched
    const int* tile_list;
    const int* tile_count;
};
// round-up magic for unsigned division by d >= 1: s = ceil(log2 d), M = floor(2^32 * (2^s - d) / d) + 1
static inline void cmu_fastdiv_init(unsigned d, unsigned* out) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    out[0] = (unsigned)((((1ull << s) - d) << 32) / d + 1);
    out[1] = s;
}

template <class TR, int MODE>
struct IGCfg {
    typedef typename TR::elem_t elem_t;
    static constexpr bool C3 = (MODE == MODE_CONV3);
    static constexpr int TAPS = C3 ? 9 : 2;  // K slices consumed per stage
    static constexpr int HALO = C3 ? 1 : 0;
    static constexpr int LW = CMU_TW + 2 * HALO;
    static constexpr int LH = CMU_TH + 2 * HALO;
    static constexpr int NPIX = LW * LH;
    static constexpr int SLICES_A = C3 ? 1 : TAPS;  // slices stored per LDS pixel
    static constexpr int PS_A = SLICES_A * 64 + 16;
    static constexpr int A_BYTES = NPIX * PS_A;
    static constexpr int PS_W = 80;
    static constexpr int BN = 64;
    static constexpr int W_BYTES = TAPS * BN * PS_W;
    static constexpr int KC = 64 / (int)sizeof(elem_t);
    static constexpr int EPC = TR::EPC;
    static constexpr int EPI_ROW = BN * (int)sizeof(elem_t) + 16;
    static constexpr int EPI_STAGE = 4 * 64 * EPI_ROW;
    static constexpr int EPI_BYTES = EPI_STAGE + 4 * BN * 2 * 4;
    static constexpr int MAIN_BYTES = A_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static constexpr int A_CHUNKS = NPIX * SLICES_A * 4;
    static constexpr int A_ITERS = (A_CHUNKS + 255) / 256;
    static constexpr int W_CHUNKS = TAPS * BN * 4;
    static constexpr int W_ITERS = (W_CHUNKS + 255) / 256;
};

// MFMA row index r (0..31) of an M-block -> pixel of its 2x16 pixel strip.  ds_read_b128 services a wave in
// four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32); giving each hardware group the 16
// CONSECUTIVE pixels of one strip row makes the 80-byte-stride A-fragment reads conflict-free (the natural
// r -> (r>>4, r&15) map mixes the two strip rows inside a group: 2-way conflicts, measured 38 % of LDS cycles).
__device__ static inline int pm_row(int r) { return __builtin_popcount(r >> 2) & 1; }
__device__ static inline int pm_col(int r) { return 4 * (r >> 3) + (r & 3); }

#ifdef CMU_IG_STAMPS
// diagnostic build only (tools/igemm_stamps.py): wave 0 of every 61st workgroup stamps s_memtime around the phases of
// its first 16 stages into a buffer nothing else reads
__device__ unsigned long long g_ig_stamps[64 * 16 * 8];
extern "C" int cmu_debug_ig_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ig_stamps), sizeof(g_ig_stamps));
}
#define IG_STAMP(k)                                                                          \
    do {                                                                                      \
        if (stamp_slot >= 0 && s < 16 && tid == 0)                                            \
            g_ig_stamps[(stamp_slot * 16 + s) * 8 + (k)] = __builtin_amdgcn_s_memtime();      \
    } while (0)
#else
#define IG_STAMP(k) do {} while (0)
#endif

template <class TR, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const IGParams p) {
    typedef IGCfg<TR, MODE> C;
    typedef typename TR::elem_t elem_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;

    // ---- block -> (spatial tile, n-block); XCD-aware: blocks b and b+8 share an XCD (speed only) ----
    int64_t bid = blockIdx.x;
    {
        const int64_t nb = p.total_blocks;
        const int64_t q = nb >> 3, rem = nb & 7;
        const int64_t xcd = bid & 7, local = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    }
    const int nbk = (int)(bid % p.nblk);
    const int64_t tile = bid / p.nblk;
    const int tx = (int)(tile % p.tilesX);
    const int ty = (int)((tile / p.tilesX) % p.tilesY);
    const int b = (int)(tile / ((int64_t)p.tilesX * p.tilesY));
    const int ty0 = ty * CMU_TH, tx0 = tx * CMU_TW;
    const int n0 = nbk * C::BN;

    const int Hin = (MODE == MODE_CONVT_DGRAD) ? 2 * p.H : p.H;
    const int Win = (MODE == MODE_CONVT_DGRAD) ? 2 * p.W : p.W;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
    const int cpi = (MODE == MODE_CONVT_DGRAD) ? (p.K + C::KC - 1) / C::KC : 1;  // slices per (i,j)

    // ---- per-thread staging roles ---------------------------------------------------------------
    const int cg = tid & 3;                                   // 16-byte chunk within the 64-byte slice
    const int aslice = C::C3 ? 0 : ((tid >> 2) & (C::SLICES_A - 1));
    // A chunk it: LDS pixel index and global pixel offset (in elements) / validity, stage independent
    int a_lds[C::A_ITERS];
    int64_t a_goff[C::A_ITERS];
    unsigned a_inb = 0;  // bit it: pixel inside the image (and chunk index in range)
#pragma unroll
    for (int it = 0; it < C::A_ITERS; ++it) {
        const int idx = it * 256 + tid;
        const int rest = idx >> 2;
        const int pix = C::C3 ? rest : rest / C::SLICES_A;
        const int py = pix / C::LW, px = pix % C::LW;
        int gy = ty0 - C::HALO + py, gx = tx0 - C::HALO + px;
        bool ok = (idx < C::A_CHUNKS) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_lds[it] = pix * C::PS_A + aslice * 64 + cg * 16;
        if (MODE == MODE_CONVT_DGRAD) { gy *= 2; gx *= 2; }
        a_goff[it] = ok ? (((int64_t)b * Hin + gy) * Win + gx) * p.ldx : 0;
        a_inb |= (ok ? 1u : 0u) << it;
    }
    int w_lds[C::W_ITERS];
    int w_n[C::W_ITERS];  // output-channel row of the packed weights
    int w_t[C::W_ITERS];  // tap / slice within the stage
#pragma unroll
    for (int it = 0; it < C::W_ITERS; ++it) {
        const int idx = it * 256 + tid;
        const int row = idx >> 2;
        const int t = row / C::BN, n = row % C::BN;
        w_lds[it] = C::A_BYTES + row * C::PS_W + cg * 16;
        w_n[it] = n0 + n;
        w_t[it] = t;
    }

    // ---- 3x3 mode: raw buffer loads (see conv_igemm3.inc): loop-invariant per-thread offsets, the slice advance is a
    // scalar offset, pixels outside the image (and the chunks of a ragged last slice) point past num_records and read
    // as zero -- no per-load address arithmetic, no predicated loads, no branches around them.  The address path of the
    // ConvTranspose modes depends on the slice per thread and keeps plain loads.
    constexpr unsigned OOBV = 0x7fff0000u;
    unsigned a_voff[C::A_ITERS], w_voff[C::W_ITERS];
    const bool use_buf = C::C3 && p.buf_ok;
    // (descriptors are built for every mode -- the type has no default state -- and only used by the 3x3 buffer path)
    const int64_t img_b = (int64_t)p.H * p.W * p.ldx * (int64_t)sizeof(elem_t);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(xb) + (use_buf ? b * img_b : 0), 0,
        (int)(unsigned)((((int64_t)p.H * p.W - 1) * p.ldx + p.K) * (int64_t)sizeof(elem_t)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wb), 0, (int)(unsigned)((int64_t)p.nslices32 * 9 * p.npad * 32), 0x00020000);
    if (C::C3) {
#pragma unroll
        for (int it = 0; it < C::A_ITERS; ++it)
            a_voff[it] = ((a_inb >> it) & 1u) ? (unsigned)((a_goff[it] - ((int64_t)b * p.H * p.W) * p.ldx + cg * C::EPC) * (int64_t)sizeof(elem_t)) : OOBV;
#pragma unroll
        for (int it = 0; it < C::W_ITERS; ++it)
            w_voff[it] = (it * 256 + tid < C::W_CHUNKS) ? (unsigned)(((int64_t)((cg >> 1) * 9 + w_t[it]) * p.npad + w_n[it]) * 32 + (cg & 1) * 16) : OOBV;
    }

    // ---- prefetch registers -----------------------------------------------------------------------
    u32x4 areg[C::A_ITERS];
    u32x4 wreg[C::W_ITERS];
    float sc[C::EPC], sh[C::EPC];
    bool relu = false, chan_ok = false;
    const bool has_tf = (p.in_scale != nullptr);

    float lo_relu = 0.f;
    auto load_stage = [&](int s) {
        if (C::C3 && use_buf) {
            const int c0 = s * C::KC + cg * C::EPC;
            chan_ok = c0 < p.K;
#pragma unroll
            for (int it = 0; it < C::A_ITERS; ++it)
                areg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(chan_ok ? a_voff[it] : OOBV), s * 64, 0));
            if (has_tf) {
                const int cc = chan_ok ? c0 : 0;
#pragma unroll
                for (int e = 0; e < C::EPC; ++e) {
                    sc[e] = p.in_scale[cc + e];
                    sh[e] = p.in_shift[cc + e];
                }
                lo_relu = cmu_relu_on(c0, p.relu_from) ? 0.f : -__builtin_inff();
            }
            const bool w_ok = (2 * s + (cg >> 1)) < p.nslices32;
            const unsigned wso = (unsigned)(2 * s * 9) * (unsigned)p.npad * 32u;
#pragma unroll
            for (int it = 0; it < C::W_ITERS; ++it)
                wreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(w_ok ? w_voff[it] : OOBV), (int)wso, 0));
            return;
        }
        // channel slice handled by this thread in stage s
        const int kslice = C::C3 ? s : s * C::TAPS + aslice;
        int c0;         // first channel of this thread's chunk (within the input tensor)
        int64_t extra;  // extra element offset (CONVT_DGRAD: (i,j) sub-pixel)
        if (MODE == MODE_CONVT_DGRAD) {
            const int ij = kslice / cpi, cc = kslice % cpi;
            c0 = cc * C::KC + cg * C::EPC;
            chan_ok = (kslice < p.nslices) && (c0 < p.K);
            extra = ((int64_t)(ij >> 1) * Win + (ij & 1)) * p.ldx;
        } else {
            c0 = kslice * C::KC + cg * C::EPC;
            chan_ok = (kslice < p.nslices) && (c0 < p.K);
            extra = 0;
        }
#pragma unroll
        for (int it = 0; it < C::A_ITERS; ++it) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (((a_inb >> it) & 1u) && chan_ok)
                v = ld_global16(xb + (a_goff[it] + extra + c0) * (int64_t)sizeof(elem_t));
            areg[it] = v;
        }
        if (has_tf && chan_ok) {
#pragma unroll
            for (int e = 0; e < C::EPC; ++e) {
                sc[e] = p.in_scale[c0 + e];
                sh[e] = p.in_shift[c0 + e];
            }
            relu = cmu_relu_on(c0, p.relu_from);
        }
#pragma unroll
        for (int it = 0; it < C::W_ITERS; ++it) {
            const int ws = C::C3 ? s : s * C::TAPS + w_t[it];
            u32x4 v = {0u, 0u, 0u, 0u};
            bool ok = (it * 256 + tid < C::W_CHUNKS) && (ws < p.nslices);
            if (C::C3) {
                // packed 3x3 weights use 32-byte K slices [slice32][tap][npad][32 B] (shared with the v2 kernel):
                // chunk cg of this 64-byte stage row lives in slice 2s + (cg >> 1), half (cg & 1)
                const int s32 = 2 * s + (cg >> 1);
                ok = ok && (s32 < p.nslices32);
                if (ok) v = ld_global16(wb + (((int64_t)s32 * 9 + w_t[it]) * p.npad + w_n[it]) * 32 + (cg & 1) * 16);
            } else if (ok) {
                const int64_t rowg = (int64_t)ws * p.npad + w_n[it];
                v = ld_global16(wb + rowg * 64 + cg * 16);
            }
            wreg[it] = v;
        }
    };

    auto store_stage = [&]() {
        if (C::C3 && use_buf) {
#pragma unroll
            for (int it = 0; it < C::A_ITERS; ++it) {
                u32x4 v = areg[it];
                if (has_tf) {
                    float f[C::EPC];
                    TR::unpack(v, f);
#pragma unroll
                    for (int e = 0; e < C::EPC; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), lo_relu);
                    v = TR::pack(f);
                    if (!(chan_ok && ((a_inb >> it) & 1u))) v = u32x4{0u, 0u, 0u, 0u};
                }
                if (it * 256 + tid < C::A_CHUNKS) *reinterpret_cast<u32x4*>(smem + a_lds[it]) = v;
            }
#pragma unroll
            for (int it = 0; it < C::W_ITERS; ++it)
                if (it * 256 + tid < C::W_CHUNKS) *reinterpret_cast<u32x4*>(smem + w_lds[it]) = wreg[it];
            return;
        }
#pragma unroll
        for (int it = 0; it < C::A_ITERS; ++it) {
            if (it * 256 + tid < C::A_CHUNKS) {
                u32x4 v = areg[it];
                if (has_tf && chan_ok && ((a_inb >> it) & 1u)) {
                    float f[C::EPC];
                    TR::unpack(v, f);
#pragma unroll
                    for (int e = 0; e < C::EPC; ++e) {
                        float t = fmaf(f[e], sc[e], sh[e]);
                        f[e] = relu ? fmaxf(t, 0.f) : t;
                    }
                    v = TR::pack(f);
                }
                *reinterpret_cast<u32x4*>(smem + a_lds[it]) = v;
            }
        }
#pragma unroll
        for (int it = 0; it < C::W_ITERS; ++it) {
            if (it * 256 + tid < C::W_CHUNKS) *reinterpret_cast<u32x4*>(smem + w_lds[it]) = wreg[it];
        }
    };

    // ---- accumulators and fragment base addresses -------------------------------------------------
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int a_base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_base[i] = ((4 * wave + 2 * i + pm_row(r)) * C::LW + pm_col(r)) * C::PS_A + h * 16;
    const int w_base = C::A_BYTES + r * C::PS_W + h * 16;

    const int nstages = C::C3 ? p.nslices : (p.nslices + C::TAPS - 1) / C::TAPS;

#ifdef CMU_IG_STAMPS
    const int stamp_slot = (blockIdx.x % 61 == 0 && blockIdx.x / 61 < 64) ? (int)(blockIdx.x / 61) : -1;
#endif
    load_stage(0);
    for (int s = 0; s < nstages; ++s) {
        IG_STAMP(0);
        __syncthreads();  // previous stage's fragment reads are done
        IG_STAMP(1);
        store_stage();
        IG_STAMP(2);
        __syncthreads();
        IG_STAMP(3);
        if (s + 1 < nstages) load_stage(s + 1);  // in flight during the MFMAs below
        IG_STAMP(4);
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) {
            const int aoff = C::C3 ? ((t / 3) * C::LW + (t % 3)) * C::PS_A : t * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(smem + a_base[0] + aoff + ks * 32);
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(smem + a_base[1] + aoff + ks * 32);
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(smem + w_base + (t * C::BN) * C::PS_W + ks * 32);
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(smem + w_base + (t * C::BN + 32) * C::PS_W + ks * 32);
                TR::mma16(a0, b0, acc[0][0]);
                TR::mma16(a0, b1, acc[0][1]);
                TR::mma16(a1, b0, acc[1][0]);
                TR::mma16(a1, b1, acc[1][1]);
            }
        }
        IG_STAMP(5);
    }
    __syncthreads();  // all waves done with the A/W images: LDS is reused by the epilogue

    // ---- epilogue -----------------------------------------------------------------------------------
    // accumulator element e of tile (i,j): pixel m32 = (e&3) + 8*(e>>2) + 4*h of M-block i, channel 32*j + r
    float* stat_lds = reinterpret_cast<float*>(smem + C::EPI_STAGE);
    if (MODE == MODE_CONV3 && p.stats != nullptr) {
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m32 = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int row = 4 * wave + 2 * i + pm_row(m32), col = pm_col(m32);
                const bool ok = (ty0 + row < p.H) && (tx0 + col < p.W);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float v = ok ? acc[i][j][e] : 0.f;
                    s1[j] += v;
                    s2[j] = fmaf(v, v, s2[j]);
                }
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s1[j] += __shfl_xor(s1[j], 32, 64);
            s2[j] += __shfl_xor(s2[j], 32, 64);
            if (h == 0) {
                stat_lds[(wave * C::BN + 32 * j + r) * 2 + 0] = s1[j];
                stat_lds[(wave * C::BN + 32 * j + r) * 2 + 1] = s2[j];
            }
        }
    }
    // (BN-backward statistics) the producer layer's raw-output chunks this lane will need: in flight before the
    // accumulators go through LDS (see conv_igemm3.inc)
    constexpr int CPRp = C::BN / C::EPC;
    u32x4 xr[CPRp];
    if ((MODE != MODE_CONVT_FWD) && p.bstats != nullptr) {
#pragma unroll
        for (int it = 0; it < CPRp; ++it) {
            const int idx = it * 64 + lane;
            const int m = idx / CPRp, q = idx % CPRp;
            const int row = 4 * wave + 2 * (m >> 5) + pm_row(m & 31), col = pm_col(m & 31);
            const int gy = ty0 + row < p.H ? ty0 + row : p.H - 1, gx = tx0 + col < p.W ? tx0 + col : p.W - 1;  // clamped
            int n = n0 + q * C::EPC;
            n = n < p.N ? n : 0;
            const int64_t xo = (((int64_t)b * p.H + gy) * p.W + gx) * p.ldbx + n;
            xr[it] = ld_global16(reinterpret_cast<const unsigned char*>(p.bx) + xo * (int64_t)sizeof(elem_t));
        }
    }
    // stage this wave's 64x64 sub-tile as [pixel m][channel n]
    unsigned char* my = smem + wave * (64 * C::EPI_ROW);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float badd = 0.f;
            if (MODE == MODE_CONVT_FWD) {
                const int n = n0 + 32 * j + r;
                if (n < p.N) badd = p.bias[n % p.Cq];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                *reinterpret_cast<elem_t*>(my + m * C::EPI_ROW + (32 * j + r) * (int)sizeof(elem_t)) =
                    TR::from_float(acc[i][j][e] + badd);
            }
        }
    __syncthreads();
    if (MODE == MODE_CONV3 && p.stats != nullptr && tid < C::BN) {
        const int n = n0 + tid;
        if (n < p.N) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                a += stat_lds[(w4 * C::BN + tid) * 2 + 0];
                q += stat_lds[(w4 * C::BN + tid) * 2 + 1];
            }
            p.stats[(tile * 2 + 0) * p.N + n] = a;
            p.stats[(tile * 2 + 1) * p.N + n] = q;
        }
    }
    // coalesced write-out: each lane moves 16-byte channel chunks of one pixel
    constexpr int CPR = C::BN / C::EPC;  // chunks per pixel row
    unsigned char* yb = reinterpret_cast<unsigned char*>(p.y);
    // optional BatchNorm+ReLU backward statistics of the layer this gradient flows into (see conv_igemm3.inc)
    const bool bs = (MODE != MODE_CONVT_FWD) && p.bstats != nullptr;
    constexpr int EPCc = C::EPC;
    float bsc[EPCc], bsh[EPCc], bmu[EPCc], bis[EPCc], bs1[EPCc], bs2[EPCc];
    const int nq = n0 + (lane % CPR) * EPCc;
    if (bs) {
#pragma unroll
        for (int e = 0; e < EPCc; ++e) {
            const int c = nq + e < p.N ? nq + e : 0;
            bsc[e] = p.b_scale[c]; bsh[e] = p.b_shift[c]; bmu[e] = p.b_mean[c]; bis[e] = p.b_invstd[c];
            bs1[e] = bs2[e] = 0.f;
        }
    }
#pragma unroll
    for (int it = 0; it < (64 * CPR) / 64; ++it) {
        const int idx = it * 64 + lane;
        const int m = idx / CPR, q = idx % CPR;
        const int row = 4 * wave + 2 * (m >> 5) + pm_row(m & 31), col = pm_col(m & 31);
        const int gy = ty0 + row, gx = tx0 + col;
        const int n = n0 + q * C::EPC;
        if (gy < p.H && gx < p.W && n < p.N) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(my + m * C::EPI_ROW + q * 16);
            int64_t off;
            if (MODE == MODE_CONVT_FWD) {
                const int ij = n / p.Cq, co = n % p.Cq;
                off = (((int64_t)b * (2 * p.H) + 2 * gy + (ij >> 1)) * (2 * p.W) + 2 * gx + (ij & 1)) * p.ldy + co;
            } else {
                off = (((int64_t)b * p.H + gy) * p.W + gx) * p.ldy + n;
            }
            st_global16(yb + off * (int64_t)sizeof(elem_t), v);
            if (bs) {
                float gv[EPCc], xv[EPCc];
                TR::unpack(v, gv);
                TR::unpack(xr[it], xv);
#pragma unroll
                for (int e = 0; e < EPCc; ++e) {
                    const float dz = fmaf(xv[e], bsc[e], bsh[e]) > 0.f ? gv[e] : 0.f;
                    bs1[e] += dz;
                    bs2[e] = fmaf(dz, xv[e], bs2[e]);   // sum(dz*x); xhat is applied per channel below
                }
            }
        }
    }
    if (bs) {
#pragma unroll
        for (int e = 0; e < EPCc; ++e) {
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) {
                bs1[e] += __shfl_xor(bs1[e], o, 64);
                bs2[e] += __shfl_xor(bs2[e], o, 64);
            }
            bs2[e] = (bs2[e] - bmu[e] * bs1[e]) * bis[e];   // sum(dz*(x-mean)*invstd) over this wave's <= 512 pixels
            if (lane < CPR) {
                stat_lds[(wave * C::BN + lane * EPCc + e) * 2 + 0] = bs1[e];
                stat_lds[(wave * C::BN + lane * EPCc + e) * 2 + 1] = bs2[e];
            }
        }
        __syncthreads();
        if (tid < C::BN && n0 + tid < p.N) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                a += stat_lds[(w4 * C::BN + tid) * 2 + 0];
                q += stat_lds[(w4 * C::BN + tid) * 2 + 1];
            }
            p.bstats[(tile * 2 + 0) * p.N + n0 + tid] = a;
            p.bstats[(tile * 2 + 1) * p.N + n0 + tid] = q;
        }
    }
}

#include "conv_igemm3.inc"
#include "conv_igemm3p.inc"
#include "conv_igemm5.inc"
#include "conv_igemm6.inc"
#include "conv_gemm.inc"
#include "conv_gather.inc"

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
template <class TR, int MODE>
static int launch_igemm(const IGParams& p, hipStream_t st, const char* name) {
    typedef IGCfg<TR, MODE> C;
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<TR, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("%s: hipFuncSetAttribute(%d B LDS): %s", name, C::LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL((conv_igemm_kernel<TR, MODE>), dim3((unsigned)p.total_blocks), dim3(256), C::LDS_BYTES, st, p);
    cmu_set_kernel_tag("conv_igemm_kernel");
    CMU_CHECK_LAUNCH(name);
    return CMU_OK;
}

static int check_act(const char* name, const void* ptr, int64_t ld, int C, int dt) {
    const int epc = 16 / cmu_dtype_size(dt);
    CMU_CHECK_ARG(ptr != nullptr, "%s: null tensor", name);
    CMU_CHECK_ARG(cmu_aligned16(ptr), "%s: tensor not 16-byte aligned", name);
    CMU_CHECK_ARG(ld >= C && ld % epc == 0, "%s: pixel stride %lld must be >= C=%d and a multiple of %d", name, (long long)ld, C, epc);
    CMU_CHECK_ARG(C % epc == 0, "%s: channels %d must be a multiple of %d", name, C, epc);
    return CMU_OK;
}

template <class TR>
static int conv3x3_fwd_t(IGParams p, hipStream_t st) {
    typedef IGCfg<TR, MODE_CONV3> C;
    // channel counts in whole 64-byte slices / 64-channel blocks: wide-tile kernel (conv_igemm3.inc); else the first kernel
    if (igemm3_eligible<TR>(p)) return launch_igemm3_any<TR>(p, st);
    p.nslices = cmu_div_up(p.K, C::KC);
    p.nslices32 = cmu_div_up(p.K, C::KC / 2);
    p.npad = cmu_conv3x3_npad(p.N);
    static const bool bufload = []() { const char* e = getenv("CMU_CONV_BUFLOAD"); return !(e && e[0] == '0'); }();
    p.buf_ok = bufload && ((int64_t)p.H * p.W * p.ldx + p.K) * (int64_t)sizeof(typename TR::elem_t) < 0x7fff0000ll &&
               (int64_t)p.nslices32 * 9 * p.npad * 32 < 0x7fff0000ll;
    return launch_igemm<TR, MODE_CONV3>(p, st, "cmu_conv3x3_fwd");
}
template <class TR>
static int convT_fwd_t(IGParams p, hipStream_t st) {
    typedef IGCfg<TR, MODE_CONVT_FWD> C;
    if (conv_gemm_s_eligible<TR>(p, false))
        return p.in_scale ? launch_conv_gemm_s<TR, false, true>(p, st, "cmu_convT2x2_fwd") : launch_conv_gemm_s<TR, false, false>(p, st, "cmu_convT2x2_fwd");
    if (conv_gemm_eligible<TR>(p, false))
        return p.in_scale ? launch_conv_gemm<TR, false, true>(p, st, "cmu_convT2x2_fwd") : launch_conv_gemm<TR, false, false>(p, st, "cmu_convT2x2_fwd");
    p.nslices = cmu_div_up(p.K, C::KC);
    return launch_igemm<TR, MODE_CONVT_FWD>(p, st, "cmu_convT2x2_fwd");
}
template <class TR>
static int convT_dgrad_t(IGParams p, hipStream_t st) {
    typedef IGCfg<TR, MODE_CONVT_DGRAD> C;
    {
        IGParams q = p;
        q.K = 4 * p.K;   // the GEMM's contraction runs over (sub-pixel position, output channel)
        if (conv_gemm_s_eligible<TR>(q, true)) return launch_conv_gemm_s<TR, true, false>(q, st, "cmu_convT2x2_dgrad");
        if (conv_gemm_eligible<TR>(q, true)) return launch_conv_gemm<TR, true, false>(q, st, "cmu_convT2x2_dgrad");
    }
    p.nslices = 4 * cmu_div_up(p.K, C::KC);
    return launch_igemm<TR, MODE_CONVT_DGRAD>(p, st, "cmu_convT2x2_dgrad");
}

static void fill_tiles(IGParams& p) {
    p.tilesX = cmu_div_up(p.W, CMU_TW);
    p.tilesY = cmu_div_up(p.H, CMU_TH);
    p.nblk = cmu_div_up(p.N, 64);
    p.npad = p.nblk * 64;
    p.total_blocks = (int64_t)p.B * p.tilesX * p.tilesY * p.nblk;
}

extern "C" int cmu_conv_ntiles(int B, int H, int W) { return B * cmu_div_up(H, CMU_TH) * cmu_div_up(W, CMU_TW); }

extern "C" int cmu_conv3x3_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                               const void* wpacked, void* y, int64_t ldy, float* stats, int B, int H, int W, int Cin,
                               int Cout, int dt, void* stream) {
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "cmu_conv3x3_fwd: bad dims");
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0, "cmu_conv3x3_fwd: bad dtype %d", dt);
    int rc;
    if ((rc = check_act("cmu_conv3x3_fwd(x)", x, ldx, Cin, dt))) return rc;
    if ((rc = check_act("cmu_conv3x3_fwd(y)", y, ldy, Cout, dt))) return rc;
    CMU_CHECK_ARG(wpacked && cmu_aligned16(wpacked), "cmu_conv3x3_fwd: packed weights null/unaligned");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_conv3x3_fwd: scale/shift must both be set");
    CMU_CHECK_ARG(relu_from % (16 / cmu_dtype_size(dt)) == 0, "cmu_conv3x3_fwd: relu_from=%d must be a multiple of the 16-byte chunk", relu_from);
    CMU_CHECK_ARG((int64_t)B * cmu_div_up(H, CMU_TH) * cmu_div_up(W, CMU_TW) * cmu_div_up(Cout, 64) < (1ll << 31),
                  "cmu_conv3x3_fwd: grid too large");
    IGParams p = {};
    p.x = x; p.ldx = ldx; p.in_scale = in_scale; p.in_shift = in_shift; p.relu_from = relu_from;
    p.w = wpacked; p.y = y; p.ldy = ldy; p.stats = stats; p.bias = nullptr;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout; p.Cq = Cout;
    fill_tiles(p);
    CMU_DISPATCH_DT(dt, conv3x3_fwd_t, p, (hipStream_t)stream);
}

extern "C" int cmu_convT2x2_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                                const void* wpacked, const float* bias, void* out, int64_t ldo, int B, int H, int W,
                                int Cin, int Cout, int dt, void* stream) {
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "cmu_convT2x2_fwd: bad dims");
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0, "cmu_convT2x2_fwd: bad dtype %d", dt);
    int rc;
    if ((rc = check_act("cmu_convT2x2_fwd(x)", x, ldx, Cin, dt))) return rc;
    if ((rc = check_act("cmu_convT2x2_fwd(out)", out, ldo, Cout, dt))) return rc;
    CMU_CHECK_ARG(wpacked && cmu_aligned16(wpacked) && bias, "cmu_convT2x2_fwd: weights/bias null or unaligned");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_convT2x2_fwd: scale/shift must both be set");
    CMU_CHECK_ARG(relu_from % (16 / cmu_dtype_size(dt)) == 0, "cmu_convT2x2_fwd: relu_from=%d must be a multiple of the 16-byte chunk", relu_from);
    IGParams p = {};
    p.x = x; p.ldx = ldx; p.in_scale = in_scale; p.in_shift = in_shift; p.relu_from = relu_from;
    p.w = wpacked; p.y = out; p.ldy = ldo; p.stats = nullptr; p.bias = bias;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = 4 * Cout; p.Cq = Cout;
    fill_tiles(p);
    CMU_DISPATCH_DT(dt, convT_fwd_t, p, (hipStream_t)stream);
}

extern "C" int cmu_convT2x2_dgrad(const void* dOut, int64_t ldd, const void* wpacked_dgrad, void* dX, int64_t ldx, int B,
                                  int H, int W, int Cin, int Cout, int dt, void* stream) {
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "cmu_convT2x2_dgrad: bad dims");
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0, "cmu_convT2x2_dgrad: bad dtype %d", dt);
    int rc;
    if ((rc = check_act("cmu_convT2x2_dgrad(dOut)", dOut, ldd, Cout, dt))) return rc;
    if ((rc = check_act("cmu_convT2x2_dgrad(dX)", dX, ldx, Cin, dt))) return rc;
    CMU_CHECK_ARG(wpacked_dgrad && cmu_aligned16(wpacked_dgrad), "cmu_convT2x2_dgrad: packed weights null/unaligned");
    IGParams p = {};
    p.x = dOut; p.ldx = ldd; p.in_scale = nullptr; p.in_shift = nullptr; p.relu_from = 0;
    p.w = wpacked_dgrad; p.y = dX; p.ldy = ldx; p.stats = nullptr; p.bias = nullptr;
    p.B = B; p.H = H; p.W = W; p.K = Cout; p.N = Cin; p.Cq = Cout;
    fill_tiles(p);
    CMU_DISPATCH_DT(dt, convT_dgrad_t, p, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// Sparse (SparK) form: the persistent kernel over a device-side list of spatial tiles
// ---------------------------------------------------------------------------------------------------
template <class TR>
static int conv3x3_tiles_ok_t(IGParams p) {
    return igemm3_eligible<TR>(p) && cmu_conv_persist_enabled() && p.H % 16 == 0 && p.W % 32 == 0 ? 1 : 0;
}
extern "C" int cmu_conv3x3_tiles_supported(int B, int H, int W, int Cin, int Cout, int dt) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || cmu_dtype_size(dt) <= 0) return 0;
    IGParams p = {};
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout; p.ldx = Cin; p.ldy = Cout;
    CMU_DISPATCH_DT(dt, conv3x3_tiles_ok_t, p);
}
extern "C" int cmu_conv3x3_fwd_tiles(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                                     const void* wpacked, void* y, int64_t ldy, const int* tile_list, const int* tile_count, int B, int H,
                                     int W, int Cin, int Cout, int dt, void* stream) {
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && tile_list && tile_count, "cmu_conv3x3_fwd_tiles: bad dims / null list");
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0, "cmu_conv3x3_fwd_tiles: bad dtype %d", dt);
    int rc;
    if ((rc = check_act("cmu_conv3x3_fwd_tiles(x)", x, ldx, Cin, dt))) return rc;
    if ((rc = check_act("cmu_conv3x3_fwd_tiles(y)", y, ldy, Cout, dt))) return rc;
    CMU_CHECK_ARG(wpacked && cmu_aligned16(wpacked), "cmu_conv3x3_fwd_tiles: packed weights null/unaligned");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_conv3x3_fwd_tiles: scale/shift must both be set");
    CMU_CHECK_ARG(relu_from % (16 / cmu_dtype_size(dt)) == 0, "cmu_conv3x3_fwd_tiles: relu_from=%d must be a multiple of the 16-byte chunk", relu_from);
    IGParams p = {};
    p.x = x; p.ldx = ldx; p.in_scale = in_scale; p.in_shift = in_shift; p.relu_from = relu_from;
    p.w = wpacked; p.y = y; p.ldy = ldy; p.stats = nullptr; p.bias = nullptr;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout; p.Cq = Cout;
    p.tile_list = tile_list; p.tile_count = tile_count;
    fill_tiles(p);
    if (!cmu_conv3x3_tiles_supported(B, H, W, Cin, Cout, dt)) {
        cmu_set_error("cmu_conv3x3_fwd_tiles: shape (H=%d W=%d Cin=%d Cout=%d) is not served by the persistent kernel "
                      "(whole 16 x 32 tiles, whole channel blocks): call cmu_conv3x3_fwd", H, W, Cin, Cout);
        return CMU_ERR_UNSUPPORTED;
    }
    CMU_DISPATCH_DT(dt, launch_igemm3_any, p, (hipStream_t)stream);
}

__global__ void sparse_tile_list_kernel(const uint8_t* __restrict__ active, int f, int sbits, int B, int tilesY, int tilesX, int th, int tw,
                                        int* __restrict__ list, int* __restrict__ count);

// ---- gather form: the convolution over a list of active pixels (conv_gather.inc) ---------------------------------------------------
template <class TR>
static int conv3x3_rows_ok_t(IGParams p) {
    constexpr int KSC = 128 / (int)sizeof(typename TR::elem_t);
    static const bool on = []() { const char* e = getenv("CMU_SPARK_GATHER"); return !(e && e[0] == '0'); }();
    if (!on || p.N % 128 != 0 || p.K % KSC != 0) return 0;
    const int64_t px = (int64_t)p.B * p.H * p.W;
    if ((px * p.ldx + p.K) * (int64_t)sizeof(typename TR::elem_t) >= 0x7fff0000ll) return 0;          // 32-bit buffer offsets over the whole tensor
    if ((int64_t)(p.K / (32 / (int)sizeof(typename TR::elem_t))) * 9 * cmu_conv3x3_npad(p.N) * 32 >= 0x7fff0000ll) return 0;
    return 1;
}
extern "C" int cmu_conv3x3_rows_supported(int B, int H, int W, int Cin, int Cout, int dt) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || cmu_dtype_size(dt) <= 0) return 0;
    IGParams p = {};
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout; p.ldx = Cin; p.ldy = Cout;
    CMU_DISPATCH_DT(dt, conv3x3_rows_ok_t, p);
}
extern "C" int cmu_conv3x3_fwd_rows(const void* x, int64_t ldx, const void* wpacked, void* y, int64_t ldy, const int* rows, const int* n_rows,
                                    int64_t max_rows, int B, int H, int W, int Cin, int Cout, int dt, void* stream) {
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && rows && n_rows && max_rows > 0, "cmu_conv3x3_fwd_rows: bad dims / null list");
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0, "cmu_conv3x3_fwd_rows: bad dtype %d", dt);
    int rc;
    if ((rc = check_act("cmu_conv3x3_fwd_rows(x)", x, ldx, Cin, dt))) return rc;
    if ((rc = check_act("cmu_conv3x3_fwd_rows(y)", y, ldy, Cout, dt))) return rc;
    CMU_CHECK_ARG(wpacked && cmu_aligned16(wpacked), "cmu_conv3x3_fwd_rows: packed weights null/unaligned");
    IGParams p = {};
    p.x = x; p.ldx = ldx; p.w = wpacked; p.y = y; p.ldy = ldy;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout; p.Cq = Cout;
    p.tile_list = rows; p.tile_count = n_rows;
    const int64_t px = (int64_t)B * H * W;
    if (!cmu_conv3x3_rows_supported(B, H, W, Cin, Cout, dt) || (px * ldx + Cin) * cmu_dtype_size(dt) >= 0x7fff0000ll ||
        (px * ldy + Cout) * cmu_dtype_size(dt) >= (1ll << 40)) {
        cmu_set_error("cmu_conv3x3_fwd_rows: needs Cout %% 128 == 0, Cin a whole number of 128-byte steps and an input tensor below 2 GiB "
                      "(Cin=%d Cout=%d): call cmu_conv3x3_fwd / cmu_conv3x3_fwd_tiles", Cin, Cout);
        return CMU_ERR_UNSUPPORTED;
    }
    CMU_CHECK_ARG(cmu_div_up64(max_rows, 256) * (Cout / 128) < (1ll << 31), "cmu_conv3x3_fwd_rows: grid too large");
    CMU_DISPATCH_DT(dt, launch_conv_gather, p, max_rows, (hipStream_t)stream);
}

// rows[r] = dense pixel index (b*H + y)*W + x of the r-th active pixel, patch-major (patch order = cmu_sparse_tile_list with tiles
// of one patch; pixels row-major inside a patch); entries past the end up to `capacity` are -1; count[0] = number of rows
__global__ void sparse_pixel_rows_kernel(const int* __restrict__ plist, const int* __restrict__ pcount, int f, int s, int H, int W,
                                         int* __restrict__ rows, int64_t capacity, int* __restrict__ count) {
    const int np = pcount[0];
    const int64_t n = (int64_t)np * s * s;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < capacity; r += (int64_t)gridDim.x * blockDim.x) {
        int v = -1;
        if (r < n) {
            const int pi = (int)(r / (s * s)), in = (int)(r % (s * s));
            const int t = plist[pi];                                   // (b*f + fy)*f + fx
            const int fx = t % f, fy = (t / f) % f, b = t / (f * f);
            v = (b * H + fy * s + in / s) * W + fx * s + in % s;
        }
        rows[r] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) count[0] = (int)n;
}
extern "C" int64_t cmu_sparse_pixel_list_ws_bytes(int B, int f) { return ((int64_t)B * f * f + 16) * (int64_t)sizeof(int); }
extern "C" int cmu_sparse_pixel_list(const uint8_t* active, int f, int B, int H, int W, int* rows, int64_t capacity, int* count, void* ws,
                                     void* stream) {
    CMU_CHECK_ARG(active && rows && count && ws && f > 0 && B > 0 && H > 0 && W == H && capacity > 0, "cmu_sparse_pixel_list: bad args");
    const int sbits = sp_shift_bits(H, f);
    CMU_CHECK_ARG(sbits >= 0 && (int64_t)B * H * W < (1ll << 31), "cmu_sparse_pixel_list: H must be f << s (H=%d, f=%d)", H, f);
    const int s = 1 << sbits;
    int* plist = (int*)ws + 16;
    int* pcount = (int*)ws;
    hipLaunchKernelGGL(sparse_tile_list_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, active, f, sbits, B, f, f, s, s, plist, pcount);
    CMU_CHECK_LAUNCH("cmu_sparse_pixel_list(patches)");
    const int grid = (int)(cmu_div_up64(capacity, 256) < 4096 ? cmu_div_up64(capacity, 256) : 4096);
    hipLaunchKernelGGL(sparse_pixel_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, plist, pcount, f, s, H, W, rows, capacity, count);
    CMU_CHECK_LAUNCH("cmu_sparse_pixel_list(rows)");
    return CMU_OK;
}

// Active-tile list of a (B, H, W) level for the patch map `active` (B, f, f): tile (b, ty, tx) of th x tw pixels is listed
// iff one of the patches it overlaps is active.  One workgroup, ascending tile order (deterministic), count in count[0].
// Round 4: two phases per trip of up to 65,536 tiles.  (1) The 1,024 threads stride over the tiles -- independent, coalesced mask
// loads, many in flight -- and each wave's ballot goes into a bitmap in LDS; (2) every thread takes a run of up to 64 CONSECUTIVE
// tiles out of the bitmap (one 64-bit mask), the workgroup scans the runs' counts once (wave prefix by shuffles + the 16 wave
// totals) and each thread writes its listed tiles behind its offset.  Five barriers per trip where the tile-per-thread loop took
// three per 1,024 tiles with one dependent mask load each (45 us per list, nine lists per SparK step).  Same ascending order.
__device__ static inline void sparse_tile_list_body(const uint8_t* __restrict__ active, int f, int sbits, int B, int tilesY, int tilesX, int th,
                                                   int tw, int* __restrict__ list, int* __restrict__ count, unsigned long long* bits, int* wsum,
                                                   int& base) {
    const int total = B * tilesY * tilesX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int want = total / 1024 + (total % 1024 != 0);
    const int per = want < 64 ? want : 64;                     // tiles per thread and trip
    for (int t0 = 0; t0 < total; t0 += 1024 * per) {
        for (int b0 = 0; b0 < 1024 * per; b0 += 1024) {
            const int t = t0 + b0 + (int)threadIdx.x;
            bool on = false;
            if (t < total) {
                const int tx = t % tilesX, ty = (t / tilesX) % tilesY, b = t / (tilesX * tilesY);
                const int py0 = (ty * th) >> sbits, py1 = (ty * th + th - 1) >> sbits;
                const int px0 = (tx * tw) >> sbits, px1 = (tx * tw + tw - 1) >> sbits;
                for (int py = py0; py <= py1 && py < f; ++py)
                    for (int px = px0; px <= px1 && px < f; ++px) on |= active[((int64_t)b * f + py) * f + px] != 0;
            }
            const unsigned long long m64 = __ballot(on);
            if (lane == 0) bits[(b0 >> 6) + wave] = m64;
        }
        __syncthreads();
        const int start = (int)threadIdx.x * per, w0 = start >> 6, sh = start & 63;
        unsigned long long m = bits[w0] >> sh;
        if (sh != 0 && w0 + 1 < 1024) m |= bits[w0 + 1] << (64 - sh);
        if (per < 64) m &= (1ull << per) - 1ull;
        const int first = t0 + start;
        const int mine = __popcll(m);
        int incl = mine;                                   // inclusive prefix over the wave (fixed shuffle order)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = base + incl - mine;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        for (int k = 0; k < per; ++k)
            if ((m >> k) & 1ull) list[off++] = first + k;
        __syncthreads();
        if (threadIdx.x == 0) {
            int sum = 0;
            for (int w = 0; w < 16; ++w) sum += wsum[w];
            base += sum;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = base;
}
__global__ __launch_bounds__(1024) void sparse_tile_list_kernel(const uint8_t* __restrict__ active, int f, int sbits, int B, int tilesY,
                                                               int tilesX, int th, int tw, int* __restrict__ list, int* __restrict__ count) {
    __shared__ unsigned long long bits[1024];
    __shared__ int wsum[16];
    __shared__ int base;
    sparse_tile_list_body(active, f, sbits, B, tilesY, tilesX, th, tw, list, count, bits, wsum, base);
}
// every list of a step in ONE launch, one workgroup per list (round 4: nine single-workgroup launches of 11 ... 72 us sat on a SparK
// step's critical path -- the mask is known before the step starts); same lists as cmu_sparse_tile_list, element for element.
constexpr int SP_MAX_LISTS = 12;
struct SpListBatch {
    int f, B;
    int sbits[SP_MAX_LISTS], tilesY[SP_MAX_LISTS], tilesX[SP_MAX_LISTS], th[SP_MAX_LISTS], tw[SP_MAX_LISTS];
    int* list[SP_MAX_LISTS];
    int* count[SP_MAX_LISTS];
};
__global__ __launch_bounds__(1024) void sparse_tile_lists_kernel(const uint8_t* __restrict__ active, SpListBatch q) {
    __shared__ unsigned long long bits[1024];
    __shared__ int wsum[16];
    __shared__ int base;
    const int i = blockIdx.x;
    sparse_tile_list_body(active, q.f, q.sbits[i], q.B, q.tilesY[i], q.tilesX[i], q.th[i], q.tw[i], q.list[i], q.count[i], bits, wsum, base);
}
extern "C" int cmu_sparse_tile_lists_max(void) { return SP_MAX_LISTS; }
extern "C" int cmu_sparse_tile_lists(const uint8_t* active, int f, int B, int n, const int* H, const int* tile_h, const int* tile_w, int* const* lists,
                                     int* const* counts, void* stream) {
    CMU_CHECK_ARG(active && f > 0 && B > 0 && n > 0 && n <= SP_MAX_LISTS && H && tile_h && tile_w && lists && counts,
                  "cmu_sparse_tile_lists: bad args (at most %d lists per call)", SP_MAX_LISTS);
    SpListBatch q = {};
    q.f = f; q.B = B;
    for (int i = 0; i < n; ++i) {
        CMU_CHECK_ARG(lists[i] && counts[i] && H[i] > 0 && tile_h[i] > 0 && tile_w[i] > 0, "cmu_sparse_tile_lists: null list / bad tile (entry %d)", i);
        const int sb = sp_shift_bits(H[i], f);
        CMU_CHECK_ARG(sb >= 0, "cmu_sparse_tile_lists: H must be f << s (entry %d: H=%d, f=%d)", i, H[i], f);
        q.sbits[i] = sb;
        q.tilesY[i] = cmu_div_up(H[i], tile_h[i]);
        q.tilesX[i] = cmu_div_up(H[i], tile_w[i]);
        q.th[i] = tile_h[i]; q.tw[i] = tile_w[i];
        CMU_CHECK_ARG((int64_t)B * q.tilesY[i] * q.tilesX[i] < (1ll << 30), "cmu_sparse_tile_lists: too many tiles (entry %d)", i);
        q.list[i] = lists[i]; q.count[i] = counts[i];
    }
    hipLaunchKernelGGL(sparse_tile_lists_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, active, q);
    CMU_CHECK_LAUNCH("cmu_sparse_tile_lists");
    return CMU_OK;
}
// the pixel lists of several levels from ONE patch list (cmu_sparse_tile_list with tiles of one patch, e.g. H = f and 1 x 1 tiles:
// entries (b*f + fy)*f + fx) in one launch: level i (side H[i] = f << s) gets rows[i][0 .. capacity[i]) and counts[i][0], exactly
// as cmu_sparse_pixel_list writes them
struct SpRowsBatch {
    int f;
    int s[SP_MAX_LISTS], H[SP_MAX_LISTS];
    int* rows[SP_MAX_LISTS];
    int* count[SP_MAX_LISTS];
    long long cap[SP_MAX_LISTS];
};
__global__ void sparse_pixel_rows_multi_kernel(const int* __restrict__ plist, const int* __restrict__ pcount, SpRowsBatch q) {
    const int i = blockIdx.y;
    const int np = pcount[0], s = q.s[i], H = q.H[i], f = q.f;
    const long long n = (long long)np * s * s, cap = q.cap[i];
    int* __restrict__ rows = q.rows[i];
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < cap; r += (long long)gridDim.x * blockDim.x) {
        int v = -1;
        if (r < n) {
            const int pi = (int)(r / (s * s)), in = (int)(r % (s * s));
            const int t = plist[pi];
            const int fx = t % f, fy = (t / f) % f, b = t / (f * f);
            v = (b * H + fy * s + in / s) * H + fx * s + in % s;
        }
        rows[r] = v;
    }
    // (never more rows than the list holds: a host-side count that overstates the capacity must not send the list-driven kernels past it)
    if (blockIdx.x == 0 && threadIdx.x == 0) q.count[i][0] = (int)(n < cap ? n : cap);
}
extern "C" int cmu_sparse_pixel_lists(const int* patches, const int* patch_count, int f, int B, int n, const int* H, int* const* rows,
                                      const int64_t* capacity, int* const* counts, void* stream) {
    CMU_CHECK_ARG(patches && patch_count && f > 0 && B > 0 && n > 0 && n <= SP_MAX_LISTS && H && rows && capacity && counts,
                  "cmu_sparse_pixel_lists: bad args (at most %d levels per call)", SP_MAX_LISTS);
    SpRowsBatch q = {};
    q.f = f;
    int64_t cmax = 0;
    for (int i = 0; i < n; ++i) {
        const int sb = sp_shift_bits(H[i], f);
        CMU_CHECK_ARG(rows[i] && counts[i] && capacity[i] > 0 && sb >= 0 && (int64_t)B * H[i] * H[i] < (1ll << 31),
                      "cmu_sparse_pixel_lists: null list, or H must be f << s (entry %d: H=%d, f=%d)", i, H[i], f);
        q.s[i] = 1 << sb; q.H[i] = H[i]; q.rows[i] = rows[i]; q.count[i] = counts[i]; q.cap[i] = capacity[i];
        if (capacity[i] > cmax) cmax = capacity[i];
    }
    const int grid = (int)(cmu_div_up64(cmax, 256) < 4096 ? cmu_div_up64(cmax, 256) : 4096);
    hipLaunchKernelGGL(sparse_pixel_rows_multi_kernel, dim3(grid, n), dim3(256), 0, (hipStream_t)stream, patches, patch_count, q);
    CMU_CHECK_LAUNCH("cmu_sparse_pixel_lists");
    return CMU_OK;
}
extern "C" int cmu_sparse_tile_list(const uint8_t* active, int f, int B, int H, int W, int tile_h, int tile_w, int* list, int* count,
                                    void* stream) {
    CMU_CHECK_ARG(active && list && count && f > 0 && B > 0 && H > 0 && W > 0 && tile_h > 0 && tile_w > 0, "cmu_sparse_tile_list: bad args");
    const int sbits = sp_shift_bits(H, f);
    CMU_CHECK_ARG(sbits >= 0 && W == H, "cmu_sparse_tile_list: H must be f << s and the level square (H=%d, W=%d, f=%d)", H, W, f);
    const int tilesY = cmu_div_up(H, tile_h), tilesX = cmu_div_up(W, tile_w);
    CMU_CHECK_ARG((int64_t)B * tilesY * tilesX < (1ll << 30), "cmu_sparse_tile_list: too many tiles");
    hipLaunchKernelGGL(sparse_tile_list_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, active, f, sbits, B, tilesY, tilesX, tile_h, tile_w,
                       list, count);
    CMU_CHECK_LAUNCH("cmu_sparse_tile_list");
    return CMU_OK;
}

// ---------------------------------------------------------------------------------------------------
// data-gradient entries that also produce the BatchNorm+ReLU backward statistics of the layer the gradient flows into
// ---------------------------------------------------------------------------------------------------
static int check_bnstats(const char* name, const void* yraw, int64_t ldy, const float* scale, const float* shift, const float* mean,
                         const float* invstd, const float* bstats, int N, int dt) {
    int rc;
    if ((rc = check_act(name, yraw, ldy, N, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && mean && invstd && bstats, "%s: null BatchNorm argument", name);
    return CMU_OK;
}

extern "C" int cmu_conv3x3_dgrad_bn(const void* dY, int64_t ldd, const void* wpacked_flip, void* dX, int64_t ldx, const void* yraw,
                                    int64_t ldy, const float* scale, const float* shift, const float* save_mean,
                                    const float* save_invstd, float* bstats, int B, int H, int W, int K, int N, int dt, void* stream) {
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && K > 0 && N > 0, "cmu_conv3x3_dgrad_bn: bad dims");
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0, "cmu_conv3x3_dgrad_bn: bad dtype %d", dt);
    int rc;
    if ((rc = check_act("cmu_conv3x3_dgrad_bn(dY)", dY, ldd, K, dt))) return rc;
    if ((rc = check_act("cmu_conv3x3_dgrad_bn(dX)", dX, ldx, N, dt))) return rc;
    if ((rc = check_bnstats("cmu_conv3x3_dgrad_bn(yraw)", yraw, ldy, scale, shift, save_mean, save_invstd, bstats, N, dt))) return rc;
    CMU_CHECK_ARG(wpacked_flip && cmu_aligned16(wpacked_flip), "cmu_conv3x3_dgrad_bn: packed weights null/unaligned");
    CMU_CHECK_ARG((int64_t)B * cmu_div_up(H, CMU_TH) * cmu_div_up(W, CMU_TW) * cmu_div_up(N, 64) < (1ll << 31), "cmu_conv3x3_dgrad_bn: grid too large");
    IGParams p = {};
    p.x = dY; p.ldx = ldd; p.w = wpacked_flip; p.y = dX; p.ldy = ldx;
    p.B = B; p.H = H; p.W = W; p.K = K; p.N = N; p.Cq = N;
    p.bx = yraw; p.ldbx = ldy; p.b_scale = scale; p.b_shift = shift; p.b_mean = save_mean; p.b_invstd = save_invstd; p.bstats = bstats;
    fill_tiles(p);
    CMU_DISPATCH_DT(dt, conv3x3_fwd_t, p, (hipStream_t)stream);
}

extern "C" int cmu_convT2x2_dgrad_bn(const void* dOut, int64_t ldd, const void* wpacked_dgrad, void* dX, int64_t ldx, const void* yraw,
                                     int64_t ldy, const float* scale, const float* shift, const float* save_mean,
                                     const float* save_invstd, float* bstats, int B, int H, int W, int Cin, int Cout, int dt,
                                     void* stream) {
    CMU_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "cmu_convT2x2_dgrad_bn: bad dims");
    CMU_CHECK_ARG(cmu_dtype_size(dt) > 0, "cmu_convT2x2_dgrad_bn: bad dtype %d", dt);
    int rc;
    if ((rc = check_act("cmu_convT2x2_dgrad_bn(dOut)", dOut, ldd, Cout, dt))) return rc;
    if ((rc = check_act("cmu_convT2x2_dgrad_bn(dX)", dX, ldx, Cin, dt))) return rc;
    if ((rc = check_bnstats("cmu_convT2x2_dgrad_bn(yraw)", yraw, ldy, scale, shift, save_mean, save_invstd, bstats, Cin, dt))) return rc;
    CMU_CHECK_ARG(wpacked_dgrad && cmu_aligned16(wpacked_dgrad), "cmu_convT2x2_dgrad_bn: packed weights null/unaligned");
    IGParams p = {};
    p.x = dOut; p.ldx = ldd; p.w = wpacked_dgrad; p.y = dX; p.ldy = ldx;
    p.B = B; p.H = H; p.W = W; p.K = Cout; p.N = Cin; p.Cq = Cout;
    p.bx = yraw; p.ldbx = ldy; p.b_scale = scale; p.b_shift = shift; p.b_mean = save_mean; p.b_invstd = save_invstd; p.bstats = bstats;
    fill_tiles(p);
    CMU_DISPATCH_DT(dt, convT_dgrad_t, p, (hipStream_t)stream);
}
