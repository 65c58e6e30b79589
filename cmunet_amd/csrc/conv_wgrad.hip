// conv_wgrad.hip -- weight gradients of Conv2d 3x3 and ConvTranspose2d 2x2 on MFMA (gfx950).
//
//   MODE_W3:  dW[n][c][kh][kw] = sum_p dY[p][n] * act(X)[p + (kh-1,kw-1)][c]        (autograd of model.py:17,20)
//   MODE_WT:  dW[c][n][i][j]   = sum_p dOut[2p+(i,j)][n] * act(X)[p][c]              (autograd of model.py:60)
//
// GEMM view: M = output channels n (rows, operand A = dY^T), N = input channels c (cols, operand B = X),
// K = pixels.  Both operands are pixel-major NHWC in HBM, i.e. K is the SLOW axis of both -- the opposite of
// what an MFMA fragment wants (8 consecutive k per lane).  CDNA4 answer: keep the natural [pixel][channel]
// image in LDS and read fragments with ds_read_b64_tr_b16 (hardware 4x16 transpose), two reads per
// fragment; for fp32 the 32x32x2 MFMA takes one k per lane-half, so plain ds_read_b32 is already right.
//
//   * workgroup = 512 threads = 8 wave64, one per CU; block tile = 128-byte channel rows on both sides
//     (64x64 channels for f16/bf16, 32x32 for f32) x all 9 taps: each wave keeps 9 accumulator tiles
//     (144 VGPRs) for its (32n x 32c) block and half (1/8 for f32) of the tile's pixel rows; the A
//     fragment of a pixel row is loaded once and reused by the 9 taps.
//   * K loop = 16x16 spatial tiles (split-K over workgroups); per tile the dY tile and the 18x18 X halo are
//     staged ONCE (global -> regs -> LDS, BatchNorm+ReLU of the producer applied to X on the way) and the
//     9 taps read the halo at shifted pixel addresses.
//   * LDS image: 128-byte pixel rows, 64-byte halves XOR-swizzled by bit 1 of the pixel index: the four
//     pixels a transposed read touches land on the four 64-byte quarters of the 256-byte bank row.
//   * accumulators of the k-parts are combined through LDS, partial slabs [split][tap][n][c] go to a
//     workspace and a second kernel sums them in split order (deterministic) into the reference layout.
// (Its eight waves multiply, then stage, in lockstep.  Running waves 4-7 one stage phase ahead -- the two-phase ping-pong of
// conv_wgrad2.inc -- on THIS loop measured 11.9 against 10.2 ms of weight-gradient time per bench step: alone on its SIMD a
// wave waits for every transposed fragment read, which two waves multiplying side by side hide for each other; the
// ping-pong needs the register double buffer of fragments that conv_wgrad2.inc has and the 144 accumulator registers
// here leave no room for.  The Cout = 64 layers that still run here are 64 x 64 blocks: 7.2 MFMAs per staged chunk
// whatever the schedule, against 10.3 for the 128 x 64 blocks of conv_wgrad2.inc.)
#include "common.h"
#include <stdlib.h>

enum { MODE_W3 = 0, MODE_WT = 1 };

struct WGParams {
    const void* a;  // dY (MODE_W3) / dOut (MODE_WT)
    int64_t lda;
    const void* b;  // X
    int64_t ldb;
    const float* b_scale;
    const float* b_shift;
    int relu_from;
    float* ws;
    int B, H, W;  // pixel grid of the contraction (MODE_WT: the low-res grid)
    int CA, CB;   // channels of A (output channels) and B (input channels)
    int CApad, CBpad;
    int nAB, nBB, splitk, ntiles, tilesX, tilesY;
    // sparse (SparK) form of the first kernel: the K loop runs over the 16 x 16 pixel tiles tile_list[0 .. *tile_count) only
    // (device arrays from cmu_sparse_tile_list; elsewhere dY is zero by construction, so the sum is unchanged)
    const int* tile_list;
    const int* tile_count;
};

template <class TR, int MODE>
struct WGCfg {
    typedef typename TR::elem_t elem_t;
    static constexpr int ES = (int)sizeof(elem_t);
    static constexpr int CW = 128 / ES;
    static constexpr int NB32 = CW / 32;
    static constexpr int NT = NB32 * NB32;
    static constexpr int KPARTS = 8 / NT;
    static constexpr int RPP = 16 / KPARTS;  // pixel rows per k-part
    static constexpr int TAPS = (MODE == MODE_W3) ? 9 : 1;
    static constexpr int HALO = (MODE == MODE_W3) ? 1 : 0;
    static constexpr int LW = 16 + 2 * HALO;
    static constexpr int LWP = (MODE == MODE_W3) ? 20 : 16;  // LDS row pitch in pixels: a multiple of 4, so that the
                                                             // swizzle bit of a fragment read is a per-lane constant
    static constexpr int NPIXB = LW * LW;
    static constexpr int A_BYTES = 256 * 128;
    static constexpr int B_BYTES = LW * LWP * 128;
    static constexpr int A_ITERS = (256 * 8) / 512;
    static constexpr int B_ITERS = (NPIXB * 8 + 511) / 512;
    static constexpr int RED_BYTES = NT * TAPS * 4096;
    static constexpr int BUF_BYTES = A_BYTES + B_BYTES;   // one stage: dY tile + X halo
    static constexpr int MAIN_BYTES = 2 * BUF_BYTES;      // double-buffered: stage t+1 is written while t is read
    static constexpr int LDS_BYTES = MAIN_BYTES > RED_BYTES ? MAIN_BYTES : RED_BYTES;
};

__device__ static inline int swz(int pix, int cbyte) { return pix * 128 + (cbyte ^ (((pix >> 1) & 1) << 6)); }

__device__ static inline u32x2 lds_tr16(const unsigned char* smem, int off) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + off));
    return __builtin_bit_cast(u32x2, v);
}

#ifdef CMU_IG_STAMPS
__device__ unsigned long long g_wg_stamps[64 * 16 * 8];
extern "C" int cmu_debug_wg_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), sizeof(g_wg_stamps));
}
#define WG_STAMP(k_)                                                                          \
    do {                                                                                       \
        if (stamp_slot >= 0 && stamp_i < 16 && tid == 0)                                       \
            g_wg_stamps[(stamp_slot * 16 + stamp_i) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define WG_STAMP(k_) do {} while (0)
#endif

template <class TR, int MODE>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const WGParams p) {
    typedef WGCfg<TR, MODE> C;
    constexpr int EPC = TR::EPC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* smA = smem;                 // buffer being read (swapped every tile)
    unsigned char* smB = smem + C::A_BYTES;
    unsigned char* stA = smem;                 // buffer being written by store_tile
    unsigned char* stB = smem + C::A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;

    int bid = blockIdx.x;
    const int ab = bid % p.nAB;
    bid /= p.nAB;
    const int bb = bid % p.nBB;
    bid /= p.nBB;
    int ij = 0;
    if (MODE == MODE_WT) { ij = bid & 3; bid >>= 2; }
    const int split = bid;

    const int tile_id = wave % C::NT;
    const int kpart = wave / C::NT;
    const int nb32 = tile_id % C::NB32, cb32 = tile_id / C::NB32;

    const int Ha = (MODE == MODE_WT) ? 2 * p.H : p.H;
    const int Wa = (MODE == MODE_WT) ? 2 * p.W : p.W;
    const unsigned char* ga = reinterpret_cast<const unsigned char*>(p.a);
    const unsigned char* gb = reinterpret_cast<const unsigned char*>(p.b);

    // staging role: 16-byte chunk cgi of the 128-byte channel row, constant per thread
    const int cgi = tid & 7;
    const int a_c0 = ab * C::CW + cgi * EPC;
    const int b_c0 = bb * C::CW + cgi * EPC;
    const bool a_cok = a_c0 < p.CA;
    const bool b_cok = b_c0 < p.CB;
    const bool has_tf = (p.b_scale != nullptr);
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = (has_tf && b_cok) ? p.b_scale[b_c0 + e] : 1.f;
        sh[e] = (has_tf && b_cok) ? p.b_shift[b_c0 + e] : 0.f;
    }
    const bool relu = has_tf && cmu_relu_on(b_c0, p.relu_from);

    f32x16 acc[C::TAPS];
#pragma unroll
    for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    u32x4 areg[C::A_ITERS];
    u32x4 breg[C::B_ITERS];
    unsigned b_ok = 0;  // bit it: B chunk is inside the image (so the transform applies)

    const int ntiles = (MODE == MODE_W3 && p.tile_list != nullptr) ? __builtin_amdgcn_readfirstlane(*p.tile_count) : p.ntiles;
    auto load_tile = [&](int tile) {
        if (MODE == MODE_W3 && p.tile_list != nullptr) tile = __builtin_amdgcn_readfirstlane(p.tile_list[tile]);
        const int tx = tile % p.tilesX, ty = (tile / p.tilesX) % p.tilesY, b = tile / (p.tilesX * p.tilesY);
        const int ty0 = ty * 16, tx0 = tx * 16;
#pragma unroll
        for (int it = 0; it < C::A_ITERS; ++it) {
            const int pix = (it * 512 + tid) >> 3;
            const int py = pix >> 4, px = pix & 15;
            int gy = ty0 + py, gx = tx0 + px;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (a_cok && gy < p.H && gx < p.W) {
                if (MODE == MODE_WT) { gy = 2 * gy + (ij >> 1); gx = 2 * gx + (ij & 1); }
                v = ld_global16(ga + ((((int64_t)b * Ha + gy) * Wa + gx) * p.lda + a_c0) * C::ES);
            }
            areg[it] = v;
        }
        b_ok = 0;
#pragma unroll
        for (int it = 0; it < C::B_ITERS; ++it) {
            const int idx = it * 512 + tid;
            const int pix = idx >> 3;
            const int py = pix / C::LW, px = pix % C::LW;
            const int gy = ty0 - C::HALO + py, gx = tx0 - C::HALO + px;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < C::NPIXB * 8 && b_cok && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                v = ld_global16(gb + ((((int64_t)b * p.H + gy) * p.W + gx) * p.ldb + b_c0) * C::ES);
                b_ok |= 1u << it;
            }
            breg[it] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < C::A_ITERS; ++it) {
            const int pix = (it * 512 + tid) >> 3;
            *reinterpret_cast<u32x4*>(stA + swz(pix, cgi * 16)) = areg[it];
        }
#pragma unroll
        for (int it = 0; it < C::B_ITERS; ++it) {
            const int idx = it * 512 + tid;
            if (idx < C::NPIXB * 8) {
                u32x4 v = breg[it];
                if (has_tf && ((b_ok >> it) & 1u)) {
                    float f[EPC];
                    TR::unpack(v, f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float t = fmaf(f[e], sc[e], sh[e]);
                        f[e] = relu ? fmaxf(t, 0.f) : t;
                    }
                    v = TR::pack(f);
                }
                const int pix = idx >> 3;
                *reinterpret_cast<u32x4*>(stB + swz((pix / C::LW) * C::LWP + pix % C::LW, cgi * 16)) = v;
            }
        }
    };

    // fragment addressing.  16-bit path: 16-lane group g = lane>>4 covers channels 16*(g&1).. of the wave's
    // 32-block and pixels 8*(g>>1) + 4*t + q of the row; lane li = lane&15: q = li>>2, quad = li&3.
    // Row pitches (16 / 20 pixels) are multiples of 4, so bit 1 of the pixel index -- the swizzle bit -- depends
    // only on (kw + q): every read is base(py,kh) + per-lane constant + immediate.
    const int li = lane & 15, g = lane >> 4;
    const int q4 = li >> 2, quad = li & 3;
    const int a_cbyte16 = (nb32 * 32 + 16 * (g & 1) + 4 * quad) * 2;
    const int b_cbyte16 = (cb32 * 32 + 16 * (g & 1) + 4 * quad) * 2;
    const int a_off16 = (8 * h + q4) * 128 + (a_cbyte16 ^ (((q4 >> 1) & 1) << 6));
    int b_off16[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) b_off16[kw] = (kw + 8 * h + q4) * 128 + (b_cbyte16 ^ ((((kw + q4) >> 1) & 1) << 6));
    // f32 path: lane-half h holds pixel 2s+h of the k-pair; swizzle bit = ((kw + h) >> 1 + s) & 1
    const int a_off32 = h * 128 + (nb32 * 32 + r) * 4;
    int b_off32[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) b_off32[kw] = (kw + h) * 128 + (((cb32 * 32 + r) * 4) ^ ((((kw + h) >> 1) & 1) << 6));

    // software pipeline over this split's tiles: LDS holds tile t (being read) and tile t+1 (being written by
    // store_tile from registers loaded one iteration earlier); registers are then refilled with tile t+2.
    // One barrier per tile; a wave's LDS writes overlap the other waves' MFMAs.
    int tile = split;
    if (tile < ntiles) {
        load_tile(tile);
        store_tile();
        if (tile + p.splitk < ntiles) load_tile(tile + p.splitk);
    }
    __syncthreads();
#ifdef CMU_IG_STAMPS
    const int stamp_slot = (blockIdx.x % 7 == 0 && blockIdx.x / 7 < 64) ? (int)(blockIdx.x / 7) : -1;
    int stamp_i = -1;
#endif
    for (; tile < ntiles; tile += p.splitk) {
#ifdef CMU_IG_STAMPS
        ++stamp_i;
#endif
        WG_STAMP(0);
        stA = (smA == smem) ? smem + C::BUF_BYTES : smem;
        stB = stA + C::A_BYTES;
#pragma unroll 1
        for (int rr = 0; rr < C::RPP; ++rr) {
            const int py = kpart * C::RPP + rr;
            if constexpr (EPC == 8) {
                // A fragment: 8 consecutive pixels (k = 8h + j) of row py for channel row r
                const unsigned char* pa = smA + py * (16 * 128) + a_off16;
                const u32x2 a_lo = lds_tr16(pa, 0);
                const u32x2 a_hi = lds_tr16(pa, 512);
                const u32x4 afrag = {a_lo[0], a_lo[1], a_hi[0], a_hi[1]};
                const unsigned char* pbrow = smB + py * (C::LWP * 128);
#pragma unroll
                for (int t = 0; t < C::TAPS; ++t) {
                    const int kh = (C::TAPS == 9) ? t / 3 : 0, kw = (C::TAPS == 9) ? t % 3 : 0;
                    const unsigned char* pb = pbrow + kh * (C::LWP * 128) + b_off16[kw];
                    const u32x2 b_lo = lds_tr16(pb, 0);
                    const u32x2 b_hi = lds_tr16(pb, 512);
                    const u32x4 bfrag = {b_lo[0], b_lo[1], b_hi[0], b_hi[1]};
                    TR::mma16(afrag, bfrag, acc[t]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float av = *reinterpret_cast<const float*>(smA + (py * 16 + 2 * s) * 128 + (a_off32 ^ ((s & 1) << 6)));
#pragma unroll
                    for (int t = 0; t < C::TAPS; ++t) {
                        const int kh = (C::TAPS == 9) ? t / 3 : 0, kw = (C::TAPS == 9) ? t % 3 : 0;
                        const float bv = *reinterpret_cast<const float*>(smB + ((py + kh) * C::LWP + 2 * s) * 128 + (b_off32[kw] ^ ((s & 1) << 6)));
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        WG_STAMP(1);
        if (tile + p.splitk < ntiles) store_tile();                       // tile t+1: registers -> the other buffer
        WG_STAMP(2);
        __syncthreads();
        WG_STAMP(3);
        if (tile + 2 * p.splitk < ntiles) load_tile(tile + 2 * p.splitk);  // tile t+2: in flight during the next compute
        WG_STAMP(4);
        smA = stA;
        smB = stB;
    }
    // ---- combine the k-parts through LDS (fixed order), then write the partial slab ----------------------
    float* red = reinterpret_cast<float*>(smem) + (int64_t)tile_id * (C::TAPS * 1024);
    for (int kp = 1; kp < C::KPARTS; ++kp) {
        __syncthreads();
        if (kpart == kp) {
#pragma unroll
            for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(t * 16 + e) * 64 + lane] = acc[t][e];
        }
        __syncthreads();
        if (kpart == 0) {
#pragma unroll
            for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] += red[(t * 16 + e) * 64 + lane];
        }
    }
    if (kpart == 0) {
        const int T = (MODE == MODE_W3) ? 9 : 4;
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) {
            const int tt = (MODE == MODE_W3) ? t : ij;
            float* base = p.ws + (((int64_t)split * T + tt) * p.CApad + ab * C::CW + nb32 * 32) * p.CBpad + bb * C::CW + cb32 * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = (e & 3) + 8 * (e >> 2) + 4 * h;
                base[(int64_t)n * p.CBpad] = acc[t][e];
            }
        }
    }
}

#include "conv_wgrad2.inc"
#include "conv_wgrad2s.inc"
#include "conv_wgrad2f.inc"

// partial slabs -> parameter-gradient layout, summed in split order
#ifndef CMU_WGR_CAP
#define CMU_WGR_CAP 8192
#endif
// sum over the split slabs k = part, part + 4, ... (fixed order), eight loads in flight per trip
__device__ static inline float sum_split(const float* __restrict__ p, int64_t stride, int part, int splitk) {
    constexpr int U = 8;
    float s = 0.f;
    int k = part;
    for (; k + 4 * (U - 1) < splitk; k += 4 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(int64_t)(k + 4 * u) * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
    }
    for (; k < splitk; k += 4) s += p[(int64_t)k * stride];
    return s;
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splitk, int T, int CApad, int CBpad, int CA,
                                                          int CB, float* __restrict__ dW, int mode) {
    // 64 outputs x 4 split-parts per block; the parts are combined through LDS in a fixed order
    __shared__ float red[4][64];
    const int64_t total = (int64_t)T * CA * CB;
    const int ol = threadIdx.x & 63, part = threadIdx.x >> 6;
    for (int64_t o0 = (int64_t)blockIdx.x * 64; o0 < total; o0 += (int64_t)gridDim.x * 64) {
        const int64_t o = o0 + ol;
        int c = (int)(o % CB);
        int n = (int)((o / CB) % CA);
        const int t = (int)(o / ((int64_t)CB * CA));
        float s = 0.f;
        if (mode == 2) {   // wide ConvTranspose kernel: slabs are [split][ij][c][n], n fastest
            n = (int)(o % CA);
            c = (int)((o / CA) % CB);
            if (o < total) s = sum_split(ws + ((int64_t)t * CB + c) * CA + n, (int64_t)T * CB * CA, part, splitk);
        } else if (o < total)
            s = sum_split(ws + ((int64_t)t * CApad + n) * CBpad + c, (int64_t)T * CApad * CBpad, part, splitk);
        __syncthreads();
        red[part][ol] = s;
        __syncthreads();
        if (part == 0 && o < total) {
            s = (red[0][ol] + red[1][ol]) + (red[2][ol] + red[3][ol]);
            if (mode == MODE_W3) dW[((int64_t)n * CB + c) * 9 + t] = s;      // (Cout,Cin,3,3)
            else dW[((int64_t)c * CA + n) * 4 + t] = s;                      // (Cin,Cout,2,2)
        }
    }
}

// the same reduction four channels per lane (16-byte loads: four times the bytes in flight; every output element still sums its
// slabs in the scalar kernel's order, so the two kernels agree bit for bit).  Needs the fast slab axis in whole 4-element groups.
__device__ static inline f32x4 sum_split4(const float* __restrict__ p, int64_t stride, int part, int splitk) {
    constexpr int U = 8;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = part;
    for (; k + 4 * (U - 1) < splitk; k += 4 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(k + 4 * u) * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
    }
    for (; k < splitk; k += 4) s += *reinterpret_cast<const f32x4*>(p + (int64_t)k * stride);
    return s;
}
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ ws, int splitk, int T, int CApad, int CBpad, int CA,
                                                           int CB, float* __restrict__ dW, int mode) {
    __shared__ f32x4 red[4][64];
    const int F = (mode == 2 ? CA : CB) / 4;                       // 4-element groups along the slab's fast axis
    const int S = mode == 2 ? CB : CA;                             // the slower channel axis
    const int64_t total = (int64_t)T * S * F;
    const int ol = threadIdx.x & 63, part = threadIdx.x >> 6;
    for (int64_t o0 = (int64_t)blockIdx.x * 64; o0 < total; o0 += (int64_t)gridDim.x * 64) {
        const int64_t o = o0 + ol;
        const int f4 = (int)(o % F), sl = (int)((o / F) % S), t = (int)(o / ((int64_t)F * S));
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (o < total) {
            if (mode == 2) s = sum_split4(ws + ((int64_t)t * CB + sl) * CA + 4 * f4, (int64_t)T * CB * CA, part, splitk);
            else s = sum_split4(ws + ((int64_t)t * CApad + sl) * CBpad + 4 * f4, (int64_t)T * CApad * CBpad, part, splitk);
        }
        __syncthreads();
        red[part][ol] = s;
        __syncthreads();
        if (part == 0 && o < total) {
            s = (red[0][ol] + red[1][ol]) + (red[2][ol] + red[3][ol]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mode == 2) dW[((int64_t)sl * CA + 4 * f4 + j) * 4 + t] = s[j];             // slab [ij][c][n]: sl = c, fast = n
                else if (mode == MODE_W3) dW[((int64_t)sl * CB + 4 * f4 + j) * 9 + t] = s[j];  // (Cout,Cin,3,3): sl = n, fast = c
                else dW[((int64_t)(4 * f4 + j) * CA + sl) * 4 + t] = s[j];                     // (Cin,Cout,2,2): sl = n, fast = c
            }
        }
    }
}
// CMU_WGR_VEC=0 keeps the scalar reduction (A/B switch)
static void launch_wgrad_reduce(const float* ws, int splitk, int T, int CApad, int CBpad, int CA, int CB, float* dW, int mode, hipStream_t st) {
    static const bool vec_on = []() { const char* e = getenv("CMU_WGR_VEC"); return !(e && e[0] == '0'); }();
    const bool vec = vec_on && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (mode == 2 ? CA % 4 == 0 : (CB % 4 == 0 && CBpad % 4 == 0));
    const int64_t total = (int64_t)T * CA * CB / (vec ? 4 : 1);
    const int grid = (int)(cmu_div_up64(total, 64) < CMU_WGR_CAP ? cmu_div_up64(total, 64) : CMU_WGR_CAP);
    if (vec) hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(grid), dim3(256), 0, st, ws, splitk, T, CApad, CBpad, CA, CB, dW, mode);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, ws, splitk, T, CApad, CBpad, CA, CB, dW, mode);
}

// per-channel sum over all pixels (ConvTranspose2d bias gradient)
template <class TR>
__global__ __launch_bounds__(256) void channel_sum_kernel(const unsigned char* __restrict__ x, int64_t ldx, float* __restrict__ ws,
                                                         int64_t npix, int C, int cpb, int ppb) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int nchunk = C / EPC;
    const int ch = blockIdx.y * cpb + tid % cpb;
    const int prow = tid / cpb;
    const bool active = prow < ppb && ch < nchunk;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    if (active)
        for (int64_t pp = (int64_t)blockIdx.x * ppb + prow; pp < npix; pp += (int64_t)gridDim.x * ppb) {
            float f[EPC];
            TR::unpack(ld_global16(x + (pp * ldx + ch * EPC) * ES), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += f[e];
        }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        __syncthreads();
        red[tid] = s[e];
        __syncthreads();
        if (tid < cpb && blockIdx.y * cpb + tid < nchunk) {
            float a = 0.f;
            for (int k = 0; k < ppb; ++k) a += red[k * cpb + tid];
            ws[(int64_t)blockIdx.x * C + (blockIdx.y * cpb + tid) * EPC + e] = a;
        }
    }
}
// 16 channels x 16 row parts per 256-thread block, eight rows in flight per thread, fixed-order combine (one thread per channel
// walking up to 256 rows was a chain of dependent round trips: 23 us per launch)
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float* __restrict__ ws, int nblocks, int C, float* out) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < C) {
        constexpr int U = 8;
        int b = part;
        for (; b + 16 * (U - 1) < nblocks; b += 16 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ws[(int64_t)(b + 16 * u) * C + c];
#pragma unroll
            for (int u = 0; u < U; ++u) s += (double)v[u];
        }
        for (; b < nblocks; b += 16) s += (double)ws[(int64_t)b * C + c];
    }
    red[part][cl] = s;
    __syncthreads();
    if (part == 0 && c < C) {
        s = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][cl];
        out[c] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static int wg_splitk(int nbase, int ntiles, int mult, int target = 512) {
    // aim at ~target workgroups (512: two rounds over 256 CUs); every split gets at least one tile
    int s = target / (nbase * mult);
    if (s < 1) s = 1;
    if (s > ntiles) s = ntiles;
    return s;
}
// the wide kernels hold one workgroup per CU (LDS): 256 workgroups are exactly one round over the chip, and every
// workgroup writes its whole accumulator tile to the split-K slab -- half the workgroups is half the slab traffic of
// the reduction (1.1 ms per bench step at 512).  CMU_WGRAD_BLOCKS overrides (A/B).
static int cmu_wg_first_target() {
    static const int v = []() { const char* e = getenv("CMU_WGRAD_BLOCKS1"); const int n = e ? atoi(e) : 512; return n >= 8 ? n : 512; }();
    return v;
}
static int cmu_wg_wide_target() {
    static const int v = []() { const char* e = getenv("CMU_WGRAD_BLOCKS"); const int n = e ? atoi(e) : 256; return n >= 8 ? n : 256; }();
    return v;
}
static void wg_geometry(int B, int H, int W, int CA, int CB, int dt, int mult, WGParams& p) {
    const int CW = 128 / cmu_dtype_size(dt);
    p.tilesX = cmu_div_up(W, 16);
    p.tilesY = cmu_div_up(H, 16);
    p.ntiles = B * p.tilesX * p.tilesY;
    p.nAB = cmu_div_up(CA, CW);
    p.nBB = cmu_div_up(CB, CW);
    p.CApad = p.nAB * CW;
    p.CBpad = p.nBB * CW;
    p.splitk = wg_splitk(p.nAB * p.nBB, p.ntiles, mult, mult == 1 ? cmu_wg_first_target() : 512);
}
constexpr int CSUM_BLOCKS = 256;

// wide-tile kernel (conv_wgrad2.inc): 16-bit dtypes, Cout in whole 128-blocks, Cin in whole 64-blocks.  CMU_WGRAD_WIDE=0 keeps
// every layer on the first kernel (A/B switch for the benches).
static bool wg2_shape_ok(int CA, int CB, int dt) {
    static const bool on = []() { const char* e = getenv("CMU_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
    return on && cmu_dtype_size(dt) == 2 && CA % 128 == 0 && CB % 64 == 0;
}
// swapped roles (conv_wgrad2.inc, SWAP): Cout = 64 with Cin in whole 128-blocks.  CMU_WGRAD_SWAP=0 keeps those layers on the
// first kernel (A/B switch)
static bool wg2_swap_ok(int CA, int CB, int dt) {
    static const bool on = []() { const char* e = getenv("CMU_WGRAD_SWAP"); return !(e && e[0] == '0'); }();
    static const bool wide = []() { const char* e = getenv("CMU_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
    return on && wide && cmu_dtype_size(dt) == 2 && CA == 64 && CB % 128 == 0;
}
static void wg2_geometry(int B, int H, int W, int CA, int CB, WGParams& p, bool swap = false) {
    p.tilesX = cmu_div_up(W, 16);
    p.tilesY = cmu_div_up(H, 8);
    p.ntiles = B * p.tilesX * p.tilesY;
    p.nAB = (swap ? CB : CA) / 128;   // blocks of the plain 128-channel operand
    p.nBB = (swap ? CA : CB) / 64;    // blocks of the haloed 64-channel operand
    p.CApad = CA;
    p.CBpad = CB;
    p.splitk = wg_splitk(p.nAB * p.nBB, p.ntiles, 1, cmu_wg_wide_target());
}
static bool wgT2_shape_ok(int CA, int CB, int dt) {
    static const bool on = []() { const char* e = getenv("CMU_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
    return on && cmu_dtype_size(dt) == 2 && CA % 64 == 0 && CB % 128 == 0;
}
// 256 X channels per workgroup (conv_wgrad2.inc, NXI = 4) where Cin allows: the deeper decoder levels (CMU_WGT2_NX256=0: A/B)
static bool wgT2_wide_x(int CB) {
    static const bool on = []() { const char* e = getenv("CMU_WGT2_NX256"); return !(e && e[0] == '0'); }();
    return on && CB % 256 == 0;
}
static void wgT2_geometry(int B, int H, int W, int CA, int CB, WGParams& p) {
    p.tilesX = cmu_div_up(W, 16);
    p.tilesY = cmu_div_up(H, 4);
    p.ntiles = B * p.tilesX * p.tilesY;
    p.nAB = CA / 64;
    p.nBB = CB / (wgT2_wide_x(CB) ? 256 : 128);
    p.CApad = CA;
    p.CBpad = CB;
    p.splitk = wg_splitk(p.nAB * p.nBB, p.ntiles, 1, cmu_wg_wide_target());
}
template <class TR, int NXI>
static int wgradT_wide_launch(const WG2Params& pp, hipStream_t st) {
    typedef WGT2Cfg<TR, NXI> C;
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgradT2_kernel<TR, NXI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("cmu_convT2x2_wgrad(wide): hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL((conv_wgradT2_kernel<TR, NXI>), dim3(pp.g.nAB * pp.g.nBB * pp.g.splitk), dim3(512), C::LDS_BYTES, st, pp);
    return CMU_OK;
}
template <class TR>
static int wgradT_wide_t(WGParams p, float* dW, float* dbias, hipStream_t st) {
    WG2Params pp;
    pp.g = p;
    pp.dtx = p.splitk % p.tilesX;
    pp.dty = (p.splitk / p.tilesX) % p.tilesY;
    pp.dtb = p.splitk / (p.tilesX * p.tilesY);
    const int rc = wgT2_wide_x(p.CB) ? wgradT_wide_launch<TR, 4>(pp, st) : wgradT_wide_launch<TR, 2>(pp, st);
    if (rc != CMU_OK) return rc;
    cmu_set_kernel_tag("conv_wgradT2_kernel");
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(wide)");
    launch_wgrad_reduce((const float*)p.ws, p.splitk, 4, p.CApad, p.CBpad, p.CA, p.CB, dW, 2, st);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(reduce)");
    const float* ws_sum = p.ws + (int64_t)p.splitk * 4 * p.CB * p.CA;
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cmu_div_up(p.CA, 16)), dim3(256), 0, st, ws_sum, p.splitk, p.CA, dbias);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(bias final)");
    return CMU_OK;
}
template <class TR, bool SWAP>
static int wgrad3_wide_t(WGParams p, float* dW, hipStream_t st) {
    typedef WG2Cfg<TR> C;
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad2_kernel<TR, SWAP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("cmu_conv3x3_wgrad(wide): hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    WG2Params pp;
    pp.g = p;
    pp.dtx = p.splitk % p.tilesX;
    pp.dty = (p.splitk / p.tilesX) % p.tilesY;
    pp.dtb = p.splitk / (p.tilesX * p.tilesY);
    hipLaunchKernelGGL((conv_wgrad2_kernel<TR, SWAP>), dim3(p.nAB * p.nBB * p.splitk), dim3(512), C::LDS_BYTES, st, pp);
    cmu_set_kernel_tag("conv_wgrad2_kernel");
    CMU_CHECK_LAUNCH("cmu_conv3x3_wgrad(wide)");
    // (the slab is [split][tap][Cout][Cin] in both forms: p.CA = Cout rows of p.CB = Cin)
    launch_wgrad_reduce((const float*)p.ws, p.splitk, 9, p.CA, p.CB, p.CA, p.CB, dW, (int)MODE_W3, st);
    CMU_CHECK_LAUNCH("cmu_conv3x3_wgrad(reduce)");
    return CMU_OK;
}

// 64 n x 64 c form for the 16-bit dtypes (conv_wgrad2s.inc): whole 64-blocks on both sides where neither form above applies (the
// 64 -> 64 layers).  CMU_WGRAD_SQUARE=0 keeps them on the first kernel (A/B switch: environment read once, cmu_set_dispatch_override in tests).
static bool wg2s_shape_ok(int CA, int CB, int dt) {
    static const bool wide = []() { const char* e = getenv("CMU_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
    return wide && cmu_switch_on(CMU_SW_WGRAD_SQUARE) && cmu_dtype_size(dt) == 2 && CA % 64 == 0 && CB % 64 == 0 && !wg2_shape_ok(CA, CB, dt) &&
           !wg2_swap_ok(CA, CB, dt);
}
static void wg2s_geometry(int B, int H, int W, int CA, int CB, WGParams& p) {
    p.tilesX = cmu_div_up(W, 16);
    p.tilesY = cmu_div_up(H, 8);
    p.ntiles = B * p.tilesX * p.tilesY;
    p.nAB = CA / 64;
    p.nBB = CB / 64;
    p.CApad = CA;
    p.CBpad = CB;
    p.splitk = wg_splitk(p.nAB * p.nBB, p.ntiles, 1, cmu_wg_wide_target());
}
template <class TR>
static int wgrad3_square_t(WGParams p, float* dW, hipStream_t st) {
    typedef WG2SCfg<TR> C;
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad2s_kernel<TR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("cmu_conv3x3_wgrad(64 x 64): hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    WG2Params pp;
    pp.g = p;
    pp.dtx = p.splitk % p.tilesX;
    pp.dty = (p.splitk / p.tilesX) % p.tilesY;
    pp.dtb = p.splitk / (p.tilesX * p.tilesY);
    hipLaunchKernelGGL((conv_wgrad2s_kernel<TR>), dim3(p.nAB * p.nBB * p.splitk), dim3(512), C::LDS_BYTES, st, pp);
    cmu_set_kernel_tag("conv_wgrad2s_kernel");
    CMU_CHECK_LAUNCH("cmu_conv3x3_wgrad(64 x 64)");
    // (the wave halves wrote separate slabs: 2 x splitk partial sums, summed in slab order)
    launch_wgrad_reduce((const float*)p.ws, p.splitk * 2, 9, p.CA, p.CB, p.CA, p.CB, dW, (int)MODE_W3, st);
    CMU_CHECK_LAUNCH("cmu_conv3x3_wgrad(reduce)");
    return CMU_OK;
}

// fp32 wide kernel (conv_wgrad2f.inc): Cout and Cin in whole 64-blocks (128 n x 64 c blocks when Cout allows, else 64 x 64 with two
// k-parts).  CMU_WGRAD_WIDE_F32=0 keeps fp32 on the first kernel (A/B switch: environment read once, cmu_set_dispatch_override in tests).
static bool wg2f_shape_ok(int CA, int CB, int dt) {
    static const bool wide = []() { const char* e = getenv("CMU_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
    return wide && cmu_switch_on(CMU_SW_WGRAD_WIDE_F32) && dt == CMU_F32 && CA % 64 == 0 && CB % 64 == 0;
}
static void wg2f_geometry(int B, int H, int W, int CA, int CB, WGParams& p) {
    p.tilesX = cmu_div_up(W, 16);
    p.tilesY = cmu_div_up(H, 4);
    p.ntiles = B * p.tilesX * p.tilesY;
    p.nAB = CA / (CA % 128 == 0 ? 128 : 64);
    p.nBB = CB / 64;
    p.CApad = CA;
    p.CBpad = CB;
    p.splitk = wg_splitk(p.nAB * p.nBB, p.ntiles, 1, cmu_wg_wide_target());
}
template <int NAI>
static int wgrad3_wide_f32_t(WGParams p, float* dW, hipStream_t st) {
    typedef WG2FCfg<NAI> C;
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad2f_kernel<NAI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("cmu_conv3x3_wgrad(wide, fp32): hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    WG2Params pp;
    pp.g = p;
    pp.dtx = p.splitk % p.tilesX;
    pp.dty = (p.splitk / p.tilesX) % p.tilesY;
    pp.dtb = p.splitk / (p.tilesX * p.tilesY);
    hipLaunchKernelGGL((conv_wgrad2f_kernel<NAI>), dim3(p.nAB * p.nBB * p.splitk), dim3(512), C::LDS_BYTES, st, pp);
    cmu_set_kernel_tag("conv_wgrad2f_kernel");
    CMU_CHECK_LAUNCH("cmu_conv3x3_wgrad(wide, fp32)");
    // (the wave halves of the 64 x 64 block wrote separate slabs: KP x splitk partial sums, summed in slab order)
    launch_wgrad_reduce((const float*)p.ws, p.splitk * C::KP, 9, p.CA, p.CB, p.CA, p.CB, dW, (int)MODE_W3, st);
    CMU_CHECK_LAUNCH("cmu_conv3x3_wgrad(reduce)");
    return CMU_OK;
}

template <class TR, int MODE>
static int launch_wgrad(const WGParams& p, hipStream_t st, const char* name) {
    typedef WGCfg<TR, MODE> C;
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<TR, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("%s: hipFuncSetAttribute(%d B LDS): %s", name, C::LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    const int grid = p.nAB * p.nBB * p.splitk * (MODE == MODE_WT ? 4 : 1);
    hipLaunchKernelGGL((conv_wgrad_kernel<TR, MODE>), dim3(grid), dim3(512), C::LDS_BYTES, st, p);
    cmu_set_kernel_tag("conv_wgrad_kernel");
    CMU_CHECK_LAUNCH(name);
    return CMU_OK;
}
template <class TR>
static int wgrad3_t(WGParams p, float* dW, hipStream_t st) {
    int rc = launch_wgrad<TR, MODE_W3>(p, st, "cmu_conv3x3_wgrad");
    if (rc) return rc;
    launch_wgrad_reduce((const float*)p.ws, p.splitk, 9, p.CApad, p.CBpad, p.CA, p.CB, dW, (int)MODE_W3, st);
    CMU_CHECK_LAUNCH("cmu_conv3x3_wgrad(reduce)");
    return CMU_OK;
}
template <class TR>
static int wgradT_t(WGParams p, float* dW, float* dbias, float* ws_sum, hipStream_t st) {
    int rc = launch_wgrad<TR, MODE_WT>(p, st, "cmu_convT2x2_wgrad");
    if (rc) return rc;
    launch_wgrad_reduce((const float*)p.ws, p.splitk, 4, p.CApad, p.CBpad, p.CA, p.CB, dW, (int)MODE_WT, st);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(reduce)");
    // bias gradient: sum of dOut over all (B,2H,2W) pixels
    const int nchunk = p.CA / TR::EPC;
    const int cpb = nchunk < 256 ? nchunk : 256, ppb = 256 / cpb, gy = cmu_div_up(nchunk, cpb);
    const int64_t npix = (int64_t)p.B * 4 * p.H * p.W;
    int gx = (int)(cmu_div_up64(npix, ppb * 4) < CSUM_BLOCKS ? cmu_div_up64(npix, ppb * 4) : CSUM_BLOCKS);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((channel_sum_kernel<TR>), dim3(gx, gy), dim3(256), 0, st, (const unsigned char*)p.a, p.lda, ws_sum, npix, p.CA, cpb, ppb);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(bias)");
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cmu_div_up(p.CA, 16)), dim3(256), 0, st, (const float*)ws_sum, gx, p.CA, dbias);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(bias final)");
    return CMU_OK;
}

// fp32 wide ConvTranspose weight gradient (conv_wgrad2f.inc): 128 c x 64 n blocks.  Follows CMU_WGRAD_WIDE_F32.
static bool wgT2f_shape_ok(int CA, int CB, int dt) { return wg2f_shape_ok(64, 64, dt) && CA % 64 == 0 && CB % 128 == 0; }
static void wgT2f_geometry(int B, int H, int W, int CA, int CB, WGParams& p) {
    p.tilesX = cmu_div_up(W, 16);
    p.tilesY = cmu_div_up(H, 2);
    p.ntiles = B * p.tilesX * p.tilesY;
    p.nAB = CA / 64;
    p.nBB = CB / 128;
    p.CApad = CA;
    p.CBpad = CB;
    p.splitk = wg_splitk(p.nAB * p.nBB, p.ntiles, 1, cmu_wg_wide_target());
}
static int wgradT_wide_f32(WGParams p, float* dW, float* dbias, float* ws_sum, hipStream_t st) {
    typedef WGT2FCfg C;
    static CmuPerDevice attr_set;   // hipFuncSetAttribute is per device
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgradT2f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) {
            cmu_set_error("cmu_convT2x2_wgrad(wide, fp32): hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CMU_ERR_LAUNCH;
        }
        attr_set.mark();
    }
    WG2Params pp;
    pp.g = p;
    pp.dtx = p.splitk % p.tilesX;
    pp.dty = (p.splitk / p.tilesX) % p.tilesY;
    pp.dtb = p.splitk / (p.tilesX * p.tilesY);
    hipLaunchKernelGGL(conv_wgradT2f_kernel, dim3(p.nAB * p.nBB * p.splitk), dim3(512), C::LDS_BYTES, st, pp);
    cmu_set_kernel_tag("conv_wgradT2f_kernel");
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(wide, fp32)");
    launch_wgrad_reduce((const float*)p.ws, p.splitk, 4, p.CApad, p.CBpad, p.CA, p.CB, dW, 2, st);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(reduce)");
    // bias gradient: sum of dOut over all (B,2H,2W) pixels (the first path's channel-sum kernels)
    typedef F32Traits TR;
    const int nchunk = p.CA / TR::EPC;
    const int cpb = nchunk < 256 ? nchunk : 256, ppb = 256 / cpb, gy = cmu_div_up(nchunk, cpb);
    const int64_t npix = (int64_t)p.B * 4 * p.H * p.W;
    int gx = (int)(cmu_div_up64(npix, ppb * 4) < CSUM_BLOCKS ? cmu_div_up64(npix, ppb * 4) : CSUM_BLOCKS);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((channel_sum_kernel<TR>), dim3(gx, gy), dim3(256), 0, st, (const unsigned char*)p.a, p.lda, ws_sum, npix, p.CA, cpb, ppb);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(bias)");
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cmu_div_up(p.CA, 16)), dim3(256), 0, st, (const float*)ws_sum, gx, p.CA, dbias);
    CMU_CHECK_LAUNCH("cmu_convT2x2_wgrad(bias final)");
    return CMU_OK;
}

static int wg_check(const char* name, const void* t, int64_t ld, int C, int dt) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0, "%s: bad dtype %d", name, dt);
    const int epc = 16 / es;
    CMU_CHECK_ARG(t && cmu_aligned16(t), "%s: null / unaligned tensor", name);
    CMU_CHECK_ARG(C > 0 && C % epc == 0 && ld % epc == 0 && ld >= C, "%s: C=%d / ld=%lld must be multiples of %d", name, C, (long long)ld, epc);
    return CMU_OK;
}

extern "C" int64_t cmu_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dt) {
    if (cmu_dtype_size(dt) == 0 || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
    WGParams p = {};
    wg_geometry(B, H, W, Cout, Cin, dt, 1, p);
    int64_t need = (int64_t)p.splitk * 9 * p.CApad * p.CBpad * (int64_t)sizeof(float);
    if (wg2_shape_ok(Cout, Cin, dt) || wg2_swap_ok(Cout, Cin, dt)) {   // either kernel may run (the wide one needs 4 GiB-addressable tensors)
        WGParams q = {};
        wg2_geometry(B, H, W, Cout, Cin, q, !wg2_shape_ok(Cout, Cin, dt));
        const int64_t w = (int64_t)q.splitk * 9 * q.CApad * q.CBpad * (int64_t)sizeof(float);
        if (w > need) need = w;
    }
    if (cmu_dtype_size(dt) == 2 && Cout % 64 == 0 && Cin % 64 == 0) {   // the 64 x 64 form: two slabs per split
        WGParams q = {};
        wg2s_geometry(B, H, W, Cout, Cin, q);
        const int64_t w = (int64_t)q.splitk * 2 * 9 * q.CApad * q.CBpad * (int64_t)sizeof(float);
        if (w > need) need = w;
    }
    if (dt == CMU_F32 && Cout % 64 == 0 && Cin % 64 == 0) {   // the fp32 wide kernel (two slabs per split for 64 x 64 blocks)
        WGParams q = {};
        wg2f_geometry(B, H, W, Cout, Cin, q);
        const int64_t w = (int64_t)q.splitk * (Cout % 128 == 0 ? 1 : 2) * 9 * q.CApad * q.CBpad * (int64_t)sizeof(float);
        if (w > need) need = w;
    }
    return need;
}
extern "C" int cmu_conv3x3_wgrad(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from, const void* dY,
                                 int64_t ldd, float* dW, int B, int H, int W, int Cin, int Cout, int dt, void* ws, void* stream) {
    int rc;
    if ((rc = wg_check("cmu_conv3x3_wgrad(x)", x, ldx, Cin, dt))) return rc;
    if ((rc = wg_check("cmu_conv3x3_wgrad(dY)", dY, ldd, Cout, dt))) return rc;
    CMU_CHECK_ARG(dW && ws && B > 0 && H > 0 && W > 0, "cmu_conv3x3_wgrad: null argument / bad dims");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_conv3x3_wgrad: scale/shift must both be set");
    CMU_CHECK_ARG(relu_from % (16 / cmu_dtype_size(dt)) == 0, "cmu_conv3x3_wgrad: relu_from=%d must be a multiple of the 16-byte chunk", relu_from);
    WGParams p = {};
    p.a = dY; p.lda = ldd; p.b = x; p.ldb = ldx; p.b_scale = in_scale; p.b_shift = in_shift; p.relu_from = relu_from;
    p.ws = (float*)ws; p.B = B; p.H = H; p.W = W; p.CA = Cout; p.CB = Cin;
    const int64_t px = (int64_t)H * W;   // per image: the buffer descriptors are per image
    if (wg2_shape_ok(Cout, Cin, dt) && (px * ldd + Cout) * 2 < 0x7fff0000ll && ((px + W + 1) * ldx + Cin) * 2 < 0x7fff0000ll &&
        (in_scale == nullptr || ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 3) == 0)) {
        wg2_geometry(B, H, W, Cout, Cin, p);
        if (dt == CMU_F16) return wgrad3_wide_t<F16Traits, false>(p, dW, (hipStream_t)stream);
        return wgrad3_wide_t<BF16Traits, false>(p, dW, (hipStream_t)stream);
    }
    // swapped roles: the halo is on dY here, the plain images are X
    if (wg2_swap_ok(Cout, Cin, dt) && ((px + W + 1) * ldd + Cout) * 2 < 0x7fff0000ll && (px * ldx + Cin) * 2 < 0x7fff0000ll &&
        (in_scale == nullptr || ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 15) == 0)) {
        wg2_geometry(B, H, W, Cout, Cin, p, true);
        if (dt == CMU_F16) return wgrad3_wide_t<F16Traits, true>(p, dW, (hipStream_t)stream);
        return wgrad3_wide_t<BF16Traits, true>(p, dW, (hipStream_t)stream);
    }
    if (wg2s_shape_ok(Cout, Cin, dt) && (px * ldd + Cout) * 2 < 0x7fff0000ll && ((px + W + 1) * ldx + Cin) * 2 < 0x7fff0000ll &&
        (in_scale == nullptr || ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 3) == 0)) {
        wg2s_geometry(B, H, W, Cout, Cin, p);
        if (dt == CMU_F16) return wgrad3_square_t<F16Traits>(p, dW, (hipStream_t)stream);
        return wgrad3_square_t<BF16Traits>(p, dW, (hipStream_t)stream);
    }
    if (wg2f_shape_ok(Cout, Cin, dt) && (px * ldd + Cout) * 4 < 0x7fff0000ll && ((px + W + 1) * ldx + Cin) * 4 < 0x7fff0000ll &&
        (in_scale == nullptr || ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 3) == 0)) {
        wg2f_geometry(B, H, W, Cout, Cin, p);
        return Cout % 128 == 0 ? wgrad3_wide_f32_t<4>(p, dW, (hipStream_t)stream) : wgrad3_wide_f32_t<2>(p, dW, (hipStream_t)stream);
    }
    wg_geometry(B, H, W, Cout, Cin, dt, 1, p);
    CMU_DISPATCH_DT(dt, wgrad3_t, p, dW, (hipStream_t)stream);
}

// Sparse (SparK) weight gradient over a device-side tile list (cmu_sparse_tile_list).  The list's tile height says which kernel
// walks it: 16 (16 x 16 pixel tiles) = the first kernel, any shape; 8 (8 x 16 tiles, the wide kernel's K tile) = the wide kernel,
// for the shapes cmu_conv3x3_wgrad_tile_h names.  ws: cmu_conv3x3_wgrad_ws_bytes.
static bool wg2_list_ok(int B, int H, int W, int64_t ldx, int64_t ldd, int Cin, int Cout, int dt) {
    const int64_t px = (int64_t)H * W;
    return wg2_shape_ok(Cout, Cin, dt) && (px * ldd + Cout) * 2 < 0x7fff0000ll && ((px + W + 1) * ldx + Cin) * 2 < 0x7fff0000ll;
}
static bool wg2s_list_ok(int B, int H, int W, int64_t ldx, int64_t ldd, int Cin, int Cout, int dt) {
    const int64_t px = (int64_t)H * W;
    return wg2s_shape_ok(Cout, Cin, dt) && (px * ldd + Cout) * 2 < 0x7fff0000ll && ((px + W + 1) * ldx + Cin) * 2 < 0x7fff0000ll;
}
extern "C" int cmu_conv3x3_wgrad_tile_h(int B, int H, int W, int Cin, int Cout, int dt) {
    if (cmu_dtype_size(dt) == 0 || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
    return (wg2_list_ok(B, H, W, Cin, Cout, Cin, Cout, dt) || wg2s_list_ok(B, H, W, Cin, Cout, Cin, Cout, dt)) ? 8 : 16;   // (dense NHWC tensors: ld = C)
}
extern "C" int cmu_conv3x3_wgrad_tiles(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                                       const void* dY, int64_t ldd, float* dW, const int* tile_list, const int* tile_count, int tile_h,
                                       int B, int H, int W, int Cin, int Cout, int dt, void* ws, void* stream) {
    int rc;
    if ((rc = wg_check("cmu_conv3x3_wgrad_tiles(x)", x, ldx, Cin, dt))) return rc;
    if ((rc = wg_check("cmu_conv3x3_wgrad_tiles(dY)", dY, ldd, Cout, dt))) return rc;
    CMU_CHECK_ARG(dW && ws && tile_list && tile_count && B > 0 && H > 0 && W > 0, "cmu_conv3x3_wgrad_tiles: null argument / bad dims");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_conv3x3_wgrad_tiles: scale/shift must both be set");
    CMU_CHECK_ARG(relu_from % (16 / cmu_dtype_size(dt)) == 0, "cmu_conv3x3_wgrad_tiles: relu_from=%d must be a multiple of the 16-byte chunk", relu_from);
    CMU_CHECK_ARG(tile_h == 16 || tile_h == 8, "cmu_conv3x3_wgrad_tiles: tile_h=%d (16: 16 x 16 tiles, 8: 8 x 16 tiles)", tile_h);
    WGParams p = {};
    p.a = dY; p.lda = ldd; p.b = x; p.ldb = ldx; p.b_scale = in_scale; p.b_shift = in_shift; p.relu_from = relu_from;
    p.ws = (float*)ws; p.B = B; p.H = H; p.W = W; p.CA = Cout; p.CB = Cin;
    p.tile_list = tile_list; p.tile_count = tile_count;
    if (tile_h == 8 && wg2s_list_ok(B, H, W, ldx, ldd, Cin, Cout, dt) &&
        (in_scale == nullptr || ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 3) == 0)) {
        wg2s_geometry(B, H, W, Cout, Cin, p);   // (the 64 x 64 form walks the same 8 x 16 lists)
        if (dt == CMU_F16) return wgrad3_square_t<F16Traits>(p, dW, (hipStream_t)stream);
        return wgrad3_square_t<BF16Traits>(p, dW, (hipStream_t)stream);
    }
    if (tile_h == 8) {
        CMU_CHECK_ARG(wg2_list_ok(B, H, W, ldx, ldd, Cin, Cout, dt) &&
                          (in_scale == nullptr || ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 3) == 0),
                      "cmu_conv3x3_wgrad_tiles: an 8 x 16 tile list needs a shape of the wide kernel (16-bit, Cout %% 128 == 0, Cin %% 64 == 0; "
                      "cmu_conv3x3_wgrad_tile_h), got %d -> %d", Cin, Cout);
        wg2_geometry(B, H, W, Cout, Cin, p);
        if (dt == CMU_F16) return wgrad3_wide_t<F16Traits, false>(p, dW, (hipStream_t)stream);
        return wgrad3_wide_t<BF16Traits, false>(p, dW, (hipStream_t)stream);
    }
    wg_geometry(B, H, W, Cout, Cin, dt, 1, p);
    CMU_DISPATCH_DT(dt, wgrad3_t, p, dW, (hipStream_t)stream);
}

extern "C" int64_t cmu_convT2x2_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dt) {
    if (cmu_dtype_size(dt) == 0 || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
    WGParams p = {};
    wg_geometry(B, H, W, Cout, Cin, dt, 4, p);
    int64_t need = ((int64_t)p.splitk * 4 * p.CApad * p.CBpad + (int64_t)CSUM_BLOCKS * Cout) * (int64_t)sizeof(float);
    if (wgT2_shape_ok(Cout, Cin, dt)) {
        WGParams q = {};
        wgT2_geometry(B, H, W, Cout, Cin, q);
        const int64_t w = (int64_t)q.splitk * ((int64_t)4 * Cout * Cin + Cout) * (int64_t)sizeof(float);
        if (w > need) need = w;
    }
    if (dt == CMU_F32 && Cout % 64 == 0 && Cin % 128 == 0) {   // the fp32 wide form
        WGParams q = {};
        wgT2f_geometry(B, H, W, Cout, Cin, q);
        const int64_t w = ((int64_t)q.splitk * 4 * Cout * Cin + (int64_t)CSUM_BLOCKS * Cout) * (int64_t)sizeof(float);
        if (w > need) need = w;
    }
    return need;
}
extern "C" int cmu_convT2x2_wgrad(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from, const void* dOut,
                                  int64_t ldd, float* dW, float* dbias, int B, int H, int W, int Cin, int Cout, int dt, void* ws,
                                  void* stream) {
    int rc;
    if ((rc = wg_check("cmu_convT2x2_wgrad(x)", x, ldx, Cin, dt))) return rc;
    if ((rc = wg_check("cmu_convT2x2_wgrad(dOut)", dOut, ldd, Cout, dt))) return rc;
    CMU_CHECK_ARG(dW && dbias && ws && B > 0 && H > 0 && W > 0, "cmu_convT2x2_wgrad: null argument / bad dims");
    CMU_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "cmu_convT2x2_wgrad: scale/shift must both be set");
    CMU_CHECK_ARG(relu_from % (16 / cmu_dtype_size(dt)) == 0, "cmu_convT2x2_wgrad: relu_from=%d must be a multiple of the 16-byte chunk", relu_from);
    WGParams p = {};
    p.a = dOut; p.lda = ldd; p.b = x; p.ldb = ldx; p.b_scale = in_scale; p.b_shift = in_shift; p.relu_from = relu_from;
    p.ws = (float*)ws; p.B = B; p.H = H; p.W = W; p.CA = Cout; p.CB = Cin;
    const int64_t px = (int64_t)H * W;   // per image: the buffer descriptors are per image
    if (wgT2_shape_ok(Cout, Cin, dt) && (4 * px * ldd + Cout) * 2 < 0x7fff0000ll && (px * ldx + Cin) * 2 < 0x7fff0000ll) {
        wgT2_geometry(B, H, W, Cout, Cin, p);
        if (dt == CMU_F16) return wgradT_wide_t<F16Traits>(p, dW, dbias, (hipStream_t)stream);
        return wgradT_wide_t<BF16Traits>(p, dW, dbias, (hipStream_t)stream);
    }
    if (wgT2f_shape_ok(Cout, Cin, dt) && (4 * px * ldd + Cout) * 4 < 0x7fff0000ll && (px * ldx + Cin) * 4 < 0x7fff0000ll &&
        (in_scale == nullptr || ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 3) == 0)) {
        wgT2f_geometry(B, H, W, Cout, Cin, p);
        return wgradT_wide_f32(p, dW, dbias, (float*)ws + (int64_t)p.splitk * 4 * Cout * Cin, (hipStream_t)stream);
    }
    wg_geometry(B, H, W, Cout, Cin, dt, 4, p);
    float* ws_sum = (float*)ws + (int64_t)p.splitk * 4 * p.CApad * p.CBpad;
    const int dtc = dt;
    switch (dtc) {
        case CMU_F32: return wgradT_t<F32Traits>(p, dW, dbias, ws_sum, (hipStream_t)stream);
        case CMU_F16: return wgradT_t<F16Traits>(p, dW, dbias, ws_sum, (hipStream_t)stream);
        case CMU_BF16: return wgradT_t<BF16Traits>(p, dW, dbias, ws_sum, (hipStream_t)stream);
    }
    cmu_set_error("cmu_convT2x2_wgrad: unknown dtype %d", dt);
    return CMU_ERR_ARG;
}
