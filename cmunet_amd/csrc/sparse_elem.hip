// sparse_elem.hip -- patch-organised forms of the sparse encoder's element-wise passes (round 4; reference semantics:
// Pretraining/Spark/encoder.py:12-56 -- every op of the sparse encoder is followed by a multiplication with the up-sampled
// active-patch map, BatchNorm statistics are taken over active positions only; spark.py:98-111 -- mask-token gradient).
//
// Why: the pixel-organised masked kernels of sparse.hip / backward_elem.hip walk ALL pixels, look the patch map up per pixel
// (integer division for the coordinates, one dependent byte load, a divergent branch) and therefore take the dense pass's time at
// 25 % of its work (s_memtime-free evidence: rocprofv3 timeline of one SparK step, profiles/r04_spark_timeline.txt -- masked
// BatchNorm-backward apply 414 us against 405 us dense at 64 ch x 512 x 512, masked pool backward 206 against 207).
// Here the unit of work is a group of PATCH ROWS: a workgroup decodes (image, patch, first row) once with scalar arithmetic, the
// branch on the patch's bit is uniform over the workgroup, every thread keeps one channel chunk (its BatchNorm constants are loaded
// once) and up to four 16-byte chunks in flight, and a masked patch costs either nothing (passes whose consumers only ever visit
// active patches) or streaming zero stores without a single load.
//
// `ring` (apply / select): zeros are written only to the one-pixel border frame of each masked patch instead of the whole patch.
// That is enough when every consumer of the output is list-driven (tile lists / pixel lists: they read active patches plus a
// one-pixel halo) -- the host passes ring = 1 only then; the interior of masked patches is left unwritten and never read.
//
// The active-patch arithmetic is the pixel kernels' own, expression for expression: outputs at active positions are bit-identical
// (tests/test_gpu_sparse_tiles.py).
#include "common.h"

#ifndef CMU_CELLS_XLOG
#define CMU_CELLS_XLOG 10     // log2 of the chunks a group of small patches is filled up to (tools/cells_bench.py: 0 / 8 / 9 / 10 ->
                              // apply 614 / 611 / 593 / 589 us over the five levels, select 627 / 593 / 552 / 527, mask-token sum 373 / 359 / 339 / 329)
#endif

struct CellGeo {
    int f, ff;      // patch map side, f * f
    int sbits;      // log2(patch side in pixels at this level)
    int cbits;      // log2(16-byte chunks per pixel)
    int rbits;      // log2(patch rows per work item)
    int gbits;      // log2(row groups per patch) = sbits - rbits
    int xbits;      // log2(horizontally adjacent patches per work item): small patches are taken several at a time
    int fg;         // f >> xbits
    int H, W;
    int nitems;     // B * f * fg << gbits
};

// work item wi = (image b, patch row fy, group of 2^xbits adjacent patches, row group): first patch cell0, first pixel row y0, first
// pixel column x0, first row inside the patch yy0.  Chunk k of the item: channel chunk k & cmask, pixel column j = (k >> cbits) &
// (2^(sbits + xbits) - 1) -- patch cell0 + (j >> sbits), column j & (ps - 1) inside it -- and row k >> (cbits + sbits + xbits).
__device__ static inline void cell_decode(const CellGeo& g, int wi, int& cell0, int& b, int& y0, int& x0, int& yy0) {
    const int cg = wi >> g.gbits;
    const int rg = wi & ((1 << g.gbits) - 1);
    const int per_img = g.f * g.fg;
    b = cg / per_img;
    const int rem = cg - b * per_img;
    const int fy = rem / g.fg, fxg = rem - fy * g.fg;
    cell0 = (b * g.f + fy) * g.f + (fxg << g.xbits);
    yy0 = rg << g.rbits;
    y0 = (fy << g.sbits) + yy0;
    x0 = (fxg << g.xbits) << g.sbits;
}

static bool cells_geometry(int B, int H, int W, int C, int epc, int f, int max_chunks_log2, bool pooled, CellGeo* g) {
    if (B <= 0 || H <= 0 || W != H || f <= 0 || C <= 0 || C % epc != 0) return false;
    const int sb = sp_shift_bits(H, f);
    if (sb < (pooled ? 1 : 0)) return false;
    const int nchunk = C / epc;
    if (nchunk > 256 || (nchunk & (nchunk - 1)) != 0) return false;
    int cb = 0;
    while ((1 << cb) < nchunk) ++cb;
    if ((int64_t)B * f * f * (1ll << sb) >= (1ll << 31) || (int64_t)B * H * W >= (1ll << 31)) return false;
    g->f = f; g->ff = f * f;
    g->sbits = pooled ? sb - 1 : sb;            // pooled form: the geometry of the POOLED level (patch side / 2)
    g->cbits = cb;
    int rb = max_chunks_log2 - g->sbits - cb;
    if (rb < 0) rb = 0;
    if (rb > g->sbits) rb = g->sbits;
    g->rbits = rb;
    g->gbits = g->sbits - rb;
    // whole patches smaller than a workgroup's four chunks per thread: several side by side (a thread whose chunks are all masked
    // skips the BatchNorm constants -- loading them in every workgroup was what made grouped items slower than small ones)
    int xb = CMU_CELLS_XLOG - g->sbits - cb - rb;
    if (xb < 0) xb = 0;
    while (xb > 0 && (f & ((1 << xb) - 1)) != 0) --xb;
    g->xbits = xb;
    g->fg = f >> xb;
    g->H = pooled ? H / 2 : H;
    g->W = pooled ? W / 2 : W;
    g->nitems = (B * f * g->fg) << g->gbits;
    return true;
}

extern "C" int cmu_cells_supported(int B, int H, int W, int f, int C, int dt) {
    const int es = cmu_dtype_size(dt);
    if (es <= 0) return 0;
    CellGeo g;
    return cells_geometry(B, H, W, C, 16 / es, f, 10, false, &g) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm+ReLU backward, apply pass, sparse form: dY = scale * (gate * dA - c1 - xhat * c2) in active patches, zeros in masked
// ones (cmu_bn_bwd_apply_masked's contract; ring = 1: zeros in the border frame of masked patches only)
// ---------------------------------------------------------------------------------------------------
template <class TR, bool MULTI>
__global__ __launch_bounds__(256) void bn_bwd_apply_cells_kernel(const unsigned char* __restrict__ dA, int64_t ldd,
                                                                const unsigned char* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ coef, unsigned char* __restrict__ dY, int64_t ldo,
                                                                const uint8_t* __restrict__ active, CellGeo g, int C, int ring) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    const int tid = threadIdx.x;
    int cell0, b, y0, x0, yy0;
    cell_decode(g, blockIdx.x, cell0, b, y0, x0, yy0);
    const int ps = 1 << g.sbits, cmask = (1 << g.cbits) - 1, wbits = g.sbits + g.xbits, wmask = (1 << wbits) - 1;
    const int total = (1 << g.rbits) << (wbits + g.cbits);
    const int64_t row0 = (int64_t)b * g.H + y0;
    if (!MULTI && active[cell0] == 0) {         // one patch per work item: a masked one costs stores only (uniform branch)
        for (int k = tid; k < total; k += 256) {
            const int c = k & cmask, j = (k >> g.cbits) & wmask, r = k >> (g.cbits + wbits);
            if (ring) {
                const int yy = yy0 + r;
                if (!(yy == 0 || yy == ps - 1 || j == 0 || j == ps - 1)) continue;
            }
            const int64_t p = (row0 + r) * g.W + x0 + j;
            st_global16(dY + (p * ldo + c * EPC) * ES, u32x4{0u, 0u, 0u, 0u});
        }
        return;
    }
    const int ch = tid & cmask;                 // 256 % chunks-per-pixel == 0: the same channel chunk on every trip
    // several patches per item (at most 1,024 chunks: one trip): their bits first -- a thread whose chunks are all masked loads no
    // BatchNorm constants (twelve 16-byte loads against two per chunk)
    uint8_t ab[4] = {1, 1, 1, 1};
    bool any = !MULTI;
    if (MULTI) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = tid + 256 * i;
            ab[i] = k < total ? active[cell0 + (((k >> g.cbits) & wmask) >> g.sbits)] : (uint8_t)0;
            any |= ab[i] != 0;
        }
    }
    float sc[EPC], sh[EPC], mu[EPC], is[EPC], c1[EPC], c2[EPC];
    if (any) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = ch * EPC + e;
            sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
            c1[e] = coef[c]; c2[e] = coef[C + c];
        }
    }
    for (int k0 = tid; k0 < total; k0 += 1024) {
        u32x4 gq[4], vq[4];
        int64_t p[4];
        int st[4];                                 // 0: nothing to do, 1: active, 2: zero store
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 256 * i;
            st[i] = 0;
            if (k < total) {
                const int j = (k >> g.cbits) & wmask, r = k >> (g.cbits + wbits);
                p[i] = (row0 + r) * g.W + x0 + j;
                if (ab[i] != 0) {
                    st[i] = 1;
                    gq[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dA + (p[i] * ldd + ch * EPC) * ES));
                    vq[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + (p[i] * ldy + ch * EPC) * ES));
                } else {
                    const int yy = yy0 + r, jj = j & (ps - 1);
                    st[i] = (!ring || yy == 0 || yy == ps - 1 || jj == 0 || jj == ps - 1) ? 2 : 0;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (st[i] == 2) st_global16(dY + (p[i] * ldo + ch * EPC) * ES, u32x4{0u, 0u, 0u, 0u});
            if (st[i] == 1) {
                float gg[EPC], v[EPC], o[EPC];
                TR::unpack(gq[i], gg);
                TR::unpack(vq[i], v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float dz = fmaf(v[e], sc[e], sh[e]) > 0.f ? gg[e] : 0.f;
                    const float xh = (v[e] - mu[e]) * is[e];
                    o[e] = sc[e] * (dz - c1[e] - xh * c2[e]);
                }
                __builtin_nontemporal_store(TR::pack(o), reinterpret_cast<u32x4*>(dY + (p[i] * ldo + ch * EPC) * ES));
            }
        }
    }
}

static int check_cells_act(const char* name, const void* a, int64_t lda, int C, int dt) {
    const int es = cmu_dtype_size(dt);
    CMU_CHECK_ARG(es > 0, "%s: bad dtype %d", name, dt);
    const int epc = 16 / es;
    CMU_CHECK_ARG(a && cmu_aligned16(a) && C > 0 && C % epc == 0 && lda % epc == 0 && lda >= C, "%s: null / unaligned tensor or C=%d, ld=%lld not multiples of %d",
                  name, C, (long long)lda, epc);
    return CMU_OK;
}
#define CMU_CELLS_GEOMETRY(name, pooled, maxlog)                                                                                             \
    CellGeo g;                                                                                                                               \
    if (!cells_geometry(B, H, W, C, 16 / cmu_dtype_size(dt), f, maxlog, pooled, &g)) {                                                       \
        cmu_set_error(name ": needs a square level with H = f << s, a power-of-two number (<= 256) of 16-byte chunks per pixel "            \
                           "(B=%d H=%d W=%d f=%d C=%d): call the pixel-organised form", B, H, W, f, C);                                      \
        return CMU_ERR_UNSUPPORTED;                                                                                                          \
    }

template <class TR>
static int bn_bwd_apply_cells_t(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const float* coef, void* dY, int64_t ldo, const uint8_t* active, CellGeo g, int C, int ring,
                                hipStream_t st) {
    if (g.xbits > 0)
        hipLaunchKernelGGL((bn_bwd_apply_cells_kernel<TR, true>), dim3(g.nitems), dim3(256), 0, st, (const unsigned char*)dA, ldd, (const unsigned char*)y,
                           ldy, scale, shift, mean, invstd, coef, (unsigned char*)dY, ldo, active, g, C, ring);
    else
        hipLaunchKernelGGL((bn_bwd_apply_cells_kernel<TR, false>), dim3(g.nitems), dim3(256), 0, st, (const unsigned char*)dA, ldd, (const unsigned char*)y,
                           ldy, scale, shift, mean, invstd, coef, (unsigned char*)dY, ldo, active, g, C, ring);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_apply_cells");
    return CMU_OK;
}
extern "C" int cmu_bn_bwd_apply_cells(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                                      const float* save_mean, const float* save_invstd, const float* coef, void* dY, int64_t ldo,
                                      const uint8_t* active, int f, int ring, int B, int H, int W, int C, int dt, void* stream) {
    int rc;
    if ((rc = check_cells_act("cmu_bn_bwd_apply_cells(dA)", dA, ldd, C, dt))) return rc;
    if ((rc = check_cells_act("cmu_bn_bwd_apply_cells(y)", y, ldy, C, dt))) return rc;
    if ((rc = check_cells_act("cmu_bn_bwd_apply_cells(dY)", dY, ldo, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && active, "cmu_bn_bwd_apply_cells: null argument");
    CMU_CELLS_GEOMETRY("cmu_bn_bwd_apply_cells", false, 10)
    CMU_DISPATCH_DT(dt, bn_bwd_apply_cells_t, dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, coef, dY, ldo, active, g, C, ring,
                    (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// out = active ? relu?(x * scale + shift) : 0   (cmu_mask_select with a zero fill; ring as above)
// ---------------------------------------------------------------------------------------------------
template <class TR, bool MULTI>
__global__ __launch_bounds__(256) void mask_select_cells_kernel(const unsigned char* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int relu, unsigned char* __restrict__ out,
                                                               int64_t ldo, const uint8_t* __restrict__ active, CellGeo g, int ring,
                                                               const float* __restrict__ fill) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    const int tid = threadIdx.x;
    int cell0, b, y0, x0, yy0;
    cell_decode(g, blockIdx.x, cell0, b, y0, x0, yy0);
    const int ps = 1 << g.sbits, cmask = (1 << g.cbits) - 1, wbits = g.sbits + g.xbits, wmask = (1 << wbits) - 1;
    const int total = (1 << g.rbits) << (wbits + g.cbits);
    const int64_t row0 = (int64_t)b * g.H + y0;
    // what masked positions receive: zeros, or the per-channel fill vector (the densify step's mask tokens, spark.py:103-107) -- the
    // thread's channel chunk is the same on every trip (256 % chunks-per-pixel == 0)
    u32x4 fillv = u32x4{0u, 0u, 0u, 0u};
    if (fill != nullptr) {
        float fl[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) fl[e] = fill[(tid & cmask) * EPC + e];
        fillv = TR::pack(fl);
    }
    if (!MULTI && active[cell0] == 0) {
        for (int k = tid; k < total; k += 256) {
            const int c = k & cmask, j = (k >> g.cbits) & wmask, r = k >> (g.cbits + wbits);
            if (ring) {
                const int yy = yy0 + r;
                if (!(yy == 0 || yy == ps - 1 || j == 0 || j == ps - 1)) continue;
            }
            const int64_t p = (row0 + r) * g.W + x0 + j;
            st_global16(out + (p * ldo + c * EPC) * ES, fillv);
        }
        return;
    }
    const int ch = tid & cmask;
    uint8_t ab[4] = {1, 1, 1, 1};
    bool any = !MULTI;
    if (MULTI) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = tid + 256 * i;
            ab[i] = k < total ? active[cell0 + (((k >> g.cbits) & wmask) >> g.sbits)] : (uint8_t)0;
            any |= ab[i] != 0;
        }
    }
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = (scale && any) ? scale[ch * EPC + e] : 1.f;
        sh[e] = (scale && any) ? shift[ch * EPC + e] : 0.f;
    }
    for (int k0 = tid; k0 < total; k0 += 1024) {
        u32x4 vq[4];
        int64_t p[4];
        int st[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 256 * i;
            st[i] = 0;
            if (k < total) {
                const int j = (k >> g.cbits) & wmask, r = k >> (g.cbits + wbits);
                p[i] = (row0 + r) * g.W + x0 + j;
                if (ab[i] != 0) {
                    st[i] = 1;
                    vq[i] = ld_global16(x + (p[i] * ldx + ch * EPC) * ES);
                } else {
                    const int yy = yy0 + r, jj = j & (ps - 1);
                    st[i] = (!ring || yy == 0 || yy == ps - 1 || jj == 0 || jj == ps - 1) ? 2 : 0;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (st[i] == 2) st_global16(out + (p[i] * ldo + ch * EPC) * ES, fillv);
            if (st[i] == 1) {
                u32x4 o = vq[i];
                if (scale || relu) {
                    float v[EPC];
                    TR::unpack(o, v);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float t = fmaf(v[e], sc[e], sh[e]);
                        v[e] = relu ? fmaxf(t, 0.f) : t;
                    }
                    o = TR::pack(v);
                }
                st_global16(out + (p[i] * ldo + ch * EPC) * ES, o);
            }
        }
    }
}
template <class TR>
static int mask_select_cells_t(const void* x, int64_t ldx, const float* scale, const float* shift, int relu, void* out, int64_t ldo,
                               const uint8_t* active, CellGeo g, int ring, const float* fill, hipStream_t st) {
    if (g.xbits > 0)
        hipLaunchKernelGGL((mask_select_cells_kernel<TR, true>), dim3(g.nitems), dim3(256), 0, st, (const unsigned char*)x, ldx, scale, shift, relu,
                           (unsigned char*)out, ldo, active, g, ring, fill);
    else
        hipLaunchKernelGGL((mask_select_cells_kernel<TR, false>), dim3(g.nitems), dim3(256), 0, st, (const unsigned char*)x, ldx, scale, shift, relu,
                           (unsigned char*)out, ldo, active, g, ring, fill);
    CMU_CHECK_LAUNCH("cmu_mask_select_cells");
    return CMU_OK;
}
extern "C" int cmu_mask_select_cells(const void* x, int64_t ldx, const float* scale, const float* shift, int relu, const uint8_t* active, int f,
                                     int ring, const float* fill, void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream) {
    int rc;
    if ((rc = check_cells_act("cmu_mask_select_cells(x)", x, ldx, C, dt))) return rc;
    if ((rc = check_cells_act("cmu_mask_select_cells(out)", out, ldo, C, dt))) return rc;
    CMU_CHECK_ARG(active && (scale == nullptr) == (shift == nullptr), "cmu_mask_select_cells: null patch map, or only one of scale / shift");
    CMU_CELLS_GEOMETRY("cmu_mask_select_cells", false, 10)
    CMU_CHECK_ARG(!(ring && fill), "cmu_mask_select_cells: border-frame zeroing and a fill vector exclude each other");
    CMU_DISPATCH_DT(dt, mask_select_cells_t, x, ldx, scale, shift, relu, out, ldo, active, g, ring, fill, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// MaxPool2d(2) backward + skip-gradient add over ACTIVE patches only (cmu_maxpool_bwd_masked's contract: dA at masked positions is
// left unwritten).  Work item: rows of POOLED pixels of one patch; per pooled chunk 1 + 4 + 4 loads and 4 stores.
// ---------------------------------------------------------------------------------------------------
template <class TR, bool MULTI>
__global__ __launch_bounds__(256) void maxpool_bwd_cells_kernel(const unsigned char* __restrict__ dP, int64_t ldp,
                                                               const unsigned char* __restrict__ dS, int64_t lds,
                                                               const unsigned char* __restrict__ y, int64_t ldy, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, unsigned char* __restrict__ dA, int64_t lda,
                                                               const uint8_t* __restrict__ active, CellGeo g) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    const int tid = threadIdx.x;
    int cell0, b, y0, x0, yy0;
    cell_decode(g, blockIdx.x, cell0, b, y0, x0, yy0);        // pooled coordinates
    if (!MULTI && active[cell0] == 0) return;
    const int cmask = (1 << g.cbits) - 1, wbits = g.sbits + g.xbits, wmask = (1 << wbits) - 1;
    const int total = (1 << g.rbits) << (wbits + g.cbits);
    const int H = 2 * g.H, W = 2 * g.W;
    const int ch = tid & cmask;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[ch * EPC + e]; sh[e] = shift[ch * EPC + e]; }
    for (int k0 = tid; k0 < total; k0 += 512) {
        u32x4 gq[2], fq[2][4], dq[2][4];
        int64_t src[2][4];
        bool on[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + 256 * i;
            on[i] = k < total && (!MULTI || active[cell0 + (((k >> g.cbits) & wmask) >> g.sbits)] != 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + 256 * i;
            {
                const int j = (k >> g.cbits) & wmask, r = k >> (g.cbits + wbits);
                if (on[i]) {
                    const int yo = y0 + r, xo = x0 + j;
                    const int64_t pp = ((int64_t)b * g.H + yo) * g.W + xo;
                    gq[i] = ld_global16_nt(dP + (pp * ldp + ch * EPC) * ES);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        src[i][q] = ((int64_t)b * H + 2 * yo + (q >> 1)) * W + 2 * xo + (q & 1);
                        fq[i][q] = ld_global16(y + (src[i][q] * ldy + ch * EPC) * ES);
                        if (dS) dq[i][q] = ld_global16_nt(dS + (src[i][q] * lds + ch * EPC) * ES);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (on[i]) {
                float best[EPC], gg[EPC];
                int arg[EPC];
                TR::unpack(gq[i], gg);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float fv[EPC];
                    TR::unpack(fq[i][q], fv);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float a = fmaxf(fmaf(fv[e], sc[e], sh[e]), 0.f);
                        if (q == 0 || a > best[e]) { best[e] = a; arg[e] = q; }   // first maximum wins (ATen max_pool2d)
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float d[EPC];
                    if (dS) TR::unpack(dq[i][q], d);
                    else {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) d[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < EPC; ++e) d[e] += (arg[e] == q) ? gg[e] : 0.f;
                    st_global16(dA + (src[i][q] * lda + ch * EPC) * ES, TR::pack(d));
                }
            }
        }
    }
}
template <class TR>
static int maxpool_bwd_cells_t(const void* dP, int64_t ldp, const void* dS, int64_t lds, const void* y, int64_t ldy, const float* scale,
                               const float* shift, void* dA, int64_t lda, const uint8_t* active, CellGeo g, hipStream_t st) {
    if (g.xbits > 0)
        hipLaunchKernelGGL((maxpool_bwd_cells_kernel<TR, true>), dim3(g.nitems), dim3(256), 0, st, (const unsigned char*)dP, ldp, (const unsigned char*)dS,
                           lds, (const unsigned char*)y, ldy, scale, shift, (unsigned char*)dA, lda, active, g);
    else
        hipLaunchKernelGGL((maxpool_bwd_cells_kernel<TR, false>), dim3(g.nitems), dim3(256), 0, st, (const unsigned char*)dP, ldp, (const unsigned char*)dS,
                           lds, (const unsigned char*)y, ldy, scale, shift, (unsigned char*)dA, lda, active, g);
    CMU_CHECK_LAUNCH("cmu_maxpool_bwd_cells");
    return CMU_OK;
}
extern "C" int cmu_maxpool_bwd_cells(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* y, int64_t ldy, const float* scale,
                                     const float* shift, const uint8_t* active, int f, void* dA, int64_t lda, int B, int H, int W, int C, int dt,
                                     void* stream) {
    int rc;
    if ((rc = check_cells_act("cmu_maxpool_bwd_cells(dP)", dP, ldp, C, dt))) return rc;
    if ((rc = check_cells_act("cmu_maxpool_bwd_cells(y)", y, ldy, C, dt))) return rc;
    if ((rc = check_cells_act("cmu_maxpool_bwd_cells(dA)", dA, lda, C, dt))) return rc;
    if (dSkip && (rc = check_cells_act("cmu_maxpool_bwd_cells(dSkip)", dSkip, lds, C, dt))) return rc;
    CMU_CHECK_ARG(active && scale && shift && H % 2 == 0 && W % 2 == 0, "cmu_maxpool_bwd_cells: null argument / odd level");
    CMU_CELLS_GEOMETRY("cmu_maxpool_bwd_cells", true, 9)
    CMU_DISPATCH_DT(dt, maxpool_bwd_cells_t, dP, ldp, dSkip, lds, y, ldy, scale, shift, dA, lda, active, g, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// Per-channel sums over the pixels of the selected patches, three forms of one walk (CSUM_ROWS workgroups, each over one contiguous
// range of work items, per-thread fp32 sums of one channel chunk, fixed-order fold inside the workgroup and over the slab, no atomics:
// bitwise reproducible):
//   MODE 0  sum of x                       -> cmu_cells_channel_sum (invert = 1: the MASKED patches -- the mask-token gradient of
//                                             spark.py:104-108)
//   MODE 1  sum and sum of squares of x    -> cmu_cells_channel_stats: the sparse BatchNorm statistics (encoder.py:26-36),
//                                             slab [CSUM_ROWS][2][C] for cmu_bn_finalize
//   MODE 2  BatchNorm+ReLU backward sums   -> cmu_bn_bwd_reduce_cells (= cmu_bn_bwd_reduce_masked): sum of dz and of dz * xhat with
//                                             dz = gate * dA, into the cmu_bn_bwd_finalize workspace layout
// ---------------------------------------------------------------------------------------------------
constexpr int CSUM_ROWS = 1024;
constexpr int CSUM_HDR = 16;      // bytes in front of the MODE 2 slab (cmu_bn_bwd_ws layout: the number of rows)
template <class TR, bool MULTI, int MODE>
__global__ __launch_bounds__(256) void cells_channel_sum_kernel(const unsigned char* __restrict__ x, int64_t ldx, const unsigned char* __restrict__ y,
                                                               int64_t ldy, const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const uint8_t* __restrict__ active, int invert, CellGeo g, int per, int C,
                                                               float* __restrict__ slab) {
    constexpr int EPC = TR::EPC;
    constexpr int ES = (int)sizeof(typename TR::elem_t);
    constexpr int NS = MODE == 0 ? 1 : 2;
    __shared__ float red[256 * EPC];
    const int tid = threadIdx.x;
    const int cmask = (1 << g.cbits) - 1, nchunk = 1 << g.cbits, wbits = g.sbits + g.xbits, wmask = (1 << wbits) - 1;
    const int total = (1 << g.rbits) << (wbits + g.cbits);
    const int ch = tid & cmask;
    float s[NS][EPC];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[q][e] = 0.f;
    float sc[EPC], sh[EPC], mu[EPC], is[EPC];
    if (MODE == 2) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = ch * EPC + e;
            sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
        }
    }
    const int w1 = (blockIdx.x + 1) * per < g.nitems ? (blockIdx.x + 1) * per : g.nitems;
    for (int wi = blockIdx.x * per; wi < w1; ++wi) {
        int cell0, b, y0, x0, yy0;
        cell_decode(g, wi, cell0, b, y0, x0, yy0);
        if (!MULTI && (active[cell0] != 0) == (invert != 0)) continue;
        const int64_t row0 = (int64_t)b * g.H + y0;
        for (int k0 = tid; k0 < total; k0 += 1024) {
            u32x4 vq[4], gq[4];
            bool on[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 256 * i;
                on[i] = k < total && (!MULTI || (active[cell0 + (((k >> g.cbits) & wmask) >> g.sbits)] != 0) != (invert != 0));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 256 * i;
                const int j = (k >> g.cbits) & wmask, r = k >> (g.cbits + wbits);
                if (on[i]) {
                    const int64_t p = (row0 + r) * g.W + x0 + j;
                    if (MODE == 2) {
                        vq[i] = ld_global16(x + (p * ldx + ch * EPC) * ES);
                        gq[i] = ld_global16(y + (p * ldy + ch * EPC) * ES);
                    } else {
                        vq[i] = MODE == 0 ? ld_global16_nt(x + (p * ldx + ch * EPC) * ES) : ld_global16(x + (p * ldx + ch * EPC) * ES);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (on[i]) {
                    float v[EPC];
                    TR::unpack(vq[i], v);
                    if (MODE == 0) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) s[0][e] += v[e];
                    } else if (MODE == 1) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            s[0][e] += v[e];
                            s[NS - 1][e] = fmaf(v[e], v[e], s[NS - 1][e]);
                        }
                    } else {                       // x = dA, y = the raw output
                        float yv[EPC];
                        TR::unpack(gq[i], yv);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            const float dz = fmaf(yv[e], sc[e], sh[e]) > 0.f ? v[e] : 0.f;
                            s[0][e] += dz;
                            s[NS - 1][e] = fmaf(dz, (yv[e] - mu[e]) * is[e], s[NS - 1][e]);
                        }
                    }
                }
            }
        }
    }
    if (MODE == 2 && blockIdx.x == 0 && tid == 0) *reinterpret_cast<int*>(slab) = (int)gridDim.x;
    float* out = MODE == 2 ? slab + CSUM_HDR / 4 : slab;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[tid * EPC + e] = s[q][e];
        __syncthreads();
        if (tid < nchunk) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float a = 0.f;
                for (int k = tid; k < 256; k += nchunk) a += red[k * EPC + e];
                out[((int64_t)blockIdx.x * NS + q) * C + tid * EPC + e] = a;
            }
        }
    }
}
// 16 channels x 16 row parts per block, eight rows in flight per thread, fixed-order combine in double
__global__ __launch_bounds__(256) void cells_channel_sum_final_kernel(const float* __restrict__ slab, int nrows, int C, float* __restrict__ out) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < C) {
        constexpr int U = 8;
        int b = part;
        for (; b + 16 * (U - 1) < nrows; b += 16 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = slab[(int64_t)(b + 16 * u) * C + c];
#pragma unroll
            for (int u = 0; u < U; ++u) s += (double)v[u];
        }
        for (; b < nrows; b += 16) s += (double)slab[(int64_t)b * C + c];
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
// launches the walk with exactly `rows` workgroups (every slab row is written; workgroups past the work write zeros)
template <class TR, int MODE>
static int cells_sums_launch(const void* x, int64_t ldx, const void* y, int64_t ldy, const float* scale, const float* shift, const float* mean,
                             const float* invstd, const uint8_t* active, int invert, CellGeo g, int C, float* slab, int rows, hipStream_t st) {
    const int per = (g.nitems + rows - 1) / rows;
    if (g.xbits > 0)
        hipLaunchKernelGGL((cells_channel_sum_kernel<TR, true, MODE>), dim3(rows), dim3(256), 0, st, (const unsigned char*)x, ldx, (const unsigned char*)y,
                           ldy, scale, shift, mean, invstd, active, invert, g, per, C, slab);
    else
        hipLaunchKernelGGL((cells_channel_sum_kernel<TR, false, MODE>), dim3(rows), dim3(256), 0, st, (const unsigned char*)x, ldx, (const unsigned char*)y,
                           ldy, scale, shift, mean, invstd, active, invert, g, per, C, slab);
    return CMU_OK;
}
static int csum_rows(const CellGeo& g) { return g.nitems < CSUM_ROWS ? g.nitems : CSUM_ROWS; }
template <class TR>
static int cells_channel_sum_t(const void* x, int64_t ldx, const uint8_t* active, int invert, CellGeo g, int C, float* out, float* ws, hipStream_t st) {
    const int rows = csum_rows(g);
    cells_sums_launch<TR, 0>(x, ldx, nullptr, 0, nullptr, nullptr, nullptr, nullptr, active, invert, g, C, ws, rows, st);
    CMU_CHECK_LAUNCH("cmu_cells_channel_sum");
    hipLaunchKernelGGL(cells_channel_sum_final_kernel, dim3(cmu_div_up(C, 16)), dim3(256), 0, st, (const float*)ws, rows, C, out);
    CMU_CHECK_LAUNCH("cmu_cells_channel_sum(final)");
    return CMU_OK;
}
extern "C" int64_t cmu_cells_channel_sum_ws_bytes(int C) { return (int64_t)CSUM_ROWS * C * (int64_t)sizeof(float); }
extern "C" int cmu_cells_channel_sum(const void* x, int64_t ldx, const uint8_t* active, int f, int invert, float* out, void* ws, int B, int H, int W,
                                     int C, int dt, void* stream) {
    int rc;
    if ((rc = check_cells_act("cmu_cells_channel_sum(x)", x, ldx, C, dt))) return rc;
    CMU_CHECK_ARG(active && out && ws, "cmu_cells_channel_sum: null argument");
    CMU_CELLS_GEOMETRY("cmu_cells_channel_sum", false, 10)
    CMU_DISPATCH_DT(dt, cells_channel_sum_t, x, ldx, active, invert, g, C, out, (float*)ws, (hipStream_t)stream);
}

template <class TR>
static int cells_channel_stats_t(const void* x, int64_t ldx, const uint8_t* active, CellGeo g, int C, float* slab, hipStream_t st) {
    cells_sums_launch<TR, 1>(x, ldx, nullptr, 0, nullptr, nullptr, nullptr, nullptr, active, 0, g, C, slab, CSUM_ROWS, st);
    CMU_CHECK_LAUNCH("cmu_cells_channel_stats");
    return CMU_OK;
}
extern "C" int cmu_cells_stats_rows(void) { return CSUM_ROWS; }
extern "C" int cmu_cells_channel_stats(const void* x, int64_t ldx, const uint8_t* active, int f, float* slab, int B, int H, int W, int C, int dt,
                                       void* stream) {
    int rc;
    if ((rc = check_cells_act("cmu_cells_channel_stats(x)", x, ldx, C, dt))) return rc;
    CMU_CHECK_ARG(active && slab, "cmu_cells_channel_stats: null argument");
    CMU_CELLS_GEOMETRY("cmu_cells_channel_stats", false, 10)
    CMU_DISPATCH_DT(dt, cells_channel_stats_t, x, ldx, active, g, C, slab, (hipStream_t)stream);
}

template <class TR>
static int bn_bwd_reduce_cells_t(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, const uint8_t* active, CellGeo g, int C, float* ws, hipStream_t st) {
    cells_sums_launch<TR, 2>(dA, ldd, y, ldy, scale, shift, mean, invstd, active, 0, g, C, ws, csum_rows(g), st);
    CMU_CHECK_LAUNCH("cmu_bn_bwd_reduce_cells");
    return CMU_OK;
}
extern "C" int cmu_bn_bwd_reduce_cells(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                                       const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                                       const uint8_t* active, int f, int64_t count, int B, int H, int W, int C, int dt, void* ws, void* stream) {
    int rc;
    if ((rc = check_cells_act("cmu_bn_bwd_reduce_cells(dA)", dA, ldd, C, dt))) return rc;
    if ((rc = check_cells_act("cmu_bn_bwd_reduce_cells(y)", y, ldy, C, dt))) return rc;
    CMU_CHECK_ARG(scale && shift && save_mean && save_invstd && coef && ws && active && count > 0, "cmu_bn_bwd_reduce_cells: null argument");
    CMU_CELLS_GEOMETRY("cmu_bn_bwd_reduce_cells", false, 10)
    {
        const int es = cmu_dtype_size(dt);
        int r2 = es == 4 ? bn_bwd_reduce_cells_t<F32Traits>(dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, active, g, C, (float*)ws, (hipStream_t)stream)
                 : dt == CMU_F16 ? bn_bwd_reduce_cells_t<F16Traits>(dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, active, g, C, (float*)ws, (hipStream_t)stream)
                                 : bn_bwd_reduce_cells_t<BF16Traits>(dA, ldd, y, ldy, scale, shift, save_mean, save_invstd, active, g, C, (float*)ws, (hipStream_t)stream);
        if (r2 != CMU_OK) return r2;
    }
    return cmu_bn_bwd_finalize(ws, count, dgamma, dbeta, coef, C, stream);      // ws: cmu_bn_bwd_ws_bytes(C)
}
