// moco.hip -- MoCo-v2 InfoNCE against the momentum queue + ring-buffer enqueue as ONE launch
// (Pretraining/MoCo/pl_bolts/models/self_supervised/moco/moco2_module.py:160-175 _dequeue_and_enqueue, :256-285 logits / CE).
//
//   logits = [q.k, q @ queue] / T  (N x (1+K), label 0)   loss = CE(logits)   dq = d loss / d q_raw
//   queue[:, ptr : ptr+Nk] = keys^T ; ptr = (ptr + Nk) % K   -- AFTER the logits were taken from the old queue (SURVEY A-8)
//
// The first form of this kernel gave one workgroup per query row (32 of 256 CUs) two passes over the whole (D, K) queue with a
// thread per column, and let a single workgroup enqueue while recomputing each key's norm per element: 19.9 ms per step at
// K = 4096, D = 1024 -- as long as both encoders together.  The work is two skinny GEMMs around a softmax:
//   P1  L (B x K)  = qn (B x D) . queue (D x K)             [queue rows contiguous in j: the "input gradient" shape of skinny.hip]
//   P2  row softmax statistics, loss, p = softmax / (B T)   [one workgroup per row]
//   P3  G (B x D)  = p (B x K) . queue^T                     [the "forward" shape of skinny.hip, split over K]
//   P4  dq from G, the enqueue, the pointer and the loss
// (P0 normalises the rows).  Here ONE persistent grid (one workgroup per CU at most) runs the phases back to back with a
// grid-wide barrier between them (agent-scope atomics + fences; every workgroup is resident: grid <= CUs and 66 KB of LDS, so
// the spin always ends), each phase spread over all workgroups, fp32 products on v_mfma_f32_32x32x2_f32 with 16-byte loads
// straight into the operand registers.  Cross-workgroup sums go through slabs in a fixed order: bitwise reproducible.
// 16 MB of queue are read twice: tens of microseconds instead of 20 ms.
#include "common.h"

typedef float f32x4m __attribute__((ext_vector_type(4)));

struct MocoParams {
    const float* q_raw;
    const float* k_raw;
    const float* keys_all;
    int Nk;
    float* queue;
    int64_t* queue_ptr;
    float* loss;
    float* dq;
    float* k_norm_out;
    int B, D, K;
    float temp;
    // workspace
    unsigned* bar;     // 8 counters, zeroed before the launch
    float* qn;         // [B][D]
    float* kn;         // [B][D]
    float* qnT;        // [RT][D][32]   qn transposed per 32-row tile (P1's A operand: contiguous 128-byte reads)
    float* rowst;      // [B][4]  qnorm, pos, gpos, loss term
    float* L;          // [B][K]  logits / T, then p
    float* slab;       // [splits][B][D]
    int splits;
    int64_t kchunk;
};

// Grid barrier between the phases.  It only ends when every workgroup of the grid is resident at once: the launcher sizes the grid
// to at most one workgroup per CU (66 KB LDS, 256 threads), which holds on a stream that owns the whole device -- NOT under a CU mask
// narrower than the CU count (tools/cu_partition.py's masked streams) or beside another resident kernel that holds the CUs it needs
// (an RCCL kernel on the process-group stream, several ranks sharing one card): do not run this entry concurrently with other
// kernels.  So that such a misuse fails instead of hanging the GPU, the spin is BOUNDED (~2 s of s_sleep polls): on expiry the
// workgroup raises bar[7] and leaves the barrier; every workgroup reads that flag behind the last barrier and the launch then
// ends with loss = NaN, dq = 0 (nothing computed on incomplete data reaches the optimiser), and the queue and its pointer
// UNTOUCHED (advisor, round 3: they used to take the garbage keys).  pretrain.MocoPretrainer raises on the NaN loss.
__device__ static inline void moco_grid_barrier(unsigned* counter, unsigned target, unsigned* timeout_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21) || __hip_atomic_load(timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence();   // every wave: what the other workgroups wrote before the barrier is visible to its loads
}

__device__ static inline float moco_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ static inline float moco_block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ static inline f32x4m moco_ld4(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const f32x4m*>(p) : f32x4m{0.f, 0.f, 0.f, 0.f};
}

__global__ __launch_bounds__(256) void moco_fused_kernel(const MocoParams p) {
    __shared__ float red[4][4][16][64];     // P1: the four waves' partial tiles (64 KB)
    float* red4 = &red[0][0][0][0];         // block reductions of the other phases (never live together with the tiles)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int B = p.B, D = p.D, K = p.K;
    const int RT = (B + 31) / 32;
    const unsigned G = gridDim.x;
    int ptr0 = (int)p.queue_ptr[0];   // read before anything in this launch can move it (the write is in the last phase)
    if (ptr0 < 0 || ptr0 >= p.K) ptr0 = 0;   // (a corrupted pointer cannot send the enqueue outside the queue)

    // ---- P0: row norms, normalised rows, the positive logit ---------------------------------------------------------------------
    for (int b = blockIdx.x; b < B; b += G) {
        const float* q = p.q_raw + (int64_t)b * D;
        const float* kk = p.k_raw + (int64_t)b * D;
        float s = 0.f, s2 = 0.f;
        for (int d = tid; d < D; d += 256) {
            s = fmaf(q[d], q[d], s);
            s2 = fmaf(kk[d], kk[d], s2);
        }
        const float qnorm = fmaxf(sqrtf(moco_block_sum(s, red4)), 1e-12f);
        const float knorm = fmaxf(sqrtf(moco_block_sum(s2, red4)), 1e-12f);
        float pos = 0.f;
        for (int d = tid; d < D; d += 256) {
            const float a = q[d] / qnorm, k_ = kk[d] / knorm;
            p.qn[(int64_t)b * D + d] = a;
            p.kn[(int64_t)b * D + d] = k_;
            p.qnT[((int64_t)(b >> 5) * D + d) * 32 + (b & 31)] = a;
            if (p.k_norm_out) p.k_norm_out[(int64_t)b * D + d] = k_;
            pos = fmaf(a, k_, pos);
        }
        pos = moco_block_sum(pos, red4) / p.temp;
        if (tid == 0) {
            p.rowst[b * 4 + 0] = qnorm;
            p.rowst[b * 4 + 1] = pos;
        }
    }
    // rows of a partial last tile that do not exist: zero columns of qnT
    for (int64_t o = (int64_t)blockIdx.x * 256 + tid; o < (int64_t)(RT * 32 - B) * D; o += (int64_t)G * 256) {
        const int m = B + (int)(o / D), d = (int)(o % D);
        p.qnT[((int64_t)(m >> 5) * D + d) * 32 + (m & 31)] = 0.f;
    }
    moco_grid_barrier(p.bar + 0, G, p.bar + 7);

    // ---- P1: L = qn . queue / T.  Work item = (row tile, 128 columns j): lane c holds columns 4c .. 4c+3 of four MFMA tiles, the four
    // waves take a quarter of the D rows each and are summed through LDS in wave order ------------------------------------------
    {
        const int ncol = (K + 127) / 128;
        const int dq4 = ((D + 3) / 4 + 1) & ~1;        // rows per wave (even: d pairs)
        for (int it = blockIdx.x; it < ncol * RT; it += G) {
            const int rt = it / ncol, cb = it % ncol;
            const int64_t jcol = (int64_t)cb * 128 + 4 * c;
            const bool jok = jcol < K;
            const int dbeg = wave * dq4, dend = dbeg + dq4 < D ? dbeg + dq4 : D;
            const float* aT = p.qnT + (int64_t)rt * D * 32;
            f32x16 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            constexpr int U = 8;
            for (int d0 = dbeg; d0 < dend; d0 += 2 * U) {
                f32x4m wv[U];
                float av[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int d = d0 + 2 * u + h;
                    wv[u] = moco_ld4(p.queue + (int64_t)(d < dend ? d : 0) * K + jcol, jok && d < dend);
                    av[u] = d < dend ? aT[(int64_t)d * 32 + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wv[u][j], acc[j], 0, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[wave][j][e][lane] = acc[j][e];
            __syncthreads();
            if (jok) {
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const int e = 4 * wave + ee;
                    const int m = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    f32x4m v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (((red[0][j][e][lane] + red[1][j][e][lane]) + red[2][j][e][lane]) + red[3][j][e][lane]) / p.temp;
                    if (m < B) *reinterpret_cast<f32x4m*>(p.L + (int64_t)m * K + jcol) = v;
                }
            }
        }
    }
    moco_grid_barrier(p.bar + 1, G, p.bar + 7);

    // ---- P2: per row: log-sum-exp over [pos, L[b][:]], the row's loss term, p = softmax / (B T) in place ----------------------------
    for (int b = blockIdx.x; b < B; b += G) {
        float* Lr = p.L + (int64_t)b * K;
        const float pos = p.rowst[b * 4 + 1];
        float m = pos;
        for (int j = tid; j < K; j += 256) m = fmaxf(m, Lr[j]);
        m = moco_block_max(m, red4);
        float se = 0.f;
        for (int j = tid; j < K; j += 256) se += expf(Lr[j] - m);
        se = moco_block_sum(se, red4) + expf(pos - m);
        const float inv = 1.f / (se * (float)B * p.temp);
        for (int j = tid; j < K; j += 256) Lr[j] = expf(Lr[j] - m) * inv;
        if (tid == 0) {
            p.rowst[b * 4 + 2] = (expf(pos - m) / se - 1.f) / ((float)B * p.temp);   // d loss / d (q.k) incl. 1/T
            p.rowst[b * 4 + 3] = (m + logf(se) - pos) / (float)B;                     // CE with label 0, mean over the batch
        }
    }
    moco_grid_barrier(p.bar + 2, G, p.bar + 7);

    // ---- P3: G = p . queue^T, split over K.  Work item = (row tile, 128 rows d of the queue, K range): wave = 32 rows ---------------
    if (p.dq != nullptr) {
        const int nblk = (D + 127) / 128;
        constexpr int U = 4;
        for (int it = blockIdx.x; it < nblk * p.splits * RT; it += G) {
            const int rt = it / (nblk * p.splits), rem = it % (nblk * p.splits);
            const int sp = rem / nblk, nb = rem % nblk;
            const int n = (nb * 4 + wave) * 32 + c;                  // queue row d of this lane
            const int m = rt * 32 + c;                               // row of p of this lane
            const int64_t k0 = (int64_t)sp * p.kchunk, k1 = k0 + p.kchunk < K ? k0 + p.kchunk : K;
            const bool nok = n < D, mok = m < B;
            const float* wp = p.queue + (int64_t)(nok ? n : 0) * K + 4 * h;
            const float* xp = p.L + (int64_t)(mok ? m : 0) * K + 4 * h;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            for (int64_t kb = k0; kb < k1; kb += 8 * U) {
                f32x4m wv[U], xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t k = kb + u * 8;
                    wv[u] = moco_ld4(wp + k, nok && k + 4 * h < k1);       // (k1 and k + 4h are multiples of 4)
                    xv[u] = moco_ld4(xp + k, mok && k + 4 * h < k1);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u][j], wv[u][j], acc, 0, 0, 0);
            }
            // D[m][n]: lane holds column n (its queue row d), rows m = (e & 3) + 8 (e >> 2) + 4 h
            if (nok) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int mm = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (mm < B) p.slab[((int64_t)sp * B + mm) * D + n] = acc[e];
                }
            }
        }
    }
    moco_grid_barrier(p.bar + 3, G, p.bar + 7);

    // ---- P4: dq = (I - qn qn^T) (gpos kn + G) / |q|;  enqueue;  pointer;  loss ------------------------------------------------------
    // (a barrier that timed out anywhere in the grid: every workgroup sees the flag here -- it is raised before the workgroup that
    // gave up arrives at the counters the others wait on, and a waiting workgroup leaves its spin as soon as it is raised)
    const bool timed_out = __hip_atomic_load(p.bar + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (timed_out) {
        if (p.dq != nullptr)
            for (int64_t o = (int64_t)blockIdx.x * 256 + tid; o < (int64_t)B * D; o += (int64_t)G * 256) p.dq[o] = 0.f;
        if (blockIdx.x == 0 && tid == 0) p.loss[0] = __builtin_nanf("");
        return;
    }
    if (p.dq != nullptr) {
        for (int b = blockIdx.x; b < B; b += G) {
            const float qnorm = p.rowst[b * 4 + 0], gpos = p.rowst[b * 4 + 2];
            float dot = 0.f;
            for (int d = tid; d < D; d += 256) {
                float g = gpos * p.kn[(int64_t)b * D + d];
                for (int s = 0; s < p.splits; ++s) g += p.slab[((int64_t)s * B + b) * D + d];
                p.dq[(int64_t)b * D + d] = g;                       // (dqn; finished below)
                dot = fmaf(g, p.qn[(int64_t)b * D + d], dot);
            }
            dot = moco_block_sum(dot, red4);
            for (int d = tid; d < D; d += 256) p.dq[(int64_t)b * D + d] = (p.dq[(int64_t)b * D + d] - p.qn[(int64_t)b * D + d] * dot) / qnorm;
        }
    }
    // queue[:, ptr + i] = key_i (moco2_module.py:172): every reader of the old queue is past the last barrier
    {
        const float* keys = p.keys_all != nullptr ? p.keys_all : p.kn;
        const int Nk = p.Nk;
        for (int64_t o = (int64_t)blockIdx.x * 256 + tid; o < (int64_t)Nk * D; o += (int64_t)G * 256) {
            const int d = (int)(o / Nk), i = (int)(o % Nk);        // consecutive threads -> consecutive columns of one queue row
            // (ring semantics: a pointer that is not a multiple of the batch -- a batch size that changed between calls, which the
            // reference refuses with a shape error at moco2_module.py:172 and Moco_v2.training_step refuses on the host -- wraps
            // instead of writing into the next row or past the buffer)
            int col = ptr0 + i;
            if (col >= K) col -= K;
            p.queue[(int64_t)d * K + col] = keys[(int64_t)i * D + d];
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        double tot = 0.0;
        for (int r = 0; r < B; ++r) tot += (double)p.rowst[r * 4 + 3];
        p.loss[0] = (float)tot;
        p.queue_ptr[0] = (int64_t)((ptr0 + p.Nk) % K);
    }
}

static int moco_splits(int D, int K, int64_t* kchunk) {
    const int nblk = cmu_div_up(D, 128);
    int splits = cmu_div_up(512, nblk);                           // ~2 work items per CU
    int64_t kc = cmu_div_up64(K, splits);
    kc = cmu_div_up64(kc, 32) * 32;
    if (kc < 32) kc = 32;
    *kchunk = kc;
    return (int)cmu_div_up64(K, kc);
}
static int64_t moco_align(int64_t n) { return (n + 63) / 64 * 64; }   // floats -> 256-byte aligned sections

extern "C" int64_t cmu_moco_ws_bytes(int B, int D, int K) {
    if (B <= 0 || D <= 0 || K <= 0) return -1;
    int64_t kc;
    const int splits = moco_splits(D, K, &kc);
    const int RT = cmu_div_up(B, 32);
    const int64_t fl = 64 + 2 * moco_align((int64_t)B * D) + moco_align((int64_t)RT * D * 32) + moco_align((int64_t)B * 4) +
                       moco_align((int64_t)B * K) + moco_align((int64_t)splits * B * D);
    return fl * (int64_t)sizeof(float);
}

extern "C" int cmu_moco_infonce_enqueue(const float* q_raw, const float* k_raw, const float* keys_all, int Nk, float* queue,
                                        int64_t* queue_ptr, float* loss, float* dq, float* k_norm_out, int B, int D, int K,
                                        float temperature, void* ws, void* stream) {
    CMU_CHECK_ARG(q_raw && k_raw && queue && queue_ptr && loss && ws && B > 0 && D > 0 && K > 0 && temperature > 0.f,
                  "cmu_moco_infonce_enqueue: bad args");
    if (!keys_all) Nk = B;
    CMU_CHECK_ARG(Nk > 0 && K % Nk == 0, "cmu_moco_infonce_enqueue: K=%d must be a multiple of the gathered batch %d (moco2_module.py:169)", K, Nk);
    CMU_CHECK_ARG(K % 4 == 0 && cmu_aligned16(queue) && cmu_aligned16(ws), "cmu_moco_infonce_enqueue: K %% 4 == 0 and 16-byte aligned queue / ws");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ws, 0, 64 * sizeof(float), st);
    if (e != hipSuccess) { cmu_set_error("cmu_moco_infonce_enqueue: memset: %s", hipGetErrorString(e)); return CMU_ERR_LAUNCH; }
    MocoParams p;
    p.q_raw = q_raw; p.k_raw = k_raw; p.keys_all = keys_all; p.Nk = Nk; p.queue = queue; p.queue_ptr = queue_ptr; p.loss = loss; p.dq = dq;
    p.k_norm_out = k_norm_out; p.B = B; p.D = D; p.K = K; p.temp = temperature;
    p.splits = moco_splits(D, K, &p.kchunk);
    const int RT = cmu_div_up(B, 32);
    float* w = (float*)ws;
    p.bar = (unsigned*)w; w += 64;
    p.qn = w; w += moco_align((int64_t)B * D);
    p.kn = w; w += moco_align((int64_t)B * D);
    p.qnT = w; w += moco_align((int64_t)RT * D * 32);
    p.rowst = w; w += moco_align((int64_t)B * 4);
    p.L = w; w += moco_align((int64_t)B * K);
    p.slab = w;
    // one workgroup per CU at most: all of them are resident (66 KB LDS, 256 threads), which the grid barriers rely on
    int cus = 0, dev = cmu_current_device();
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 64;
    const int want = cmu_div_up(K, 128) * RT > cmu_div_up(D, 128) * p.splits * RT ? cmu_div_up(K, 128) * RT : cmu_div_up(D, 128) * p.splits * RT;
    int grid = want < cus ? want : cus;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(moco_fused_kernel, dim3(grid), dim3(256), 0, st, p);
    CMU_CHECK_LAUNCH("cmu_moco_infonce_enqueue");
    return CMU_OK;
}
