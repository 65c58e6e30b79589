// store_probe.hip -- how fast does ONE CU drain a burst of epilogue stores?  (round 5: the 16x16x32 conv kernel's epilogue issues 16 x 1 KiB
// store instructions per wave; the next weight load behind them waits until they have retired -- vmcnt retires in order.)
// Each workgroup (one per CU, 256 or 512 threads) issues `n` 16-byte stores per lane in one of these shapes and times issue and drain with s_memtime:
//   0  64-byte pieces: 4 lanes x 16 B contiguous per pixel, 16 pixels per instruction at a stride of `ld` bytes (what conv_igemm5 writes)
//   1  full lines: 8 lanes x 16 B = 128 B per pixel, 8 pixels per instruction
//   2  1 KiB contiguous per instruction
// hipcc -O3 --offload-arch=gfx950 -o tools/_diag/store_probe tools/store_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int SHAPE, bool NT>
__global__ __launch_bounds__(512) void probe(unsigned char* out, int ld, int nst, long long wg_stride, unsigned long long* stamps, int rounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* base = out + (long long)blockIdx.x * wg_stride + (long long)wave * 16 * 16 * ld;   // each wave: its own 16 pixel rows
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    unsigned long long t_issue = 0, t_drain = 0;
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < nst; ++i) {
            unsigned off;
            if (SHAPE == 0) off = (unsigned)(((i >> 1) * 16 + (lane & 15)) * ld + (i & 1) * 64 + (lane >> 4) * 16);
            else if (SHAPE == 1) off = (unsigned)((i * 8 + (lane >> 3)) * ld + (lane & 7) * 16);
            else if (SHAPE == 3) {
                // what conv_igemm5's epilogue writes: row i of the wave's 8 x 16-pixel block of a 256-pixel-wide, 128-channel image (pixel 256 B,
                // image row 64 KB), two full-line instructions per row (8 pixels x 128 B each); wave w: rows 8 (w >> 1).., columns 16 (w & 1)..
                const int row = 8 * (wave >> 1) + (i >> 1), col = 16 * (wave & 1) + 8 * (i & 1) + (lane >> 3);
                off = (unsigned)(row * 65536 + col * 256 + (lane & 7) * 16);      // relative to the workgroup's region (< 32 * 65536 = 2 MiB)
            } else off = (unsigned)(i * 1024 + lane * 16);
            u32x4* p = reinterpret_cast<u32x4*>((SHAPE == 3 ? base - (long long)wave * 16 * 16 * ld : base) + off);
            if (NT) __builtin_nontemporal_store(v, p);
            else *p = v;
            v[0] += 1;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_issue += t1 - t0;
        t_drain += t2 - t0;
        base += 64;   // another 64-byte column of the same lines next round (stays inside the row for ld >= 256)
        if ((r & 3) == 3) base += 8 * 16 * 16 * (long long)ld - 256;
    }
    if (lane == 0) {
        stamps[(blockIdx.x * 8 + wave) * 2 + 0] = t_issue / rounds;
        stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t_drain / rounds;
    }
}

template <int SHAPE, bool NT>
static void run(const char* name, int threads, int ld, int nst, int grid, unsigned char* buf, long long wg_stride, unsigned long long* dstamps) {
    const int rounds = 16;
    hipLaunchKernelGGL((probe<SHAPE, NT>), dim3(grid), dim3(threads), 0, 0, buf, ld, nst, wg_stride, dstamps, rounds);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE, NT>), dim3(grid), dim3(threads), 0, 0, buf, ld, nst, wg_stride, dstamps, rounds);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 16);
    (void)hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> is, dr;
    const int waves = threads / 64;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < waves; ++w) { is.push_back(h[(b * 8 + w) * 2]); dr.push_back(h[(b * 8 + w) * 2 + 1]); }
    std::sort(is.begin(), is.end()); std::sort(dr.begin(), dr.end());
    const double bytes_cu = (double)waves * nst * 1024;
    printf("%-34s %3d thr  grid %3d  ld %4d  %2d stores/lane: issue %6llu  drain median %6llu  max %6llu cycles  -> %5.1f B/clk/CU   (%.2f TB/s over the launch)\n", name, threads, grid,
           ld, nst, is[is.size() / 2], dr[dr.size() / 2], dr.back(), bytes_cu / (double)dr[dr.size() / 2], bytes_cu * grid * rounds / (ms * 1e-3) / 1e12);
}

int main() {
    const long long wg_stride = 8ll << 20;
    unsigned char* buf;
    unsigned long long* dstamps;
    (void)hipMalloc(&buf, 256 * wg_stride);
    (void)hipMalloc(&dstamps, 256 * 16 * 8);
    (void)hipMemset(buf, 0, 256 * wg_stride);
    for (int grid : {1, 32, 256}) {
        for (int thr : {256, 512}) {
            run<0, true>("64-byte pieces, nt", thr, 256, 16, grid, buf, wg_stride, dstamps);
            run<0, false>("64-byte pieces", thr, 256, 16, grid, buf, wg_stride, dstamps);
            run<1, true>("full lines, nt", thr, 256, 16, grid, buf, wg_stride, dstamps);
            run<1, false>("full lines", thr, 256, 16, grid, buf, wg_stride, dstamps);
            run<3, true>("conv_igemm5 rows (64 KB apart), nt", thr, 256, 16, grid, buf, wg_stride, dstamps);
            run<3, false>("conv_igemm5 rows (64 KB apart)", thr, 256, 16, grid, buf, wg_stride, dstamps);
            run<2, true>("1 KiB contiguous, nt", thr, 256, 16, grid, buf, wg_stride, dstamps);
            run<2, false>("1 KiB contiguous", thr, 256, 16, grid, buf, wg_stride, dstamps);
        }
        run<0, true>("64-byte pieces, nt, ld 1024", 256, 1024, 16, grid, buf, wg_stride, dstamps);
        run<1, true>("full lines, nt, ld 1024", 256, 1024, 16, grid, buf, wg_stride, dstamps);
        run<0, true>("64-byte pieces, nt, 4 stores", 256, 256, 4, grid, buf, wg_stride, dstamps);
    }
    return 0;
}
