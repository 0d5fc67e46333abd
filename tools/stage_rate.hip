// How fast can ONE CU pull L2-resident bytes into LDS?  The GraphLayer GEMM's time tracks the bytes it stages (DESIGN.md
// section 8), so this measures the staging path alone: every wave loops over 1-KiB pieces of a small window (all workgroups
// share it: L2 hits, larger than the 32 KB L1) and lands them in LDS, no MFMA, no barrier.
//   mode 0: LDS-DMA (global_load_lds_dwordx4), counted vmcnt
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only (the vector memory path without the LDS write)
// pattern 0: a piece = 8 rows x 128 B, rows 4 KiB apart (a k-tile of a K = 2048 bf16 operand); 1: 1 KiB contiguous
// build: hipcc --offload-arch=gfx950 -O3 tools/stage_rate.hip -o tools/stage_rate ; run: tools/stage_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ inline void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}

template <int MODE, int DEPTH>  // DEPTH pieces in flight per wave
__global__ __launch_bounds__(512) void stage_kernel(const unsigned char* __restrict__ win, size_t win_bytes, int iters, int pattern,
                                                    float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* mine = smem + wave * (DEPTH * 1024);
    // piece q: strided -> rows 8*(q/32) .. +7 of the window seen as 4 KiB rows, 128-byte column q % 32 (consecutive pieces walk
    // the k axis of the same 8 rows, as a GEMM's k loop does); contiguous -> KiB q of the window
    const unsigned npieces = (unsigned)(win_bytes / 1024);
    unsigned q = ((blockIdx.x * 8 + wave) * 7919u) % npieces;
    const size_t lane_off = pattern == 0 ? (size_t)(lane >> 3) * 4096 + (lane & 7) * 16 : (size_t)lane * 16;
    uint4 r[DEPTH];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t base = pattern == 0 ? (size_t)(q >> 5) * 32768 + (q & 31) * 128 : (size_t)q * 1024;
            const unsigned char* src = win + base + lane_off;
            if (MODE == 0) dma16(src, mine + d * 1024);
            else r[d] = *reinterpret_cast<const uint4*>(src);
            q = q + 1 == npieces ? 0 : q + 1;
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += *reinterpret_cast<const float*>(mine + lane * 4);
        } else if (MODE == 1) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<uint4*>(mine + d * 1024 + lane * 16) = r[d];
            acc += *reinterpret_cast<const float*>(mine + lane * 4);
        } else {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += __uint_as_float(r[d].x ^ r[d].w);
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

template <int MODE, int DEPTH>
static void run(const unsigned char* win, size_t win_bytes, int wgs_per_cu, int pattern, float* sink, int cus) {
    const int iters = 4000 / DEPTH;
    const int grid = cus * wgs_per_cu;
    const size_t lds = 8 * DEPTH * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stage_kernel<MODE, DEPTH>), dim3(grid), dim3(512), lds, 0, win, win_bytes, iters, pattern, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)grid * 8 * iters * DEPTH * 1024;
    printf("mode %d depth %2d wgs/cu %d pattern %d: %7.3f ms  %6.2f TB/s chip  %6.1f GB/s per CU (%.1f B/clk at 2.1 GHz)\n", MODE, DEPTH,
           wgs_per_cu, pattern, best, bytes / best * 1e-9, bytes / best * 1e-6 / cus, bytes / best * 1e-6 / cus / 2.1);
}

int main(int argc, char** argv) {
    const size_t win_bytes = (argc > 1 ? atoi(argv[1]) : 128) * 1024;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned char* win;
    float* sink;
    hipMalloc(&win, win_bytes + 65536);
    hipMemset(win, 1, win_bytes + 65536);
    hipMalloc(&sink, 4);
    printf("%s, %d CUs, window %zu KiB shared by all workgroups\n", prop.name, cus, win_bytes / 1024);
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int w = 1; w <= 2; ++w) {
            run<0, 2>(win, win_bytes, w, pattern, sink, cus);
            run<0, 4>(win, win_bytes, w, pattern, sink, cus);
            run<0, 8>(win, win_bytes, w, pattern, sink, cus);
            run<1, 4>(win, win_bytes, w, pattern, sink, cus);
            run<1, 8>(win, win_bytes, w, pattern, sink, cus);
            run<2, 4>(win, win_bytes, w, pattern, sink, cus);
            run<2, 8>(win, win_bytes, w, pattern, sink, cus);
        }
    return 0;
}
