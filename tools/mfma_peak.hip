// Where is the matrix-pipe ceiling of THIS chip for the two bf16 MFMA shapes, sustained (power-limited clock included)?
// Settles item 4(i) of the round-1 review: is the 1.56 PFLOP/s of the wide kernel's MFMA-only ablation (16x16x32) a property
// of the shape or of the chip?  Register-only loops: no LDS, no memory, independent accumulators, 1 or 2 waves per SIMD,
// launches of 0.2 ms .. 20 ms (a short launch runs at boost clock, a long one at the clock the power limit allows).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak ; run: tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// wave tile 64 x 128 as the wide kernel has it: 4 A fragments x 8 B fragments of 16x16x32 = 32 accumulators of 4 registers
__global__ __launch_bounds__(512) void k16(int iters, float* sink, const bf16x8* src) {
    bf16x8 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = src[threadIdx.x + 64 * i];
    for (int i = 0; i < 8; ++i) b[i] = src[threadIdx.x + 64 * (4 + i)];
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) *sink = s;
}
// the fp16 instruction on the same registers (the operand bits are then read as fp16: random sign / mantissa around 1)
__global__ __launch_bounds__(512) void k16h(int iters, float* sink, const bf16x8* src) {
    bf16x8 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = src[threadIdx.x + 64 * i];
    for (int i = 0; i < 8; ++i) b[i] = src[threadIdx.x + 64 * (4 + i)];
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) *sink = s;
}
// the same wave tile with 32x32x16: 2 A fragments x 4 B fragments per 16-wide k step, 8 accumulators of 16 registers; the
// two k steps of a 32-wide chunk use different operand registers (as a real kernel's would)
__global__ __launch_bounds__(512) void k32(int iters, float* sink, const bf16x8* src) {
    bf16x8 a[2][2], b[2][4];
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 2; ++i) a[k][i] = src[threadIdx.x + 64 * (k * 2 + i)];
        for (int i = 0; i < 4; ++i) b[k][i] = src[threadIdx.x + 64 * (4 + k * 4 + i)];
    }
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][i], b[k][j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) *sink = s;
}

// exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, the train step's arithmetic): 8 accumulator tiles (wave tile 32 x 64 of the fp32
// implicit GEMM), four MFMAs per tile and 16-byte chunk. CHAINED: the four back to back on one accumulator (program order of
// Frag<float>::mma); otherwise tile-interleaved (dependent distance 8 MFMAs).
template <bool CHAINED>
__global__ __launch_bounds__(512) void kf32(int iters, float* sink, const bf16x8* src) {
    f32x4 a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&src[threadIdx.x + 64 * i]);
    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f32x4*>(&src[threadIdx.x + 64 * (2 + i)]);
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (CHAINED) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) *sink = s;
}

template <typename K>
static void run_f32(const char* name, K kern, int waves, int iters, float* sink, const bf16x8* src) {
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    float best = 1e9f, ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, iters, sink, src);
        hipEventRecord(e);
        hipEventSynchronize(e);
        hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double fl = 2.0 * 32 * 64 * 16 * (double)iters * waves * 256;   // one iteration = 32 x 64 x 16
    printf("%-34s waves/CU %d iters %7d  best %8.3f ms %7.1f TFLOP/s\n", name, waves, iters, best, fl / best / 1e9);
}

template <typename K>
static void run(const char* name, K kern, int waves, int iters, float* sink, const bf16x8* src) {
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    const int grid = 256;   // one workgroup per CU
    float best = 1e9f, ms = 0.f, last = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 0, 0, iters, sink, src);
        hipEventRecord(e);
        hipEventSynchronize(e);
        hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
        last = ms;
    }
    // one iteration = one 64 x 128 x 32 wave tile = 2 * 64 * 128 * 32 flop
    const double fl = 2.0 * 64 * 128 * 32 * (double)iters * waves * grid;
    printf("%-28s waves/CU %d iters %7d  best %8.3f ms %7.1f TFLOP/s   last (6th back-to-back) %8.3f ms %7.1f TFLOP/s\n", name, waves, iters, best,
           fl / best / 1e9, last, fl / last / 1e9);
}

int main() {
    float* sink;
    bf16x8* src;
    hipMalloc(&sink, 4);
    hipMalloc(&src, 64 * 12 * 8 * 8 * 2);
    for (int random : {0, 1}) {
        // constant operands toggle few wires; random ones (bf16 in +-[0.5, 2)) are what a real layer feeds the pipe
        unsigned short host[64 * 12 * 8 * 8];
        unsigned lcg = 12345u;
        for (auto& v : host) {
            lcg = lcg * 1664525u + 1013904223u;
            v = random ? (unsigned short)(((lcg >> 16) & 0x80ffu) | 0x3f00u) : (unsigned short)0x3c00u;
        }
        hipMemcpy(src, host, sizeof(host), hipMemcpyHostToDevice);
        printf("operands: %s\n", random ? "random" : "constant");
        for (int iters : {2000, 20000, 100000}) {
            run("16x16x32 (32 acc x 4 regs)", k16, 4, iters, sink, src);
            run("16x16x32 (32 acc x 4 regs)", k16, 8, iters, sink, src);
            run("16x16x32 f16 (32 acc x 4 regs)", k16h, 8, iters, sink, src);
            run("32x32x16 (8 acc x 16 regs)", k32, 4, iters, sink, src);
            run("32x32x16 (8 acc x 16 regs)", k32, 8, iters, sink, src);
        }
        for (int waves : {4, 8}) {
            run_f32("fp32 16x16x4, chained per tile", kf32<true>, waves, 100000, sink, src);
            run_f32("fp32 16x16x4, tile-interleaved", kf32<false>, waves, 100000, sink, src);
        }
    }
    return 0;
}
