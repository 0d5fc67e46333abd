// What does a bare MFMA stream reach on this chip with random operands? 256 workgroups x 4 waves (one per SIMD), each wave
// issues ITER x 64 MFMAs on 16 independent accumulators with operands held in registers; no memory traffic inside the loop.
//   mode 0: v_mfma_f32_16x16x32_f16, accumulators in VGPRs (compiler-allocated)
//   mode 1: the same, accumulators in asm-owned AGPRs (the form of conv3x3_fat.hip / bottleneck_seam.hip)
//   mode 2: v_mfma_f32_32x32x16_f16, 16 accumulators of 16 AGPRs (asm-owned), half the instructions for the same FLOPs
//   mode 3: mode 1 with 8 waves per workgroup (two per SIMD, 256 registers each) and the same work per SIMD
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_stream.hip -o gpurun_out/mfma_stream ; run: ./mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int AQ> __device__ __forceinline__ void mf16(const u32x4_t& a, const u32x4_t& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * AQ), "n"(4 * AQ + 3));
}
template <int AQ> __device__ __forceinline__ void mf32(const u32x4_t& a, const u32x4_t& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(16 * AQ), "n"(16 * AQ + 15));
}
template <int R> __device__ __forceinline__ float aread() { float x; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(R)); return x; }
template <int R> __device__ __forceinline__ void azero() { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"n"(R)); }

template <int MODE>
__global__ __launch_bounds__(MODE == 3 ? 512 : 256) void k(const u32x4_t* __restrict__ src, float* __restrict__ dst, int iters) {
    const int lane = threadIdx.x & 255;
    u32x4_t a[4], b[16];
    for (int i = 0; i < 4; ++i) a[i] = src[(blockIdx.x * 20 + i) * 256 + lane];
    for (int i = 0; i < 16; ++i) b[i] = src[(blockIdx.x * 20 + 4 + i) * 256 + lane];
    float s = 0.f;
    if constexpr (MODE == 0) {
        f4_t acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f4_t{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a[q]), __builtin_bit_cast(h8_t, b[i]), acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (MODE == 1 || MODE == 3) {
        asm volatile("" ::: "a63");
        sfor<64>([&](auto r) { azero<decltype(r)::value>(); });
        for (int it = 0; it < iters; ++it) {
            sfor<4>([&](auto q) { sfor<16>([&](auto i) { mf16<decltype(i)::value>(a[decltype(q)::value], b[decltype(i)::value]); }); });
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        sfor<64>([&](auto r) { s += aread<decltype(r)::value>(); });
    } else {
        asm volatile("" ::: "a255");
        sfor<256>([&](auto r) { azero<decltype(r)::value>(); });
        for (int it = 0; it < iters; ++it) {  // 32 MFMAs of twice the FLOPs: (2 A x 8 B) x 2 k-halves
            sfor<2>([&](auto h) {
                sfor<2>([&](auto q) {
                    sfor<8>([&](auto i) { mf32<decltype(q)::value * 8 + decltype(i)::value>(a[decltype(h)::value * 2 + decltype(q)::value], b[decltype(h)::value * 8 + decltype(i)::value]); });
                });
            });
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        sfor<256>([&](auto r) { s += aread<decltype(r)::value>(); });
    }
    dst[blockIdx.x * 256 + lane] = s;  // (mode 3: the two waves of a SIMD write the same value)
}

int main() {
    const int nblk = 256, iters = 2000;
    std::vector<unsigned> h(nblk * 20 * 256 * 4);
    srand(1);
    for (auto& v : h) {  // two random halves in [-1, 1): realistic toggling, no inf / nan
        auto half = []() { unsigned m = rand() & 0x3ff, e = 10 + rand() % 5, s = rand() & 1; return (s << 15) | (e << 10) | m; };
        v = half() | (half() << 16);
    }
    u32x4_t* src; float* dst;
    hipMalloc(&src, h.size() * 4); hipMalloc(&dst, nblk * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        if (zero) hipMemset(src, 0, h.size() * 4);
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, src, dst, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, src, dst, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(256), 0, 0, src, dst, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nblk), dim3(512), 0, 0, src, dst, iters / 2);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
            }
            const double flops = (double)nblk * 4 * iters * 64 * 16384.0;
            printf("%s operands, mode %d: %.3f ms, %.0f TFLOP/s, %.2f ns per 16x16x32-equivalent MFMA per SIMD (%.1f cycles at 2.4 GHz)\n",
                   zero ? "zero  " : "random", mode, best, flops / best * 1e-9, best * 1e6 / (iters * 64.0), best * 1e6 / (iters * 64.0) * 2.4);
        }
    }
    return 0;
}
