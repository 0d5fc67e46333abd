// HBM -> L2 prefetch on DEDICATED CUs, running beside a GEMM launch on another stream. ROUND-5 EXPERIMENT, NOT PART OF THE LIBRARY: it was
// built into libagrl_hip.so as agrl_l2_prefetch (csrc/, Makefile SRCS, include/agrl_hip.h, _hip.SIGNATURES) for tools/ubench/l2_prefetch_bench.py,
// measured SLOWER (profiles/r05_l2_prefetch_experiment.txt: conv3 + residual 119 -> 132-144 us, pooled 115 -> 135-144, conv3 + downsample 185 -> 193-205)
// and taken out again: an unpaced prefetcher needs 60-87 us for what the GEMM's first tiles want after 15, both then fetch the same lines, and the GEMM
// has lost 32-64 CUs. Kept as the record of the experiment.
// tools/ubench/mix_cus.hip: the adding-up of L2 hits and HBM streams is a property of a CU's own vector-memory path --
// with the HBM streams on OTHER CUs the hit traffic runs at its own speed. So: a few workgroups that each fill a CU (1024 threads, 150 KB
// of LDS: nothing else fits beside them) read the HBM-resident operand of the GEMM running on the remaining CUs -- the residual map of
// conv3 + shortcut -- into the L2 of the XCD that will consume it, so that the GEMM's own loads of it are L2 hits.
//   * XCD x (workgroups b with b mod 8 == x: the round-robin placement this library relies on for speed elsewhere, never for
//     correctness) takes the x-th eighth of each range -- conv1x1_duo_kernel gives XCD x the x-th eighth of the pixel rows;
//   * its workgroups walk that share in 16 KB chunks, interleaved, eight chunks in flight per workgroup;
//   * the data is discarded: nothing depends on this kernel having run (a pure hint).
#include "agrl_common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void l2_prefetch_kernel(const u32x4_t* __restrict__ a, size_t a16, const u32x4_t* __restrict__ b, size_t b16,
                                                           unsigned* __restrict__ sink) {
    extern __shared__ unsigned char dyn[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    u32x4_t acc = {0, 0, 0, 0};
    auto walk = [&](const u32x4_t* p, size_t n16) {
        if (!p || !n16) return;
        const size_t share = (n16 / 8) & ~(size_t)1023;          // 16-byte units of this XCD's share (whole chunks)
        const u32x4_t* base = p + (size_t)xcd * share;
        const size_t nchunk = share >> 10;                          // 1024 lanes x 16 B
        size_t c = slot;
        for (; c + 7 * (size_t)per < nchunk; c += 8 * (size_t)per) {
            u32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[((c + (size_t)u * per) << 10) + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
        for (; c < nchunk; c += per) acc ^= base[(c << 10) + threadIdx.x];
    };
    walk(a, a16);
    walk(b, b16);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && dyn[threadIdx.x] == 77) sink[0] = acc.x;  // keeps the loads alive
}

}  // namespace

extern "C" int agrl_l2_prefetch(const void* a, size_t a_bytes, const void* b, size_t b_bytes, void* sink, int cus_per_xcd, agrl_stream_t stream) {
    AGRL_CHECK_ARG(a && sink, "agrl_l2_prefetch: null pointer");
    AGRL_CHECK_ARG(cus_per_xcd >= 1 && cus_per_xcd <= 16, "agrl_l2_prefetch: 1 .. 16 workgroups per XCD (got %d)", cus_per_xcd);
    AGRL_CHECK_ARG(((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0, "agrl_l2_prefetch: pointers must be 16-byte aligned");
    static bool attr = false;
    constexpr int kLds = 150 * 1024;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)l2_prefetch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 1;
        attr = true;
    }
    hipLaunchKernelGGL(l2_prefetch_kernel, dim3(8 * cus_per_xcd), dim3(1024), kLds, (hipStream_t)stream, reinterpret_cast<const u32x4_t*>(a),
                       a_bytes / 16, reinterpret_cast<const u32x4_t*>(b), b_bytes / 16, reinterpret_cast<unsigned*>(sink));
    AGRL_CHECK_LAUNCH("agrl_l2_prefetch");
    return 0;
}
