// Achievable HBM / MALL READ bandwidth on this chip: the yardstick for the read-dominated kernels (distmat, GCN).
// build: hipcc --offload-arch=gfx950 -O3 tools/readbw.hip -o tools/readbw ; run: tools/readbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, size_t n4, float* sink) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { float4 v = src[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}
int main() {
    const size_t maxb = 2048ull << 20;
    float4 *a, *b; float* sink;
    hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&sink, 4);
    hipMemset(a, 1, maxb); hipMemset(b, 0, maxb);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (size_t mb : {16, 50, 100, 400, 1600}) {
        const size_t n4 = (mb << 20) / 16;
        for (int grid : {512, 1024, 2048, 4096}) {
            float best = 1e9, ms;
            for (int rep = 0; rep < 8; ++rep) {
                // rotate through the 2 GB buffer so that a "cold" pass cannot hit in the 256 MB MALL
                const size_t off = (mb >= 400) ? 0 : ((size_t)rep * (256ull << 20) / 16) % ((maxb - (mb << 20)) / 16);
                hipEventRecord(s);
                hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, a + off, n4, sink);
                hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms, s, e);
                if (rep >= 2 && ms < best) best = ms;
            }
            float bestw = 1e9;
            for (int rep = 0; rep < 8; ++rep) {  // same region every time: MALL / L2 warm
                hipEventRecord(s);
                hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, a, n4, sink);
                hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms, s, e);
                if (rep >= 2 && ms < bestw) bestw = ms;
            }
            float bestc = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(s);
                hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n4);
                hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms, s, e);
                if (rep >= 2 && ms < bestc) bestc = ms;
            }
            printf("%5zu MB grid %4d: read rotating %7.1f us %5.2f TB/s | read same %7.1f us %5.2f TB/s | copy %7.1f us %5.2f TB/s (r+w)\n",
                   mb, grid, best * 1e3, (mb << 20) / (best * 1e-3) / 1e12, bestw * 1e3, (mb << 20) / (bestw * 1e-3) / 1e12,
                   bestc * 1e3, 2.0 * (mb << 20) / (bestc * 1e-3) / 1e12);
        }
    }
    return 0;
}
