// Do three kinds of memory traffic issued by DIFFERENT waves of one CU overlap or add up? (The question behind conv1x1_duo.hip: a
// conv3 + residual tile moves L2-resident weights / pixel rows, an HBM residual stream and an HBM result stream through each CU.)
// 256 workgroups x 8 waves, one workgroup per CU (96 KB of dynamic LDS), roles by wave:
//   waves 0-3  "hit":   re-read a 2 MB region (the weights of a 512 -> 2048 conv) A bytes per workgroup, 16 B per lane
//   waves 4-5  "read":  stream B bytes per workgroup of a region nobody else touches (the residual)
//   waves 6-7  "write": stream C bytes per workgroup of stores (the result)
// Reported: microseconds for each kind alone and for the combinations, against the sum and the maximum of the parts.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mix_stream.hip -o gpurun_out/mix_stream ; run: ./mix_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mix(const u32x4_t* __restrict__ hit, const u32x4_t* __restrict__ rd, u32x4_t* __restrict__ wr,
                                          unsigned* __restrict__ sink, int nhit, int nrd, int nwr, int rot) {
    // n*: 1 KiB wave-instructions per wave of the role
    extern __shared__ unsigned char dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4_t acc = {0, 0, 0, 0};
    if (wave < 4) {
        // 2 MB = 2048 wave-instructions; every workgroup walks the region from a different start
        unsigned pos = (blockIdx.x * 37 + wave * 512) & 2047;
        for (int i = 0; i < nhit; i += 8) {
            u32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = hit[(size_t)((pos + u) & 2047) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
            pos += 8;
        }
    } else if (wave < 6) {
        const size_t base = ((size_t)(blockIdx.x + rot * 256) * 2 + (wave - 4)) * (size_t)nrd * 64;
        for (int i = 0; i < nrd; i += 8) {
            u32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = rd[base + (size_t)(i + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    } else {
        const size_t base = ((size_t)(blockIdx.x + rot * 256) * 2 + (wave - 6)) * (size_t)nwr * 64;
        const u32x4_t v = {(unsigned)lane, (unsigned)wave, (unsigned)blockIdx.x, 7u};
        for (int i = 0; i < nwr; ++i) wr[base + (size_t)i * 64 + lane] = v;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x + dyn[0];
}

int main() {
    const size_t big = 1536ull << 20;
    u32x4_t *hit, *rd, *wr; unsigned* sink;
    hipMalloc(&hit, 2 << 20); hipMalloc(&rd, big); hipMalloc(&wr, big); hipMalloc(&sink, 4);
    hipMemset(hit, 1, 2 << 20); hipMemset(rd, 2, big); hipMemset(wr, 0, big);
    hipFuncSetAttribute((const void*)mix, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    // per workgroup: A = 3 MB of hits (what a CU moves from L2 for one conv3 launch), B = C = 512 KB (residual / result per CU)
    struct Case { const char* name; int a, b, c; };
    const int A = 3072 / 4, B = 512 / 2, C = 512 / 2;   // KiB-instructions per wave of the role
    const Case cases[] = {{"hit only", A, 0, 0}, {"read only", 0, B, 0}, {"write only", 0, 0, C}, {"hit + read", A, B, 0}, {"hit + write", A, 0, C},
                          {"read + write", 0, B, C}, {"hit + read + write", A, B, C}, {"hit/2 + read + write", A / 2, B, C}};
    for (const Case& c : cases) {
        float best = 1e9, ms;
        for (int rep = 0; rep < 6; ++rep) {   // rotate through 1.5 GB: the read / write regions never sit in the 256 MB memory-side cache
            hipEventRecord(s);
            hipLaunchKernelGGL(mix, dim3(256), dim3(512), 96 * 1024, 0, hit, rd, wr, sink, c.a, c.b, c.c, rep % 5);
            hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms, s, e);
            if (rep >= 1 && ms < best) best = ms;
        }
        const double mb = 256.0 * (4.0 * c.a + 2.0 * c.b + 2.0 * c.c) / 1024.0;
        printf("%-22s %7.1f us   (%.0f MB through the CUs: hits %.0f MB, HBM reads %.0f MB, HBM writes %.0f MB)\n", c.name, best * 1e3, mb,
               256.0 * 4 * c.a / 1024, 256.0 * 2 * c.b / 1024, 256.0 * 2 * c.c / 1024);
    }
    return 0;
}
