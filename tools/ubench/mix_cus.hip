// Follow-up to mix_stream.hip: is the adding-up of L2 hits and HBM streams a property of the CU's own vector-memory path (requests of
// all waves of a CU queue in one in-order miss path: an L2 hit waits behind an HBM miss) or of the chip (L2 / fabric)? Same traffic as
// mix_stream -- 768 MB of L2 hits, 128 MB of HBM reads, 128 MB of HBM writes -- but the kinds on DIFFERENT CUs: workgroup b (one per
// CU, 256 of them) is a "hit" workgroup unless b mod 8 == 7 ("stream": 32 CUs, four per XCD, whose eight waves read / write the HBM regions).
// Reported: the hit CUs alone, the stream CUs alone, both together.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mix_cus.hip -o gpurun_out/mix_cus ; run: ./mix_cus
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mix(const u32x4_t* __restrict__ hit, const u32x4_t* __restrict__ rd, u32x4_t* __restrict__ wr,
                                          unsigned* __restrict__ sink, int nhit, int nrd, int nwr, int rot, int nstream) {
    extern __shared__ unsigned char dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const bool stream = slot >= 32 - nstream;   // the last `nstream` workgroups of every XCD
    u32x4_t acc = {0, 0, 0, 0};
    if (!stream) {
        if (wave < 4) {
            unsigned pos = (blockIdx.x * 37 + wave * 512) & 2047;
            for (int i = 0; i < nhit; i += 8) {
                u32x4_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = hit[(size_t)((pos + u) & 2047) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u];
                pos += 8;
            }
        }
    } else {
        const int sid = xcd * nstream + (slot - (32 - nstream));   // 0 .. 8 nstream - 1
        if (wave < 4) {
            const size_t base = ((size_t)(sid + rot * 8 * nstream) * 4 + wave) * (size_t)nrd * 64;
            for (int i = 0; i < nrd; i += 8) {
                u32x4_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = rd[base + (size_t)(i + u) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u];
            }
        } else {
            const size_t base = ((size_t)(sid + rot * 8 * nstream) * 4 + (wave - 4)) * (size_t)nwr * 64;
            const u32x4_t v = {(unsigned)lane, (unsigned)wave, (unsigned)blockIdx.x, 7u};
            for (int i = 0; i < nwr; ++i) wr[base + (size_t)i * 64 + lane] = v;
        }
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
    for (int nstream : {4, 8}) {   // stream CUs per XCD (of 32)
        const int ncu_hit = 256 - 8 * nstream, ncu_str = 8 * nstream;
        // totals as in mix_stream: 768 MB of hits over the hit CUs, 128 MB read + 128 MB written over the stream CUs
        const int A = (int)(768.0 * 1024 / ncu_hit / 4) & ~7;          // KiB-instructions per hit wave
        const int B = (int)(128.0 * 1024 / ncu_str / 4) & ~7, C = B;   // per stream wave
        struct Case { const char* name; int a, b, c; };
        const Case cases[] = {{"hit CUs only", A, 0, 0}, {"stream CUs: read", 0, B, 0}, {"stream CUs: write", 0, 0, C}, {"stream CUs: read + write", 0, B, C},
                              {"hit CUs + read", A, B, 0}, {"hit CUs + write", A, 0, C}, {"hit CUs + read + write", A, B, C}};
        printf("%d stream CUs per XCD (%d hit CUs, %d stream CUs)\n", nstream, ncu_hit, ncu_str);
        for (const Case& c : cases) {
            float best = 1e9, ms;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(s);
                hipLaunchKernelGGL(mix, dim3(256), dim3(512), 96 * 1024, 0, hit, rd, wr, sink, c.a, c.b, c.c, rep % 5, nstream);
                hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms, s, e);
                if (rep >= 1 && ms < best) best = ms;
            }
            printf("  %-26s %7.1f us   (hits %.0f MB, HBM reads %.0f MB, HBM writes %.0f MB)\n", c.name, best * 1e3, ncu_hit * 4.0 * c.a / 1024,
                   ncu_str * 4.0 * c.b / 1024, ncu_str * 4.0 * c.c / 1024);
        }
    }
    return 0;
}
