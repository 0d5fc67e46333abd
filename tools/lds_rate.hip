// LDS read bandwidth of one CU on gfx950: 8 or 16 waves per CU, every lane ds_read_b128 (conflict-free: lane L reads 16 bytes
// at 16 L inside a 1-KiB piece, pieces walked with a stride), results folded into a register. Reports B/clk/CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_rate.hip -o tools/lds_rate ; run: tools/lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int UNROLL>
__global__ __launch_bounds__(512) void lds_read_kernel(int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    uint4 acc = make_uint4(0, 0, 0, 0);
    unsigned off = wave * 1024 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {   // asm: the loads are neither hoisted nor merged
            const unsigned a = (off + u * 8192) & 65535;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
        off = (off + 1024) & 65535;
    }
    if (acc.x == 0x12345678u && acc.y == 17) *sink = 1.f;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* sink;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs = 1; wgs <= 2; ++wgs) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(lds_read_kernel<8>, dim3(cus * wgs), dim3(512), 64 * 1024, 0, iters, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double bytes = (double)cus * wgs * 512 * 16.0 * 8 * iters;
        printf("%d workgroup(s) of 8 waves per CU, ds_read_b128: %.3f ms, %.1f TB/s chip, %.1f GB/s per CU = %.1f B/clk at 2.1 GHz (%.1f at 2.4)\n",
               wgs, best, bytes / best * 1e-9, bytes / best * 1e-6 / cus, bytes / best * 1e-6 / cus / 2.1, bytes / best * 1e-6 / cus / 2.4);
    }
    return 0;
}
