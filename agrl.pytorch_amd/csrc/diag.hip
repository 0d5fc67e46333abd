// Measurement yardstick of the HBM-bound kernels ("the builder must re-measure achievable peaks
// with a stream kernel"): a pure read stream over a caller-supplied buffer, 16-byte loads, eight in flight per lane.
// bench.py times it on the launch stream at the byte counts of the distance matrix and of the GCN message pass and prints
// the rate beside those kernels' own (the 8 TB/s HBM3E figure is a datasheet number no single-pass read of 15-100 MB
// reaches on this chip). Not part of the hot path.
#include "agrl_common.h"

namespace {

__global__ __launch_bounds__(256) void read_stream_kernel(const float4* __restrict__ src, size_t n4, float* sink) {
    constexpr int U = 8;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
    }
    for (; i < n4; i += stride) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;  // keeps the loads alive, never true in practice
}

}  // namespace

extern "C" int agrl_diag_read_stream(const void* src, size_t bytes, float* sink, int workgroups, agrl_stream_t stream) {
    AGRL_CHECK_ARG(src && sink, "agrl_diag_read_stream: null pointer");
    AGRL_CHECK_ARG(bytes >= 16 && (((uintptr_t)src) & 15) == 0, "agrl_diag_read_stream: need >= 16 bytes, 16-byte aligned");
    AGRL_CHECK_ARG(workgroups > 0 && workgroups <= 65536, "agrl_diag_read_stream: bad grid");
    hipLaunchKernelGGL(read_stream_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), bytes / 16, sink);
    AGRL_CHECK_LAUNCH("agrl_diag_read_stream");
    return 0;
}
