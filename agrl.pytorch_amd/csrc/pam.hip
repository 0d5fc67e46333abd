// Position-attention part nodes of the sibling model ganet (torchreid/models/ganet.py:98-136 PAM_Module, :384-400 the
// per-slice use): for every frame and every pyramid slice (rows [start, end) of the h x w map, L = rows * w positions)
//     attention = softmax_q(query_p . key_q)            (L x L, over the key axis)
//     node      = avgpool(gamma * value . attention^T + 2 slice)
// The value conv (C -> C on every position, three times per frame because the pyramid covers the map three times) never
// has to be evaluated per position: average pooling is linear and every attention row sums to one, so
//     avgpool(value . attention^T) = Wv (X abar) + bv,      abar_q = mean_p attention[p][q]
// i.e. ONE matrix-vector product per node on the attention-weighted mean of the slice. This kernel produces the two
// per-node vectors the host needs -- xbar = X abar and xmean = mean of the slice -- from the map and the stacked
// query / key conv output; the host then runs the (F*P, C) x (C, C) Linear on xbar (agrl_linear_nobias) and combines
// (agrl_pam_combine). With the module's gamma == 0 (its value at construction) only xmean is needed.
// grid = (frames, parts), 256 threads. HBM-bound on the map (read once per pyramid level).
#include "agrl_common.h"

namespace {

constexpr int PAM_MAXL = 128;           // positions per slice (16 x 8 map)
constexpr int PAM_ES = PAM_MAXL + 1;    // energy row stride (floats): column sweeps are conflict-free
constexpr int PAM_CH = 32;              // query / key channels staged per step
constexpr int PAM_TS = PAM_CH + 1;

struct PamBins {
    int nparts;
    int start[16], end[16];
};

template <typename T>
__device__ inline float ldf(const T* p);
template <>
__device__ inline float ldf<float>(const float* p) { return *p; }
template <>
__device__ inline float ldf<lp16_t>(const lp16_t* p) { return lp16_to_f32(*p); }

template <typename T>
__global__ __launch_bounds__(256) void pam_pool_kernel(const T* __restrict__ x, const T* __restrict__ qk, float* __restrict__ xbar,
                                                       float* __restrict__ xmean, int h, int w, int C, int Cq, PamBins bins,
                                                       int with_attention) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_e = s_mem;                               // [L][PAM_ES] energy -> attention
    float* s_q = s_e + PAM_MAXL * PAM_ES;             // [L][PAM_TS]
    float* s_k = s_q + PAM_MAXL * PAM_TS;             // [L][PAM_TS]
    float* s_abar = s_k + PAM_MAXL * PAM_TS;          // [L]
    const int frame = blockIdx.x, part = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = bins.start[part] * w;
    const int L = (bins.end[part] - bins.start[part]) * w;
    const size_t pix0 = (size_t)frame * h * w + p0;
    if (with_attention) {
        const int ty = tid >> 4, tx = tid & 15;
        float acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        const T* qkb = qk + pix0 * (size_t)(2 * Cq);
        for (int c0 = 0; c0 < Cq; c0 += PAM_CH) {
            for (int e = tid; e < L * PAM_CH; e += 256) {
                const int p = e / PAM_CH, c = e - p * PAM_CH;
                s_q[p * PAM_TS + c] = ldf<T>(qkb + (size_t)p * 2 * Cq + c0 + c);
                s_k[p * PAM_TS + c] = ldf<T>(qkb + (size_t)p * 2 * Cq + Cq + c0 + c);
            }
            __syncthreads();
#pragma unroll 4
            for (int c = 0; c < PAM_CH; ++c) {
                float a[8], b[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = s_q[(ty + 16 * i) * PAM_TS + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = s_k[(tx + 16 * j) * PAM_TS + c];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = ty + 16 * i, q = tx + 16 * j;
                if (p < L && q < L) s_e[p * PAM_ES + q] = acc[i][j];
            }
        __syncthreads();
        // softmax over the key axis, one wavefront per query row
        for (int p = wave; p < L; p += 4) {
            const float v0 = lane < L ? s_e[p * PAM_ES + lane] : -INFINITY;
            const float v1 = lane + 64 < L ? s_e[p * PAM_ES + lane + 64] : -INFINITY;
            const float m = wave_max(fmaxf(v0, v1));
            const float e0 = lane < L ? expf(v0 - m) : 0.f, e1 = lane + 64 < L ? expf(v1 - m) : 0.f;
            const float s = wave_sum(e0 + e1);
            if (lane < L) s_e[p * PAM_ES + lane] = e0 / s;
            if (lane + 64 < L) s_e[p * PAM_ES + lane + 64] = e1 / s;
        }
        __syncthreads();
        if (tid < L) {
            float s = 0.f;
            for (int p = 0; p < L; ++p) s += s_e[p * PAM_ES + tid];
            s_abar[tid] = s / (float)L;
        }
        __syncthreads();
    }
    // xbar[c] = sum_q abar[q] x[q][c], xmean[c] = mean_q x[q][c]; thread -> channels c, c + 256, ..: coalesced rows
    const T* xb = x + pix0 * (size_t)C;
    const size_t node = (size_t)frame * bins.nparts + part;
    for (int c = tid; c < C; c += 256) {
        float sb = 0.f, sm = 0.f;
        for (int q0 = 0; q0 < L; q0 += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = q0 + i < L ? ldf<T>(xb + (size_t)(q0 + i) * C + c) : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (q0 + i < L) {
                    sm += v[i];
                    if (with_attention) sb = fmaf(s_abar[q0 + i], v[i], sb);
                }
        }
        xmean[node * C + c] = sm / (float)L;
        if (with_attention) xbar[node * C + c] = sb;
    }
}

// nodes = gamma * (y + bv) + 2 * xmean (y = Wv xbar), + optional bf16 copy (operand of the next Linear)
__global__ __launch_bounds__(256) void pam_combine_kernel(const float* __restrict__ y, const float* __restrict__ bv,
                                                          const float* __restrict__ xmean, float gamma, float* __restrict__ nodes,
                                                          lp16_t* __restrict__ nodes_lp, size_t total, int C) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        float v = 2.f * xmean[e];
        if (y) v = fmaf(gamma, y[e] + bv[c], v);
        nodes[e] = v;
        if (nodes_lp) nodes_lp[e] = f32_to_lp16(v);
    }
}

}  // namespace

extern "C" int agrl_pam_pool(const void* x, const void* qk, float* xbar, float* xmean, int F, int h, int w, int C, int Cq,
                             const int* splits, int n_splits, int dtype, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && xmean && splits, "agrl_pam_pool: null pointer");
    AGRL_CHECK_ARG((qk == nullptr) == (xbar == nullptr), "agrl_pam_pool: qk and xbar go together (both NULL when the module's gamma is 0)");
    AGRL_CHECK_ARG(F > 0 && h > 0 && w > 0 && C > 0 && n_splits > 0, "agrl_pam_pool: bad shape");
    AGRL_CHECK_ARG(dtype == AGRL_F32 || dtype == AGRL_LP16, "agrl_pam_pool: bad dtype %d", dtype);
    AGRL_CHECK_ARG(!qk || (Cq > 0 && Cq % PAM_CH == 0), "agrl_pam_pool: Cq=%d must be a multiple of %d", Cq, PAM_CH);
    PamBins bins;
    int P = 0;
    for (int i = 0; i < n_splits; ++i) {
        const int n = splits[i];
        AGRL_CHECK_ARG(n > 0 && P + n <= 16 && h / n > 0, "agrl_pam_pool: at most 16 parts, each at least one map row");
        const int step = h / n;  // ganet.py:387-390: h // n rows per slice, remainder rows dropped
        for (int j = 0; j < n; ++j) {
            bins.start[P] = step * j;
            bins.end[P] = step * (j + 1);
            AGRL_CHECK_ARG(step * w <= PAM_MAXL, "agrl_pam_pool: a slice has %d positions, at most %d supported", step * w, PAM_MAXL);
            ++P;
        }
    }
    bins.nparts = P;
    for (int i = P; i < 16; ++i) bins.start[i] = bins.end[i] = 0;
    const size_t lds = (size_t)(PAM_MAXL * PAM_ES + 2 * PAM_MAXL * PAM_TS + PAM_MAXL) * sizeof(float);
    const int att = qk != nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AGRL_F32) {
        hipError_t e = hipFuncSetAttribute((const void*)pam_pool_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        AGRL_CHECK_ARG(e == hipSuccess, "agrl_pam_pool: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(pam_pool_kernel<float>, dim3(F, P), dim3(256), lds, st, (const float*)x, (const float*)qk, xbar, xmean,
                           h, w, C, Cq, bins, att);
    } else {
        hipError_t e = hipFuncSetAttribute((const void*)pam_pool_kernel<lp16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        AGRL_CHECK_ARG(e == hipSuccess, "agrl_pam_pool: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(pam_pool_kernel<lp16_t>, dim3(F, P), dim3(256), lds, st, (const lp16_t*)x, (const lp16_t*)qk, xbar, xmean,
                           h, w, C, Cq, bins, att);
    }
    AGRL_CHECK_LAUNCH("agrl_pam_pool");
    return 0;
}

extern "C" int agrl_pam_combine(const float* y, const float* bv, const float* xmean, float gamma, float* nodes, void* nodes_lp,
                                int rows, int C, agrl_stream_t stream) {
    AGRL_CHECK_ARG(xmean && nodes && rows > 0 && C > 0, "agrl_pam_combine: bad arguments");
    AGRL_CHECK_ARG((y == nullptr) == (bv == nullptr), "agrl_pam_combine: y and bv go together");
    const size_t total = (size_t)rows * C;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pam_combine_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bv, xmean, gamma, nodes,
                       (lp16_t*)nodes_lp, total, C);
    AGRL_CHECK_LAUNCH("agrl_pam_combine");
    return 0;
}
