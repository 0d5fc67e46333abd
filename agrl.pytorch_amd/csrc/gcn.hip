// Pose-guided adaptive graph convolution (GraphLayer, torchreid/models/vmgn.py:68-172) for gfx950.
//
//   agrl_graph_gram      partial Gram matrices F F^T over channel slices    (vmgn.py:116-118)   MFMA fp32, exact
//   agrl_graph_finalize  d -> sim -> row-L1 normalise -> mix with pose adj  (vmgn.py:118-120, :155-166)
//   agrl_graph_propagate G h -> BN1d(eval) -> LeakyReLU -> (1-g) f + g h'   (vmgn.py:168-172)   HBM-bound
//
// The Linear h = f W^T between them is agrl_linear_nobias (igemm.hip).
// Everything is deterministic: no atomics, fixed summation order.
#include "agrl_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// Gram partials. grid = (B, nz). One workgroup stages f[b, :, z*CS : (z+1)*CS] (V x CS fp32) in LDS and
// computes the V x V partial Gram with v_mfma_f32_16x16x4_f32 (bitwise an fp32 fma chain).
// LDS rows are CS*4 + 16 bytes so the 16 lanes of a fragment read hit distinct 16-byte bank slots.
constexpr int GRAM_CS = 128;
constexpr int GRAM_ROWB = GRAM_CS * 4 + 16;

__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ f, float* __restrict__ gram_part, int V,
                                                   int C, int nz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_f[];
    const int b = blockIdx.x, z = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Vp = (V + 15) & ~15;
    const int nf = Vp >> 4;
    // stage: thread -> float4 (tid&31) of row (tid>>5) + 8*i
    const float* src = f + (size_t)b * V * C + (size_t)z * GRAM_CS;
    for (int r = tid >> 5; r < Vp; r += 8) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < V) v = *reinterpret_cast<const float4*>(src + (size_t)r * C + (tid & 31) * 4);
        *reinterpret_cast<float4*>(s_f + r * GRAM_ROWB + (tid & 31) * 16) = v;
    }
    __syncthreads();
    const int frow = lane & 15, fch = lane >> 4;
    float* dst = gram_part + ((size_t)b * nz + z) * V * V;
    for (int fr = wave; fr < nf * nf; fr += 4) {
        const int fi = fr / nf, fj = fr - fi * nf;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* pa = s_f + (fi * 16 + frow) * GRAM_ROWB + fch * 16;
        const unsigned char* pb = s_f + (fj * 16 + frow) * GRAM_ROWB + fch * 16;
#pragma unroll
        for (int ks = 0; ks < GRAM_CS / 16; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(pa + ks * 64);
            const float4 bb = *reinterpret_cast<const float4*>(pb + ks * 64);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, acc, 0, 0, 0);
        }
        // D[row = 4*(lane>>4) + r][col = lane&15]
        const int j = fj * 16 + frow;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = fi * 16 + fch * 4 + r;
            if (i < V && j < V) dst[(size_t)i * V + j] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Finalize. grid = B, 256 threads; one wavefront per graph row.
__global__ __launch_bounds__(256) void graph_finalize_kernel(const float* __restrict__ gram_part, int nz,
                                                             const float* __restrict__ adj, float* __restrict__ G,
                                                             int V, int use_pose, int learn_graph) {
    extern __shared__ __attribute__((aligned(16))) float s_g[];  // V*V gram, then V norms
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* s_n = s_g + V * V;
    if (learn_graph) {
        const float* gp = gram_part + (size_t)b * nz * V * V;
        for (int e = tid; e < V * V; e += 256) {
            float s = 0.f;
            for (int z = 0; z < nz; ++z) s += gp[(size_t)z * V * V + e];
            s_g[e] = s;
        }
        __syncthreads();
        for (int i = tid; i < V; i += 256) s_n[i] = s_g[i * V + i];
        __syncthreads();
    }
    for (int i = wave; i < V; i += 4) {
        // similarity row: 2 / (exp(sqrt(clamp(n_j + n_i - 2 g_ij, 1e-12))) + 1), then /max(sum,1e-12)
        float ssum = 0.f, asum = 0.f;
        for (int j = lane; j < V; j += 64) {
            if (learn_graph) {
                float d2 = (s_n[j] + s_n[i]) - 2.f * s_g[i * V + j];
                d2 = fmaxf(d2, 1e-12f);
                const float sim = 2.f / (expf(sqrtf(d2)) + 1.f);
                s_g[i * V + j] = sim;
                ssum += fabsf(sim);
            }
            if (use_pose) asum += fabsf(adj[((size_t)b * V + i) * V + j]);
        }
        ssum = wave_sum(ssum);
        asum = wave_sum(asum);
        const float sden = fmaxf(ssum, 1e-12f), aden = fmaxf(asum, 1e-12f);
        for (int j = lane; j < V; j += 64) {
            float g;
            if (learn_graph) {
                g = s_g[i * V + j] / sden;
                if (use_pose) g = (adj[((size_t)b * V + i) * V + j] / aden + g) / 2.f;
            } else {
                g = adj[((size_t)b * V + i) * V + j] / aden;
            }
            G[((size_t)b * V + i) * V + j] = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Propagate. grid = (B, C/256); thread -> one channel; the thread's h column sits in LDS (private to
// the thread: no barrier needed), graph rows come through the scalar cache (wave-uniform addresses),
// RB output rows are register-blocked so each LDS read feeds RB FMAs.
template <int RB>
__global__ __launch_bounds__(256) void graph_propagate_kernel(const float* __restrict__ f, const float* __restrict__ h,
                                                              const float* __restrict__ G,
                                                              const float* __restrict__ bn_scale,
                                                              const float* __restrict__ bn_shift, float one_minus_gamma,
                                                              float gamma, float slope, float* __restrict__ out,
                                                              bf16_t* __restrict__ out_lp, int V, int C) {
    extern __shared__ __attribute__((aligned(16))) float s_h[];  // [V][256]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int c = blockIdx.y * 256 + tid;
    const bool live = c < C;
    const int cc = live ? c : C - 1;
    const float* hb = h + (size_t)b * V * C + cc;
    for (int u = 0; u < V; ++u) s_h[u * 256 + tid] = hb[(size_t)u * C];
    const float sc = bn_scale[cc], sh = bn_shift[cc];
    const float* Gb = G + (size_t)b * V * V;
    const float* fb = f + (size_t)b * V * C + cc;
    for (int v0 = 0; v0 < V; v0 += RB) {
        float acc[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) acc[k] = 0.f;
        const float* grow[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) grow[k] = Gb + (size_t)(v0 + k < V ? v0 + k : V - 1) * V;
        for (int u = 0; u < V; ++u) {
            const float hv = s_h[u * 256 + tid];
#pragma unroll
            for (int k = 0; k < RB; ++k) acc[k] = fmaf(grow[k][u], hv, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int v = v0 + k;
            if (v < V && live) {
                float y = fmaf(acc[k], sc, sh);
                y = y > 0.f ? y : slope * y;
                const float o = one_minus_gamma * fb[(size_t)v * C] + gamma * y;
                const size_t idx = ((size_t)b * V + v) * C + c;
                out[idx] = o;
                if (out_lp) out_lp[idx] = f32_to_bf16(o);
            }
        }
    }
}

}  // namespace

extern "C" int agrl_graph_gram(const float* f, float* gram_part, int B, int V, int C, int cslice,
                               agrl_stream_t stream) {
    AGRL_CHECK_ARG(f && gram_part, "agrl_graph_gram: null pointer");
    AGRL_CHECK_ARG(B > 0 && V > 0 && C > 0, "agrl_graph_gram: bad shape");
    AGRL_CHECK_ARG(cslice == GRAM_CS && C % GRAM_CS == 0, "agrl_graph_gram: cslice must be %d and divide C", GRAM_CS);
    const int Vp = (V + 15) & ~15;
    const size_t lds = (size_t)Vp * GRAM_ROWB;
    AGRL_CHECK_ARG(lds <= 160 * 1024, "agrl_graph_gram: V=%d too large", V);
    const int nz = C / GRAM_CS;
    hipLaunchKernelGGL(gram_kernel, dim3(B, nz), dim3(256), lds, (hipStream_t)stream, f, gram_part, V, C, nz);
    AGRL_CHECK_LAUNCH("agrl_graph_gram");
    return 0;
}

extern "C" int agrl_graph_finalize(const float* gram_part, int nz, const float* adj, float* G, int B, int V,
                                   int use_pose, int learn_graph, agrl_stream_t stream) {
    AGRL_CHECK_ARG(G && B > 0 && V > 0, "agrl_graph_finalize: bad arguments");
    AGRL_CHECK_ARG(use_pose || learn_graph, "agrl_graph_finalize: use_pose or learn_graph must be set");
    AGRL_CHECK_ARG(!use_pose || adj, "agrl_graph_finalize: use_pose needs adj");
    AGRL_CHECK_ARG(!learn_graph || (gram_part && nz > 0), "agrl_graph_finalize: learn_graph needs the Gram partials");
    const size_t lds = ((size_t)V * V + V) * sizeof(float);
    AGRL_CHECK_ARG(lds <= 160 * 1024, "agrl_graph_finalize: V=%d too large", V);
    hipLaunchKernelGGL(graph_finalize_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, gram_part, nz, adj, G, V,
                       use_pose, learn_graph);
    AGRL_CHECK_LAUNCH("agrl_graph_finalize");
    return 0;
}

extern "C" int agrl_graph_propagate(const float* f, const float* h, const float* G, const float* bn_scale,
                                    const float* bn_shift, float gamma, float slope, float* out, void* out_lp, int B,
                                    int V, int C, agrl_stream_t stream) {
    AGRL_CHECK_ARG(f && h && G && bn_scale && bn_shift && out, "agrl_graph_propagate: null pointer");
    AGRL_CHECK_ARG(B > 0 && V > 0 && C > 0, "agrl_graph_propagate: bad shape");
    const size_t lds = (size_t)V * 256 * sizeof(float);
    AGRL_CHECK_ARG(lds <= 160 * 1024, "agrl_graph_propagate: V=%d too large", V);
    // (1 - gamma) is evaluated in double like the reference's Python float, then rounded once
    const float omg = (float)(1.0 - (double)gamma);
    if (lds > 64 * 1024) {  // per-device attribute, idempotent: set it whenever the launch needs it
        hipError_t e = hipFuncSetAttribute((const void*)graph_propagate_kernel<8>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        AGRL_CHECK_ARG(e == hipSuccess, "agrl_graph_propagate: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(graph_propagate_kernel<8>, dim3(B, cdiv(C, 256)), dim3(256), lds, (hipStream_t)stream, f, h, G,
                       bn_scale, bn_shift, omg, gamma, slope, out, (bf16_t*)out_lp, V, C);
    AGRL_CHECK_LAUNCH("agrl_graph_propagate");
    return 0;
}
