// Pose-guided adaptive graph convolution (GraphLayer, torchreid/models/vmgn.py:68-172) for gfx950.
//
//   agrl_graph_gram      partial Gram matrices F F^T over channel slices    (vmgn.py:116-118)   MFMA fp32, exact
//   agrl_graph_finalize  d -> sim -> row-L1 normalise -> mix with pose adj  (vmgn.py:118-120, :155-166)
//   agrl_graph_propagate G h -> BN1d(eval) -> LeakyReLU -> (1-g) f + g h'   (vmgn.py:168-172)   HBM-bound
//
// The Linear h = f W^T between them is agrl_linear_nobias (igemm.hip).
// Everything is deterministic: no atomics, fixed summation order.
#include <stdlib.h>

#include "agrl_common.h"
#include "igemm_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// Gram partials. grid = (B, nz). One workgroup stages f[b, :, z*CS : (z+1)*CS] (V x CS fp32) in LDS and
// computes the V x V partial Gram with v_mfma_f32_16x16x4_f32 (bitwise an fp32 fma chain).
// LDS rows are CS*4 + 16 bytes so the 16 lanes of a fragment read hit distinct 16-byte bank slots.
constexpr int GRAM_CS = 128;
constexpr int GRAM_ROWB = GRAM_CS * 4 + 16;

// PAIR: partials of A B^T for two node matrices (the message pass's d loss / d G = dmsg h^T per tracklet) -- the second
// matrix is staged behind the first
template <bool PAIR>
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ f, const float* __restrict__ f2,
                                                   float* __restrict__ gram_part, int V, int C, int nz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_f[];
    const int b = blockIdx.x, z = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Vp = (V + 15) & ~15;
    const int nf = Vp >> 4;
    // stage: thread -> float4 (tid&31) of row (tid>>5) + 8*i
#pragma unroll
    for (int which = 0; which < (PAIR ? 2 : 1); ++which) {
    const float* src = (which ? f2 : f) + (size_t)b * V * C + (size_t)z * GRAM_CS;
    unsigned char* s_dst = s_f + which * Vp * GRAM_ROWB;
    // all of a thread's loads are issued before the first LDS store (a load -> store loop is a chain of HBM round trips)
    for (int r0 = tid >> 5; r0 < Vp; r0 += 64) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = r0 + 8 * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < V) v[i] = *reinterpret_cast<const float4*>(src + (size_t)r * C + (tid & 31) * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = r0 + 8 * i;
            if (r < Vp) *reinterpret_cast<float4*>(s_dst + r * GRAM_ROWB + (tid & 31) * 16) = v[i];
        }
    }
    }
    __syncthreads();
    const int frow = lane & 15, fch = lane >> 4;
    float* dst = gram_part + ((size_t)b * nz + z) * V * V;
    for (int fr = wave; fr < nf * nf; fr += 4) {
        const int fi = fr / nf, fj = fr - fi * nf;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* pa = s_f + (fi * 16 + frow) * GRAM_ROWB + fch * 16;
        const unsigned char* pb = s_f + (PAIR ? Vp * GRAM_ROWB : 0) + (fj * 16 + frow) * GRAM_ROWB + fch * 16;
        // all sixteen operand reads of the fragment pair first, then the 32 MFMAs (one read -> wait -> MFMA per k-step is
        // a chain of LDS round trips)
        float4 av[GRAM_CS / 16], bv[GRAM_CS / 16];
#pragma unroll
        for (int ks = 0; ks < GRAM_CS / 16; ++ks) {
            av[ks] = *reinterpret_cast<const float4*>(pa + ks * 64);
            bv[ks] = *reinterpret_cast<const float4*>(pb + ks * 64);
        }
#pragma unroll
        for (int ks = 0; ks < GRAM_CS / 16; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].x, bv[ks].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].y, bv[ks].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].z, bv[ks].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].w, bv[ks].w, acc, 0, 0, 0);
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

// sum of the nz Gram partials of one element, z ascending (fixed order); sixteen loads in flight at a time instead of a
// load -> add chain (nz = 16 for C = 2048: one batch)
__device__ inline float zsum(const float* p, int nz, int zstride) {
    float s = 0.f;
    for (int z0 = 0; z0 < nz; z0 += 16) {
        float part[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) part[i] = z0 + i < nz ? p[(size_t)(z0 + i) * zstride] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (z0 + i < nz) s += part[i];
    }
    return s;
}

// ---------------------------------------------------------------------------------------------------
// Finalize. grid = (B, ceil(V/4)), 256 threads: one wavefront per graph row, so B*V/4 workgroups are in
// flight and every row's loads are independent (the first version, one workgroup per tracklet walking its
// rows, was latency-bound at ~70 us).
// adjacency entry (i, j) of tracklet b from the fp32 (B,V,V) tensor of the reference's loader or from the bit-packed form
// (B, V, ceil(V/32)) uint32 words, bit j & 31 of word j >> 5 of row i (agrl_pose_adjacency_bits / agrl_adjacency_pack)
__device__ inline float adj_at(const float* __restrict__ adj, const uint32_t* __restrict__ bits, size_t row, int V, int j) {
    if (bits) return (float)((bits[row * ((V + 31) >> 5) + (j >> 5)] >> (j & 31)) & 1u);
    return adj[row * V + j];
}

__global__ __launch_bounds__(256) void graph_finalize_kernel(const float* __restrict__ gram_part, int nz,
                                                             const float* __restrict__ adj, const uint32_t* __restrict__ adj_bits,
                                                             float* __restrict__ G,
                                                             int V, int use_pose, int learn_graph, int mask_diag) {
    extern __shared__ __attribute__((aligned(16))) float s_n[];  // V squared norms (Gram diagonal)
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* gp = gram_part + (size_t)b * nz * V * V;
    if (learn_graph) {
        for (int j = tid; j < V; j += 256) s_n[j] = zsum(gp + (size_t)j * V + j, nz, V * V);
        __syncthreads();
    }
    const int i = blockIdx.y * 4 + wave;
    if (i >= V) return;
    constexpr int JPL = 4;  // columns per lane held in registers: V <= 256
    float sim[JPL], av[JPL];
    float ssum = 0.f, asum = 0.f;
#pragma unroll
    for (int q = 0; q < JPL; ++q) {
        const int j = lane + 64 * q;
        sim[q] = 0.f;
        av[q] = 0.f;
        if (j < V) {
            if (learn_graph) {
                const float g = zsum(gp + (size_t)i * V + j, nz, V * V);
                // similarity: 2 / (exp(sqrt(clamp(n_j + n_i - 2 g_ij, 1e-12))) + 1)
                float d2 = (s_n[j] + s_n[i]) - 2.f * g;
                d2 = fmaxf(d2, 1e-12f);
                sim[q] = 2.f / (expf(sqrtf(d2)) + 1.f);
                if (mask_diag && j == i) sim[q] = 0.f;  // ganet.py:259-268: self-loops masked out before the normalisation
                ssum += fabsf(sim[q]);
            }
            if (use_pose) {
                av[q] = adj_at(adj, adj_bits, (size_t)b * V + i, V, j);
                if (mask_diag && j == i) av[q] = 0.f;
                asum += fabsf(av[q]);
            }
        }
    }
    ssum = wave_sum(ssum);
    asum = wave_sum(asum);
    const float sden = fmaxf(ssum, 1e-12f), aden = fmaxf(asum, 1e-12f);
#pragma unroll
    for (int q = 0; q < JPL; ++q) {
        const int j = lane + 64 * q;
        if (j < V) {
            float g;
            if (learn_graph) {
                g = sim[q] / sden;
                if (use_pose) g = (av[q] / aden + g) / 2.f;
            } else {
                g = av[q] / aden;
            }
            G[((size_t)b * V + i) * V + j] = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Propagate. grid = (B, C/128), 128 threads; thread -> one channel. The graph is staged in LDS transposed
// (GT[u][v]) so the 8 graph values a register block needs are two broadcast ds_read_b128; the thread's h column
// sits in a thread-private LDS column (VT == 0; a register-resident variant VT > 0 exists but spills).
// 8 output rows are register-blocked: each h value feeds 8 FMAs.
constexpr int PROP_THREADS = 128;
constexpr int PROP_RB = 8;

template <int VT>  // VT > 0: compile-time V, h in registers; VT == 0: generic, h in LDS
__global__ __launch_bounds__(PROP_THREADS) void graph_propagate_kernel(
    const float* __restrict__ f, const float* __restrict__ h, const float* __restrict__ G,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, float one_minus_gamma, float gamma,
    float slope, float* __restrict__ out, lp16_t* __restrict__ out_lp, int Vrt, int C) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int V = VT > 0 ? VT : Vrt;
    const int Vp = (V + PROP_RB - 1) & ~(PROP_RB - 1);
    float* s_gt = s_mem;            // [V][Vp] transposed graph, zero padded columns
    float* s_h = s_mem + V * Vp;    // generic path only: [V][PROP_THREADS]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int c = blockIdx.y * PROP_THREADS + tid;
    const bool live = c < C;
    const int cc = live ? c : C - 1;
    const float* Gb = G + (size_t)b * V * V;
    for (int e = tid; e < V * Vp; e += PROP_THREADS) {
        const int u = e / Vp, v = e - u * Vp;
        s_gt[e] = v < V ? Gb[(size_t)v * V + u] : 0.f;
    }
    const float* hb = h + (size_t)b * V * C + cc;
    float hreg[VT > 0 ? VT : 1];
    if constexpr (VT > 0) {
#pragma unroll
        for (int u = 0; u < VT; ++u) hreg[u] = hb[(size_t)u * C];
    } else {
        for (int u = 0; u < V; ++u) s_h[u * PROP_THREADS + tid] = hb[(size_t)u * C];
    }
    const float sc = bn_scale[cc], sh = bn_shift[cc];
    const float* fb = f + (size_t)b * V * C + cc;
    __syncthreads();
    for (int v0 = 0; v0 < V; v0 += PROP_RB) {
        float acc[PROP_RB], fv[PROP_RB];
#pragma unroll
        for (int k = 0; k < PROP_RB; ++k) {
            acc[k] = 0.f;
            fv[k] = fb[(size_t)(v0 + k < V ? v0 + k : V - 1) * C];  // residual input, in flight under the FMA sweep
        }
        if constexpr (VT > 0) {
#pragma unroll
            for (int u = 0; u < VT; ++u) {
                const float4 g0 = *reinterpret_cast<const float4*>(&s_gt[u * Vp + v0]);
                const float4 g1 = *reinterpret_cast<const float4*>(&s_gt[u * Vp + v0 + 4]);
                const float hv = hreg[u];
                acc[0] = fmaf(g0.x, hv, acc[0]); acc[1] = fmaf(g0.y, hv, acc[1]);
                acc[2] = fmaf(g0.z, hv, acc[2]); acc[3] = fmaf(g0.w, hv, acc[3]);
                acc[4] = fmaf(g1.x, hv, acc[4]); acc[5] = fmaf(g1.y, hv, acc[5]);
                acc[6] = fmaf(g1.z, hv, acc[6]); acc[7] = fmaf(g1.w, hv, acc[7]);
            }
        } else {
            // unrolled by 4: 12 LDS reads are issued before the first FMA needs its operands (one wave per SIMD
            // has nobody else to hide the ~100-cycle LDS latency behind)
#pragma unroll 4
            for (int u = 0; u < V; ++u) {
                const float4 g0 = *reinterpret_cast<const float4*>(&s_gt[u * Vp + v0]);
                const float4 g1 = *reinterpret_cast<const float4*>(&s_gt[u * Vp + v0 + 4]);
                const float hv = s_h[u * PROP_THREADS + tid];
                acc[0] = fmaf(g0.x, hv, acc[0]); acc[1] = fmaf(g0.y, hv, acc[1]);
                acc[2] = fmaf(g0.z, hv, acc[2]); acc[3] = fmaf(g0.w, hv, acc[3]);
                acc[4] = fmaf(g1.x, hv, acc[4]); acc[5] = fmaf(g1.y, hv, acc[5]);
                acc[6] = fmaf(g1.z, hv, acc[6]); acc[7] = fmaf(g1.w, hv, acc[7]);
            }
        }
#pragma unroll
        for (int k = 0; k < PROP_RB; ++k) {
            const int v = v0 + k;
            if (v < V && live) {
                float y = fmaf(acc[k], sc, sh);
                y = y > 0.f ? y : slope * y;
                const float o = one_minus_gamma * fv[k] + gamma * y;
                const size_t idx = ((size_t)b * V + v) * C + c;
                out[idx] = o;
                if (out_lp) out_lp[idx] = f32_to_lp16(o);
            }
        }
    }
}


// Any V (the LDS-resident forms stop at V = 128 / V (V + 128) 4 bytes <= 160 KB): grid = (B, C / 128, ceil(V / 16)). A
// workgroup holds 16 rows of G (16 x V) in LDS and streams h[u][c], u = 0..V-1, from L2 (h is re-read once per 16 output rows).
constexpr int PROPT_ROWS = 16;
__global__ __launch_bounds__(PROP_THREADS) void graph_propagate_tiled_kernel(
    const float* __restrict__ f, const float* __restrict__ h, const float* __restrict__ G,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, float one_minus_gamma, float gamma,
    float slope, float* __restrict__ out, lp16_t* __restrict__ out_lp, int V, int C) {
    extern __shared__ __attribute__((aligned(16))) float s_gr[];   // [V][16]: s_gr[u * 16 + k] = G[v0 + k][u]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int c = blockIdx.y * PROP_THREADS + tid;
    const bool live = c < C;
    const int cc = live ? c : C - 1;
    const int v0 = blockIdx.z * PROPT_ROWS;
    const float* Gb = G + (size_t)b * V * V;
    for (int e = tid; e < V * PROPT_ROWS; e += PROP_THREADS) {
        const int k = e / V, u = e - k * V;     // coalesced over u along a graph row
        s_gr[u * PROPT_ROWS + k] = v0 + k < V ? Gb[(size_t)(v0 + k) * V + u] : 0.f;
    }
    __syncthreads();
    const float* hb = h + (size_t)b * V * C + cc;
    float acc[PROPT_ROWS];
#pragma unroll
    for (int k = 0; k < PROPT_ROWS; ++k) acc[k] = 0.f;
#pragma unroll 4
    for (int u = 0; u < V; ++u) {
        const float hv = hb[(size_t)u * C];
#pragma unroll
        for (int k4 = 0; k4 < PROPT_ROWS / 4; ++k4) {
            const float4 g4 = *reinterpret_cast<const float4*>(&s_gr[u * PROPT_ROWS + 4 * k4]);
            acc[4 * k4 + 0] = fmaf(g4.x, hv, acc[4 * k4 + 0]);
            acc[4 * k4 + 1] = fmaf(g4.y, hv, acc[4 * k4 + 1]);
            acc[4 * k4 + 2] = fmaf(g4.z, hv, acc[4 * k4 + 2]);
            acc[4 * k4 + 3] = fmaf(g4.w, hv, acc[4 * k4 + 3]);
        }
    }
    const float sc = bn_scale[cc], sh = bn_shift[cc];
#pragma unroll
    for (int k = 0; k < PROPT_ROWS; ++k) {
        const int v = v0 + k;
        if (v < V && live) {
            const size_t idx = ((size_t)b * V + v) * C + c;
            float y = fmaf(acc[k], sc, sh);
            y = y > 0.f ? y : slope * y;
            const float o = one_minus_gamma * f[idx] + gamma * y;
            out[idx] = o;
            if (out_lp) out_lp[idx] = f32_to_lp16(o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Propagate, MFMA form (V <= 128). grid = (B, C/128), 256 threads. Per workgroup: D'[c][v] = sum_u H[u][c] G[v][u]
// for a 128-channel slab, with v_mfma_f32_16x16x4_f32 (exact fp32): A operand = H^T (rows = channels), B operand =
// G^T (cols = graph rows), so a lane ends with 4 consecutive channels of one node -> float4 epilogue.
//   H slab  : V4 rows x 512 B, LDS-DMA straight from HBM (rows >= V from a zero block), 2 rows per 1-KiB piece
//   G^T     : [V4][NVF*16] fp32, zero padded
//   f       : this lane's residual inputs are fetched before the MFMA sweep and consumed in the epilogue
// wave w owns channel fragments {2w, 2w+1} of the slab and all NVF node fragments.
typedef __attribute__((address_space(3))) void gcn_lds_void_t;
typedef __attribute__((address_space(1))) const void gcn_gbl_void_t;
__device__ __attribute__((aligned(16))) uint4 g_gcn_zero16;

template <int NVF>
__global__ __launch_bounds__(256) void graph_propagate_mfma_kernel(
    const float* __restrict__ f, const float* __restrict__ h, const float* __restrict__ G,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, float one_minus_gamma, float gamma,
    float slope, float* __restrict__ out, lp16_t* __restrict__ out_lp, int V, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int V4 = (V + 3) & ~3;
    constexpr int VP = NVF * 16;
    float* s_h = reinterpret_cast<float*>(s_raw);                 // [V4 (even-padded)][128]
    const int hrows = (V4 + 1) & ~1;
    float* s_gt = s_h + hrows * 128;                              // [V4][VP]
    const int b = blockIdx.x;
    const int c0 = blockIdx.y * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // H slab by LDS-DMA: piece p holds rows 2p (lanes 0-31) and 2p+1 (lanes 32-63), 512 B each
    const unsigned char* hb = reinterpret_cast<const unsigned char*>(h + (size_t)b * V * C + c0);
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_gcn_zero16);
    for (int p = wave; p < hrows / 2; p += 4) {
        const int row = 2 * p + (lane >> 5);
        const unsigned char* src = row < V ? hb + (size_t)row * C * 4 + (lane & 31) * 16 : zsrc;
        __builtin_amdgcn_global_load_lds((gcn_gbl_void_t*)src, (gcn_lds_void_t*)(s_raw + p * 1024), 16, 0, 0);
    }
    // G^T, zero padded in both directions
    const float* Gb = G + (size_t)b * V * V;
    for (int e = tid; e < V4 * VP; e += 256) {
        const int u = e / VP, v = e - u * VP;
        s_gt[e] = (u < V && v < V) ? Gb[(size_t)v * V + u] : 0.f;
    }
    // residual inputs + BN constants of this lane's outputs
    const int vl = lane & 15, g4 = lane >> 4;
    float4 fin[2][NVF];
    float4 sc[2], sh[2];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        const int c = c0 + (wave * 2 + cf) * 16 + g4 * 4;
        sc[cf] = *reinterpret_cast<const float4*>(bn_scale + c);
        sh[cf] = *reinterpret_cast<const float4*>(bn_shift + c);
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) {
            const int v = vf * 16 + vl;
            fin[cf][vf] = v < V ? *reinterpret_cast<const float4*>(f + ((size_t)b * V + v) * C + c)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    f32x4_t acc[2][NVF];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) acc[cf][vf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nt = V4 >> 2;
    for (int t = 0; t < nt; ++t) {
        const int u = 4 * t + g4;
        float a[2], bq[NVF];
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) a[cf] = s_h[u * 128 + (wave * 2 + cf) * 16 + vl];
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) bq[vf] = s_gt[u * VP + vf * 16 + vl];
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
#pragma unroll
            for (int vf = 0; vf < NVF; ++vf)
                acc[cf][vf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cf], bq[vf], acc[cf][vf], 0, 0, 0);
    }
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        const int c = c0 + (wave * 2 + cf) * 16 + g4 * 4;
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) {
            const int v = vf * 16 + vl;
            if (v < V) {
                const float scv[4] = {sc[cf].x, sc[cf].y, sc[cf].z, sc[cf].w};
                const float shv[4] = {sh[cf].x, sh[cf].y, sh[cf].z, sh[cf].w};
                const float fv[4] = {fin[cf][vf].x, fin[cf][vf].y, fin[cf][vf].z, fin[cf][vf].w};
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = fmaf(acc[cf][vf][r], scv[r], shv[r]);
                    y = y > 0.f ? y : slope * y;
                    o[r] = one_minus_gamma * fv[r] + gamma * y;
                }
                const size_t idx = ((size_t)b * V + v) * C + c;
                *reinterpret_cast<float4*>(out + idx) = make_float4(o[0], o[1], o[2], o[3]);
                if (out_lp) {
                    uint2 pk;
                    pk.x = pack_lp16x2(o[0], o[1]);
                    pk.y = pack_lp16x2(o[2], o[3]);
                    *reinterpret_cast<uint2*>(out_lp + idx) = pk;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Propagate, streaming MFMA form (V <= 64, V % 4 == 0: the MARS / PRID configurations). grid = (B, C/128), 128
// threads; each WAVE owns 64 channels of one tracklet and streams them HBM -> registers -> MFMA -> HBM, no LDS round
// trip for h or f:
//   * h: one 16-byte load per lane per 4 graph rows u: lane (i = lane&15, kg = lane>>4) reads
//     h[4t+kg][c0 + 4 sigma(i) .. +3] -> 256 contiguous bytes per row, whole cache lines. The four floats feed FOUR
//     MFMAs (v_mfma_f32_16x16x4_f32, exact fp32) whose row i stands for channel c0 + 4 sigma(i) + j: the channel <->
//     MFMA-row assignment is free, so it is chosen to match the load; sigma = the 4 x 4 index transpose, which makes
//     the lanes kg = 0..3 of every epilogue access cover 64 contiguous bytes.
//   * D_j[row 4 kg + r][col v] = channel c0 + 4 kg + 16 r + j of node v: f is read and the output written as float4s.
//   * G (V x V, L2-resident) is copied as it lies into LDS by LDS-DMA (B operand G[v][u] by ds_read_b32; rows >= V of
//     the padded node fragment read whatever follows and only feed output nodes that are never stored).
// Issue order = arrival order: G, BN constants, h in sweep order, f. EVERY load of the prologue is inline asm with a
// hand-counted s_waitcnt tied to its destination registers: left to hipcc, the loads get sunk behind one another (G
// ends up last in the queue, one load lands under a branch with a vmcnt(0) in front of the sweep); this way 46 KB per
// workgroup is in flight at once and step t of the sweep waits for h row-group t alone, so the matrix work overlaps
// the rest of the stream.
__device__ inline f32x4_t gload16(const void* p) {
    f32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// at most n younger vector-memory operations still in flight (n folds to a constant in the unrolled callers)
__device__ inline void landed(int n, f32x4_t& v) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" : "+v"(v) : : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(v) : : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(v) : : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(v) : : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" : "+v"(v) : : "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(v) : : "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" : "+v"(v) : : "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" : "+v"(v) : : "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" : "+v"(v) : : "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" : "+v"(v) : : "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" : "+v"(v) : : "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" : "+v"(v) : : "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" : "+v"(v) : : "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" : "+v"(v) : : "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" : "+v"(v) : : "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" : "+v"(v) : : "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" : "+v"(v) : : "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" : "+v"(v) : : "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" : "+v"(v) : : "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" : "+v"(v) : : "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" : "+v"(v) : : "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" : "+v"(v) : : "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" : "+v"(v) : : "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" : "+v"(v) : : "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" : "+v"(v) : : "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" : "+v"(v) : : "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" : "+v"(v) : : "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" : "+v"(v) : : "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" : "+v"(v) : : "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" : "+v"(v) : : "memory"); break;
        case 31: asm volatile("s_waitcnt vmcnt(31)" : "+v"(v) : : "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" : "+v"(v) : : "memory"); break;
        case 33: asm volatile("s_waitcnt vmcnt(33)" : "+v"(v) : : "memory"); break;
        case 34: asm volatile("s_waitcnt vmcnt(34)" : "+v"(v) : : "memory"); break;
        case 35: asm volatile("s_waitcnt vmcnt(35)" : "+v"(v) : : "memory"); break;
        case 36: asm volatile("s_waitcnt vmcnt(36)" : "+v"(v) : : "memory"); break;
        case 37: asm volatile("s_waitcnt vmcnt(37)" : "+v"(v) : : "memory"); break;
        case 38: asm volatile("s_waitcnt vmcnt(38)" : "+v"(v) : : "memory"); break;
        case 39: asm volatile("s_waitcnt vmcnt(39)" : "+v"(v) : : "memory"); break;
        case 40: asm volatile("s_waitcnt vmcnt(40)" : "+v"(v) : : "memory"); break;
        case 41: asm volatile("s_waitcnt vmcnt(41)" : "+v"(v) : : "memory"); break;
        case 42: asm volatile("s_waitcnt vmcnt(42)" : "+v"(v) : : "memory"); break;
        case 43: asm volatile("s_waitcnt vmcnt(43)" : "+v"(v) : : "memory"); break;
        case 44: asm volatile("s_waitcnt vmcnt(44)" : "+v"(v) : : "memory"); break;
        case 45: asm volatile("s_waitcnt vmcnt(45)" : "+v"(v) : : "memory"); break;
        case 46: asm volatile("s_waitcnt vmcnt(46)" : "+v"(v) : : "memory"); break;
        case 47: asm volatile("s_waitcnt vmcnt(47)" : "+v"(v) : : "memory"); break;
        case 48: asm volatile("s_waitcnt vmcnt(48)" : "+v"(v) : : "memory"); break;
        case 49: asm volatile("s_waitcnt vmcnt(49)" : "+v"(v) : : "memory"); break;
        case 50: asm volatile("s_waitcnt vmcnt(50)" : "+v"(v) : : "memory"); break;
        case 51: asm volatile("s_waitcnt vmcnt(51)" : "+v"(v) : : "memory"); break;
        case 52: asm volatile("s_waitcnt vmcnt(52)" : "+v"(v) : : "memory"); break;
        case 53: asm volatile("s_waitcnt vmcnt(53)" : "+v"(v) : : "memory"); break;
        case 54: asm volatile("s_waitcnt vmcnt(54)" : "+v"(v) : : "memory"); break;
        case 55: asm volatile("s_waitcnt vmcnt(55)" : "+v"(v) : : "memory"); break;
        case 56: asm volatile("s_waitcnt vmcnt(56)" : "+v"(v) : : "memory"); break;
        case 57: asm volatile("s_waitcnt vmcnt(57)" : "+v"(v) : : "memory"); break;
        case 58: asm volatile("s_waitcnt vmcnt(58)" : "+v"(v) : : "memory"); break;
        case 59: asm volatile("s_waitcnt vmcnt(59)" : "+v"(v) : : "memory"); break;
        case 60: asm volatile("s_waitcnt vmcnt(60)" : "+v"(v) : : "memory"); break;
        case 61: asm volatile("s_waitcnt vmcnt(61)" : "+v"(v) : : "memory"); break;
        case 62: asm volatile("s_waitcnt vmcnt(62)" : "+v"(v) : : "memory"); break;
        case 63: asm volatile("s_waitcnt vmcnt(63)" : "+v"(v) : : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); break;
    }
}
__device__ inline void tie(f32x4_t& v) { asm volatile("" : "+v"(v)); }

template <int PS_NT, int NWV>  // V = 4 PS_NT exactly; NWV waves (64 channels each) per workgroup
__global__ __launch_bounds__(64 * NWV) void graph_propagate_stream_kernel(
    const float* __restrict__ f, const float* __restrict__ h, const float* __restrict__ G,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, float one_minus_gamma, float gamma,
    float slope, float* __restrict__ out, lp16_t* __restrict__ out_lp, int C) {
    extern __shared__ __attribute__((aligned(16))) float s_g[];  // [16*NVF][V]
    constexpr int V = 4 * PS_NT, NVF = (PS_NT + 3) / 4;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = (blockIdx.y * NWV + wave) * 64;
    const int i16 = lane & 15, kg = lane >> 4;

    {
        const unsigned char* Gb = reinterpret_cast<const unsigned char*>(G + (size_t)b * V * V);
        const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
        constexpr int GBYTES = V * V * 4, NPIECE = (GBYTES + 1023) / 1024;
#pragma unroll
        for (int pc = 0; pc < (NPIECE + NWV - 1) / NWV; ++pc) {  // every wave issues the same number of pieces (vmcnt below)
            const int piece = NWV * pc + wave;
            const int off = piece * 1024 + lane * 16;
            const bool real = piece < NPIECE;
            dma16(real && off < GBYTES ? Gb + off : zsrc,
                  reinterpret_cast<unsigned char*>(s_g) + (real ? piece : NPIECE) * 1024);  // spare KiB for the odd one
        }
    }
    const size_t node0 = (size_t)b * V;
    const int cl = c0 + 4 * kg;  // float4 r of this lane = channels cl + 16 r .. +3
    f32x4_t sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc[r] = gload16(bn_scale + cl + 16 * r);
        sh[r] = gload16(bn_shift + cl + 16 * r);
    }
    const int sig = 4 * (i16 & 3) + (i16 >> 2);
    const float* hb = h + (size_t)b * V * C + c0 + 4 * sig;
    f32x4_t hreg[PS_NT];
#pragma unroll
    for (int t = 0; t < PS_NT; ++t) hreg[t] = gload16(hb + (size_t)(4 * t + kg) * C);
    f32x4_t fin[NVF][4];
#pragma unroll
    for (int vf = 0; vf < NVF; ++vf) {
        const int v = min(vf * 16 + i16, V - 1);  // clamped: nodes >= V are computed on a copy and never stored
#pragma unroll
        for (int r = 0; r < 4; ++r) fin[vf][r] = gload16(f + (node0 + v) * C + cl + 16 * r);
    }
    constexpr int YOUNGER_THAN_G = 8 + PS_NT + 4 * NVF;
    wait_vmcnt<YOUNGER_THAN_G>();  // the DMA pieces are the oldest entries of the queue
    wg_barrier();

    f32x4_t acc[NVF][4];
#pragma unroll
    for (int vf = 0; vf < NVF; ++vf)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[vf][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < PS_NT; ++t) {
        float gq[NVF];
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) gq[vf] = s_g[(vf * 16 + i16) * V + 4 * t + kg];
        landed(PS_NT - 1 - t + 4 * NVF, hreg[t]);
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[vf][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[t][j], gq[vf], acc[vf][j], 0, 0, 0);
    }
    // epilogue: D_j row 4 kg + r = channel c0 + 4 sigma(4 kg + r) + j = cl + 16 r + j -> float4 r = {acc[vf][0..3][r]}
    landed(0, fin[0][0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tie(sc[r]);
        tie(sh[r]);
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) tie(fin[vf][r]);
    }
#pragma unroll
    for (int vf = 0; vf < NVF; ++vf) {
        const int v = vf * 16 + i16;
        if (v >= V) continue;
        float* op = out + (node0 + v) * C + cl;
        lp16_t* lp = out_lp ? out_lp + (node0 + v) * C + cl : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = fmaf(acc[vf][j][r], sc[r][j], sh[r][j]);
                y = y > 0.f ? y : slope * y;
                o[j] = one_minus_gamma * fin[vf][r][j] + gamma * y;
            }
            *reinterpret_cast<float4*>(op + 16 * r) = make_float4(o[0], o[1], o[2], o[3]);
            if (lp) *reinterpret_cast<uint2*>(lp + 16 * r) = make_uint2(pack_lp16x2(o[0], o[1]), pack_lp16x2(o[2], o[3]));
        }
    }
}


// P = G f, nothing else: the message pass applied to the layer INPUT (vmgn.py:168 with the Linear commuted behind it,
// G (f W^T) = (G f) W^T), written ONCE in the dtype the following GEMM (agrl_graph_linear_mix) consumes. Same streaming
// structure as graph_propagate_stream_kernel -- G by LDS-DMA at the head of the queue, one 16-byte load per lane per 4 graph
// rows feeding four exact-fp32 MFMAs, the free channel <-> MFMA-row assignment chosen so that loads are whole cache lines --
// minus the residual / BatchNorm operands: f crosses HBM once (V C 4 bytes per tracklet in, V C 2 or 4 out).
// MODE 0: fp32, 1: the 16-bit type, 2: fp32-sized rows PRE-SPLIT for the split-fp16 GEMM (AGRL_F32H3P: the layout of
// agrl_split16_weights_inloop -- lane group kg's float4s 2 t, 2 t + 1 are exactly chunk kg of 32-channel group t, so the halves are
// formed here, once, and agrl_graph_linear_mix's k-loop has no VALU work left)
template <int PS_NT, int NWV, int MODE>
__global__ __launch_bounds__(64 * NWV) void graph_apply_stream_kernel(const float* __restrict__ f, const float* __restrict__ G,
                                                                      float* __restrict__ out, lp16_t* __restrict__ out_lp, int C) {
    extern __shared__ __attribute__((aligned(16))) float s_g[];  // [16*NVF][V]
    constexpr int V = 4 * PS_NT, NVF = (PS_NT + 3) / 4;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = (blockIdx.y * NWV + wave) * 64;
    const int i16 = lane & 15, kg = lane >> 4;
    {
        const unsigned char* Gb = reinterpret_cast<const unsigned char*>(G + (size_t)b * V * V);
        const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
        constexpr int GBYTES = V * V * 4, NPIECE = (GBYTES + 1023) / 1024;
#pragma unroll
        for (int pc = 0; pc < (NPIECE + NWV - 1) / NWV; ++pc) {  // every wave issues the same number of pieces (vmcnt below)
            const int piece = NWV * pc + wave;
            const int off = piece * 1024 + lane * 16;
            const bool real = piece < NPIECE;
            dma16(real && off < GBYTES ? Gb + off : zsrc,
                  reinterpret_cast<unsigned char*>(s_g) + (real ? piece : NPIECE) * 1024);  // spare KiB for the odd one
        }
    }
    const size_t node0 = (size_t)b * V;
    const int cl = c0 + 4 * kg;  // float4 r of this lane's results = channels cl + 16 r .. +3
    const int sig = 4 * (i16 & 3) + (i16 >> 2);
    const float* fb = f + (size_t)b * V * C + c0 + 4 * sig;
    f32x4_t freg[PS_NT];
#pragma unroll
    for (int t = 0; t < PS_NT; ++t) freg[t] = gload16(fb + (size_t)(4 * t + kg) * C);
    wait_vmcnt<PS_NT>();  // the DMA pieces are the oldest entries of the queue
    wg_barrier();

    f32x4_t acc[NVF][4];
#pragma unroll
    for (int vf = 0; vf < NVF; ++vf)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[vf][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < PS_NT; ++t) {
        float gq[NVF];
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) gq[vf] = s_g[(vf * 16 + i16) * V + 4 * t + kg];
        landed(PS_NT - 1 - t, freg[t]);
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[vf][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(freg[t][j], gq[vf], acc[vf][j], 0, 0, 0);
    }
    // D_j row 4 kg + r = channel c0 + 4 sigma(4 kg + r) + j = cl + 16 r + j -> float4 r = {acc[vf][0..3][r]}
#pragma unroll
    for (int vf = 0; vf < NVF; ++vf) {
        const int v = vf * 16 + i16;
        if (v >= V) continue;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4 c0_ = make_uint4(__float_as_uint(acc[vf][0][2 * t]), __float_as_uint(acc[vf][1][2 * t]), __float_as_uint(acc[vf][2][2 * t]),
                                             __float_as_uint(acc[vf][3][2 * t]));
                const uint4 c1_ = make_uint4(__float_as_uint(acc[vf][0][2 * t + 1]), __float_as_uint(acc[vf][1][2 * t + 1]),
                                             __float_as_uint(acc[vf][2][2 * t + 1]), __float_as_uint(acc[vf][3][2 * t + 1]));
                uint4 hi, lo;
                Frag<f32h_t>::split8(c0_, c1_, hi, lo);
                unsigned char* dst = reinterpret_cast<unsigned char*>(out + (node0 + v) * C + c0 + 32 * t) + kg * 16;
                *reinterpret_cast<uint4*>(dst) = hi;
                *reinterpret_cast<uint4*>(dst + 64) = lo;
            }
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float o0 = acc[vf][0][r], o1 = acc[vf][1][r], o2 = acc[vf][2][r], o3 = acc[vf][3][r];
            if constexpr (MODE == 1) *reinterpret_cast<uint2*>(out_lp + (node0 + v) * C + cl + 16 * r) = make_uint2(pack_lp16x2(o0, o1), pack_lp16x2(o2, o3));
            else *reinterpret_cast<float4*>(out + (node0 + v) * C + cl + 16 * r) = make_float4(o0, o1, o2, o3);
        }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Graph + P = G f of ONE TRACKLET PER WORKGROUP (many tracklets per GPU: B >= 128, so that one workgroup per tracklet fills
// the chip): the whole HBM-bound part of the commuted GraphLayer in one launch, no Gram partials and no G round trip.
//   1  Gram: wave w streams channels [w C/4, (w+1) C/4) of the tracklet's V rows straight into exact-fp32 MFMAs (lane = row
//      i16 of a 16-row fragment, k-group kg: one 16-byte load per fragment and 16 channels feeds four k-steps of the ten
//      fragment pairs I <= J); the four wave partials meet in LDS and are added in wave order (deterministic);
//   2  graph: d2 -> sim -> row-L1 normalise -> mix with the pose graph (graph_finalize_kernel's arithmetic), into LDS;
//   3  P = G f: the streaming message pass of graph_apply_stream_kernel over the tracklet's channels, 64 per wave and step --
//      f comes a second time, out of the memory-side cache (458 KB per tracklet, just read).
// f crosses HBM once per tracklet (V C 4 bytes), P leaves once. The Gram is summed in a different order than the
// slice-partial form (4 wave partials of C/4 channels, not 16 slices of 128): the graph agrees to fp32 roundoff.
// out[b][e] = sum over the nz slice partials, z ascending
__global__ __launch_bounds__(256) void gram_sum_kernel(const float* __restrict__ part, int nz, float* __restrict__ out, int vv, size_t total) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t b = e / vv;
    out[e] = zsum(part + b * nz * vv + (e - b * vv), nz, vv);
}

constexpr int GT_WAVES = 8;   // two waves per SIMD: one wave's exact-fp32 MFMA chain covers the other's memory latency
template <int PS_NT, bool LP>
__global__ __launch_bounds__(64 * GT_WAVES) void graph_tracklet_kernel(const float* __restrict__ f, const float* __restrict__ adj,
                                                                       const uint32_t* __restrict__ adj_bits, float* __restrict__ G_out, float* __restrict__ out, lp16_t* __restrict__ out_lp,
                                                                       int C, int use_pose, int learn_graph, int mask_diag) {
    constexpr int V = 4 * PS_NT, NVF = (PS_NT + 3) / 4, VP = NVF * 16, NPAIR = NVF * (NVF + 1) / 2, NT = 64 * GT_WAVES;
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_part = s_mem;                              // [GT_WAVES][NPAIR][64 lanes][4]  (phase 1 -> 2)
    float* s_gram = s_part + GT_WAVES * NPAIR * 256;    // [VP][VP + 1]
    float* s_G = s_gram + VP * (VP + 1);                // [VP][V]   rows >= V zero
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    const size_t node0 = (size_t)b * V;
    // ---- 1: Gram partial of this wave's channel slice (C / GT_WAVES channels), loads two 16-channel steps ahead
    if (learn_graph) {
        f32x4_t acc[NPAIR];
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int cq = C / GT_WAVES;
        const float* base[NVF];
#pragma unroll
        for (int I = 0; I < NVF; ++I) base[I] = f + (node0 + min(16 * I + i16, V - 1)) * C + wave * cq + 4 * kg;   // rows >= V: a copy, never used
        float4 cur[NVF], nx1[NVF], nx2[NVF];
#pragma unroll
        for (int I = 0; I < NVF; ++I) {
            cur[I] = *reinterpret_cast<const float4*>(base[I]);
            nx1[I] = *reinterpret_cast<const float4*>(base[I] + (16 < cq ? 16 : 0));
        }
        for (int c = 0; c < cq; c += 16) {
            const int c2 = c + 32 < cq ? c + 32 : c;
#pragma unroll
            for (int I = 0; I < NVF; ++I) nx2[I] = *reinterpret_cast<const float4*>(base[I] + c2);
            // k-step outermost: consecutive MFMAs go to different accumulators (a dependent v_mfma_f32_16x16x4_f32 waits 40
            // cycles, an independent one issues after 32)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int q = 0;
#pragma unroll
                for (int I = 0; I < NVF; ++I)
#pragma unroll
                    for (int J = I; J < NVF; ++J, ++q) {
                        const float a = e == 0 ? cur[I].x : e == 1 ? cur[I].y : e == 2 ? cur[I].z : cur[I].w;
                        const float bb = e == 0 ? cur[J].x : e == 1 ? cur[J].y : e == 2 ? cur[J].z : cur[J].w;
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc[q], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int I = 0; I < NVF; ++I) {
                cur[I] = nx1[I];
                nx1[I] = nx2[I];
            }
        }
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) *reinterpret_cast<f32x4_t*>(s_part + ((wave * NPAIR + q) * 64 + lane) * 4) = acc[q];
    }
    __syncthreads();
    if (learn_graph) {
        // D of pair (I, J): element r of lane l = Gram[16 I + 4 (l >> 4) + r][16 J + (l & 15)]; a thread adds the wave partials
        // of its (pair, lane, r) in wave order and writes both mirror images
        for (int e = tid; e < NPAIR * 256; e += NT) {
            const int q = e >> 8, l = e & 63, r = (e >> 6) & 3;
            float g = 0.f;
#pragma unroll
            for (int w = 0; w < GT_WAVES; ++w) g += s_part[((w * NPAIR + q) * 64 + l) * 4 + r];
            int I = 0, rem = q;   // q -> (I, J), I <= J, row-major over the upper triangle
            while (rem >= NVF - I) { rem -= NVF - I; ++I; }
            const int J = I + rem;
            const int row = 16 * I + 4 * (l >> 4) + r, col = 16 * J + (l & 15);
            s_gram[row * (VP + 1) + col] = g;
            if (I != J) s_gram[col * (VP + 1) + row] = g;
        }
    }
    __syncthreads();
    // ---- 2: the graph, one wavefront per row (V <= 64: one column per lane) -- graph_finalize_kernel's arithmetic
    for (int i = wave; i < VP; i += GT_WAVES) {
        float g = 0.f;
        if (i < V) {
            const bool live = lane < V;
            float sim = 0.f, av = 0.f;
            if (learn_graph && live) {
                float d2 = (s_gram[lane * (VP + 1) + lane] + s_gram[i * (VP + 1) + i]) - 2.f * s_gram[i * (VP + 1) + lane];
                d2 = fmaxf(d2, 1e-12f);
                sim = 2.f / (expf(sqrtf(d2)) + 1.f);
                if (mask_diag && lane == i) sim = 0.f;
            }
            if (use_pose && live) {
                av = adj_at(adj, adj_bits, node0 + i, V, lane);
                if (mask_diag && lane == i) av = 0.f;
            }
            const float sden = fmaxf(wave_sum(fabsf(sim)), 1e-12f), aden = fmaxf(wave_sum(fabsf(av)), 1e-12f);
            if (learn_graph) {
                g = sim / sden;
                if (use_pose) g = (av / aden + g) / 2.f;
            } else {
                g = av / aden;
            }
            if (live && G_out) G_out[(node0 + i) * V + lane] = g;
        }
        if (lane < V) s_G[i * V + lane] = i < V ? g : 0.f;
    }
    __syncthreads();
    // ---- 3: P = G f, 64 channels per wave and step (the lane <-> channel assignment of graph_apply_stream_kernel); the next
    // step's rows are requested before this step's MFMAs
    const int sig = 4 * (i16 & 3) + (i16 >> 2);
    const float* fb = f + node0 * C + 4 * sig + (size_t)kg * C;
    float4 freg[PS_NT], fnext[PS_NT];
    int c0 = wave * 64;
    if (c0 < C) {
#pragma unroll
        for (int t = 0; t < PS_NT; ++t) freg[t] = *reinterpret_cast<const float4*>(fb + c0 + (size_t)(4 * t) * C);
    }
    for (; c0 < C; c0 += 64 * GT_WAVES) {
        const int cn = c0 + 64 * GT_WAVES < C ? c0 + 64 * GT_WAVES : c0;
#pragma unroll
        for (int t = 0; t < PS_NT; ++t) fnext[t] = *reinterpret_cast<const float4*>(fb + cn + (size_t)(4 * t) * C);
        f32x4_t acc[NVF][4];
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[vf][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < PS_NT; ++t) {
            float gq[NVF];
#pragma unroll
            for (int vf = 0; vf < NVF; ++vf) gq[vf] = s_G[(vf * 16 + i16) * V + 4 * t + kg];
#pragma unroll
            for (int vf = 0; vf < NVF; ++vf) {
                acc[vf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(freg[t].x, gq[vf], acc[vf][0], 0, 0, 0);
                acc[vf][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(freg[t].y, gq[vf], acc[vf][1], 0, 0, 0);
                acc[vf][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(freg[t].z, gq[vf], acc[vf][2], 0, 0, 0);
                acc[vf][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(freg[t].w, gq[vf], acc[vf][3], 0, 0, 0);
            }
        }
        const int cl = c0 + 4 * kg;   // D_j row 4 kg + r = channel c0 + 4 sigma(4 kg + r) + j = cl + 16 r + j
#pragma unroll
        for (int vf = 0; vf < NVF; ++vf) {
            const int v = vf * 16 + i16;
            if (v >= V) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float o0 = acc[vf][0][r], o1 = acc[vf][1][r], o2 = acc[vf][2][r], o3 = acc[vf][3][r];
                if constexpr (LP) *reinterpret_cast<uint2*>(out_lp + (node0 + v) * C + cl + 16 * r) = make_uint2(pack_lp16x2(o0, o1), pack_lp16x2(o2, o3));
                else *reinterpret_cast<float4*>(out + (node0 + v) * C + cl + 16 * r) = make_float4(o0, o1, o2, o3);
            }
        }
#pragma unroll
        for (int t = 0; t < PS_NT; ++t) freg[t] = fnext[t];
    }
}

// ---------------------------------------------------------------------------------------------------
// Pose adjacency on the device: generate_graph + adj_graph(method 'same'), torchreid/dataset_loader.py:218-388.
// One workgroup per tracklet. Per frame and body part (head / body / leg keypoint groups) the confident keypoints'
// y coordinates are bucketed into horizontal stripes (bisect_right on the stripe borders, clamped), the stripes
// of a part are made contiguous, every stripe also marks its coarser pyramid ancestors -> one node bitmask per
// (frame, part). adj[i][j] = 1 iff i != j and some part marks both nodes. Borders are evaluated in fp64 exactly as
// numpy's arange(0, height + 1, height / num_split) does.
__global__ __launch_bounds__(256) void pose_adjacency_kernel(const float* __restrict__ poses, const unsigned char* __restrict__ detected,
                                                             float* __restrict__ adj, uint32_t* __restrict__ adj_bits, int S, int num_split, int levels,
                                                             int pyramid, int P, double height, float threshold) {
    __shared__ unsigned s_mask[3 * 64];  // [frame][part] node bitmask (P <= 31), S <= 64
    const int b = blockIdx.x, tid = threadIdx.x;
    const int V = S * P;
    if (tid < 3 * S) {
        const int t = tid / 3, part = tid - 3 * t;
        unsigned mask = 0;
        if (detected[(size_t)b * S + t]) {
            // keypoint groups of the reference: head {0,1,14,15,16,17}, body {2..7}, leg {8..13}
            const int ids[3][6] = {{0, 1, 14, 15, 16, 17}, {2, 3, 4, 5, 6, 7}, {8, 9, 10, 11, 12, 13}};
            const float* kp = poses + ((size_t)b * S + t) * 18 * 3;
            const double step = height / (double)num_split;
            int lo = 1 << 30, hi = -1;
            for (int i = 0; i < 6; ++i) {
                const float* q = kp + ids[part][i] * 3;
                if (q[2] > threshold) {
                    // bisect_right(borders, y): number of borders <= y; borders = 0, step, 2 step, .. (< height + 1)
                    int cnt = 0;
                    for (int e = 0;; ++e) {
                        const double border = (double)e * step;
                        if (!(border < height + 1.0)) break;
                        if (border <= (double)q[1]) ++cnt;
                    }
                    const int sid = min(num_split, max(1, cnt));
                    lo = min(lo, sid);
                    hi = max(hi, sid);
                }
            }
            for (int sid = lo; sid <= hi; ++sid) {   // contiguous stripes (empty when no confident keypoint)
                mask |= 1u << (sid - 1);
                if (pyramid) {
                    for (int i = 1; i <= levels; ++i) {
                        const int anc = (sid + (1 << i) - 1) >> i;                                  // ceil(sid / 2^i)
                        mask |= 1u << (anc + (1 << (levels + 1)) - (1 << (levels + 1 - i)) - 1);
                    }
                }
            }
        }
        s_mask[tid] = mask;
    }
    __syncthreads();
    auto edge = [&](int i, int j) {
        const int ti = i / P, pi = i - ti * P, tj = j / P, pj = j - tj * P;
        bool on = false;
        if (i != j) {
#pragma unroll
            for (int part = 0; part < 3; ++part)
                on = on || (((s_mask[ti * 3 + part] >> pi) & 1u) && ((s_mask[tj * 3 + part] >> pj) & 1u));
        }
        return on;
    };
    if (adj) {
        float* ab = adj + (size_t)b * V * V;
        for (int e = tid; e < V * V; e += 256) {
            const int i = e / V, j = e - i * V;
            ab[e] = edge(i, j) ? 1.f : 0.f;
        }
    }
    if (adj_bits) {   // bit j & 31 of word j >> 5 of row i: 56 x 56 nodes = 448 bytes instead of 12.5 KB
        const int W = (V + 31) >> 5;
        uint32_t* bb = adj_bits + (size_t)b * V * W;
        for (int e = tid; e < V * W; e += 256) {
            const int i = e / W, w = e - i * W;
            uint32_t word = 0;
            for (int q = 0; q < 32; ++q) {
                const int j = 32 * w + q;
                if (j < V && edge(i, j)) word |= 1u << q;
            }
            bb[e] = word;
        }
    }
}

// fp32 {0, 1} adjacency (B,V,V) (what the reference's loader produces, dataset_loader.py:345-388) -> the bit-packed form
__global__ __launch_bounds__(256) void adjacency_pack_kernel(const float* __restrict__ adj, uint32_t* __restrict__ bits, int V, size_t total_words) {
    const int W = (V + 31) >> 5;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total_words; e += (size_t)gridDim.x * 256) {
        const size_t row = e / W;
        const int w = (int)(e - row * W);
        uint32_t word = 0;
        for (int q = 0; q < 32; ++q) {
            const int j = 32 * w + q;
            if (j < V && adj[row * V + j] != 0.f) word |= 1u << q;
        }
        bits[e] = word;
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
    if (lds > 64 * 1024) {  // V >= 125 (e.g. seq_len 20 x 7 parts): above the default dynamic-LDS limit of a launch
        hipError_t e = hipFuncSetAttribute((const void*)gram_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        AGRL_CHECK_ARG(e == hipSuccess, "agrl_graph_gram: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(gram_kernel<false>, dim3(B, nz), dim3(256), lds, (hipStream_t)stream, f, (const float*)nullptr, gram_part, V, C, nz);
    AGRL_CHECK_LAUNCH("agrl_graph_gram");
    return 0;
}

extern "C" int agrl_graph_pair_product(const float* a, const float* b, float* part, float* out, int B, int V, int C, agrl_stream_t stream) {
    AGRL_CHECK_ARG(a && b && part && out, "agrl_graph_pair_product: null pointer");
    AGRL_CHECK_ARG(B > 0 && V > 0 && C > 0 && C % GRAM_CS == 0, "agrl_graph_pair_product: bad shape (C must be a multiple of %d)", GRAM_CS);
    const int Vp = (V + 15) & ~15;
    const size_t lds = (size_t)2 * Vp * GRAM_ROWB;
    AGRL_CHECK_ARG(lds <= 160 * 1024, "agrl_graph_pair_product: V=%d too large (two V x 128 slices must fit the LDS)", V);
    const int nz = C / GRAM_CS;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gram_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        AGRL_CHECK_ARG(e == hipSuccess, "agrl_graph_pair_product: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(gram_kernel<true>, dim3(B, nz), dim3(256), lds, (hipStream_t)stream, a, b, part, V, C, nz);
    AGRL_CHECK_LAUNCH("agrl_graph_pair_product");
    const size_t total = (size_t)B * V * V;
    hipLaunchKernelGGL(gram_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, nz, out, V * V, total);
    AGRL_CHECK_LAUNCH("agrl_graph_pair_product(sum)");
    return 0;
}

extern "C" int agrl_graph_finalize(const float* gram_part, int nz, const float* adj, float* G, int B, int V,
                                   int use_pose, int learn_graph, int mask_diag, agrl_stream_t stream) {
    AGRL_CHECK_ARG(G && B > 0 && V > 0, "agrl_graph_finalize: bad arguments");
    AGRL_CHECK_ARG(use_pose || learn_graph, "agrl_graph_finalize: use_pose or learn_graph must be set");
    AGRL_CHECK_ARG(!use_pose || adj, "agrl_graph_finalize: use_pose needs adj");
    AGRL_CHECK_ARG(!learn_graph || (gram_part && nz > 0), "agrl_graph_finalize: learn_graph needs the Gram partials");
    AGRL_CHECK_ARG(V <= 256, "agrl_graph_finalize: V=%d > 256 not supported", V);
    hipLaunchKernelGGL(graph_finalize_kernel, dim3(B, cdiv(V, 4)), dim3(256), (size_t)V * sizeof(float),
                       (hipStream_t)stream, gram_part, nz, adj, (const uint32_t*)nullptr, G, V, use_pose, learn_graph, mask_diag);
    AGRL_CHECK_LAUNCH("agrl_graph_finalize");
    return 0;
}

extern "C" int agrl_graph_finalize_bits(const float* gram_part, int nz, const uint32_t* adj_bits, float* G, int B, int V,
                                        int use_pose, int learn_graph, int mask_diag, agrl_stream_t stream) {
    AGRL_CHECK_ARG(G && B > 0 && V > 0, "agrl_graph_finalize_bits: bad arguments");
    AGRL_CHECK_ARG(use_pose || learn_graph, "agrl_graph_finalize_bits: use_pose or learn_graph must be set");
    AGRL_CHECK_ARG(!use_pose || adj_bits, "agrl_graph_finalize_bits: use_pose needs the packed adjacency");
    AGRL_CHECK_ARG(!learn_graph || (gram_part && nz > 0), "agrl_graph_finalize_bits: learn_graph needs the Gram partials");
    AGRL_CHECK_ARG(V <= 256, "agrl_graph_finalize_bits: V=%d > 256 not supported", V);
    hipLaunchKernelGGL(graph_finalize_kernel, dim3(B, cdiv(V, 4)), dim3(256), (size_t)V * sizeof(float),
                       (hipStream_t)stream, gram_part, nz, (const float*)nullptr, adj_bits, G, V, use_pose, learn_graph, mask_diag);
    AGRL_CHECK_LAUNCH("agrl_graph_finalize_bits");
    return 0;
}

extern "C" int agrl_graph_apply(const float* G, const float* f, void* out, int out_dtype, int B, int V, int C, agrl_stream_t stream) {
    AGRL_CHECK_ARG(G && f && out, "agrl_graph_apply: null pointer");
    AGRL_CHECK_ARG(B > 0 && V > 0 && C > 0, "agrl_graph_apply: bad shape");
    AGRL_CHECK_ARG(out_dtype == AGRL_F32 || out_dtype == AGRL_LP16 || out_dtype == AGRL_F32H3P, "agrl_graph_apply: out dtype must be fp32, the 16-bit type or AGRL_F32H3P");
    const bool lp = out_dtype == AGRL_LP16, pre = out_dtype == AGRL_F32H3P;
    const bool aligned = ((((uintptr_t)f | (uintptr_t)out | (uintptr_t)G) & 15) == 0);
    AGRL_CHECK_ARG(V <= 64 && (V % 4) == 0 && (C % 128) == 0 && aligned,
                   "agrl_graph_apply: the streaming form needs V <= 64, V %% 4 == 0, C %% 128 == 0, 16-byte aligned operands (V=%d C=%d); "
                   "use agrl_graph_propagate with h = f and a unit BatchNorm otherwise", V, C);
    const int V4 = (V + 3) & ~3;
    const int nvf = (V + 15) / 16;
    const int nwv = (C % 256) == 0 ? 4 : 2;
    const size_t lds_s = (size_t)16 * nvf * V4 * sizeof(float) + 2048;
#define LAUNCH_GA(NT_)                                                                                                          \
    case NT_:                                                                                                                   \
        if (nwv == 4) {                                                                                                         \
            if (lp) hipLaunchKernelGGL((graph_apply_stream_kernel<NT_, 4, 1>), dim3(B, C / 256), dim3(256), lds_s, (hipStream_t)stream, f, G, nullptr, (lp16_t*)out, C); \
            else if (pre) hipLaunchKernelGGL((graph_apply_stream_kernel<NT_, 4, 2>), dim3(B, C / 256), dim3(256), lds_s, (hipStream_t)stream, f, G, (float*)out, nullptr, C); \
            else hipLaunchKernelGGL((graph_apply_stream_kernel<NT_, 4, 0>), dim3(B, C / 256), dim3(256), lds_s, (hipStream_t)stream, f, G, (float*)out, nullptr, C);   \
        } else {                                                                                                                \
            if (lp) hipLaunchKernelGGL((graph_apply_stream_kernel<NT_, 2, 1>), dim3(B, C / 128), dim3(128), lds_s, (hipStream_t)stream, f, G, nullptr, (lp16_t*)out, C); \
            else if (pre) hipLaunchKernelGGL((graph_apply_stream_kernel<NT_, 2, 2>), dim3(B, C / 128), dim3(128), lds_s, (hipStream_t)stream, f, G, (float*)out, nullptr, C); \
            else hipLaunchKernelGGL((graph_apply_stream_kernel<NT_, 2, 0>), dim3(B, C / 128), dim3(128), lds_s, (hipStream_t)stream, f, G, (float*)out, nullptr, C);   \
        }                                                                                                                       \
        break
    switch (V4 >> 2) {
        LAUNCH_GA(1); LAUNCH_GA(2); LAUNCH_GA(3); LAUNCH_GA(4); LAUNCH_GA(5); LAUNCH_GA(6); LAUNCH_GA(7); LAUNCH_GA(8);
        LAUNCH_GA(9); LAUNCH_GA(10); LAUNCH_GA(11); LAUNCH_GA(12); LAUNCH_GA(13); LAUNCH_GA(14); LAUNCH_GA(15); LAUNCH_GA(16);
    }
#undef LAUNCH_GA
    AGRL_CHECK_LAUNCH("agrl_graph_apply");
    return 0;
}

extern "C" int agrl_graph_tracklet_operand(const float* f, const void* adj, int adj_packed, float* G_out, void* out, int out_dtype, int B, int V,
                                           int C, int use_pose, int learn_graph, int mask_diag, agrl_stream_t stream) {
    const float* adj_f = adj_packed ? nullptr : (const float*)adj;
    const uint32_t* adj_b = adj_packed ? (const uint32_t*)adj : nullptr;
    AGRL_CHECK_ARG(f && out && (use_pose || learn_graph) && (!use_pose || adj), "agrl_graph_tracklet_operand: bad arguments");
    AGRL_CHECK_ARG(out_dtype == AGRL_F32 || out_dtype == AGRL_LP16, "agrl_graph_tracklet_operand: out dtype must be fp32 or bf16");
    AGRL_CHECK_ARG(B > 0 && V > 0 && V <= 64 && (V % 4) == 0 && C >= 512 && (C % 512) == 0 && ((((uintptr_t)f | (uintptr_t)out) & 15) == 0),
                   "agrl_graph_tracklet_operand: built for V <= 64, V %% 4 == 0, C %% 512 == 0, 16-byte aligned f / out (V=%d C=%d)", V, C);
    const int nvf = (V + 15) / 16, VP = nvf * 16, npair = nvf * (nvf + 1) / 2;
    const size_t lds = ((size_t)GT_WAVES * npair * 256 + (size_t)VP * (VP + 1) + (size_t)VP * V) * sizeof(float);
    const bool lp = out_dtype == AGRL_LP16;
#define LAUNCH_GT(NT_)                                                                                                          \
    case NT_: {                                                                                                                 \
        if (lds > 64 * 1024) {                                                                                                  \
            if (lp) (void)hipFuncSetAttribute((const void*)graph_tracklet_kernel<NT_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            else (void)hipFuncSetAttribute((const void*)graph_tracklet_kernel<NT_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
        }                                                                                                                       \
        if (lp) hipLaunchKernelGGL((graph_tracklet_kernel<NT_, true>), dim3(B), dim3(64 * GT_WAVES), lds, (hipStream_t)stream, f, adj_f, adj_b, G_out, nullptr, (lp16_t*)out, C, use_pose, learn_graph, mask_diag); \
        else hipLaunchKernelGGL((graph_tracklet_kernel<NT_, false>), dim3(B), dim3(64 * GT_WAVES), lds, (hipStream_t)stream, f, adj_f, adj_b, G_out, (float*)out, nullptr, C, use_pose, learn_graph, mask_diag);   \
    } break
    (void)hipGetLastError();
    switch (V / 4) {
        LAUNCH_GT(1); LAUNCH_GT(2); LAUNCH_GT(3); LAUNCH_GT(4); LAUNCH_GT(5); LAUNCH_GT(6); LAUNCH_GT(7); LAUNCH_GT(8);
        LAUNCH_GT(9); LAUNCH_GT(10); LAUNCH_GT(11); LAUNCH_GT(12); LAUNCH_GT(13); LAUNCH_GT(14); LAUNCH_GT(15); LAUNCH_GT(16);
    }
#undef LAUNCH_GT
    AGRL_CHECK_LAUNCH("agrl_graph_tracklet_operand");
    return 0;
}

extern "C" int agrl_graph_propagate(const float* f, const float* h, const float* G, const float* bn_scale,
                                    const float* bn_shift, float keep, float gamma, float slope, float* out, void* out_lp,
                                    int B, int V, int C, agrl_stream_t stream) {
    AGRL_CHECK_ARG(f && h && G && bn_scale && bn_shift && out, "agrl_graph_propagate: null pointer");
    AGRL_CHECK_ARG(B > 0 && V > 0 && C > 0, "agrl_graph_propagate: bad shape");
    // keep = the coefficient of f: (1 - gamma) for vmgn / gsta, evaluated by the host in double like the reference's Python
    // float and rounded once; 1 for ganet's ``input + gamma * h'``
    const float omg_m = keep;
    const bool aligned = ((((uintptr_t)f | (uintptr_t)h | (uintptr_t)out | (uintptr_t)out_lp | (uintptr_t)bn_scale |
                            (uintptr_t)bn_shift) & 15) == 0);
    if (V <= 64 && (V % 4) == 0 && (C % 128) == 0 && aligned && (((uintptr_t)G) & 15) == 0) {
        const int V4 = (V + 3) & ~3;
        const int nvf = (V + 15) / 16;
        const int nwv = (C % 256) == 0 ? 4 : 2;  // 4-wave workgroups: one wave per SIMD of a CU by construction
        const size_t lds_s = (size_t)16 * nvf * V4 * sizeof(float) + 2048;  // padded fragment rows + DMA piece rounding + spare
#define LAUNCH_PS(NT_)                                                                                                \
    case NT_:                                                                                                         \
        if (nwv == 4)                                                                                                 \
            hipLaunchKernelGGL((graph_propagate_stream_kernel<NT_, 4>), dim3(B, C / 256), dim3(256), lds_s,           \
                               (hipStream_t)stream, f, h, G, bn_scale, bn_shift, omg_m, gamma, slope, out,            \
                               (lp16_t*)out_lp, C);                                                                   \
        else                                                                                                          \
            hipLaunchKernelGGL((graph_propagate_stream_kernel<NT_, 2>), dim3(B, C / 128), dim3(128), lds_s,           \
                               (hipStream_t)stream, f, h, G, bn_scale, bn_shift, omg_m, gamma, slope, out,            \
                               (lp16_t*)out_lp, C);                                                                   \
        break
        switch (V4 >> 2) {
            LAUNCH_PS(1); LAUNCH_PS(2); LAUNCH_PS(3); LAUNCH_PS(4); LAUNCH_PS(5); LAUNCH_PS(6); LAUNCH_PS(7); LAUNCH_PS(8);
            LAUNCH_PS(9); LAUNCH_PS(10); LAUNCH_PS(11); LAUNCH_PS(12); LAUNCH_PS(13); LAUNCH_PS(14); LAUNCH_PS(15); LAUNCH_PS(16);
        }
#undef LAUNCH_PS
        AGRL_CHECK_LAUNCH("agrl_graph_propagate");
        return 0;
    }
    if (V <= 128 && (C % 128) == 0) {
        const int V4 = (V + 3) & ~3;
        const int hrows = (V4 + 1) & ~1;
        const int nvf = V <= 64 ? 4 : 8;
        const size_t lds_m = (size_t)hrows * 512 + (size_t)V4 * nvf * 16 * 4;
        const dim3 grid_m(B, C / 128);
        if (nvf == 4) {
            hipLaunchKernelGGL(graph_propagate_mfma_kernel<4>, grid_m, dim3(256), lds_m, (hipStream_t)stream, f, h, G,
                               bn_scale, bn_shift, omg_m, gamma, slope, out, (lp16_t*)out_lp, V, C);
        } else {
            if (lds_m > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)graph_propagate_mfma_kernel<8>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                AGRL_CHECK_ARG(e == hipSuccess, "agrl_graph_propagate: cannot raise dynamic LDS: %s", hipGetErrorString(e));
            }
            hipLaunchKernelGGL(graph_propagate_mfma_kernel<8>, grid_m, dim3(256), lds_m, (hipStream_t)stream, f, h, G,
                               bn_scale, bn_shift, omg_m, gamma, slope, out, (lp16_t*)out_lp, V, C);
        }
        AGRL_CHECK_LAUNCH("agrl_graph_propagate");
        return 0;
    }
    const int Vp = (V + PROP_RB - 1) & ~(PROP_RB - 1);
    const bool fixed = false;  // register-resident h (VT > 0) spills: hipcc hoists every LDS graph read; keep h in LDS
    const size_t lds = ((size_t)V * Vp + (fixed ? 0 : (size_t)V * PROP_THREADS)) * sizeof(float);
    if (lds > 160 * 1024) {   // graph + h slab no longer fit the LDS (V > ~125): 16 graph rows per workgroup, h streamed from L2
        AGRL_CHECK_ARG((size_t)V * PROPT_ROWS * 4 <= 64 * 1024, "agrl_graph_propagate: V=%d too large", V);
        hipLaunchKernelGGL(graph_propagate_tiled_kernel, dim3(B, cdiv(C, PROP_THREADS), cdiv(V, PROPT_ROWS)), dim3(PROP_THREADS),
                           (size_t)V * PROPT_ROWS * 4, (hipStream_t)stream, f, h, G, bn_scale, bn_shift, keep, gamma, slope, out, (lp16_t*)out_lp, V, C);
        AGRL_CHECK_LAUNCH("agrl_graph_propagate");
        return 0;
    }
    const float omg = keep;
    const dim3 grid(B, cdiv(C, PROP_THREADS));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_PROP(VT)                                                                                              \
    do {                                                                                                             \
        if (lds > 64 * 1024) {                                                                                       \
            hipError_t e = hipFuncSetAttribute((const void*)graph_propagate_kernel<VT>,                              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
            AGRL_CHECK_ARG(e == hipSuccess, "agrl_graph_propagate: cannot raise dynamic LDS: %s", hipGetErrorString(e)); \
        }                                                                                                            \
        hipLaunchKernelGGL(graph_propagate_kernel<VT>, grid, dim3(PROP_THREADS), lds, st, f, h, G, bn_scale, bn_shift, \
                           omg, gamma, slope, out, (lp16_t*)out_lp, V, C);                                           \
    } while (0)
    LAUNCH_PROP(0);
#undef LAUNCH_PROP
    AGRL_CHECK_LAUNCH("agrl_graph_propagate");
    return 0;
}


extern "C" int agrl_pose_adjacency(const float* poses, const unsigned char* detected, float* adj, int B, int S, int num_split,
                                   int pyramid_part, float height, float threshold, agrl_stream_t stream) {
    AGRL_CHECK_ARG(poses && detected && adj, "agrl_pose_adjacency: null pointer");
    AGRL_CHECK_ARG(B > 0 && S > 0 && S <= 64, "agrl_pose_adjacency: 1 <= S <= 64 frames (got %d)", S);
    AGRL_CHECK_ARG(num_split >= 1 && num_split <= 16 && (num_split & (num_split - 1)) == 0,
                   "agrl_pose_adjacency: num_split must be a power of two <= 16 (got %d)", num_split);
    int levels = 0;
    while ((1 << levels) < num_split) ++levels;
    const int P = pyramid_part ? 2 * num_split - 1 : num_split;
    hipLaunchKernelGGL(pose_adjacency_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, poses, detected, adj, (uint32_t*)nullptr, S, num_split,
                       levels, pyramid_part ? 1 : 0, P, (double)height, threshold);
    AGRL_CHECK_LAUNCH("agrl_pose_adjacency");
    return 0;
}

extern "C" int agrl_pose_adjacency_bits(const float* poses, const unsigned char* detected, uint32_t* adj_bits, int B, int S, int num_split,
                                        int pyramid_part, float height, float threshold, agrl_stream_t stream) {
    AGRL_CHECK_ARG(poses && detected && adj_bits, "agrl_pose_adjacency_bits: null pointer");
    AGRL_CHECK_ARG(B > 0 && S > 0 && S <= 64, "agrl_pose_adjacency_bits: 1 <= S <= 64 frames (got %d)", S);
    AGRL_CHECK_ARG(num_split >= 1 && num_split <= 16 && (num_split & (num_split - 1)) == 0,
                   "agrl_pose_adjacency_bits: num_split must be a power of two <= 16 (got %d)", num_split);
    int levels = 0;
    while ((1 << levels) < num_split) ++levels;
    const int P = pyramid_part ? 2 * num_split - 1 : num_split;
    hipLaunchKernelGGL(pose_adjacency_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, poses, detected, (float*)nullptr, adj_bits, S, num_split,
                       levels, pyramid_part ? 1 : 0, P, (double)height, threshold);
    AGRL_CHECK_LAUNCH("agrl_pose_adjacency_bits");
    return 0;
}

extern "C" int agrl_adjacency_pack(const float* adj, uint32_t* adj_bits, int B, int V, agrl_stream_t stream) {
    AGRL_CHECK_ARG(adj && adj_bits && B > 0 && V > 0, "agrl_adjacency_pack: bad arguments");
    const size_t total = (size_t)B * V * ((V + 31) >> 5);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(adjacency_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, adj, adj_bits, V, total);
    AGRL_CHECK_LAUNCH("agrl_adjacency_pack");
    return 0;
}
