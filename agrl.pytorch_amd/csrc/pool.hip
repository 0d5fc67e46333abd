// Pooling kernels of the per-tracklet tail (all HBM-bound, one pass over the layer4 maps):
//   agrl_part_pool        : vmgn.py:298-300 (per-frame part of the global pool) + :304-308 (part pooling)
//   agrl_row_sqnorm       : vmgn.py:276 (feat.norm over channels), distance.py:70-71 (row norms)
//   agrl_attn_pool_bnneck : vmgn.py:270-278, :313-321 (attention-weighted temporal pooling, BNNeck, cat)
//   agrl_row_l2_normalize : distance.py:86-87 (F.normalize p=2) / dtype conversion of embeddings
#include "agrl_common.h"

namespace {

constexpr int MAX_PARTS = 16;

struct PartBins {
    int nparts;
    int start[MAX_PARTS];
    int end[MAX_PARTS];
};

template <typename T>
__device__ inline void load_vec(const T* p, float v[DT<T>::epc]);
template <>
__device__ inline void load_vec<float>(const float* p, float v[4]) {
    const float4 f = *reinterpret_cast<const float4*>(p);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
template <>
__device__ inline void load_vec<lp16_t>(const lp16_t* p, float v[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unpack_lp16x2(w[i], v[2 * i], v[2 * i + 1]);
    }
}

// grid = (frames, 2): y == 0 -> global sums of x4_1, y == 1 -> part means of x4_2.
// thread -> one 16-byte channel vector; a wavefront reads 1 KiB contiguous per pixel.
template <typename T, int NP>
__global__ __launch_bounds__(256) void part_pool_kernel(const T* __restrict__ x41, const T* __restrict__ x42,
                                                        float* __restrict__ gsum, float* __restrict__ nodes,
                                                        lp16_t* __restrict__ nodes_lp, int h, int w, int C,
                                                        PartBins bins) {
    constexpr int VEC = DT<T>::epc;
    const int frame = blockIdx.x;
    const int which = blockIdx.y;
    const int P = bins.nparts;
    for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
        if (which == 0) {
            const T* src = x41 + (size_t)frame * h * w * C + c;
            float acc[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
            const int npix = h * w;
#pragma unroll 8
            for (int pix = 0; pix < npix; ++pix) {
                float v[VEC];
                load_vec<T>(src + (size_t)pix * C, v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += v[j];
            }
            float* dst = gsum + (size_t)frame * C + c;
#pragma unroll
            for (int j = 0; j < VEC; ++j) dst[j] = acc[j];
        } else {
            const T* src = x42 + (size_t)frame * h * w * C + c;
            float acc[NP][VEC];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[q][j] = 0.f;
            for (int y = 0; y < h; ++y) {
                float rs[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) rs[j] = 0.f;
#pragma unroll 8
                for (int xx = 0; xx < w; ++xx) {
                    float v[VEC];
                    load_vec<T>(src + (size_t)(y * w + xx) * C, v);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) rs[j] += v[j];
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const bool in = q < P && y >= bins.start[q] && y < bins.end[q];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[q][j] += in ? rs[j] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (q < P) {
                    const float inv = 1.f / (float)((bins.end[q] - bins.start[q]) * w);
                    float* dst = nodes + ((size_t)frame * P + q) * C + c;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const float m = acc[q][j] * inv;
                        dst[j] = m;
                        if (nodes_lp) nodes_lp[((size_t)frame * P + q) * C + c + j] = f32_to_lp16(m);
                    }
                }
            }
        }
    }
}

// one wavefront per row
template <typename T>
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const T* __restrict__ x, float* __restrict__ sqn, int R,
                                                         int C) {
    constexpr int VEC = DT<T>::epc;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const T* src = x + (size_t)row * C;
    float s = 0.f;
    if ((C % VEC) == 0) {
        // eight 16-byte loads in flight per lane, then the fmas in the same order as a plain loop
        for (int c0 = lane * VEC; c0 < C; c0 += 8 * 64 * VEC) {
            float v[8][VEC];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 + i * 64 * VEC;
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[i][j] = 0.f;
                if (c < C) load_vec<T>(src + c, v[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i * 64 * VEC < C) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) s = fmaf(v[i][j], v[i][j], s);
                }
        }
    } else {
        for (int c = lane; c < C; c += 64) {
            const float v = DT<T>::ld(src + c);
            s = fmaf(v, v, s);
        }
    }
    s = wave_sum(s);
    if (lane == 0) sqn[row] = s;
}

// y = x / max(||x||, 1e-12) (or plain conversion); one 256-thread workgroup per row, float4 loads
template <typename TOUT>
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ x, TOUT* __restrict__ y, int R,
                                                            int C, int ldy, int normalize) {
    __shared__ float s_part[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = x + (size_t)row * C;
    const bool vec = (C & 3) == 0 && ((((uintptr_t)src) & 15) == 0);
    float den = 1.f;
    if (normalize) {
        float s = 0.f;
        if (vec) {
            for (int c = tid * 4; c < C; c += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(src + c);
                s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
            }
        } else {
            for (int c = tid; c < C; c += 256) s = fmaf(src[c], src[c], s);
        }
        s = wave_sum(s);
        if (lane == 0) s_part[wave] = s;
        __syncthreads();
        den = fmaxf(sqrtf((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])), 1e-12f);
    }
    TOUT* dst = y + (size_t)row * ldy;
    for (int c = tid; c < ldy; c += 256) {
        const float v = c < C ? (normalize ? src[c] / den : src[c]) : 0.f;  // zero padding up to ldy
        DT<TOUT>::st(dst + c, v);
    }
}

// grid = (B, C/256): thread -> one channel of one tracklet.
__global__ __launch_bounds__(256) void attn_pool_bnneck_kernel(
    const float* __restrict__ nodes, const float* __restrict__ sqn, const float* __restrict__ gsum,
    const float* __restrict__ g_scale, const float* __restrict__ g_shift, const float* __restrict__ a_scale,
    const float* __restrict__ a_shift, float* __restrict__ out, float* __restrict__ g_f, float* __restrict__ att_f,
    int S, int P, int C, float inv_ghw) {
    extern __shared__ __attribute__((aligned(16))) float s_att[];  // S*P attention weights
    const int b = blockIdx.x;
    const int V = S * P;
    // a[s,p] = ||f[s,p]|| / max(sum_s ||f[s,p]||, 1e-12)   (F.normalize p=1 over the frame axis)
    // every node's norm by its own thread (one round trip), then P threads normalise over the frame axis out of LDS -- the
    // same sums in the same order as a per-part loop over global memory, without its chain of 2 S dependent loads
    for (int t = threadIdx.x; t < V; t += blockDim.x) s_att[t] = sqrtf(sqn[(size_t)b * V + t]);
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += blockDim.x) {
        float tot = 0.f;
        for (int s = 0; s < S; ++s) tot += s_att[s * P + q];
        const float den = fmaxf(tot, 1e-12f);
        for (int s = 0; s < S; ++s) s_att[s * P + q] = s_att[s * P + q] / den;
    }
    __syncthreads();
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // attention branch: mean over parts of the attention-weighted sum over frames
    float att = 0.f;
    // eight loads in flight per part instead of a load -> fma chain over the S x P nodes (the kernel was a string of HBM
    // round trips); the arithmetic keeps its order: fuse over s ascending, then over the parts
    for (int q = 0; q < P; ++q) {
        float fuse = 0.f;
        for (int s0 = 0; s0 < S; s0 += 8) {
            float nv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) nv[i] = s0 + i < S ? nodes[((size_t)b * V + (s0 + i) * P + q) * C + c] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (s0 + i < S) fuse = fmaf(s_att[(s0 + i) * P + q], nv[i], fuse);
        }
        att += fuse;
    }
    att /= (float)P;
    // global branch: mean over (S, h, w)
    float g = 0.f;
    for (int s0 = 0; s0 < S; s0 += 8) {
        float gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) gv[i] = s0 + i < S ? gsum[((size_t)b * S + s0 + i) * C + c] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (s0 + i < S) g += gv[i];
    }
    g *= inv_ghw;
    if (g_f) g_f[(size_t)b * C + c] = g;
    if (att_f) att_f[(size_t)b * C + c] = att;
    out[(size_t)b * 2 * C + c] = fmaf(g, g_scale[c], g_shift[c]);
    out[(size_t)b * 2 * C + C + c] = fmaf(att, a_scale[c], a_shift[c]);
}

// The whole per-tracklet tail in ONE launch (round 5: the review's "two extra launches in front of the distance matrix"): node norms
// (row_sqnorm_kernel's job), attention pooling + BNNeck + cat (attn_pool_bnneck_kernel's), and the query operand of the distance
// matrix -- the embedding row's squared norm and its L2-normalised copy in the distance matrix's operand type (row_sqnorm_kernel /
// row_normalize_kernel on the (B, 2C) output). One 1024-thread workgroup per tracklet; every sum runs in the order of the kernel
// it replaces (a wave per node / per row with the same eight loads in flight and the same shuffle tree; a thread per channel over
// frames then parts; 256 threads x float4 strides and the four-partial tree for the row norm), so all outputs are BIT-IDENTICAL to
// the three separate launches. vmgn.py:270-278, :313-321; distance.py:70-71, :86-87.
__global__ __launch_bounds__(1024) void attn_tail_kernel(
    const float* __restrict__ nodes, const float* __restrict__ gsum, const float* __restrict__ g_scale,
    const float* __restrict__ g_shift, const float* __restrict__ a_scale, const float* __restrict__ a_shift,
    float* __restrict__ out, float* __restrict__ g_f, float* __restrict__ att_f, float* __restrict__ node_sqn,
    float* __restrict__ out_sqn, lp16_t* __restrict__ q_lp, float* __restrict__ q_f32, int S, int P, int C, float inv_ghw) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];  // [V] attention weights | [2 C] the output row | [4] partials
    const int b = blockIdx.x;
    const int V = S * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* s_att = s_mem;
    float* s_row = s_mem + ((V + 3) & ~3);
    float* s_part = s_row + 2 * C;
    // ---- node norms: one wavefront per node, row_sqnorm_kernel<float>'s loads and order
    for (int v = wave; v < V; v += 16) {
        const float* src = nodes + ((size_t)b * V + v) * C;
        float s = 0.f;
        for (int c0 = lane * 4; c0 < C; c0 += 8 * 64 * 4) {
            float x[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 + i * 64 * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) x[i][j] = 0.f;
                if (c < C) load_vec<float>(src + c, x[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i * 64 * 4 < C) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = fmaf(x[i][j], x[i][j], s);
                }
        }
        s = wave_sum(s);
        if (lane == 0) {
            s_att[v] = sqrtf(s);
            if (node_sqn) node_sqn[(size_t)b * V + v] = s;
        }
    }
    __syncthreads();
    // a[s,p] = ||f[s,p]|| / max(sum_s ||f[s,p]||, 1e-12)   (F.normalize p=1 over the frame axis)
    for (int q = tid; q < P; q += blockDim.x) {
        float tot = 0.f;
        for (int s = 0; s < S; ++s) tot += s_att[s * P + q];
        const float den = fmaxf(tot, 1e-12f);
        for (int s = 0; s < S; ++s) s_att[s * P + q] = s_att[s * P + q] / den;
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        // attention branch: mean over parts of the attention-weighted sum over frames (attn_pool_bnneck_kernel's order)
        float att = 0.f;
        for (int q = 0; q < P; ++q) {
            float fuse = 0.f;
            for (int s0 = 0; s0 < S; s0 += 8) {
                float nv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) nv[i] = s0 + i < S ? nodes[((size_t)b * V + (s0 + i) * P + q) * C + c] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (s0 + i < S) fuse = fmaf(s_att[(s0 + i) * P + q], nv[i], fuse);
            }
            att += fuse;
        }
        att /= (float)P;
        float g = 0.f;
        for (int s0 = 0; s0 < S; s0 += 8) {
            float gv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) gv[i] = s0 + i < S ? gsum[((size_t)b * S + s0 + i) * C + c] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (s0 + i < S) g += gv[i];
        }
        g *= inv_ghw;
        if (g_f) g_f[(size_t)b * C + c] = g;
        if (att_f) att_f[(size_t)b * C + c] = att;
        const float og = fmaf(g, g_scale[c], g_shift[c]), oa = fmaf(att, a_scale[c], a_shift[c]);
        out[(size_t)b * 2 * C + c] = og;
        out[(size_t)b * 2 * C + C + c] = oa;
        s_row[c] = og;
        s_row[C + c] = oa;
    }
    __syncthreads();
    const int D = 2 * C;
    if (out_sqn && wave == 4) {  // the row's squared norm as row_sqnorm_kernel<float> forms it (one wavefront)
        float s = 0.f;
        for (int c0 = lane * 4; c0 < D; c0 += 8 * 64 * 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 + i * 64 * 4;
                if (c < D) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = fmaf(s_row[c + j], s_row[c + j], s);
                }
            }
        }
        s = wave_sum(s);
        if (lane == 0) out_sqn[b] = s;
    }
    if (q_lp || q_f32) {  // x / max(||x||, 1e-12): row_normalize_kernel's 256 threads x float4 strides, four wave partials
        if (tid < 256) {
            float s = 0.f;
            for (int c = tid * 4; c < D; c += 1024) {
                s = fmaf(s_row[c], s_row[c], s); s = fmaf(s_row[c + 1], s_row[c + 1], s);
                s = fmaf(s_row[c + 2], s_row[c + 2], s); s = fmaf(s_row[c + 3], s_row[c + 3], s);
            }
            s = wave_sum(s);
            if (lane == 0) s_part[wave] = s;
        }
        __syncthreads();
        const float den = fmaxf(sqrtf((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])), 1e-12f);
        for (int c = tid; c < D; c += blockDim.x) {
            const float v = s_row[c] / den;
            if (q_lp) q_lp[(size_t)b * D + c] = f32_to_lp16(v);
            if (q_f32) q_f32[(size_t)b * D + c] = v;
        }
    }
}

// out[t][c] = mean / max over i < n of feats[t*n + i][c]; thread -> one channel of one tracklet, clips in ascending order
__global__ __launch_bounds__(256) void clip_pool_kernel(const float* __restrict__ feats, float* __restrict__ out, int n, int D,
                                                        int mode) {
    const int t = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= D) return;
    const float* src = feats + (size_t)t * n * D + c;
    float acc = src[0];
    for (int i0 = 1; i0 < n; i0 += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = i0 + i < n ? src[(size_t)(i0 + i) * D] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i0 + i < n) acc = mode ? __builtin_elementwise_maximum(acc, v[i]) : acc + v[i];  // IEEE-754-2019 maximum: a NaN clip stays a NaN (fmaxf drops it)
    }
    out[(size_t)t * D + c] = mode ? acc : acc / (float)n;
}

}  // namespace

extern "C" int agrl_clip_pool(const float* feats, float* out, int T, int n, int D, int mode, agrl_stream_t stream) {
    AGRL_CHECK_ARG(feats && out && T > 0 && n > 0 && D > 0, "agrl_clip_pool: bad arguments");
    AGRL_CHECK_ARG(mode == 0 || mode == 1, "agrl_clip_pool: mode must be 0 (mean) or 1 (max), got %d", mode);
    hipLaunchKernelGGL(clip_pool_kernel, dim3(T, cdiv(D, 256)), dim3(256), 0, (hipStream_t)stream, feats, out, n, D, mode);
    AGRL_CHECK_LAUNCH("agrl_clip_pool");
    return 0;
}

extern "C" int agrl_part_pool(const void* x4_1, const void* x4_2, float* gsum, float* nodes, void* nodes_lp, int F,
                              int h, int w, int C, const int* splits, int n_splits, int dtype,
                              agrl_stream_t stream) {
    AGRL_CHECK_ARG(x4_1 && x4_2 && gsum && nodes && splits, "agrl_part_pool: null pointer");
    AGRL_CHECK_ARG(F > 0 && h > 0 && w > 0 && C > 0 && n_splits > 0, "agrl_part_pool: bad shape");
    AGRL_CHECK_ARG(dtype == AGRL_F32 || dtype == AGRL_LP16, "agrl_part_pool: bad dtype %d", dtype);
    PartBins bins;
    int P = 0;
    for (int i = 0; i < n_splits; ++i) {
        const int n = splits[i];
        AGRL_CHECK_ARG(n > 0 && P + n <= MAX_PARTS, "agrl_part_pool: at most %d parts supported", MAX_PARTS);
        for (int j = 0; j < n; ++j) {  // AdaptiveAvgPool2d bins: [floor(j*h/n), ceil((j+1)*h/n))
            bins.start[P] = (j * h) / n;
            bins.end[P] = ((j + 1) * h + n - 1) / n;
            ++P;
        }
    }
    bins.nparts = P;
    for (int i = P; i < MAX_PARTS; ++i) bins.start[i] = bins.end[i] = 0;
    const int vec = dtype == AGRL_F32 ? 4 : 8;
    AGRL_CHECK_ARG(C % vec == 0, "agrl_part_pool: C=%d must be a multiple of %d", C, vec);
    const int threads = 256;
    dim3 grid(F, 2);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_PP(T, NP)                                                                                         \
    hipLaunchKernelGGL((part_pool_kernel<T, NP>), grid, dim3(threads), 0, st, (const T*)x4_1, (const T*)x4_2, gsum, \
                       nodes, (lp16_t*)nodes_lp, h, w, C, bins)
    if (dtype == AGRL_F32) {
        if (P <= 8) LAUNCH_PP(float, 8); else LAUNCH_PP(float, 16);
    } else {
        if (P <= 8) LAUNCH_PP(lp16_t, 8); else LAUNCH_PP(lp16_t, 16);
    }
#undef LAUNCH_PP
    AGRL_CHECK_LAUNCH("agrl_part_pool");
    return 0;
}

extern "C" int agrl_row_sqnorm(const void* x, float* sqn, int R, int C, int dtype, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && sqn && R > 0 && C > 0, "agrl_row_sqnorm: bad arguments");
    AGRL_CHECK_ARG(dtype == AGRL_F32 || dtype == AGRL_LP16, "agrl_row_sqnorm: bad dtype %d", dtype);
    if (dtype == AGRL_F32)
        hipLaunchKernelGGL(row_sqnorm_kernel<float>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, sqn, R, C);
    else
        hipLaunchKernelGGL(row_sqnorm_kernel<lp16_t>, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const lp16_t*)x, sqn, R, C);
    AGRL_CHECK_LAUNCH("agrl_row_sqnorm");
    return 0;
}

extern "C" int agrl_row_l2_normalize(const float* x, void* y, int R, int C, int ldy, int normalize, int out_dtype,
                                     agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && y && R > 0 && C > 0 && ldy >= C, "agrl_row_l2_normalize: bad arguments");
    AGRL_CHECK_ARG(out_dtype == AGRL_F32 || out_dtype == AGRL_LP16, "agrl_row_l2_normalize: bad dtype %d", out_dtype);
    if (out_dtype == AGRL_F32)
        hipLaunchKernelGGL(row_normalize_kernel<float>, dim3(R), dim3(256), 0, (hipStream_t)stream, x,
                           (float*)y, R, C, ldy, normalize);
    else
        hipLaunchKernelGGL(row_normalize_kernel<lp16_t>, dim3(R), dim3(256), 0, (hipStream_t)stream, x,
                           (lp16_t*)y, R, C, ldy, normalize);
    AGRL_CHECK_LAUNCH("agrl_row_l2_normalize");
    return 0;
}

extern "C" int agrl_attn_pool_bnneck(const float* nodes, const float* sqn, const float* gsum, const float* g_scale,
                                     const float* g_shift, const float* a_scale, const float* a_shift, float* out,
                                     float* g_f, float* att_f, int B, int S, int P, int C, int hw,
                                     agrl_stream_t stream) {
    AGRL_CHECK_ARG(nodes && sqn && gsum && g_scale && g_shift && a_scale && a_shift && out,
                   "agrl_attn_pool_bnneck: null pointer");
    AGRL_CHECK_ARG(B > 0 && S > 0 && P > 0 && C > 0 && hw > 0, "agrl_attn_pool_bnneck: bad shape");
    const size_t lds = (size_t)S * P * sizeof(float);
    AGRL_CHECK_ARG(lds <= 64 * 1024, "agrl_attn_pool_bnneck: S*P too large");
    dim3 grid(B, cdiv(C, 256));
    hipLaunchKernelGGL(attn_pool_bnneck_kernel, grid, dim3(256), lds, (hipStream_t)stream, nodes, sqn, gsum, g_scale,
                       g_shift, a_scale, a_shift, out, g_f, att_f, S, P, C, 1.f / ((float)S * (float)hw));
    AGRL_CHECK_LAUNCH("agrl_attn_pool_bnneck");
    return 0;
}

extern "C" int agrl_attn_tail(const float* nodes, const float* gsum, const float* g_scale, const float* g_shift, const float* a_scale,
                              const float* a_shift, float* out, float* g_f, float* att_f, float* node_sqn, float* out_sqn,
                              void* q_lp, float* q_f32, int B, int S, int P, int C, int hw, agrl_stream_t stream) {
    AGRL_CHECK_ARG(nodes && gsum && g_scale && g_shift && a_scale && a_shift && out, "agrl_attn_tail: null pointer");
    AGRL_CHECK_ARG(B > 0 && S > 0 && P > 0 && C > 0 && hw > 0 && C % 4 == 0, "agrl_attn_tail: bad shape (C must be a multiple of 4)");
    AGRL_CHECK_ARG((((uintptr_t)nodes | (uintptr_t)out) & 15) == 0, "agrl_attn_tail: nodes / out must be 16-byte aligned");
    const size_t lds = ((size_t)((S * P + 3) & ~3) + 2 * (size_t)C + 4) * sizeof(float);
    AGRL_CHECK_ARG(lds <= 64 * 1024, "agrl_attn_tail: S * P + 2 C floats must fit 64 KB of LDS");
    hipLaunchKernelGGL(attn_tail_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, nodes, gsum, g_scale, g_shift, a_scale, a_shift,
                       out, g_f, att_f, node_sqn, out_sqn, reinterpret_cast<lp16_t*>(q_lp), q_f32, S, P, C, 1.f / ((float)S * (float)hw));
    AGRL_CHECK_LAUNCH("agrl_attn_tail");
    return 0;
}
