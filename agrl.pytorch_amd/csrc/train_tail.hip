// Train-step kernels of the model's TAIL (BASELINE config 4; GSTA.forward under model.train(), vmgn.py:296-357, and the losses
// of train_vidreid_xent_htri.py:401-408): the backward passes that have no forward twin among the eval kernels, and the fused
// label-smoothed cross entropy. Forward passes reuse the eval entry points (agrl_part_pool, agrl_graph_gram / _finalize /
// _propagate, agrl_row_sqnorm, agrl_attn_pool_bnneck); Linear layers run as 1x1 convs on the conv nodes of csrc/train.hip.
//
//   agrl_axpby                      out = a x + b y                                       (residual mix, vmgn.py:172, and its backward)
//   agrl_part_pool_backward         gradients of the global / part average pooling        (vmgn.py:298-308)
//   agrl_attn_pool_backward         gradient of the attention temporal pooling            (vmgn.py:270-278, :313-317)
//   agrl_graph_matrix_backward      d loss / d G  ->  the (V x V) matrix M with d loss / d f = M f   (vmgn.py:114-120, :155-166)
//   agrl_xent_label_smooth          loss value + d loss / d logits in one call            (losses/cross_entropy_loss.py:26-37)
// All deterministic (no atomics), fp32.
#include "agrl_common.h"

namespace {

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b,
                                                    float* __restrict__ out, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256)
        out[e] = y ? fmaf(a, x[e], b * y[e]) : a * x[e];
}

struct PoolBins {
    int nparts;
    int start[16], end[16];
};

// dx1[f][pix][c] = dg[f / S][c] * inv_global ; dx2[f][pix][c] = sum over the parts whose row band holds pix of dnodes[f][part][c] / (rows * w)
__global__ __launch_bounds__(256) void part_pool_backward_kernel(const float* __restrict__ dg, const float* __restrict__ dnodes,
                                                                 float* __restrict__ dx1, float* __restrict__ dx2, int S, int h, int w,
                                                                 int C, float inv_global, PoolBins bins, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        size_t q = e / C;
        const int pix = (int)(q % (h * w));
        const int fr = (int)(q / (h * w));
        const int row = pix / w;
        if (dx1) dx1[e] = dg[(size_t)(fr / S) * C + c] * inv_global;
        float g = 0.f;
        for (int p = 0; p < bins.nparts; ++p)
            if (row >= bins.start[p] && row < bins.end[p])
                g += dnodes[((size_t)fr * bins.nparts + p) * C + c] / (float)((bins.end[p] - bins.start[p]) * w);
        dx2[e] = g;
    }
}

// attention pooling backward. forward: n_sp = |f_sp|, a_sp = n_sp / max(N_p, 1e-12), N_p = sum_s n_sp, att_f = mean_p sum_s a_sp f_sp.
// With g = d loss / d att_f, u_sp = g . f_sp, ubar_p = sum_s a_sp u_sp:
//     d f_sp = (a_sp / P) g + (u_sp - ubar_p) / (P N_p n_sp) f_sp          (second term 0 where n_sp == 0 or N_p is clamped)
// grid = B, 256 threads: the V dot products by wavefronts, then the per-part sums, then the channel sweep.
__global__ __launch_bounds__(256) void attn_pool_backward_kernel(const float* __restrict__ nodes, const float* __restrict__ g,
                                                                 float* __restrict__ dnodes, int S, int P, int C) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int V = S * P;
    float* s_n = s_mem;          // [V] norms
    float* s_u = s_mem + V;      // [V] g . f
    float* s_a = s_mem + 2 * V;  // [V] attention weights
    float* s_k = s_mem + 3 * V;  // [V] coefficient of f_sp
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* gb = g + (size_t)b * C;
    for (int v = wave; v < V; v += 4) {
        const float* fv = nodes + ((size_t)b * V + v) * C;
        float nn = 0.f, uu = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 a = *reinterpret_cast<const float4*>(fv + c);
            const float4 q = *reinterpret_cast<const float4*>(gb + c);
            nn = fmaf(a.x, a.x, nn); nn = fmaf(a.y, a.y, nn); nn = fmaf(a.z, a.z, nn); nn = fmaf(a.w, a.w, nn);
            uu = fmaf(a.x, q.x, uu); uu = fmaf(a.y, q.y, uu); uu = fmaf(a.z, q.z, uu); uu = fmaf(a.w, q.w, uu);
        }
        nn = wave_sum(nn);
        uu = wave_sum(uu);
        if (lane == 0) { s_n[v] = sqrtf(nn); s_u[v] = uu; }
    }
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        float tot = 0.f;
        for (int s = 0; s < S; ++s) tot += s_n[s * P + p];
        const bool clamped = !(tot > 1e-12f);
        const float den = clamped ? 1e-12f : tot;
        float ubar = 0.f;
        for (int s = 0; s < S; ++s) {
            const float a = s_n[s * P + p] / den;
            s_a[s * P + p] = a;
            ubar = fmaf(a, s_u[s * P + p], ubar);
        }
        for (int s = 0; s < S; ++s) {
            const float n = s_n[s * P + p];
            s_k[s * P + p] = (clamped || !(n > 0.f)) ? 0.f : (s_u[s * P + p] - ubar) / ((float)P * den * n);
        }
    }
    __syncthreads();
    const float invP = 1.f / (float)P;
    for (int v = 0; v < V; ++v) {
        const float a = s_a[v] * invP, k = s_k[v];
        const float* fv = nodes + ((size_t)b * V + v) * C;
        float* dv = dnodes + ((size_t)b * V + v) * C;
        for (int c = tid * 4; c < C; c += 1024) {
            const float4 f4 = *reinterpret_cast<const float4*>(fv + c);
            const float4 g4 = *reinterpret_cast<const float4*>(gb + c);
            *reinterpret_cast<float4*>(dv + c) = make_float4(fmaf(a, g4.x, k * f4.x), fmaf(a, g4.y, k * f4.y), fmaf(a, g4.z, k * f4.z), fmaf(a, g4.w, k * f4.w));
        }
    }
}

// Backward of the adaptive graph (forward: agrl_graph_gram + agrl_graph_finalize). Per tracklet, from the summed Gram matrix:
//   D2_ij = n_i + n_j - 2 g_ij (n = diagonal), D = sqrt(max(D2, 1e-12)), S = 2 / (exp(D) + 1), r_i = sum_j S_ij, Shat = S / r_i,
//   G = Shat (learned only) or (Ahat + Shat) / 2; only Shat depends on f.
//   dShat = dG * (use_pose ? 1/2 : 1);  dS_ij = (dShat_ij - sum_k dShat_ik Shat_ik) / r_i;  dD = -S (1 - S/2) dS;
//   E_ij = dD2_ij = dD_ij / (2 D_ij) where D2_ij > 1e-12 and i != j, else 0   (D2_ii == 0 identically: no gradient through the diagonal;
//   the reference's autograd forms it as two cancelling fp32 terms);   T = E + E^T;   M = 2 (diag(rowsum T) - T)   =>   d loss / d f = M f.
// grid = B, 256 threads; V <= 128 (three V x V fp32 images in LDS).
__global__ __launch_bounds__(256) void graph_matrix_backward_kernel(const float* __restrict__ gram_part, int nz,
                                                                    const float* __restrict__ dG, float* __restrict__ Mout, int V,
                                                                    int use_pose, int mask_diag) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_s = s_mem;               // S, then E
    float* s_d = s_mem + V * V;       // D
    float* s_x = s_mem + 2 * V * V;   // dShat, then T
    float* s_n = s_mem + 3 * V * V;   // [V] norms / row sums / row sums of T
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* gp = gram_part + (size_t)b * nz * V * V;
    const float half = use_pose ? 0.5f : 1.f;
    for (int e = tid; e < V * V; e += 256) {
        float acc = 0.f;
        for (int z = 0; z < nz; ++z) acc += gp[(size_t)z * V * V + e];
        s_d[e] = acc;                                  // Gram
        s_x[e] = dG[(size_t)b * V * V + e] * half;     // dShat
    }
    __syncthreads();
    for (int j = tid; j < V; j += 256) s_n[j] = s_d[j * V + j];
    __syncthreads();
    for (int e = tid; e < V * V; e += 256) {
        const int i = e / V, j = e - i * V;
        const float d2 = fmaxf((s_n[j] + s_n[i]) - 2.f * s_d[e], 1e-12f);
        const float d = sqrtf(d2);
        float sv = 2.f / (expf(d) + 1.f);
        if (mask_diag && i == j) sv = 0.f;
        s_s[e] = sv;
        // marks entries that pass no gradient: the diagonal and clamped distances
        s_d[e] = (i == j || !(d2 > 1e-12f)) ? 0.f : d;
    }
    __syncthreads();
    // per row: r_i and c_i = sum_k dShat_ik Shat_ik, then dS -> dD -> E in place of S
    for (int i = wave; i < V; i += 4) {
        float r = 0.f;
        for (int j = lane; j < V; j += 64) r += fabsf(s_s[i * V + j]);
        r = wave_sum(r);
        const float den = fmaxf(r, 1e-12f);
        float cdot = 0.f;
        for (int j = lane; j < V; j += 64) cdot = fmaf(s_x[i * V + j], s_s[i * V + j] / den, cdot);
        cdot = wave_sum(cdot);
        for (int j = lane; j < V; j += 64) {
            const float sv = s_s[i * V + j], d = s_d[i * V + j];
            const float dS = r > 1e-12f ? (s_x[i * V + j] - cdot) / den : 0.f;
            const float dD = -sv * (1.f - 0.5f * sv) * dS;
            s_s[i * V + j] = d > 0.f ? dD / (2.f * d) : 0.f;   // E_ij
        }
    }
    __syncthreads();
    for (int e = tid; e < V * V; e += 256) {
        const int i = e / V, j = e - i * V;
        s_x[e] = s_s[e] + s_s[j * V + i];              // T
    }
    __syncthreads();
    for (int i = wave; i < V; i += 4) {
        float t = 0.f;
        for (int j = lane; j < V; j += 64) t += s_x[i * V + j];
        t = wave_sum(t);
        if (lane == 0) s_n[i] = t;
    }
    __syncthreads();
    for (int e = tid; e < V * V; e += 256) {
        const int i = e / V, j = e - i * V;
        Mout[(size_t)b * V * V + e] = 2.f * ((i == j ? s_n[i] : 0.f) - s_x[e]);
    }
}

// label-smoothed cross entropy: loss = sum_k mean_i(-q_ik log p_ik), q = (1 - eps) onehot + eps / K; dlogits = (p - q) / n.
// grid = n rows (one workgroup per sample) + a finishing pass that adds the row losses in row order.
__global__ __launch_bounds__(256) void xent_rows_kernel(const float* __restrict__ logits, const int32_t* __restrict__ targets, int K,
                                                        float eps, float inv_n, float* __restrict__ row_loss, float* __restrict__ dlogits) {
    __shared__ float s_red[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* z = logits + (size_t)i * K;
    float m = -INFINITY;
    for (int k = tid; k < K; k += 256) m = fmaxf(m, z[k]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    __syncthreads();
    float se = 0.f, sz = 0.f;
    for (int k = tid; k < K; k += 256) {
        se += expf(z[k] - m);
        sz += z[k];
    }
    se = wave_sum(se);
    if (lane == 0) s_red[wave] = se;
    __syncthreads();
    se = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    sz = wave_sum(sz);
    if (lane == 0) s_red[wave] = sz;
    __syncthreads();
    sz = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    const float lse = m + logf(se);
    const int y = targets[i];
    // a label outside [0, K) (class count mismatch, un-relabelled pids): the reference's scatter_ raises a device assert
    // (cross_entropy_loss.py:31); nothing can be raised from here without a host sync, so the row's loss and gradient
    // become NaN -- the step's loss is NaN and says so -- and z[y] is never read
    const bool bad = y < 0 || y >= K;
    // -sum_k q_k log p_k = -(1 - eps) (z_y - lse) - (eps / K) (sum_k z_k - K lse)
    if (tid == 0) row_loss[i] = bad ? NAN : -(1.f - eps) * (z[y] - lse) - (eps / (float)K) * (sz - (float)K * lse);
    for (int k = tid; k < K; k += 256) {
        const float p = expf(z[k] - lse);
        const float q = (k == y ? 1.f - eps : 0.f) + eps / (float)K;
        dlogits[(size_t)i * K + k] = bad ? NAN : (p - q) * inv_n;
    }
}

__global__ void xent_finish_kernel(const float* __restrict__ row_loss, int n, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < n; ++i) t += row_loss[i];
        loss[0] = t / (float)n;
    }
}

}  // namespace

extern "C" int agrl_axpby(const float* x, const float* y, float a, float b, float* out, size_t total, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && out && total > 0, "agrl_axpby: bad arguments");
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, a, b, out, total);
    AGRL_CHECK_LAUNCH("agrl_axpby");
    return 0;
}

extern "C" int agrl_part_pool_backward(const float* dg, const float* dnodes, float* dx1, float* dx2, int F, int S, int h, int w, int C,
                                       const int* splits, int n_splits, agrl_stream_t stream) {
    AGRL_CHECK_ARG(dnodes && dx2 && splits && (dx1 == nullptr) == (dg == nullptr), "agrl_part_pool_backward: bad pointers");
    AGRL_CHECK_ARG(F > 0 && S > 0 && F % S == 0 && h > 0 && w > 0 && C > 0 && n_splits > 0, "agrl_part_pool_backward: bad shape");
    PoolBins bins;
    int P = 0;
    for (int i = 0; i < n_splits; ++i) {
        const int n = splits[i];
        AGRL_CHECK_ARG(n > 0 && P + n <= 16, "agrl_part_pool_backward: at most 16 parts");
        for (int j = 0; j < n; ++j) {  // AdaptiveAvgPool2d bins: [floor(j*h/n), ceil((j+1)*h/n))
            bins.start[P] = (j * h) / n;
            bins.end[P] = ((j + 1) * h + n - 1) / n;
            ++P;
        }
    }
    bins.nparts = P;
    for (int i = P; i < 16; ++i) bins.start[i] = bins.end[i] = 0;
    const size_t total = (size_t)F * h * w * C;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(part_pool_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dg, dnodes, dx1, dx2, S, h, w, C,
                       1.f / ((float)S * (float)h * (float)w), bins, total);
    AGRL_CHECK_LAUNCH("agrl_part_pool_backward");
    return 0;
}

extern "C" int agrl_attn_pool_backward(const float* nodes, const float* datt, float* dnodes, int B, int S, int P, int C,
                                       agrl_stream_t stream) {
    AGRL_CHECK_ARG(nodes && datt && dnodes && B > 0 && S > 0 && P > 0 && C > 0 && (C % 4) == 0, "agrl_attn_pool_backward: bad arguments");
    const size_t lds = (size_t)4 * S * P * sizeof(float);
    AGRL_CHECK_ARG(lds <= 64 * 1024, "agrl_attn_pool_backward: S*P too large");
    hipLaunchKernelGGL(attn_pool_backward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, nodes, datt, dnodes, S, P, C);
    AGRL_CHECK_LAUNCH("agrl_attn_pool_backward");
    return 0;
}

extern "C" int agrl_graph_matrix_backward(const float* gram_part, int nz, const float* dG, float* M, int B, int V, int use_pose,
                                          int mask_diag, agrl_stream_t stream) {
    AGRL_CHECK_ARG(gram_part && dG && M && nz > 0 && B > 0 && V > 0, "agrl_graph_matrix_backward: bad arguments");
    const size_t lds = ((size_t)3 * V * V + V) * sizeof(float);
    AGRL_CHECK_ARG(lds <= 160 * 1024, "agrl_graph_matrix_backward: V=%d too large (V <= 115)", V);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)graph_matrix_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        AGRL_CHECK_ARG(e == hipSuccess, "agrl_graph_matrix_backward: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(graph_matrix_backward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, gram_part, nz, dG, M, V, use_pose, mask_diag);
    AGRL_CHECK_LAUNCH("agrl_graph_matrix_backward");
    return 0;
}

extern "C" int agrl_xent_label_smooth(const float* logits, const int32_t* targets, int n, int K, float eps, float* loss, float* dlogits,
                                      float* row_loss, agrl_stream_t stream) {
    AGRL_CHECK_ARG(logits && targets && loss && dlogits && row_loss && n > 0 && K > 0, "agrl_xent_label_smooth: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(xent_rows_kernel, dim3(n), dim3(256), 0, st, logits, targets, K, eps, 1.f / (float)n, row_loss, dlogits);
    hipLaunchKernelGGL(xent_finish_kernel, dim3(1), dim3(64), 0, st, row_loss, n, loss);
    AGRL_CHECK_LAUNCH("agrl_xent_label_smooth");
    return 0;
}
