// Train-step kernels of the conv trunk (BASELINE config 4; reference train(), train_vidreid_xent_htri.py:397-413, running
// Bottleneck.forward / backward of torchreid/models/vmgn.py:45-65 under model.train()): everything around the three conv
// GEMMs (forward, data gradient, weight gradient -- all on the implicit-GEMM kernel of igemm.hip in exact-fp32 MFMA) that
// torch's autograd would otherwise run as stock kernels.
//
//   agrl_bn_stats          batch statistics of a conv output (BatchNorm2d in train mode, vmgn.py:49/53/57)
//   agrl_bn_apply          y * scale + shift (+ residual) (ReLU)
//   agrl_bn_backward       ReLU mask + BatchNorm backward: dgamma / dbeta reductions, then dy (and the masked gradient that
//                          flows on to the residual branch)
//   agrl_im2col_t          channel-major, tap-expanded transpose of an NHWC tensor: the K-contiguous operand of the weight-
//                          gradient GEMM dW[co][tap,ci] = sum_pixels dy[pixel][co] x[pixel + tap][ci]
//   agrl_gemm_nt_splitk    y = x w^T with K split over workgroups (weight gradients: K = pixels, few output tiles)
//   agrl_maxpool3x3s2      forward with the arg-max tap, and its backward (gather form, deterministic)
//
// All HBM-bound, deterministic (no atomics: fixed-order two-stage reductions with double partials).
#include "agrl_common.h"
#include "igemm_dev.h"

namespace {

// ---- per-channel reductions over the rows of an (M, C) fp32 matrix ---------------------------------------------------------
// MODE 0: s1 = sum y,  s2 = sum y^2                      (batch statistics)
// MODE 1: s1 = sum dz, s2 = sum dz * xhat                (BatchNorm backward), dz = relu ? (out > 0 ? dout : 0) : dout,
//                                                         xhat = (y - mean) * invstd
// grid = (ceil(C/64), chunks), 256 threads = 4 row lanes x 64 channels; partial[chunk][2][C] in double.
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                        const float* __restrict__ out, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int relu, float slope, int M, int C,
                                                        int rows_per_chunk, double* __restrict__ partial) {
    __shared__ float s_a[4][64], s_b[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    float a = 0.f, b = 0.f;
    if (c < C) {
        float mu = 0.f, is = 0.f;
        if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
        for (int r = r0 + rl; r < r1; r += 4) {
            const size_t i = (size_t)r * C + c;
            if (MODE == 0) {
                const float v = y[i];
                a += v;
                b = fmaf(v, v, b);
            } else {
                float dz = dout[i];
                if (relu && !(out[i] > 0.f)) dz *= slope;   // ReLU: slope 0; LeakyReLU keeps the sign, so out > 0 <=> pre-activation > 0
                a += dz;
                b = fmaf(dz, (y[i] - mu) * is, b);
            }
        }
    }
    s_a[rl][threadIdx.x & 63] = a;
    s_b[rl][threadIdx.x & 63] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int t = threadIdx.x;
        const double sa = ((double)s_a[0][t] + (double)s_a[1][t]) + ((double)s_a[2][t] + (double)s_a[3][t]);
        const double sb = ((double)s_b[0][t] + (double)s_b[1][t]) + ((double)s_b[2][t] + (double)s_b[3][t]);
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = sa;
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = sb;
    }
}

// the four sign bits of float4 number e4 (written by bn_apply_kernel) as +1 / -1: stands in for the forward output wherever
// only "out > 0" is asked of it
__device__ inline float4 mask_as_float4(const unsigned char* __restrict__ mask, size_t e4) {
    const unsigned nib = ((unsigned)mask[e4 >> 1] >> ((unsigned)(e4 & 1) * 4)) & 15u;
    return make_float4((nib & 1u) ? 1.f : -1.f, (nib & 2u) ? 1.f : -1.f, (nib & 4u) ? 1.f : -1.f, (nib & 8u) ? 1.f : -1.f);
}

// The same reductions for C % 4 == 0 and 16-byte aligned operands (every layer of the model): a thread owns FOUR channels
// (16-byte loads; a 64-channel row is one 256-byte segment of 16 lanes, so a wave covers 4 rows of layer 1 per load instead of
// one 4-byte element per lane) and keeps four row-steps in flight. ``lanes`` = 16 / 32 / 64 float4 columns per block;
// 256 / lanes row lanes. grid = (ceil(C/4/lanes), chunks).
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_vec_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                            const float* __restrict__ out, const unsigned char* __restrict__ mask,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                            float slope, int M, int C, int rows_per_chunk, int lanes,
                                                            double* __restrict__ partial) {
    __shared__ float4 s_a[256], s_b[256];
    const int lc = threadIdx.x % lanes, rl = threadIdx.x / lanes, nrl = 256 / lanes;
    const int c4 = blockIdx.x * lanes + lc, C4 = C >> 2;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c4 < C4) {
        float4 mu = a, is = a;
        if (MODE == 1) { mu = reinterpret_cast<const float4*>(mean)[c4]; is = reinterpret_cast<const float4*>(invstd)[c4]; }
        const float4* y4 = reinterpret_cast<const float4*>(y) + c4;
        const float4* d4 = reinterpret_cast<const float4*>(dout) + c4;
        const float4* o4 = reinterpret_cast<const float4*>(out) + c4;
        auto acc = [&](const float4 v, float4 dz, const float4 o) {
            if (MODE == 0) {
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                b.x = fmaf(v.x, v.x, b.x); b.y = fmaf(v.y, v.y, b.y); b.z = fmaf(v.z, v.z, b.z); b.w = fmaf(v.w, v.w, b.w);
            } else {
                if (relu) {
                    if (!(o.x > 0.f)) dz.x *= slope;
                    if (!(o.y > 0.f)) dz.y *= slope;
                    if (!(o.z > 0.f)) dz.z *= slope;
                    if (!(o.w > 0.f)) dz.w *= slope;
                }
                a.x += dz.x; a.y += dz.y; a.z += dz.z; a.w += dz.w;
                b.x = fmaf(dz.x, (v.x - mu.x) * is.x, b.x); b.y = fmaf(dz.y, (v.y - mu.y) * is.y, b.y);
                b.z = fmaf(dz.z, (v.z - mu.z) * is.z, b.z); b.w = fmaf(dz.w, (v.w - mu.w) * is.w, b.w);
            }
        };
        int r = r0 + rl;
        for (; r + 3 * nrl < r1; r += 4 * nrl) {
            float4 v[4], d[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = (size_t)(r + u * nrl) * C4;
                v[u] = y4[i];
                if (MODE == 1) {
                    d[u] = d4[i];
                    o[u] = relu ? (mask ? mask_as_float4(mask, i + c4) : o4[i]) : v[u];
                } else {
                    d[u] = v[u]; o[u] = v[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc(v[u], d[u], o[u]);
        }
        for (; r < r1; r += nrl) {
            const size_t i = (size_t)r * C4;
            const float4 v = y4[i];
            acc(v, MODE == 1 ? d4[i] : v, (MODE == 1 && relu) ? (mask ? mask_as_float4(mask, i + c4) : o4[i]) : v);
        }
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = b;
    __syncthreads();
    if (rl == 0 && c4 < C4) {
        double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < nrl; ++k) {                      // fixed order: deterministic
            const float4 pa = s_a[k * lanes + lc], pb = s_b[k * lanes + lc];
            sa[0] += (double)pa.x; sa[1] += (double)pa.y; sa[2] += (double)pa.z; sa[3] += (double)pa.w;
            sb[0] += (double)pb.x; sb[1] += (double)pb.y; sb[2] += (double)pb.z; sb[3] += (double)pb.w;
        }
        double* pa = partial + ((size_t)blockIdx.y * 2 + 0) * C + 4 * c4;
        double* pb = partial + ((size_t)blockIdx.y * 2 + 1) * C + 4 * c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j] = sa[j]; pb[j] = sb[j]; }
    }
}

// MODE 0 -> mean, biased variance; MODE 1 -> the two sums themselves (dbeta, dgamma)
// grid = ceil(C/16) blocks of 16 channels x 16 chunk lanes: a lane adds every 16th partial (loads independent, four in flight),
// the 16 lane sums are added in lane order by the first 16 threads -- the summation order depends on ``chunks`` only.
template <int MODE, typename TP = double>
__global__ __launch_bounds__(256) void colreduce_final_kernel(const TP* __restrict__ partial, int chunks, int C, int M,
                                                              float* __restrict__ o1, float* __restrict__ o2, int kstride) {
    // chunk k: s1 at partial[k * kstride + c], s2 at partial[k * kstride + C + c] (kstride = 2 C for this file's reductions)
    __shared__ double s_a[16][17], s_b[16][17];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double sa = 0.0, sb = 0.0;
    if (c < C) {
        int k = kl;
        for (; k + 48 < chunks; k += 64) {
            double pa[4], pb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pa[u] = (double)partial[(size_t)(k + 16 * u) * kstride + c];
                pb[u] = (double)partial[(size_t)(k + 16 * u) * kstride + C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa += pa[u]; sb += pb[u]; }
        }
        for (; k < chunks; k += 16) {
            sa += (double)partial[(size_t)k * kstride + c];
            sb += (double)partial[(size_t)k * kstride + C + c];
        }
    }
    s_a[kl][cl] = sa;
    s_b[kl][cl] = sb;
    __syncthreads();
    if (kl != 0 || c >= C) return;
    sa = 0.0; sb = 0.0;
    for (int k = 0; k < 16; ++k) { sa += s_a[k][cl]; sb += s_b[k][cl]; }
    if (MODE == 0) {
        const double mu = sa / (double)M;
        o1[c] = (float)mu;
        o2[c] = (float)fmax(sb / (double)M - mu * mu, 0.0);
    } else {
        o1[c] = (float)sa;
        o2[c] = (float)sb;
    }
}

// ``mask`` (optional, with relu): one bit per element in linear order, set where the pre-activation is positive -- the only thing
// the backward pass needs from the output (8 x fewer bytes than re-reading it, twice)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res,
                                                       float* __restrict__ out, unsigned char* __restrict__ mask, int relu, float slope,
                                                       size_t total4, int C4) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t iters = (total4 + stride - 1) / stride;
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t it = 0; it < iters; ++it, e += stride) {        // uniform trip count: the mask nibbles travel through a shuffle
        unsigned bits = 0;
        if (e < total4) {
            const int c4 = (int)(e % C4);
            const float4 v = reinterpret_cast<const float4*>(y)[e];
            const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
            const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
            float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
            if (res) {
                const float4 r = reinterpret_cast<const float4*>(res)[e];
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            if (relu) {
                bits = (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
                o.x = o.x > 0.f ? o.x : slope * o.x; o.y = o.y > 0.f ? o.y : slope * o.y;
                o.z = o.z > 0.f ? o.z : slope * o.z; o.w = o.w > 0.f ? o.w : slope * o.w;
            }
            reinterpret_cast<float4*>(out)[e] = o;
        }
        if (mask) {
            const unsigned hi = __shfl_xor(bits, 1);             // the odd neighbour's nibble (0 past the end)
            if (!(e & 1) && e < total4) mask[e >> 1] = (unsigned char)(bits | (hi << 4));
        }
    }
}

// dy = k1[c] * (dz - k2[c] - xhat * k3[c]),  k1 = gamma * invstd, k2 = sum dz / M, k3 = sum(dz xhat) / M; dz optionally stored
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ s1, const float* __restrict__ s2, int relu,
                                                           float slope, float inv_m, float* __restrict__ dy, float* __restrict__ dz_out,
                                                           size_t total, int C) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        float dz = dout[e];
        if (relu && !(out[e] > 0.f)) dz *= slope;
        const float is = invstd[c];
        const float xhat = (y[e] - mean[c]) * is;
        dy[e] = gamma[c] * is * (dz - s1[c] * inv_m - xhat * (s2[c] * inv_m));
        if (dz_out) dz_out[e] = dz;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                               const float* __restrict__ y, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ s1, const float* __restrict__ s2, int relu,
                                                               float slope, float inv_m, float* __restrict__ dy,
                                                               float* __restrict__ dz_out, const unsigned char* __restrict__ mask,
                                                               size_t total4, int C4) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        float4 dz = reinterpret_cast<const float4*>(dout)[e];
        const float4 v = reinterpret_cast<const float4*>(y)[e];
        if (relu) {
            const float4 o = mask ? mask_as_float4(mask, e) : reinterpret_cast<const float4*>(out)[e];
            if (!(o.x > 0.f)) dz.x *= slope;
            if (!(o.y > 0.f)) dz.y *= slope;
            if (!(o.z > 0.f)) dz.z *= slope;
            if (!(o.w > 0.f)) dz.w *= slope;
        }
        const float4 is = reinterpret_cast<const float4*>(invstd)[c4], mu = reinterpret_cast<const float4*>(mean)[c4];
        const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
        const float4 a = reinterpret_cast<const float4*>(s1)[c4], b = reinterpret_cast<const float4*>(s2)[c4];
        float4 r;
        r.x = g.x * is.x * (dz.x - a.x * inv_m - (v.x - mu.x) * is.x * (b.x * inv_m));
        r.y = g.y * is.y * (dz.y - a.y * inv_m - (v.y - mu.y) * is.y * (b.y * inv_m));
        r.z = g.z * is.z * (dz.z - a.z * inv_m - (v.z - mu.z) * is.z * (b.z * inv_m));
        r.w = g.w * is.w * (dz.w - a.w * inv_m - (v.w - mu.w) * is.w * (b.w * inv_m));
        reinterpret_cast<float4*>(dy)[e] = r;
        if (dz_out) reinterpret_cast<float4*>(dz_out)[e] = dz;
    }
}

// ---- channel-major tap-expanded transpose ------------------------------------------------------------------------------------
// T[(tap * C + c)][m] = x[f][oh * stride - pad + r][ow * stride - pad + s][c] (0 outside the frame), m = (f, oh, ow),
// tap = r * S + s. grid = (ceil(Mout/32), ceil(C/32), R*S), 256 threads: a 32 x 32 tile through LDS (both sides coalesced).
__global__ __launch_bounds__(256) void im2col_t_kernel(const float* __restrict__ x, float* __restrict__ T, int H, int W, int C,
                                                       int OH, int OW, int R, int S, int stride, int pad, int Mout, int ldT) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tap = blockIdx.z;
    const int r = tap / S, s = tap - r * S;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty 0..7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = ty + 8 * i;
        const int m = m0 + ml;
        float v = 0.f;
        if (m < Mout && c0 + tx < C) {
            const int f = m / (OH * OW), rem = m - f * OH * OW;
            const int oh = rem / OW, ow = rem - oh * OW;
            const int ih = oh * stride - pad + r, iw = ow * stride - pad + s;
            if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[(((size_t)f * H + ih) * W + iw) * C + c0 + tx];
        }
        tile[ml][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cl = ty + 8 * i;
        if (c0 + cl < C && m0 + tx < ldT) T[((size_t)tap * C + c0 + cl) * ldT + m0 + tx] = tile[tx][cl];  // columns >= Mout: zeros
    }
}

// ---- pixel-major patches (the stem conv as a pointwise GEMM) ---------------------------------------------------------------
// P[m][(r * S + s) * C + c] = x[f][oh * stride - pad + r][ow * stride - pad + s][c] (0 outside the frame and in the padding
// columns R*S*C .. ldP-1), m = (f, oh, ow). With 3 input channels the implicit-GEMM kernels would pad every tap to their
// 32-channel k granularity (10.7 x the arithmetic); 7*7*3 = 147 patch columns padded to 160 make conv1 a 160 -> 64 pointwise
// layer whose forward and weight gradient run on the same kernels as every other 1x1. x is NHWC or (``nchw``) the frames as
// the data loader delivers them. CC / RR / SS: compile-time copies of C / R / S (0 = run time) so that the column -> (tap,
// channel) split costs multiplications, not divisions.
template <int CC, int RR, int SS>
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ x, float* __restrict__ P, int ldP4, int H, int W,
                                                          int C_, int OH, int OW, int R_, int S_, int stride, int pad, int nchw,
                                                          size_t total4) {
    const int C = CC ? CC : C_, R = RR ? RR : R_, S = SS ? SS : S_;
    const int ncol = R * S * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
        const size_t m = e / (size_t)ldP4;
        const int c4 = (int)(e - m * (size_t)ldP4);
        const int ow = (int)(m % (size_t)OW);
        const size_t t = m / (size_t)OW;
        const int oh = (int)(t % (size_t)OH), f = (int)(t / (size_t)OH);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * c4 + j;
            v[j] = 0.f;
            if (n < ncol) {
                const int tap = n / C, c = n - tap * C;
                const int r = tap / S, s_ = tap - r * S;
                const int ih = oh * stride - pad + r, iw = ow * stride - pad + s_;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W)
                    v[j] = nchw ? x[(((size_t)f * C + c) * H + ih) * W + iw] : x[(((size_t)f * H + ih) * W + iw) * C + c];
            }
        }
        reinterpret_cast<float4*>(P)[e] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- 3x3 / stride 2 / pad 1 max pooling (nn.MaxPool2d of the stem, vmgn.py:284) ------------------------------------------------
// forward: first maximum in window scan order wins (strict >), its tap index 0..8 is kept for the backward pass
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          unsigned char* __restrict__ idx, int H, int W, int C, int OH, int OW,
                                                          size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        size_t p = e / C;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int f = (int)(p / OH);
        float best = -INFINITY;
        int bi = 0;
        bool any = false;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ih = 2 * oh - 1 + r, iw = 2 * ow - 1 + s;
                if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                const float v = x[(((size_t)f * H + ih) * W + iw) * C + c];
                if (!any || v > best) { best = v; bi = r * 3 + s; any = true; }
            }
        out[e] = best;
        idx[e] = (unsigned char)bi;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                          float* __restrict__ dx, int H, int W, int C, int OH, int OW, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        size_t p = e / C;
        const int iw = (int)(p % W); p /= W;
        const int ih = (int)(p % H);
        const int f = (int)(p / H);
        float g = 0.f;
        // windows (oh, ow) that contain (ih, iw): 2 oh - 1 <= ih <= 2 oh + 1
        for (int oh = (ih >> 1); oh <= ((ih + 1) >> 1); ++oh) {
            if (oh < 0 || oh >= OH) continue;
            const int r = ih - (2 * oh - 1);
            if (r < 0 || r > 2) continue;
            for (int ow = (iw >> 1); ow <= ((iw + 1) >> 1); ++ow) {
                if (ow < 0 || ow >= OW) continue;
                const int s = iw - (2 * ow - 1);
                if (s < 0 || s > 2) continue;
                const size_t o = (((size_t)f * OH + oh) * OW + ow) * C + c;
                if (idx[o] == (unsigned char)(r * 3 + s)) g += dout[o];
            }
        }
        dx[e] = g;
    }
}

// float4 columns per block of the vectorised reduction (0: the scalar kernel -- C % 4 != 0)
static int reduce_lanes(int C) { return (C & 3) ? 0 : (C >= 256 ? 64 : (C >= 128 ? 32 : 16)); }

static int reduce_chunks(int M, int C, int* rows_per_chunk) {
    const int lanes = reduce_lanes(C);
    const int cg = lanes ? cdiv(C / 4, lanes) : cdiv(C, 64);
    int chunks = 2048 / cg;
    if (chunks < 1) chunks = 1;
    if (chunks > 512) chunks = 512;
    int rpc = cdiv(M, chunks);
    rpc = (rpc + 15) & ~15;                 // whole row-lane passes of either kernel
    *rows_per_chunk = rpc;
    return cdiv(M, rpc);
}

}  // namespace

// Everything nn.BatchNorm{1,2}d does with the batch statistics besides normalising, in ONE launch (it was eleven tiny torch
// kernels per BatchNorm layer, ~770 launches per train step): invstd = rsqrt(var + eps), the folded scale = gamma * invstd and
// shift = beta - mean * scale for agrl_bn_apply, and the running-statistics update of torch.nn.functional.batch_norm --
// running = (1 - momentum) * running + momentum * {mean, var * n / (n - 1)} -- plus num_batches_tracked += 1.
__global__ __launch_bounds__(256) void bn_fold_train_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float momentum, float unbias, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, long long* __restrict__ num_batches,
                                                            float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ invstd, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && num_batches) num_batches[0] += 1;
    if (c >= C) return;
    const float mu = mean[c], v = var[c];
    const float is = rsqrtf(v + eps);
    const float sc = gamma[c] * is;
    invstd[c] = is;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
    if (running_mean) {
        running_mean[c] = fmaf(momentum, mu, running_mean[c] * (1.f - momentum));
        running_var[c] = fmaf(momentum, v * unbias, running_var[c] * (1.f - momentum));
    }
}

extern "C" int agrl_bn_fold_train(const float* mean, const float* var, const float* gamma, const float* beta, float eps, float momentum,
                                  long long n, float* running_mean, float* running_var, long long* num_batches_tracked, float* scale,
                                  float* shift, float* invstd, int C, agrl_stream_t stream) {
    AGRL_CHECK_ARG(mean && var && gamma && beta && scale && shift && invstd && C > 0 && n > 0, "agrl_bn_fold_train: bad arguments");
    AGRL_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "agrl_bn_fold_train: running_mean and running_var go together");
    const float unbias = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
    hipLaunchKernelGGL(bn_fold_train_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, var, gamma, beta, eps, momentum, unbias,
                       running_mean, running_var, num_batches_tracked, scale, shift, invstd, C);
    AGRL_CHECK_LAUNCH("agrl_bn_fold_train");
    return 0;
}

extern "C" size_t agrl_bn_workspace(int M, int C) {
    int rpc;
    const int chunks = reduce_chunks(M, C, &rpc);
    return (size_t)chunks * 2 * C * sizeof(double);
}

extern "C" int agrl_bn_stats(const float* y, float* mean, float* var, int M, int C, void* workspace, size_t workspace_bytes,
                             agrl_stream_t stream) {
    AGRL_CHECK_ARG(y && mean && var && workspace && M > 0 && C > 0, "agrl_bn_stats: bad arguments");
    AGRL_CHECK_ARG(workspace_bytes >= agrl_bn_workspace(M, C) && (((uintptr_t)workspace) & 7) == 0, "agrl_bn_stats: workspace too small");
    int rpc;
    const int chunks = reduce_chunks(M, C, &rpc);
    hipStream_t st = (hipStream_t)stream;
    const int lanes = ((uintptr_t)y & 15) ? 0 : reduce_lanes(C);
    if (lanes)
        hipLaunchKernelGGL(colreduce_vec_kernel<0>, dim3(cdiv(C / 4, lanes), chunks), dim3(256), 0, st, y, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                           0.f, M, C, rpc, lanes, (double*)workspace);
    else
        hipLaunchKernelGGL(colreduce_kernel<0>, dim3(cdiv(C, 64), chunks), dim3(256), 0, st, y, nullptr, nullptr, nullptr, nullptr, 0, 0.f, M,
                           C, rpc, (double*)workspace);
    hipLaunchKernelGGL(colreduce_final_kernel<0>, dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)workspace, chunks, C, M, mean, var, 2 * C);
    AGRL_CHECK_LAUNCH("agrl_bn_stats");
    return 0;
}

// mean / biased variance from the per-tile sums agrl_conv2d_stats left ([rows][2][C] floats = a rows x 2C matrix whose column
// sums are wanted): the vectorised column reduction over the rows (double partials per chunk), then one small final kernel
extern "C" int agrl_bn_stats_from_partials(const float* partial, int rows, int C, int M, float* mean, float* var, void* workspace,
                                           size_t workspace_bytes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(partial && mean && var && workspace && rows > 0 && C > 0 && M > 0, "agrl_bn_stats_from_partials: bad arguments");
    AGRL_CHECK_ARG((C % 2) == 0 && (((uintptr_t)partial) & 15) == 0, "agrl_bn_stats_from_partials: C must be even, partial 16-byte aligned");
    AGRL_CHECK_ARG(workspace_bytes >= agrl_bn_workspace(rows, 2 * C) && (((uintptr_t)workspace) & 7) == 0,
                   "agrl_bn_stats_from_partials: workspace too small (agrl_bn_workspace(rows, 2 * C))");
    int rpc;
    const int chunks = reduce_chunks(rows, 2 * C, &rpc);
    const int lanes = reduce_lanes(2 * C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colreduce_vec_kernel<0>, dim3(cdiv(2 * C / 4, lanes), chunks), dim3(256), 0, st, partial, nullptr, nullptr, nullptr, nullptr,
                       nullptr, 0, 0.f, rows, 2 * C, rpc, lanes, (double*)workspace);
    // plane 0 of chunk k holds the 2C column sums [sum | sum of squares]: the usual final reduce with a chunk stride of 4C
    hipLaunchKernelGGL((colreduce_final_kernel<0, double>), dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)workspace, chunks, C, M, mean, var,
                       4 * C);
    AGRL_CHECK_LAUNCH("agrl_bn_stats_from_partials");
    return 0;
}

extern "C" int agrl_bn_apply(const float* y, const float* scale, const float* shift, const float* residual, float* out,
                             unsigned char* mask, int M, int C, int relu, float slope, agrl_stream_t stream) {
    AGRL_CHECK_ARG(y && scale && shift && out && M > 0 && C > 0 && (C % 4) == 0, "agrl_bn_apply: bad arguments (C %% 4 == 0)");
    const uintptr_t al = (uintptr_t)y | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)residual | (uintptr_t)out;
    AGRL_CHECK_ARG((al & 15) == 0, "agrl_bn_apply: operands must be 16-byte aligned");
    const size_t total4 = (size_t)M * C / 4;
    const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, scale, shift, residual, out, relu ? mask : nullptr, relu,
                       slope, total4, C / 4);
    AGRL_CHECK_LAUNCH("agrl_bn_apply");
    return 0;
}

extern "C" int agrl_bn_backward(const float* dout, const float* out, const unsigned char* mask, const float* y, const float* mean,
                                const float* invstd, const float* gamma, int relu, float slope, float* dy, float* dz, float* dgamma,
                                float* dbeta, int M, int C, void* workspace, size_t workspace_bytes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(dout && y && mean && invstd && gamma && dy && dgamma && dbeta && workspace, "agrl_bn_backward: null pointer");
    AGRL_CHECK_ARG(!relu || out || mask, "agrl_bn_backward: the ReLU mask needs the forward output or the sign mask agrl_bn_apply wrote");
    AGRL_CHECK_ARG(M > 0 && C > 0, "agrl_bn_backward: bad shape");
    AGRL_CHECK_ARG(workspace_bytes >= agrl_bn_workspace(M, C) && (((uintptr_t)workspace) & 7) == 0, "agrl_bn_backward: workspace too small");
    int rpc;
    const int chunks = reduce_chunks(M, C, &rpc);
    hipStream_t st = (hipStream_t)stream;
    const uintptr_t al = (uintptr_t)dout | (uintptr_t)out | (uintptr_t)y | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma |
                         (uintptr_t)dy | (uintptr_t)dz | (uintptr_t)dgamma | (uintptr_t)dbeta;
    const int lanes = (al & 15) ? 0 : reduce_lanes(C);
    AGRL_CHECK_ARG(!relu || out || lanes, "agrl_bn_backward: the sign-mask form needs C %% 4 == 0 and 16-byte aligned operands");
    const unsigned char* mk = relu ? mask : nullptr;
    if (lanes)
        hipLaunchKernelGGL(colreduce_vec_kernel<1>, dim3(cdiv(C / 4, lanes), chunks), dim3(256), 0, st, y, dout, out, mk, mean, invstd, relu,
                           slope, M, C, rpc, lanes, (double*)workspace);
    else
        hipLaunchKernelGGL(colreduce_kernel<1>, dim3(cdiv(C, 64), chunks), dim3(256), 0, st, y, dout, out, mean, invstd, relu, slope, M, C,
                           rpc, (double*)workspace);
    hipLaunchKernelGGL(colreduce_final_kernel<1>, dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)workspace, chunks, C, M, dbeta, dgamma, 2 * C);
    const size_t total = (size_t)M * C;
    if (lanes) {
        const size_t total4 = total / 4;
        const int blocks = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
        hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(blocks), dim3(256), 0, st, dout, out, y, mean, invstd, gamma, dbeta, dgamma, relu,
                           slope, 1.f / (float)M, dy, dz, mk, total4, C / 4);
    } else {
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, dout, out, y, mean, invstd, gamma, dbeta, dgamma, relu,
                           slope, 1.f / (float)M, dy, dz, total, C);
    }
    AGRL_CHECK_LAUNCH("agrl_bn_backward");
    return 0;
}

extern "C" int agrl_im2col_t(const float* x, float* T, int ldT, int F, int H, int W, int C, int R, int S, int stride, int pad,
                             agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && T && F > 0 && H > 0 && W > 0 && C > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "agrl_im2col_t: bad arguments");
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    AGRL_CHECK_ARG(OH > 0 && OW > 0 && R * S <= 65535, "agrl_im2col_t: empty output");
    const int Mout = F * OH * OW;
    AGRL_CHECK_ARG(ldT >= Mout, "agrl_im2col_t: ldT=%d < %d output pixels", ldT, Mout);
    hipLaunchKernelGGL(im2col_t_kernel, dim3(cdiv(ldT, 32), cdiv(C, 32), R * S), dim3(256), 0, (hipStream_t)stream, x, T, H, W, C, OH,
                       OW, R, S, stride, pad, Mout, ldT);
    AGRL_CHECK_LAUNCH("agrl_im2col_t");
    return 0;
}

extern "C" int agrl_im2col_rows(const float* x, float* P, int ldP, int F, int H, int W, int C, int R, int S, int stride, int pad,
                                int nchw, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && P && F > 0 && H > 0 && W > 0 && C > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "agrl_im2col_rows: bad arguments");
    AGRL_CHECK_ARG(ldP >= R * S * C && (ldP % 4) == 0 && (((uintptr_t)P) & 15) == 0, "agrl_im2col_rows: ldP=%d must be >= R*S*C=%d, a multiple of 4, P 16-byte aligned", ldP, R * S * C);
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    AGRL_CHECK_ARG(OH > 0 && OW > 0, "agrl_im2col_rows: empty output");
    const size_t total4 = (size_t)F * OH * OW * (ldP / 4);
    const int blocks = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    if (C == 3 && R == 7 && S == 7)
        hipLaunchKernelGGL((im2col_rows_kernel<3, 7, 7>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, P, ldP / 4, H, W, C, OH, OW, R, S,
                           stride, pad, nchw, total4);
    else
        hipLaunchKernelGGL((im2col_rows_kernel<0, 0, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, P, ldP / 4, H, W, C, OH, OW, R, S,
                           stride, pad, nchw, total4);
    AGRL_CHECK_LAUNCH("agrl_im2col_rows");
    return 0;
}

extern "C" int agrl_maxpool3x3s2(const float* x, float* out, unsigned char* idx, int F, int H, int W, int C, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && out && idx && F > 0 && H > 0 && W > 0 && C > 0, "agrl_maxpool3x3s2: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)F * OH * OW * C;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, idx, H, W, C, OH, OW, total);
    AGRL_CHECK_LAUNCH("agrl_maxpool3x3s2");
    return 0;
}

extern "C" int agrl_maxpool3x3s2_backward(const float* dout, const unsigned char* idx, float* dx, int F, int H, int W, int C,
                                          agrl_stream_t stream) {
    AGRL_CHECK_ARG(dout && idx && dx && F > 0 && H > 0 && W > 0 && C > 0, "agrl_maxpool3x3s2_backward: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)F * H * W * C;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, idx, dx, H, W, C, OH, OW, total);
    AGRL_CHECK_LAUNCH("agrl_maxpool3x3s2_backward");
    return 0;
}
