// Stem of the frame feature extractor: conv 7x7/2 (3->64, BN folded) + ReLU + maxpool 3x3/2, fused.
// Replaces torchreid/models/vmgn.py:281-284. Reads the driver's fp32 NCHW frames directly and writes
// the NHWC layout every later kernel uses, so no separate layout-conversion pass exists.
//
// One 256-thread workgroup produces a 4x8 tile of POOLED pixels x 64 channels of one frame:
//   input patch 3 x 23 x 39 (zero padded)   -> LDS
//   weights 147 x 64 (k-major)              -> LDS
//   conv tile 9 x 17 x 64, +bias, ReLU      -> registers -> LDS (over the weight region)
//   3x3/2 max                               -> NHWC store, 64 channels contiguous per wavefront
// Out-of-range conv positions are stored as 0, which is exact for a max over post-ReLU values whose
// window always contains at least one in-range element.
#include "agrl_common.h"

namespace {
constexpr int PT_H = 4, PT_W = 8;                // pooled tile
constexpr int CT_H = 2 * PT_H + 1, CT_W = 2 * PT_W + 1;  // conv tile 9 x 17
constexpr int NPOS = CT_H * CT_W;                // 153
constexpr int IT_H = 2 * (CT_H - 1) + 7, IT_W = 2 * (CT_W - 1) + 7;  // 23 x 39
constexpr int IT_WP = 40;
constexpr int KTAPS = 147;
constexpr int POS_PER_THREAD = (NPOS + 15) / 16;  // 10

template <typename TOUT>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, TOUT* __restrict__ out,
                                                   int H, int W, int CH, int CW, int PH, int PW, int tiles_w,
                                                   int tiles_hw) {
    __shared__ __attribute__((aligned(16))) float s_patch[3 * IT_H * IT_WP];
    __shared__ __attribute__((aligned(16))) float s_wc[NPOS * 64];  // weights (147x64) then conv tile (153x64)

    const int tid = threadIdx.x;
    const int n = blockIdx.x / tiles_hw;
    const int trem = blockIdx.x - n * tiles_hw;
    const int ph0 = (trem / tiles_w) * PT_H;
    const int pw0 = (trem % tiles_w) * PT_W;
    const int cr0 = 2 * ph0 - 1, cc0 = 2 * pw0 - 1;  // first conv row/col of the tile
    const int iy0 = 2 * cr0 - 3, ix0 = 2 * cc0 - 3;  // first input row/col of the patch

    const float* xn = x + (size_t)n * 3 * H * W;
    for (int e = tid; e < 3 * IT_H * IT_WP; e += 256) {
        const int c = e / (IT_H * IT_WP);
        const int r = (e / IT_WP) % IT_H;
        const int q = e % IT_WP;
        const int iy = iy0 + r, ix = ix0 + q;
        float v = 0.f;
        if (q < IT_W && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xn[((size_t)c * H + iy) * W + ix];
        s_patch[e] = v;
    }
    // w is (64, 147) row-major (OHWI flattened); stage transposed to [k][o]
    for (int e = tid; e < KTAPS * 64; e += 256) {
        const int o = e / KTAPS, k = e - o * KTAPS;
        s_wc[k * 64 + o] = w[e];
    }
    __syncthreads();

    const int og = tid & 15;  // channels 4*og .. 4*og+3
    const int pg = tid >> 4;  // positions pg + 16*i
    float acc[POS_PER_THREAD][4];
    int pbase[POS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < POS_PER_THREAD; ++i) {
        int pos = pg + 16 * i;
        pos = pos < NPOS ? pos : NPOS - 1;
        const int cy = pos / CT_W, cx = pos - cy * CT_W;
        pbase[i] = (2 * cy) * IT_WP + 2 * cx;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    }
    for (int r = 0; r < 7; ++r) {
        for (int s = 0; s < 7; ++s) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 wv = *reinterpret_cast<const float4*>(&s_wc[((r * 7 + s) * 3 + c) * 64 + og * 4]);
                const float* pc = s_patch + c * (IT_H * IT_WP) + r * IT_WP + s;
#pragma unroll
                for (int i = 0; i < POS_PER_THREAD; ++i) {
                    const float xv = pc[pbase[i]];
                    acc[i][0] = fmaf(xv, wv.x, acc[i][0]);
                    acc[i][1] = fmaf(xv, wv.y, acc[i][1]);
                    acc[i][2] = fmaf(xv, wv.z, acc[i][2]);
                    acc[i][3] = fmaf(xv, wv.w, acc[i][3]);
                }
            }
        }
    }
    __syncthreads();  // all weight reads done; reuse the region for the conv tile
    const float4 bv = *reinterpret_cast<const float4*>(bias + og * 4);
#pragma unroll
    for (int i = 0; i < POS_PER_THREAD; ++i) {
        const int pos = pg + 16 * i;
        if (pos < NPOS) {
            const int cy = pos / CT_W, cx = pos - cy * CT_W;
            const int cr = cr0 + cy, cc = cc0 + cx;
            const bool in = (unsigned)cr < (unsigned)CH && (unsigned)cc < (unsigned)CW;
            float4 v;
            v.x = in ? relu_nan(acc[i][0] + bv.x) : 0.f;
            v.y = in ? relu_nan(acc[i][1] + bv.y) : 0.f;
            v.z = in ? relu_nan(acc[i][2] + bv.z) : 0.f;
            v.w = in ? relu_nan(acc[i][3] + bv.w) : 0.f;
            *reinterpret_cast<float4*>(&s_wc[pos * 64 + og * 4]) = v;
        }
    }
    __syncthreads();

    const int ch = tid & 63;
#pragma unroll
    for (int i = 0; i < (PT_H * PT_W) / 4; ++i) {
        const int pp = (tid >> 6) + 4 * i;
        const int py = pp / PT_W, px = pp - py * PT_W;
        const int ph = ph0 + py, pw = pw0 + px;
        if (ph < PH && pw < PW) {
            float m = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, s_wc[((2 * py + dy) * CT_W + 2 * px + dx) * 64 + ch]);
            DT<TOUT>::st(out + (((size_t)n * PH + ph) * PW + pw) * 64 + ch, m);
        }
    }
}
}  // namespace

extern "C" int agrl_stem_conv_bn_relu_maxpool(const float* x, const float* w, const float* bias, void* out,
                                              int N, int H, int W, int out_dtype, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w && bias && out, "agrl_stem: null pointer");
    AGRL_CHECK_ARG(N > 0 && H >= 7 && W >= 7, "agrl_stem: bad shape N=%d H=%d W=%d", N, H, W);
    AGRL_CHECK_ARG(out_dtype == AGRL_F32 || out_dtype == AGRL_LP16, "agrl_stem: bad dtype %d", out_dtype);
    const int CH = (H + 6 - 7) / 2 + 1, CW = (W + 6 - 7) / 2 + 1;
    const int PH = (CH + 2 - 3) / 2 + 1, PW = (CW + 2 - 3) / 2 + 1;
    const int tiles_h = cdiv(PH, PT_H), tiles_w = cdiv(PW, PT_W);
    const long long grid = (long long)N * tiles_h * tiles_w;
    AGRL_CHECK_ARG(grid < (1ll << 31), "agrl_stem: grid too large");
    if (out_dtype == AGRL_F32)
        hipLaunchKernelGGL(stem_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                           (float*)out, H, W, CH, CW, PH, PW, tiles_w, tiles_h * tiles_w);
    else
        hipLaunchKernelGGL(stem_kernel<lp16_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                           (lp16_t*)out, H, W, CH, CW, PH, PW, tiles_w, tiles_h * tiles_w);
    AGRL_CHECK_LAUNCH("agrl_stem");
    return 0;
}
