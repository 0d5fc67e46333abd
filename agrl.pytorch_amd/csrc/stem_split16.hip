// Split-fp16 stem (round 6, the conforming mode at speed): conv 7x7/2 (3->64, BN folded) + ReLU + maxpool 3x3/2, fused, in the
// arithmetic of agrl_conv2d_bn_act_split16 -- every product as three fp16 MFMAs on fp16 high / low halves, fp32 accumulation -- with
// fp32 NHWC output. vmgn.py:281-284. The exact-fp32 stem (stem.hip, a VALU kernel) takes 1.0 ms of the conforming mode's 12.6 ms
// step; this one runs the 16-bit stem's structure (stem_mfma.hip: the 7x7x3 filter as a K = 7 x 32 contraction straight out of a
// [y][x][4] patch in LDS, no im2col) three times over:
//   patch     39 x 40 x 4 as TWO fp16 images, hi = fp16(x) and lo = fp16(x - hi) (frames are O(1): lo is unscaled; MFMA keeps fp16
//             subnormals, tests/test_gpu_kernels.py::test_fp16_mfma_keeps_subnormal_operands)
//   weights   wh, wl of w 2^k in the 16-bit stem's packed form (64 x 240 fp16 each, resident in LDS); acc += wh xl + wl xh + wh xh
//   conv tile fp32, 17 x 17 positions x 32 channels at a time (two channel halves per tile: 37 KB instead of 74), v = relu(2^-k acc + b)
//   pooling   3 x 3 / 2 max on the fp32 values, fp32 NHWC store
// LDS 25 + 60 + 37 KB: one 512-thread workgroup per CU, persistent over tiles, the next tile's pixels prefetched into registers.
#include <stdlib.h>

#include "agrl_common.h"

namespace {
constexpr int PT = 8;                 // pooled tile edge
constexpr int CT = 2 * PT + 1;        // conv tile edge 17
constexpr int NPOS = CT * CT;         // 289
constexpr int NFRAG = (NPOS + 15) / 16;  // 19
constexpr int NWV = 8;
constexpr int NTH = 64 * NWV;
constexpr int FPW = (NFRAG + NWV - 1) / NWV;  // 3
constexpr int IT = 2 * (CT - 1) + 7;  // 39
constexpr int PWP = 40;
constexpr int PATCH_BYTES = IT * PWP * 8;      // 12480
constexpr int WROW_BYTES = 480;                // stem_mfma.hip: the conflict-free row stride of the packed weights
constexpr int W_BYTES = 64 * WROW_BYTES;       // 30 KiB
constexpr int CT_BYTES = (NPOS + 3) * 128;     // [pos][32 channels] fp32

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ f32x4_t mfma_f16_16x16x32(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    const h2_t v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}

__global__ __launch_bounds__(NTH) void stem_split16_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wh_pk,
                                                           const unsigned char* __restrict__ wl_pk, const float* __restrict__ bias,
                                                           float* __restrict__ out, float alpha, int H, int W, int CH, int CW, int PH,
                                                           int PW, int tiles_w, int tiles_hw, int ntiles) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PATCH_BYTES + CT_BYTES + 2 * W_BYTES + 256];
    unsigned char* s_ph = smem;
    unsigned char* s_pl = smem + PATCH_BYTES;
    unsigned char* s_ct = smem + 2 * PATCH_BYTES;
    unsigned char* s_wh = s_ct + CT_BYTES;
    unsigned char* s_wl = s_wh + W_BYTES;
    float* s_bias = reinterpret_cast<float*>(s_wl + W_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;

    for (int piece = wave; piece < W_BYTES / 1024; piece += NWV) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wh_pk + piece * 1024 + lane * 16), (lds_void_t*)(s_wh + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wl_pk + piece * 1024 + lane * 16), (lds_void_t*)(s_wl + piece * 1024), 16, 0, 0);
    }
    if (tid < 64) s_bias[tid] = bias[tid];

    constexpr int NPASS = (IT * PWP + NTH - 1) / NTH;  // 4
    float pv[NPASS][3];
    auto load_patch = [&](int T) {
        const int n = T / tiles_hw;
        const int trem = T - n * tiles_hw;
        const int ph0 = (trem / tiles_w) * PT, pw0 = (trem % tiles_w) * PT;
        const int iy0 = 2 * (2 * ph0 - 1) - 3, ix0 = 2 * (2 * pw0 - 1) - 3;
        const float* xn = x + (size_t)n * 3 * H * W;
        int td = tid;
        asm volatile("" : "+v"(td));
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int e = td + NTH * i;
            const int py = e / PWP, px = e - py * PWP;
            const int iy = iy0 + py, ix = ix0 + px;
            pv[i][0] = pv[i][1] = pv[i][2] = 0.f;
            if (e < IT * PWP && px < IT && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const size_t o = (size_t)iy * W + ix;
                pv[i][0] = xn[o];
                pv[i][1] = xn[(size_t)H * W + o];
                pv[i][2] = xn[2 * (size_t)H * W + o];
            }
        }
    };
    const int frow = lane & 15, g = lane >> 4;
    int a_off[FPW];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        int pos = (wave + NWV * i) * 16 + frow;
        pos = pos < NPOS ? pos : NPOS - 1;
        const int cy = pos / CT, cx = pos - cy * CT;
        a_off[i] = ((2 * cy) * PWP + 2 * cx + 2 * g) * 8;
    }
    auto ct_row = [](int pos) { return (pos & ~3) | ((pos & 1) << 1) | ((pos >> 1) & 1); };

    int q = blockIdx.x;
    if (q < ntiles) load_patch(q);
    for (; q < ntiles; q += G) {
        const int T = q;
        const int n = T / tiles_hw;
        const int trem = T - n * tiles_hw;
        const int ph0 = (trem / tiles_w) * PT, pw0 = (trem % tiles_w) * PT;
        const int cr0 = 2 * ph0 - 1, cc0 = 2 * pw0 - 1;
        // ---- this tile's pixels (requested one tile ago) -> fp16 hi / lo patches in LDS
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int e = tid + NTH * i;
            if (e < IT * PWP) {
                typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
                const h2_t h01 = {(_Float16)pv[i][0], (_Float16)pv[i][1]};
                const _Float16 h2 = (_Float16)pv[i][2];
                uint2 uh, ul;
                uh.x = __builtin_bit_cast(uint32_t, h01);
                uh.y = (uint32_t)__builtin_bit_cast(unsigned short, h2);
                ul.x = pack_f16x2(pv[i][0] - (float)h01[0], pv[i][1] - (float)h01[1]);
                ul.y = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)(pv[i][2] - (float)h2));
                *reinterpret_cast<uint2*>(s_ph + e * 8) = uh;
                *reinterpret_cast<uint2*>(s_pl + e * 8) = ul;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: the weight DMA (invisible to the compiler) has landed
        __syncthreads();
        if (q + G < ntiles) load_patch(q + G);  // in flight until the top of the next iteration

        const bool interior = cr0 >= 0 && cc0 >= 0 && cr0 + CT <= CH && cc0 + CT <= CW;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            f32x4_t acc[FPW][2];
#pragma unroll
            for (int i = 0; i < FPW; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[i][a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                uint4 wh[2], wl[2], xh[FPW], xl[FPW];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int o = ((2 * half + a) * 16 + frow) * WROW_BYTES + r * 64 + g * 16;
                    wh[a] = *reinterpret_cast<const uint4*>(s_wh + o);
                    wl[a] = *reinterpret_cast<const uint4*>(s_wl + o);
                }
#pragma unroll
                for (int i = 0; i < FPW; ++i) {
                    xh[i] = *reinterpret_cast<const uint4*>(s_ph + a_off[i] + r * (PWP * 8));
                    xl[i] = *reinterpret_cast<const uint4*>(s_pl + a_off[i] + r * (PWP * 8));
                }
#pragma unroll
                for (int i = 0; i < FPW; ++i)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        acc[i][a] = mfma_f16_16x16x32(wh[a], xl[i], acc[i][a]);
                        acc[i][a] = mfma_f16_16x16x32(wl[a], xh[i], acc[i][a]);
                        acc[i][a] = mfma_f16_16x16x32(wh[a], xh[i], acc[i][a]);
                    }
            }
            if (half == 1) __syncthreads();  // every thread has pooled half 0: the conv tile may be rewritten
            int fr = frow, gg = g, tq = tid;
            asm volatile("" : "+v"(fr), "+v"(gg), "+v"(tq));
            // conv tile [pos][32 ch] fp32: 16-byte slot s (4 channels) of row p at slot s ^ (p & 7), row p at row index ct_row(p)
            // (stem_mfma.hip's layout with fp32 x 32 channels in place of fp16 x 64: the same 128-byte rows)
#pragma unroll
            for (int i = 0; i < FPW; ++i) {
                const int pos = (wave + NWV * i) * 16 + fr;
                if (pos < NPOS) {
                    const int cy = pos / CT, cx = pos - cy * CT;
                    const bool in = interior || ((unsigned)(cr0 + cy) < (unsigned)CH && (unsigned)(cc0 + cx) < (unsigned)CW);
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int chl = a * 16 + gg * 4;   // channel inside the half
                        const float4 bv = *reinterpret_cast<const float4*>(s_bias + half * 32 + chl);
                        float4 v = make_float4(relu_nan(fmaf(alpha, acc[i][a][0], bv.x)), relu_nan(fmaf(alpha, acc[i][a][1], bv.y)),
                                               relu_nan(fmaf(alpha, acc[i][a][2], bv.z)), relu_nan(fmaf(alpha, acc[i][a][3], bv.w)));
                        if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);   // positions outside the conv map count as 0 in the pool (post-ReLU values are >= 0)
                        *reinterpret_cast<float4*>(s_ct + ct_row(pos) * 128 + (((chl >> 2) ^ (pos & 7)) << 4)) = v;
                    }
                }
            }
            __syncthreads();
            // 3x3/2 max pool: thread -> 4 channels (one 16-byte slot) of one pooled pixel
            {
                const int cq = tq & 7;
                const int pp = tq >> 3;
                const int py = pp / PT, px = pp - py * PT;
                const int ph = ph0 + py, pw = pw0 + px;
                if (ph < PH && pw < PW) {
                    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int pos = (2 * py + dy) * CT + 2 * px + dx;
                            const float4 u = *reinterpret_cast<const float4*>(s_ct + ct_row(pos) * 128 + ((cq ^ (pos & 7)) << 4));
                            // (NaN-propagating like the ReLU above: an out-of-range activation must reach the embedding)
                            m.x = __builtin_elementwise_maximum(m.x, u.x); m.y = __builtin_elementwise_maximum(m.y, u.y);
                            m.z = __builtin_elementwise_maximum(m.z, u.z); m.w = __builtin_elementwise_maximum(m.w, u.w);
                        }
                    *reinterpret_cast<float4*>(out + (((size_t)n * PH + ph) * PW + pw) * 64 + half * 32 + cq * 4) = m;
                }
            }
        }
        __syncthreads();  // the conv tile and the patches are consumed: the next tile may overwrite them
    }
}
}  // namespace

extern "C" int agrl_stem_split16(const float* x, const void* wh_packed, const void* wl_packed, const float* bias, float* out, int N,
                                 int H, int W, float w_unscale, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && wh_packed && wl_packed && bias && out, "agrl_stem_split16: null pointer");
    AGRL_CHECK_ARG(N > 0 && H >= 7 && W >= 7, "agrl_stem_split16: bad shape N=%d H=%d W=%d", N, H, W);
    AGRL_CHECK_ARG(((((uintptr_t)wh_packed) | ((uintptr_t)wl_packed) | ((uintptr_t)bias) | ((uintptr_t)out)) & 15) == 0, "agrl_stem_split16: misaligned pointer");
    AGRL_CHECK_ARG(w_unscale > 0.f && w_unscale <= 3.4e38f, "agrl_stem_split16: w_unscale must be a positive finite power of two");
    {
        int e = 0;
        AGRL_CHECK_ARG(frexpf(w_unscale, &e) == 0.5f, "agrl_stem_split16: w_unscale=%g is not a power of two", (double)w_unscale);
    }
    const int CH = (H + 6 - 7) / 2 + 1, CW = (W + 6 - 7) / 2 + 1;
    const int PH = (CH + 2 - 3) / 2 + 1, PW = (CW + 2 - 3) / 2 + 1;
    const int tiles_h = cdiv(PH, PT), tiles_w = cdiv(PW, PT);
    const long long grid = (long long)N * tiles_h * tiles_w;
    AGRL_CHECK_ARG(grid < (1ll << 31), "agrl_stem_split16: grid too large");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const unsigned launch = (unsigned)(grid < cus ? grid : cus);   // one persistent workgroup per CU (123 KB of LDS)
    hipLaunchKernelGGL(stem_split16_kernel, dim3(launch), dim3(NTH), 0, (hipStream_t)stream, x, (const unsigned char*)wh_packed,
                       (const unsigned char*)wl_packed, bias, out, w_unscale, H, W, CH, CW, PH, PW, tiles_w, tiles_h * tiles_w, (int)grid);
    AGRL_CHECK_LAUNCH("agrl_stem_split16");
    return 0;
}
