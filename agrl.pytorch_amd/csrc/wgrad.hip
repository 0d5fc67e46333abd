// Weight gradient of a conv layer straight from the NHWC activations (train step, BASELINE config 4; what autograd runs for
// loss.backward(), reference train_vidreid_xent_htri.py:411, through every nn.Conv2d of torchreid/models/vmgn.py:45-65):
//
//     dW[co][ci][r][s] = sum over pixels p = (f, oy, ox) of  dy[p][co] * x[f][oy * stride + r - pad][ox * stride + s - pad][ci]
//
// A GEMM whose contraction axis is the PIXEL axis -- the slow axis of both NHWC operands. The first version transposed both
// operands into channel-major copies (agrl_im2col_t, nine-fold for a 3x3) and ran the K-contiguous GEMM on them: 24 ms of
// transposes per config-4 step beside 31 ms of GEMM. Here the transpose happens on the way into the LDS: a thread loads a
// 4 pixel x 4 channel micro-tile (four 16-byte loads, 128-byte runs per pixel row across 8 lanes), and writes it as four
// 16-byte chunks "4 consecutive pixels of one channel" -- exactly the k-chunk the MFMA fragments of igemm_dev.h read (128-byte
// swizzled LDS rows, one row per channel, 32 pixels per k-tile). The 3x3 taps are a shifted row pointer per pixel (zero rows
// outside the frame), so nothing is expanded in memory. Exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) or the split-bf16 recipe
// (AGRL_F32X3), the same two arithmetic modes as the forward / data-gradient GEMMs.
//
// Few output tiles, 10^4..10^6 pixels: the pixel axis is split over workgroups (blockIdx.y), fp32 partials per slice in the
// caller's workspace, summed in slice order by wgrad_reduce_kernel (deterministic), which also writes the OIHW layout of
// nn.Conv2d.weight.grad.
#include "agrl_common.h"
#include "igemm_dev.h"

namespace {

struct WgradParams {
    const float* x;
    const float* dy;
    float* ws;     // [ks][Cout][taps * Cin]
    int F, H, W, Cin, Cout, OH, OW, R, S, stride, pad;
    int Mtot;      // F * OH * OW pixels
    int m_tiles, n_tiles;
    int nk, cps;   // 32-pixel k-tiles in total / per slice
    int pointwise; // 1x1, stride 1, no padding: the x row of pixel p is x + p * Cin
};

__device__ inline float comp(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

template <typename TIN, int BM, int BN>
__global__ __launch_bounds__(512) void wgrad_tn_kernel(const WgradParams p) {
    // 8 waves as 4 (co) x 2 (ci); wave tile (BM / 4) x (BN / 2)
    constexpr int FM = BM / 64, FN = BN / 32;   // 16-channel fragments per wave: co (MFMA columns), ci (MFMA rows)
    // ONE k-tile in the LDS (32 KB at 128 x 128). A second buffer (one barrier per k-tile instead of two) was measured with the
    // 4-wave form: one workgroup less per CU, 11.1 -> 13.8 ms over the model's shapes -- occupancy is worth more here.
    __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * 128];
    unsigned char* sA = smem;              // dy tile: BM channel rows x 32 pixels
    unsigned char* sB = smem + BM * 128;   // x tile:  BN channel rows x 32 pixels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    int bx = blockIdx.x;
    const int tm = bx % p.m_tiles;
    bx /= p.m_tiles;
    const int tn = bx % p.n_tiles;
    const int tap = bx / p.n_tiles;
    const int r = tap / p.S, s = tap - r * p.S;
    const int co0 = tm * BM, ci0 = tn * BN;
    const int k0 = blockIdx.y * p.cps, k1 = min(p.nk, k0 + p.cps);

    // loader: waves 0-3 stage the dy tile, waves 4-7 the x tile; a thread owns pixel quad q of the k-tile and channel quad cq
    const bool ldx = wave >= 4;                       // wave-uniform
    const int q = lane & 7, cq = (wave & 3) * 8 + (lane >> 3);
    const bool in_t = 4 * cq < (ldx ? BN : BM);
    const bool ld_ok = in_t && (ldx ? ci0 + 4 * cq < p.Cin : co0 + 4 * cq < p.Cout);
    const float* src = ldx ? p.x + ci0 + 4 * cq : p.dy + co0 + 4 * cq;
    float4 rg[4];
    auto load = [&](int kt) {
        const int pb = kt * 32 + 4 * q;
        int ox = 0, oy = 0, f = 0;
        if (ldx && !p.pointwise) {
            ox = pb % p.OW;
            const int t = pb / p.OW;
            oy = t % p.OH;
            f = t / p.OH;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int pix = pb + rr;
            rg[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < p.Mtot && ld_ok) {
                if (!ldx) {
                    rg[rr] = *reinterpret_cast<const float4*>(src + (size_t)pix * p.Cout);
                } else if (p.pointwise) {
                    rg[rr] = *reinterpret_cast<const float4*>(src + (size_t)pix * p.Cin);
                } else {
                    const int iy = oy * p.stride + r - p.pad, ix = ox * p.stride + s - p.pad;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                        rg[rr] = *reinterpret_cast<const float4*>(src + ((size_t)(f * p.H + iy) * p.W + ix) * p.Cin);
                }
            }
            if (ldx && !p.pointwise && ++ox == p.OW) {
                ox = 0;
                if (++oy == p.OH) { oy = 0; ++f; }
            }
        }
    };
    auto stash = [&]() {
        if (in_t) {
            unsigned char* dst = ldx ? sB : sA;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(dst + lds_off(4 * cq + j, q)) = make_float4(comp(rg[0], j), comp(rg[1], j), comp(rg[2], j), comp(rg[3], j));
        }
    };

    const int frow = lane & 15, fchunk = lane >> 4;
    f32x4_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (k0 < k1) load(k0);
    for (int kt = k0; kt < k1; ++kt) {
        __syncthreads();   // the previous k-tile's fragment reads are done
        stash();
        __syncthreads();
        if (kt + 1 < k1) load(kt + 1);   // lands under this k-tile's matrix work
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 xf[FN], df[FM];
#pragma unroll
            for (int a = 0; a < FN; ++a) xf[a] = *reinterpret_cast<const uint4*>(sB + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int b = 0; b < FM; ++b) df[b] = *reinterpret_cast<const uint4*>(sA + lds_off(wm * (BM / 4) + b * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] = Frag<TIN>::mma(xf[a], df[b], acc[a][b]);
        }
    }
    // lane: 4 consecutive input channels (MFMA rows 4 * fchunk + j) of one output channel (MFMA column frow)
    const int ncol = p.R * p.S * p.Cin;
    float* out = p.ws + (size_t)blockIdx.y * p.Cout * ncol + (size_t)tap * p.Cin;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int co = co0 + wm * (BM / 4) + b * 16 + frow;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int ci = ci0 + wn * (BN / 2) + a * 16 + 4 * fchunk;
            if (ci < p.Cin)
                *reinterpret_cast<float4*>(out + (size_t)co * ncol + ci) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
    }
}

// dw[co][ci][tap] = sum_z ws[z][co][tap][ci]; 16 float4 columns x 16 slice lanes per block (a lane adds every 16th slice, four
// loads in flight), lane sums added in lane order: the summation order depends on the slice count only
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int ks, float* __restrict__ dw, int Cout, int Cin,
                                                           int taps) {
    __shared__ float4 s_p[16][17];
    const int col = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const size_t total4 = (size_t)Cout * taps * (Cin >> 2);
    const size_t e4 = (size_t)blockIdx.x * 16 + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 < total4) {
        const float4* src = reinterpret_cast<const float4*>(ws) + e4;
        int z = zl;
        for (; z + 48 < ks; z += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(z + 16 * u) * total4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; z < ks; z += 16) {
            const float4 v = src[(size_t)z * total4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    s_p[zl][col] = acc;
    __syncthreads();
    if (zl != 0 || e4 >= total4) return;
    float4 t = s_p[0][col];
#pragma unroll
    for (int k = 1; k < 16; ++k) { t.x += s_p[k][col].x; t.y += s_p[k][col].y; t.z += s_p[k][col].z; t.w += s_p[k][col].w; }
    const int c4n = Cin >> 2;
    const int ci = 4 * (int)(e4 % c4n);
    const size_t rest = e4 / c4n;
    const int tap = (int)(rest % taps);
    const int co = (int)(rest / taps);
    float* o = dw + ((size_t)co * Cin + ci) * taps + tap;
    if (taps == 1) {
        *reinterpret_cast<float4*>(o) = t;
    } else {
        o[0] = t.x; o[(size_t)taps] = t.y; o[2 * (size_t)taps] = t.z; o[3 * (size_t)taps] = t.w;
    }
}

struct WgradPlan {
    int bm, bn, m_tiles, n_tiles, nk, ks, cps;
};

// Workgroups of one kernel variant the device holds at once (occupancy x compute units), queried once per variant. The slices
// are equally long, so a grid of exactly one resident round has no tail: 144 tiles x 6 slices = 864 workgroups on 768 slots ran
// two rounds (56 % of the slots busy on average), 144 x 5 = 720 runs one.
template <typename TIN, int BM, int BN>
static int wgrad_capacity_of() {
    static int cap = 0;
    if (cap == 0) {
        int dev = 0, cus = 256, per_cu = 2;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_tn_kernel<TIN, BM, BN>, 512, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        (void)hipGetLastError();
        cap = per_cu * (cus > 0 ? cus : 256);
    }
    return cap;
}

static int wgrad_capacity(int dtype, int bm, int bn) {
    if (dtype == AGRL_F32) {
        if (bm == 128 && bn == 128) return wgrad_capacity_of<float, 128, 128>();
        if (bm == 128) return wgrad_capacity_of<float, 128, 64>();
        if (bn == 128) return wgrad_capacity_of<float, 64, 128>();
        return wgrad_capacity_of<float, 64, 64>();
    }
    if (bm == 128 && bn == 128) return wgrad_capacity_of<f32s_t, 128, 128>();
    if (bm == 128) return wgrad_capacity_of<f32s_t, 128, 64>();
    if (bn == 128) return wgrad_capacity_of<f32s_t, 64, 128>();
    return wgrad_capacity_of<f32s_t, 64, 64>();
}

static WgradPlan wgrad_plan(int Mtot, int Cin, int Cout, int taps, int dtype) {
    WgradPlan pl;
    pl.bm = Cout >= 128 ? 128 : 64;
    pl.bn = Cin >= 128 ? 128 : 64;
    if (cdiv(Cout, pl.bm) * cdiv(Cin, pl.bn) * taps < 8) pl.bm = pl.bn = 64;   // tiny weights: more tiles, shorter slice list
    pl.m_tiles = cdiv(Cout, pl.bm);
    pl.n_tiles = cdiv(Cin, pl.bn);
    pl.nk = cdiv(Mtot, 32);
    const int tiles = pl.m_tiles * pl.n_tiles * taps;
    const int target = wgrad_capacity(dtype, pl.bm, pl.bn);
    // slice count: fill whole resident rounds (at most three) as evenly as the tile count allows -- 144 tiles on 512 slots: 3 slices
    // fill 84 % of one round, 7 slices 98 % of two
    int ks_max = pl.nk / 4;               // at least four k-tiles per slice
    if (ks_max > 1024) ks_max = 1024;
    if (ks_max < 1) ks_max = 1;
    int ks = 1;
    double best = -1.0;
    for (int c = 1; c <= ks_max; ++c) {
        const long wgs = (long)tiles * c;
        const long rounds = (wgs + target - 1) / target;
        if (rounds > 3) break;
        const double fill = (double)wgs / (double)(rounds * target);
        if (fill > best + 0.02) { best = fill; ks = c; }
    }
    pl.cps = cdiv(pl.nk, ks);
    pl.ks = cdiv(pl.nk, pl.cps);
    return pl;
}

template <typename TIN>
static void launch_wgrad(const WgradParams& p, const WgradPlan& pl, int taps, hipStream_t st) {
    const dim3 grid(pl.m_tiles * pl.n_tiles * taps, pl.ks);
    if (pl.bm == 128 && pl.bn == 128) hipLaunchKernelGGL((wgrad_tn_kernel<TIN, 128, 128>), grid, dim3(512), 0, st, p);
    else if (pl.bm == 128) hipLaunchKernelGGL((wgrad_tn_kernel<TIN, 128, 64>), grid, dim3(512), 0, st, p);
    else if (pl.bn == 128) hipLaunchKernelGGL((wgrad_tn_kernel<TIN, 64, 128>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((wgrad_tn_kernel<TIN, 64, 64>), grid, dim3(512), 0, st, p);
}

}  // namespace

extern "C" size_t agrl_conv_wgrad_workspace(int F, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
    if (F <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return 0;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    if (OH <= 0 || OW <= 0) return 0;
    const WgradPlan p0 = wgrad_plan(F * OH * OW, Cin, Cout, R * S, AGRL_F32), p1 = wgrad_plan(F * OH * OW, Cin, Cout, R * S, AGRL_F32X3);
    return (size_t)(p0.ks > p1.ks ? p0.ks : p1.ks) * Cout * R * S * Cin * sizeof(float);   // enough for either arithmetic mode
}

extern "C" int agrl_conv_wgrad(const float* x, const float* dy, float* dw, int F, int H, int W, int Cin, int Cout, int R, int S,
                               int stride, int pad, int dtype, void* workspace, size_t workspace_bytes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && dy && dw && workspace, "agrl_conv_wgrad: null pointer");
    AGRL_CHECK_ARG(F > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "agrl_conv_wgrad: bad shape");
    AGRL_CHECK_ARG((Cin % 4) == 0 && (Cout % 4) == 0, "agrl_conv_wgrad: Cin=%d and Cout=%d must be multiples of 4", Cin, Cout);
    AGRL_CHECK_ARG(dtype == AGRL_F32 || dtype == AGRL_F32X3, "agrl_conv_wgrad: dtype must be fp32 (0) or split-bf16 fp32 (2), got %d", dtype);
    const uintptr_t al = (uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace;
    AGRL_CHECK_ARG((al & 15) == 0, "agrl_conv_wgrad: operands must be 16-byte aligned");
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    AGRL_CHECK_ARG(OH > 0 && OW > 0, "agrl_conv_wgrad: empty output");
    AGRL_CHECK_ARG((long long)F * OH * OW < (1ll << 31) - 64, "agrl_conv_wgrad: too many pixels");
    const int taps = R * S;
    const WgradPlan pl = wgrad_plan(F * OH * OW, Cin, Cout, taps, dtype);
    AGRL_CHECK_ARG(workspace_bytes >= (size_t)pl.ks * Cout * taps * Cin * sizeof(float), "agrl_conv_wgrad: workspace too small (agrl_conv_wgrad_workspace)");
    WgradParams p;
    p.x = x; p.dy = dy; p.ws = (float*)workspace;
    p.F = F; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.OH = OH; p.OW = OW; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.Mtot = F * OH * OW;
    p.m_tiles = pl.m_tiles; p.n_tiles = pl.n_tiles; p.nk = pl.nk; p.cps = pl.cps;
    p.pointwise = (R == 1 && S == 1 && stride == 1 && pad == 0) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AGRL_F32) launch_wgrad<float>(p, pl, taps, st);
    else launch_wgrad<f32s_t>(p, pl, taps, st);
    AGRL_CHECK_LAUNCH("agrl_conv_wgrad");
    const size_t total4 = (size_t)Cout * taps * (Cin / 4);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total4 + 15) / 16)), dim3(256), 0, st, (const float*)workspace, pl.ks, dw, Cout, Cin,
                       taps);
    AGRL_CHECK_LAUNCH("agrl_conv_wgrad(reduce)");
    return 0;
}
