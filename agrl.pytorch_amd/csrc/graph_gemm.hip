// The Linear of a GraphLayer in its commuted form, out = keep f + gamma lrelu(bn(P W^T)) with P = G f (vmgn.py:148,
// :168-172), as a kernel shaped for THIS problem: few rows (M = tracklets x V = 1792 at the bench size) against a
// 2048 x 2048 weight matrix, bf16 operands, fp32 result.
//
// Why not igemm_kernel: M x N / 256 CUs = 14 336 outputs per CU, so a CU owns about one 128 x 128 tile. A wave whose tile is
// FM x FN fragments reads FM + FN fragments (1 KiB each) from LDS per FM x FN MFMAs: igemm_kernel's 16 x 64 wave tile (FM 1,
// FN 4) reads five fragments per four MFMAs, its 32 x 64 one (8 waves on 128 x 128) six per eight; a 64 x 64 wave tile
// (FM = FN = 4) eight per sixteen. Four waves of 64 x 64 cover the 128 x 128 tile but leave ONE wave per SIMD, and the k loop
// then shows every LDS round trip (measured: 43-48 us against 28-32 us for the 8-wave forms).
//
// So: 8 waves = 2 x 2 wave tiles of 64 x 64 TIMES 2 k-halves. The two waves of a SIMD (w, w + 4) own the same 64 x 64
// outputs and split every 64-deep k-tile between them (k-step 0 / k-step 1), so the LDS is read as by four waves
// (64 KiB per 32 KiB k-tile) while every SIMD still has two waves to interleave; the partial sums meet once, in the
// epilogue, through the then idle ring (each wave hands over the half of its fragments it does not finalise).
// Ring of four 32 KiB k-tiles (LDS-DMA, counted vmcnt, one raw barrier per k-tile); one workgroup per CU; every XCD owns a
// range of N-tiles so its slice of W stays in its L2.
//
// Measured at 32 tracklets (1792 x 2048 x 2048, rocprofv3): 27.9 us against 31.7 us for igemm_kernel's 64 x 128 tiles (26.8 us
// inside a layer). Where the time goes (compile-time ablations of this kernel): 5.6 us launch + first k-tiles + the exchange
// with a one-k-tile loop and no epilogue memory; ~6 us the epilogue's 29 MB of f reads and result writes (HBM-bandwidth, not
// latency: requesting f twelve k-tiles early made the kernel SLOWER -- loads retire in order, so the ring behind them starved
// for an HBM round trip); ~17 us the 32 k-tiles, of which the steady-state DMA is 6.2, the fragment reads 3.7, the barrier
// 2.4 and the MFMAs nothing. Collapsing both operands onto 128 L2-resident rows changes nothing (the operand fetch is not
// the limit). Counters (profiles/r03_pmc_graph_gemm.txt): matrix pipe busy 23 % of the launch, LDS index-active 15 %, 43 % of
// the wave cycles parked at s_waitcnt / the barrier -- a latency chain per k-tile (wait, barrier, eight reads, sixteen MFMAs)
// that one workgroup per CU cannot hide; double-buffering the fragment registers across k-tiles did not help (29.5 us).
#include "igemm_dev.h"

namespace {

constexpr int GBM = 128, GBN = 128, GNS = 4;
constexpr int G_A_BYTES = GBM * 128, G_B_BYTES = GBN * 128, G_BUF = G_A_BYTES + G_B_BYTES;  // 32 KiB per ring slot
constexpr int G_DPT = (GBM + GBN) / 64;  // 1-KiB DMA pieces per wave and k-tile (8 waves): 2 pixel-row + 2 weight-row

struct GraphGemmParams {
    const unsigned char* x;   // (M, K) bf16
    const unsigned char* w;   // (N, K) bf16
    const float* f;           // (M, N) fp32
    const float* scale;       // (N)
    const float* shift;       // (N)
    float* out;               // (M, N) fp32
    float keep, gamma, slope;
    int M, N, K;
};

// Wave (ws, KH) finalises channel fragments 2 KH, 2 KH + 1 of its 64 x 64 wave tile and hands the other two to its partner
// (ws, 1 - KH) through the idle ring. Slot layout: [receiver][ws][fragment (a & 1) * 4 + b][lane] x 16 bytes (lane-contiguous:
// conflict-free). Then BatchNorm1d (folded) + LeakyReLU + the residual mix, 16-byte loads of f and stores of out.
template <int KH>
__device__ inline void graph_linear_epilogue(const GraphGemmParams& p, f32x4_t (&acc)[4][4], unsigned char* smem, int ws, int lane,
                                             int wm0, int wn0) {
    constexpr int OTH = 1 - KH;
    uint4* xch = reinterpret_cast<uint4*>(smem);
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < 4; ++b) xch[(((OTH * 4 + ws) * 8) + a2 * 4 + b) * 64 + lane] = __builtin_bit_cast(uint4, acc[2 * OTH + a2][b]);
    __syncthreads();
    const int frow = lane & 15, fchunk = lane >> 4;
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
        constexpr int A0 = 2 * KH;
        const int gn = wn0 + (A0 + a2) * 16 + fchunk * 4;  // this lane's 4 channels
        const float4 s4 = *reinterpret_cast<const float4*>(p.scale + gn), c4 = *reinterpret_cast<const float4*>(p.shift + gn);
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gm = wm0 + b * 16 + frow;
            if (gm >= p.M) continue;
            const f32x4_t other = __builtin_bit_cast(f32x4_t, xch[(((KH * 4 + ws) * 8) + a2 * 4 + b) * 64 + lane]);
            const size_t o = (size_t)gm * p.N + gn;
            const float4 f4 = *reinterpret_cast<const float4*>(p.f + o);
            const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = fmaf(acc[A0 + a2][b][r] + other[r], sv[r], cv[r]);  // (k-step 0 sum) + (k-step 1 sum): one fp32 add, order-free
                y = y > 0.f ? y : p.slope * y;
                v[r] = p.keep * fv[r] + p.gamma * y;
            }
            *reinterpret_cast<float4*>(p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

__global__ __launch_bounds__(512) void graph_linear_kernel(const GraphGemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[GNS * G_BUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wave & 3;        // spatial wave: 2 x 2 grid of 64 x 64
    const int kh = wave >> 2;       // k-half: the k-step of every k-tile this wave multiplies
    const int wm = ws & 1, wn = ws >> 1;

    // XCD map (workgroups b, b + 8, .. share an XCD): tile ids are N-tile major, every XCD gets a contiguous range
    const int nMt = (p.M + GBM - 1) / GBM;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, within = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int nt = bid / nMt, mt = bid - nt * nMt;
    const int m0 = mt * GBM, n0 = nt * GBN;

    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
    // DMA piece j of this wave: tile rows wave*16 + 8j .. +7 of the pixel tile (j < 2) / the weight tile (j >= 2);
    // lane L -> row + (L >> 3), physical chunk L & 7 = global chunk (L & 7) ^ ((row >> 1) & 7)
    const int lrow = lane >> 3, lchk = lane & 7;
    const size_t row_bytes = (size_t)p.K * 2;
    const unsigned char* src[G_DPT];
    bool src_ok[G_DPT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave * 16 + j * 8 + lrow;
        const int gm = m0 + row;
        src_ok[j] = gm < p.M;
        src[j] = p.x + (size_t)(src_ok[j] ? gm : 0) * row_bytes + ((lchk ^ ((row >> 1) & 7)) << 4);
        src_ok[2 + j] = true;
        src[2 + j] = p.w + (size_t)(n0 + row) * row_bytes + ((lchk ^ ((row >> 1) & 7)) << 4);
    }
    size_t kbyte = 0;
    auto stage_piece = [&](int buf, int j) {
        unsigned char* dst = smem + buf * G_BUF + (j < 2 ? 0 : G_A_BYTES) + (wave * 16 + (j & 1) * 8) * 128;
        dma16(src_ok[j] ? src[j] + kbyte : zsrc, dst);
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K >> 6;
#pragma unroll
    for (int s = 0; s < GNS - 1; ++s)
        if (s < nk) {
#pragma unroll
            for (int j = 0; j < G_DPT; ++j) stage_piece(s, j);
            kbyte += 128;
        }
    const int frow = lane & 15, fchunk = lane >> 4;
    const int xrow = wm * 64 + frow, wrow = wn * 64 + frow, kchunk = kh * 4 + fchunk;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = min(GNS - 2, nk - 1 - kt);  // k-tiles requested after tile kt so far
        if (younger == 2) wait_vmcnt<2 * G_DPT>();
        else if (younger == 1) wait_vmcnt<G_DPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int fill = cur + GNS - 1;  // the slot read in iteration kt - 1
        fill = fill >= GNS ? fill - GNS : fill;
        const bool do_stage = kt + GNS - 1 < nk;
        const unsigned char* sa = smem + cur * G_BUF;
        const unsigned char* sb = sa + G_A_BYTES;
        // (fragment registers double-buffered across k-tiles -- the reads of k-tile kt + 1 under the MFMAs of k-tile kt --
        // measured 29.5 us against 27.9 us for this form: with two waves per SIMD the hardware already interleaves them)
        uint4 xf[4], wf[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) xf[b] = *reinterpret_cast<const uint4*>(sa + lds_off(xrow + b * 16, kchunk));
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const uint4*>(sb + lds_off(wrow + a * 16, kchunk));
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (do_stage) stage_piece(fill, a);  // one piece of the k-tile three ahead rides with each group of MFMAs
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = Frag<lp16_t>::mma(wf[a], xf[b], acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_stage) kbyte += 128;
        cur = cur + 1 == GNS ? 0 : cur + 1;
    }
    wait_vmcnt<0>();
    __syncthreads();  // everybody is done reading the ring

    // ---- the two k-halves meet (compile-time fragment indices per half: a runtime index into acc would go through scratch)
    if (kh == 0) graph_linear_epilogue<0>(p, acc, smem, ws, lane, m0 + wm * 64, n0 + wn * 64);
    else graph_linear_epilogue<1>(p, acc, smem, ws, lane, m0 + wm * 64, n0 + wn * 64);
}

}  // namespace

static int graph_gemm_cus() {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return n_cu;
}

// One round of tiles only: with more tiles than CUs the generic kernel's two workgroups per CU (one's epilogue under the
// other's k loop) are as fast or faster (measured at 64 / 256 tracklets: 47.9 / 187 us against 47.9 / 200 us for a
// two-slot form of this kernel), at one round this kernel is (27.9 against 31.7 us at 32 tracklets).
bool graph_gemm_applicable(int M, int K, int Nout) {
    return M > 0 && (K % 64) == 0 && K >= 64 && (Nout % GBN) == 0 && ((M + GBM - 1) / GBM) * (Nout / GBN) <= graph_gemm_cus();
}

int launch_graph_gemm(const void* p_op, const void* w, const float* f, const float* bn_scale, const float* bn_shift, float keep,
                      float gamma, float slope, float* out, int M, int K, int Nout, hipStream_t stream) {
    GraphGemmParams p;
    p.x = reinterpret_cast<const unsigned char*>(p_op);
    p.w = reinterpret_cast<const unsigned char*>(w);
    p.f = f; p.scale = bn_scale; p.shift = bn_shift; p.out = out;
    p.keep = keep; p.gamma = gamma; p.slope = slope;
    p.M = M; p.N = Nout; p.K = K;
    const int grid = ((M + GBM - 1) / GBM) * (Nout / GBN);
    hipLaunchKernelGGL(graph_linear_kernel, dim3(grid), dim3(512), 0, stream, p);
    AGRL_CHECK_LAUNCH("agrl_graph_linear_mix");
    return 0;
}
