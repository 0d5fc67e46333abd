// 3x3 / stride 1 / pad 1 convolution, bf16, with TWO 16x8 pixel blocks (two whole frames in layers 3/4) per
// workgroup: 256 output pixels x 128 output channels.
//
// The one-block kernel (conv3x3_patch_kernel, igemm.hip) stages per 64-channel slab one 23 KB halo patch and
// 9 x 16 KB of weights for 9 x 515 MFMA cycles: the weight stream is as expensive as the matrix work it feeds, and a
// step between two barriers is only 16 MFMAs per wave. Doubling the pixel tile halves the weight bytes per flop
// (2 x 23 + 9 x 16 KB per 9 x 1030 MFMA cycles = 20 B/clk through the CU's 64 B/clk vector-memory path), doubles the
// matrix work between barriers and widens the wave tile to 64 x 64 (8 fragment reads per 16 MFMAs instead of 6 per 8).
//
//   * 512 threads = 8 waves as 4 (pixels) x 2 (channels); wave tile 64 pixels x 64 channels = 4 x 4 MFMA 16x16x32
//   * LDS: halo patches 2 (slab ring) x 2 (blocks) x 23.5 KB + weight ring 3 x 16 KB = 142 KB; one workgroup per CU.
//     Weights are requested TWO tap-steps ahead with counted vmcnt waits (a step is ~0.45 us, an L2 round trip under
//     load is about that long); the next slab's patches ride one piece per tap-step.
//   * fragment reads software-pipelined inside the step (weight fragment two MFMA groups ahead, the second k-step's
//     pixel fragments under the first's MFMAs)
//   * epilogue: bias + ReLU in fp32, bf16 tile parked over the patch buffers, written out as whole 16-byte chunks
#include <stdlib.h>

#include "igemm_dev.h"

namespace {

// BN_ = 256 (Cout % 256 == 0 and enough tiles: the 512-channel convs of layer 4): wave tile 64 pixels x 128 channels. The LDS
// delivers 128 B/clk/CU (tools/lds_rate.hip); with 64 x 64 wave tiles a tap-step asks it for 8 waves x 16 fragment reads =
// 128 KiB + 21 KiB of DMA writes per 1024 MFMA cycles per SIMD -- 146 B/clk, MORE than it has, so the matrix pipe cannot
// exceed 0.88 and sits at 0.5 once the two are not perfectly overlapped. 64 x 128 wave tiles read 24 fragments per 64 MFMAs:
// 192 + 37 KiB per 2048 cycles = 112 B/clk. Two 32 KiB weight slots (the weights of step + 1 land during step), the out tile
// (128 KiB) overlays patches + weight ring, the lane's 36 first-k-step patch offsets are held and the second k-step's derived
// (offset ^ 64) to stay inside 256 registers with 128 of them accumulators.
template <int DBG, int BN_ = 128, int WRING = 3>  // WRING: weight-fragment ring of the 256-channel form (look-ahead WRING - 1 groups; 3 / 4 / 5: 127.2 / 123.8 / 123.5 us); DBG: ablation bits (profiling only): 1 no weight DMA after the prologue, 2 no MFMA, 4 no fragment reads, 8 no waits / barriers
__global__ __launch_bounds__(512) void conv3x3_wide_kernel(const IgemmParams p, int nblocks) {
    constexpr int BN = BN_, NW = 8, WM = 4, BM = 256;
    constexpr int FM = BM / (16 * WM), FN = BN / 32;  // 4 x 4 (x 8) fragments per wave
    constexpr int PW = 10, PPIX = 18 * PW;            // 18 x 10 halo patch per block
    constexpr int PPIECES = (PPIX + 7) / 8;           // 23 one-KiB pieces (8 patch pixels each)
    constexpr int PATCH_BYTES = PPIECES * 1024;       // 23552
    constexpr int SLAB_BYTES = 2 * PATCH_BYTES;       // both blocks
    constexpr int PJ = (2 * PPIECES + NW - 1) / NW;   // patch pieces per wave per slab (6, two of the 48 are dummies)
    constexpr int B_BYTES = BN * 128, WSLOTS = BN == 256 ? 2 : 3;
    constexpr int BJ = BN / 64;                       // weight pieces per wave per tap-step (2)
    constexpr int CPR = BN * 2 / 16, ROWB = BN * 2;
    static_assert(BM * ROWB <= 2 * SLAB_BYTES + (BN == 256 ? WSLOTS * B_BYTES : 0), "out tile must fit over the patch buffers (+ the weight ring)");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SLAB_BYTES + WSLOTS * B_BYTES + 1024];
    unsigned char* s_b = smem + 2 * SLAB_BYTES;
    unsigned char* s_dummy = s_b + WSLOTS * B_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int lrow = lane >> 3, lchk = lane & 7;

    const int nNt = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, within = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int mt = bid / nNt;
    const int nt = bid - mt * nNt;
    const int n0 = nt * BN;
    const int tw = p.W >> 3, th = p.H >> 4;
    // pixel block bsel of this tile = block 2 mt + bsel in (image, block row, block column) order
    auto block_origin = [&](int bsel, int& img, int& oy0, int& ox0) {
        const int blk = min(2 * mt + bsel, nblocks - 1);
        img = blk / (tw * th);
        const int trem = blk - img * (tw * th);
        oy0 = (trem / tw) << 4;
        ox0 = (trem % tw) << 3;
    };
    const bool second_valid = 2 * mt + 1 < nblocks;

    const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* __restrict__ wg = reinterpret_cast<const unsigned char*>(p.w);
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);

    // ---- DMA coordinates. Patch piece index pi = wave + 8 i (i < 6) over the 46 pieces of the two blocks
    unsigned poff[PJ];
    bool pok[PJ];
    int pdst[PJ];
#pragma unroll
    for (int i = 0; i < PJ; ++i) {
        const int pi = wave + NW * i;
        const int bsel = pi >= PPIECES ? 1 : 0;
        const int piece = pi - bsel * PPIECES;
        int img, oy0, ox0;
        block_origin(bsel, img, oy0, ox0);
        const int row = piece * 8 + lrow;  // patch pixel index
        const int py = row / PW, px = row - py * PW;
        const int iy = oy0 + py - 1, ix = ox0 + px - 1;
        const bool real = pi < 2 * PPIECES;
        pok[i] = real && row < PPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                 (bsel == 0 || second_valid);
        poff[i] = (unsigned)((((size_t)img * p.H + iy) * p.W + ix) * p.Cin * 2 + ((lchk ^ patch_g(py, px)) << 4));
        pdst[i] = real ? bsel * PATCH_BYTES + piece * 1024 : -1;
    }
    unsigned boff[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = wave * (BN / NW) + j * 8 + lrow;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        boff[j] = (unsigned)(((size_t)gn * p.K) * 2 + ((lchk ^ ((row >> 1) & 7)) << 4));
    }
    auto stage_patch_piece = [&](int slab, int i) {  // always exactly one DMA (dummies keep every wave's vmcnt equal)
        dma16(pok[i] ? xg + poff[i] + slab * 128 : zsrc,
              pdst[i] >= 0 ? smem + (slab & 1) * SLAB_BYTES + pdst[i] : s_dummy);
    };
    auto stage_b = [&](int slab, int tap, int slot) {
        const unsigned koff = (unsigned)(tap * p.Cin + slab * 64) * 2;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            dma16(wg + boff[j] + koff, s_b + slot * B_BYTES + (wave * (BN / NW) + j * 8) * 128);
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchunk = lane >> 4;
    // this wave's 64 pixels lie in block wm >> 1 (pixels (wm & 1) * 64 .. + 63 of it)
    const int pbase = (wm >> 1) * PATCH_BYTES;
    int py0[FM], px0[FM];  // patch pixel of this lane's output pixel at tap (0,0)
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = (wm & 1) * 64 + b * 16 + frag_px(frow);
        py0[b] = m >> 3;
        px0[b] = m & 7;
    }

    const int nslab = p.Cin >> 6;
    const int nsteps = 9 * nslab;
    // prologue: slab 0 patches, weights of steps 0 and 1
#pragma unroll
    for (int i = 0; i < PJ; ++i) stage_patch_piece(0, i);
    stage_b(0, 0, 0);
    if (WSLOTS == 3) stage_b(0, 1, 1);
    // BN = 256: first-k-step patch offsets of the nine taps (the second k-step's chunk index differs in bit 2: offset ^ 64)
    int xoff[BN == 256 ? 9 : 1][FM];
    if constexpr (BN == 256) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int b = 0; b < FM; ++b) xoff[t][b] = patch_off<PW>(py0[b] + t / 3, px0[b] + t % 3, fchunk);
    }

    int wslot = 0;  // weight ring slot of the current step
    for (int slab = 0; slab < nslab; ++slab) {
        const bool next_slab = slab + 1 < nslab;
        const unsigned char* sp = smem + (slab & 1) * SLAB_BYTES + pbase;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = slab * 9 + tap;
            // Younger than this step's weights in this wave's queue: the weights of step+1 (BJ pieces) and the patch
            // pieces issued since. Patch pieces of slab+1 are issued in taps 0..PJ-1 of slab; this step's weights were
            // issued at step-2, so the pieces of steps step-2 (after the weights), step-1 are younger; waiting them
            // out is harmless except for the count, so count them exactly:
            //   issue order inside a step: [patch piece (tap < PJ, next_slab)] then [weights of step+2]
            // => after weights(step): patch(step-1)? , weights(step+1)  -> BJ + (tap-1 in [0,PJ) && had_next ? 1 : 0)
            // At a slab boundary (tap == 0) the current slab's own patches must have landed too: they are older than
            // weights(step) except the piece issued at tap PJ-1 <= step-2 ... all older. So the same wait covers them.
            {
                const bool more = step + 1 < nsteps;
                const int ptap = tap - 1;  // the tap of step-1 (same slab unless tap == 0)
                const bool patch_younger = (tap >= 1) && (ptap < PJ) && next_slab;
                if (DBG & 8) {
                } else if (!more || WSLOTS == 2) wait_vmcnt<0>();  // two slots: this step's weights are the youngest request
                else if (patch_younger) wait_vmcnt<BJ + 1>();
                else wait_vmcnt<BJ>();
            }
            if (!(DBG & 8)) wg_barrier();
            // refill: next slab's patch piece, then the weights two steps ahead (into the slot read at step-1)
            if (tap < PJ && next_slab && !(DBG & 1)) stage_patch_piece(slab + 1, tap);
            if (step + (WSLOTS - 1) < nsteps && !(DBG & 1)) {
                int t2 = tap + (WSLOTS - 1), s2 = slab;
                if (t2 >= 9) { t2 -= 9; ++s2; }
                int slot2 = wslot + (WSLOTS - 1);
                slot2 = slot2 >= WSLOTS ? slot2 - WSLOTS : slot2;
                stage_b(s2, t2, slot2);
            }
            const int tr = tap / 3, ts = tap - tr * 3;
            const unsigned char* sb = s_b + wslot * B_BYTES;
            // 2 FN groups of FM MFMAs: group g = (k-step g / FN, channel fragment g % FN)
            constexpr int NG = 2 * FN;
            constexpr int WR = BN == 256 ? WRING : 3;   // weight-fragment ring: WR - 1 groups of look-ahead
            uint4 xfr[2][FM], wfr[WR];
            int k64 = 64;
            if constexpr (BN == 256) asm volatile("" : "+v"(k64));  // keeps offset ^ 64 a per-step instruction, not 36 more registers
            auto ldx = [&](int kk, int b) {
                if ((DBG & 4) && step) return make_uint4(step, kk, b, lane);
                if constexpr (BN == 256) return *reinterpret_cast<const uint4*>(sp + (kk ? xoff[tap][b] ^ k64 : xoff[tap][b]));
                else return *reinterpret_cast<const uint4*>(sp + patch_off<PW>(py0[b] + tr, px0[b] + ts, kk * 4 + fchunk));
            };
            auto ldw = [&](int g) {
                if ((DBG & 4) && step) return make_uint4(step, g, 7, lane);
                return *reinterpret_cast<const uint4*>(sb + lds_off(wn * (BN / 2) + (g % FN) * 16 + frow, (g / FN) * 4 + fchunk));
            };
            wfr[0] = ldw(0);
#pragma unroll
            for (int b = 0; b < FM; ++b) xfr[0][b] = ldx(0, b);
#pragma unroll
            for (int q = 1; q < WR - 1; ++q) wfr[q] = ldw(q);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + WR - 1 < NG) wfr[(g + WR - 1) % WR] = ldw(g + WR - 1);
                if (g < FM) xfr[1][g] = ldx(1, g);
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    if (!(DBG & 2)) acc[g % FN][b] = Frag<lp16_t>::mma(wfr[g % WR], xfr[g / FN][b], acc[g % FN][b]);
                    else asm volatile("" ::"v"(wfr[g % WR].x), "v"(wfr[g % WR].w), "v"(xfr[g / FN][b].x), "v"(xfr[g / FN][b].w));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (BN = 256: carrying the first k-step's pixel fragments from step to step -- read under the previous step's second
            // k-step so that a step starts with two reads instead of six -- was measured: the longer live ranges spill inside
            // the slab loop, 154 us against 125 us)
            wslot = wslot + 1 == WSLOTS ? 0 : wslot + 1;
        }
    }
    wait_vmcnt<0>();
    wg_barrier();  // all fragment reads done: the patch buffers become the out tile

    unsigned char* so = smem;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int c = wn * (BN / 2) + a * 16 + fchunk * 4;
        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.colv && n0 + c < p.N) cv = *reinterpret_cast<const float4*>(p.colv + n0 + c);
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int prow = wm * (BM / WM) + b * 16 + frag_px(frow);
            float v[4] = {acc[a][b][0] + cv.x, acc[a][b][1] + cv.y, acc[a][b][2] + cv.z, acc[a][b][3] + cv.w};
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = relu_nan(v[r]);
            }
            store4<lp16_t>(reinterpret_cast<lp16_t*>(so + prow * ROWB + (((c >> 3) ^ (prow & (CPR - 1))) << 4) + ((c & 4) << 1)), v);
        }
    }
    wg_barrier();
    constexpr int RPT = 64 * NW / CPR;  // 32 rows per pass
    const int pch = tid % CPR;
    const int r0 = tid / CPR;
#pragma unroll
    for (int i = 0; i < BM / RPT; ++i) {
        const int row = r0 + i * RPT;
        const int bsel = row >> 7, m = row & 127;
        if (bsel == 1 && !second_valid) continue;
        const int gch = pch ^ (row & (CPR - 1));
        const int gn = n0 + gch * 8;
        if (gn < p.N) {
            int img, oy0, ox0;
            block_origin(bsel, img, oy0, ox0);
            const size_t gm = ((size_t)img * p.H + oy0 + (m >> 3)) * p.W + ox0 + (m & 7);
            const uint4 v = *reinterpret_cast<const uint4*>(so + row * ROWB + (pch << 4));
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + (gm * p.ldo + gn) * 2) = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1, 64 -> 64 channels (layer 1, 64 x 32 maps): HBM-bound (67 MB in, 67 MB out per 256 frames
// against 39 GFLOP), and with one 64-channel slab the whole filter bank is 72 KB. Persistent workgroups keep ALL NINE
// taps' weights resident in LDS and walk 16 x 8 pixel blocks: per block only the 23 KB halo patch comes in (LDS-DMA,
// prefetched one block ahead) and 16 KB goes out; no weight traffic, no barrier inside the 9-tap sweep (72 MFMAs per
// wave between barriers instead of 8). (Weights in registers -- 144 VGPRs of A fragments -- was tried: with the 36
// shifted patch addresses hipcc runs out of registers and spills, 60 us against 51 us for this form.)
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const IgemmParams p, int nblocks) {
    constexpr int NW = 8, WM = 4, BM = 128, BN = 64;
    constexpr int FM = BM / (16 * WM), FN = BN / 32;  // 2 x 2 fragments per wave
    constexpr int PW = 10, PPIX = 18 * PW, PPIECES = (PPIX + 7) / 8, PATCH_BYTES = PPIECES * 1024;
    constexpr int PJ = (PPIECES + NW - 1) / NW;       // 3 DMA pieces per wave per block (one of the 24 is a dummy)
    constexpr int W_TAP = BN * 128, W_BYTES = 9 * W_TAP;  // 72 KB
    constexpr int O_BYTES = BM * BN * 2;                   // 16 KB out tile, 128-byte rows
    __shared__ __attribute__((aligned(16))) unsigned char smem[W_BYTES + 2 * PATCH_BYTES + O_BYTES + 1024];
    unsigned char* s_w = smem;
    unsigned char* s_p = s_w + W_BYTES;
    unsigned char* s_o = s_p + 2 * PATCH_BYTES;
    unsigned char* s_dummy = s_o + O_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int lrow = lane >> 3, lchk = lane & 7;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int G = gridDim.x;
    const int tw = p.W >> 3, th = p.H >> 4;
    const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);

    // resident weights: tap t, rows 8 wave .. +7 (64 output channels); K order of the OHWI weight = (tap, cin)
    {
        const int row = wave * 8 + lrow;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            dma16(reinterpret_cast<const unsigned char*>(p.w) + ((size_t)row * p.K + t * 64) * 2 + ((lchk ^ ((row >> 1) & 7)) << 4),
                  s_w + t * W_TAP + wave * 8 * 128);
    }
    auto stage_patch = [&](int blk, int buf) {
        const int img = blk / (tw * th);
        const int trem = blk - img * (tw * th);
        const int oy0 = (trem / tw) << 4, ox0 = (trem % tw) << 3;
#pragma unroll
        for (int i = 0; i < PJ; ++i) {
            const int piece = wave + NW * i;
            const int row = piece * 8 + lrow;
            const int py = row / PW, px = row - py * PW;
            const int iy = oy0 + py - 1, ix = ox0 + px - 1;
            const bool ok = piece < PPIECES && row < PPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            dma16(ok ? xg + (((size_t)img * p.H + iy) * p.W + ix) * 128 + ((lchk ^ patch_g(py, px)) << 4) : zsrc,
                  piece < PPIECES ? s_p + buf * PATCH_BYTES + piece * 1024 : s_dummy);
        }
    };
    float4 cv[FN];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        cv[a] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.colv) cv[a] = *reinterpret_cast<const float4*>(p.colv + wn * 32 + a * 16 + fchunk * 4);
    }
    int py0[FM], px0[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = wm * 32 + b * 16 + frag_px(frow);
        py0[b] = m >> 3;
        px0[b] = m & 7;
    }

    int blk = blockIdx.x;
    if (blk < nblocks) stage_patch(blk, 0);
    int buf = 0;
    bool first = true;
    for (; blk < nblocks; blk += G, buf ^= 1) {
        // this block's patch (and, the first time, the weights) are the oldest entries of the queue: the previous
        // block's 2 output stores may stay in flight
        if (first) wait_vmcnt<0>();
        else wait_vmcnt<2>();
        first = false;
        wg_barrier();
        if (blk + G < nblocks) stage_patch(blk + G, buf ^ 1);
        const unsigned char* sp = s_p + buf * PATCH_BYTES;
        f32x4_t acc[FN][FM];
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 xf[FM], wf[FN];
#pragma unroll
                for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const uint4*>(sp + patch_off<PW>(py0[b] + t / 3, px0[b] + t % 3, kk * 4 + fchunk));
#pragma unroll
                for (int a = 0; a < FN; ++a)
                    wf[a] = *reinterpret_cast<const uint4*>(s_w + t * W_TAP + lds_off(wn * 32 + a * 16 + frow, kk * 4 + fchunk));
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b) acc[a][b] = Frag<lp16_t>::mma(wf[a], xf[b], acc[a][b]);
            }
        }
        // bias + ReLU -> bf16 out tile (128-byte rows, chunk c at c ^ (row & 7)); the previous block's stores read it
        // before the barrier at the top of this iteration
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int prow = wm * 32 + b * 16 + frag_px(frow);
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int c = wn * 32 + a * 16 + fchunk * 4;
                float v[4] = {acc[a][b][0] + cv[a].x, acc[a][b][1] + cv[a].y, acc[a][b][2] + cv[a].z, acc[a][b][3] + cv[a].w};
                if (p.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = relu_nan(v[r]);
                }
                store4<lp16_t>(reinterpret_cast<lp16_t*>(s_o + prow * 128 + (((c >> 3) ^ (prow & 7)) << 4) + ((c & 4) << 1)), v);
            }
        }
        wg_barrier();
        {
            const int img = blk / (tw * th);
            const int trem = blk - img * (tw * th);
            const int oy0 = (trem / tw) << 4, ox0 = (trem % tw) << 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (tid >> 3) + 64 * i, pch = tid & 7;
                const size_t gm = ((size_t)img * p.H + oy0 + (row >> 3)) * p.W + ox0 + (row & 7);
                const uint4 v = *reinterpret_cast<const uint4*>(s_o + row * 128 + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + (gm * p.ldo + (pch ^ (row & 7)) * 8) * 2) = v;
            }
        }
    }
}

}  // namespace

int launch_conv3x3_wide(const IgemmParams& p, hipStream_t stream) {
    const int nblocks = (p.M / p.OH / p.OW) * (p.H / 16) * (p.W / 8);
    const int grid = cdiv(nblocks, 2) * cdiv(p.N, 128);
    switch (agrl_opts().conv3x3_dbg) {  // non-zero only in an -DAGRL_ABLATE build
#ifdef AGRL_ABLATE
#define C3_CASE(D) case D: hipLaunchKernelGGL(conv3x3_wide_kernel<D>, dim3(grid), dim3(512), 0, stream, p, nblocks); break
        C3_CASE(1); C3_CASE(2); C3_CASE(4); C3_CASE(8); C3_CASE(5); C3_CASE(13); C3_CASE(6); C3_CASE(9);
#undef C3_CASE
#endif
        default:
            // 256-channel tiles when they still cover the chip (layer 4: 128 pixel tiles x 2 = 256 workgroups); AGRL_CONV3X3_N128=1: A/B
            if ((p.N % 256) == 0 && cdiv(nblocks, 2) * (p.N / 256) >= 192 && !agrl_opts().conv3x3_n128)
                hipLaunchKernelGGL((conv3x3_wide_kernel<0, 256, 5>), dim3(cdiv(nblocks, 2) * (p.N / 256)), dim3(512), 0, stream, p, nblocks);
            else hipLaunchKernelGGL(conv3x3_wide_kernel<0>, dim3(grid), dim3(512), 0, stream, p, nblocks);
    }
    AGRL_CHECK_LAUNCH("agrl_conv2d_bn_act(3x3 wide)");
    return 0;
}

int launch_conv3x3_c64(const IgemmParams& p, hipStream_t stream) {
    const int nblocks = (p.M / p.OH / p.OW) * (p.H / 16) * (p.W / 8);
    const int grid = nblocks < 256 ? nblocks : 256;
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(grid), dim3(512), 0, stream, p, nblocks);
    AGRL_CHECK_LAUNCH("agrl_conv2d_bn_act(3x3 c64)");
    return 0;
}
