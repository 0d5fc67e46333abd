// Whole identity-shortcut Bottleneck of layer 3 in ONE kernel, one 16 x 8 frame per workgroup (bf16):
//     y1  = relu(x  @ W1^T + b1)              1x1, Cin -> 256            (vmgn.py:48-50)
//     y2  = relu(conv3x3(y1, W2) + b2)        3x3 pad 1, 256 -> 256      (vmgn.py:52-54)
//     out = relu(y2 @ W3^T + b3 + x)          1x1, 256 -> Cin, + shortcut (vmgn.py:56-64)
// with every BatchNorm folded (eval). A 16 x 8 frame IS the natural tile of the 3x3 conv: its zero padding is the frame
// border, no halo from another tile exists. So y1 and y2 (128 pixels x 256 channels = 64 KB each) live in the LDS for the whole
// block, x is read from HBM once (as the k-tiles of the first GEMM; its second use as the shortcut hits L2) and out is written
// once: 128 MB of HBM traffic per block at 256 frames instead of ~270 MB for the three separate launches, and the K = 256
// GEMMs (4 k-tiles: pure latency as separate kernels) run out of a warm pipeline.
//
// Wave arrangement 1 (pixels) x 8 (channels): every wave owns 32 output channels of ALL 128 pixels (64 accumulator registers)
// and therefore reads only ITS OWN 32 weight rows -- it streams them from L2 through a PRIVATE ring of 4 KB k-tiles (LDS-DMA)
// and waits for nothing but its own vmcnt. The pixel operand of phases 2 and 3 is a y image that nobody writes during the
// phase, so those phases run WITHOUT a single workgroup barrier: the eight waves drift apart and cover each other's LDS / DMA
// latencies (a first version in the usual 4 x 2 arrangement with shared weight tiles and one barrier per 0.5 us k-tile spent two
// thirds of its time in those barriers and the fragment-read round trips behind them: 98 us per block, no better than the three
// separate launches). Only phase 1 shares a streamed operand (the x k-tiles, 32 KB = two k-tiles per barrier).
//
// LDS (160 KB):  R0 [0, 64K)  y1 image            R2 [64K, 96K) spare            R1 [96K, 160K)  y2 image
//   phase 1: x ring 2 x 32 KB in R0, private W1 rings (8 waves x 3 x 4 KB) in R2|R1; y1 -> R0 after the k-loop
//   phase 2: private W2 rings in R2|R1 (tap-major: 9 taps x 4 k-tiles);                 y2 -> R1 after the k-loop
//   phase 3: private W3 rings in R0|R2 (4 channel chunks x 4 k-tiles);                  out -> HBM per 256-channel chunk
// y images: 512-byte pixel rows, 16-byte chunk c of row r at chunk (c & 16) | ((c ^ r) & 15): the 16 lanes a ds_read_b128 is
// served in hit 16 different slots of the 256-byte bank window for every tap shift (rows r + const keep r mod 16 distinct).
// A wave's 32 weight rows are staged in the order (a, i) -> channel 8 (i >> 2) + 4 a + (i & 3): the two MFMA results of a lane
// are then 8 consecutive channels -> 16-byte LDS / HBM epilogue accesses, 64 contiguous bytes per pixel and instruction.
#include "igemm_dev.h"

#ifndef AGRL_FRAME_ABL
#define AGRL_FRAME_ABL 0   // development ablations (compile time, a separate library only): 1 no MFMA, 2 no steady-state DMA, 8 skip phase 1, 16 skip phase 2, 32 skip phase 3
#endif

namespace {

constexpr int FABL = AGRL_FRAME_ABL;
constexpr int FPX = 128;      // pixels per frame
constexpr int FCM = 256;      // bottleneck width
constexpr int R0_OFF = 0, R2_OFF = 65536, R1_OFF = 98304;
constexpr int NSW = 3;        // private ring slots per wave (8 KB each)

struct FrameParams {
    const void* x;
    const void* w1;
    const float* b1;
    const void* w2;
    const float* b2;
    const void* w3;
    const float* b3;
    void* out;
    int F, Cin;
};

__device__ inline int yimg_off(int row, int chunk) { return row * 512 + (((chunk & 16) | ((chunk ^ row) & 15)) << 4); }

template <int N>
__device__ inline void wait_vm_le() {  // wait until at most N vector-memory operations of this wave are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ inline void wait_vm_dyn(int n) {  // n in {0, 4, 8, 12, 16, 24, 40}
    if (n >= 40) wait_vm_le<40>();
    else if (n >= 24) wait_vm_le<24>();
    else if (n >= 16) wait_vm_le<16>();
    else if (n >= 12) wait_vm_le<12>();
    else if (n >= 8) wait_vm_le<8>();
    else if (n >= 4) wait_vm_le<4>();
    else wait_vm_le<0>();
}

__global__ __launch_bounds__(256) void bottleneck_frame_kernel(const FrameParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FB = 8;   // pixel fragments per wave (all 128 pixels)
    constexpr int NA = 4;   // weight fragments per wave (64 output channels)
    constexpr int TILE = NA * 16 * 128;   // private weight k-tile: 64 rows x 128 B = 8 KB = 8 DMA pieces
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, fchunk = lane >> 4;
    const int lrow = lane >> 3, lchk = lane & 7;
    const int Cin = p.Cin;
    const unsigned char* __restrict__ w1g = reinterpret_cast<const unsigned char*>(p.w1);
    const unsigned char* __restrict__ w2g = reinterpret_cast<const unsigned char*>(p.w2);
    const unsigned char* __restrict__ w3g = reinterpret_cast<const unsigned char*>(p.w3);

    // private weight tile: 64 rows x 128 B, DMA piece j = rows 8j .. 8j+7; row (a = row >> 4, i = row & 15) holds the wave's
    // channel 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3): a lane's results acc[2j], acc[2j+1] are 8 consecutive channels
    auto piece_ch = [&](int j) {
        const int row = j * 8 + lrow;
        const int a = row >> 4, i = row & 15;
        return wave * 64 + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
    };
    auto piece_sw = [&](int j) {
        const int row = j * 8 + lrow;
        return (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
    };

    for (int frame = blockIdx.x; frame < p.F; frame += gridDim.x) {
        const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x) + (size_t)frame * FPX * Cin * 2;
        unsigned char* __restrict__ og = reinterpret_cast<unsigned char*>(p.out) + (size_t)frame * FPX * Cin * 2;
        // the lane's fragment / epilogue addresses are re-derived from an opaque lane id at every phase boundary: hoisted out of
        // the frame loop they would all stay live across the three phases
        auto launder = [&]() { asm volatile("" : "+v"(frow), "+v"(fchunk)); };
        f32x4_t acc[NA][FB];
        auto zero_acc = [&]() {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < FB; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        };
        // one 32-deep k-step: 4 weight fragments x 8 pixel fragments = 32 MFMAs for 12 fragment reads
        auto mma_kstep = [&](const uint4 (&wf)[NA], const uint4 (&xf)[FB]) {
#pragma unroll
            for (int b = 0; b < FB; ++b)
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    if (!(FABL & 1)) acc[a][b] = Frag<bf16_t>::mma(wf[a], xf[b], acc[a][b]);
                    else asm volatile("" ::"v"(wf[a].x), "v"(wf[a].w), "v"(xf[b].x), "v"(xf[b].w));
                }
        };
        auto ld_wf = [&](const unsigned char* sw, int kk, uint4 (&wf)[NA]) {
#pragma unroll
            for (int a = 0; a < NA; ++a) wf[a] = (FABL & 64) ? make_uint4(a, kk, lane, 3) : *reinterpret_cast<const uint4*>(sw + lds_off(a * 16 + frow, kk * 4 + fchunk));
        };
        // bias + ReLU + bf16: this lane's 2 x 8 consecutive channels of 8 pixels -> the y image at `img`
        auto store_yimg = [&](unsigned char* img, const float* __restrict__ bias) {
            launder();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cw = wave * 64 + 32 * j + 8 * fchunk;
                const float4 b0 = *reinterpret_cast<const float4*>(bias + cw);
                const float4 b1 = *reinterpret_cast<const float4*>(bias + cw + 4);
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    float v[8] = {acc[2 * j][b][0] + b0.x, acc[2 * j][b][1] + b0.y, acc[2 * j][b][2] + b0.z, acc[2 * j][b][3] + b0.w,
                                  acc[2 * j + 1][b][0] + b1.x, acc[2 * j + 1][b][1] + b1.y, acc[2 * j + 1][b][2] + b1.z, acc[2 * j + 1][b][3] + b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    const uint4 pk = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                    *reinterpret_cast<uint4*>(img + yimg_off(b * 16 + frow, cw >> 3)) = pk;
                }
            }
        };

        // =========================== phase 1: y1 = relu(x W1^T + b1), K = Cin ===========================
        // x: shared ring of FOUR 16 KB k-tiles in R0 (three in flight: a CU streams its 256 KB of x latency-bound, so bytes in
        // flight are what sets the rate), one barrier per k-tile; W1: private rings in R2|R1.
        {
            const int nk = (FABL & 8) ? 1 : (Cin >> 6);
            unsigned char* wring = smem + R2_OFF + wave * (NSW * TILE);
            auto stage_x = [&](int t, int q) {  // x k-tile t: 128 rows x 128 B in lds_off layout; piece q (0..3) of this wave = rows wave*32 + 8q ..
                const int row = wave * 32 + q * 8 + lrow;
                const unsigned off = (unsigned)row * (unsigned)Cin * 2u + (unsigned)((lchk ^ ((row >> 1) & 7)) << 4) + (unsigned)t * 128u;
                dma16(xg + off, smem + R0_OFF + (t & 3) * 16384 + (wave * 32 + q * 8) * 128);
            };
            auto stage_w = [&](int t, int j) {  // private W1 k-tile t, piece j
                dma16(w1g + (size_t)piece_ch(j) * Cin * 2 + piece_sw(j) + (unsigned)t * 128u, wring + (t % NSW) * TILE + j * 1024);
            };
            launder();
            zero_acc();
            // prologue; queue afterwards (old -> young): x0 x1 x2 (4 each), w0 w1 (8 each)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t < nk) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) stage_x(t, q);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (t < nk) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) stage_w(t, j);
                }
#pragma unroll 1
            for (int t = 0; t < nk; ++t) {
                // issue order: prologue x0 x1 x2 w0 w1, then per iteration i: x(i+3), w(i+2). x(t) and w(t) must have landed; younger
                // than w(t): [t >= 1: x(t+2) (4)] and w(t+1) (8)
                int younger = (t + 1 < nk ? 8 : 0);
                if (t >= 1 && t + 2 < nk) younger += 4;
                wait_vm_dyn(younger);
                __builtin_amdgcn_s_barrier();   // everybody's pieces of x k-tile t are in; slot (t+3) & 3 (tile t-1) is free
                asm volatile("" ::: "memory");
                if (t + 3 < nk) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) stage_x(t + 3, q);
                }
                const unsigned char* sx = smem + R0_OFF + (t & 3) * 16384;
                const unsigned char* sw = wring + (t % NSW) * TILE;
                const bool more_w = t + 2 < nk;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 wf[NA], xf[FB];
                    ld_wf(sw, kk, wf);
#pragma unroll
                    for (int b = 0; b < FB; ++b) xf[b] = (FABL & 4) ? make_uint4(b, kk, lane, 5) : *reinterpret_cast<const uint4*>(sx + lds_off(b * 16 + frow, kk * 4 + fchunk));
                    if (more_w && !(FABL & 2)) {   // W1 tile t+2 into the slot of tile t-1 (this wave read it last)
#pragma unroll
                        for (int j = 0; j < 4; ++j) stage_w(t + 2, 4 * kk + j);
                    }
                    mma_kstep(wf, xf);
                }
            }
            wait_vm_le<0>();
            wg_barrier();  // every slot read is done: R0 may become the y1 image
            store_yimg(smem + R0_OFF, p.b1);
        }
        wg_barrier();  // y1 complete

        // =========================== phase 2: y2 = relu(conv3x3(y1) + b2): 9 taps x 4 k-tiles, no barrier ===========================
        {
            constexpr int NT = 36;
            unsigned char* wring = smem + R2_OFF + wave * (NSW * TILE);
            auto stage_w = [&](int t, int j) {  // W2 is OHWI (256, 3, 3, 256): k-tile t = (tap t >> 2, input channels 64 (t & 3) ..)
                const unsigned koff = (unsigned)(t >> 2) * 512u + (unsigned)(t & 3) * 128u;
                dma16(w2g + (size_t)piece_ch(j) * (9 * FCM * 2) + piece_sw(j) + koff, wring + (t % NSW) * TILE + j * 1024);
            };
            launder();
            zero_acc();
#pragma unroll
            for (int j = 0; j < 8; ++j) stage_w(0, j);
#pragma unroll
            for (int j = 0; j < 8; ++j) stage_w(1, j);
            const unsigned char* y1 = smem + R0_OFF;
#pragma unroll 1
            for (int tap = 0; tap < ((FABL & 16) ? 0 : 9); ++tap) {
                launder();   // per-tap address set: keep the other taps' out of the registers
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                // The lane's pixel of fragment b is b*16 + frow = image row 2b + (frow >> 3), column frow & 7; the tap reads pixel
                // + 8 dy + dx. Column validity is the same for all eight fragments; a row can only fall off the frame for b == 0
                // (dy = -1, upper image row of the fragment) and b == 7 (dy = +1, lower one). (row & 15) -- the swizzle key -- is the
                // same for all b as well, so the eight reads of a k-step are ONE address + immediate offsets b * 8 KB.
                const int sh = 8 * dy + dx;
                const bool xok = (unsigned)((frow & 7) + dx) < 8u;
                const bool bad0 = dy < 0 && frow < 8, bad7 = dy > 0 && frow >= 8;
                const int rr = (frow + sh) & 15;
                const int rowb = (frow + sh) * 512;                       // byte offset of the b = 0 row (may lie outside for bad0)
                const int rowb0 = bad0 ? frow * 512 : rowb;               // in-range stand-in (the value is zeroed below)
                const int rowb7 = (bad7 ? frow * 512 : rowb) + 7 * 8192;
#pragma unroll 1
                for (int kc = 0; kc < 4; ++kc) {
                    const int t = tap * 4 + kc;
                    if (t + 1 < NT) wait_vm_le<8>(); else wait_vm_le<0>();   // own tile t landed; own tile t+1 may stay in flight
                    const unsigned char* sw = wring + (t % NSW) * TILE;
                    const bool more = t + 2 < NT;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        uint4 wf[NA], xf[FB];
                        ld_wf(sw, kk, wf);
                        const int chunk = kc * 8 + kk * 4 + fchunk;
                        const int swz = (((chunk ^ rr) & 15) | (chunk & 16)) << 4;
#pragma unroll
                        for (int b = 0; b < FB; ++b) {
                            const int off = (b == 0 ? rowb0 : b == 7 ? rowb7 : rowb + b * 8192) + swz;
                            uint4 v = (FABL & 4) ? make_uint4(off, kc, kk, lane) : *reinterpret_cast<const uint4*>(y1 + off);
                            const bool ok = xok && !(b == 0 && bad0) && !(b == 7 && bad7);
                            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                            xf[b] = v;
                        }
                        if (more && !(FABL & 2)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) stage_w(t + 2, 4 * kk + j);
                        }
                        mma_kstep(wf, xf);
                    }
                }
            }
            wait_vm_le<0>();
            wg_barrier();  // every wave is done with y1 and with its ring (R2|R1): R1 may become the y2 image
            store_yimg(smem + R1_OFF, p.b2);
        }
        wg_barrier();  // y2 complete; y1 (R0) is dead

        // =========================== phase 3: out = relu(y2 W3^T + b3 + x), chunks of 256 channels, no barrier ===========================
        {
            unsigned char* wring = smem + R0_OFF + wave * (NSW * TILE);
            auto stage_w = [&](int t, int j) {  // W3 is (Cin, 256): k-tile t = (channel chunk t >> 2, input channels 64 (t & 3) ..)
                dma16(w3g + (size_t)((t >> 2) * 256 + piece_ch(j)) * (FCM * 2) + piece_sw(j) + (unsigned)(t & 3) * 128u, wring + (t % NSW) * TILE + j * 1024);
            };
#pragma unroll
            for (int j = 0; j < 8; ++j) stage_w(0, j);
#pragma unroll
            for (int j = 0; j < 8; ++j) stage_w(1, j);
            const unsigned char* y2 = smem + R1_OFF;
            const int nchunks = (FABL & 32) ? 0 : (Cin >> 8);
            const int ntiles = nchunks * 4;
#pragma unroll 1
            for (int nc = 0; nc < nchunks; ++nc) {
                launder();
                zero_acc();
                uint4 rres[2][FB];
                // vmcnt bookkeeping (one in-order counter for loads, DMA and stores). Issue order around chunk nc (tiles T0..T3):
                //   .. stage(T1) [previous epilogue: 16 stores] | T0: 16 shortcut loads, stage(T2) | T1: stage(T3) | T2: stage(T4) | T3: stage(T5) | 16 stores
                // T0 waits for tile T0 only: tile T1 (8) and the previous chunk's stores (16) stay in flight -- the stores get two
                // k-tiles to drain before anything has to wait for them (tile T2's wait), the shortcut loads a whole chunk.
                auto ktile3 = [&](int kc) {
                    const int t = nc * 4 + kc;
                    const int nxt = t + 1 < ntiles ? 8 : 0;
                    if (kc == 0) wait_vm_dyn(nxt + (nc > 0 ? 16 : 0));
                    else if (kc == 1) wait_vm_dyn(nxt + (nc > 0 ? 16 : 0) + 16);   // + the shortcut loads issued in T0
                    else wait_vm_dyn(nxt);
                    if (kc == 0) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int b = 0; b < FB; ++b)
                                rres[j][b] = *reinterpret_cast<const uint4*>(xg + ((size_t)(b * 16 + frow) * Cin + nc * 256 + wave * 64 + 32 * j + 8 * fchunk) * 2);
                    }
                    const unsigned char* sw = wring + (t % NSW) * TILE;
                    const bool more = t + 2 < ntiles;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        uint4 wf[NA], xf[FB];
                        ld_wf(sw, kk, wf);
#pragma unroll
                        for (int b = 0; b < FB; ++b) xf[b] = (FABL & 4) ? make_uint4(b, kc, kk, lane) : *reinterpret_cast<const uint4*>(y2 + yimg_off(b * 16 + frow, kc * 8 + kk * 4 + fchunk));
                        if (more && !(FABL & 2)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) stage_w(t + 2, 4 * kk + j);
                        }
                        mma_kstep(wf, xf);
                    }
                };
                ktile3(0);
                ktile3(1);
                ktile3(2);
                ktile3(3);
                launder();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cwe = nc * 256 + wave * 64 + 32 * j + 8 * fchunk;
                    const float4 b0 = *reinterpret_cast<const float4*>(p.b3 + cwe);
                    const float4 b1 = *reinterpret_cast<const float4*>(p.b3 + cwe + 4);
#pragma unroll
                    for (int b = 0; b < FB; ++b) {
                        float v[8] = {acc[2 * j][b][0] + b0.x, acc[2 * j][b][1] + b0.y, acc[2 * j][b][2] + b0.z, acc[2 * j][b][3] + b0.w,
                                      acc[2 * j + 1][b][0] + b1.x, acc[2 * j + 1][b][1] + b1.y, acc[2 * j + 1][b][2] + b1.z, acc[2 * j + 1][b][3] + b1.w};
                        const uint32_t w4[4] = {rres[j][b].x, rres[j][b].y, rres[j][b].z, rres[j][b].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += __uint_as_float(w4[e] << 16);
                            v[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        const uint4 pk = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                        *reinterpret_cast<uint4*>(og + ((size_t)(b * 16 + frow) * Cin + cwe) * 2) = pk;
                    }
                }
            }
        }
        wait_vm_le<0>();
        wg_barrier();  // the next frame's phase 1 stages x into R0 (W3 rings) and W1 into R2|R1 (y2)
    }
}

}  // namespace

extern "C" int agrl_bottleneck_frame(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                                     const float* b3, void* out, int F, int H, int W, int Cin, int Cmid, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && out, "agrl_bottleneck_frame: null pointer");
    AGRL_CHECK_ARG(F > 0 && H == 16 && W == 8, "agrl_bottleneck_frame: frames must be 16 x 8 (got %d x %d)", H, W);
    AGRL_CHECK_ARG(Cmid == FCM && Cin >= 256 && (Cin % 256) == 0, "agrl_bottleneck_frame: built for width 256 and Cin %% 256 == 0 (got %d, %d)", Cmid, Cin);
    const uintptr_t al = (uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)out | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)b3;
    AGRL_CHECK_ARG((al & 15) == 0, "agrl_bottleneck_frame: operands must be 16-byte aligned");
    AGRL_CHECK_ARG((size_t)FPX * Cin * 2 < (1ull << 32), "agrl_bottleneck_frame: frame too large");
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    hipError_t e = hipFuncSetAttribute((const void*)bottleneck_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    AGRL_CHECK_ARG(e == hipSuccess, "agrl_bottleneck_frame: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    FrameParams p;
    p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.out = out; p.F = F; p.Cin = Cin;
    hipLaunchKernelGGL(bottleneck_frame_kernel, dim3(F < n_cu ? F : n_cu), dim3(256), 160 * 1024, (hipStream_t)stream, p);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_frame");
    return 0;
}
