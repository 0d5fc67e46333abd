// Whole identity-shortcut Bottleneck of layer 3 in ONE kernel, one 16 x 8 frame per workgroup (bf16):
//     y1  = relu(x  @ W1^T + b1)              1x1, Cin -> 256            (vmgn.py:48-50)
//     y2  = relu(conv3x3(y1, W2) + b2)        3x3 pad 1, 256 -> 256      (vmgn.py:52-54)
//     out = relu(y2 @ W3^T + b3 + x)          1x1, 256 -> Cin, + shortcut (vmgn.py:56-64)
// with every BatchNorm folded (eval). A 16 x 8 frame IS the natural tile of the 3x3 conv: its zero padding is the frame
// border, no halo from another tile exists. So y1 and y2 (128 pixels x 256 channels = 64 KB each) live in the LDS for the whole
// block, x is read from HBM once (as the k-tiles of the first GEMM; its second use as the shortcut hits L2) and out is written
// once: 128 MB of HBM traffic per block at 256 frames instead of ~270 MB for the three separate launches, three launches'
// prologues / epilogues / drains become one, and the K = 256 GEMMs (4 k-tiles: pure latency as separate kernels) run out of
// a warm pipeline. All three weight matrices (2.2 MB) are STREAMED from L2 through a ring of 32 KB k-tiles (every workgroup
// reads the same bytes at about the same time).
//
// LDS (160 KB):  R0 [0, 64K)  y1 image            R2 [64K, 96K) spare            R1 [96K, 160K)  y2 image
//   phase 1: two 48 KB slots (x k-tile 16 KB + W1 k-tile 32 KB) in R2|R1;   y1 -> R0
//   phase 2: three 32 KB W2 slots in R2|R1 (tap-major: 9 taps x 4 k-tiles);  y2 -> R1 after the last read
//   phase 3: three 32 KB W3 slots in R0|R2 (4 channel chunks x 4 k-tiles);   out -> HBM per 256-channel chunk
// y images: 512-byte pixel rows, 16-byte chunk c of row r at chunk (c & 16) | ((c ^ r) & 15): the 16 lanes a ds_read_b128 is
// served in hit 16 different slots of the 256-byte bank window for every tap shift (rows r + const keep r mod 16 distinct).
// 8 waves as 4 (pixels) x 2 (channels), wave tile 32 x 128, 64 accumulator registers; weight rows are staged in the
// permuted order of igemm_wide.hip so that a lane's results are 8 consecutive channels: 16-byte LDS / HBM epilogue accesses.
#include "igemm_dev.h"

namespace {

constexpr int FPX = 128;      // pixels per frame
constexpr int FCM = 256;      // bottleneck width
constexpr int R0_OFF = 0, R2_OFF = 65536, R1_OFF = 98304;

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
__device__ inline void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(512) void bottleneck_frame_kernel(const FrameParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WM = 4, FM = 2, FN = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    int frow = lane & 15, fchunk = lane >> 4;
    const int lrow = lane >> 3, lchk = lane & 7;
    const int Cin = p.Cin;
    const unsigned char* __restrict__ w1g = reinterpret_cast<const unsigned char*>(p.w1);
    const unsigned char* __restrict__ w2g = reinterpret_cast<const unsigned char*>(p.w2);
    const unsigned char* __restrict__ w3g = reinterpret_cast<const unsigned char*>(p.w3);

    // weight-tile DMA geometry: piece j of this wave = tile rows wave*32 + 8j .. +7; LDS row -> output channel through the
    // register-epilogue permutation (igemm_wide.hip): row a*16 + i of a 128-channel slab holds channel 32(a>>1) + 8(i>>2) + 4(a&1) + (i&3)
    int b_ch[4];
    unsigned b_sw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + lrow;
        const int rp = row & 127, a = rp >> 4, i = rp & 15;
        b_ch[j] = (row & ~127) + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
        b_sw[j] = (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
    }
    // pixel-tile DMA geometry of phase 1: piece j = frame rows wave*16 + 8j .. +7
    unsigned a_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave * 16 + j * 8 + lrow;
        a_off[j] = (unsigned)row * (unsigned)Cin * 2u + (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
    }

    for (int frame = blockIdx.x; frame < p.F; frame += gridDim.x) {
        const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x) + (size_t)frame * FPX * Cin * 2;
        unsigned char* __restrict__ og = reinterpret_cast<unsigned char*>(p.out) + (size_t)frame * FPX * Cin * 2;
        // the lane's fragment / epilogue addresses are re-derived from an opaque lane id at every phase boundary: hoisted out of
        // the frame loop they would all stay live across the three phases and spill (a spill reload's vmcnt(0) drains the ring)
        auto launder = [&]() { asm volatile("" : "+v"(frow), "+v"(fchunk)); };
        f32x4_t acc[FN][FM];
        auto zero_acc = [&]() {
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        };
        // 32 MFMAs of one k-tile: weight fragments from sb (128-byte-row tile), pixel fragments supplied by the caller
        auto mma_ktile = [&](const unsigned char* sb, const uint4 (&xf)[2][FM]) {
            // weight fragments four at a time (16 registers), the next four requested before the first four's MFMAs
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 wf[2][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    wf[0][a] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * 128 + a * 16 + frow, kk * 4 + fchunk));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 0) {
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            wf[1][a] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * 128 + (4 + a) * 16 + frow, kk * 4 + fchunk));
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < FM; ++b) acc[4 * h + a][b] = Frag<bf16_t>::mma(wf[h][a], xf[kk][b], acc[4 * h + a][b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // bias + ReLU + bf16, 8 channels per (b, j) -> the y image at `img`
        auto store_yimg = [&](unsigned char* img, const float* __restrict__ bias) {
            launder();
            const int cb = wn * 128 + 8 * fchunk;  // this lane's channels: cb + 32 j + {0..7}, j = 0..3 (acc[2j], acc[2j+1])
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias + cb + 32 * j);
                const float4 b1 = *reinterpret_cast<const float4*>(bias + cb + 32 * j + 4);
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int row = wm * 32 + b * 16 + frow;
                    float v[8] = {acc[2 * j][b][0] + b0.x, acc[2 * j][b][1] + b0.y, acc[2 * j][b][2] + b0.z, acc[2 * j][b][3] + b0.w,
                                  acc[2 * j + 1][b][0] + b1.x, acc[2 * j + 1][b][1] + b1.y, acc[2 * j + 1][b][2] + b1.z, acc[2 * j + 1][b][3] + b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    const uint4 pk = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                    *reinterpret_cast<uint4*>(img + yimg_off(row, (cb + 32 * j) >> 3)) = pk;
                }
            }
        };

        // =========================== phase 1: y1 = relu(x W1^T + b1), K = Cin ===========================
        {
            const int nk = Cin >> 6;
            unsigned kbyte = 0;
            auto stage1 = [&](int slot) {
                unsigned char* sa = smem + R2_OFF + slot * 49152;
                unsigned char* sb = sa + 16384;
#pragma unroll
                for (int j = 0; j < 2; ++j) dma16(xg + a_off[j] + kbyte, sa + (wave * 16 + j * 8) * 128);
#pragma unroll
                for (int j = 0; j < 4; ++j) dma16(w1g + (size_t)b_ch[j] * Cin * 2 + b_sw[j] + kbyte, sb + (wave * 32 + j * 8) * 128);
                kbyte += 128;
            };
            launder();
            zero_acc();
            stage1(0);
            for (int kt = 0; kt < nk; ++kt) {
                wait_vm<0>();
                __builtin_amdgcn_s_barrier();   // k-tile kt is in its slot for everybody; slot (kt+1)&1 is free (read in kt-1)
                asm volatile("" ::: "memory");
                if (kt + 1 < nk) stage1((kt + 1) & 1);
                const unsigned char* sa = smem + R2_OFF + (kt & 1) * 49152;
                uint4 xf[2][FM];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int b = 0; b < FM; ++b)
                        xf[kk][b] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * 32 + b * 16 + frow, kk * 4 + fchunk));
                mma_ktile(sa + 16384, xf);
            }
            store_yimg(smem + R0_OFF, p.b1);
        }
        wg_barrier();  // y1 complete; every read of the phase-1 slots is done

        // =========================== phase 2: y2 = relu(conv3x3(y1) + b2): 9 taps x 4 k-tiles ===========================
        {
            constexpr int NT = 36;
            auto stage2 = [&](int t) {  // W2 is OHWI (256, 3, 3, 256): k-tile t = (tap t >> 2, input channels 64 (t & 3) ..)
                unsigned char* sb = smem + R2_OFF + (t % 3) * 32768;
                const unsigned koff = (unsigned)(t >> 2) * 512u + (unsigned)(t & 3) * 128u;
#pragma unroll
                for (int j = 0; j < 4; ++j) dma16(w2g + (size_t)b_ch[j] * (9 * FCM * 2) + b_sw[j] + koff, sb + (wave * 32 + j * 8) * 128);
            };
            launder();
            zero_acc();
            stage2(0);
            stage2(1);
            const unsigned char* y1 = smem + R0_OFF;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                // this lane's two shifted pixel rows and their validity (the frame border is the conv's zero padding)
                int qrow[FM];
                bool qok[FM];
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int pxl = wm * 32 + b * 16 + frow;
                    const int qy = (pxl >> 3) + dy, qx = (pxl & 7) + dx;
                    qok[b] = (unsigned)qy < 16u && (unsigned)qx < 8u;
                    qrow[b] = qok[b] ? qy * 8 + qx : pxl;
                }
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const int t = tap * 4 + kc;
                    if (t + 1 < NT) wait_vm<4>(); else wait_vm<0>();   // tile t landed; tile t+1 may stay in flight
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (t + 2 < NT) stage2(t + 2);
                    uint4 xf[2][FM];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int b = 0; b < FM; ++b) {
                            uint4 v = *reinterpret_cast<const uint4*>(y1 + yimg_off(qrow[b], kc * 8 + kk * 4 + fchunk));
                            if (!qok[b]) v = make_uint4(0u, 0u, 0u, 0u);
                            xf[kk][b] = v;
                        }
                    mma_ktile(smem + R2_OFF + (t % 3) * 32768, xf);
                }
            }
            wg_barrier();  // every weight-slot read (R2|R1) is done: R1 may become the y2 image
            store_yimg(smem + R1_OFF, p.b2);
        }
        wg_barrier();  // y2 complete; y1 (R0) is dead

        // =========================== phase 3: out = relu(y2 W3^T + b3 + x), 4 chunks of 256 channels ===========================
        {
            constexpr int NT = 16;
            auto stage3 = [&](int t) {  // W3 is (Cin, 256): k-tile t = (channel chunk t >> 2, input channels 64 (t & 3) ..)
                unsigned char* sb = smem + R0_OFF + (t % 3) * 32768;
                const int nc = t >> 2;
                const unsigned koff = (unsigned)(t & 3) * 128u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dma16(w3g + (size_t)(nc * 256 + b_ch[j]) * (FCM * 2) + b_sw[j] + koff, sb + (wave * 32 + j * 8) * 128);
            };
            stage3(0);
            stage3(1);
            const unsigned char* y2 = smem + R1_OFF;
            const int nchunks = Cin >> 8;
            const int ntiles = nchunks * 4;
            (void)NT;
            bool stores_pending = false;  // the previous iteration ended with this lane's 8 epilogue stores (younger than tile t+1's DMA)
            for (int nc = 0; nc < nchunks; ++nc) {
                launder();
                const int cb = wn * 128 + 8 * fchunk;
                zero_acc();
                uint4 rres[FM][4];
                // one k-tile of the chunk; `last` (a literal at both call sites) = the chunk's 4th k-tile, which also requests the
                // shortcut x of the chunk -- a whole k-tile before its use, ahead of the next DMA pieces
                auto ktile3 = [&](int kc, bool last) {
                    const int t = nc * 4 + kc;
                    // vmcnt bookkeeping (one in-order counter for loads, DMA and stores): tile t's 4 pieces are the oldest entries;
                    // younger than them: tile t+1's 4 pieces (if any) and, right after an epilogue, its 8 stores
                    if (t + 1 < ntiles) {
                        if (stores_pending) wait_vm<12>(); else wait_vm<4>();
                    } else {
                        if (stores_pending) wait_vm<8>(); else wait_vm<0>();
                    }
                    stores_pending = false;
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (last) {
#pragma unroll
                        for (int b = 0; b < FM; ++b)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                rres[b][j] = *reinterpret_cast<const uint4*>(xg + ((size_t)(wm * 32 + b * 16 + frow) * Cin + nc * 256 + cb + 32 * j) * 2);
                    }
                    if (t + 2 < ntiles) stage3(t + 2);
                    uint4 xf[2][FM];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int b = 0; b < FM; ++b)
                            xf[kk][b] = *reinterpret_cast<const uint4*>(y2 + yimg_off(wm * 32 + b * 16 + frow, kc * 8 + kk * 4 + fchunk));
                    mma_ktile(smem + R0_OFF + (t % 3) * 32768, xf);
                };
#pragma unroll 1
                for (int kc = 0; kc < 3; ++kc) ktile3(kc, false);
                ktile3(3, true);
                // epilogue of the chunk: the 8 shortcut loads are older than tile t+2's pieces (issued after them)
                if (nc * 4 + 5 < ntiles) wait_vm<4>(); else wait_vm<0>();
                launder();
                const int cbe = wn * 128 + 8 * fchunk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.b3 + nc * 256 + cbe + 32 * j);
                    const float4 b1 = *reinterpret_cast<const float4*>(p.b3 + nc * 256 + cbe + 32 * j + 4);
#pragma unroll
                    for (int b = 0; b < FM; ++b) {
                        float v[8] = {acc[2 * j][b][0] + b0.x, acc[2 * j][b][1] + b0.y, acc[2 * j][b][2] + b0.z, acc[2 * j][b][3] + b0.w,
                                      acc[2 * j + 1][b][0] + b1.x, acc[2 * j + 1][b][1] + b1.y, acc[2 * j + 1][b][2] + b1.z, acc[2 * j + 1][b][3] + b1.w};
                        const uint32_t w4[4] = {rres[b][j].x, rres[b][j].y, rres[b][j].z, rres[b][j].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += __uint_as_float(w4[e] << 16);
                            v[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        const uint4 pk = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                        *reinterpret_cast<uint4*>(og + ((size_t)(wm * 32 + b * 16 + frow) * Cin + nc * 256 + cbe + 32 * j) * 2) = pk;
                    }
                }
                stores_pending = true;
            }
        }
        wait_vm<0>();
        wg_barrier();  // the next frame's phase 1 stages into R2|R1 (y2) and writes y1 into R0 (W3 slots)
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
    hipLaunchKernelGGL(bottleneck_frame_kernel, dim3(F < n_cu ? F : n_cu), dim3(512), 160 * 1024, (hipStream_t)stream, p);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_frame");
    return 0;
}
