// Fused tail of a layer-1 Bottleneck and head of the next one (torchreid/models/vmgn.py:45-65), bf16:
//
//     x_out = relu( W3 y2 + b3 + residual )      1x1 conv   64 -> 256, folded BN, the block's output   (written)
//     z     = relu( W1' x_out + b1' )            1x1 conv  256 ->  64 of the NEXT block, folded BN     (written)
//
// Layer 1 works on 64 x 32 maps (524 288 pixels per 256-frame step) with 64 / 256 channels: every conv there is
// HBM-bound (arithmetic intensity 32-170 flop/B), and the next block's first conv re-reads the 268 MB map the previous
// kernel has just written. Here that conv is computed from the output tile while it is still in LDS: the map is
// written once and not read back (-25 % of a block's HBM bytes, one launch less).
//
//   * persistent workgroups (one per CU, 8 waves), tiles of 64 pixels; BOTH weight matrices stay resident in LDS
//     (32 KB + 32 KB) for the whole kernel
//   * per tile only y2 (8 KB) and the residual (32 KB) come in by LDS-DMA, prefetched one tile ahead (double-buffered);
//     the residual tile is combined in place (fp32 math, one rounding) and IS the out tile: drained to HBM as whole
//     16-byte chunks and re-read as the B operand of the second GEMM (row = pixel, 512 B, chunk c at c ^ (row & 31):
//     conflict-free ds_read_b128)
//   * waits are counted: the prefetch DMA sits in front of the tile's stores in the queue, so waiting for it leaves
//     the stores in flight (loads and stores retire in order on one counter)
#include "igemm_dev.h"

namespace {

struct TailParams {
    const void* y2;      // (M, 64)   bf16
    const void* w3;      // (256, 64) bf16, BN folded
    const float* b3;     // (256)
    const void* res;     // (M, 256)  bf16
    void* out;           // (M, 256)  bf16
    const void* w1n;     // (64, 256) bf16, BN folded (next block's conv1)
    const float* b1n;    // (64)
    void* z;             // (M, 64)   bf16
    const void* xs;      // CAT form: (M, 64) bf16 input of the block's 1x1 stride-1 downsample conv (instead of res)
    const void* ws;      // CAT form: (256, 64) bf16 downsample weights, BN folded
    const float* bs;     // CAT form: (256) downsample bias
    int M;
};

constexpr int TBM = 64, TK1 = 64, TN1 = 256, TK2 = 256, TN2 = 64;

// CAT: the residual is the block's own downsample conv (first block of the layer): computed here as a second k-tile,
// out = relu([W3 | Ws] [y2 ; xs] + b3 + bs) -- the 268 MB residual map is neither written nor read.
template <bool CAT>
__global__ __launch_bounds__(512) void bottleneck_tail_kernel(const TailParams p, int ntiles) {
    constexpr int NKT1 = CAT ? 2 : 1;                  // k-tiles of the first GEMM
    constexpr int W3_BYTES = NKT1 * TN1 * 128;         // 32 / 64 KB: [k-tile][256 rows][128 B]
    constexpr int W1_KT = TN2 * 128;                   // one k-tile of W1': 64 rows x 128 B
    constexpr int W1_BYTES = (TK2 / 64) * W1_KT;       // 32 KB: 4 k-tiles
    constexpr int A_BYTES = NKT1 * TBM * 128;          // 8 / 16 KB input tile (its first 8 KB double as the z staging tile)
    constexpr int R_BYTES = TBM * TN1 * 2;             // 32 KB residual / out tile, 512-byte rows
    constexpr int NR = CAT ? 1 : 2;                    // out tile slots: the residual prefetch needs the second one
    __shared__ __attribute__((aligned(16))) unsigned char smem[W3_BYTES + W1_BYTES + 2 * A_BYTES + NR * R_BYTES];
    unsigned char* s_w3 = smem;
    unsigned char* s_w1 = s_w3 + W3_BYTES;
    unsigned char* s_a = s_w1 + W1_BYTES;
    unsigned char* s_r = s_a + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm2 = wave & 1, wn4 = wave >> 1;  // 2 (pixels) x 4 (channels) wave grid for both GEMMs
    const int frow = lane & 15, fchunk = lane >> 4;
    const int lrow = lane >> 3, lchk = lane & 7;
    const int G = gridDim.x;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
    const unsigned char* y2g = reinterpret_cast<const unsigned char*>(p.y2);
    const unsigned char* resg = reinterpret_cast<const unsigned char*>(p.res);

    // ---- resident weights: W3 rows (wave*32 + 8j + lrow), W1' k-tile (j), rows (wave*8 + lrow)
#pragma unroll
    for (int kt = 0; kt < NKT1; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave * 32 + j * 8 + lrow;
            const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(kt == 0 ? p.w3 : p.ws);
            dma16(wsrc + (size_t)row * (TK1 * 2) + ((lchk ^ ((row >> 1) & 7)) << 4), s_w3 + kt * (TN1 * 128) + (wave * 32 + j * 8) * 128);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 8 + lrow;
        dma16(reinterpret_cast<const unsigned char*>(p.w1n) + (size_t)row * (TK2 * 2) + j * 128 + ((lchk ^ ((row >> 1) & 7)) << 4),
              s_w1 + j * W1_KT + wave * 8 * 128);
    }
    // ---- per-tile DMA: y2 piece = rows 8 wave .. +7; residual pieces = rows 2 (4 wave + j) + (lane >> 5)
    auto stage_tile = [&](int T, int slot) {
        const int m0 = T * TBM;
        {
            const int row = wave * 8 + lrow;
            const int gm = m0 + row;
            dma16(gm < p.M ? y2g + (size_t)gm * (TK1 * 2) + ((lchk ^ ((row >> 1) & 7)) << 4) : zsrc,
                  s_a + slot * A_BYTES + wave * 1024);
            if constexpr (CAT)
                dma16(gm < p.M ? reinterpret_cast<const unsigned char*>(p.xs) + (size_t)gm * (TK1 * 2) + ((lchk ^ ((row >> 1) & 7)) << 4) : zsrc,
                      s_a + slot * A_BYTES + TBM * 128 + wave * 1024);
        }
        if constexpr (!CAT)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = wave * 4 + j;
            const int row = piece * 2 + (lane >> 5);
            const int gm = m0 + row;
            const int gch = (lane & 31) ^ (row & 31);
            dma16(gm < p.M ? resg + ((size_t)gm * TN1 + gch * 8) * 2 : zsrc, s_r + slot * R_BYTES + piece * 1024);
        }
    };
    // biases of this lane's output channels (fixed for the whole kernel)
    float4 b3v[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        b3v[a] = *reinterpret_cast<const float4*>(p.b3 + wn4 * 64 + a * 16 + fchunk * 4);
        if constexpr (CAT) {
            const float4 t = *reinterpret_cast<const float4*>(p.bs + wn4 * 64 + a * 16 + fchunk * 4);
            b3v[a].x += t.x; b3v[a].y += t.y; b3v[a].z += t.z; b3v[a].w += t.w;
        }
    }
    const float4 b1v = *reinterpret_cast<const float4*>(p.b1n + wn4 * 16 + fchunk * 4);

    int T = blockIdx.x;
    if (T < ntiles) stage_tile(T, 0);
    wait_vmcnt<0>();
    wg_barrier();
    int slot = 0;
    for (; T < ntiles; T += G, slot ^= 1) {
        const int m0 = T * TBM;
        const bool has_next = T + G < ntiles;
        if (has_next) stage_tile(T + G, slot ^ 1);  // 5 (CAT: 2) DMA pieces, at the head of this iteration's queue
        const unsigned char* sa = s_a + slot * A_BYTES;
        unsigned char* sr = s_r + (CAT ? 0 : slot) * R_BYTES;
        // ---- GEMM 1: 64 px x 256 ch, K = 64. Wave tile 32 px x 64 ch.
        f32x4_t acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2 * NKT1; ++kk) {
            uint4 xf[2], wf[4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
                xf[b] = *reinterpret_cast<const uint4*>(sa + (kk >> 1) * (TBM * 128) + lds_off(wm2 * 32 + b * 16 + frow, (kk & 1) * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < 4; ++a)
                wf[a] = *reinterpret_cast<const uint4*>(s_w3 + (kk >> 1) * (TN1 * 128) + lds_off(wn4 * 64 + a * 16 + frow, (kk & 1) * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = Frag<lp16_t>::mma(wf[a], xf[b], acc[a][b]);
        }
        // bias + residual (in place) + ReLU -> bf16 out tile
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = wm2 * 32 + b * 16 + frow;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = wn4 * 64 + a * 16 + fchunk * 4;
                unsigned char* cell = sr + px * 512 + (((c >> 3) ^ (px & 31)) << 4) + ((c & 4) << 1);
                float rr[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (!CAT) load4<lp16_t>(reinterpret_cast<const lp16_t*>(cell), rr);
                float v[4];
                v[0] = relu_nan(acc[a][b][0] + b3v[a].x + rr[0]);
                v[1] = relu_nan(acc[a][b][1] + b3v[a].y + rr[1]);
                v[2] = relu_nan(acc[a][b][2] + b3v[a].z + rr[2]);
                v[3] = relu_nan(acc[a][b][3] + b3v[a].w + rr[3]);
                store4<lp16_t>(reinterpret_cast<lp16_t*>(cell), v);
            }
        }
        wg_barrier();  // out tile complete
        // drain the out tile: 2048 16-byte chunks, 4 per thread, whole 512-byte rows per 32 lanes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 5) + 16 * i;
            const int pch = tid & 31;
            const int gm = m0 + row;
            if (gm < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(sr + row * 512 + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + ((size_t)gm * TN1 + (pch ^ (row & 31)) * 8) * 2) = v;
            }
        }
        // ---- GEMM 2: 64 px x 64 ch, K = 256, B operand straight from the out tile. Wave tile 32 px x 16 ch.
        f32x4_t acc2[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < TK2 / 32; ++ks) {
            const uint4 wf = *reinterpret_cast<const uint4*>(s_w1 + (ks >> 1) * W1_KT + lds_off(wn4 * 16 + frow, (ks & 1) * 4 + fchunk));
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int px = wm2 * 32 + b * 16 + frow;
                const uint4 xf = *reinterpret_cast<const uint4*>(sr + px * 512 + (((ks * 4 + fchunk) ^ (px & 31)) << 4));
                acc2[b] = Frag<lp16_t>::mma(wf, xf, acc2[b]);
            }
        }
        // z tile -> the y2 slot of this tile (every wave finished reading it before the barrier above)
        unsigned char* sz = s_a + slot * A_BYTES;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = wm2 * 32 + b * 16 + frow;
            const int c = wn4 * 16 + fchunk * 4;
            float v[4];
            v[0] = relu_nan(acc2[b][0] + b1v.x);
            v[1] = relu_nan(acc2[b][1] + b1v.y);
            v[2] = relu_nan(acc2[b][2] + b1v.z);
            v[3] = relu_nan(acc2[b][3] + b1v.w);
            store4<lp16_t>(reinterpret_cast<lp16_t*>(sz + px * 128 + (((c >> 3) ^ (px & 7)) << 4) + ((c & 4) << 1)), v);
        }
        wg_barrier();  // z tile complete; every read of the out tile done
        {
            const int row = tid >> 3, pch = tid & 7;
            const int gm = m0 + row;
            if (gm < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(sz + row * 128 + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.z) + ((size_t)gm * TN2 + (pch ^ (row & 7)) * 8) * 2) = v;
            }
        }
        // the next tile's DMA (issued first) has landed once at most the 5 stores above are still outstanding
        wait_vmcnt<5>();
        wg_barrier();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Layer transition form: the next block's conv1 has 128 output channels (layer1 -> layer2, 256 -> 128). Its 64 KB of
// weights leave room for ONE out tile only, so the residual tile is fetched at the top of its own tile (under the
// first GEMM's MFMAs) instead of one tile ahead; the y2 tile is still prefetched. z (64 px x 128 ch) is staged over the
// out tile once every wave has finished reading it.
__global__ __launch_bounds__(512) void bottleneck_tail128_kernel(const TailParams p, int ntiles) {
    constexpr int CN = 128;
    constexpr int W3_BYTES = TN1 * 128, W1_KT = CN * 128, W1_BYTES = (TK2 / 64) * W1_KT;  // 32 KB, 64 KB
    constexpr int A_BYTES = TBM * 128, R_BYTES = TBM * TN1 * 2;                            // 8 KB, 32 KB
    __shared__ __attribute__((aligned(16))) unsigned char smem[W3_BYTES + W1_BYTES + 2 * A_BYTES + R_BYTES];
    unsigned char* s_w3 = smem;
    unsigned char* s_w1 = s_w3 + W3_BYTES;
    unsigned char* s_a = s_w1 + W1_BYTES;
    unsigned char* sr = s_a + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm2 = wave & 1, wn4 = wave >> 1;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int lrow = lane >> 3, lchk = lane & 7;
    const int G = gridDim.x;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
    const unsigned char* y2g = reinterpret_cast<const unsigned char*>(p.y2);
    const unsigned char* resg = reinterpret_cast<const unsigned char*>(p.res);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + lrow;
        dma16(reinterpret_cast<const unsigned char*>(p.w3) + (size_t)row * (TK1 * 2) + ((lchk ^ ((row >> 1) & 7)) << 4),
              s_w3 + (wave * 32 + j * 8) * 128);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = (wave + 8 * h) * 8 + lrow;
            dma16(reinterpret_cast<const unsigned char*>(p.w1n) + (size_t)row * (TK2 * 2) + kt * 128 + ((lchk ^ ((row >> 1) & 7)) << 4),
                  s_w1 + kt * W1_KT + (wave + 8 * h) * 8 * 128);
        }
    auto stage_y2 = [&](int T, int slot) {
        const int row = wave * 8 + lrow;
        const int gm = T * TBM + row;
        dma16(gm < p.M ? y2g + (size_t)gm * (TK1 * 2) + ((lchk ^ ((row >> 1) & 7)) << 4) : zsrc, s_a + slot * A_BYTES + wave * 1024);
    };
    auto stage_res = [&](int T) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = wave * 4 + j;
            const int row = piece * 2 + (lane >> 5);
            const int gm = T * TBM + row;
            const int gch = (lane & 31) ^ (row & 31);
            dma16(gm < p.M ? resg + ((size_t)gm * TN1 + gch * 8) * 2 : zsrc, sr + piece * 1024);
        }
    };
    float4 b3v[4], b1v[2];
#pragma unroll
    for (int a = 0; a < 4; ++a) b3v[a] = *reinterpret_cast<const float4*>(p.b3 + wn4 * 64 + a * 16 + fchunk * 4);
#pragma unroll
    for (int a = 0; a < 2; ++a) b1v[a] = *reinterpret_cast<const float4*>(p.b1n + wn4 * 32 + a * 16 + fchunk * 4);

    int T = blockIdx.x;
    if (T < ntiles) stage_y2(T, 0);
    wait_vmcnt<0>();
    wg_barrier();
    int slot = 0;
    for (; T < ntiles; T += G, slot ^= 1) {
        const int m0 = T * TBM;
        const bool has_next = T + G < ntiles;
        stage_res(T);                               // 4 pieces, oldest in this iteration's queue
        if (has_next) stage_y2(T + G, slot ^ 1);    // 1 piece
        const unsigned char* sa = s_a + slot * A_BYTES;
        f32x4_t acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 xf[2], wf[4];
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(sa + lds_off(wm2 * 32 + b * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const uint4*>(s_w3 + lds_off(wn4 * 64 + a * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = Frag<lp16_t>::mma(wf[a], xf[b], acc[a][b]);
        }
        if (has_next) wait_vmcnt<1>();  // the residual pieces are older than the y2 prefetch
        else wait_vmcnt<0>();
        wg_barrier();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = wm2 * 32 + b * 16 + frow;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = wn4 * 64 + a * 16 + fchunk * 4;
                unsigned char* cell = sr + px * 512 + (((c >> 3) ^ (px & 31)) << 4) + ((c & 4) << 1);
                float rr[4];
                load4<lp16_t>(reinterpret_cast<const lp16_t*>(cell), rr);
                float v[4];
                v[0] = relu_nan(acc[a][b][0] + b3v[a].x + rr[0]);
                v[1] = relu_nan(acc[a][b][1] + b3v[a].y + rr[1]);
                v[2] = relu_nan(acc[a][b][2] + b3v[a].z + rr[2]);
                v[3] = relu_nan(acc[a][b][3] + b3v[a].w + rr[3]);
                store4<lp16_t>(reinterpret_cast<lp16_t*>(cell), v);
            }
        }
        wg_barrier();  // out tile complete
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 5) + 16 * i;
            const int pch = tid & 31;
            const int gm = m0 + row;
            if (gm < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(sr + row * 512 + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + ((size_t)gm * TN1 + (pch ^ (row & 31)) * 8) * 2) = v;
            }
        }
        // GEMM 2: 64 px x 128 ch, K = 256. Wave tile 32 px x 32 ch.
        f32x4_t acc2[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < TK2 / 32; ++ks) {
            uint4 wf[2], xf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                wf[a] = *reinterpret_cast<const uint4*>(s_w1 + (ks >> 1) * W1_KT + lds_off(wn4 * 32 + a * 16 + frow, (ks & 1) * 4 + fchunk));
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int px = wm2 * 32 + b * 16 + frow;
                xf[b] = *reinterpret_cast<const uint4*>(sr + px * 512 + (((ks * 4 + fchunk) ^ (px & 31)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc2[a][b] = Frag<lp16_t>::mma(wf[a], xf[b], acc2[a][b]);
        }
        wg_barrier();  // every read of the out tile (drain + GEMM 2) is done: stage z over it, 256-byte rows
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = wm2 * 32 + b * 16 + frow;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int c = wn4 * 32 + a * 16 + fchunk * 4;
                float v[4];
                v[0] = relu_nan(acc2[a][b][0] + b1v[a].x);
                v[1] = relu_nan(acc2[a][b][1] + b1v[a].y);
                v[2] = relu_nan(acc2[a][b][2] + b1v[a].z);
                v[3] = relu_nan(acc2[a][b][3] + b1v[a].w);
                store4<lp16_t>(reinterpret_cast<lp16_t*>(sr + px * 256 + (((c >> 3) ^ (px & 15)) << 4) + ((c & 4) << 1)), v);
            }
        }
        wg_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = tid >> 3, pch = (tid & 7) * 2 + i;
            const int gm = m0 + row;
            if (gm < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(sr + row * 256 + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.z) + ((size_t)gm * CN + (pch ^ (row & 15)) * 8) * 2) = v;
            }
        }
        wait_vmcnt<6>();  // the y2 prefetch is older than this tile's 6 stores
        wg_barrier();     // ... and nobody still reads the z tile when the next residual lands on it
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Layer-2 form: conv3 128 -> 512 + residual + ReLU, then the next block's conv1 512 -> 128 (32 x 16 maps, 131 072
// pixels per 256-frame step). The two weight matrices are 128 KB each -- they do not fit the LDS beside the tiles --
// but a wave only ever needs ITS output channels of them: with the eight waves splitting the channels (64 of conv3,
// 16 of the next conv1) each lane keeps its 16 + 16 MFMA weight fragments in 128 VGPRs for the whole kernel, and the
// LDS holds nothing but tiles: y2 (16 KB) and the residual / out tile (64 KB), both double-buffered = 160 KB.
// Per 64-pixel tile the kernel moves 160 KB of HBM traffic (y2, residual in; out, z out) for 2 x 64 MFMAs per wave:
// HBM-bound, like the layer-1 form.
constexpr int L2K1 = 128, L2N1 = 512, L2K2 = 512, L2N2 = 128;

__global__ __launch_bounds__(512) void bottleneck_tail_l2_kernel(const TailParams p, int ntiles) {
    constexpr int A_BYTES = 2 * TBM * 128;      // 16 KB: two 64-channel k-tiles of [64 rows][128 B]
    constexpr int R_BYTES = TBM * L2N1 * 2;     // 64 KB: 1024-byte rows, chunk c at c ^ (row & 31)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 2 * R_BYTES];
    unsigned char* s_a = smem;
    unsigned char* s_r = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int G = gridDim.x;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
    const unsigned char* y2g = reinterpret_cast<const unsigned char*>(p.y2);
    const unsigned char* resg = reinterpret_cast<const unsigned char*>(p.res);

    // ---- this wave's weight fragments, resident in registers: conv3 channels 64 wave + 16 a + frow, k = 32 kk + 8 fchunk;
    // next conv1 channel 16 wave + frow, k = 32 ks + 8 fchunk (the MFMA A operand: row = lane & 15, k-chunk = lane >> 4)
    uint4 w3f[4][4], w1f[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            w3f[a][kk] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.w3) +
                                                         ((size_t)(wave * 64 + a * 16 + frow) * L2K1 + kk * 32 + fchunk * 8) * 2);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
        w1f[ks] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.w1n) +
                                                  ((size_t)(wave * 16 + frow) * L2K2 + ks * 32 + fchunk * 8) * 2);
    // biases: one value per lane (lane L: conv3 channel 64 wave + L, next conv1 channel 16 wave + (L & 15)), fetched
    // through the cross-lane network where they are used -- 2 registers instead of 20
    float b3l = p.b3[wave * 64 + lane];
    float b1l = p.b1n[wave * 16 + (lane & 15)];
    asm volatile("" : "+v"(b3l), "+v"(b1l));
    auto lane_bias = [&](float held, int src_lane) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(held)));
    };
    // opaque from here on: the compiler must keep them in registers rather than re-load them inside the tile loop (its
    // own loads come with vmcnt waits that would drain the DMA queue)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            asm volatile("" : "+v"(w3f[a][kk].x), "+v"(w3f[a][kk].y), "+v"(w3f[a][kk].z), "+v"(w3f[a][kk].w));
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(w1f[ks].x), "+v"(w1f[ks].y), "+v"(w1f[ks].z), "+v"(w1f[ks].w));

    // ---- per-tile DMA: 2 y2 pieces (rows 8 wave .. +7 of both k-tiles) + 8 residual pieces (whole rows 8 wave + j)
    auto stage_tile = [&](int T, int slot) {
        const int m0 = T * TBM;
        int ln = lane;
        asm volatile("" : "+v"(ln));  // addresses are recomputed per tile: hoisted out of the loop they end up in scratch
        {
            const int row = wave * 8 + (ln >> 3);
            const int gm = m0 + row;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
                dma16(gm < p.M ? y2g + (size_t)gm * (L2K1 * 2) + kt * 128 + (((ln & 7) ^ ((row >> 1) & 7)) << 4) : zsrc,
                      s_a + slot * A_BYTES + kt * (TBM * 128) + wave * 1024);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = wave * 8 + j;
            const int gm = m0 + row;
            dma16(gm < p.M ? resg + (size_t)gm * (L2N1 * 2) + ((ln ^ (row & 31)) << 4) : zsrc, s_r + slot * R_BYTES + row * 1024);
        }
    };

    int T = blockIdx.x;
    if (T < ntiles) stage_tile(T, 0);
    wait_vmcnt<0>();
    wg_barrier();
    int slot = 0;
    for (; T < ntiles; T += G, slot ^= 1) {
        const int m0 = T * TBM;
        if (T + G < ntiles) stage_tile(T + G, slot ^ 1);  // 10 DMA pieces at the head of this iteration's queue
        const unsigned char* sa = s_a + slot * A_BYTES;
        unsigned char* sr = s_r + slot * R_BYTES;
        // ---- GEMM 1: 64 px x 512 ch, K = 128. Wave tile 64 px x 64 ch, weights from registers; two passes of 32
        // channels (32 accumulator registers at a time beside the 128 weight registers), each followed by its
        // bias + residual (in place) + ReLU -> bf16 out tile
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
            f32x4_t acc[2][4];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            uint4 xp[2][4];
            auto ld1 = [&](int kk, int b) {
                return *reinterpret_cast<const uint4*>(sa + (kk >> 1) * (TBM * 128) + lds_off(b * 16 + frow, (kk & 1) * 4 + fchunk));
            };
#pragma unroll
            for (int b = 0; b < 4; ++b) xp[0][b] = ld1(0, b);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) xp[(kk + 1) & 1][b] = ld1(kk + 1, b);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = Frag<lp16_t>::mma(w3f[2 * ah + a][kk], xp[kk & 1][b], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
            }
            float bv[2][4];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[a][r] = lane_bias(b3l, (2 * ah + a) * 16 + fchunk * 4 + r);
            // all eight residual cells are read before the first is written back (a read behind a write to the same array is
            // not hoisted by the compiler: eight LDS round trips in a row otherwise)
            uint2 rcell[4][2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int px = b * 16 + frow;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int c = wave * 64 + (2 * ah + a) * 16 + fchunk * 4;
                    rcell[b][a] = *reinterpret_cast<const uint2*>(sr + px * 1024 + (((c >> 3) ^ (px & 31)) << 4) + ((c & 4) << 1));
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int px = b * 16 + frow;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int c = wave * 64 + (2 * ah + a) * 16 + fchunk * 4;
                    unsigned char* cell = sr + px * 1024 + (((c >> 3) ^ (px & 31)) << 4) + ((c & 4) << 1);
                    float rr[4];
                    unpack_lp16x2(rcell[b][a].x, rr[0], rr[1]);
                    unpack_lp16x2(rcell[b][a].y, rr[2], rr[3]);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = relu_nan(acc[a][b][r] + bv[a][r] + rr[r]);
                    store4<lp16_t>(reinterpret_cast<lp16_t*>(cell), v);
                }
            }
        }
        wg_barrier();  // out tile complete; every read of the y2 tile done
        // drain the out tile: 4096 16-byte chunks, 8 per thread, one whole 1024-byte row per wave instruction
        int ld = lane, td = tid;
        asm volatile("" : "+v"(ld), "+v"(td));  // store addresses recomputed per tile (see stage_tile)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = wave + 8 * i;
            const int gm = m0 + row;
            if (gm < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(sr + row * 1024 + (ld << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + ((size_t)gm * L2N1 + (ld ^ (row & 31)) * 8) * 2) = v;
            }
        }
        // ---- GEMM 2: 64 px x 128 ch, K = 512, B operand straight from the out tile. Wave tile 64 px x 16 ch.
        f32x4_t acc2[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc2[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // fragment reads one k-step ahead of the MFMAs; sched_barrier keeps the compiler from hoisting all 64 reads
        uint4 xq[2][4];
        auto ld2 = [&](int ks, int b) {
            const int px = b * 16 + frow;
            return *reinterpret_cast<const uint4*>(sr + px * 1024 + (((ks * 4 + fchunk) ^ (px & 31)) << 4));
        };
#pragma unroll
        for (int b = 0; b < 4; ++b) xq[0][b] = ld2(0, b);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + 1 < 16) {
#pragma unroll
                for (int b = 0; b < 4; ++b) xq[(ks + 1) & 1][b] = ld2(ks + 1, b);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) acc2[b] = Frag<lp16_t>::mma(w1f[ks], xq[ks & 1][b], acc2[b]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // z tile (64 px x 128 ch, 256-byte rows, chunk c at c ^ (row & 15)) over this tile's y2 slot
        unsigned char* sz = s_a + slot * A_BYTES;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int px = b * 16 + frow;
            const int c = wave * 16 + fchunk * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = relu_nan(acc2[b][r] + lane_bias(b1l, fchunk * 4 + r));
            store4<lp16_t>(reinterpret_cast<lp16_t*>(sz + px * 256 + (((c >> 3) ^ (px & 15)) << 4) + ((c & 4) << 1)), v);
        }
        wg_barrier();  // z tile complete; every read of the out tile done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (td >> 4) + 32 * i, pch = td & 15;
            const int gm = m0 + row;
            if (gm < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(sz + row * 256 + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.z) + ((size_t)gm * L2N2 + (pch ^ (row & 15)) * 8) * 2) = v;
            }
        }
        // the next tile's DMA (issued first) has landed once at most the 10 stores above are still outstanding (a ragged
        // tile issues fewer, but it is the last tile of its workgroup: nothing is in flight then)
        wait_vmcnt<10>();
        wg_barrier();
    }
}

}  // namespace

extern "C" int agrl_bottleneck_tail(const void* y2, const void* w3, const float* b3, const void* residual,
                                    const void* x_short, const void* w_short, const float* b_short, void* out,
                                    const void* w1_next, const float* b1_next, void* z, int M, int Cmid, int Cout,
                                    int Cnext, int Cshort, agrl_stream_t stream) {
    AGRL_CHECK_ARG(y2 && w3 && b3 && out && w1_next && b1_next && z, "agrl_bottleneck_tail: null pointer");
    AGRL_CHECK_ARG((residual != nullptr) != (x_short != nullptr),
                   "agrl_bottleneck_tail: pass either the residual map or the downsample conv's input, not both");
    AGRL_CHECK_ARG(!x_short || (w_short && b_short), "agrl_bottleneck_tail: the downsample form needs its weights and bias");
    AGRL_CHECK_ARG(M > 0, "agrl_bottleneck_tail: empty problem");
    const bool layer2 = Cmid == L2K1 && Cout == L2N1 && Cnext == L2N2 && !x_short;
    AGRL_CHECK_ARG(layer2 || (Cmid == TK1 && Cout == TN1 && (Cnext == TN2 || (Cnext == 128 && !x_short)) && (!x_short || Cshort == TK1)),
                   "agrl_bottleneck_tail: built for Cmid=64, Cout=256, Cnext=64 (128 without downsample), Cshort=64 "
                   "(layer 1) and Cmid=128, Cout=512, Cnext=128 without downsample (layer 2), got %d/%d/%d/%d", Cmid, Cout, Cnext, Cshort);
    const uintptr_t al = (uintptr_t)y2 | (uintptr_t)w3 | (uintptr_t)b3 | (uintptr_t)residual | (uintptr_t)out |
                         (uintptr_t)w1_next | (uintptr_t)b1_next | (uintptr_t)z | (uintptr_t)x_short | (uintptr_t)w_short |
                         (uintptr_t)b_short;
    AGRL_CHECK_ARG((al & 15) == 0, "agrl_bottleneck_tail: pointers must be 16-byte aligned");
    TailParams p;
    p.y2 = y2; p.w3 = w3; p.b3 = b3; p.res = residual; p.out = out; p.w1n = w1_next; p.b1n = b1_next; p.z = z; p.M = M;
    p.xs = x_short; p.ws = w_short; p.bs = b_short;
    const int ntiles = cdiv(M, TBM);
    const int grid = ntiles < 256 ? ntiles : 256;
    if (layer2) hipLaunchKernelGGL(bottleneck_tail_l2_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    else if (Cnext == 128) hipLaunchKernelGGL(bottleneck_tail128_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    else if (x_short) hipLaunchKernelGGL(bottleneck_tail_kernel<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    else hipLaunchKernelGGL(bottleneck_tail_kernel<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_tail");
    return 0;
}
