// Whole layer-1 Bottleneck body behind its first conv, fused with the head of the next block (bf16;
// torchreid/models/vmgn.py:45-65):
//
//     y2    = relu( W2 (*) z + b2 )               3x3 conv   64 ->  64, stride 1, pad 1 (z = this block's conv1 output)
//     x_out = relu( W3 y2 + b3 + R )              1x1 conv   64 -> 256 + shortcut R, the block's output      (written)
//     z'    = relu( W1' x_out + b1' )             1x1 conv  256 ->  64 / 128 of the NEXT block               (written)
//
// R is the residual map (identity shortcut) or, in the first block, Ws x + bs computed in the same pass (CAT form).
// Layer 1 is HBM-bound (64 x 32 maps, 524 288 pixels per 256-frame step): the separate 3x3 kernel writes the 67 MB y2
// map only for the tail kernel to read it back. Here y2 never leaves the CU: per 8 x 8 pixel tile the kernel reads a
// 10 x 10 halo patch of z (12.5 KB) and the residual tile (32 KB), and writes x_out (32 KB) and z' (8 / 16 KB).
//
//   * persistent workgroups (one per CU, 8 waves); the nine taps of W2 stay resident in LDS (72 KB), W3 / Ws / W1' as MFMA
//     fragments in registers (each wave only needs its own output channels: 32 + 32 [+ 32] VGPRs)
//   * 3x3 sweep straight from the patch with the conflict-free column permutation (frag_px / patch_off, igemm_dev.h)
//   * the next tile's patch is prefetched one tile ahead (LDS-DMA, double-buffered); the residual tile is requested at the
//     top of its own tile and lands under the 3x3 sweep; waits are counted (loads and stores retire in order)
#include "igemm_dev.h"

namespace {

struct BlockParams {
    const void* z;     // (F, H, W, 64)  bf16: conv1 output of this block
    const void* w2;    // (64, 3, 3, 64) bf16 OHWI, BN folded
    const float* b2;   // (64)
    const void* w3;    // (256, 64)      bf16
    const float* b3;   // (256)
    const void* res;   // (F, H, W, 256) bf16 residual map (identity shortcut) or null (CAT)
    const void* xs;    // CAT: (F, H, W, 64) bf16 input of the block's 1x1 stride-1 downsample conv
    const void* ws;    // CAT: (256, 64) bf16
    const float* bs;   // CAT: (256)
    void* out;         // (F, H, W, 256) bf16
    const void* w1n;   // (CN, 256) bf16: the next block's conv1
    const float* b1n;  // (CN)
    void* zn;          // (F, H, W, CN) bf16
    int F, H, W;
};

template <bool CAT, int CN>
__global__ __launch_bounds__(512) void bottleneck_block_kernel(const BlockParams p, int ntiles) {
    static_assert(CN == 64 || (CN == 128 && !CAT), "next conv1: 64 channels, or 128 with the identity shortcut");
    constexpr int NW = 8, PW = 10, PPIX = 100, PPIECES = 13, PATCH_BYTES = PPIECES * 1024, PJ = 2;
    constexpr int W_TAP = 64 * 128, W2_BYTES = 9 * W_TAP;  // 72 KB
    constexpr int Y_BYTES = 64 * 128;                      // y2 tile (and the z' staging tile when CN == 64)
    constexpr int X_BYTES = CAT ? 64 * 128 : 0;            // CAT: the shortcut conv's input tile
    constexpr int R_BYTES = 64 * 512;                      // residual / out tile, 512-byte rows, chunk c at c ^ (row & 31)
    constexpr int NST = CN == 64 ? 5 : 6;                  // global stores per thread and tile (4 out + 1 or 2 z')
    constexpr int NA2 = CN / 64;                           // 16-channel fragments of the next conv1 per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[W2_BYTES + 2 * PATCH_BYTES + Y_BYTES + X_BYTES + R_BYTES + 1024];
    unsigned char* s_w2 = smem;
    unsigned char* s_p = s_w2 + W2_BYTES;
    unsigned char* s_y = s_p + 2 * PATCH_BYTES;
    unsigned char* s_x = s_y + Y_BYTES;
    unsigned char* s_r = s_x + X_BYTES;
    unsigned char* s_dummy = s_r + R_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int lrow = lane >> 3, lchk = lane & 7;
    const int G = gridDim.x;
    const int tw = p.W >> 3, th = p.H >> 3;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
    const unsigned char* zg = reinterpret_cast<const unsigned char*>(p.z);
    const unsigned char* resg = reinterpret_cast<const unsigned char*>(p.res);
    const unsigned char* xsg = reinterpret_cast<const unsigned char*>(p.xs);

    // ---- resident 3x3 weights: tap t, rows 8 wave .. +7 (K order of the OHWI weight = (tap, cin))
    {
        const int row = wave * 8 + lrow;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            dma16(reinterpret_cast<const unsigned char*>(p.w2) + ((size_t)row * 576 + t * 64) * 2 + ((lchk ^ ((row >> 1) & 7)) << 4),
                  s_w2 + t * W_TAP + wave * 8 * 128);
    }
    // ---- 1x1 weights as MFMA A fragments in registers. GEMM 1 (64 px x 256 ch): wave grid 2 (px) x 4 (ch), wave tile
    // 32 px x 64 ch; GEMM 2 (64 px x CN ch, K = 256): same grid, wave tile 32 px x CN/4 ch
    const int wm2 = wave & 1, wn4 = wave >> 1;
    uint4 w3f[4][2], wsf[CAT ? 4 : 1][2], w1f[NA2][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const size_t o = ((size_t)(wn4 * 64 + a * 16 + frow) * 64 + kk * 32 + fchunk * 8) * 2;
            w3f[a][kk] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.w3) + o);
            if constexpr (CAT) wsf[a][kk] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.ws) + o);
        }
#pragma unroll
    for (int a = 0; a < NA2; ++a)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            w1f[a][ks] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.w1n) +
                                                         ((size_t)(wn4 * (CN / 4) + a * 16 + frow) * 256 + ks * 32 + fchunk * 8) * 2);
    // biases of this lane's output channels
    const int wm = wave & 3, wn = wave >> 2;  // GEMM 0 (3x3, 64 px x 64 ch): wave grid 4 (px) x 2 (ch), wave tile 16 px x 32 ch
    float4 b2v[2], b3v[4], b1v[NA2];
#pragma unroll
    for (int a = 0; a < 2; ++a) b2v[a] = *reinterpret_cast<const float4*>(p.b2 + wn * 32 + a * 16 + fchunk * 4);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        b3v[a] = *reinterpret_cast<const float4*>(p.b3 + wn4 * 64 + a * 16 + fchunk * 4);
        if constexpr (CAT) {
            const float4 t = *reinterpret_cast<const float4*>(p.bs + wn4 * 64 + a * 16 + fchunk * 4);
            b3v[a].x += t.x; b3v[a].y += t.y; b3v[a].z += t.z; b3v[a].w += t.w;
        }
    }
#pragma unroll
    for (int a = 0; a < NA2; ++a) b1v[a] = *reinterpret_cast<const float4*>(p.b1n + wn4 * (CN / 4) + a * 16 + fchunk * 4);
    // pin everything loaded so far: re-loading inside the tile loop would come with vmcnt(0) waits that drain the DMA queue
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            asm volatile("" : "+v"(w3f[a][kk].x), "+v"(w3f[a][kk].y), "+v"(w3f[a][kk].z), "+v"(w3f[a][kk].w));
            if constexpr (CAT) asm volatile("" : "+v"(wsf[a][kk].x), "+v"(wsf[a][kk].y), "+v"(wsf[a][kk].z), "+v"(wsf[a][kk].w));
        }
#pragma unroll
    for (int a = 0; a < NA2; ++a)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(w1f[a][ks].x), "+v"(w1f[a][ks].y), "+v"(w1f[a][ks].z), "+v"(w1f[a][ks].w));
#pragma unroll
    for (int a = 0; a < 4; ++a) asm volatile("" : "+v"(b3v[a].x), "+v"(b3v[a].y), "+v"(b3v[a].z), "+v"(b3v[a].w));
#pragma unroll
    for (int a = 0; a < 2; ++a) asm volatile("" : "+v"(b2v[a].x), "+v"(b2v[a].y), "+v"(b2v[a].z), "+v"(b2v[a].w));
#pragma unroll
    for (int a = 0; a < NA2; ++a) asm volatile("" : "+v"(b1v[a].x), "+v"(b1v[a].y), "+v"(b1v[a].z), "+v"(b1v[a].w));

    // per-lane parts of the residual-tile and out-tile global offsets (pixel (row >> 3, row & 7) of the tile and the
    // swizzled chunk), computed ONCE; per tile only the tile's base pixel (uniform) is added
    unsigned res_lane[4], out_lane[CN == 64 ? 4 : 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rrow = (wave * 4 + j) * 2 + (lane >> 5);
        res_lane[j] = (unsigned)(((rrow >> 3) * p.W + (rrow & 7)) * 512 + (((lane & 31) ^ (rrow & 31)) << 4));
        const int orow = (tid >> 5) + 16 * j;
        if constexpr (CN == 64) out_lane[j] = (unsigned)(((orow >> 3) * p.W + (orow & 7)) * 512 + (((tid & 31) ^ (orow & 31)) << 4));
    }

    auto tile_origin = [&](int T, int& img, int& oy0, int& ox0) {
        img = T / (tw * th);
        const int trem = T - img * (tw * th);
        oy0 = (trem / tw) << 3;
        ox0 = (trem % tw) << 3;
    };
    // global pixel index of tile pixel q (row-major 8 x 8)
    auto gpix = [&](int img, int oy0, int ox0, int q) { return ((size_t)img * p.H + oy0 + (q >> 3)) * p.W + ox0 + (q & 7); };

    auto stage_patch = [&](int T, int buf) {  // 2 DMA pieces per wave (13 real ones + 3 dummies)
        int img, oy0, ox0;
        tile_origin(T, img, oy0, ox0);
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < PJ; ++i) {
            const int piece = wave + NW * i;
            const int row = piece * 8 + (ln >> 3);
            const int py = row / PW, px = row - py * PW;
            const int iy = oy0 + py - 1, ix = ox0 + px - 1;
            const bool ok = piece < PPIECES && row < PPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            dma16(ok ? zg + (((size_t)img * p.H + iy) * p.W + ix) * 128 + (((ln & 7) ^ patch_g(py, px)) << 4) : zsrc,
                  piece < PPIECES ? s_p + buf * PATCH_BYTES + piece * 1024 : s_dummy);
        }
    };
    auto stage_shortcut = [&](int T) {  // residual: 4 pieces per wave (two 512-byte rows each); CAT: 1 piece (8 rows of x)
        int img, oy0, ox0;
        tile_origin(T, img, oy0, ox0);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if constexpr (CAT) {
            const int row = wave * 8 + (ln >> 3);
            dma16(xsg + gpix(img, oy0, ox0, row) * 128 + (((ln & 7) ^ ((row >> 1) & 7)) << 4), s_x + wave * 1024);
        } else {
            const unsigned char* tbase = resg + gpix(img, oy0, ox0, 0) * 512;  // uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16(tbase + res_lane[j], s_r + (wave * 4 + j) * 1024);
        }
    };

    // LDS byte offsets of the 3x3 sweep, computed ONCE: the swizzles make them 18 + 4 different per-lane values, and
    // recomputing them per tile (as the other phases do to stay out of scratch) was a quarter of the kernel's VALU work
    int xoff[18], woff[2][2];
    {
        const int q = frag_px(frow);
        const int py0 = 2 * wm + (q >> 3), px0 = q & 7;
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            const int t = st >> 1, kk = st & 1;
            xoff[st] = patch_off<PW>(py0 + t / 3, px0 + t % 3, kk * 4 + fchunk);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int a = 0; a < 2; ++a) woff[kk][a] = lds_off(wn * 32 + a * 16 + frow, kk * 4 + fchunk);
    }

    // XCD-aware tile walk (workgroups b, b + 8, .. share an XCD and its L2): an XCD owns a contiguous range of tiles, so the
    // tiles in flight on it at any time are neighbours and their overlapping halo rows / columns are L2 hits
    const int xcd = blockIdx.x & 7;
    const int xq = ntiles >> 3, xr = ntiles & 7;
    const int xbase = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int tend = xbase + xq + (xcd < xr ? 1 : 0);                         // end of this XCD's range
    const int tstep = (G >> 3) + (xcd < (G & 7) ? 1 : 0);                     // workgroups on this XCD
    int T = xbase + (int)(blockIdx.x >> 3);
    if (T < tend) stage_patch(T, 0);
    int buf = 0;
    bool first = true;
    for (; T < tend; T += tstep, buf ^= 1) {
        int img, oy0, ox0;
        tile_origin(T, img, oy0, ox0);
        const bool has_next = T + tstep < tend;
        // this tile's patch (and, the first time, the weights) are the oldest entries of the queue; the previous tile's
        // stores may stay in flight
        if (first) wait_vmcnt<0>();
        else wait_vmcnt<NST>();
        first = false;
        wg_barrier();
        stage_shortcut(T);
        if (has_next) stage_patch(T + tstep, buf ^ 1);
        int fr = frow, fc = fchunk, td = tid;
        asm volatile("" : "+v"(fr), "+v"(fc), "+v"(td));  // per-tile address arithmetic, nothing hoisted into scratch

        // ---- GEMM 0: 3x3 conv from the patch. Wave (wm, wn): pixels of block rows 2 wm, 2 wm + 1; channels 32 wn + 16 a.
        {
            const unsigned char* sp = s_p + buf * PATCH_BYTES;
            const int q = frag_px(fr);
            f32x4_t acc0[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
            // fragments of step st + 2 (st = 2 tap + k-step) are requested before the MFMAs of step st: the sweep is a chain of
            // LDS round trips otherwise (hipcc issues one read, waits, issues one MFMA)
            uint4 xq[3], wq[3][2];
            auto ld0 = [&](int st, int slot) {
                const int t = st >> 1, kk = st & 1;
                xq[slot] = *reinterpret_cast<const uint4*>(sp + xoff[st]);
#pragma unroll
                for (int a = 0; a < 2; ++a) wq[slot][a] = *reinterpret_cast<const uint4*>(s_w2 + t * W_TAP + woff[kk][a]);
            };
            ld0(0, 0);
            ld0(1, 1);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + 2 < 18) ld0(st + 2, (st + 2) % 3);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc0[a] = Frag<lp16_t>::mma(wq[st % 3][a], xq[st % 3], acc0[a]);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int prow = wm * 16 + q;  // tile pixel (row-major) of this lane's MFMA column
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int c = wn * 32 + a * 16 + fc * 4;
                float v[4] = {relu_nan(acc0[a][0] + b2v[a].x), relu_nan(acc0[a][1] + b2v[a].y),
                              relu_nan(acc0[a][2] + b2v[a].z), relu_nan(acc0[a][3] + b2v[a].w)};
                store4<lp16_t>(reinterpret_cast<lp16_t*>(s_y + prow * 128 + (((c >> 3) ^ ((prow >> 1) & 7)) << 4) + ((c & 4) << 1)), v);
            }
        }
        if constexpr (CAT) {  // the x tile is an operand of GEMM 1: it must have landed (it is older than the patch prefetch)
            if (has_next) wait_vmcnt<PJ>();
            else wait_vmcnt<0>();
        }
        wg_barrier();  // y2 tile complete

        // ---- GEMM 1: 64 px x 256 ch, K = 64 (+ 64 of the shortcut conv). Wave tile 32 px x 64 ch, weights from registers.
        f32x4_t acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 xf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(s_y + lds_off(wm2 * 32 + b * 16 + fr, kk * 4 + fc));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = Frag<lp16_t>::mma(w3f[a][kk], xf[b], acc[a][b]);
        }
        if constexpr (CAT) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 xf[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(s_x + lds_off(wm2 * 32 + b * 16 + fr, kk * 4 + fc));
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = Frag<lp16_t>::mma(wsf[a][kk], xf[b], acc[a][b]);
            }
        } else {
            if (has_next) wait_vmcnt<PJ>();  // the residual pieces are older than the patch prefetch
            else wait_vmcnt<0>();
            wg_barrier();                    // everybody's residual pieces are in the out tile
        }
        // bias + residual (in place) + ReLU -> bf16 out tile; all eight residual cells are read before the first is
        // written back (a read behind a write to the same array is not hoisted by the compiler: eight LDS round trips)
        uint2 rcell[2][4];
        if constexpr (!CAT) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int px = wm2 * 32 + b * 16 + fr;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int c = wn4 * 64 + a * 16 + fc * 4;
                    rcell[b][a] = *reinterpret_cast<const uint2*>(s_r + px * 512 + (((c >> 3) ^ (px & 31)) << 4) + ((c & 4) << 1));
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = wm2 * 32 + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = wn4 * 64 + a * 16 + fc * 4;
                unsigned char* cell = s_r + px * 512 + (((c >> 3) ^ (px & 31)) << 4) + ((c & 4) << 1);
                float rr[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (!CAT) {
                    unpack_lp16x2(rcell[b][a].x, rr[0], rr[1]);
                    unpack_lp16x2(rcell[b][a].y, rr[2], rr[3]);
                }
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
        {
            unsigned char* obase = reinterpret_cast<unsigned char*>(p.out) + gpix(img, oy0, ox0, 0) * 512;  // uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (td >> 5) + 16 * i;
                const int pch = td & 31;
                const uint4 v = *reinterpret_cast<const uint4*>(s_r + row * 512 + (pch << 4));
                if constexpr (CN == 64) *reinterpret_cast<uint4*>(obase + out_lane[i]) = v;  // (CN 128: no registers to spare)
                else *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + (gpix(img, oy0, ox0, row) * 256 + (pch ^ (row & 31)) * 8) * 2) = v;
            }
        }
        // ---- GEMM 2: 64 px x CN ch, K = 256, B operand straight from the out tile. Wave tile 32 px x CN/4 ch.
        f32x4_t acc2[NA2][2];
#pragma unroll
        for (int a = 0; a < NA2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            uint4 x2[3][2];  // fragments two k-steps ahead of the MFMAs
            auto ld2 = [&](int ks, int slot) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int px = wm2 * 32 + b * 16 + fr;
                    x2[slot][b] = *reinterpret_cast<const uint4*>(s_r + px * 512 + (((ks * 4 + fc) ^ (px & 31)) << 4));
                }
            };
            ld2(0, 0);
            ld2(1, 1);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 2 < 8) ld2(ks + 2, (ks + 2) % 3);
#pragma unroll
                for (int a = 0; a < NA2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc2[a][b] = Frag<lp16_t>::mma(w1f[a][ks], x2[ks % 3][b], acc2[a][b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // z' tile: CN == 64 -> the y2 tile (every read of it was before the last barrier), 128-byte rows;
        //          CN == 128 -> over the out tile once every wave has finished reading it, 256-byte rows
        unsigned char* sz = CN == 64 ? s_y : s_r;
        if constexpr (CN == 128) wg_barrier();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = wm2 * 32 + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < NA2; ++a) {
                const int c = wn4 * (CN / 4) + a * 16 + fc * 4;
                float v[4];
                v[0] = relu_nan(acc2[a][b][0] + b1v[a].x);
                v[1] = relu_nan(acc2[a][b][1] + b1v[a].y);
                v[2] = relu_nan(acc2[a][b][2] + b1v[a].z);
                v[3] = relu_nan(acc2[a][b][3] + b1v[a].w);
                store4<lp16_t>(reinterpret_cast<lp16_t*>(sz + px * (CN * 2) + (((c >> 3) ^ (px & (CN / 8 - 1))) << 4) + ((c & 4) << 1)), v);
            }
        }
        wg_barrier();  // z' tile complete (CN == 64: and every read of the out tile done)
#pragma unroll
        for (int i = 0; i < CN / 64; ++i) {
            const int row = td >> 3, pch = (td & 7) * (CN / 64) + i;
            const uint4 v = *reinterpret_cast<const uint4*>(sz + row * (CN * 2) + (pch << 4));
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.zn) + (gpix(img, oy0, ox0, row) * CN + (pch ^ (row & (CN / 8 - 1))) * 8) * 2) = v;
        }
        // the top of the next iteration waits for the next patch (older than these NST stores) and its barrier keeps the
        // next residual from landing on a tile somebody still reads
    }
}

}  // namespace

extern "C" int agrl_bottleneck_block(const void* z, const void* w2, const float* b2, const void* w3, const float* b3,
                                     const void* residual, const void* x_short, const void* w_short, const float* b_short,
                                     void* out, const void* w1_next, const float* b1_next, void* z_next, int F, int H, int W,
                                     int Cmid, int Cout, int Cnext, agrl_stream_t stream) {
    AGRL_CHECK_ARG(z && w2 && b2 && w3 && b3 && out && w1_next && b1_next && z_next, "agrl_bottleneck_block: null pointer");
    AGRL_CHECK_ARG((residual != nullptr) != (x_short != nullptr),
                   "agrl_bottleneck_block: pass either the residual map or the downsample conv's input, not both");
    AGRL_CHECK_ARG(!x_short || (w_short && b_short), "agrl_bottleneck_block: the downsample form needs its weights and bias");
    AGRL_CHECK_ARG(F > 0 && H > 0 && W > 0 && (H % 8) == 0 && (W % 8) == 0, "agrl_bottleneck_block: maps must be multiples of 8 x 8 (got %d x %d)", H, W);
    AGRL_CHECK_ARG(Cmid == 64 && Cout == 256 && (Cnext == 64 || (Cnext == 128 && !x_short)),
                   "agrl_bottleneck_block: built for Cmid=64, Cout=256, Cnext=64 (128 without downsample), got %d/%d/%d", Cmid, Cout, Cnext);
    AGRL_CHECK_ARG((size_t)F * H * W * 256 * 2 < (1ull << 40), "agrl_bottleneck_block: problem too large");
    const uintptr_t al = (uintptr_t)z | (uintptr_t)w2 | (uintptr_t)b2 | (uintptr_t)w3 | (uintptr_t)b3 | (uintptr_t)residual |
                         (uintptr_t)x_short | (uintptr_t)w_short | (uintptr_t)b_short | (uintptr_t)out | (uintptr_t)w1_next |
                         (uintptr_t)b1_next | (uintptr_t)z_next;
    AGRL_CHECK_ARG((al & 15) == 0, "agrl_bottleneck_block: pointers must be 16-byte aligned");
    BlockParams p;
    p.z = z; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.res = residual; p.xs = x_short; p.ws = w_short; p.bs = b_short;
    p.out = out; p.w1n = w1_next; p.b1n = b1_next; p.zn = z_next; p.F = F; p.H = H; p.W = W;
    const int ntiles = F * (H / 8) * (W / 8);
    const int grid = ntiles < 256 ? ntiles : 256;
    if (x_short) hipLaunchKernelGGL((bottleneck_block_kernel<true, 64>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    else if (Cnext == 128) hipLaunchKernelGGL((bottleneck_block_kernel<false, 128>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    else hipLaunchKernelGGL((bottleneck_block_kernel<false, 64>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_block");
    return 0;
}
