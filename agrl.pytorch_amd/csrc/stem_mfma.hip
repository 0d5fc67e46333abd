// bf16-MFMA stem: conv 7x7/2 (3->64, BN folded) + ReLU + maxpool 3x3/2, fused. vmgn.py:281-284.
//
// The 7x7x3 filter is turned into a K = 7 x 32 contraction that needs NO im2col: the input patch sits in LDS as
// bf16 [y][x][4] (3 channels + a zero), so for a fixed filter row r the 7 taps x 4 channels an output pixel
// needs are 28 CONTIGUOUS bf16; padding that to 32 (the extra pixel meets zero weights) makes one k-step of
// v_mfma_f32_16x16x32_bf16 per filter row, and every lane's 8-element operand slice is one aligned ds_read_b128
// straight out of the patch. 1.5x redundant MFMA work buys zero gather instructions.
//
// One 512-thread workgroup -> an 8x8 tile of POOLED pixels x 64 channels of one frame:
//   patch 39x40x4 bf16 (12.5 KB) + packed weights 64 x 240 bf16 (30 KB, LDS-DMA)      -> LDS
//   conv tile 17x17 = 289 positions x 64 ch: 19 position fragments over 8 waves, 7 k-steps, +bias, ReLU
//   -> bf16 conv tile in LDS (overlaying patch+weights) -> 3x3/2 max -> NHWC store (128 B per pooled pixel)
#include <stdlib.h>

#include "agrl_common.h"

namespace {
constexpr int PT = 8;                 // pooled tile edge
constexpr int CT = 2 * PT + 1;        // conv tile edge 17
constexpr int NPOS = CT * CT;         // 289
constexpr int NFRAG = (NPOS + 15) / 16;  // 19
constexpr int NWV = 8;                // waves per workgroup
constexpr int NTH = 64 * NWV;
constexpr int FPW = (NFRAG + NWV - 1) / NWV;  // position fragments per wave: 3
constexpr int IT = 2 * (CT - 1) + 7;  // input patch edge 39
constexpr int PWP = 40;               // padded patch width (pixels)
constexpr int PATCH_BYTES = IT * PWP * 8;      // 12480
// 7*32 bf16 = 448 + 32 pad = 30 sixteen-byte slots per row. A ds_read_b128 is served in groups of 16 lanes: rows
// (lane & 15) 0-3, 12-15 at k-chunk g with rows 4-11 at k-chunk g + 1; 30 r mod 16 sends the first set to the even slots and
// the second (+1) to the odd ones: conflict-free (29 slots, the "odd stride" choice, collides on five of sixteen)
constexpr int WROW_BYTES = 480;
constexpr int W_BYTES = 64 * WROW_BYTES;       // 30720 = 30 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// Persistent form: a workgroup keeps the packed weights in LDS and walks tiles; the NEXT tile's input pixels are requested
// (12 floats per thread, in registers) right after the current patch has been written to LDS, so the HBM round trip runs
// under the MFMA sweep, the conv-tile epilogue and the pooling of the current tile. The one-tile-per-workgroup form spent
// most of a workgroup's life waiting: for its weights (30 KB per 8 x 8 pooled pixels), its pixels, its three barriers.
// LDS: weights 30 KB + one region that holds the patch, then the conv tile (37 KB) = 67 KB: two workgroups per CU.
constexpr int CT_BYTES = (NPOS + 3) * 128;
constexpr int REGION_BYTES = CT_BYTES > PATCH_BYTES ? CT_BYTES : PATCH_BYTES;

// SPLIT: the patch and the conv tile in SEPARATE LDS regions (12.2 + 36.5 + 30 KB + biases = 78.9 KB: still two workgroups per CU) -- the
// two barriers that guarded the overlay (sweep done -> conv tile may be written; pool done -> next patch may be written) disappear: a wave
// that has finished pooling writes its pixels of the next patch while the others still pool, two barriers per tile instead of four.
template <bool SPLIT>
__global__ __launch_bounds__(NTH, 4) void stem_mfma_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wpk,
                                                        const float* __restrict__ bias, lp16_t* __restrict__ out, int H,
                                                        int W, int CH, int CW, int PH, int PW, int tiles_w, int tiles_hw,
                                                        int ntiles, int xcd_map) {
    constexpr int TILES_BYTES = SPLIT ? PATCH_BYTES + CT_BYTES : REGION_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILES_BYTES + W_BYTES + 256];
    unsigned char* s_patch = smem;
    unsigned char* s_ct = SPLIT ? smem + PATCH_BYTES : smem;
    unsigned char* s_w = smem + TILES_BYTES;
    float* s_bias = reinterpret_cast<float*>(smem + TILES_BYTES + W_BYTES);  // 64 biases: LDS reads in the epilogue instead
                                                                             // of global loads whose waits also cover the prefetch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;

    // weights: 30 one-KiB DMA pieces, contiguous, once per workgroup
    for (int piece = wave; piece < W_BYTES / 1024; piece += NWV)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wpk + piece * 1024 + lane * 16), (lds_void_t*)(s_w + piece * 1024), 16, 0, 0);

    if (tid < 64) s_bias[tid] = bias[tid];
    // patch: one pixel (3 channels -> 4 bf16) per thread per pass; all loads of all passes are issued together
    constexpr int NPASS = (IT * PWP + NTH - 1) / NTH;  // 4
    float pv[NPASS][3];
    auto load_patch = [&](int T) {
        const int n = T / tiles_hw;
        const int trem = T - n * tiles_hw;
        const int ph0 = (trem / tiles_w) * PT, pw0 = (trem % tiles_w) * PT;
        const int iy0 = 2 * (2 * ph0 - 1) - 3, ix0 = 2 * (2 * pw0 - 1) - 3;
        const float* xn = x + (size_t)n * 3 * H * W;
        int td = tid;
        asm volatile("" : "+v"(td));  // per-tile address arithmetic (hoisted out of the tile loop it costs 100 registers)
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
    int a_off[FPW];  // byte offset of this lane's patch slice at filter row 0
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        int pos = (wave + NWV * i) * 16 + frow;
        pos = pos < NPOS ? pos : NPOS - 1;
        const int cy = pos / CT, cx = pos - cy * CT;
        a_off[i] = ((2 * cy) * PWP + 2 * cx + 2 * g) * 8;
    }
    auto ct_row = [](int pos) { return (pos & ~3) | ((pos & 1) << 1) | ((pos >> 1) & 1); };

    // Tile order. Neighbouring tiles share 7 of their 39 input columns / rows, and a 39-pixel row segment of a tile straddles 2-3 of the
    // 4 128-byte lines of its image row: with tile T on workgroup T mod G (XCD T mod 8) the four tiles across an image row sat on four
    // XCDs, each pulling its own copy of the shared lines into its own L2 -- counter fetch 300 MB for 100.7 MB of frames (2.5 x across, 1.22 x
    // down: round-5 PMC pass). Here every FRAME belongs to one XCD (frame n -> XCD n mod 8), whose G / 8 workgroups walk its frames' tiles
    // together (four frames in flight per XCD): the overlaps are L2 hits.
    const int nframes = ntiles / tiles_hw;
    const bool xmap = xcd_map && (G & 7) == 0 && nframes >= 8;
    const int xcd = blockIdx.x & 7;
    const int qstep = xmap ? (G >> 3) : G;
    const int qlimit = xmap ? ((nframes - xcd + 7) >> 3) * tiles_hw : ntiles;
    auto tile_of = [&](int q) {
        if (!xmap) return q;
        const int fl = q / tiles_hw;
        return (xcd + 8 * fl) * tiles_hw + (q - fl * tiles_hw);
    };
    int q = xmap ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (q < qlimit) load_patch(tile_of(q));
    for (; q < qlimit; q += qstep) {
        const int T = tile_of(q);
        const int n = T / tiles_hw;
        const int trem = T - n * tiles_hw;
        const int ph0 = (trem / tiles_w) * PT, pw0 = (trem % tiles_w) * PT;
        const int cr0 = 2 * ph0 - 1, cc0 = 2 * pw0 - 1;
        // ---- this tile's pixels (requested one tile ago) -> bf16 patch in LDS
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int e = tid + NTH * i;
            if (e < IT * PWP) {
                uint2 u;
                u.x = pack_lp16x2(pv[i][0], pv[i][1]);
                u.y = (uint32_t)f32_to_lp16(pv[i][2]);
                *reinterpret_cast<uint2*>(s_patch + e * 8) = u;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: the weight DMA (invisible to the compiler) has landed
        __syncthreads();
        if (q + qstep < qlimit) load_patch(tile_of(q + qstep));  // in flight until the top of the next iteration

        f32x4_t acc[FPW][4];
#pragma unroll
        for (int i = 0; i < FPW; ++i)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[i][a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            uint4 wf[4], xf[FPW];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                wf[a] = *reinterpret_cast<const uint4*>(s_w + (a * 16 + frow) * WROW_BYTES + r * 64 + g * 16);
#pragma unroll
            for (int i = 0; i < FPW; ++i) xf[i] = *reinterpret_cast<const uint4*>(s_patch + a_off[i] + r * (PWP * 8));
#pragma unroll
            for (int i = 0; i < FPW; ++i)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[i][a] = mfma_lp16_16x16x32(wf[a], xf[i], acc[i][a]);
        }
        if constexpr (!SPLIT) __syncthreads();  // every wave is done with the patch: the conv tile may overlay it
        int fr = frow, gg = g, tq = tid;
        asm volatile("" : "+v"(fr), "+v"(gg), "+v"(tq));  // epilogue / pooling addresses are recomputed per tile

        // conv tile [pos][64 ch] bf16, 8-byte slot s of row p stored at slot s ^ (p & 15); row p lives at row index ct_row(p)
        // = p with bits 0 and 1 swapped: a 32-lane group of the pooling reads below covers two pooled pixels = rows p and
        // p + 2, which would share every bank (rows alternate between the halves of the 256-byte bank window by bit 0)
        // conv positions outside the conv map (only tiles on the top / left image border have any) count as 0 in the pool
        const bool interior = cr0 >= 0 && cc0 >= 0 && cr0 + CT <= CH && cc0 + CT <= CW;
        int rowo[FPW], swz[FPW];
        bool live[FPW], in[FPW];
#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            const int pos = (wave + NWV * i) * 16 + fr;
            const int cy = pos / CT, cx = pos - cy * CT;
            live[i] = pos < NPOS;
            in[i] = interior || ((unsigned)(cr0 + cy) < (unsigned)CH && (unsigned)(cc0 + cx) < (unsigned)CW);
            rowo[i] = ct_row(pos) * 128;
            swz[i] = pos & 7;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ch = a * 16 + gg * 4;
            const float4 bv = *reinterpret_cast<const float4*>(s_bias + ch);
#pragma unroll
            for (int i = 0; i < FPW; ++i) {
                if (live[i]) {
                    const float v0 = relu_nan(acc[i][a][0] + bv.x), v1 = relu_nan(acc[i][a][1] + bv.y);
                    const float v2 = relu_nan(acc[i][a][2] + bv.z), v3 = relu_nan(acc[i][a][3] + bv.w);
                    uint2 u;
                    u.x = pack_lp16x2(v0, v1);
                    u.y = pack_lp16x2(v2, v3);
                    if (!in[i]) u = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(s_ct + rowo[i] + (((ch >> 3) ^ swz[i]) << 4) + ((ch & 4) << 1)) = u;
                }
            }
        }
        __syncthreads();

        // 3x3/2 max pool: thread -> 8 channels (one 16-byte slot) of ONE pooled pixel. The activations are post-ReLU bf16,
        // i.e. non-negative: their bit patterns order like unsigned 16-bit integers, so the maximum is two v_pk_max_u16 per
        // dword pair instead of unpack + fmax per channel (the kernel is VALU-bound)
        {
            typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
            const int cq = tq & 7;
            const int pp = tq >> 3;
            const int py = pp / PT, px = pp - py * PT;
            const int ph = ph0 + py, pw = pw0 + px;
            if (ph < PH && pw < PW) {
                u16x2_t m[4] = {u16x2_t{0, 0}, u16x2_t{0, 0}, u16x2_t{0, 0}, u16x2_t{0, 0}};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int pos = (2 * py + dy) * CT + 2 * px + dx;
                        const uint4 u = *reinterpret_cast<const uint4*>(s_ct + ct_row(pos) * 128 + ((cq ^ (pos & 7)) << 4));
                        const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) m[e] = __builtin_elementwise_max(m[e], __builtin_bit_cast(u16x2_t, w4[e]));
                    }
                const uint4 o = make_uint4(__builtin_bit_cast(uint32_t, m[0]), __builtin_bit_cast(uint32_t, m[1]),
                                           __builtin_bit_cast(uint32_t, m[2]), __builtin_bit_cast(uint32_t, m[3]));
                *reinterpret_cast<uint4*>(out + (((size_t)n * PH + ph) * PW + pw) * 64 + cq * 8) = o;
            }
        }
        if constexpr (!SPLIT) __syncthreads();  // the conv tile is consumed: the next patch may overwrite it
    }
}
}  // namespace

extern "C" int agrl_stem_conv_bn_relu_maxpool_lp16(const float* x, const void* w_packed, const float* bias, void* out,
                                                   int N, int H, int W, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w_packed && bias && out, "agrl_stem_lp16: null pointer");
    AGRL_CHECK_ARG(N > 0 && H >= 7 && W >= 7, "agrl_stem_lp16: bad shape N=%d H=%d W=%d", N, H, W);
    AGRL_CHECK_ARG((((uintptr_t)w_packed) & 15) == 0 && (((uintptr_t)bias) & 15) == 0 && (((uintptr_t)out) & 15) == 0,
                   "agrl_stem_lp16: misaligned pointer");
    const int CH = (H + 6 - 7) / 2 + 1, CW = (W + 6 - 7) / 2 + 1;
    const int PH = (CH + 2 - 3) / 2 + 1, PW = (CW + 2 - 3) / 2 + 1;
    const int tiles_h = cdiv(PH, PT), tiles_w = cdiv(PW, PT);
    const long long grid = (long long)N * tiles_h * tiles_w;
    AGRL_CHECK_ARG(grid < (1ll << 31), "agrl_stem_lp16: grid too large");
    const int wgs = 512;  // two persistent workgroups per CU (67 KB of LDS each)
    const unsigned launch = (unsigned)(grid < wgs ? grid : wgs);
    if (agrl_opts().stem_split_lds != 0)
        hipLaunchKernelGGL(stem_mfma_kernel<true>, dim3(launch), dim3(NTH), 0, (hipStream_t)stream, x,
                       (const unsigned char*)w_packed, bias, (lp16_t*)out, H, W, CH, CW, PH, PW, tiles_w, tiles_h * tiles_w,
                       (int)grid, agrl_opts().stem_xcd_map != 0);
    else
        hipLaunchKernelGGL(stem_mfma_kernel<false>, dim3(launch), dim3(NTH), 0, (hipStream_t)stream, x,
                       (const unsigned char*)w_packed, bias, (lp16_t*)out, H, W, CH, CW, PH, PW, tiles_w, tiles_h * tiles_w,
                       (int)grid, agrl_opts().stem_xcd_map != 0);
    AGRL_CHECK_LAUNCH("agrl_stem_lp16");
    return 0;
}
