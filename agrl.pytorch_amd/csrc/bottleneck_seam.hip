// The seam between two Bottlenecks of the MFMA-bound stages (layers 3 / 4), back to back in ONE kernel (16-bit build type):
//   out (M, N1) = relu(y2 (M, K1) @ w3 (N1, K1)^T + b3 + residual)        conv3 / bn3 / += / relu of block i,   vmgn.py:56-64
//   z   (M, N2) = relu(out @ w1n (N2, N1)^T + b1n)                        conv1 / bn1 / relu    of block i + 1, vmgn.py:48-50
// `out` (the 1024- / 2048-channel map) is written once -- it is the next block's residual -- and never read back: as two
// launches it crosses HBM three times (written, read by conv1, read as residual). The two GEMMs run on the same 128-pixel
// tile: conv3's output is produced in 256-channel chunks, each chunk is finished (bias + residual + ReLU, rounded once, stored)
// and at once contracted against the matching 256-deep k-slice of the next conv1 into an accumulator set that lives across
// the chunks (128 x N2 fp32: 64 / 128 registers per lane).
//
//   * 512 threads = 8 waves as 2 (pixels) x 4 (channels). GEMM 1 wave tile 64 px x 64 ch (16 accumulator quads), GEMM 2 wave
//     tile 64 px x N2 / 4 ch. Weights are the MFMA A operand (result rows = channels), pixels the B operand.
//   * The finished chunk changes hands through LDS (X, 64 KB): a lane's packed epilogue registers (8 consecutive channels of
//     one pixel) ARE the B fragment of one 32-deep k-step of GEMM 2, so X is an array of 1-KiB fragment blocks [k-step][pixel
//     half][fragment] written and read lane-linearly (conflict free), and the same registers go to HBM as 16-byte stores.
//   * The residual never touches LDS: a chunk's accumulators START as the residual (8 x 16-byte loads per lane, issued one
//     GEMM 2 earlier in inline asm with hand-counted waits, so that the weight ring's waits never drain behind them).
//   * LDS = 160 KB exactly: R (96 KB) + X (64 KB). GEMM 1 streams 48-KB k-tiles (128 pixel rows + 256 weight rows, 64 deep)
//     through THREE slots -- RA, RB and X itself, which is idle while a chunk is being accumulated -- two tiles ahead; GEMM 2
//     streams 32-KB weight units (N2 = 256: 64 deep, 128-byte rows; N2 = 512: 32 deep, 64-byte rows with their own
//     conflict-free swizzle) through the three 32-KB thirds of R, two units ahead. The slot orders are chosen at compile
//     time so that the last k-tile of a chunk sits in X (all of R is then free for the first three weight units) and the last
//     unit sits in the first third of R (RB and the consumed part of X are then free for the next chunk's first two k-tiles).
//     X is k-step major so that it frees up front to back while GEMM 2 walks it.
//   * All staging is LDS-DMA in inline asm with counted vmcnt waits; one workgroup barrier per k-tile / unit.
#include <utility>

#include "igemm_dev.h"

namespace {

struct SeamParams {
    const unsigned char* y2;
    const unsigned char* w3;
    const float* b3;
    const unsigned char* res;
    unsigned char* out;
    const unsigned char* w1n;
    const float* b1n;
    unsigned char* z;
    int M;
};

// Ablation bits for profiling builds (-DSEAM_ABL=n through tools/seam_ablate.sh; wrong results by design, never in the shipped
// library): 1 no residual loads / out stores, 2 no GEMM 1 MFMAs, 4 no GEMM 2 MFMAs, 8 no DMA, 16 no fragment reads, 32 no barriers,
// 64 weight pieces from contiguous addresses, 128 pixel pieces from contiguous addresses (is the row stride what the DMA waits for?)
#ifndef SEAM_ABL
#define SEAM_ABL 0
#endif
constexpr int SBM = 128;          // pixels per tile
constexpr int SCH = 256;          // conv3 output channels per chunk
constexpr int RA = 0, RB = 48 * 1024, XO = 96 * 1024;
constexpr int KT_Y = 16 * 1024;   // pixel part of a GEMM 1 k-tile (128 rows x 128 B); the weight part (256 x 128 B) follows
constexpr int BIAS_OFF = XO + 48 * 1024;  // + wave * 1024: the chunk's bias, one DMA piece per wave

constexpr int seq1(int nk1, int t) {
    const int xpos = (nk1 - 1) % 3, i = t % 3;
    if (i == xpos) return XO;
    const int first_other = xpos == 0 ? 1 : 0;
    return i == first_other ? RB : RA;
}
constexpr int gseq(int nu, int u) { return ((u % 3) + 3 - (nu - 1) % 3) % 3; }

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// 16 bytes global -> VGPRs behind hipcc's back (its waits would drain the DMA ring): the destination is valid only after a
// counted wait and the "+v" fence below
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;      // every LDS access through 32-bit address-space-3 pointers:
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;         // generic pointers cost a 64-bit add + null check per address
typedef __attribute__((address_space(3))) f32x4_t lds_f32x4_t;
__device__ __forceinline__ u32x4_t lds_ld16(const lds_u8_t* p) {
#if defined(SEAM_ABL) && (SEAM_ABL & 16)
    return u32x4_t{(unsigned)(size_t)p, 1u, 2u, 3u};
#else
    return *reinterpret_cast<const lds_u32x4_t*>(p);
#endif
}
template <int IMM>
__device__ __forceinline__ void gload16_asm(u32x4_t& dst, unsigned off, const unsigned char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(off), "s"(base), "n"(IMM) : "memory");
}
__device__ __forceinline__ void fence16(u32x4_t& v) { asm volatile("" : "+v"(v)); }

// LDS-DMA with the source as SGPR base + 32-bit lane offset and the destination as SGPR base + immediate: no per-piece
// VECTOR address arithmetic that hipcc could hoist out of the unrolled loops (there it costs two VGPRs per source and one SGPR
// per destination, per piece and k-tile: hundreds of spills). The instruction's own offset field is NOT used: it is added to
// the global AND the LDS address.
template <int LDS_IMM>
__device__ __forceinline__ void dma16s(const unsigned char* sbase, unsigned voff, unsigned lds_wave_base) {
    if constexpr (SEAM_ABL & 8) return;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_wave_base), "n"(LDS_IMM)
        : "memory", "scc");
}

template <int K1, int N1, int N2>
__global__ __launch_bounds__(512) void bottleneck_seam_kernel(const SeamParams p) {
    static_assert(K1 % 64 == 0 && N1 % SCH == 0 && (N2 == 256 || N2 == 512), "shapes");
    constexpr bool WIDE2 = N2 == 512;            // GEMM 2 units: 32 deep x 512 rows of 64 B instead of 64 deep x 256 rows of 128 B
    constexpr int NC = N1 / SCH;                 // chunks
    constexpr int NK1 = K1 / 64;                 // GEMM 1 k-tiles per chunk
    constexpr int NU = WIDE2 ? 8 : 4;            // GEMM 2 units per chunk
    constexpr int FN2 = N2 / 64;                 // GEMM 2 channel fragments per wave (4 / 8)
    constexpr int P1 = 6, P2 = 4;                // DMA pieces per wave: k-tile (2 pixel + 4 weight), unit
    static_assert(NK1 >= 4, "the slot schedule needs four k-tiles per chunk");
    // slot of k-tile t = seq1(NK1, t): the last k-tile of a chunk in X, the first non-X position RB, the other RA
    constexpr bool EARLY1 = seq1(NK1, 1) != RA;  // k-tile 1 can be requested under the previous chunk's last unit (RA is busy then)
    // third of R of unit u = gseq(NU, u) (x 32 KB): the last unit in third 0
    static_assert(seq1(NK1, NK1 - 1) == XO && gseq(NU, NU - 1) == 0, "slot schedule");

    __shared__ __attribute__((aligned(16))) unsigned char smem_[160 * 1024];
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int m0 = blockIdx.x * SBM;

    // ---- per-lane DMA source offsets (bytes; the host checks every operand is < 4 GB)
    const int lrow = lane >> 3, lchk = lane & 7;
    // Few lane offsets + per-piece constants on the scalar side of the address. LDS row a*16 + i of a wave's channel slab holds
    // channel sigma(a, i) = 32 (a>>1) + 8 (i>>2) + 4 (a&1) + (i&3): the MFMA result rows 4 f + r of a fragment PAIR of a lane
    // are then 8 consecutive channels (one 16-byte piece: igemm_wide.hip). For the row r = wave*32 + 8j + lrow of piece j
    // that is channel(j) = channel(0) + 16 (j&1) + 4 (j>>1), and the swizzle term of odd pieces differs by ^ 4: two lane
    // offsets (even / odd piece) + the constant 4 (j>>1) rows; the 64-byte-row units: one offset + 32 (j>>1) + 4 (j&1) rows.
    auto sigma = [](int a, int i) { return 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3); };
    unsigned a_off[2], b_off[2], d_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // pixel rows wave*16 + 8j + lrow of the tile
        const int row = wave * 16 + j * 8 + lrow;
        a_off[j] = (unsigned)(m0 + row) * (K1 * 2) + (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // conv3 weight rows wave*32 + 8j + lrow of the chunk (64-row slabs)
        const int r = wave * 32 + j * 8 + lrow;
        const int ch = (r & ~63) + sigma((r & 63) >> 4, r & 15);
        b_off[j] = (unsigned)ch * (K1 * 2) + (unsigned)((lchk ^ ((r >> 1) & 7)) << 4);
    }
    if constexpr (WIDE2) {  // 512 rows x 64 B: a piece = 16 rows, lane -> row + (lane >> 2), 16-byte chunk lane & 3; 128-row slabs
        const int r = wave * 64 + (lane >> 2);
        const int och = (r & ~127) + sigma((r & 127) >> 4, r & 15);
        d_off[0] = d_off[1] = (unsigned)och * (N1 * 2) + (unsigned)((((lane & 3) ^ (-(r >> 2))) & 3) << 4);
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // 256 rows x 128 B, 64-row slabs
            const int r = wave * 32 + j * 8 + lrow;
            const int och = (r & ~63) + sigma((r & 63) >> 4, r & 15);
            d_off[j] = (unsigned)och * (N1 * 2) + (unsigned)((lchk ^ ((r >> 1) & 7)) << 4);
        }
    }
    // residual / out: lane (f = fchunk, pixel frow of fragment b) owns channels 64 wn + 32 j + 8 f .. + 7 of a chunk
    const unsigned r_off = (unsigned)(m0 + wm * 64 + frow) * (N1 * 2) + (unsigned)(wn * 64 + 8 * fchunk) * 2;

    // ---- fragment read offsets inside a slot
    const int lo128 = frow * 128 + ((fchunk ^ ((frow >> 1) & 7)) << 4);         // 128-byte rows, k-step 0 (k-step 1: ^ 64)
    const int lo64 = frow * 64 + (((fchunk ^ (-(frow >> 2))) & 3) << 4);        // 64-byte rows: chunk ^ {0,3,2,1}[(row >> 2) & 3]
    const int xrd = lane * 16;
    (void)lo64;

    // ---- DMA issue helpers. Piece I of a GEMM 1 k-tile: 0-1 pixel rows, 2-5 weight rows. w3c / w1c = the chunk's weight slices.
    // Destinations = a per-wave SGPR base + compile-time offset.
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + wave * 2048);          // pixel rows wave*16 .. of a k-tile
    const unsigned ldsB = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);          // weight rows wave*32 .. of a k-tile / a unit's pieces
    const unsigned ldsC = __builtin_amdgcn_readfirstlane(lds0 + BIAS_OFF + wave * 1024);
    auto issue_k = [&](auto slot_c, const unsigned char* w3c, auto t_c, auto i_c) {
        constexpr int SLOT = decltype(slot_c)::value, T = decltype(t_c)::value, I = decltype(i_c)::value;
        if constexpr (I < 2) {
            if constexpr (SEAM_ABL & 128) dma16s<SLOT + I * 1024>(p.y2 + (size_t)m0 * (K1 * 2) + (T * 16 + I) * 1024 + wave * 2048, (unsigned)lane * 16u, ldsA);
            else dma16s<SLOT + I * 1024>(p.y2 + T * 128, a_off[I], ldsA);
        } else {
            if constexpr (SEAM_ABL & 64) dma16s<SLOT + KT_Y + (I - 2) * 1024>(w3c + (T * 32 + (I - 2)) * 1024 + wave * 4096, (unsigned)lane * 16u, ldsB);
            else dma16s<SLOT + KT_Y + (I - 2) * 1024>(w3c + (4 * ((I - 2) >> 1) * (K1 * 2) + T * 128), b_off[(I - 2) & 1], ldsB);
        }
    };
    auto issue_u = [&](auto third_c, const unsigned char* w1c, auto u_c, auto i_c) {
        constexpr int THIRD = decltype(third_c)::value, U = decltype(u_c)::value, I = decltype(i_c)::value;
        if constexpr (SEAM_ABL & 64) dma16s<THIRD * 32768 + I * 1024>(w1c + (U * 32 + I) * 1024 + wave * 4096, (unsigned)lane * 16u, ldsB);
        else if constexpr (WIDE2) dma16s<THIRD * 32768 + I * 1024>(w1c + ((32 * (I >> 1) + 4 * (I & 1)) * (N1 * 2) + U * 64), d_off[0], ldsB);
        else dma16s<THIRD * 32768 + I * 1024>(w1c + (4 * (I >> 1) * (N1 * 2) + U * 128), d_off[I & 1], ldsB);
    };
    auto issue_bias = [&](int c) {
        dma16s<0>(reinterpret_cast<const unsigned char*>(p.b3 + c * SCH + wn * 64), (unsigned)(lane & 15) * 16u, ldsC);
    };
    auto w3_chunk = [&](int c) { return p.w3 + (size_t)c * (SCH * K1 * 2); };
    auto w1_chunk = [&](int c) { return p.w1n + (size_t)c * (SCH * 2); };

    f32x4_t acc1[4][4];
    f32x4_t acc2[FN2][4];
#pragma unroll
    for (int a = 0; a < FN2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t rres[4][2];
    constexpr int S_OPS = (SEAM_ABL & 1) ? 0 : 16;  // the epilogue's 8 stores + 8 residual loads per lane
    auto load_residual = [&](int c) {  // one lane offset; fragment / chunk offsets on the scalar side, the channel half as immediate
        const unsigned char* rc = p.res + (size_t)c * (SCH * 2);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (SEAM_ABL & 1) { rres[b][0] = u32x4_t{(unsigned)c, r_off, 1u, 2u}; rres[b][1] = rres[b][0]; continue; }
            gload16_asm<0>(rres[b][0], r_off, rc + b * (16 * N1 * 2));
            gload16_asm<64>(rres[b][1], r_off, rc + b * (16 * N1 * 2));
        }
    };
    auto init_acc1 = [&]() {  // the chunk's accumulators start as the residual (the loads were waited for by the caller)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fence16(rres[b][j]);
                const uint32_t w4[4] = {rres[b][j][0], rres[b][j][1], rres[b][j][2], rres[b][j][3]};
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) unpack_lp16x2(w4[e], v[2 * e], v[2 * e + 1]);
                acc1[2 * j][b] = f32x4_t{v[0], v[1], v[2], v[3]};
                acc1[2 * j + 1][b] = f32x4_t{v[4], v[5], v[6], v[7]};
            }
    };

    // NP DMA pieces ride between the 8 MFMA groups of a k-tile / unit: pieces [G NP / 8, (G + 1) NP / 8) in front of group G
    auto issue_for_group = [&](auto g_c, auto np_c, auto&& issue) {
        constexpr int G = decltype(g_c)::value, NP = decltype(np_c)::value;
        constexpr int LO = G * NP / 8, HI = (G + 1) * NP / 8;
        static_for<HI - LO>([&](auto ic) { issue(std::integral_constant<int, LO + decltype(ic)::value>{}); });
    };

    // ---- one GEMM 1 k-tile: 2 k-steps x 4 channel fragments x 4 pixel fragments. With N2 = 512 the two accumulator sets
    // leave 64 registers for everything else: the second k-step's pixel fragments then replace the first's one by one, each
    // right behind the last MFMA that reads its register (operands are read at issue), instead of into a second set.
    auto gemm1_ktile = [&](auto slot_c, auto np_c, auto&& issue) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr bool TIGHT = WIDE2;
        const lds_u8_t* sa = smem + SLOT + (wm * 64) * 128;
        const lds_u8_t* sb = smem + SLOT + KT_Y + (wn * 64) * 128;
        constexpr int WR = TIGHT ? 2 : 3;  // weight-fragment ring: WR - 1 groups of read-ahead
        u32x4_t xfr[TIGHT ? 1 : 2][4], wfr[WR];
        auto ldx = [&](int kk, int b) { return lds_ld16(sa + b * 2048 + (lo128 ^ (kk * 64))); };
        auto ldw = [&](int g) { return lds_ld16(sb + (g & 3) * 2048 + (lo128 ^ ((g >> 2) * 64))); };
        wfr[0] = ldw(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) xfr[0][b] = ldx(0, b);
        if constexpr (WR == 3) wfr[1] = ldw(1);
        static_for<8>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + WR - 1 < 8) wfr[(g + WR - 1) % WR] = ldw(g + WR - 1);
            if constexpr (!TIGHT && g < 4) xfr[1][g] = ldx(1, g);
            issue_for_group(gc, np_c, issue);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (SEAM_ABL & 2) asm volatile("" ::"v"(wfr[g % WR]), "v"(xfr[TIGHT ? 0 : g >> 2][b]));
                else acc1[g & 3][b] = mfma_lp16_16x16x32(wfr[g % WR], xfr[TIGHT ? 0 : g >> 2][b], acc1[g & 3][b]);
                if constexpr (TIGHT && g == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    xfr[0][b] = ldx(1, b);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- one GEMM 2 unit
    auto gemm2_unit = [&](auto third_c, auto u_c, auto np_c, auto&& issue) {
        constexpr int THIRD = decltype(third_c)::value;
        constexpr int U = decltype(u_c)::value;
        if constexpr (WIDE2) {
            // 32 deep: k-step U; 8 channel fragments x 4 pixel fragments
            const lds_u8_t* sb = smem + THIRD * 32768 + (wn * 128) * 64 + lo64;
            const lds_u8_t* sx = smem + XO + ((U * 2 + wm) * 4) * 1024 + xrd;
            u32x4_t xfr[4], wfr[3];
            wfr[0] = lds_ld16(sb);
#pragma unroll
            for (int b = 0; b < 4; ++b) xfr[b] = lds_ld16(sx + b * 1024);
            wfr[1] = lds_ld16(sb + 1024);
            static_for<8>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g + 2 < 8) wfr[(g + 2) % 3] = lds_ld16(sb + (g + 2) * 1024);
                issue_for_group(gc, np_c, issue);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if constexpr (SEAM_ABL & 4) asm volatile("" ::"v"(wfr[g % 3]), "v"(xfr[b]));
                    else acc2[g][b] = mfma_lp16_16x16x32(wfr[g % 3], xfr[b], acc2[g][b]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            // 64 deep: k-steps 2U, 2U + 1; 4 channel fragments x 4 pixel fragments each
            const lds_u8_t* sb = smem + THIRD * 32768 + (wn * 64) * 128;
            const lds_u8_t* sx = smem + XO + ((2 * U * 2 + wm) * 4) * 1024 + xrd;
            u32x4_t xfr[2][4], wfr[3];
            auto ldx = [&](int kk, int b) { return lds_ld16(sx + kk * 8192 + b * 1024); };
            auto ldw = [&](int g) { return lds_ld16(sb + (g & 3) * 2048 + (lo128 ^ ((g >> 2) * 64))); };
            wfr[0] = ldw(0);
#pragma unroll
            for (int b = 0; b < 4; ++b) xfr[0][b] = ldx(0, b);
            wfr[1] = ldw(1);
            static_for<8>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g + 2 < 8) wfr[(g + 2) % 3] = ldw(g + 2);
                if constexpr (g < 4) xfr[1][g] = ldx(1, g);
                issue_for_group(gc, np_c, issue);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if constexpr (SEAM_ABL & 4) asm volatile("" ::"v"(wfr[g % 3]), "v"(xfr[g >> 2][b]));
                    else acc2[g & 3][b] = mfma_lp16_16x16x32(wfr[g % 3], xfr[g >> 2][b], acc2[g & 3][b]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    };

    auto step_sync = [&](auto allowed_c) {  // the awaited pieces of THIS wave have landed; the barrier covers everybody else's
        if constexpr (!(SEAM_ABL & 8)) wait_vmcnt<decltype(allowed_c)::value>();
        if constexpr (!(SEAM_ABL & 32)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using std::integral_constant;
    constexpr integral_constant<int, 0> I0{};
    constexpr integral_constant<int, 1> I1{};
    constexpr integral_constant<int, 2> I2{};

    // ---- GEMM 1 of chunk c (k-tile 0 [and 1] already requested, residual registers loading)
    auto run_gemm1 = [&](int c) {
        const unsigned char* w3c = w3_chunk(c);
        const unsigned char* w1c = w1_chunk(c);
        static_for<NK1>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr integral_constant<int, seq1(NK1, T)> slot{};
            // younger than k-tile T at this point: k-tile T+1 (if requested by now) and, at T == 2, the bias piece
            constexpr int ALLOWED = T == 0 ? (EARLY1 ? P1 : 0) : ((T + 1 < NK1 ? P1 : 0) + (T == 2 ? 1 : 0));
            step_sync(integral_constant<int, ALLOWED>{});
            if constexpr (T == 0) {
                init_acc1();
                __builtin_amdgcn_sched_barrier(0);  // the k-tile's fragment reads stay behind the unpacking (32 + 64 live registers there)
                constexpr int NP = (EARLY1 ? 0 : P1) + P1;  // [k-tile 1 ->] k-tile 2
                gemm1_ktile(slot, integral_constant<int, NP>{}, [&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    if constexpr (!EARLY1 && I < P1) issue_k(integral_constant<int, seq1(NK1, 1)>{}, w3c, I1, ic);
                    else issue_k(integral_constant<int, seq1(NK1, 2)>{}, w3c, I2, integral_constant<int, EARLY1 ? I : I - P1>{});
                });
            } else if constexpr (T + 1 == NK1) {
                // all of R is free: the first three weight units of GEMM 2
                gemm1_ktile(slot, integral_constant<int, 3 * P2>{}, [&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    issue_u(integral_constant<int, gseq(NU, I / P2)>{}, w1c, integral_constant<int, I / P2>{}, integral_constant<int, I % P2>{});
                });
            } else if constexpr (T + 2 < NK1) {
                constexpr int NP = P1 + (T == 1 ? 1 : 0);
                gemm1_ktile(slot, integral_constant<int, NP>{}, [&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    if constexpr (I < P1) issue_k(integral_constant<int, seq1(NK1, T + 2)>{}, w3c, integral_constant<int, T + 2>{}, ic);
                    else issue_bias(c);
                });
            } else {
                gemm1_ktile(slot, I0, [&](auto) {});
            }
        });
    };

    // ---- prologue: residual of chunk 0, then its first k-tile(s)
    load_residual(0);
    static_for<P1>([&](auto ic) { issue_k(integral_constant<int, seq1(NK1, 0)>{}, p.w3, I0, ic); });
    if constexpr (EARLY1) static_for<P1>([&](auto ic) { issue_k(integral_constant<int, seq1(NK1, 1)>{}, p.w3, I1, ic); });

    for (int c = 0;; ++c) {
        const bool last = c + 1 == NC;
        run_gemm1(c);

        // ---- chunk epilogue: + bias, ReLU, round once; the packed registers go to HBM (next block's residual) and to X
        float bias[2][8];
        {
            const lds_u8_t* bs = smem + BIAS_OFF + wave * 1024 + fchunk * 32;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4_t b0 = *reinterpret_cast<const lds_f32x4_t*>(bs + j * 128);
                const f32x4_t b1 = *reinterpret_cast<const lds_f32x4_t*>(bs + j * 128 + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bias[j][e] = b0[e]; bias[j][4 + e] = b1[e]; }
            }
        }
        wg_barrier();  // every wave is past its reads of the last k-tile (it sits in X) and holds its bias
        unsigned char* oc = p.out + (size_t)c * (SCH * 2);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = relu_nan(acc1[2 * j][b][e] + bias[j][e]);
                    v[4 + e] = relu_nan(acc1[2 * j + 1][b][e] + bias[j][4 + e]);
                }
                const u32x4_t pk = {pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7])};
                if (!(SEAM_ABL & 1) || pk[0] == 0x12345678u) *reinterpret_cast<u32x4_t*>(oc + b * (16 * N1 * 2) + j * 64 + r_off) = pk;
                *reinterpret_cast<lds_u32x4_t*>(smem + XO + ((((2 * wn + j) * 2 + wm) * 4 + b) * 1024) + xrd) = pk;
            }
        // the last chunk requests chunk 0's residual and first k-tiles again (never used, drained at the end): the code stays
        // branch-free -- two variants of a unit meet in accumulator copies, and the wait counts stay the same
        const int cn = last ? 0 : c + 1;
        load_residual(cn);
        wg_barrier();  // X is complete

        // ---- GEMM 2 over the chunk
        const unsigned char* w1c = w1_chunk(c);
        const unsigned char* w3n = w3_chunk(cn);
        static_for<NU>([&](auto uc) {
            constexpr int U = decltype(uc)::value;
            constexpr integral_constant<int, gseq(NU, U)> third{};
            // younger than unit U: the later units requested so far and the epilogue's 8 stores (+ 8 residual loads)
            constexpr int AFTER = U == 0 ? 2 * P2 : (U == 1 ? P2 : (U + 1 < NU ? P2 : 0));
            constexpr bool WITH_S = U <= 2;
            step_sync(integral_constant<int, AFTER + (WITH_S ? S_OPS : 0)>{});
            if constexpr (U + 1 == NU) {
                // RB and the consumed front of X are free: the next chunk's first k-tile(s)
                constexpr int NP = EARLY1 ? 2 * P1 : P1;
                gemm2_unit(third, uc, integral_constant<int, NP>{}, [&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    if constexpr (I < P1) issue_k(integral_constant<int, seq1(NK1, 0)>{}, w3n, I0, ic);
                    else issue_k(integral_constant<int, seq1(NK1, 1)>{}, w3n, I1, integral_constant<int, I - P1>{});
                });
            } else if constexpr (U >= 1 && U + 2 < NU) {
                gemm2_unit(third, uc, integral_constant<int, P2>{}, [&](auto ic) {
                    issue_u(integral_constant<int, gseq(NU, U + 2)>{}, w1c, integral_constant<int, U + 2>{}, ic);
                });
            } else {
                gemm2_unit(third, uc, I0, [&](auto) {});
            }
        });
        if (last) break;
    }

    // ---- z = relu(acc2 + b1n), rounded, 16-byte stores (64 contiguous bytes per pixel row and instruction)
    lp16_t* __restrict__ zp = reinterpret_cast<lp16_t*>(p.z);
    const int cb2 = wn * (N2 / 4) + 8 * fchunk;
#pragma unroll
    for (int j = 0; j < FN2 / 2; ++j) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.b1n + cb2 + 32 * j);
        const float4 b1 = *reinterpret_cast<const float4*>(p.b1n + cb2 + 32 * j + 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gm = m0 + wm * 64 + b * 16 + frow;
            float v[8];
            v[0] = acc2[2 * j][b][0] + b0.x; v[1] = acc2[2 * j][b][1] + b0.y; v[2] = acc2[2 * j][b][2] + b0.z; v[3] = acc2[2 * j][b][3] + b0.w;
            v[4] = acc2[2 * j + 1][b][0] + b1.x; v[5] = acc2[2 * j + 1][b][1] + b1.y; v[6] = acc2[2 * j + 1][b][2] + b1.z; v[7] = acc2[2 * j + 1][b][3] + b1.w;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
            *reinterpret_cast<uint4*>(zp + (size_t)gm * N2 + cb2 + 32 * j) =
                make_uint4(pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7]));
        }
    }
}

}  // namespace

bool bottleneck_seam_applicable(int M, int Cmid, int Cout, int Cnext) {
    if (M <= 0 || M % SBM) return false;
    if ((size_t)M * Cout * 2 >= (1ull << 32)) return false;
    return (Cmid == 256 && Cout == 1024 && Cnext == 256) || (Cmid == 512 && Cout == 2048 && Cnext == 512) ||
           (Cmid == 256 && Cout == 1024 && Cnext == 512);
}

int launch_bottleneck_seam(const void* y2, const void* w3, const float* b3, const void* residual, void* out, const void* w1_next,
                           const float* b1_next, void* z, int M, int Cmid, int Cout, int Cnext, hipStream_t stream) {
    AGRL_CHECK_ARG(bottleneck_seam_applicable(M, Cmid, Cout, Cnext), "agrl_bottleneck_tail: layer-3/4 form needs M %% 128 == 0 and 256/1024/256, 256/1024/512 or 512/2048/512 channels");
    SeamParams p;
    p.y2 = reinterpret_cast<const unsigned char*>(y2);
    p.w3 = reinterpret_cast<const unsigned char*>(w3);
    p.b3 = b3;
    p.res = reinterpret_cast<const unsigned char*>(residual);
    p.out = reinterpret_cast<unsigned char*>(out);
    p.w1n = reinterpret_cast<const unsigned char*>(w1_next);
    p.b1n = b1_next;
    p.z = reinterpret_cast<unsigned char*>(z);
    p.M = M;
    const dim3 grid(M / SBM), block(512);
    if (Cmid == 256 && Cnext == 256) hipLaunchKernelGGL((bottleneck_seam_kernel<256, 1024, 256>), grid, block, 0, stream, p);
    else if (Cmid == 256) hipLaunchKernelGGL((bottleneck_seam_kernel<256, 1024, 512>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((bottleneck_seam_kernel<512, 2048, 512>), grid, block, 0, stream, p);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_tail");
    return 0;
}
