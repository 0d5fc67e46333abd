// 256 x 256 tile form of the pointwise (1x1, stride 1) implicit GEMM for the MFMA-bound layers (layer 3/4 convs:
// M = 32768 pixels, N = 256..2048 channels, K = 256..2048), bf16 in / bf16 out.
//
// Why a second tile shape. Every byte of a k-tile goes global -> LDS through the CU's 64 B/clk vector-memory path
// and comes back out of the LDS through the 256 B/clk read port. For the 128 x 128 tile (igemm.hip) both cost as many
// cycles as the MFMAs of that k-tile (32 KB in, 96 KB of fragment reads at 8 waves, 515 MFMA cycles), so the matrix
// pipe can never be more than half busy. A 256 x 256 tile with 64 x 128 wave tiles needs half the bytes per flop on
// both paths: 64 KB in and 192 KB of fragment reads per 2060 MFMA cycles.
//
//   * 512 threads = 8 waves as 4 (pixels) x 2 (channels); wave tile 64 x 128 = 4 x 8 MFMA 16x16x32 fragments,
//     128 accumulator registers per lane; one workgroup per CU (128 KB of LDS: 2-slot ring of 64 KB k-tiles)
//   * staging: LDS-DMA, 8 one-KiB pieces per wave per k-tile, one piece per 4 MFMAs in the first half of the k-tile;
//     fragment reads software-pipelined two MFMA groups ahead; same swizzled
//     128-byte-row layout and the same weight-as-A-operand fragment assignment as igemm.hip
//   * epilogue in two column halves through the two ring slots (a 256 x 256 bf16 tile is the whole ring): half h =
//     the h-th 64 channels of BOTH wave columns, so all 8 waves work in both halves; the residual half-tiles are
//     DMA'd into the slots, combined in fp32 in place, rounded once and written out as whole 16-byte chunks
#include <stdlib.h>

#include "igemm_dev.h"

namespace {

constexpr int WBM = 256, WBN = 256;

constexpr bool REGEPI_OK(int dbg) { return (dbg & 8192) == 0; }
template <int DBG, int WBN_ = 256, bool RES = false>  // RES: + residual (its 64 staging registers leave no room for the persistent form's carried state); WBN_: channels per tile (256, or 128 for N = 256 layers: twice the tiles); DBG: ablation bits, compile time (a runtime test inside the k-loop wrecks the schedule): 2 no MFMA, 8 no steady-state DMA, 16 no fragment reads, 32 no DMA waits, 128 no barrier
__global__ __launch_bounds__(512) void igemm_wide_kernel(const IgemmParams p) {
    static_assert(WBN_ == 256 || ((WBN_ == 128 || WBN_ == 192) && (DBG & (4096 | 8192 | 16384)) == 0), "the 128- / 192-channel tiles have the plain schedule + register epilogue only");
    static_assert(WBN_ != 192 || (DBG & 65536) != 0, "the 192-column tile exists for the fp32-output distance form only");
    constexpr bool POOL = (DBG & 16384) != 0;   // fused frame pooling epilogue (a tile = two whole 16 x 8 frames)
    // DUAL (bit 32768): TWO pixel-row operands concatenated along K -- out = [x | x2] [W1 | W2]^T: a Bottleneck's last conv
    // and its 1x1 downsample conv as ONE GEMM (vmgn.py:56-64: bn3(conv3(y2)) + downsample(x), both BatchNorms folded), so the
    // shortcut map is neither written nor read back and the sum is formed in fp32. x rows are exactly twice as long as x2
    // rows (ResNet: inplanes = 2 planes in the first block of layers 2-4), which lets ONE stored offset per piece serve both.
    constexpr bool DUAL = (DBG & 32768) != 0;
    // F32OUT (bit 65536): fp32 output with the distance epilogue out = alpha * acc + rowv[m] (or rowc) + colv[n] -- the full
    // query x gallery distance matrix of BASELINE configs[4] (metrics/distance.py:59-89) through the 256 x 256 tile: the tiled
    // igemm_kernel it used before sits at 0.30 of the MFMA peak on that shape. M and N need not be tile multiples: rows of
    // either operand beyond the matrix are staged as zeros and their results are not stored (N % 4 == 0 for the 16-byte stores).
    constexpr bool F32OUT = (DBG & 65536) != 0;
    static_assert(!F32OUT || ((WBN_ == 256 || WBN_ == 192) && !RES && !POOL && !DUAL && REGEPI_OK(DBG)), "fp32 output: plain 256- / 192-column tiles, register epilogue");
    constexpr bool REGEPI = (DBG & 8192) == 0;  // bit 8192: the LDS-staged two-half epilogue (kept for A/B)
    constexpr int BM = WBM, BN = WBN_, NW = 8, WM = 4;
    constexpr int FM = BM / (16 * WM);  // 4 pixel fragments per wave
    constexpr int FN = BN / 32;         // 8 channel fragments per wave
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;  // 64 KB per ring slot
    constexpr int AJ = BM / (8 * NW), BJ = BN / (8 * NW);                                  // 4 + 4 DMA pieces per wave
    constexpr int DPT = AJ + BJ;
    constexpr int NS = BN == 128 ? 3 : 2;  // ring slots: the 48 KB k-tiles of the 128-channel tile fit three times
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM;
    const int wn = wave / WM;

    // XCD-aware block -> tile map (blocks b, b+8, .. share an XCD): every XCD owns a contiguous range of tiles, so the
    // N-tiles of one M-tile are neighbours in one L2. PERSISTENT when the grid is smaller than the tile count: a
    // workgroup walks its XCD's range with the stride of the workgroups on that XCD; the first k-tile of its next tile
    // is requested BEFORE the epilogue's stores are issued, so the stores drain under the next tile's matrix work
    // (a workgroup that ends instead holds its LDS until its last store is acknowledged, and its successor on the CU
    // starts with a cold pipeline).
    const int nNt = F32OUT ? (p.N + BN - 1) / BN : p.N / BN;
    const int ntiles = ((p.M + BM - 1) / BM) * nNt;
    const int xcd = blockIdx.x & 7;
    int tl = blockIdx.x >> 3;                                                  // tile index inside the XCD's range
    const int xq = ntiles >> 3, xr = ntiles & 7;
    const int xbase = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int xcnt = xq + (xcd < xr ? 1 : 0);
    const int wstep = (int)(gridDim.x >> 3) + (xcd < (int)(gridDim.x & 7) ? 1 : 0);  // workgroups on this XCD
    if (tl >= xcnt) return;

    const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* __restrict__ xg2 = reinterpret_cast<const unsigned char*>(p.x2);
    const unsigned char* __restrict__ wg = reinterpret_cast<const unsigned char*>(p.w);
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);

    // per-lane DMA sources (32-bit byte offsets; the host checks the operands are < 4 GB): piece j of this wave
    // covers tile rows wave*32 + 8j .. +7, lane L -> row + (L>>3), physical chunk L&7 = global chunk (L&7)^((row>>1)&7)
    const int lrow = lane >> 3, lchk = lane & 7;
    const unsigned row_bytes = (unsigned)p.K * 2u;
    // DUAL: a_off is kept for the SHORT rows (x2, K - K1 = K1 / 2 elements); the long-row offset of the same pixel and chunk
    // is 2 a_off - (a_off & 127) (row starts are multiples of 128 bytes, the swizzled chunk sits in the low 7 bits)
    const unsigned row_bytes_a = DUAL ? (unsigned)(p.K - p.K1) * 2u : row_bytes;
    const unsigned split_bytes = DUAL ? (unsigned)p.K1 * 2u : 0u;
    unsigned a_off[AJ], b_off[BJ];
    unsigned a_okmask = 0, b_okmask = 0;
    int m0 = 0, n0 = 0;
    auto setup_tile = [&](int tile) {
        const int mt = tile / nNt;
        const int nt = tile - mt * nNt;
        m0 = mt * BM;
        n0 = nt * BN;
        a_okmask = 0;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int row = wave * (BM / NW) + j * 8 + lrow;
            const int gm = m0 + row;
            if (gm < p.M) a_okmask |= 1u << j;
            int pix = gm < p.M ? gm : 0;
            if (p.stride > 1) {  // strided 1x1 (the downsample convs): output pixel -> the input pixel it reads
                const int ohw = p.OH * p.OW;
                const int n = pix / ohw, rem = pix - n * ohw;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                pix = (n * p.H + oy * p.stride) * p.W + ox * p.stride;
            }
            a_off[j] = (unsigned)pix * row_bytes_a + (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = wave * (BN / NW) + j * 8 + lrow;
            int ch = row;
            if constexpr (REGEPI) {
                // LDS row a*16 + i of a wave column's 128-channel slab holds channel sigma(a, i) = 32 (a>>1) + 8 (i>>2) +
                // 4 (a&1) + (i&3): the MFMA result rows 4 f + r of the 8 fragments of a lane are then 32 channels that the
                // register epilogue reads / writes as 16-byte pieces, 64 contiguous bytes per pixel row and instruction
                constexpr int SLAB = BN / 2;  // channels per wave column
                const int rp = row % SLAB, a = rp >> 4, i = rp & 15;   // (SLAB = 128 / 64, or 96 for the 192-column distance tile)
                ch = (row - rp) + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
            }
            if constexpr (F32OUT) {
                if (j == 0) b_okmask = 0;
                if (n0 + ch < p.N) b_okmask |= 1u << j;
                else ch = 0;
            }
            b_off[j] = (unsigned)(n0 + ch) * row_bytes + (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
        }
    };
    setup_tile(xbase + tl);
    unsigned kbyte = 0;  // byte offset of the k-tile being STAGED inside a row
    auto stage_piece = [&](int buf, int idx) {
        if (idx < AJ) {
            unsigned char* sa = smem + buf * BUF_BYTES + wave * (BM / NW) * 128;
            if constexpr (DUAL) {
                const bool second = kbyte >= split_bytes;  // wave-uniform
                const unsigned char* src = second ? xg2 + a_off[idx] + (kbyte - split_bytes)
                                                  : xg + (2u * a_off[idx] - (a_off[idx] & 127u)) + kbyte;
                dma16((a_okmask >> idx) & 1u ? src : zsrc, sa + idx * 1024);
            } else
            dma16((a_okmask >> idx) & 1u ? xg + a_off[idx] + kbyte : zsrc, sa + idx * 1024);
        } else {
            const int j = idx - AJ;
            unsigned char* sb = smem + buf * BUF_BYTES + A_BYTES + wave * (BN / NW) * 128;
            if constexpr (F32OUT) dma16((b_okmask >> j) & 1u ? wg + b_off[j] + kbyte : zsrc, sb + j * 1024);
            else
            dma16(wg + b_off[j] + kbyte, sb + j * 1024);
        }
    };

    // epilogue half-tile image: 256 rows x 256 B; 16-byte chunk c (8 channels) of row r sits at chunk c ^ (r & 15).
    // Half h holds channels [64h, 64h+64) of wave column 0 as chunks 0-7 and of wave column 1 as chunks 8-15.
    auto half_col = [&](int h, int vchunk) { return n0 + (vchunk >> 3) * (BN / 2) + h * 64 + (vchunk & 7) * 8; };
    auto stage_residual_half = [&](int h, unsigned char* so) {
        const unsigned char* rg = reinterpret_cast<const unsigned char*>(p.res);
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // 64 four-row pieces per half, 8 per wave
            const int row0 = (wave * 8 + j) * 4;
            const int row = row0 + (lane >> 4);
            const int vch = (lane & 15) ^ (row & 15);
            const int gm = m0 + row;
            dma16(gm < p.M ? rg + ((size_t)gm * p.ldo + half_col(h, vch)) * 2 : zsrc, so + row0 * 256);
        }
    };

    const int nk = p.K >> 6;
    constexpr bool has_res = RES;
    int frow = lane & 15;
    int fchunk = lane >> 4;
    int cur = 0;
    // the first NS-1 k-tiles of the CURRENT tile (a_off / b_off) into the ring slots cur, cur+1, ..
    auto stage_head = [&]() {
        kbyte = 0;
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) {
            if (s < nk) {
                int slot = cur + s;
                slot = slot >= NS ? slot - NS : slot;
#pragma unroll
                for (int i = 0; i < DPT; ++i) stage_piece(slot, i);
                kbyte += 128;
            }
        }
    };
    stage_head();
    constexpr int EPI_STORES = FM * (FN / 2);  // 16-byte stores per lane in the register epilogue
    bool carried_any = false;  // not the workgroup's first tile
    bool carried = false;  // this tile's head was requested in front of the previous tile's stores (still in flight)
    for (;;) {
    // per-tile recomputation instead of registers held across the epilogue: the lane's fragment addresses (derived from
    // frow / fchunk) and, for a carried tile, its DMA offsets
    __builtin_amdgcn_sched_barrier(0);  // keep the next tile's accumulator initialisation out of this tile's epilogue
    asm volatile("" : "+v"(frow), "+v"(fchunk));
    if (carried_any) {
        asm volatile("" : "+s"(tl));
        setup_tile(xbase + tl);
    }
    f32x4_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr ((DBG & 4096) != 0 && BN == 256) {
        // ---- ping-pong schedule. The two waves of a SIMD (w and w + 4: wave column 0 / 1) run the same phase
        // sequence {L: fragment reads (+ DMA issue) | barrier | M: 32 MFMAs | barrier}, column 1 ONE BARRIER LATER:
        // while one column's waves are in their matrix section the other column's are reading LDS / issuing DMA, so
        // the matrix pipe never waits for both at once. Phase p = (k-tile p >> 1, k-step p & 1).
        //   column 0: L_p in barrier interval 2p,   M_p in 2p+1
        //   column 1: L_p in barrier interval 2p+1, M_p in 2p+2
        // k-tile kt+1 goes into the slot last read by column 1's L_{2kt-1} (interval 4kt-1): it is issued at the start
        // of each wave's L_{2kt} (intervals 4kt / 4kt+1) and must be complete before barrier 4kt+4, the first barrier
        // after which anybody (column 0, L_{2kt+2}) reads it: column 0 waits for its pieces at the end of M_{2kt+1},
        // column 1 at the end of L_{2kt+1} -- both just before that barrier.
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // k-tile 0 is in LDS for everybody
        if (wn == 1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const int fill = cur ^ 1;
            const bool do_stage = kt + 1 < nk;
            const unsigned char* sa = smem + cur * BUF_BYTES;
            const unsigned char* sb = sa + A_BYTES;
            if (do_stage) {
#pragma unroll
                for (int i = 0; i < DPT; ++i) stage_piece(fill, i);
            } else if (has_res && !REGEPI) {
                stage_residual_half(0, smem + fill * BUF_BYTES);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 xf[FM], wf[FN];
#pragma unroll
                for (int b = 0; b < FM; ++b)
                    xf[b] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * (BM / WM) + b * 16 + frow, kk * 4 + fchunk));
#pragma unroll
                for (int a = 0; a < FN; ++a)
                    wf[a] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fchunk));
                if (kk == 1 && wn == 1) wait_vmcnt<0>();
                wg_barrier();  // lgkmcnt(0): the fragments are in registers
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b) acc[a][b] = Frag<lp16_t>::mma(wf[a], xf[b], acc[a][b]);
                __builtin_amdgcn_s_setprio(0);
                if (kk == 1 && wn == 0) wait_vmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            kbyte += 128;
            cur ^= 1;
        }
        if (wn == 0) __builtin_amdgcn_s_barrier();
    } else
    for (int kt = 0; kt < nk; ++kt) {
        // k-tile kt has landed (this wave's pieces; the barrier covers everybody else's). With three slots the NEXT
        // k-tile stays in flight across the barrier (counted wait): a 48 KB k-tile is 0.56 us of MFMA, less than a
        // memory round trip, so one tile of look-ahead is latency-bound.
        if (!(DBG & 32)) {
            if (kt == 0 && carried) {  // the previous tile's stores are younger than this k-tile: leave them in flight
                if (NS == 3 && nk > 1) wait_vmcnt<DPT + EPI_STORES>();
                else wait_vmcnt<EPI_STORES>();
            } else if (NS == 3 && kt + 1 < nk) wait_vmcnt<DPT>();
            else wait_vmcnt<0>();
        }
        if (!(DBG & 128)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int fill = cur + NS - 1;  // the slot read in iteration kt-1, free now
        fill = fill >= NS ? fill - NS : fill;
        const bool do_stage = kt + NS - 1 < nk && !(DBG & 8);
        if (!do_stage && has_res && !REGEPI) stage_residual_half(0, smem + fill * BUF_BYTES);
        const unsigned char* sa = smem + cur * BUF_BYTES;
        const unsigned char* sb = sa + A_BYTES;
        // Software-pipelined fragment reads: 16 groups of 4 MFMAs (group g = k-step g>>3, channel fragment g&7). The
        // weight fragment of group g+2 and the next k-step's pixel fragments are requested before group g's MFMAs, so
        // the matrix pipe waits on the LDS only right after the barrier. One DMA piece of the next k-tile rides in each
        // of groups 0-7: spread out they hide under the matrix work (a burst of 8 stalls the wave on the memory
        // pipeline's queue) and every piece still has half a k-tile to land. sched_barrier pins that order.
        uint4 xfr[2][FM], wfr[3];
        auto ldx = [&](int kk, int b) {
            if ((DBG & 16) && kt) return make_uint4(kt, kk, b, lane);
            return *reinterpret_cast<const uint4*>(sa + lds_off(wm * (BM / WM) + b * 16 + frow, kk * 4 + fchunk));
        };
        auto ldw = [&](int g) {
            if ((DBG & 16) && kt) return make_uint4(kt, g, 1, lane);
            return *reinterpret_cast<const uint4*>(sb + lds_off(wn * (BN / 2) + (g % FN) * 16 + frow, (g / FN) * 4 + fchunk));
        };
        constexpr int NG = 2 * FN;  // groups of FM MFMAs per k-tile: group g = (k-step g / FN, channel fragment g % FN)
        wfr[0] = ldw(0);
#pragma unroll
        for (int b = 0; b < FM; ++b) xfr[0][b] = ldx(0, b);
        wfr[1] = ldw(1);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 2 < NG) wfr[(g + 2) % 3] = ldw(g + 2);
            if (g < FM) xfr[1][g] = ldx(1, g);
            if (do_stage && g < DPT) stage_piece(fill, g);
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                if (!(DBG & 2)) acc[g % FN][b] = Frag<lp16_t>::mma(wfr[g % 3], xfr[g / FN][b], acc[g % FN][b]);
                else asm volatile("" ::"v"(wfr[g % 3].x), "v"(wfr[g % 3].w), "v"(xfr[g / FN][b].x), "v"(xfr[g / FN][b].w));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        kbyte += 128;
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    if constexpr (REGEPI) {
        const int em0 = m0, en0 = n0;  // the tile whose results are in the accumulators
        bool has_next = false;
        if constexpr (!POOL && !RES) {
            tl += wstep;
            has_next = tl < xcnt;
            if (has_next) {  // the ring is free (every wave is past the barrier of the last k-tile's predecessor)
                setup_tile(xbase + tl);
                stage_head();
            }
        }
        // counted wait at the next tile's first k-tile: valid only if every one of the EPI_STORES stores is issued
        carried = has_next && em0 + BM <= p.M && (DBG & 1) == 0;
        carried_any = has_next;
        // ---- register epilogue: no LDS, no barrier. Lane (f = lane>>4, pixel = lane&15 of fragment b) holds channels
        // cb + 32 j + {0..7} (cb = n0 + 128 wn + 8 f) in acc[2j][b], acc[2j+1][b]: residual in / result out as one
        // 16-byte access per (b, j); the four lanes of a pixel cover 64 contiguous bytes per instruction.
        const int cb = en0 + wn * (BN / 2) + 8 * fchunk;
        if constexpr (F32OUT) {
            // lane (f, row = lane & 15 of fragment b): columns cb + 32 j + {0..7} of acc[2j][b], acc[2j+1][b]: two 16-byte stores,
            // the four lanes of a row cover 128 contiguous bytes per instruction
            float* __restrict__ out32 = reinterpret_cast<float*>(p.out);
            constexpr int NJ = FN / 2;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c0 = cb + 32 * j;
                float4 cv0 = make_float4(0.f, 0.f, 0.f, 0.f), cv1 = cv0;
                if (p.colv && c0 + 3 < p.N) cv0 = *reinterpret_cast<const float4*>(p.colv + c0);
                if (p.colv && c0 + 7 < p.N) cv1 = *reinterpret_cast<const float4*>(p.colv + c0 + 4);
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int gm = em0 + wm * (BM / WM) + b * 16 + frow;
                    if (gm >= p.M) continue;
                    const float rv = p.rowv ? p.rowv[gm] : p.rowc;
                    float* dst = out32 + (size_t)gm * p.ldo + c0;
                    if (c0 + 3 < p.N)
                        *reinterpret_cast<float4*>(dst) = make_float4(fmaf(p.alpha, acc[2 * j][b][0], rv + cv0.x), fmaf(p.alpha, acc[2 * j][b][1], rv + cv0.y),
                                                                      fmaf(p.alpha, acc[2 * j][b][2], rv + cv0.z), fmaf(p.alpha, acc[2 * j][b][3], rv + cv0.w));
                    if (c0 + 7 < p.N)
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(fmaf(p.alpha, acc[2 * j + 1][b][0], rv + cv1.x), fmaf(p.alpha, acc[2 * j + 1][b][1], rv + cv1.y),
                                                                          fmaf(p.alpha, acc[2 * j + 1][b][2], rv + cv1.z), fmaf(p.alpha, acc[2 * j + 1][b][3], rv + cv1.w));
                }
            }
            carried = false;  // (the counted wait of a carried tile assumes EPI_STORES stores per lane: not with masked fp32 stores)
            if (!has_next) return;
            continue;
        }
        const lp16_t* __restrict__ resp = reinterpret_cast<const lp16_t*>(p.res);
        lp16_t* __restrict__ outp = reinterpret_cast<lp16_t*>(p.out);
        constexpr int NJ = FN / 2;  // 16-byte pieces (8 channels) per lane and pixel fragment
        // every residual piece is requested before the first store: loads and stores retire in order on one counter, so a
        // load behind a store would wait for that store's acknowledgement
        uint4 rres[FM][NJ];
        if (has_res) {
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int gm = min(em0 + wm * (BM / WM) + b * 16 + frow, p.M - 1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) rres[b][j] = *reinterpret_cast<const uint4*>(resp + (size_t)gm * p.ldo + cb + 32 * j);
            }
        }
        if (p.colv) {  // bias first, eight values in registers at a time (same order of operations: acc + bias, then + residual)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.colv + cb + 32 * j);
                const float4 b1 = *reinterpret_cast<const float4*>(p.colv + cb + 32 * j + 4);
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    acc[2 * j][b][0] += b0.x; acc[2 * j][b][1] += b0.y; acc[2 * j][b][2] += b0.z; acc[2 * j][b][3] += b0.w;
                    acc[2 * j + 1][b][0] += b1.x; acc[2 * j + 1][b][1] += b1.y; acc[2 * j + 1][b][2] += b1.z; acc[2 * j + 1][b][3] += b1.w;
                }
            }
        }
        float psum[POOL ? 2 : 1][NJ][8];
        if constexpr (POOL) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) psum[q][j][e] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int gm = em0 + wm * (BM / WM) + b * 16 + frow;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v[8];
                v[0] = acc[2 * j][b][0]; v[1] = acc[2 * j][b][1]; v[2] = acc[2 * j][b][2]; v[3] = acc[2 * j][b][3];
                v[4] = acc[2 * j + 1][b][0]; v[5] = acc[2 * j + 1][b][1]; v[6] = acc[2 * j + 1][b][2]; v[7] = acc[2 * j + 1][b][3];
                if (has_res) {
                    const uint32_t w4[4] = {rres[b][j].x, rres[b][j].y, rres[b][j].z, rres[b][j].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        unpack_lp16x2(w4[e], lo, hi);
                        v[2 * e] += lo;
                        v[2 * e + 1] += hi;
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
                }
                const uint4 pk = make_uint4(pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7]));
                if (gm < p.M && (!POOL || p.pool_store_out) && (!(DBG & 1) || p.relu == 12345)) *reinterpret_cast<uint4*>(outp + (size_t)gm * p.ldo + cb + 32 * j) = pk;
                if constexpr (POOL) {  // pool the bf16-rounded activations (what a separate pooling pass would read)
                    const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
                    const float live = gm < p.M ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        unpack_lp16x2(w4[e], lo, hi);
                        psum[b >> 1][j][2 * e] += live * lo;
                        psum[b >> 1][j][2 * e + 1] += live * hi;
                    }
                }
            }
        }
        if constexpr (POOL) {
            // vmgn.py:298-308. This wave's 64 pixels are image rows 8 (wm & 1) .. +7 of frame (wm >> 1): fragments
            // b = 0,1 / 2,3 are two QUARTER bins (4 image rows = 32 pixels each). Sum over the 16 pixel lanes of a
            // fragment (same f), park the quarter sums in LDS, then every output bin is a sum of whole quarters.
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = psum[q][j][e];
                        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                        psum[q][j][e] = t;
                    }
            wg_barrier();  // every wave is past its last fragment read: the ring can be overwritten
            float* s_q = reinterpret_cast<float*>(smem);  // [2 frames][4 quarters][256 channels]
            if (frow == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float* d = s_q + (((wm >> 1) * 4 + 2 * (wm & 1) + q) * BN) + wn * (BN / 2) + 8 * fchunk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *reinterpret_cast<float4*>(d + 32 * j) = make_float4(psum[q][j][0], psum[q][j][1], psum[q][j][2], psum[q][j][3]);
                        *reinterpret_cast<float4*>(d + 32 * j + 4) = make_float4(psum[q][j][4], psum[q][j][5], psum[q][j][6], psum[q][j][7]);
                    }
                }
            }
            wg_barrier();
            const int P = p.pool_nparts;
            for (int o = tid; o < 2 * P * BN; o += 512) {
                const int c = o % BN, fp = o / BN, part = fp % P, fr = fp / P;
                const int frame = (em0 >> 7) + fr;
                if (frame * 128 >= p.M) continue;
                const int q0 = p.pool_start[part] >> 2, q1 = p.pool_end[part] >> 2;  // bins are whole quarters (host-checked)
                float t = 0.f;
                for (int q = q0; q < q1; ++q) t += s_q[(fr * 4 + q) * BN + c];
                if (p.pool_mean) t *= 1.f / (float)((q1 - q0) * 32);
                const size_t oi = ((size_t)frame * P + part) * p.N + en0 + c;
                p.pool_out[oi] = t;
                if (p.pool_out_lp) reinterpret_cast<lp16_t*>(p.pool_out_lp)[oi] = f32_to_lp16(t);
            }
        }
        if (!has_next) return;
    }
    if constexpr (!REGEPI) {
    // cur = slot F (free since the last iteration: holds residual half 0), cur ^ 1 = slot L (the last k-tile)
    unsigned char* soF = smem + cur * BUF_BYTES;
    unsigned char* soL = smem + (cur ^ 1) * BUF_BYTES;
    wait_vmcnt<0>();
    wg_barrier();  // every fragment read of slot L is done
    if (has_res) stage_residual_half(1, soL);

    auto combine_half = [&](int h, unsigned char* so) {
#pragma unroll
        for (int aa = 0; aa < FN / 2; ++aa) {
            const int a = h * (FN / 2) + aa;
            const int vc = wn * 64 + aa * 16 + fchunk * 4;  // channel inside the half image (0..127)
            float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.colv) cv = *reinterpret_cast<const float4*>(p.colv + n0 + wn * (BN / 2) + a * 16 + fchunk * 4);
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int prow = wm * (BM / WM) + b * 16 + frow;
                unsigned char* slot = so + prow * 256 + (((vc >> 3) ^ (prow & 15)) << 4) + ((vc & 4) << 1);
                float v[4];
                v[0] = acc[a][b][0] + cv.x;
                v[1] = acc[a][b][1] + cv.y;
                v[2] = acc[a][b][2] + cv.z;
                v[3] = acc[a][b][3] + cv.w;
                if (has_res) {
                    float rr[4];
                    load4<lp16_t>(reinterpret_cast<const lp16_t*>(slot), rr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                }
                if (p.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = relu_nan(v[r]);
                }
                store4<lp16_t>(reinterpret_cast<lp16_t*>(slot), v);
            }
        }
    };
    auto drain_half = [&](int h, const unsigned char* so) {  // whole 16-byte chunks, 4 full 128-byte lines per wave op
        const int pch = tid & 15;
        const int r0 = tid >> 4;
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int row = r0 + i * 32;
            const int vch = pch ^ (row & 15);
            const int gm = m0 + row;
            if (gm < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(so + row * 256 + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + ((size_t)gm * p.ldo + half_col(h, vch)) * 2) = v;
            }
        }
    };

    combine_half(0, soF);
    wait_vmcnt<0>();  // residual half 1 has landed
    wg_barrier();     // half 0 complete
    drain_half(0, soF);
    combine_half(1, soL);
    wg_barrier();
    drain_half(1, soL);
    return;
    }
    }  // tile loop
}

}  // namespace

// fp32-output distance form (F32OUT): 16-bit operands, any M, N % 4 == 0, no residual / pooling / second source
bool igemm_wide_f32out_applicable(const IgemmParams& p) {
    const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;
    if (!pointwise || p.x2 || p.res || p.pool_nparts > 0 || p.ksplit > 1 || p.mix_f || p.stats || p.relu) return false;
    if (p.K % 64 || p.N % 4 || p.ldo % 4) return false;
    if ((size_t)p.M * p.K * 2 >= (1ull << 32) || (size_t)p.N * p.K * 2 >= (1ull << 32)) return false;
    const uintptr_t al = (uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.out | (uintptr_t)p.colv;
    return (al & 15) == 0;
}

int launch_igemm_wide_f32out(const IgemmParams& p, hipStream_t stream, const char* who) {
    // One workgroup per tile and per CU: the launch takes ceil(tiles / CUs) rounds of a tile's time, and a 256 x 192 tile costs 3/4
    // of a 256 x 256 one. The MARS matrix (1980 x 12 180: 8 x 48 = 384 tiles of 256 columns = 1.5 rounds, paid as 2) is 8 x 64 = 512
    // tiles of 192 columns = 2 exact rounds at 3/4 of the cost each: take the 192-column tile whenever it is cheaper by that count.
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int mt = cdiv(p.M, WBM);
    const int tiles256 = mt * cdiv(p.N, 256), tiles192 = mt * cdiv(p.N, 192);
    const long long cost256 = (long long)cdiv(tiles256, cus) * 256, cost192 = (long long)cdiv(tiles192, cus) * 192;
    const int force = agrl_opts().distmat_tile_n;   // AGRL_DISTMAT_TILE_N = 192 / 256: A/B
    const bool use192 = agrl_opt_set(force) ? force == 192 : cost192 < cost256;
    if (use192) hipLaunchKernelGGL((igemm_wide_kernel<65536, 192, false>), dim3(tiles192), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((igemm_wide_kernel<65536, 256, false>), dim3(tiles256), dim3(512), 0, stream, p);
    AGRL_CHECK_LAUNCH(who);
    return 0;
}

bool igemm_wide_applicable(const IgemmParams& p) {
    if (p.x2 && (p.stride != 1 || p.res || p.pool_nparts > 0 || p.K1 <= 0 || p.K1 != 2 * (p.K - p.K1) || (p.K1 % 64) || (p.N % WBN) ||
                 (((uintptr_t)p.x2) & 15)))
        return false;
    const bool pointwise = p.R == 1 && p.S == 1 && p.stride >= 1 && p.pad == 0;  // stride > 1: rows are gathered
    if (!pointwise || p.rowv || p.ksplit > 1) return false;
    if (p.stride > 1 && (p.pool_nparts > 0 || (size_t)(p.M / (p.OH * p.OW)) * p.H * p.W * p.K * 2 >= (1ull << 32))) return false;
    if (p.pool_nparts > 0) {  // fused pooling: 16 x 8 frames, two per tile, bins made of whole 4-row quarters
        if (p.OH != 16 || p.OW != 8 || p.pool_w != 8) return false;
        for (int i = 0; i < p.pool_nparts; ++i)
            if ((p.pool_start[i] & 3) || (p.pool_end[i] & 3)) return false;
    }
    if (p.alpha != 1.f || p.rowc != 0.f) return false;
    if (p.N % 128 || p.K % 64 || p.ldo % 8) return false;
    if (p.N % WBN && p.pool_nparts > 0) return false;
    if ((size_t)p.M * p.K * 2 >= (1ull << 32) || (size_t)p.N * p.K * 2 >= (1ull << 32)) return false;
    const uintptr_t al = (uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.colv;
    return (al & 15) == 0;
}

int launch_igemm_wide(const IgemmParams& p, hipStream_t stream, const char* who) {
    // 128-channel tiles where 256-channel ones would leave CUs idle (N = 256 layers) or do not divide N
    bool half_n = (p.N % WBN) != 0 || (p.pool_nparts == 0 && p.dbg == 0 && !p.x2 && cdiv(p.M, WBM) * (p.N / WBN) < 224);
    const AgrlOpts& opt = agrl_opts();
    // AGRL_IGEMM_WIDE = 2 / 3 force the 256- / 128-channel tile (tests, A/B)
    if (opt.igemm_wide == 2 && (p.N % WBN) == 0) half_n = false;
    if (opt.igemm_wide == 3 && p.pool_nparts == 0 && !p.x2) half_n = true;
    // persistent form: one workgroup per CU walks its share of the tiles (AGRL_IGEMM_WIDE_PERSIST=0: one per tile)
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const bool persist = true;  // fewer workgroups than tiles: each walks its XCD's range (one workgroup per tile measured slower)
    const bool res = p.res != nullptr;
    if (half_n) {
        const int tiles = cdiv(p.M, WBM) * (p.N / 128);
        if (res) hipLaunchKernelGGL((igemm_wide_kernel<0, 128, true>), dim3(tiles), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((igemm_wide_kernel<0, 128, false>), dim3(persist ? min(tiles, n_cu) : tiles), dim3(512), 0, stream, p);
        AGRL_CHECK_LAUNCH(who);
        return 0;
    }
    const int tiles256 = cdiv(p.M, WBM) * (p.N / WBN);
    const int grid = persist && !res && p.pool_nparts == 0 && (p.dbg & (4096 | 8192)) == 0 ? min(tiles256, n_cu) : tiles256;
    if (p.pool_nparts > 0) {
        if (res) hipLaunchKernelGGL((igemm_wide_kernel<16384, 256, true>), dim3(grid), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((igemm_wide_kernel<16384, 256, false>), dim3(grid), dim3(512), 0, stream, p);
        AGRL_CHECK_LAUNCH(who);
        return 0;
    }
    if (p.x2) {  // two-source form (agrl_conv1x1_dual_bn_act): no residual, no pooling, 256-channel tiles
        hipLaunchKernelGGL((igemm_wide_kernel<32768, 256, false>), dim3(grid), dim3(512), 0, stream, p);
        AGRL_CHECK_LAUNCH(who);
        return 0;
    }
    if (res) {
#ifdef AGRL_ABLATE
        if (p.dbg == 8192) hipLaunchKernelGGL((igemm_wide_kernel<8192, 256, true>), dim3(grid), dim3(512), 0, stream, p);
        else
#endif
        hipLaunchKernelGGL((igemm_wide_kernel<0, 256, true>), dim3(grid), dim3(512), 0, stream, p);
        AGRL_CHECK_LAUNCH(who);
        return 0;
    }
    switch (p.dbg) {  // non-zero only in an -DAGRL_ABLATE build
#ifdef AGRL_ABLATE
#define WIDE_CASE(D) case D: hipLaunchKernelGGL((igemm_wide_kernel<D, 256>), dim3(grid), dim3(512), 0, stream, p); break
        WIDE_CASE(1); WIDE_CASE(16384); WIDE_CASE(4096); WIDE_CASE(8192); WIDE_CASE(178); WIDE_CASE(50); WIDE_CASE(146); WIDE_CASE(2); WIDE_CASE(8); WIDE_CASE(16); WIDE_CASE(32); WIDE_CASE(160); WIDE_CASE(184); WIDE_CASE(18);
#undef WIDE_CASE
#endif
        default: hipLaunchKernelGGL((igemm_wide_kernel<0, 256>), dim3(grid), dim3(512), 0, stream, p);
    }
    AGRL_CHECK_LAUNCH(who);
    return 0;
}
