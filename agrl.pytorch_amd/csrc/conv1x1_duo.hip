// The last conv of a layer-4 branch -- conv3 / bn3 + identity shortcut + ReLU (torchreid/models/vmgn.py:56-64, 512 -> 2048 on 16 x 8
// maps) with the frame pooling of vmgn.py:298-308 in its epilogue (agrl_conv1x1_bn_act_pool's job: the 2048-channel map of a
// branch's last block never exists) -- and the same conv with the map stored, 16-bit build type, as TWO CO-RESIDENT four-wave
// workgroups per CU.
//   out (M, Cout) = act(x (M, K) @ W (Cout, K)^T + bias + residual (M, Cout))
// This layer is 68.7 GFLOP against 302 MB (33 MB of rows, 134 MB of residual, 134 MB of result; pooled: 168 MB): ~45 us of matrix
// work and ~55 us of HBM time, and every form of rounds 1-4 (igemm_wide_kernel<0, 256, true>, conv1x1_fat.hip with the residual as
// slabs / in registers / persistent, the back-to-back seam kernel) ran at their SUM, 105-121 us. Round 5 measured why, and what
// does and does not help (tools/conv1x1_duo_bench.py, tools/duo_timeline.py, tools/ubench/mix_stream.hip):
//   * THIS form: a workgroup is 4 waves x <= 256 registers x 64 KB of LDS, so a CU holds two; one's epilogue runs under the other's
//     k-loop on different waves (different memory counters). Tile = 128 pixel rows (one frame) x 256 channels; wave w owns 64
//     channels (4 MFMA A fragments) of all 128 pixels (8 B fragments): 32 accumulator quads = 128 asm-owned AGPRs; weights from
//     agrl_conv1x1_pack's per-(channel tile, wave) fragment streams (the SAME packed tensor serves conv1x1_fat.hip) global -> 8-
//     fragment VGPR ring; pixel rows of the current / next 128-channel slab in LDS (2 x 32 KB) by LDS-DMA, one barrier per slab.
//     Epilogue through a wave-private 16 KB LDS image of the wave's 128 rows x 128 B in the pixel buffers' swizzled layout, so that
//     every global access is a whole 128-byte line per eight lanes: residual in by LDS-DMA, combined in place in the MFMA layout
//     (+ bias, + residual, ReLU, one rounding: igemm_wide_kernel's order of operations -- results BIT-IDENTICAL to
//     agrl_conv2d_bn_act(residual=...) / agrl_conv1x1_bn_act_pool), result out row by row. POOL: the rounded activations' quarter
//     sums (4 image rows) per lane, 16-lane shuffles, quarters through LDS, bins = sums of whole quarters (igemm_wide_kernel<16384>'s
//     order). Measured, same box, inside a Bottleneck (conv1 and the 3x3 run before every timed call), HIP events, this / wide:
//     POOLED 92.0 / 101.2 us (global branch), the model's dispatch; map stored 117.4 / 119.5 us (back to back 115.6 / 122.4): dispatched; WITHOUT a
//     residual (layer 4's conv1s, tools/conv1x1_duo_vs_fat.py: this / conv1x1_fat_kernel / wide) 2048 -> 512 69.2 / 72.7 / 77.5 us,
//     1024 -> 512 43.5 / 45.4 / 45.4 -- inside the step equal mid-round (six conv1 launches 12 us slower by events), 7-13 us per step AHEAD on the final tree (two boxes, eight A/B pairs): dispatched late in round 5; 512 -> 256 on 32 x 16 maps 56 / 60.5 / 54: not; the TWO-SOURCE form (conv3 + downsample
//     conv of a first block over [x | y2], agrl_conv1x1_packed_dual_duo) 167.1 us against conv1x1_fat_kernel's 185.2: dispatched.
//     Ablations of the stored form (127 us with the accumulator-layout epilogue): no result stores 76, no residual loads 110, neither
//     62, no MFMA 124 (!), no weight loads 107, no pixel DMA 114, one workgroup per CU 142. Timeline of a workgroup (s_memtime): 5.3 us
//     from start to the first barrier (first loads), k-loop 9.0, residual wait 2.6, combine 4.5, stores 2.4.
//   * a weight ring of 12 fragments instead of 8 (248 / 252 registers; the ring then rotates through the 16-fragment slab and the slab body
//     exists three times): bit-identical, layer 4's pointwise family 0.981-0.982 ms against 0.974-0.978 in the step (three A/B pairs, one
//     box) -- no gain, not kept: the k-loop does not wait for its weights.
//   * second form (git history, c2f7fb3): ONE persistent 8-wave workgroup per frame walking its eight channel tiles, four matrix
//     waves (k-loop + combine, the fragment stream running on across tiles, an LDS arrival counter instead of s_barrier for the
//     pixel buffers) + four memory waves bringing the residual into the image and streaming the result out under the next tile's
//     k-loop. Bit-identical; 138 / 120 us stored, 113 / 100 pooled -- SLOWER: with the memory waves' traffic in flight the matrix
//     waves' k-loop takes 14 us per tile instead of 4-5.5.
//   * why: tools/ubench/mix_stream.hip -- three kinds of traffic issued by DIFFERENT waves of every CU (3 MB of L2 hits, 0.5 MB of
//     HBM reads, 0.5 MB of HBM writes per CU: this layer's budget) take 31 / 27 / 25 us alone and 47-52 us in pairs, 73 us all three:
//     a CU's vector-memory pipe serves them ADDITIVELY, whoever issues them. An operand ring of 8 fragments (1 us of cover) cannot
//     ride out the queueing behind HBM-latency requests, and the LDS cannot hold a tile's worth of operands ahead. What is left
//     is fewer bytes per CU (the 256 x 256 tile of igemm_wide_kernel moves 3 MB per CU, this 128 x 256 tile 4 MB) or fewer HBM
//     bytes per launch (the pooled form: no result map) -- which is where this kernel wins.
#include "fat_dev.h"

namespace {

struct DuoParams {
    const unsigned char* x;     // (M, K1) 16-bit pixel rows
    const unsigned char* x2;    // (M, K - K1) second source (the K axis is [x | x2]) or nullptr
    int K1;                     // columns of x (= K without a second source)
    const unsigned char* wpk;   // packed weight streams (agrl_conv1x1_pack)
    const float* bias;          // (Cout)
    const unsigned char* res;   // (M, Cout) residual or nullptr
    unsigned char* out;         // (M, Cout) or nullptr (POOL only)
    int M, K, Cout, relu;
    // strided first source (the stride-s 1x1 downsample conv of a first block): x is the (frames, Hi, Wi, K1) map and row m = (f, ho, wo)
    // of the GEMM reads its pixel (f, s ho, s wo); gHoWo = 0: x holds the M rows themselves
    int gHoWo, gWo, gS, gWi, gHiWi;
    // POOL
    float* pool_out;            // fp32 (frames, nparts, Cout)
    unsigned short* pool_out_lp;  // optional 16-bit copy
    int pool_nparts, pool_mean;
    int pool_q0[16], pool_q1[16];  // bins in quarters (4 image rows) [q0, q1)
    // split-fp16 planes (round 6, agrl_conv1x1_split16*): x / x2 / res / out hold three fp16 planes per pixel, [hi | lo 2^11 | hi] -- for the
    // k-loop simply rows of 3 x the channels (K counts them); res / out: (M, 3 Cout). alpha un-does the weights' power-of-two pre-scale.
    float alpha;
    int planes;
    // plane PAIRS (round 6, late): a wide tensor (a block's input / output: the residual stream) holds [hi | lo 2^11] only -- the third
    // plane repeats the first, and every kernel here is HBM-bound on exactly these tensors. x_pair: x is a pair of K1 / 3 channels and the
    // K axis of its weights runs per 128-channel slab c as [hi_c | hi_c | lo_c] (against [wh_c | wl_c | wh_c 2^-11]): the repeated slab is
    // requested again right behind its first use -- an L2 hit, not an HBM read. ro_pair: res / out are pairs, (M, 2 Cout).
    int x_pair, ro_pair;
};

#ifndef DUO_ABL
#define DUO_ABL 0  // timing ablations (results wrong; tools/duo_ablate.sh): 1 no residual loads, 2 no stores, 4 no MFMA, 8 no weight loads in the loop, 16 no pixel DMA in the loop, 32 one workgroup per CU (48 KB of dynamic LDS on top), 64 phase stamps (agrl_duo_trace_buffer, tools/duo_timeline.py)
#endif
constexpr int DRING = 8;                 // weight fragments in flight per wave
constexpr int DPS = 4 * 4;               // weight fragments per 128-channel slab and wave: 4 k-steps x 4 channel fragments
constexpr int DROWS = 128;               // pixel rows per tile
constexpr int DHALF = DROWS * 128;       // 128 pixel rows x 64 channels
constexpr int DSLAB = 2 * DHALF;         // one 128-channel slab of the pixel tile: 32 KB
constexpr int DPPW = 8;                  // DMA pieces (8 rows x 128 B) per wave and slab
constexpr int DBARRIER_AT = 12;          // see conv1x1_fat.hip
constexpr int dpieces_at(int p) { return p >= DBARRIER_AT ? 2 : 0; }
constexpr int dpiece_first(int p) { int n = 0; for (int q = 0; q < p; ++q) n += dpieces_at(q); return n; }
static_assert(dpiece_first(DPS) == DPPW && DBARRIER_AT >= DRING, "all pieces placed, behind fragments whose successors' ring slots the prologue fills");
struct DuoSched {
    int allowed[DPS];
};
constexpr DuoSched make_duo_sched() {  // vmcnt budget of the wait in front of fragment p of a slab (steady state)
    DuoSched s{};
    int issued[4][DPS] = {};
    int seq = 0;
    for (int p = 0; p < DRING; ++p) issued[0][p] = seq++;
    for (int k = 0; k < 3; ++k)
        for (int p = 0; p < DPS; ++p) {
            if (k == 1) s.allowed[p] = seq - 1 - issued[k][p];
            const int q = p + DRING;
            if (q >= DPS) issued[k + 1][q - DPS] = seq++;
            else issued[k][q] = seq++;
            seq += dpieces_at(p);  // the next slab's pieces
        }
    return s;
}
struct DuoSchedOf {
    static constexpr DuoSched value = make_duo_sched();
};

#if DUO_ABL & 64   // profiling build: per-workgroup phase stamps (s_memtime) + placement, 12 x 8 bytes per workgroup
__device__ unsigned long long* g_duo_trace = nullptr;
#define DUO_STAMP(k)                                                                          \
    do {                                                                                      \
        if (g_duo_trace && tid == 0) g_duo_trace[(size_t)blockIdx.x * 12 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define DUO_STAMP(k) do { } while (0)
#endif

template <bool POOL, bool PLANES = false>   // PLANES: the split-fp16 epilogue (its own instantiations: both epilogues in one kernel do not fit the 128 arch VGPRs beside the 128 asm-owned AGPRs)
__global__ __launch_bounds__(256, 2) void conv1x1_duo_kernel(const DuoParams p) {
    using SCHED = DuoSchedOf;
    using std::integral_constant;
    __shared__ __attribute__((aligned(16))) unsigned char smem_[2 * DSLAB];
    // pooled planes form: the quarter sums of a pass are parked here (4 waves x [4 quarters][64 channels]) -- the wave's image is the
    // residual planes' landing zone in both passes, and carrying all 64 sums per lane through them overflows the 128 arch VGPRs
    __shared__ __attribute__((aligned(16))) float s_pool_[(POOL && PLANES) ? 4 * 256 : 4];
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int frow = lane & 15, fchunk = lane >> 4;

    DUO_STAMP(0);
#if DUO_ABL & 64
    if (g_duo_trace && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_duo_trace[(size_t)blockIdx.x * 12 + 10] = hw;
        g_duo_trace[(size_t)blockIdx.x * 12 + 11] = xcc;
    }
#endif
    // tile = (pixel tile mt, channel tile nt): neighbouring workgroups (same XCD: blockIdx % 8) share the pixel tile
    DUO_STAMP(1);
    const int nNt = p.Cout >> 8;
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, within = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const int m0 = mt * DROWS;

    // ---- pixel staging: piece i = 2 j + h of this wave -> rows (wave + 4 j) * 8 .. + 7 of 64-channel half h; lane (lrow = lane >> 3,
    // lchk = lane & 7) fetches chunk lchk ^ swizzle(row) of its row (16-byte chunk c of row r at c ^ ((r >> 1) & 7))
    unsigned roff[4], roff2[4];
    const int lrow = lane >> 3, lchk = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave + 4 * j) * 8 + lrow;
        const int gm = min(m0 + row, p.M - 1);
        const unsigned sw = (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
        unsigned srow = (unsigned)gm;
        if (p.gHoWo) {
            const unsigned f = (unsigned)gm / (unsigned)p.gHoWo, r = (unsigned)gm - f * (unsigned)p.gHoWo;
            const unsigned ho = r / (unsigned)p.gWo, wo = r - ho * (unsigned)p.gWo;
            srow = f * (unsigned)p.gHiWi + (ho * (unsigned)p.gWi + wo) * (unsigned)p.gS;
        }
        unsigned rowlen1 = (unsigned)p.K1;
        if constexpr (PLANES) rowlen1 = p.x_pair ? (unsigned)p.K1 / 3u * 2u : (unsigned)p.K1;
        roff[j] = srow * rowlen1 * 2u + sw;
        roff2[j] = (unsigned)gm * (unsigned)(p.K - p.K1) * 2u + sw;
    }
    const int nslab = p.K >> 7, nslab1 = p.K1 >> 7;
    auto stage_piece = [&](int slab, int buf, auto i_c) {  // piece i of slab `slab` into buffer `buf`
        constexpr int I = decltype(i_c)::value, J = I >> 1, H = I & 1;
        const bool second = slab >= nslab1;  // uniform: the slab lies in x2 (two-source form: conv3 + downsample conv as one GEMM)
        int slab1 = slab;
        if constexpr (PLANES) {
            if (p.x_pair) {   // (scalar) slab 3 c + r of the K axis -> slab c of the hi plane (r = 0, 1) or of the lo plane (r = 2)
                const int c = slab / 3, r = slab - 3 * c;
                slab1 = r == 2 ? nslab1 / 3 + c : c;
            }
        }
        const unsigned char* src = second ? p.x2 + roff2[J] + (size_t)((slab - nslab1) * 256 + H * 128) : p.x + roff[J] + (size_t)(slab1 * 256 + H * 128);
        fat_dma(src, __builtin_amdgcn_readfirstlane(lds0 + buf * DSLAB + H * DHALF + (wave + 4 * J) * 1024));
    };

    // ---- pixel fragment b (rows 16 b + (lane & 15)) of k-step kk: half kk >> 1, chunk 4 (kk & 1) + (lane >> 4)
    const int xbase = frow * 128 + ((fchunk ^ ((frow >> 1) & 7)) << 4);

    // ---- weight stream of this wave: fragment q of slab s at wpk + ((nt * 4 + wave) * nslab * DPS + s * DPS + q) KiB
    const unsigned char* wstream = p.wpk + (size_t)(nt * 4 + wave) * nslab * (DPS * 1024);
#if DUO_ABL   // (profiling builds: with parts of the loop compiled out hipcc no longer proves the stream pointer uniform)
    wstream = reinterpret_cast<const unsigned char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((size_t)wstream >> 32)) << 32) |
                                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)wstream));   // (the builtin returns a SIGNED int)
#endif
    u32x4_t wr[DRING];
    auto issue_w = [&](auto slot_c, const unsigned char* slab_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        fat_gload<(POS & 3) * 1024>(wr[SLOT], lane16, slab_base + (POS & ~3) * 1024);
    };

    asm volatile("" ::: "a127");
    sfor<32>([&](auto qc) { fat_zero<decltype(qc)::value>(); });

    // ---- prologue: slab 0's pixel rows; then the first ring of weight fragments with slab 1's pieces behind fragments 4 .. 7 --
    // the order the loop issues them in behind fragments 12 .. 15 of the slab before, so that its counted waits hold from slab 0 on
    sfor<DPPW>([&](auto ic) { stage_piece(0, 0, ic); });
    sfor<DRING>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        issue_w(ic, wstream, ic);
        sfor<dpieces_at(I + DRING)>([&](auto jc) {
            stage_piece(nslab > 1 ? 1 : 0, 1, integral_constant<int, dpiece_first(I + DRING) + decltype(jc)::value>{});
        });
    });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DRING + DPPW) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    DUO_STAMP(2);
    u32x4_t xf[8];
    auto ldx = [&](const lds_u8_t* sp, auto ks_c, auto b_c) {
        constexpr int KS = decltype(ks_c)::value, B = decltype(b_c)::value;
        const lds_u8_t* a = sp + (xbase ^ ((KS & 1) * 64));
        return *reinterpret_cast<const lds_u32x4_t*>(a + (KS >> 1) * DHALF + B * 2048);
    };
    sfor<8>([&](auto bc) { xf[decltype(bc)::value] = ldx(smem, integral_constant<int, 0>{}, bc); });
    for (int slab = 0; slab < nslab; ++slab) {
        const bool more = slab + 1 < nslab;
        const unsigned char* ws = wstream + (size_t)slab * (DPS * 1024);
        const unsigned char* wsn = wstream + (size_t)(more ? slab + 1 : 0) * (DPS * 1024);  // past the end: slab 0 again (never used)
        const int ahead = slab + 2 < nslab ? slab + 2 : slab;  // (last two slabs: their own rows again, into the freed buffer)
        const lds_u8_t* sp = smem + (slab & 1) * DSLAB;
        const lds_u8_t* spn = smem + ((slab + 1) & 1) * DSLAB;

        sfor<DPS>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            constexpr int KS = P >> 2, A = P & 3, SL = P % DRING;
            fat_wait<SCHED::value.allowed[P]>(wr[SL]);
            if constexpr (P == DBARRIER_AT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            sfor<8>([&](auto bc) {
                constexpr int B = decltype(bc)::value;
                if constexpr (!(DUO_ABL & 4)) fat_mfma<A * 8 + B>(wr[SL], xf[B]);
                if constexpr (A == 3) {  // the next k-step's fragment replaces this one right behind its last reader
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (KS + 1 < 4) xf[B] = ldx(sp, integral_constant<int, KS + 1>{}, bc);
                    else xf[B] = ldx(spn, integral_constant<int, 0>{}, bc);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            constexpr int Q = P + DRING;
            if constexpr (!(DUO_ABL & 8)) {
            if constexpr (Q >= DPS) issue_w(integral_constant<int, SL>{}, wsn, integral_constant<int, Q - DPS>{});
            else issue_w(integral_constant<int, SL>{}, ws, integral_constant<int, Q>{});
            }
            if constexpr (!(DUO_ABL & 16))
            sfor<dpieces_at(P)>([&](auto ic) { stage_piece(ahead, slab & 1, integral_constant<int, dpiece_first(P) + decltype(ic)::value>{}); });
        });
    }
    DUO_STAMP(3);
    // fragments and pieces requested past the end are still landing
#pragma unroll
    for (int i = 0; i < DRING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < DRING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // ---- epilogue: + bias, + residual, ReLU, round once -- through a wave-private 16 KB LDS image of the wave's 128 rows x 128 B
    // (its 64 channels) in the pixel buffers' swizzled row layout, so that EVERY global access is a whole 128-byte line per eight
    // lanes. (Straight from the accumulator layout a 16-lane group touches 16 different lines with 16 bytes each: measured, the
    // result stores alone then cost 51 of 127 us and the residual loads 17.) Residual in by LDS-DMA, combined in place in the
    // MFMA layout (conflict-free: the pixel fragments' addressing), result out row by row. No barrier but the one below.
    __syncthreads();  // every wave is past its last fragment read: the pixel buffers are free
    DUO_STAMP(4);
    lds_u8_t* const wt = smem + wave * 16384;
    const size_t colb = (size_t)(nt * 256 + wave * 64) * 2;  // byte column of the wave's 128-byte row segments
    const bool has_res = p.res != nullptr && !(DUO_ABL & 1);
    const bool has_out = (!POOL || p.out != nullptr) && !(DUO_ABL & 2);
    auto row_off = [&](int i) {  // piece i = rows 8 i + lrow: byte offset of this lane's 16 bytes (chunk lchk ^ swizzle(row)) in res / out
        const int row = 8 * i + lrow;
        const int gm = min(m0 + row, p.M - 1);
        return (size_t)gm * p.Cout * 2 + colb + (size_t)((lchk ^ ((row >> 1) & 7)) << 4);
    };
    const int cb = nt * 256 + wave * 64 + 8 * fchunk;
    const float alpha = p.alpha;
    float psum[POOL ? 4 : 1][2][8];
    if constexpr (POOL) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) psum[q][j][e] = 0.f;
    }
    if constexpr (PLANES) {
        // ---- split-fp16 planes: the fp32 value v = alpha acc + bias (+ residual hi + residual lo 2^-11) (ReLU) leaves as fp16 hi = fp16(v)
        // and lo = fp16((v - hi) 2^11). The wave's 16 KB image holds 64 rows at a time: hi rows in its lower half, lo rows in the upper
        // half (same swizzled row layout), so two passes of 64 rows; the residual's two planes arrive in the same two halves by LDS-DMA
        // and each cell pair is combined in place. POOL: the UNROUNDED v is pooled (what the fp32 reference pools, vmgn.py:298-308).
        const size_t ld3 = (size_t)p.Cout * (p.ro_pair ? 2 : 3) * 2;   // bytes per pixel row of a planes tensor (triple or pair)
        const size_t plane = (size_t)p.Cout * 2;
        // (opaque copies: hipcc would otherwise compute the sixteen request offsets before the k-loop and carry them through it, and
        // the pooled instantiation then overflows its 128 arch VGPRs into AGPRs the asm blocks own)
        int lrow_e = lrow, lchk_e = lchk;
        asm volatile("" : "+v"(lrow_e), "+v"(lchk_e));
        auto row_off3 = [&](int i) {  // piece i = rows 8 i + lrow of the TILE: byte offset of this lane's 16 bytes in plane 0 of res / out
            const int row = 8 * i + lrow_e;
            const int gm = min(m0 + row, p.M - 1);
            return (size_t)gm * ld3 + colb + (size_t)((lchk_e ^ ((row >> 1) & 7)) << 4);
        };
        sfor<2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            float ps[POOL ? 2 : 1][2][8];   // this pass's quarters 2 h, 2 h + 1
            if constexpr (POOL) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) ps[q][j][e] = 0.f;
            }
            if (has_res) {
                // (scalar base + 32-bit lane offset: with 64-bit lane addresses for 16 requests beside the live pooling sums the
                // instantiation would not fit its 128 arch VGPRs; the entry points bound every map by 4 GB)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned off = (unsigned)row_off3(8 * h + i);
                    fat_dma_s(p.res, off, __builtin_amdgcn_readfirstlane(lds0 + wave * 16384 + i * 1024));
                    fat_dma_s(p.res + plane, off, __builtin_amdgcn_readfirstlane(lds0 + wave * 16384 + 8192 + i * 1024));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sfor<2>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j);
                const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j + 4);
                sfor<4>([&](auto bc) {
                    constexpr int BL = decltype(bc)::value, B = 4 * h + BL;
                    const f32x4_t lo = fat_read<(2 * j) * 8 + B>(), hi = fat_read<(2 * j + 1) * 8 + B>();
                    float v[8] = {fmaf(alpha, lo[0], b0.x), fmaf(alpha, lo[1], b0.y), fmaf(alpha, lo[2], b0.z), fmaf(alpha, lo[3], b0.w),
                                  fmaf(alpha, hi[0], b1.x), fmaf(alpha, hi[1], b1.y), fmaf(alpha, hi[2], b1.z), fmaf(alpha, hi[3], b1.w)};
                    lds_u32x4_t* const cell_h = reinterpret_cast<lds_u32x4_t*>(wt + (xbase ^ (j * 64)) + BL * 2048);  // row 16 BL + frow of this pass
                    lds_u32x4_t* const cell_l = reinterpret_cast<lds_u32x4_t*>(wt + 8192 + (xbase ^ (j * 64)) + BL * 2048);
                    if (has_res) {
                        float r[8];
                        {
                            const u32x4_t rl = *cell_l;
                            const uint32_t wl4[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) unpack_lp16x2(wl4[e], r[2 * e], r[2 * e + 1]);
                        }
                        {
                            const u32x4_t rh = *cell_h;
                            const uint32_t wh4[4] = {rh.x, rh.y, rh.z, rh.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float a0, a1;
                                unpack_lp16x2(wh4[e], a0, a1);
                                v[2 * e] += fmaf(r[2 * e], 1.f / 2048.f, a0);       // hi + lo 2^-11: exact (22 bits)
                                v[2 * e + 1] += fmaf(r[2 * e + 1], 1.f / 2048.f, a1);
                            }
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
                    }
                    if constexpr (!POOL) {   // (the pooled plane form never writes the map: agrl_conv1x1_split16_pool passes no out)
                      if (has_out) {
                        uint32_t ph[4], pl[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ph[e] = pack_lp16x2(v[2 * e], v[2 * e + 1]);
                            float a0, a1;
                            unpack_lp16x2(ph[e], a0, a1);
                            pl[e] = pack_lp16x2((v[2 * e] - a0) * 2048.f, (v[2 * e + 1] - a1) * 2048.f);
                        }
                        *cell_h = u32x4_t{ph[0], ph[1], ph[2], ph[3]};
                        *cell_l = u32x4_t{pl[0], pl[1], pl[2], pl[3]};
                      }
                    }
                    if constexpr (POOL) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ps[BL >> 1][j][e] += v[e];
                    }
                });
            });
            if constexpr (!POOL)
            if (has_out) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u32x4_t vh = *reinterpret_cast<const lds_u32x4_t*>(wt + i * 1024 + lane * 16);
                    const u32x4_t vl = *reinterpret_cast<const lds_u32x4_t*>(wt + 8192 + i * 1024 + lane * 16);
                    if (m0 + 8 * (8 * h + i) + lrow < p.M) {
                        unsigned char* o = p.out + row_off3(8 * h + i);
                        *reinterpret_cast<u32x4_t*>(o) = vh;
                        *reinterpret_cast<u32x4_t*>(o + plane) = vl;
                        if (!p.ro_pair) *reinterpret_cast<u32x4_t*>(o + 2 * plane) = vh;
                    }
                }
            }
            if constexpr (POOL) {   // sum over the 16 pixel lanes of a fragment (same f), park the quarter sums (igemm_wide_kernel<16384>'s order)
                float* const s_w = s_pool_ + wave * 256;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float t = ps[q][j][e];
                            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                            ps[q][j][e] = t;
                        }
                        if (frow == 0) {
                            float* d = s_w + (2 * h + q) * 64 + 8 * fchunk + 32 * j;
                            *reinterpret_cast<float4*>(d) = make_float4(ps[q][j][0], ps[q][j][1], ps[q][j][2], ps[q][j][3]);
                            *reinterpret_cast<float4*>(d + 4) = make_float4(ps[q][j][4], ps[q][j][5], ps[q][j][6], ps[q][j][7]);
                        }
                    }
            }
            // the image is re-filled by the next pass's DMA: every LDS read of this pass must have returned first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
    } else {
    if (has_res) {
#pragma unroll
        for (int i = 0; i < 16; ++i) fat_dma(p.res + row_off(i), __builtin_amdgcn_readfirstlane(lds0 + wave * 16384 + i * 1024));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DUO_STAMP(5);
    sfor<2>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j + 4);
        sfor<8>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            const f32x4_t lo = fat_read<(2 * j) * 8 + B>(), hi = fat_read<(2 * j + 1) * 8 + B>();
            float v[8] = {fmaf(alpha, lo[0], b0.x), fmaf(alpha, lo[1], b0.y), fmaf(alpha, lo[2], b0.z), fmaf(alpha, lo[3], b0.w),
                          fmaf(alpha, hi[0], b1.x), fmaf(alpha, hi[1], b1.y), fmaf(alpha, hi[2], b1.z), fmaf(alpha, hi[3], b1.w)};   // (alpha = 1: acc + b bit for bit)
            lds_u32x4_t* const cell = reinterpret_cast<lds_u32x4_t*>(wt + (xbase ^ (j * 64)) + B * 2048);  // row 16 B + frow, chunk 4 j + f
            if (has_res) {
                const u32x4_t r = *cell;
                const uint32_t w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float l, h;
                    unpack_lp16x2(w4[e], l, h);
                    v[2 * e] += l;
                    v[2 * e + 1] += h;
                }
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
            }
            const u32x4_t pk = {pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7])};
            if (has_out) *cell = pk;
            if constexpr (POOL) {  // pool the rounded activations (what a separate pooling pass would read)
                const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float l, h;
                    unpack_lp16x2(w4[e], l, h);
                    psum[B >> 1][j][2 * e] += l;
                    psum[B >> 1][j][2 * e + 1] += h;
                }
            }
        });
    });
    DUO_STAMP(6);
    if (has_out) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const u32x4_t v = *reinterpret_cast<const lds_u32x4_t*>(wt + i * 1024 + lane * 16);
            if (m0 + 8 * i + lrow < p.M) *reinterpret_cast<u32x4_t*>(p.out + row_off(i)) = v;
        }
    }
    }  // (!p.planes)
    DUO_STAMP(7);
#if DUO_ABL & 64
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DUO_STAMP(8);
#endif
    if constexpr (POOL) {
        // vmgn.py:298-308. The tile is ONE frame: fragments 2 q, 2 q + 1 are quarter q (4 image rows = 32 pixels). Sum over the 16
        // pixel lanes of a fragment (same f), park the quarter sums in LDS (the head of the wave's own image: its row reads above
        // are issued, and the LDS serves a wave in order); every output bin is a sum of whole quarters.
        float* const s_w = reinterpret_cast<float*>(smem_ + wave * 16384);  // [4 quarters][64 channels of this wave]
        if constexpr (!PLANES) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = psum[q][j][e];
                    t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                    psum[q][j][e] = t;
                }
                if (frow == 0) {
                    float* d = s_w + q * 64 + 8 * fchunk + 32 * j;
                    *reinterpret_cast<float4*>(d) = make_float4(psum[q][j][0], psum[q][j][1], psum[q][j][2], psum[q][j][3]);
                    *reinterpret_cast<float4*>(d + 4) = make_float4(psum[q][j][4], psum[q][j][5], psum[q][j][6], psum[q][j][7]);
                }
            }
        }
        __syncthreads();
        const int P = p.pool_nparts;
        for (int o = tid; o < P * 256; o += 256) {
            const int c = o & 255, part = o >> 8;
            const int q0 = p.pool_q0[part], q1 = p.pool_q1[part];
            float t = 0.f;
            for (int q = q0; q < q1; ++q)
                t += PLANES ? s_pool_[(c >> 6) * 256 + q * 64 + (c & 63)] : reinterpret_cast<const float*>(smem_ + (c >> 6) * 16384)[q * 64 + (c & 63)];
            if (p.pool_mean) t *= 1.f / (float)((q1 - q0) * 32);
            const size_t oi = ((size_t)mt * P + part) * p.Cout + nt * 256 + c;
            p.pool_out[oi] = t;
            if (p.pool_out_lp) p.pool_out_lp[oi] = f32_to_lp16(t);
        }
    }
}

// ---- PERSISTENT form (round 6, 16-bit non-plane entry points): 2 workgroups per CU walk the tiles, and the software pipeline of the
// k-loop runs on ACROSS tiles -- during a tile's second-last slab the pieces that used to be dummies (the slab's own rows again) bring
// the NEXT tile's slab 0 into the buffer that has just been freed, and during its last slab the weight ring refills with the next
// tile's first eight fragments instead of re-reading slab 0. The epilogue works in the OTHER buffer only (the last slab's): 4 wave
// images of 8 KB = 64 rows x 128 B, two passes of 64 rows (the plane epilogue's scheme), so the prefetched slab survives it. What a
// tile of the one-shot form spends before its first barrier (5.3 of 23.8 us: first pixel rows and weight fragments arriving behind
// the co-resident workgroup's traffic, profiles/r05_duo_timeline.txt) is requested a whole slab + an epilogue ahead here. Results bit-identical
// (same k order, same epilogue arithmetic). Every counted wait of the one-shot form stays valid: a wait "at most N operations outstanding"
// only ever meets MORE already-retired operations in this order, never fewer.
template <bool POOL>
__global__ __launch_bounds__(256, 2) void conv1x1_duo_persist_kernel(const DuoParams p, int ntiles) {
    using SCHED = DuoSchedOf;
    using std::integral_constant;
    __shared__ __attribute__((aligned(16))) unsigned char smem_[2 * DSLAB];
    __shared__ __attribute__((aligned(16))) float s_pool_[POOL ? 4 * 256 : 4];
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int lrow = lane >> 3, lchk = lane & 7;
    const int nNt = p.Cout >> 8;
    const int nslab = p.K >> 7, nslab1 = p.K1 >> 7;

    // virtual block id -> tile, the one-shot form's XCD map over ALL tiles (gridDim.x is a multiple of 8: a workgroup's tiles keep its XCD)
    auto tile_of = [&](int vb, int& mt_, int& nt_) {
        const int q = ntiles >> 3, r = ntiles & 7;
        const int xcd = vb & 7, within = vb >> 3;
        const int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
        mt_ = b / nNt;
        nt_ = b - mt_ * nNt;
    };
    // byte offset of this lane's 16 bytes of pixel row (wave + 4 j) * 8 + lrow of the tile at m0_ in source 1 (second = false) / source 2
    // (by value, element by element: an array handed to a lambda by reference lives in scratch, and a scratch access is a vector-memory
    // operation that would sit in the counted vmcnt queue)
    auto roff_of = [&](int m0_, int j, bool second, int lrow_, int lchk_) -> unsigned {   // (lrow_ / lchk_: opaque copies at the per-tile call
        // sites, so that hipcc recomputes the row terms there instead of carrying them through the k-loop: the kernel sits at its 128 arch VGPRs)
        const int row = (wave + 4 * j) * 8 + lrow_;
        const int gm = min(m0_ + row, p.M - 1);
        const unsigned sw = (unsigned)((lchk_ ^ ((row >> 1) & 7)) << 4);
        if (second) return (unsigned)gm * (unsigned)(p.K - p.K1) * 2u + sw;
        unsigned srow = (unsigned)gm;
        if (p.gHoWo) {
            const unsigned f = (unsigned)gm / (unsigned)p.gHoWo, r = (unsigned)gm - f * (unsigned)p.gHoWo;
            const unsigned ho = r / (unsigned)p.gWo, wo = r - ho * (unsigned)p.gWo;
            srow = f * (unsigned)p.gHiWi + (ho * (unsigned)p.gWi + wo) * (unsigned)p.gS;
        }
        return srow * (unsigned)p.K1 * 2u + sw;
    };
    int vb = blockIdx.x, mt, nt;
    tile_of(vb, mt, nt);
    int m0 = mt * DROWS;
    unsigned roff[4], roff2[4], roffd[4];   // roffd: next tile's source-1 offsets minus this tile's
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        roff[j] = roff_of(m0, j, false, lrow, lchk);
        roff2[j] = roff_of(m0, j, true, lrow, lchk);
        roffd[j] = 0u;
    }
    auto stage_piece = [&](int slab, int buf, auto i_c, bool nx) {  // piece i of slab `slab` (nx: of the NEXT tile's slab 0) into buffer `buf`
        constexpr int I = decltype(i_c)::value, J = I >> 1, H = I & 1;
        const bool second = !nx && slab >= nslab1;
        // (next tile's offset as current + delta: a select between two array elements becomes a select of ADDRESSES in LLVM, which pins
        // both arrays in scratch -- and a scratch access is a vector-memory operation in the counted queue)
        const unsigned ro1 = roff[J] + (nx ? roffd[J] : 0u);
        const unsigned char* src = second ? p.x2 + roff2[J] + (size_t)((slab - nslab1) * 256 + H * 128)
                                          : p.x + ro1 + (size_t)(slab * 256 + H * 128);
        fat_dma(src, __builtin_amdgcn_readfirstlane(lds0 + buf * DSLAB + H * DHALF + (wave + 4 * J) * 1024));
    };
    const int xbase = frow * 128 + ((fchunk ^ ((frow >> 1) & 7)) << 4);
    auto stream_of = [&](int nt_) { return p.wpk + (size_t)(nt_ * 4 + wave) * nslab * (DPS * 1024); };
    const unsigned char* wstream = stream_of(nt);
    u32x4_t wr[DRING];
    auto issue_w = [&](auto slot_c, const unsigned char* slab_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        fat_gload<(POS & 3) * 1024>(wr[SLOT], lane16, slab_base + (POS & ~3) * 1024);
    };
    asm volatile("" ::: "a127");

    // ---- prologue of the FIRST tile (the one-shot form's)
    sfor<DPPW>([&](auto ic) { stage_piece(0, 0, ic, false); });
    sfor<DRING>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        issue_w(ic, wstream, ic);
        sfor<dpieces_at(I + DRING)>([&](auto jc) {
            stage_piece(1, 1, integral_constant<int, dpiece_first(I + DRING) + decltype(jc)::value>{}, false);
        });
    });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DRING + DPPW) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int par = 0;   // buffer of the current tile's slab 0
    auto ldx = [&](const lds_u8_t* sp, auto ks_c, auto b_c) {
        constexpr int KS = decltype(ks_c)::value, B = decltype(b_c)::value;
        const lds_u8_t* a = sp + (xbase ^ ((KS & 1) * 64));
        return *reinterpret_cast<const lds_u32x4_t*>(a + (KS >> 1) * DHALF + B * 2048);
    };
    const bool has_res = p.res != nullptr;
    const bool has_out = !POOL || p.out != nullptr;
    const float alpha = p.alpha;
    for (;;) {
        sfor<32>([&](auto qc) { fat_zero<decltype(qc)::value>(); });
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < ntiles;
        int mtn = mt, ntn = nt;
        const unsigned char* wstream_n = wstream;
        if (has_next) {
            tile_of(vbn, mtn, ntn);
            int lr = lrow, lc = lchk;
            asm volatile("" : "+v"(lr), "+v"(lc));
#pragma unroll
            for (int j = 0; j < 4; ++j) roffd[j] = roff_of(mtn * DROWS, j, false, lr, lc) - roff[j];
            wstream_n = stream_of(ntn);
        }
        u32x4_t xf[8];
        sfor<8>([&](auto bc) { xf[decltype(bc)::value] = ldx(smem + par * DSLAB, integral_constant<int, 0>{}, bc); });
        for (int slab = 0; slab < nslab; ++slab) {
            const bool more = slab + 1 < nslab;
            const unsigned char* ws = wstream + (size_t)slab * (DPS * 1024);
            const unsigned char* wsn = more ? wstream + (size_t)(slab + 1) * (DPS * 1024) : wstream_n;   // last slab: the NEXT tile's first ring
            const bool nx = has_next && slab + 2 == nslab;                   // second-last slab: the next tile's slab 0 ...
            const int ahead = nx ? 0 : (slab + 2 < nslab ? slab + 2 : slab);  // ... (last slab, last tile: dummies -- the slab's own rows again)
            const int cur = (slab + par) & 1;
            const lds_u8_t* sp = smem + cur * DSLAB;
            const lds_u8_t* spn = smem + (cur ^ 1) * DSLAB;
            sfor<DPS>([&](auto pc) {
                constexpr int P = decltype(pc)::value;
                constexpr int KS = P >> 2, A = P & 3, SL = P % DRING;
                fat_wait<SCHED::value.allowed[P]>(wr[SL]);
                if constexpr (P == DBARRIER_AT) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                sfor<8>([&](auto bc) {
                    constexpr int B = decltype(bc)::value;
                    fat_mfma<A * 8 + B>(wr[SL], xf[B]);
                    if constexpr (A == 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (KS + 1 < 4) xf[B] = ldx(sp, integral_constant<int, KS + 1>{}, bc);
                        else xf[B] = ldx(spn, integral_constant<int, 0>{}, bc);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
                constexpr int Q = P + DRING;
                if constexpr (Q >= DPS) issue_w(integral_constant<int, SL>{}, wsn, integral_constant<int, Q - DPS>{});
                else issue_w(integral_constant<int, SL>{}, ws, integral_constant<int, Q>{});
                sfor<dpieces_at(P)>([&](auto ic) { stage_piece(ahead, cur, integral_constant<int, dpiece_first(P) + decltype(ic)::value>{}, nx); });
            });
        }
        // the next tile's first ring and its slab 0 (or the dummies of the last tile) are still landing
#pragma unroll
        for (int i = 0; i < DRING; ++i) asm volatile("" : "+v"(wr[i]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < DRING; ++i) asm volatile("" : "+v"(wr[i]));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __syncthreads();  // every wave is past its last fragment read of the LAST slab's buffer: it becomes the image area

        // ---- epilogue in the last slab's buffer: wave image 8 KB = 64 rows x 128 B (the pixel buffers' swizzled row layout), two passes
        const int bufE = (nslab - 1 + par) & 1;
        lds_u8_t* const wt = smem + bufE * DSLAB + wave * 8192;
        const unsigned ldsE = lds0 + bufE * DSLAB + wave * 8192;
        const size_t colb = (size_t)(nt * 256 + wave * 64) * 2;
        const int cb = nt * 256 + wave * 64 + 8 * fchunk;
        int lrow_e = lrow, lchk_e = lchk;
        asm volatile("" : "+v"(lrow_e), "+v"(lchk_e));
        auto row_off = [&](int i) {
            const int row = 8 * i + lrow_e;
            const int gm = min(m0 + row, p.M - 1);
            return (unsigned)((size_t)gm * p.Cout * 2 + colb + (size_t)((lchk_e ^ ((row >> 1) & 7)) << 4));
        };
        sfor<2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            float ps[POOL ? 2 : 1][2][8];
            if constexpr (POOL) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) ps[q][j][e] = 0.f;
            }
            if (has_res) {
#pragma unroll
                for (int i = 0; i < 8; ++i) fat_dma_s(p.res, row_off(8 * h + i), __builtin_amdgcn_readfirstlane(ldsE + i * 1024));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sfor<2>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j);
                const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j + 4);
                sfor<4>([&](auto bc) {
                    constexpr int BL = decltype(bc)::value, B = 4 * h + BL;
                    const f32x4_t lo = fat_read<(2 * j) * 8 + B>(), hi = fat_read<(2 * j + 1) * 8 + B>();
                    float v[8] = {fmaf(alpha, lo[0], b0.x), fmaf(alpha, lo[1], b0.y), fmaf(alpha, lo[2], b0.z), fmaf(alpha, lo[3], b0.w),
                                  fmaf(alpha, hi[0], b1.x), fmaf(alpha, hi[1], b1.y), fmaf(alpha, hi[2], b1.z), fmaf(alpha, hi[3], b1.w)};
                    lds_u32x4_t* const cell = reinterpret_cast<lds_u32x4_t*>(wt + (xbase ^ (j * 64)) + BL * 2048);
                    if (has_res) {
                        const u32x4_t r = *cell;
                        const uint32_t w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float l, hh;
                            unpack_lp16x2(w4[e], l, hh);
                            v[2 * e] += l;
                            v[2 * e + 1] += hh;
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
                    }
                    const u32x4_t pk = {pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7])};
                    if (has_out) *cell = pk;
                    if constexpr (POOL) {  // pool the rounded activations (what a separate pooling pass would read)
                        const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float l, hh;
                            unpack_lp16x2(w4[e], l, hh);
                            ps[BL >> 1][j][2 * e] += l;
                            ps[BL >> 1][j][2 * e + 1] += hh;
                        }
                    }
                });
            });
            if (has_out) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u32x4_t v = *reinterpret_cast<const lds_u32x4_t*>(wt + i * 1024 + lane * 16);
                    if (m0 + 8 * (8 * h + i) + lrow < p.M) *reinterpret_cast<u32x4_t*>(p.out + row_off(8 * h + i)) = v;
                }
            }
            if constexpr (POOL) {
                float* const s_w = s_pool_ + wave * 256;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float t = ps[q][j][e];
                            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                            ps[q][j][e] = t;
                        }
                        if (frow == 0) {
                            float* d = s_w + (2 * h + q) * 64 + 8 * fchunk + 32 * j;
                            *reinterpret_cast<float4*>(d) = make_float4(ps[q][j][0], ps[q][j][1], ps[q][j][2], ps[q][j][3]);
                            *reinterpret_cast<float4*>(d + 4) = make_float4(ps[q][j][4], ps[q][j][5], ps[q][j][6], ps[q][j][7]);
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is re-filled by the next pass's DMA
        });
        if constexpr (POOL) {
            __syncthreads();
            const int P = p.pool_nparts;
            for (int o = tid; o < P * 256; o += 256) {
                const int c = o & 255, part = o >> 8;
                const int q0 = p.pool_q0[part], q1 = p.pool_q1[part];
                float t = 0.f;
                for (int q = q0; q < q1; ++q) t += s_pool_[(c >> 6) * 256 + q * 64 + (c & 63)];
                if (p.pool_mean) t *= 1.f / (float)((q1 - q0) * 32);
                const size_t oi = ((size_t)mt * P + part) * p.Cout + nt * 256 + c;
                p.pool_out[oi] = t;
                if (p.pool_out_lp) p.pool_out_lp[oi] = f32_to_lp16(t);
            }
        }
        if (!has_next) break;
        // ---- on to the next tile: its slab 0 sits in the other buffer, its first ring in wr[]; slab 1 goes where the images were
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();   // every wave has read its image rows (and the pooled sums): the buffer is free
        vb = vbn; mt = mtn; nt = ntn; m0 = mt * DROWS;
        int lr2 = lrow, lc2 = lchk;
        asm volatile("" : "+v"(lr2), "+v"(lc2));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            roff[j] += roffd[j];
            roff2[j] = roff_of(m0, j, true, lr2, lc2);
        }
        wstream = wstream_n;
        par = (nslab + par) & 1;
        sfor<DPPW>([&](auto ic) { stage_piece(nslab > 1 ? 1 : 0, par ^ 1, ic, false); });
    }
}

int duo_launch(DuoParams& p, bool pool, hipStream_t stream, const char* who) {
    if (!p.planes) p.alpha = 1.f;   // the 16-bit entry points: acc + bias, bit for bit what round 5 computed
    const int grid = ((p.M + DROWS - 1) / DROWS) * (p.Cout >> 8);
    constexpr int dyn = (DUO_ABL & 32) ? 48 * 1024 : 0;
    if (p.planes) {
        if (pool) hipLaunchKernelGGL((conv1x1_duo_kernel<true, true>), dim3(grid), dim3(256), dyn, stream, p);
        else hipLaunchKernelGGL((conv1x1_duo_kernel<false, true>), dim3(grid), dim3(256), dyn, stream, p);
    } else if (agrl_opts().duo_persist != 0 && (p.K >> 7) >= 2 && grid > 512) {
        // round 6: two persistent workgroups per CU, the next tile's first slab and weight ring requested during the current tile's last
        // slabs (conv1x1_duo_persist_kernel; bit-identical). Same-box A/B of the whole step, four pairs: 3.440-3.454 against 3.457-3.461 ms
        // (every run ahead of every one-shot run; per launch: conv3 + residual 98 -> 92 / 95 -> 90.5 us, the strided first blocks 83 -> 78 /
        // 58 -> 54, pooled 98 -> 95; profiles/r06_ab_duo_persist.txt). AGRL_DUO_PERSIST=0: the one-shot form. Launches of <= 512 tiles
        // (layer 4's conv1s: one tile per resident slot) have nothing to carry over and stay one-shot.
        if (pool) hipLaunchKernelGGL((conv1x1_duo_persist_kernel<true>), dim3(512), dim3(256), 0, stream, p, grid);
        else hipLaunchKernelGGL((conv1x1_duo_persist_kernel<false>), dim3(512), dim3(256), 0, stream, p, grid);
    } else if (pool) hipLaunchKernelGGL((conv1x1_duo_kernel<true, false>), dim3(grid), dim3(256), dyn, stream, p);
    else hipLaunchKernelGGL((conv1x1_duo_kernel<false, false>), dim3(grid), dim3(256), dyn, stream, p);
    AGRL_CHECK_LAUNCH(who);
    return 0;
}

}  // namespace

#if DUO_ABL & 64
extern "C" int agrl_duo_trace_buffer(void* buf) {  // profiling build only
    return hipMemcpyToSymbol(HIP_SYMBOL(g_duo_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int agrl_conv1x1_packed_res_bn_act(const void* x, const void* packed, const float* bias, const void* residual, void* out,
                                              int M, int K, int Cout, int relu, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && packed && bias && out, "agrl_conv1x1_packed_res_bn_act: null pointer");
    AGRL_CHECK_ARG(M > 0 && K > 0 && K % 128 == 0 && Cout > 0 && Cout % 256 == 0,
                   "agrl_conv1x1_packed_res_bn_act: needs K %% 128 == 0 and Cout %% 256 == 0; got M=%d K=%d Cout=%d", M, K, Cout);
    AGRL_CHECK_ARG((size_t)M * (size_t)(K > Cout ? K : Cout) * 2 < (1ull << 32), "agrl_conv1x1_packed_res_bn_act: maps beyond 4 GB are not addressed");
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)out) & 15) == 0,
                   "agrl_conv1x1_packed_res_bn_act: pointers must be 16-byte aligned");
    DuoParams p{};
    p.x = reinterpret_cast<const unsigned char*>(x);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.bias = bias;
    p.res = reinterpret_cast<const unsigned char*>(residual);
    p.out = reinterpret_cast<unsigned char*>(out);
    p.M = M; p.K = K; p.K1 = K; p.Cout = Cout; p.relu = relu;
    return duo_launch(p, false, (hipStream_t)stream, "agrl_conv1x1_packed_res_bn_act");
}

extern "C" int agrl_conv1x1_packed_res_pool(const void* x, const void* packed, const float* bias, const void* residual,
                                            float* pool_out, void* pool_out_lp, int N, int H, int W, int K, int Cout, int relu,
                                            const int* splits, int n_splits, int mean, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && packed && bias && pool_out && splits, "agrl_conv1x1_packed_res_pool: null pointer");
    AGRL_CHECK_ARG(H == 16 && W == 8, "agrl_conv1x1_packed_res_pool: a frame must be 16 x 8 pixels (got %dx%d)", H, W);
    AGRL_CHECK_ARG(N > 0 && K > 0 && K % 128 == 0 && Cout > 0 && Cout % 256 == 0,
                   "agrl_conv1x1_packed_res_pool: needs K %% 128 == 0 and Cout %% 256 == 0; got N=%d K=%d Cout=%d", N, K, Cout);
    AGRL_CHECK_ARG((size_t)N * 128 * (size_t)(K > Cout ? K : Cout) * 2 < (1ull << 32), "agrl_conv1x1_packed_res_pool: maps beyond 4 GB are not addressed");
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)pool_out) & 15) == 0,
                   "agrl_conv1x1_packed_res_pool: pointers must be 16-byte aligned");
    DuoParams p{};
    int P = 0;
    for (int i = 0; i < n_splits; ++i) {
        const int n = splits[i];
        AGRL_CHECK_ARG(n > 0 && P + n <= 16, "agrl_conv1x1_packed_res_pool: at most 16 bins");
        for (int j = 0; j < n; ++j) {  // AdaptiveAvgPool2d bins over image rows; here: whole quarters of the frame
            const int r0 = (j * H) / n, r1 = ((j + 1) * H + n - 1) / n;
            AGRL_CHECK_ARG((r0 & 3) == 0 && (r1 & 3) == 0, "agrl_conv1x1_packed_res_pool: bins must be made of whole 4-row quarters (split %d)", n);
            p.pool_q0[P] = r0 >> 2;
            p.pool_q1[P] = r1 >> 2;
            ++P;
        }
    }
    p.pool_nparts = P; p.pool_mean = mean;
    p.pool_out = pool_out;
    p.pool_out_lp = reinterpret_cast<unsigned short*>(pool_out_lp);
    p.x = reinterpret_cast<const unsigned char*>(x);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.bias = bias;
    p.res = reinterpret_cast<const unsigned char*>(residual);
    p.out = nullptr;
    p.M = N * 128; p.K = K; p.K1 = K; p.Cout = Cout; p.relu = relu;
    return duo_launch(p, true, (hipStream_t)stream, "agrl_conv1x1_packed_res_pool");
}

extern "C" int agrl_conv1x1_packed_dual_duo(const void* x, const void* x2, const void* packed, const float* bias, void* out, int M, int K1,
                                            int K2, int Cout, int relu, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && x2 && packed && bias && out, "agrl_conv1x1_packed_dual_duo: null pointer");
    AGRL_CHECK_ARG(M > 0 && K1 > 0 && K1 % 128 == 0 && K2 > 0 && K2 % 128 == 0 && Cout > 0 && Cout % 256 == 0,
                   "agrl_conv1x1_packed_dual_duo: needs K1, K2 %% 128 == 0 and Cout %% 256 == 0; got M=%d K1=%d K2=%d Cout=%d", M, K1, K2, Cout);
    const size_t widest = (size_t)(K1 > Cout ? (K1 > K2 ? K1 : K2) : (Cout > K2 ? Cout : K2));
    AGRL_CHECK_ARG((size_t)M * widest * 2 < (1ull << 32), "agrl_conv1x1_packed_dual_duo: maps beyond 4 GB are not addressed");
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)out) & 15) == 0,
                   "agrl_conv1x1_packed_dual_duo: pointers must be 16-byte aligned");
    DuoParams p{};
    p.x = reinterpret_cast<const unsigned char*>(x);
    p.x2 = reinterpret_cast<const unsigned char*>(x2);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.bias = bias;
    p.out = reinterpret_cast<unsigned char*>(out);
    p.M = M; p.K = K1 + K2; p.K1 = K1; p.Cout = Cout; p.relu = relu;
    return duo_launch(p, false, (hipStream_t)stream, "agrl_conv1x1_packed_dual_duo");
}

// The same GEMM for the first block of a STRIDED layer (layers 2 / 3: conv2 and the downsample conv both have stride s, vmgn.py:56-64 with
// the downsample of torchvision's _make_layer): x is the block input (N, Hi, Wi, K1), of which the 1x1 / stride-s / pad-0 downsample conv
// reads pixel (s ho, s wo) for output pixel (ho, wo); x2 = conv2's output (N, Ho, Wo, K2), Ho = (Hi - 1) / s + 1. The shortcut map
// (N, Ho, Wo, Cout) is neither written nor read back.
extern "C" int agrl_conv1x1_packed_dual_strided(const void* x, const void* x2, const void* packed, const float* bias, void* out, int N,
                                                int Hi, int Wi, int stride, int K1, int K2, int Cout, int relu, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && x2 && packed && bias && out, "agrl_conv1x1_packed_dual_strided: null pointer");
    AGRL_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && stride >= 1, "agrl_conv1x1_packed_dual_strided: bad map %dx%dx%d stride %d", N, Hi, Wi, stride);
    AGRL_CHECK_ARG(K1 > 0 && K1 % 128 == 0 && K2 > 0 && K2 % 128 == 0 && Cout > 0 && Cout % 256 == 0,
                   "agrl_conv1x1_packed_dual_strided: needs K1, K2 %% 128 == 0 and Cout %% 256 == 0; got K1=%d K2=%d Cout=%d", K1, K2, Cout);
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    const size_t M = (size_t)N * Ho * Wo;
    AGRL_CHECK_ARG((size_t)N * Hi * Wi * K1 * 2 < (1ull << 32) && M * (size_t)(Cout > K2 ? Cout : K2) * 2 < (1ull << 32),
                   "agrl_conv1x1_packed_dual_strided: maps beyond 4 GB are not addressed");
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)out) & 15) == 0,
                   "agrl_conv1x1_packed_dual_strided: pointers must be 16-byte aligned");
    DuoParams p{};
    p.x = reinterpret_cast<const unsigned char*>(x);
    p.x2 = reinterpret_cast<const unsigned char*>(x2);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.bias = bias;
    p.out = reinterpret_cast<unsigned char*>(out);
    p.M = (int)M; p.K = K1 + K2; p.K1 = K1; p.Cout = Cout; p.relu = relu;
    if (stride > 1) { p.gHoWo = Ho * Wo; p.gWo = Wo; p.gS = stride; p.gWi = Wi; p.gHiWi = Hi * Wi; }
    return duo_launch(p, false, (hipStream_t)stream, "agrl_conv1x1_packed_dual_strided");
}

// ---- split-fp16 planes (round 6: the conforming mode at speed; torchreid/models/vmgn.py:45-65 in fp32-equivalent arithmetic) ----------
// Every activation tensor is (M, 3 C) fp16 per pixel [hi | lo 2^11 | hi], hi = fp16(v), lo = fp16((v - hi) 2^11); `packed` is
// agrl_conv1x1_pack of the fp16 weight (Cout, 3 K) = [wh | wh 2^-11 | wl] of w 2^k (hip_ops.split16_plane_weights): the unchanged k-loop
// then sums xh wh + xl wh + xh wl -- 22 significand bits per operand -- into one fp32 accumulator, which the epilogue un-scales by
// w_unscale = 2^-k. K3 / K1_3 / K2_3 count the plane channels (3 x the true ones).
static int duo_split16_common(DuoParams& p, const void* x, const void* packed, const float* bias, int M, int K3, int Cout, int relu,
                              float w_unscale, int layout, const char* who) {
    AGRL_CHECK_ARG(layout >= 0 && layout <= 3, "%s: layout = %d (bit 0: x is a plane pair, bit 1: residual / out are pairs)", who, layout);
    p.x_pair = layout & 1; p.ro_pair = (layout >> 1) & 1;
    AGRL_CHECK_ARG(agrl_lp16_is_f16(), "%s: the split planes are fp16 (load libagrl_hip.so, not the bf16 build)", who);
    AGRL_CHECK_ARG(x && packed && bias, "%s: null pointer", who);
    AGRL_CHECK_ARG(M > 0 && K3 > 0 && K3 % 384 == 0 && Cout > 0 && Cout % 256 == 0, "%s: needs K3 %% 384 == 0 (3 planes of whole 128-channel slabs) and Cout %% 256 == 0; got M=%d K3=%d Cout=%d", who, M, K3, Cout);
    AGRL_CHECK_ARG((size_t)M * (size_t)(K3 > 3 * Cout ? K3 : 3 * Cout) * 2 < (1ull << 32), "%s: maps beyond 4 GB are not addressed", who);
    AGRL_CHECK_ARG(w_unscale > 0.f && w_unscale <= 3.4e38f, "%s: w_unscale must be a positive finite power of two", who);
    {
        int e = 0;
        AGRL_CHECK_ARG(frexpf(w_unscale, &e) == 0.5f, "%s: w_unscale=%g is not a power of two", who, (double)w_unscale);
    }
    p.x = reinterpret_cast<const unsigned char*>(x);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.bias = bias;
    p.M = M; p.K = K3; p.K1 = K3; p.Cout = Cout; p.relu = relu;
    p.alpha = w_unscale; p.planes = 1;
    return 0;
}

extern "C" int agrl_conv1x1_split16(const void* x, const void* packed, const float* bias, const void* residual, void* out, int M, int K3,
                                    int Cout, int relu, float w_unscale, int layout, agrl_stream_t stream) {
    DuoParams p{};
    if (int rc = duo_split16_common(p, x, packed, bias, M, K3, Cout, relu, w_unscale, layout, "agrl_conv1x1_split16")) return rc;
    AGRL_CHECK_ARG(out, "agrl_conv1x1_split16: null pointer");
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)out) & 15) == 0, "agrl_conv1x1_split16: pointers must be 16-byte aligned");
    p.res = reinterpret_cast<const unsigned char*>(residual);
    p.out = reinterpret_cast<unsigned char*>(out);
    return duo_launch(p, false, (hipStream_t)stream, "agrl_conv1x1_split16");
}

extern "C" int agrl_conv1x1_split16_dual(const void* x, const void* x2, const void* packed, const float* bias, void* out, int M, int K1_3,
                                         int K2_3, int Cout, int relu, float w_unscale, int layout, agrl_stream_t stream) {
    DuoParams p{};
    AGRL_CHECK_ARG(x2 && out && K1_3 > 0 && K1_3 % 384 == 0 && K2_3 > 0 && K2_3 % 384 == 0, "agrl_conv1x1_split16_dual: needs two sources of whole plane triples (K %% 384 == 0)");
    if (int rc = duo_split16_common(p, x, packed, bias, M, K1_3 + K2_3, Cout, relu, w_unscale, layout, "agrl_conv1x1_split16_dual")) return rc;
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)out) & 15) == 0, "agrl_conv1x1_split16_dual: pointers must be 16-byte aligned");
    p.x2 = reinterpret_cast<const unsigned char*>(x2);
    p.K1 = K1_3;
    p.out = reinterpret_cast<unsigned char*>(out);
    return duo_launch(p, false, (hipStream_t)stream, "agrl_conv1x1_split16_dual");
}

extern "C" int agrl_conv1x1_split16_pool(const void* x, const void* packed, const float* bias, const void* residual, float* pool_out, int N,
                                         int H, int W, int K3, int Cout, int relu, const int* splits, int n_splits, int mean, float w_unscale,
                                         int layout, agrl_stream_t stream) {
    DuoParams p{};
    AGRL_CHECK_ARG(pool_out && splits, "agrl_conv1x1_split16_pool: null pointer");
    AGRL_CHECK_ARG(H == 16 && W == 8 && N > 0, "agrl_conv1x1_split16_pool: a frame must be 16 x 8 pixels (got %dx%d)", H, W);
    if (int rc = duo_split16_common(p, x, packed, bias, N * 128, K3, Cout, relu, w_unscale, layout, "agrl_conv1x1_split16_pool")) return rc;
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)pool_out) & 15) == 0, "agrl_conv1x1_split16_pool: pointers must be 16-byte aligned");
    int P = 0;
    for (int i = 0; i < n_splits; ++i) {
        const int n = splits[i];
        AGRL_CHECK_ARG(n > 0 && P + n <= 16, "agrl_conv1x1_split16_pool: at most 16 bins");
        for (int j = 0; j < n; ++j) {
            const int r0 = (j * H) / n, r1 = ((j + 1) * H + n - 1) / n;
            AGRL_CHECK_ARG((r0 & 3) == 0 && (r1 & 3) == 0, "agrl_conv1x1_split16_pool: bins must be made of whole 4-row quarters (split %d)", n);
            p.pool_q0[P] = r0 >> 2;
            p.pool_q1[P] = r1 >> 2;
            ++P;
        }
    }
    p.pool_nparts = P; p.pool_mean = mean;
    p.pool_out = pool_out;
    p.pool_out_lp = nullptr;
    p.res = reinterpret_cast<const unsigned char*>(residual);
    p.out = nullptr;
    return duo_launch(p, true, (hipStream_t)stream, "agrl_conv1x1_split16_pool");
}

// fp32 (rows, C) -> the three fp16 planes (rows, 3 C) = [hi | lo 2^11 | hi] the split16 kernels consume: the seam between the fp32-tensor
// part of the conforming mode (stem .. first block of layer 3, agrl_conv2d_bn_act_split16) and its plane part.
__global__ __launch_bounds__(256) void split16_planes_kernel(const float4* __restrict__ x, uint2* __restrict__ out, long long groups, int c4, int nplanes) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < groups; t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / c4;
        const int g = (int)(t - row * c4);
        const float4 v = x[t];
        const uint32_t h0 = pack_lp16x2(v.x, v.y), h1 = pack_lp16x2(v.z, v.w);
        float a, b, c, d;
        unpack_lp16x2(h0, a, b);
        unpack_lp16x2(h1, c, d);
        const uint2 hi = make_uint2(h0, h1);
        const uint2 lo = make_uint2(pack_lp16x2((v.x - a) * 2048.f, (v.y - b) * 2048.f), pack_lp16x2((v.z - c) * 2048.f, (v.w - d) * 2048.f));
        uint2* o = out + row * nplanes * c4 + g;
        o[0] = hi;
        o[c4] = lo;
        if (nplanes == 3) o[2 * c4] = hi;
    }
}

extern "C" int agrl_split16_planes(const float* x, void* out, long long rows, int C, int nplanes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(nplanes == 2 || nplanes == 3, "agrl_split16_planes: nplanes = %d (3: [hi | lo 2^11 | hi], 2: the pair [hi | lo 2^11])", nplanes);
    AGRL_CHECK_ARG(agrl_lp16_is_f16(), "agrl_split16_planes: the split planes are fp16 (load libagrl_hip.so, not the bf16 build)");
    AGRL_CHECK_ARG(x && out && rows > 0 && C > 0 && C % 4 == 0, "agrl_split16_planes: needs C %% 4 == 0 (got rows=%lld C=%d)", rows, C);
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "agrl_split16_planes: pointers must be 16-byte aligned");
    const long long groups = rows * (C / 4);
    const int grid = (int)((groups + 255) / 256 < 8192 ? (groups + 255) / 256 : 8192);
    hipLaunchKernelGGL(split16_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<uint2*>(out), groups, C / 4, nplanes);
    AGRL_CHECK_LAUNCH("agrl_split16_planes");
    return 0;
}

// fp32 (rows, C) -> the WEIGHT-side plane triple (rows, 3 C) = [h | h 2^-11 | l] of x * scale (scale a power of two: exact), h = fp16(x scale),
// l = fp16(x scale - h): the operand that meets activation planes [xh | xl 2^11 | xh] in a k-loop (gallery rows of agrl_distmat_split16; the
// conv weights are packed by the host at pack time, hip_ops.split16_plane_weights: the same arithmetic).
__global__ __launch_bounds__(256) void split16_weight_planes_kernel(const float4* __restrict__ x, uint2* __restrict__ out, long long groups, int c4, float scale) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < groups; t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / c4;
        const int g = (int)(t - row * c4);
        float4 v = x[t];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        const uint32_t h0 = pack_lp16x2(v.x, v.y), h1 = pack_lp16x2(v.z, v.w);
        float a, b, c, d;
        unpack_lp16x2(h0, a, b);
        unpack_lp16x2(h1, c, d);
        uint2* o = out + row * 3 * c4 + g;
        o[0] = make_uint2(h0, h1);
        o[c4] = make_uint2(pack_lp16x2(a * (1.f / 2048.f), b * (1.f / 2048.f)), pack_lp16x2(c * (1.f / 2048.f), d * (1.f / 2048.f)));
        o[2 * c4] = make_uint2(pack_lp16x2(v.x - a, v.y - b), pack_lp16x2(v.z - c, v.w - d));
    }
}

extern "C" int agrl_split16_weight_planes(const float* x, void* out, long long rows, int C, float scale, agrl_stream_t stream) {
    AGRL_CHECK_ARG(agrl_lp16_is_f16(), "agrl_split16_weight_planes: the split planes are fp16 (load libagrl_hip.so, not the bf16 build)");
    AGRL_CHECK_ARG(x && out && rows > 0 && C > 0 && C % 4 == 0, "agrl_split16_weight_planes: needs C %% 4 == 0 (got rows=%lld C=%d)", rows, C);
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "agrl_split16_weight_planes: pointers must be 16-byte aligned");
    {
        int e = 0;
        AGRL_CHECK_ARG(scale > 0.f && scale <= 3.4e38f && frexpf(scale, &e) == 0.5f, "agrl_split16_weight_planes: scale=%g is not a positive power of two", (double)scale);
    }
    const long long groups = rows * (C / 4);
    const int grid = (int)((groups + 255) / 256 < 8192 ? (groups + 255) / 256 : 8192);
    hipLaunchKernelGGL(split16_weight_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<uint2*>(out), groups, C / 4, scale);
    AGRL_CHECK_LAUNCH("agrl_split16_weight_planes");
    return 0;
}
