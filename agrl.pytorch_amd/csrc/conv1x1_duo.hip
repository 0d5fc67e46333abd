// conv3 / bn3 + identity shortcut + ReLU of a layer-4 Bottleneck (torchreid/models/vmgn.py:56-64, 512 -> 2048 on 16 x 8 maps), 16-bit
// build type, as a workgroup of TWO KINDS of waves -- and, as a second instantiation, the same conv with the frame pooling of
// vmgn.py:298-308 in its epilogue (agrl_conv1x1_bn_act_pool's job: the 2048-channel map of a branch's last block never exists).
//   out (M, Cout) = act(x (M, K) @ W (Cout, K)^T + bias + residual (M, Cout))
// Why its own kernel. This layer is 68.7 GFLOP against 302 MB (33 MB of rows, 134 MB of residual, 134 MB of result): ~45 us of
// matrix work and ~55 us of HBM time. Every earlier form (igemm_wide_kernel<0, 256, true>, conv1x1_fat.hip with the residual as
// slabs / in registers / persistent, the back-to-back seam kernel) ran at their SUM, 105-121 us: a wave's loads, LDS-DMA and
// stores retire in order on ONE counter, so the waves that wait for an L2-latency operand ring cannot also keep an HBM-latency
// residual / store stream in flight. (Round 5, first attempt, measured: two independent four-wave workgroups per CU, one in its
// epilogue while the other multiplies -- 117 us against 119.5: per workgroup 5 us of first-load latency, 9 us of k-loop, 2.6 us
// waiting for the residual, 4.5 us combining, 2 us issuing stores, nothing of it ahead of time.)
// Here ONE persistent workgroup of 8 waves owns a 128-pixel tile (one 16 x 8 frame) and walks its 256-channel tiles:
//   * waves 0-3, one per SIMD ("matrix waves"): conv1x1_fat.hip's loop at half height -- wave w owns 64 channels (4 MFMA A
//     fragments) of all 128 pixels (8 B fragments), 32 accumulator quads = 128 asm-owned AGPRs; weights from agrl_conv1x1_pack's
//     per-(channel tile, wave) fragment streams (the SAME packed tensor serves both kernels) global -> 8-fragment VGPR ring;
//     pixel rows of the current / next 128-channel slab in LDS (2 x 32 KB) by LDS-DMA, one barrier per slab. The slab / fragment
//     stream runs on ACROSS channel tiles (the next tile's first weights and rows are in flight while this tile finishes: no
//     start-up latency per tile), and these waves never issue a load or store that goes to HBM for the residual or the result;
//   * waves 4-7, the second wave of each SIMD ("memory waves"), own the residual / result traffic on their own counters: wave
//     4 + w DMAs the residual tile of matrix wave w (128 rows x 128 B) into a wave-private 16 KB LDS image while the k-loop runs,
//     and after the combine step streams the image's rows out -- every global access a whole 128-byte line per eight lanes;
//   * per channel tile: k-loop, barrier, combine (+ bias, + residual from the image, ReLU, one rounding, result written in place:
//     conflict-free, the pixel fragments' addressing), barrier -- the order of operations of igemm_wide_kernel's register
//     epilogue, so the results are BIT-IDENTICAL to agrl_conv2d_bn_act(residual=...) / agrl_conv1x1_bn_act_pool;
//   * POOL: a tile is one frame; the rounded activations' quarter sums (4 image rows) per lane, 16-lane shuffles, quarters parked in
//     the image, bins = sums of whole quarters (formed and written by the memory waves) -- igemm_wide_kernel<16384>'s order.
// Measured numbers: DESIGN.md section 5, round 5.
#include "fat_dev.h"

namespace {

struct DuoParams {
    const unsigned char* x;     // (M, K) 16-bit pixel rows
    const unsigned char* wpk;   // packed weight streams (agrl_conv1x1_pack)
    const float* bias;          // (Cout)
    const unsigned char* res;   // (M, Cout) residual or nullptr
    unsigned char* out;         // (M, Cout) or nullptr (POOL only)
    int M, K, Cout, relu;
    int nsplit;                 // workgroups per pixel tile: each walks Cout / 256 / nsplit channel tiles
    // POOL
    float* pool_out;            // fp32 (frames, nparts, Cout)
    unsigned short* pool_out_lp;  // optional 16-bit copy
    int pool_nparts, pool_mean;
    int pool_q0[16], pool_q1[16];  // bins in quarters (4 image rows) [q0, q1)
};

#ifndef DUO_ABL
#define DUO_ABL 0  // timing ablations (results wrong): 1 no residual loads, 2 no stores, 4 no MFMA, 8 no weight loads in the loop, 16 no pixel DMA in the loop, 64 phase stamps (agrl_duo_trace_buffer)
#endif
constexpr int DRING = 8;                 // weight fragments in flight per wave
constexpr int DPS = 4 * 4;               // weight fragments per 128-channel slab and wave: 4 k-steps x 4 channel fragments
constexpr int DROWS = 128;               // pixel rows per tile
constexpr int DHALF = DROWS * 128;       // 128 pixel rows x 64 channels
constexpr int DSLAB = 2 * DHALF;         // one 128-channel slab of the pixel tile: 32 KB
constexpr int DIMG = 16384;              // a wave's residual / result image: 128 rows x 128 B
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

#if DUO_ABL & 64   // profiling build: phase stamps (s_memtime), 8 x 8 bytes per (workgroup, channel tile, role)
__device__ unsigned long long* g_duo_trace = nullptr;
#define DUO_STAMP(role, t, k)                                                                                                   \
    do {                                                                                                                        \
        if (g_duo_trace && lane == 0 && (wave & 3) == 0) g_duo_trace[(((size_t)blockIdx.x * 8 + (t)) * 2 + (role)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define DUO_STAMP(role, t, k) do { } while (0)
#endif

__device__ __forceinline__ void duo_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// The pixel buffers' barrier, among the FOUR MATRIX WAVES only: an arrival counter in LDS (one lane adds 1, everybody polls until
// 4 (g + 1) arrivals are in). s_barrier would make the memory waves take part -- and every cycle they spend queueing a store or a
// DMA behind a busy memory pipe would then stall the matrix waves at the next slab (measured: k-loops of 16 us instead of 4).
__device__ __forceinline__ void duo_slab_sync(unsigned cnt_addr, unsigned target) {
    unsigned v, sv;
    unsigned long long keep;
    const unsigned one = 1u;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"          // this wave's reads of the buffer that is about to be refilled are done
        "s_mov_b64 %[keep], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "ds_add_u32 %[addr], %[one]\n\t"
        "s_mov_b64 exec, %[keep]\n"
        "1:\n\t"
        "ds_read_b32 %[v], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[sv], %[v]\n\t"
        "s_cmp_lt_u32 %[sv], %[target]\n\t"
        "s_cbranch_scc1 1b"
        : [v] "=&v"(v), [sv] "=&s"(sv), [keep] "=&s"(keep)
        : [addr] "v"(cnt_addr), [one] "v"(one), [target] "s"(target)
        : "memory", "scc");
}

template <bool POOL>
__global__ __launch_bounds__(512) void conv1x1_duo_kernel(const DuoParams p) {
    using SCHED = DuoSchedOf;
    using std::integral_constant;
    __shared__ __attribute__((aligned(16))) unsigned char smem_[2 * DSLAB + 4 * DIMG + 4 * 1024 + 16];  // pixel buffers, images, quarter sums (POOL), the matrix waves' slab counter
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int lrow = lane >> 3, lchk = lane & 7;

    // workgroup = (pixel tile mt, part): channel tiles nt0 .. nt0 + ntile - 1
    const int nNt = p.Cout >> 8, ntile = nNt / p.nsplit;
    const int mt = blockIdx.x / p.nsplit, nt0 = (blockIdx.x - mt * p.nsplit) * ntile;
    const int m0 = mt * DROWS;
    const int nslab = p.K >> 7;
    const int w4 = wave & 3;  // the 64-channel column of the tile this wave computes (matrix wave) / moves (memory wave)
    const bool has_res = p.res != nullptr && !(DUO_ABL & 1);
    const bool has_out = (!POOL || p.out != nullptr) && !(DUO_ABL & 2);
    // the image of column w4: 128 rows x 128 B in the pixel buffers' layout (16-byte chunk c of row r at c ^ ((r >> 1) & 7))
    lds_u8_t* const img = smem + 2 * DSLAB + w4 * DIMG;
    const unsigned img0 = lds0 + 2 * DSLAB + w4 * DIMG;

    if (wave >= 4) {
        // ================= memory wave: residual in, result out, pooled bins; one barrier per barrier of the matrix waves
        auto row_off = [&](int i, int nt) {  // piece i = rows 8 i + lrow: byte offset of this lane's 16 bytes (chunk lchk ^ swizzle(row)) in res / out
            const int row = 8 * i + lrow;
            const int gm = min(m0 + row, p.M - 1);
            return (size_t)gm * p.Cout * 2 + (size_t)(nt * 256 + w4 * 64) * 2 + (size_t)((lchk ^ ((row >> 1) & 7)) << 4);
        };
        auto fetch_residual = [&](int nt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) fat_dma(p.res + row_off(i, nt), __builtin_amdgcn_readfirstlane(img0 + i * 1024));
        };
        if (has_res) fetch_residual(nt0);
        duo_barrier();  // [P]
        for (int t = 0; t < ntile; ++t) {
            const int nt = nt0 + t;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's residual has landed (and the last tile's rows are out)
            DUO_STAMP(1, t, 0);
            duo_barrier();  // [E1]
            duo_barrier();  // [E2]: the matrix wave has combined: the image holds the result (POOL: the quarter sums are parked)
            DUO_STAMP(1, t, 1);
            const bool more = has_res && t + 1 < ntile;
            if constexpr (POOL) {
                // vmgn.py:298-308: every output bin is a sum of whole quarters; this wave's 64 channels
                if (more) fetch_residual(nt + 1);   // (POOL never writes the image: the next residual can go in at once)
                const float* s_w = reinterpret_cast<const float*>(smem_ + 2 * DSLAB + 4 * DIMG + w4 * 1024);
                const int P = p.pool_nparts;
                for (int o = lane; o < P * 64; o += 64) {
                    const int c = o & 63, part = o >> 6;
                    const int q0 = p.pool_q0[part], q1 = p.pool_q1[part];
                    float v = 0.f;
                    for (int q = q0; q < q1; ++q) v += s_w[q * 64 + c];
                    if (p.pool_mean) v *= 1.f / (float)((q1 - q0) * 32);
                    const size_t oi = ((size_t)mt * P + part) * p.Cout + nt * 256 + w4 * 64 + c;
                    p.pool_out[oi] = v;
                    if (p.pool_out_lp) p.pool_out_lp[oi] = f32_to_lp16(v);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the quarter sums are read before [E1] lets the next ones in
            } else {
                // the image's rows into registers; then, row group by row group, the next residual in and the result out (two halves
                // of 64 rows: the register budget of the matrix waves' 128 accumulators binds this wave too)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4_t rows[8];
                    if (has_out) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) rows[i] = *reinterpret_cast<const lds_u32x4_t*>(img + (8 * h + i) * 1024 + lane * 16);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (h == 0) DUO_STAMP(1, t, 2);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (more) fat_dma(p.res + row_off(8 * h + i, nt + 1), __builtin_amdgcn_readfirstlane(img0 + (8 * h + i) * 1024));
                        if (has_out && m0 + 8 * (8 * h + i) + lrow < p.M) *reinterpret_cast<u32x4_t*>(p.out + row_off(8 * h + i, nt)) = rows[i];
                    }
                }
            }
            DUO_STAMP(1, t, 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================= matrix wave
    // ---- pixel staging: piece i = 2 j + h of this wave -> rows (wave + 4 j) * 8 .. + 7 of 64-channel half h; lane (lrow, lchk)
    // fetches chunk lchk ^ swizzle(row) of its row
    unsigned roff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave + 4 * j) * 8 + lrow;
        const int gm = min(m0 + row, p.M - 1);
        roff[j] = (unsigned)gm * (unsigned)p.K * 2u + (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
    }
    auto stage_piece = [&](int slab, int buf, auto i_c) {  // piece i of slab `slab` into buffer `buf`
        constexpr int I = decltype(i_c)::value, J = I >> 1, H = I & 1;
        fat_dma(p.x + roff[J] + (size_t)(slab * 256 + H * 128),
                __builtin_amdgcn_readfirstlane(lds0 + buf * DSLAB + H * DHALF + (wave + 4 * J) * 1024));
    };

    // ---- pixel fragment b (rows 16 b + (lane & 15)) of k-step kk: half kk >> 1, chunk 4 (kk & 1) + (lane >> 4)
    const int xbase = frow * 128 + ((fchunk ^ ((frow >> 1) & 7)) << 4);

    // ---- weight stream of this wave for channel tile nt: fragment q of slab s at wpk + ((nt * 4 + wave) * nslab * DPS + s * DPS + q) KiB;
    // global slab g = t * nslab + s of this workgroup's walk
    const size_t tile_stride = (size_t)4 * nslab * (DPS * 1024);
    const unsigned char* const wstream = p.wpk + (size_t)(nt0 * 4 + wave) * nslab * (DPS * 1024);
    const int gslabs = ntile * nslab;
    auto slab_weights = [&](int g) {  // (beyond the walk: its first slab again -- requested, never used)
        const int gg = g < gslabs ? g : 0;
        const int t = gg / nslab, s = gg - t * nslab;
        return wstream + (size_t)t * tile_stride + (size_t)s * (DPS * 1024);
    };
    u32x4_t wr[DRING];
    auto issue_w = [&](auto slot_c, const unsigned char* slab_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        fat_gload<(POS & 3) * 1024>(wr[SLOT], lane16, slab_base + (POS & ~3) * 1024);
    };

    asm volatile("" ::: "a127");
    sfor<32>([&](auto qc) { fat_zero<decltype(qc)::value>(); });
    const unsigned cnt_addr = lds0 + 2 * DSLAB + 4 * DIMG + 4 * 1024;
    if (tid == 0) *reinterpret_cast<__attribute__((address_space(3))) unsigned*>(smem + 2 * DSLAB + 4 * DIMG + 4 * 1024) = 0u;

    // ---- prologue: slab 0's pixel rows; then the first ring of weight fragments with slab 1's pieces behind fragments 4 .. 7 --
    // the order the loop issues them in behind fragments 12 .. 15 of the slab before, so that its counted waits hold from slab 0 on
    sfor<DPPW>([&](auto ic) { stage_piece(0, 0, ic); });
    sfor<DRING>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        issue_w(ic, wstream, ic);
        sfor<dpieces_at(I + DRING)>([&](auto jc) {
            stage_piece(1 % nslab, 1, integral_constant<int, dpiece_first(I + DRING) + decltype(jc)::value>{});
        });
    });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DRING + DPPW) : "memory");
    duo_barrier();  // [P]

    u32x4_t xf[8];
    auto ldx = [&](const lds_u8_t* sp, auto ks_c, auto b_c) {
        constexpr int KS = decltype(ks_c)::value, B = decltype(b_c)::value;
        const lds_u8_t* a = sp + (xbase ^ ((KS & 1) * 64));
        return *reinterpret_cast<const lds_u32x4_t*>(a + (KS >> 1) * DHALF + B * 2048);
    };
    sfor<8>([&](auto bc) { xf[decltype(bc)::value] = ldx(smem, integral_constant<int, 0>{}, bc); });
    const int cb0 = w4 * 64 + 8 * fchunk;
    int g = 0;
    for (int t = 0; t < ntile; ++t) {
        DUO_STAMP(0, t, 0);
        for (int s = 0; s < nslab; ++s, ++g) {
            const unsigned char* ws = slab_weights(g);
            const unsigned char* wsn = slab_weights(g + 1);
            // pixel rows two slabs ahead (the same pixel tile, whatever the channel tile); the walk's last two slabs: their own rows
            // again into the freed buffer (requested, never used)
            const int ahead = g + 2 < gslabs ? (g + 2) % nslab : s;
            const lds_u8_t* sp = smem + (g & 1) * DSLAB;
            const lds_u8_t* spn = smem + ((g + 1) & 1) * DSLAB;

            sfor<DPS>([&](auto pc) {
                constexpr int P = decltype(pc)::value;
                constexpr int KS = P >> 2, A = P & 3, SL = P % DRING;
                fat_wait<SCHED::value.allowed[P]>(wr[SL]);
                if constexpr (P == DBARRIER_AT) duo_slab_sync(cnt_addr, 4u * (unsigned)(g + 1));  // [S]
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
                sfor<dpieces_at(P)>([&](auto ic) { stage_piece(ahead, g & 1, integral_constant<int, dpiece_first(P) + decltype(ic)::value>{}); });
            });
        }
        DUO_STAMP(0, t, 1);
        // ---- combine: + bias, + residual, ReLU, round once; lane (f, row) holds channels 64 w + 32 j + 8 f .. + 7 of (b, j). The
        // ring's loads and the next slabs' pieces stay in flight: nothing here touches the vector-memory counter.
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        duo_barrier();  // [E1]: the memory wave has seen this tile's residual land in the image
        DUO_STAMP(0, t, 2);
        const int cb = (nt0 + t) * 256 + cb0;
        sfor<2>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j);
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j + 4);
            // POOL: the tile is ONE frame; fragments 2 q, 2 q + 1 are quarter q (4 image rows = 32 pixels): per quarter the sum of
            // the ROUNDED activations (what a separate pooling pass would read) over its two fragments, then over the 16 pixel lanes
            // of a fragment (same f), parked in the wave's 1 KB behind the images; the memory wave turns quarters into bins.
            float* const s_w = reinterpret_cast<float*>(smem_ + 2 * DSLAB + 4 * DIMG + w4 * 1024);  // [4 quarters][64 channels of this wave]
            float psum[8];
            sfor<8>([&](auto bc) {
                constexpr int B = decltype(bc)::value;
                const f32x4_t lo = fat_read<(2 * j) * 8 + B>(), hi = fat_read<(2 * j + 1) * 8 + B>();
                fat_zero<(2 * j) * 8 + B>();
                fat_zero<(2 * j + 1) * 8 + B>();
                float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
                lds_u32x4_t* const cell = reinterpret_cast<lds_u32x4_t*>(img + (xbase ^ (j * 64)) + B * 2048);  // row 16 B + frow, chunk 4 j + f
                if (has_res) {
                    const u32x4_t r = *cell;
                    const uint32_t r4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float l, h;
                        unpack_lp16x2(r4[e], l, h);
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
                if constexpr (POOL) {
                    const uint32_t k4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float l, h;
                        unpack_lp16x2(k4[e], l, h);
                        if constexpr ((B & 1) == 0) {   // (0.f + x first, as the accumulating form does: -0 + 0 = +0)
                            psum[2 * e] = 0.f + l;
                            psum[2 * e + 1] = 0.f + h;
                        } else {
                            psum[2 * e] += l;
                            psum[2 * e + 1] += h;
                        }
                    }
                    if constexpr (B & 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float u = psum[e];
                            u += __shfl_xor(u, 1, 64); u += __shfl_xor(u, 2, 64); u += __shfl_xor(u, 4, 64); u += __shfl_xor(u, 8, 64);
                            psum[e] = u;
                        }
                        if (frow == 0) {
                            float* d = s_w + (B >> 1) * 64 + 8 * fchunk + 32 * j;
                            *reinterpret_cast<float4*>(d) = make_float4(psum[0], psum[1], psum[2], psum[3]);
                            *reinterpret_cast<float4*>(d + 4) = make_float4(psum[4], psum[5], psum[6], psum[7]);
                        }
                    }
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DUO_STAMP(0, t, 3);
        duo_barrier();  // [E2]
    }
    // fragments and pieces requested past the end are still landing
#pragma unroll
    for (int i = 0; i < DRING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < DRING; ++i) asm volatile("" : "+v"(wr[i]));
}

int duo_launch(DuoParams& p, bool pool, hipStream_t stream, const char* who) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    // one workgroup per pixel tile walks all channel tiles; few pixel tiles: split the walk so that the grid still covers the chip
    const int mtiles = (p.M + DROWS - 1) / DROWS, nNt = p.Cout >> 8;
    int nsplit = 1;
    while (mtiles * nsplit < cus && nsplit * 2 <= nNt && nNt % (nsplit * 2) == 0) nsplit *= 2;
    if (agrl_opt_set(agrl_opts().duo_nsplit) && agrl_opts().duo_nsplit > 0 && nNt % agrl_opts().duo_nsplit == 0) nsplit = agrl_opts().duo_nsplit;
    p.nsplit = nsplit;
    const int grid = mtiles * nsplit;
    if (pool) hipLaunchKernelGGL(conv1x1_duo_kernel<true>, dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(conv1x1_duo_kernel<false>, dim3(grid), dim3(512), 0, stream, p);
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
    p.M = M; p.K = K; p.Cout = Cout; p.relu = relu;
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
    p.M = N * 128; p.K = K; p.Cout = Cout; p.relu = relu;
    return duo_launch(p, true, (hipStream_t)stream, "agrl_conv1x1_packed_res_pool");
}
