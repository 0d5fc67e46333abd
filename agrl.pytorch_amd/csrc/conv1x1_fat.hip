// 1x1 convolution + folded BatchNorm (+ ReLU) over pixel rows (16-bit build type) as FOUR waves of 512 registers: the pointwise
// twin of conv3x3_fat.hip for the MFMA-bound 1x1 convs of layers 3 and 4 -- Bottleneck.conv1 / bn1 / relu (torchreid/models/
// vmgn.py:48-50) and, with a second source, conv3 / bn3 + downsample conv / BN of a layer's first block as ONE GEMM over the
// concatenated K axis (vmgn.py:56-64; both BatchNorms folded, the sum formed in fp32).
//   out (M, Cout) = act([x (M, K1) | x2 (M, K2)] @ W (Cout, K1 + K2)^T + bias)
// igemm_wide_kernel (8 waves, both operands through a 2-slot LDS-DMA ring, two barriers per 64-channel k-tile) reaches 0.42-0.46
// of the MFMA peak on these shapes. Here, as in conv3x3_fat.hip:
//   * a workgroup = 256 pixel rows x 256 output channels, 4 waves, one per SIMD; wave w owns 64 channels (4 MFMA A fragments) of
//     all 256 pixels (16 B fragments): 64 accumulator quads = 256 asm-owned AGPRs;
//   * weights: packed once (agrl_conv1x1_pack) into per-(channel tile, wave) streams of 1-KiB fragments in the order (128-channel
//     slab, k-step, fragment), streamed global -> VGPR ring (8 fragments), each refilled right behind its 16 MFMAs;
//   * LDS holds the pixel rows of the current and the next 128-channel slab (2 x 64 KB; two 64-channel halves of 256 rows x 128 B
//     in igemm_kernel's swizzled row layout), LDS-DMA in 1-KiB pieces behind the slab's first weight fragments; a k-step's 16 pixel fragments are read
//     once per wave and held while its four weight fragments pass, each replaced by its successor behind its last reader;
//   * one barrier per slab (256 MFMAs per wave), placed where no wave has to wait for LDS data behind it (see BARRIER_AT).
// M need not be a tile multiple: rows beyond M are staged from row M - 1 and not stored.
//
// Measured (rocprofv3 kernel durations, 256 frames of 16 x 8, fp16, tools/kernel_trace.sh tools/conv1x1_bench.py), this kernel /
// igemm_wide_kernel: 2048 -> 512: 58.4 / 61.0 us; 1024 -> 512: 35.3 / 35.2; [1024 | 512] -> 2048: 176-186 / 192-206;
// 1024 -> 256 (128 workgroups): 31.9 / 24.1; 512 -> 2048: 82.0 / 77.0. The model routes the first and third shape here.
// With the loop's weight loads, pixel DMA and LDS reads all compiled out (-DFAT1_ABL=7) the 2048 -> 512 case still takes 49.4 us:
// the MFMA stream itself, on random operands, runs at 1.78 PFLOP/s (tools/ubench/mfma_stream.hip; 2.39 on zeros) -- 38.7 us here.
// Tried on top of it and withdrawn (round 4, same tool):
//   * residual as two more slabs: the residual tile through the same LDS pipeline, multiplied by a 256 x 256 identity appended
//     to each channel tile's weight stream (exact; no epilogue loads, no staging registers): correct, 512 -> 2048 + residual
//     121.4 us against igemm_wide_kernel<0, 256, true>'s 114.9 -- the two extra slabs cost what the epilogue loads did;
//   * a persistent form (one workgroup per CU walks its tiles, pixel pieces and weight fragments requested across tile
//     boundaries, bias through a per-wave LDS copy, first products written with C = 0, epilogue stores counted in the next
//     tile's first waits): correct, but 126 us (residual form) / 205.8 us (two-source form) against 121.4 / 186.8 for one
//     workgroup per tile. Ablations of the persistent residual form: no pixel DMA -42 us, no stores -31, no weight loads -11,
//     no LDS reads -5, none of them 73.4: the tile is bound by its 256 + 128 KB of DMA'd rows and 128 KB of stores per 31 us,
//     not by per-tile start-up -- what a longer-lived workgroup cannot change;
//   * the residual in registers (first channel half requested before the main loop, second half behind it into the registers the
//     pixel fragments leave; added behind the bias: bit-identical to conv_bn_act(residual=...)) and, on top, the frame pooling of
//     agrl_conv1x1_bn_act_pool from the accumulators (wave-local bins, 16-lane shuffles): correct (stored map equal bit for bit,
//     pooled sums to 7e-8), 512 -> 2048 + residual 119.4 us against igemm_wide_kernel<0, 256, true>'s 115.8, pooled 175.0
//     against igemm_wide_kernel<16384, 256, true>'s 106.7.
//   * (round 5) the result rows through a wave-private 32 KB LDS image so that every global store is a whole 128-byte line per
//     eight lanes (what took 10 us off conv1x1_duo.hip's epilogue, where a second workgroup on the CU covers the round trip):
//     bit-identical, [1024 | 512] -> 2048 183.7-185.0 us against 176.7-178.1, 512 -> 2048 87-88 against 86-87, 2048 -> 512 equal --
//     with one workgroup per CU the barrier + LDS round trip is serial time that the better store pattern does not buy back.
// Four designs for conv3 + residual of layer 4 (8-wave LDS ring, this kernel with the residual as slabs / in registers /
// persistent, the back-to-back seam kernel) land within 5 % of each other at 2.6 TB/s of HBM traffic: 300 MB per launch, half of
// it written, with 33 MB of operand rows re-read through L2 by eight channel tiles.
#include "fat_dev.h"

namespace {

struct Fat1Params {
    const unsigned char* x;     // (M, K1) 16-bit pixel rows
    const unsigned char* x2;    // (M, K2) second source, or nullptr
    const unsigned char* wpk;   // packed weight streams (agrl_conv1x1_pack)
    const float* bias;          // (Cout)
    unsigned char* out;         // (M, Cout)
    int M, K1, K2, Cout, relu;
};

#ifndef FAT1_ABL
#define FAT1_ABL 0  // timing ablations (results wrong): 1 no weight loads in the loop, 2 no pixel DMA in the loop, 4 no LDS reads in the loop
#endif
constexpr int F1RING = 8;                // weight fragments in flight per wave
constexpr int F1PS = 4 * 4;              // weight fragments per 128-channel slab and wave: 4 k-steps x 4 channel fragments
constexpr int HALF_BYTES = 256 * 128;    // 256 pixel rows x 64 channels
constexpr int SLAB1 = 2 * HALF_BYTES;    // one 128-channel slab of the pixel tile
constexpr int PPW1 = 16;                 // DMA pieces (8 rows x 128 B) per wave and slab
// Slab s is read during the weight fragments 0 .. 15 of slab s -- its LAST k-step's pixel fragments behind fragment 11 -- and the
// first k-step of slab s + 1 behind fragment 15. One barrier per slab, in front of fragment 12: there every wave has issued (and
// waited out) its last reads of slab s's buffer and has waited for its own pieces of slab s + 1 (requested behind fragments
// 12 .. 15 of slab s - 1, i.e. older than the weight fragments it has consumed since), so behind the barrier (a) slab s + 1 is
// complete for everybody and (b) slab s's buffer is free: the pieces of slab s + 2 go into it behind fragments 12 .. 15, four each.
// No wave ever waits for LDS data at a slab boundary.
constexpr int BARRIER_AT = 12;
constexpr int pieces_at(int p) { return p >= BARRIER_AT ? 4 : 0; }
constexpr int piece_first(int p) { int n = 0; for (int q = 0; q < p; ++q) n += pieces_at(q); return n; }
static_assert(piece_first(F1PS) == PPW1 && BARRIER_AT >= F1RING, "all pieces placed, behind fragments whose successors' ring slots the prologue fills");
// vmcnt budget of the wait in front of fragment p of a slab (steady state)
struct Fat1Sched {
    int allowed[F1PS];
};
constexpr Fat1Sched make_fat1_sched() {
    Fat1Sched s{};
    int issued[4][F1PS] = {};
    int seq = 0;
    for (int p = 0; p < F1RING; ++p) issued[0][p] = seq++;
    for (int k = 0; k < 3; ++k)
        for (int p = 0; p < F1PS; ++p) {
            if (k == 1) s.allowed[p] = seq - 1 - issued[k][p];
            const int q = p + F1RING;
            if (q >= F1PS) issued[k + 1][q - F1PS] = seq++;
            else issued[k][q] = seq++;
            seq += pieces_at(p);  // the next slab's pieces
        }
    return s;
}
struct Fat1SchedOf {
    static constexpr Fat1Sched value = make_fat1_sched();
};

__global__ __launch_bounds__(256) void conv1x1_fat_kernel(const Fat1Params p) {
    using SCHED = Fat1SchedOf;
    __shared__ __attribute__((aligned(16))) unsigned char smem_[2 * SLAB1];
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int frow = lane & 15, fchunk = lane >> 4;

    // tile = (pixel tile mt, channel tile nt): neighbouring workgroups (same XCD: blockIdx % 8) share the pixel tile
    const int nNt = p.Cout >> 8;
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, within = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const int m0 = mt << 8;

    // ---- pixel staging: piece i of this wave = rows 64 wave' ... of one 64-channel half: i = 2 j + h -> rows (wave + 4 j) * 8 .. + 7
    // of half h; lane (lrow = lane >> 3, lchk = lane & 7) fetches chunk lchk ^ swizzle(row) of its row (igemm_kernel's layout:
    // 16-byte chunk c of row r at c ^ ((r >> 1) & 7))
    unsigned roff1[8], roff2[8];  // byte offset of the lane's row in x / x2 (+ its swizzled chunk)
    const int lrow = lane >> 3, lchk = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (wave + 4 * j) * 8 + lrow;
        const int gm = min(m0 + row, p.M - 1);
        const unsigned sw = (unsigned)((lchk ^ ((row >> 1) & 7)) << 4);
        roff1[j] = (unsigned)gm * (unsigned)p.K1 * 2u + sw;
        roff2[j] = (unsigned)gm * (unsigned)p.K2 * 2u + sw;
    }
    const int nslab1 = p.K1 >> 7, nslab = nslab1 + (p.K2 >> 7);
    auto stage_piece = [&](int slab, int buf, auto i_c) {  // piece i of slab `slab` into buffer `buf`
        constexpr int I = decltype(i_c)::value, J = I >> 1, H = I & 1;
        const bool second = slab >= nslab1;  // uniform
        const unsigned char* src = second ? p.x2 + roff2[J] + (size_t)((slab - nslab1) * 256 + H * 128)
                                          : p.x + roff1[J] + (size_t)(slab * 256 + H * 128);
        fat_dma(src, __builtin_amdgcn_readfirstlane(lds0 + buf * SLAB1 + H * HALF_BYTES + (wave + 4 * J) * 1024));
    };

    // ---- pixel fragment b (rows 16 b + (lane & 15)) of k-step kk: half kk >> 1, chunk 4 (kk & 1) + (lane >> 4)
    const int xbase = frow * 128 + ((fchunk ^ ((frow >> 1) & 7)) << 4);

    // ---- weight stream of this wave: fragment q of slab s at wpk + ((nt * 4 + wave) * nslab * F1PS + s * F1PS + q) KiB
    const unsigned char* wstream = p.wpk + (size_t)(nt * 4 + wave) * nslab * (F1PS * 1024);
    u32x4_t wr[F1RING];
    auto issue_w = [&](auto slot_c, const unsigned char* slab_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        fat_gload<(POS & 3) * 1024>(wr[SLOT], lane16, slab_base + (POS & ~3) * 1024);
    };

    asm volatile("" ::: "a255");
    sfor<64>([&](auto qc) { fat_zero<decltype(qc)::value>(); });

    // ---- prologue: slab 0's pixel rows; then the first ring of weight fragments with slab 1's pieces behind fragments 4 .. 7 --
    // the order the loop issues them in behind fragments 12 .. 15 of the slab before, so that its counted waits hold from slab 0 on
    using std::integral_constant;
    sfor<PPW1>([&](auto ic) { stage_piece(0, 0, ic); });
    sfor<F1RING>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        issue_w(ic, wstream, ic);
        sfor<pieces_at(I + F1RING)>([&](auto jc) {
            stage_piece(nslab > 1 ? 1 : 0, 1, integral_constant<int, piece_first(I + F1RING) + decltype(jc)::value>{});
        });
    });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F1RING + PPW1) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    u32x4_t xf[16];
    auto ldx = [&](const lds_u8_t* sp, auto ks_c, auto b_c) {
        constexpr int KS = decltype(ks_c)::value, B = decltype(b_c)::value;
        const lds_u8_t* a = sp + (xbase ^ ((KS & 1) * 64));
        return *reinterpret_cast<const lds_u32x4_t*>(a + (KS >> 1) * HALF_BYTES + B * 2048);
    };
    sfor<16>([&](auto bc) { xf[decltype(bc)::value] = ldx(smem, integral_constant<int, 0>{}, bc); });
    for (int slab = 0; slab < nslab; ++slab) {
        const bool more = slab + 1 < nslab;
        const unsigned char* ws = wstream + (size_t)slab * (F1PS * 1024);
        const unsigned char* wsn = wstream + (size_t)(more ? slab + 1 : 0) * (F1PS * 1024);  // past the end: slab 0 again (never used)
        const int ahead = slab + 2 < nslab ? slab + 2 : slab;  // (last two slabs: their own rows again, into the freed buffer)
        const lds_u8_t* sp = smem + (slab & 1) * SLAB1;
        const lds_u8_t* spn = smem + ((slab + 1) & 1) * SLAB1;

        sfor<F1PS>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            constexpr int KS = P >> 2, A = P & 3, SL = P % F1RING;
            fat_wait<SCHED::value.allowed[P]>(wr[SL]);
            if constexpr (P == BARRIER_AT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            sfor<16>([&](auto bc) {
                constexpr int B = decltype(bc)::value;
                fat_mfma<A * 16 + B>(wr[SL], xf[B]);
                if constexpr (A == 3 && !(FAT1_ABL & 4)) {  // the next k-step's fragment replaces this one right behind its last reader
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (KS + 1 < 4) xf[B] = ldx(sp, integral_constant<int, KS + 1>{}, bc);
                    else xf[B] = ldx(spn, integral_constant<int, 0>{}, bc);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            constexpr int Q = P + F1RING;
            if constexpr (!(FAT1_ABL & 1)) {
            if constexpr (Q >= F1PS) issue_w(integral_constant<int, SL>{}, wsn, integral_constant<int, Q - F1PS>{});
            else issue_w(integral_constant<int, SL>{}, ws, integral_constant<int, Q>{});
            }
            if constexpr (!(FAT1_ABL & 2))
            sfor<pieces_at(P)>([&](auto ic) { stage_piece(ahead, slab & 1, integral_constant<int, piece_first(P) + decltype(ic)::value>{}); });
        });
    }
    // fragments requested past the end are still landing
#pragma unroll
    for (int i = 0; i < F1RING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < F1RING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // ---- epilogue: + bias, ReLU, round once; lane (f, row) holds channels 64 wave + 32 j + 8 f .. + 7 of (b, j): 16-byte stores
    const int cb = nt * 256 + wave * 64 + 8 * fchunk;
    sfor<2>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j + 4);
        sfor<16>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            const f32x4_t lo = fat_read<(2 * j) * 16 + B>(), hi = fat_read<(2 * j + 1) * 16 + B>();
            float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
            }
            const int gm = m0 + B * 16 + frow;
            if (gm < p.M)
                *reinterpret_cast<uint4*>(p.out + ((size_t)gm * p.Cout + cb + 32 * j) * 2) =
                    make_uint4(pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7]));
        });
    });
}

// ---- one-off packing: (Cout, K) row-major -> per (channel tile nt, wave w) streams [128-channel slab][k-step][fragment a] of 1-KiB
// MFMA A fragments: lane (i = lane & 15, f = lane >> 4) holds the k-elements 128 slab + 32 kk + 8 f .. + 7 of output channel
// 256 nt + 64 w + sigma(a, i), sigma(a, i) = 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3) (igemm_wide.hip)
__global__ void conv1x1_fat_pack_kernel(const lp16_t* __restrict__ w, uint4* __restrict__ wpk, int K, int Cout) {
    const int nslab = K >> 7;
    const long long total = (long long)(Cout >> 8) * 4 * nslab * F1PS * 64;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        long long r = t >> 6;
        const int q = (int)(r % F1PS);
        r /= F1PS;
        const int slab = (int)(r % nslab);
        r /= nslab;
        const int wv = (int)(r & 3), nt = (int)(r >> 2);
        const int a = q & 3, kk = q >> 2;
        const int i = lane & 15, f = lane >> 4;
        const int ch = nt * 256 + wv * 64 + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
        wpk[t] = *reinterpret_cast<const uint4*>(w + (size_t)ch * K + slab * 128 + kk * 32 + f * 8);
    }
}

}  // namespace

extern "C" long long agrl_conv1x1_packed_bytes(int K, int Cout) {
    if (K <= 0 || K % 128 || Cout <= 0 || Cout % 256) return 0;
    return (long long)Cout * K * 2;
}

extern "C" int agrl_conv1x1_pack(const void* w, void* packed, int K, int Cout, agrl_stream_t stream) {
    AGRL_CHECK_ARG(w && packed, "agrl_conv1x1_pack: null pointer");
    AGRL_CHECK_ARG(K > 0 && K % 128 == 0 && Cout > 0 && Cout % 256 == 0, "agrl_conv1x1_pack: needs K %% 128 == 0 and Cout %% 256 == 0, got %d / %d", K, Cout);
    AGRL_CHECK_ARG((((uintptr_t)w | (uintptr_t)packed) & 15) == 0, "agrl_conv1x1_pack: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(conv1x1_fat_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const lp16_t*>(w),
                       reinterpret_cast<uint4*>(packed), K, Cout);
    AGRL_CHECK_LAUNCH("agrl_conv1x1_pack");
    return 0;
}

extern "C" int agrl_conv1x1_packed_bn_act(const void* x, const void* x2, const void* packed, const float* bias, void* out, int M, int K1,
                                          int K2, int Cout, int relu, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && packed && bias && out, "agrl_conv1x1_packed_bn_act: null pointer");
    AGRL_CHECK_ARG((x2 != nullptr) == (K2 > 0), "agrl_conv1x1_packed_bn_act: x2 and K2 go together");
    AGRL_CHECK_ARG(M > 0 && K1 > 0 && K1 % 128 == 0 && K2 >= 0 && K2 % 128 == 0 && Cout > 0 && Cout % 256 == 0,
                   "agrl_conv1x1_packed_bn_act: needs K1, K2 %% 128 == 0 and Cout %% 256 == 0; got M=%d K1=%d K2=%d Cout=%d", M, K1, K2, Cout);
    const size_t widest = (size_t)(K1 > Cout ? (K1 > K2 ? K1 : K2) : (Cout > K2 ? Cout : K2));
    AGRL_CHECK_ARG((size_t)M * widest * 2 < (1ull << 32), "agrl_conv1x1_packed_bn_act: maps beyond 4 GB are not addressed");
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)out) & 15) == 0,
                   "agrl_conv1x1_packed_bn_act: pointers must be 16-byte aligned");
    Fat1Params p;
    p.x = reinterpret_cast<const unsigned char*>(x);
    p.x2 = reinterpret_cast<const unsigned char*>(x2);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.bias = bias;
    p.out = reinterpret_cast<unsigned char*>(out);
    p.M = M; p.K1 = K1; p.K2 = K2; p.Cout = Cout; p.relu = relu;
    hipLaunchKernelGGL(conv1x1_fat_kernel, dim3(((M + 255) / 256) * (Cout >> 8)), dim3(256), 0, (hipStream_t)stream, p);
    AGRL_CHECK_LAUNCH("agrl_conv1x1_packed_bn_act");
    return 0;
}
