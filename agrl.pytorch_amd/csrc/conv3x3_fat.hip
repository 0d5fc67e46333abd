// 3x3 / stride 1 / pad 1 convolution + folded BatchNorm + ReLU on 16 x 8 pixel blocks (16-bit build type), as FOUR waves of 512
// registers -- the form the seam kernel (bottleneck_seam.hip) arrived at, applied to the dominant kernel family of the step.
//   Bottleneck.conv2 / bn2 / relu, torchreid/models/vmgn.py:52-54.
//
// conv3x3_wide_kernel (8 waves, 64 x 128 wave tiles) reads 24 LDS fragments per 64 MFMAs and stages pixels AND weights through
// LDS-DMA: 0.51 of the MFMA peak on layer 4, 0.39 on layer 3 (64 x 64 wave tiles). Here:
//   * a workgroup = PB pixel blocks (PB x 128 pixels) x 256 output channels, 4 waves, one per SIMD; wave w owns 64 channels
//     (4 MFMA A fragments) of ALL the tile's pixels (8 PB B fragments): 4 x 8 PB accumulator quads = 256 (PB = 2) / 128 (PB = 1)
//     registers, asm-owned AGPRs a[0 : ...) for the whole kernel (hipcc never allocates an AGPR: every MFMA is asm, and
//     tools/seam_check_isa.sh checks that it neither spills into them nor to scratch);
//   * the weights are static: packed once (agrl_conv3x3_pack) into per-(channel tile, wave) streams of 1-KiB fragments in the
//     order (64-channel slab, tap, k-step, fragment) and streamed global -> VGPR ring (8 fragments) with plain coalesced loads,
//     a fragment refilled the moment its 8 PB MFMAs have issued. No weight touches the LDS, no wave waits for another's weights;
//   * LDS holds only the halo patches of the current and the next 64-channel slab (PB x 23.5 KB each, LDS-DMA one piece per
//     k-step); the 8 PB pixel fragments of a (tap, k-step) are read once per wave -- shifted, swizzled reads as in
//     conv3x3_wide_kernel (igemm_dev.h: frag_px / patch_g) -- and held in registers while the k-step's four weight fragments pass;
//     each is replaced by its successor right behind its last reader: 8 PB LDS reads per 32 PB MFMAs;
//   * one barrier per slab (1152 PB MFMAs per wave) instead of one per tap-step.
// Measured (rocprofv3 kernel durations, 256 frames, fp16, tools/kernel_trace.sh tools/conv3x3_bench.py; this kernel / conv3x3_wide_kernel):
// 512 -> 512 on 16 x 8: 103-120 / 121-132 us; 256 -> 256: 35 / 40 us. A 128-channel-tile variant (waves 2 x 2) for layer 2's
// 128 -> 128 convs on 32 x 16 maps was built (bit-identical) and withdrawn: 45.9 us against conv3x3_patch_kernel's 45.3 -- two
// 64-channel slabs per tile are over before the pipeline has paid for its prologue. The same variant on layer 3 (256 pixels x 128
// channels, so that the two waves of a channel half fetch each weight fragment together): 37.1-37.3 us against 35.8-37.2.
// Round 5, two more measurements (tools/conv3x3_bench.py, HIP events, same box): (a) the one-block form now asks for 128 AGPRs
// instead of 256 (240 registers in all), so that two of its workgroups share a CU: on layer 4 (AGRL_CONV3X3_FAT_PB=1: 512 one-block
// workgroups, two per CU, against 256 two-block ones) 126.5-127.8 us against 126.3-128.2 -- equal: the kernel is at its matrix
// stream either way; (b) weight-ring depth 8 / 12 / 18 fragments on layer 3 (256 workgroups, one per CU): 45.4 / 45.2 / 45.6 us --
// the 17 us that layer 3's launch spends beside its 17.5 us of matrix work are not weight latency in the loop; (c) two sets of pixel
// fragments for the one-block form (the next k-step's fragments read one weight fragment earlier: 256 instead of 128 cycles ahead
// of their first use): 42.8 / 42.8 us -- not LDS latency either.
// Waits are hand-counted (every load is inline asm; hipcc's own waits would drain the ring) from a constexpr simulation of
// one slab's issue order.
#include "fat_dev.h"

namespace {

struct FatParams {
    const unsigned char* x;     // (F, H, W, Cin) 16-bit NHWC
    const unsigned char* wpk;   // packed weight streams (agrl_conv3x3_pack)
    const float* bias;          // (Cout)
    unsigned char* out;         // (F, H, W, Cout); split planes: (F, H, W, 3 Cout) = [hi | lo 2^11 | hi] per pixel
    int H, W, Cin, Cout, nblocks, relu;
    float alpha;                // out = act(alpha acc + bias): 1, or the power of two that un-does the split-fp16 mode's weight pre-scale
    int planes;                 // 1: write the fp32 result as fp16 planes [hi | (v - hi) 2^11 | hi] (agrl_conv3x3_packed_split16)
};

// fp32 -> the split-fp16 planes of round 6's conforming mode: hi = fp16(v) (round to nearest), lo = fp16((v - hi) 2^11) -- the difference is
// exact in fp32, the scale keeps lo a NORMAL fp16 wherever hi is one (the consumer's weight segment for the lo plane carries the 2^-11)
__device__ __forceinline__ void split16_pack8(const float v[8], uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = pack_lp16x2(v[2 * e], v[2 * e + 1]);
        float a, b;
        unpack_lp16x2(h[e], a, b);
        l[e] = pack_lp16x2((v[2 * e] - a) * 2048.f, (v[2 * e + 1] - b) * 2048.f);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

#ifndef FAT_ABL
#define FAT_ABL 0  // timing ablations (results wrong): 1 no weight loads in the loop, 2 no patch DMA, 4 no LDS reads, 8 no MFMA; 16: phase stamps (s_memtime, results right; agrl_fat3_trace_buffer, tools/fat3_timeline.py)
#endif
#ifndef FAT_RING
#define FAT_RING 8
#endif
constexpr int FRING = FAT_RING;  // weight fragments in flight per wave
constexpr int FPS = 9 * 2 * 4;  // weight fragments per slab and wave: 9 taps x 2 k-steps x 4 channel fragments
constexpr int PATCH_PIECES = 23, PATCH_BYTES_F = PATCH_PIECES * 1024;  // 18 x 10 halo pixels x 128 B, rounded to whole DMA pieces

// vmcnt budget of the wait in front of fragment p of a slab (steady state): operations issued after that fragment's load
template <int PPW>  // patch DMA pieces per wave and slab, one behind the first fragment of k-steps 0 .. PPW - 1
struct FatSched {
    int allowed[FPS];
};
template <int PPW>
constexpr FatSched<PPW> make_fat_sched() {
    FatSched<PPW> s{};
    int issued[4][FPS] = {};
    int seq = 0;
    for (int p = 0; p < FRING; ++p) issued[0][p] = seq++;
    for (int k = 0; k < 3; ++k)
        for (int p = 0; p < FPS; ++p) {
            if (k == 1) s.allowed[p] = seq - 1 - issued[k][p];
            const int q = p + FRING;
            if (q >= FPS) issued[k + 1][q - FPS] = seq++;
            else issued[k][q] = seq++;
            if ((p & 3) == 0 && (p >> 2) < PPW) seq += 1;  // the next slab's patch piece
        }
    return s;
}
template <int PPW>
struct FatSchedOf {
    static constexpr FatSched<PPW> value = make_fat_sched<PPW>();
};

#if FAT_ABL & 16
__device__ unsigned long long* g_fat3_trace = nullptr;   // profiling build: 16 stamps per workgroup
#define FAT3_STAMP(k)                                                                                                      \
    do {                                                                                                                   \
        if (g_fat3_trace && threadIdx.x == 0) g_fat3_trace[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#else
#define FAT3_STAMP(k) do { } while (0)
#endif

template <int PB>  // pixel blocks per workgroup
__global__ __launch_bounds__(256) void conv3x3_fat_kernel(const FatParams p) {
    FAT3_STAMP(0);
    constexpr int NBF = 8 * PB;                                   // pixel (B) fragments per wave
    constexpr int SLAB = PB * PATCH_BYTES_F;                      // one slab's patches
    constexpr int PPW = (PB * PATCH_PIECES + 3) / 4;              // patch pieces per wave and slab (the last ones may be dummies)
    using SCHED = FatSchedOf<PPW>;
    static_assert(PPW <= 18, "one patch piece per k-step");
    static_assert(FPS % FRING == 0, "a slab is a whole number of ring turns");
    __shared__ __attribute__((aligned(16))) unsigned char smem_[2 * SLAB + 1024];
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
    const int tw = p.W >> 3, th = p.H >> 4;
    auto block_origin = [&](int bsel, int& img, int& oy0, int& ox0) {
        const int blk = min(PB * mt + bsel, p.nblocks - 1);
        img = blk / (tw * th);
        const int trem = blk - img * (tw * th);
        oy0 = (trem / tw) << 4;
        ox0 = (trem % tw) << 3;
    };

    // ---- patch staging: piece pi = wave + 4 i of the PB x 23 pieces (8 patch pixels x 128 B each)
    unsigned poff[PPW];
    bool pok[PPW];
    int pdst[PPW];
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pi = wave + 4 * i;
        const int bsel = pi / PATCH_PIECES;
        const int piece = pi - bsel * PATCH_PIECES;
        const bool real = pi < PB * PATCH_PIECES;
        int img, oy0, ox0;
        block_origin(real ? bsel : 0, img, oy0, ox0);
        const int row = piece * 8 + (lane >> 3);
        const int py = row / 10, px = row - py * 10;
        const int iy = oy0 + py - 1, ix = ox0 + px - 1;
        pok[i] = real && row < 180 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && (PB * mt + bsel < p.nblocks);
        poff[i] = (unsigned)((((size_t)img * p.H + iy) * p.W + ix) * p.Cin * 2 + (((lane & 7) ^ patch_g(py, px)) << 4));
        pdst[i] = real ? bsel * PATCH_BYTES_F + piece * 1024 : -1;
    }
    auto stage_patch_piece = [&](int slab, int buf, int i) {  // always exactly one DMA (dummies keep every wave's vmcnt equal)
        if ((FAT_ABL & 2) && buf) return;
        const unsigned dst = pdst[i] >= 0 ? lds0 + buf * SLAB + pdst[i] : lds0 + 2 * SLAB;
        fat_dma(pok[i] ? p.x + poff[i] + slab * 128 : zsrc, __builtin_amdgcn_readfirstlane(dst));
    };

    // ---- pixel fragment b = block b >> 3, pixels 16 (b & 7) + frag_px(lane & 15): patch pixel (2 (b & 7) + r0 + tr, x0 + ts) at tap
    // (tr, ts); the swizzle term depends on (py & 1, px): not on b. One lane offset per tap, the rest immediates.
    const int fp = frag_px(frow);
    const int r0 = fp >> 3, x0 = fp & 7;
    int tbase[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tbase[t] = patch_off<10>(r0 + t / 3, x0 + t % 3, fchunk);

    // ---- weight stream of this wave: fragment q of slab s at wpk + ((nt * 4 + wave) * nslab * FPS + s * FPS + q) KiB
    const int nslab = p.Cin >> 6;
    const unsigned char* wstream = p.wpk + (size_t)(nt * 4 + wave) * nslab * (FPS * 1024);
    u32x4_t wr[FRING];
    auto issue_w = [&](auto slot_c, const unsigned char* slab_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        if ((FAT_ABL & 1) && slab_base != wstream) return;
        fat_gload<(POS & 3) * 1024>(wr[SLOT], lane16, slab_base + (POS & ~3) * 1024);
    };

    // the accumulators' AGPRs: 256 (PB = 2) / 128 (PB = 1: 240 registers in all, so that TWO one-block workgroups share a CU)
    if constexpr (PB == 2) asm volatile("" ::: "a255");
    else asm volatile("" ::: "a127");
    sfor<4 * NBF>([&](auto qc) { fat_zero<decltype(qc)::value>(); });

    // ---- prologue: slab 0's patches, then the first FRING weight fragments
#pragma unroll
    for (int i = 0; i < PPW; ++i) stage_patch_piece(0, 0, i);
    sfor<FRING>([&](auto ic) { issue_w(ic, wstream, ic); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FRING) : "memory");
    FAT3_STAMP(1);

    using std::integral_constant;
    u32x4_t xf[NBF];
    for (int slab = 0; slab < nslab; ++slab) {
        // this wave's pieces of the slab are older than fragments it has waited for; the barrier covers the other waves' and says
        // that the other buffer (read during the previous slab) is free for the next slab's pieces
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        FAT3_STAMP(2 + (slab < 8 ? slab : 8));
        const bool more = slab + 1 < nslab;
        const unsigned char* ws = wstream + (size_t)slab * (FPS * 1024);
        const unsigned char* wsn = wstream + (size_t)(more ? slab + 1 : 0) * (FPS * 1024);  // past the end: slab 0 again (never used)
        const lds_u8_t* sp = smem + (slab & 1) * SLAB;
        auto ldx = [&](auto ks_c, auto b_c) {
            constexpr int KS = decltype(ks_c)::value, B = decltype(b_c)::value;
            const lds_u8_t* a = sp + (tbase[KS >> 1] ^ ((KS & 1) * 64));
            return *reinterpret_cast<const lds_u32x4_t*>(a + (B >> 3) * PATCH_BYTES_F + (B & 7) * 2560);
        };
        sfor<NBF>([&](auto bc) { xf[decltype(bc)::value] = ldx(integral_constant<int, 0>{}, bc); });

        sfor<FPS>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            constexpr int KS = P >> 2, A = P & 3, SL = P % FRING;
            fat_wait<SCHED::value.allowed[P]>(wr[SL]);
            sfor<NBF>([&](auto bc) {
                constexpr int B = decltype(bc)::value;
                if constexpr (!(FAT_ABL & 8)) fat_mfma<A * NBF + B>(wr[SL], xf[B]);
                else asm volatile("" ::"v"(wr[SL]), "v"(xf[B]));
                if constexpr (A == 3 && KS + 1 < 18 && !(FAT_ABL & 4)) {  // the next k-step's fragment replaces this one right behind its last reader
                    __builtin_amdgcn_sched_barrier(0);
                    xf[B] = ldx(integral_constant<int, KS + 1>{}, bc);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            constexpr int Q = P + FRING;
            if constexpr (Q >= FPS) issue_w(integral_constant<int, SL>{}, wsn, integral_constant<int, Q - FPS>{});
            else issue_w(integral_constant<int, SL>{}, ws, integral_constant<int, Q>{});
            if constexpr (A == 0 && KS < PPW) stage_patch_piece(more ? slab + 1 : slab, (slab + 1) & 1, KS);  // (last slab: its own pieces again, into the idle buffer)
        });
    }
    FAT3_STAMP(11);
    // fragments requested past the end are still landing
#pragma unroll
    for (int i = 0; i < FRING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < FRING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    FAT3_STAMP(12);

    // ---- epilogue: + bias, ReLU, round once; lane (f, pixel) holds channels 64 wave + 32 j + 8 f .. + 7 of (b, j): 16-byte stores
    const int cb = nt * 256 + wave * 64 + 8 * fchunk;
    const float alpha = p.alpha;
    sfor<2>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cb + 32 * j + 4);
        sfor<NBF>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            const f32x4_t lo = fat_read<(2 * j) * NBF + B>(), hi = fat_read<(2 * j + 1) * NBF + B>();
            // (alpha = 1: fmaf(1, acc, b) == acc + b, the round-5 results bit for bit)
            float v[8] = {fmaf(alpha, lo[0], b0.x), fmaf(alpha, lo[1], b0.y), fmaf(alpha, lo[2], b0.z), fmaf(alpha, lo[3], b0.w),
                          fmaf(alpha, hi[0], b1.x), fmaf(alpha, hi[1], b1.y), fmaf(alpha, hi[2], b1.z), fmaf(alpha, hi[3], b1.w)};
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
            }
            if (PB * mt + (B >> 3) < p.nblocks) {
                int img, oy0, ox0;
                block_origin(B >> 3, img, oy0, ox0);
                const int m = (B & 7) * 16 + fp;
                const size_t gm = ((size_t)img * p.H + oy0 + (m >> 3)) * p.W + ox0 + (m & 7);
                if (p.planes) {
                    uint4 ph, pl;
                    split16_pack8(v, ph, pl);
                    unsigned char* o = p.out + (gm * 3 * p.Cout + cb + 32 * j) * 2;
                    *reinterpret_cast<uint4*>(o) = ph;
                    *reinterpret_cast<uint4*>(o + (size_t)p.Cout * 2) = pl;
                    *reinterpret_cast<uint4*>(o + (size_t)p.Cout * 4) = ph;
                } else {
                    *reinterpret_cast<uint4*>(p.out + (gm * p.Cout + cb + 32 * j) * 2) =
                        make_uint4(pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7]));
                }
            }
        });
    });
    FAT3_STAMP(13);
#if FAT_ABL & 16
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FAT3_STAMP(14);
#endif
}

// ---- the one-block form split over TWO workgroups per pixel block: 128 output channels each (wave w owns 32: the A fragments
// 2 (w & 1), 2 (w & 1) + 1 of the 64-channel stream of channel group 2 half + (w >> 1)), 64 accumulator AGPRs, ~190 registers and 48 KB
// of LDS, so that a CU holds two (or three) of them. For layers whose one-block form is ONE workgroup per CU (layer 3: 256 frames x 256
// channels): there every CU spends 10 % of the launch in its start-up and 13 % in its store burst, in lock step with every other CU
// (profiles/r05_conv3x3_fat_timeline.txt); two half-width workgroups share a CU's matrix pipes and hide part of each other's waits.
// Same k order: bit-identical. Measured (rocprofv3 kernel durations, tools/half_trace.sh): layer 3 (256 -> 256) 33.7 us against
// conv3x3_fat_kernel<1>'s 35.5 -- taken there; layer 4 (512 -> 512) 120.8 against conv3x3_fat_kernel<2>'s 115 -- not taken. Starting
// Late in round 5 the same kernel took layer 2's 128 -> 128 convs (32 x 16 maps: `halves` = 1, the weights packed as the lower half of a 256-channel
// tile, one workgroup per block): 37.5-39.0 us against conv3x3_patch_kernel<128>'s 45.2-45.9 (events, same box), step -0.6 %. Starting
// the second resident round `stagger` clocks late (so that one's start-up and store burst would fall under the other's k-loop) makes
// it SLOWER: 41.4 us by events without, 42.5-43.7 with 4 k .. 20 k clocks -- the default is 0.
constexpr int HRING = 12;                // weight fragments in flight per wave (a slab = 36 = 3 ring turns)
constexpr int HPS = 9 * 2 * 2;           // weight fragments per slab and wave: 9 taps x 2 k-steps x 2 channel fragments
constexpr int HPPW = (PATCH_PIECES + 3) / 4;
struct HalfSched {
    int allowed[HPS];
};
constexpr HalfSched make_half_sched() {
    HalfSched s{};
    int issued[4][HPS] = {};
    int seq = 0;
    for (int p = 0; p < HRING; ++p) issued[0][p] = seq++;
    for (int k = 0; k < 3; ++k)
        for (int p = 0; p < HPS; ++p) {
            if (k == 1) s.allowed[p] = seq - 1 - issued[k][p];
            const int q = p + HRING;
            if (q >= HPS) issued[k + 1][q - HPS] = seq++;
            else issued[k][q] = seq++;
            if ((p & 1) == 0 && (p >> 1) < HPPW) seq += 1;  // the next slab's patch piece
        }
    return s;
}
struct HalfSchedOf {
    static constexpr HalfSched value = make_half_sched();
};

__global__ __launch_bounds__(256, 2) void conv3x3_half_kernel(const FatParams p, int stagger, int first, int halves) {
    using SCHED = HalfSchedOf;
    using std::integral_constant;
    constexpr int NBF = 8;
    constexpr int SLAB = PATCH_BYTES_F;
    static_assert(HPS % HRING == 0 && HPPW <= 18, "a slab is a whole number of ring turns; one patch piece per k-step");
    __shared__ __attribute__((aligned(16))) unsigned char smem_[2 * SLAB + 1024];
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int frow = lane & 15, fchunk = lane >> 4;

    if (stagger > 0 && (int)blockIdx.x >= first && (int)blockIdx.x < 2 * first) {  // the second resident round starts late
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)stagger) __builtin_amdgcn_s_sleep(16);
    }

    // tile = (pixel block mt, 128-channel half ht of channel tile nt): the two halves of a block are `first` workgroups apart, i.e. in
    // different resident rounds (whichever CU they land on)
    // (halves == 1: a 128-channel conv -- layer 2 -- whose weights were packed as the lower half of a 256-channel tile: one workgroup per block)
    const int nNt = halves == 1 ? 1 : p.Cout >> 8;
    const int per_half = halves == 1 ? (int)gridDim.x : (int)(gridDim.x >> 1);
    const int half = (int)blockIdx.x >= per_half ? 1 : 0;
    int bid = (int)blockIdx.x - half * per_half;
    {
        const int nblk = per_half, q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, within = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const int tw = p.W >> 3, th = p.H >> 4;
    const int img = mt / (tw * th);
    const int trem = mt - img * (tw * th);
    const int oy0 = (trem / tw) << 4, ox0 = (trem % tw) << 3;

    // ---- patch staging: piece pi = wave + 4 i of the 23 pieces (8 patch pixels x 128 B each)
    unsigned poff[HPPW];
    bool pok[HPPW];
    int pdst[HPPW];
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
#pragma unroll
    for (int i = 0; i < HPPW; ++i) {
        const int piece = wave + 4 * i;
        const bool real = piece < PATCH_PIECES;
        const int row = piece * 8 + (lane >> 3);
        const int py = row / 10, px = row - py * 10;
        const int iy = oy0 + py - 1, ix = ox0 + px - 1;
        pok[i] = real && row < 180 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        poff[i] = (unsigned)((((size_t)img * p.H + iy) * p.W + ix) * p.Cin * 2 + (((lane & 7) ^ patch_g(py, px)) << 4));
        pdst[i] = real ? piece * 1024 : -1;
    }
    auto stage_patch_piece = [&](int slab, int buf, int i) {  // always exactly one DMA (dummies keep every wave's vmcnt equal)
        const unsigned dst = pdst[i] >= 0 ? lds0 + buf * SLAB + pdst[i] : lds0 + 2 * SLAB;
        fat_dma(pok[i] ? p.x + poff[i] + slab * 128 : zsrc, __builtin_amdgcn_readfirstlane(dst));
    };

    const int fp = frag_px(frow);
    const int r0 = fp >> 3, x0 = fp & 7;
    int tbase[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tbase[t] = patch_off<10>(r0 + t / 3, x0 + t % 3, fchunk);

    // ---- weight stream: the 64-channel stream of channel group grp = 2 half + (wave >> 1) of tile nt holds, per (tap, k-step), four
    // 1-KiB fragments a = 0 .. 3; this wave takes a = 2 (wave & 1) + {0, 1}: fragment q of a slab at (q >> 1) * 4 + 2 (wave & 1) + (q & 1)
    const int nslab = p.Cin >> 6;
    const int grp = 2 * half + (wave >> 1);
    const unsigned char* wstream = p.wpk + (size_t)(nt * 4 + grp) * nslab * (FPS * 1024) + (size_t)(wave & 1) * 2048;
    u32x4_t wr[HRING];
    auto issue_w = [&](auto slot_c, const unsigned char* slab_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        fat_gload<(POS & 1) * 1024>(wr[SLOT], lane16, slab_base + (POS >> 1) * 4096);
    };

    asm volatile("" ::: "a63");
    sfor<2 * NBF>([&](auto qc) { fat_zero<decltype(qc)::value>(); });

#pragma unroll
    for (int i = 0; i < HPPW; ++i) stage_patch_piece(0, 0, i);
    sfor<HRING>([&](auto ic) { issue_w(ic, wstream, ic); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HRING) : "memory");

    u32x4_t xf[NBF];
    for (int slab = 0; slab < nslab; ++slab) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = slab + 1 < nslab;
        const unsigned char* ws = wstream + (size_t)slab * (FPS * 1024);
        const unsigned char* wsn = wstream + (size_t)(more ? slab + 1 : 0) * (FPS * 1024);  // past the end: slab 0 again (never used)
        const lds_u8_t* sp = smem + (slab & 1) * SLAB;
        auto ldx = [&](auto ks_c, auto b_c) {
            constexpr int KS = decltype(ks_c)::value, B = decltype(b_c)::value;
            const lds_u8_t* a = sp + (tbase[KS >> 1] ^ ((KS & 1) * 64));
            return *reinterpret_cast<const lds_u32x4_t*>(a + B * 2560);
        };
        sfor<NBF>([&](auto bc) { xf[decltype(bc)::value] = ldx(integral_constant<int, 0>{}, bc); });

        sfor<HPS>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            constexpr int KS = P >> 1, A = P & 1, SL = P % HRING;
            fat_wait<SCHED::value.allowed[P]>(wr[SL]);
            sfor<NBF>([&](auto bc) {
                constexpr int B = decltype(bc)::value;
                fat_mfma<A * NBF + B>(wr[SL], xf[B]);
                if constexpr (A == 1 && KS + 1 < 18) {  // the next k-step's fragment replaces this one right behind its last reader
                    __builtin_amdgcn_sched_barrier(0);
                    xf[B] = ldx(integral_constant<int, KS + 1>{}, bc);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            constexpr int Q = P + HRING;
            if constexpr (Q >= HPS) issue_w(integral_constant<int, SL>{}, wsn, integral_constant<int, Q - HPS>{});
            else issue_w(integral_constant<int, SL>{}, ws, integral_constant<int, Q>{});
            if constexpr (A == 0 && KS < HPPW) stage_patch_piece(more ? slab + 1 : slab, (slab + 1) & 1, KS);
        });
    }
#pragma unroll
    for (int i = 0; i < HRING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < HRING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // ---- epilogue: + bias, ReLU, round once; lane (f, pixel) holds channels 64 grp + 32 (wave & 1) + 8 f .. + 7 of pixel fragment b
    const int cb = nt * 256 + grp * 64 + 32 * (wave & 1) + 8 * fchunk;
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cb + 4);
    const float alpha = p.alpha;
    sfor<NBF>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        const f32x4_t lo = fat_read<B>(), hi = fat_read<NBF + B>();
        float v[8] = {fmaf(alpha, lo[0], b0.x), fmaf(alpha, lo[1], b0.y), fmaf(alpha, lo[2], b0.z), fmaf(alpha, lo[3], b0.w),
                      fmaf(alpha, hi[0], b1.x), fmaf(alpha, hi[1], b1.y), fmaf(alpha, hi[2], b1.z), fmaf(alpha, hi[3], b1.w)};
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
        }
        const int m = B * 16 + fp;
        const size_t gm = ((size_t)img * p.H + oy0 + (m >> 3)) * p.W + ox0 + (m & 7);
        if (p.planes) {
            uint4 ph, pl;
            split16_pack8(v, ph, pl);
            unsigned char* o = p.out + (gm * 3 * p.Cout + cb) * 2;
            *reinterpret_cast<uint4*>(o) = ph;
            *reinterpret_cast<uint4*>(o + (size_t)p.Cout * 2) = pl;
            *reinterpret_cast<uint4*>(o + (size_t)p.Cout * 4) = ph;
        } else {
            *reinterpret_cast<uint4*>(p.out + (gm * p.Cout + cb) * 2) =
                make_uint4(pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7]));
        }
    });
}

// ---- one-off packing: OHWI (Cout, 3, 3, Cin) -> per (channel tile nt, wave w) streams [slab][tap][k-step][fragment a] of 1-KiB
// MFMA A fragments: lane (i = lane & 15, f = lane >> 4) holds the k-elements 64 slab + 32 kk + 8 f .. + 7 of tap t of output
// channel 256 nt + 64 w + sigma(a, i), sigma(a, i) = 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3) (igemm_wide.hip)
__global__ void conv3x3_fat_pack_kernel(const lp16_t* __restrict__ w, uint4* __restrict__ wpk, int Cin, int Cout) {
    const int nslab = Cin >> 6;
    const long long total = (long long)(Cout >> 8) * 4 * nslab * FPS * 64;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        long long r = t >> 6;
        const int q = (int)(r % FPS);
        r /= FPS;
        const int slab = (int)(r % nslab);
        r /= nslab;
        const int wv = (int)(r & 3), nt = (int)(r >> 2);
        const int a = q & 3, kk = (q >> 2) & 1, tap = q >> 3;
        const int i = lane & 15, f = lane >> 4;
        const int ch = nt * 256 + wv * 64 + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
        wpk[t] = *reinterpret_cast<const uint4*>(w + ((size_t)ch * 9 + tap) * Cin + slab * 64 + kk * 32 + f * 8);
    }
}

bool fat_shape_ok(int H, int W, int Cin, int Cout) { return H % 16 == 0 && W % 8 == 0 && Cin % 64 == 0 && Cin >= 128 && (Cout % 256 == 0 || Cout == 128); }

}  // namespace

#if FAT_ABL & 16
extern "C" int agrl_fat3_trace_buffer(void* buf) {  // profiling build only
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fat3_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" long long agrl_conv3x3_packed_bytes(int Cin, int Cout) {
    if (Cin % 64 || Cout % 256) return 0;
    return (long long)Cout * 9 * Cin * 2;
}

extern "C" int agrl_conv3x3_pack(const void* w_ohwi, void* packed, int Cin, int Cout, agrl_stream_t stream) {
    AGRL_CHECK_ARG(w_ohwi && packed, "agrl_conv3x3_pack: null pointer");
    AGRL_CHECK_ARG(Cin % 64 == 0 && Cout % 256 == 0, "agrl_conv3x3_pack: needs Cin %% 64 == 0 and Cout %% 256 == 0, got %d / %d", Cin, Cout);
    AGRL_CHECK_ARG((((uintptr_t)w_ohwi | (uintptr_t)packed) & 15) == 0, "agrl_conv3x3_pack: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(conv3x3_fat_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const lp16_t*>(w_ohwi),
                       reinterpret_cast<uint4*>(packed), Cin, Cout);
    AGRL_CHECK_LAUNCH("agrl_conv3x3_pack");
    return 0;
}

static int fat3_launch(const void* x, const void* packed, const float* bias, void* out, int N, int H, int W, int Cin, int Cout, int relu,
                       float alpha, int planes, agrl_stream_t stream);

extern "C" int agrl_conv3x3_packed_bn_act(const void* x, const void* packed, const float* bias, void* out, int N, int H, int W, int Cin,
                                          int Cout, int relu, agrl_stream_t stream) {
    return fat3_launch(x, packed, bias, out, N, H, W, Cin, Cout, relu, 1.f, 0, stream);
}

// Split-fp16 planes (round 6, the conforming mode at speed): x is (N, H, W, Cin3) fp16 with Cin3 = 3 Cin_true laid out per pixel as
// [hi | lo 2^11 | hi]; `packed` = agrl_conv3x3_pack of the fp16 OHWI weight (Cout, 3, 3, Cin3) = [wh | wh 2^-11 | wl] per tap (w
// pre-scaled by a power of two, hip_ops.split16_plane_weights): the unchanged k-loop then sums xh wh + xl wh + xh wl into ONE fp32
// accumulator; the epilogue un-scales (w_unscale), adds the bias, applies ReLU and writes the result as the same three planes
// (N, H, W, 3 Cout). vmgn.py:52-54.
extern "C" int agrl_conv3x3_packed_split16(const void* x, const void* packed, const float* bias, void* out, int N, int H, int W, int Cin3,
                                           int Cout, int relu, float w_unscale, agrl_stream_t stream) {
    AGRL_CHECK_ARG(agrl_lp16_is_f16(), "agrl_conv3x3_packed_split16: the split planes are fp16 (load libagrl_hip.so, not the bf16 build)");
    AGRL_CHECK_ARG(Cin3 % 3 == 0 && w_unscale > 0.f, "agrl_conv3x3_packed_split16: Cin3 = 3 x channels, w_unscale > 0");
    AGRL_CHECK_ARG((size_t)N * H * W * 3 * (size_t)Cout * 2 < (1ull << 32), "agrl_conv3x3_packed_split16: maps beyond 4 GB are not addressed");
    return fat3_launch(x, packed, bias, out, N, H, W, Cin3, Cout, relu, w_unscale, 1, stream);
}

static int fat3_launch(const void* x, const void* packed, const float* bias, void* out, int N, int H, int W, int Cin, int Cout, int relu,
                       float alpha, int planes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && packed && bias && out, "agrl_conv3x3_packed_bn_act: null pointer");
    AGRL_CHECK_ARG(N > 0 && fat_shape_ok(H, W, Cin, Cout),
                   "agrl_conv3x3_packed_bn_act: needs 16 x 8-divisible maps, Cin %% 64 == 0 (>= 128), Cout %% 256 == 0 or Cout == 128; got %dx%d %d->%d", H, W, Cin, Cout);
    AGRL_CHECK_ARG((size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 2 < (1ull << 32), "agrl_conv3x3_packed_bn_act: maps beyond 4 GB are not addressed");
    AGRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)out) & 15) == 0, "agrl_conv3x3_packed_bn_act: pointers must be 16-byte aligned");
    FatParams p;
    p.x = reinterpret_cast<const unsigned char*>(x);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.bias = bias;
    p.out = reinterpret_cast<unsigned char*>(out);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.alpha = alpha; p.planes = planes;
    p.nblocks = N * (H >> 4) * (W >> 3);
    if (Cout == 128) {   // layer 2's 128 -> 128 convs: `packed` holds the weights as the lower half of one 256-channel tile (upper half zero,
                         // never read); one half-width workgroup per 16 x 8 block, two or three resident per CU
        hipLaunchKernelGGL(conv3x3_half_kernel, dim3(p.nblocks), dim3(256), 0, (hipStream_t)stream, p, 0, p.nblocks, 1);
        AGRL_CHECK_LAUNCH("agrl_conv3x3_packed_bn_act");
        return 0;
    }
    const int nNt = Cout >> 8;
    // two pixel blocks per workgroup where that still gives every CU a workgroup, else one
    // one-block launches that would put ONE workgroup on a CU (layer 3): two half-width workgroups per block instead
    // (AGRL_CONV3X3_HALF = 0 / 1 forces it off / on; AGRL_CONV3X3_HALF_STAGGER: start delay of the second resident round, clocks)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int want_half = agrl_opts().conv3x3_half;
    if (agrl_opt_set(want_half) ? want_half != 0 : ((p.nblocks / 2) * nNt < 224 && p.nblocks * nNt <= cus && p.nblocks * nNt >= cus / 2)) {
        const int stagger = agrl_opt_set(agrl_opts().conv3x3_half_stagger) ? agrl_opts().conv3x3_half_stagger : 0;
        hipLaunchKernelGGL(conv3x3_half_kernel, dim3(2 * p.nblocks * nNt), dim3(256), 0, (hipStream_t)stream, p, stagger, p.nblocks * nNt, 2);
        AGRL_CHECK_LAUNCH("agrl_conv3x3_packed_bn_act");
        return 0;
    }
    // One pixel block per workgroup (240 registers: TWO workgroups per CU) unless AGRL_CONV3X3_FAT_PB=2 asks for the two-block form (one
    // 460-register workgroup per CU, the default until late in round 5). Back to back the two forms of layer 4 measure the same (126.5-127.8
    // against 126.3-128.2 us by events); INSIDE the step, where every launch starts cold behind a different kernel, the one-block form is
    // ~5 us per launch ahead: step 3.508-3.540 against 3.536-3.573 ms, eight A/B pairs on two boxes (profiles/r05_ab_conv3x3_fat_pb.txt).
    const int force_pb = agrl_opts().conv3x3_fat_pb;
    if (agrl_opt_set(force_pb) && force_pb == 2) hipLaunchKernelGGL(conv3x3_fat_kernel<2>, dim3(((p.nblocks + 1) / 2) * nNt), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv3x3_fat_kernel<1>, dim3(p.nblocks * nNt), dim3(256), 0, (hipStream_t)stream, p);
    AGRL_CHECK_LAUNCH("agrl_conv3x3_packed_bn_act");
    return 0;
}
