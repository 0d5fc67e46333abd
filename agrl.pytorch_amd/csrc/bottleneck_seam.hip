// The seam between two Bottlenecks of the MFMA-bound stages (layers 3 / 4), back to back in ONE kernel (16-bit build type):
//   out (M, N1) = relu(y2 (M, K1) @ w3 (N1, K1)^T + b3 + residual)        conv3 / bn3 / += / relu of block i,   vmgn.py:56-64
//   z   (M, N2) = relu(out @ w1n (N2, N1)^T + b1n)                        conv1 / bn1 / relu    of block i + 1, vmgn.py:48-50
// `out` (the 1024- / 2048-channel map) is written once -- it is the next block's residual -- and never read back: as two
// launches it crosses HBM three times (written, read by conv1, read as residual). The two GEMMs run on the same 128-pixel
// tile: conv3's output is produced in 256-channel chunks, each chunk is finished (bias + residual + ReLU, rounded once, stored)
// and at once contracted against the matching 256-deep k-slice of the next conv1 into an accumulator set that lives across
// the chunks (128 x N2 fp32: 64 / 128 registers per lane).
//
// How it works (form 3 of the history below):
//   * The weights are static, so they are packed once (agrl_bottleneck_seam_pack) into the exact order the kernel consumes them:
//     per chunk and wave one contiguous stream of 1-KiB MFMA A-fragments (lane-linear: 16 rows x 64 B of one 32-deep k-step).
//     A wave streams ITS fragments global -> AGPRs with plain 16-byte loads (one KiB per instruction, perfectly coalesced) through
//     a ring of 16 (8) fragments that is refilled the moment a fragment's eight MFMAs have issued: no LDS staging, no barrier.
//   * 256 threads = 4 waves, one per SIMD with 512 registers each. A wave owns 64 conv3 channels of the chunk (GEMM 1: 4 channel
//     fragments x 8 pixel fragments) and N2 / 4 conv1 channels (GEMM 2) of ALL 128 pixels: no weight fragment is needed by two
//     waves, and the 8 pixel fragments of a k-step are read from LDS once per wave and held in registers while the k-step's
//     weight fragments pass (each replaced by its successor right behind its last reader).
//   * LDS holds activations only: the y2 tile (K1 = 256: resident, 64 KB; K1 = 512: 16-KB k-tiles through a 5-slot ring, three
//     tiles ahead, LDS-DMA), the conv3 bias, and the hand-over buffer X (64 KB): a lane's packed epilogue registers (8 consecutive
//     channels of one pixel) ARE the B fragment of one k-step of GEMM 2, so X is an array of 1-KiB fragment blocks [k-step][pixel
//     fragment], written and read lane-linearly, and the same registers go to HBM as 16-byte stores.
//   * The residual never touches LDS: a chunk's accumulators START as residual + bias (16 loads per lane, issued one GEMM 2
//     earlier). Every load of the kernel is inline asm with hand-counted vmcnt waits (hipcc's own waits would drain the ring);
//     the counts come from a constexpr simulation of one chunk's issue order (make_sched).
//
// Round-4 history (every number: rocprofv3 kernel-trace minimum over 10 launches, 256 frames of 16 x 8, fp16, same box):
//   1. BOTH weight matrices staged through an LDS ring (48-KB k-tiles, LDS-DMA, one barrier per k-tile, two tiles ahead -- all
//      that fits beside the 64-KB hand-over buffer): parity-green, layer 4 185 us against 190 us for the two launches. Ablations:
//      171 us without a single MFMA, a lone tile on an idle chip 103 us -- 0.8 us per 32-MFMA step: a chain of barrier-gated
//      LDS-DMA round trips (issue -> landed ~1.1 us) with at most two steps in flight.
//   2. Weights out of the LDS: packed once (agrl_bottleneck_seam_pack) into the order the kernel consumes them and streamed
//      global -> registers; eight waves, each owning 32 channels of all 128 pixels. Same time (layer 3: 52 us, layer 4: 180 us):
//      now every wave reads every pixel fragment from LDS -- 12 MB per layer-4 tile at the ~128 B/clk the LDS delivers = 45 us
//      beside 62 us of MFMA -- and the skeleton without MFMAs, weight loads and HBM traffic still took 50 us.
//   3. This form: FOUR waves of 512 registers (one per SIMD), each owning 64 channels: half the LDS reads, B fragments of a k-step
//      held in registers, weight ring and GEMM 2's accumulators in asm-owned AGPRs. Layer 3 51.4 us against 59.1 us (37.5 + 22.6)
//      for the two launches; layer 4 178 us against 180 us; 256/1024/512 67.5 us against 70 us. What the time is: layer 3 / 4
//      without the chunk epilogues' HBM accesses 36 / 116 us (residual loads 10 / 17 us, out stores 8 / 37 us of the rest),
//      without weight loads as well 29 / 94 us, MFMA alone 16 / 62 us.
//   Tried on top of 3 and withdrawn: the weight ring 32 deep instead of 16 (52.3 us); the chunk's 32 HBM accesses spread one by
//   one behind GEMM 2's fragments instead of in a burst (52.1 us); lanes p / p + 8 swapping a piece by DPP so that every store /
//   load covers 8 pixels x 128 B = whole cache lines instead of 16 half lines (53.2 us); odd tiles starting half a chunk late so
//   that the CUs' bursts interleave (layer 4 +3.5 us with a 9 us delay). The HBM accesses of a chunk cost the same 16 us (layer 3)
//   however they are issued: they do not overlap the matrix work of the wave that issues them, and with one wave per SIMD nothing
//   else does either. The model uses the kernel where it wins (the five seams of layer 3: 3.578 -> 3.552 ms per step, same box).
//
#include <utility>

#include "igemm_dev.h"

namespace {

struct SeamParams {
    const unsigned char* y2;
    const unsigned char* wpk;   // packed weight stream (agrl_bottleneck_seam_pack)
    const float* b3;
    const unsigned char* res;
    unsigned char* out;
    const float* b1n;
    unsigned char* z;
    int M;
};

// Ablation bits for profiling builds (-DSEAM_ABL=n through tools/seam_ablate.sh; wrong results by design, never in the shipped
// library): 1 no residual loads / out stores, 2 no GEMM 1 MFMAs, 4 no GEMM 2 MFMAs, 8 no weight loads, 16 no fragment reads
#ifndef SEAM_ABL
#define SEAM_ABL 0
#endif
constexpr int SBM = 128;   // pixels per tile
constexpr int SCH = 256;   // conv3 output channels per chunk

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;      // every LDS access through 32-bit address-space-3 pointers:
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;         // generic pointers cost a 64-bit add + null check per address
__device__ __forceinline__ u32x4_t lds_ld16(const lds_u8_t* p) {
#if SEAM_ABL & 16
    return u32x4_t{(unsigned)(size_t)p, 1u, 2u, 3u};
#else
    return *reinterpret_cast<const lds_u32x4_t*>(p);
#endif
}

// 16 bytes global -> VGPRs behind hipcc's back: SGPR base + 32-bit lane offset + immediate. The destination is valid only
// after a counted wait that names it (wait_for below).
template <int IMM>
__device__ __forceinline__ void gload16_asm(u32x4_t& dst, unsigned off, const unsigned char* base) {
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(off), "s"(base), "n"(IMM) : "memory");
}
// Register files by hand. One wave per SIMD owns 512 registers: 256 architectural VGPRs + 256 accumulation registers (AGPRs).
// The AGPRs are asm-owned for the whole kernel, NAMED in the asm text and invisible to hipcc:
//   a[0 : 4 RING)            the weight-fragment ring -- global_load straight into AGPRs, read by the MFMAs as their A operand.
//                            hipcc never sees a register with a load in flight (given the chance it "spills" such registers
//                            while the data is still on its way: silent garbage);
//   a[4 RING : ...)          GEMM 2's accumulators (N2 / 4 channels x 128 pixels per wave), as many quads as fit; with N2 = 512
//                            the last few live in VGPRs as ordinary operands.
// hipcc itself never allocates an AGPR (every MFMA is asm) unless it spills VGPRs into them -- which the register budget excludes
// and tools/seam_check_isa.sh checks on every build (no v_accvgpr_* outside the asm blocks, no scratch). Everything the VALU
// touches (GEMM 1's accumulators start as the residual and end in the epilogue) lives in VGPRs. Left to hipcc, which picks ONE
// register file for all MFMA accumulators of a function, the 384 accumulator registers of the layer-4 shape end in scratch.
// hipcc pads no hazards around these statements: consecutive MFMAs here never share an accumulator (eight apart), A / B
// operands come from loads behind counted waits, and the VALU touches accumulators only behind explicit s_nops.
template <int RQ>  // acc (VGPRs) += ring quad RQ x b
__device__ __forceinline__ void mfma_v(f32x4_t& acc, const u32x4_t& b) {
    if constexpr (kLpF16) asm volatile("v_mfma_f32_16x16x32_f16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(b), "n"(4 * RQ), "n"(4 * RQ + 3));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(b), "n"(4 * RQ), "n"(4 * RQ + 3));
}
template <int AQ, int RQ>  // AGPR quad AQ += ring quad RQ x b
__device__ __forceinline__ void mfma_a(const u32x4_t& b) {
    if constexpr (kLpF16)
        asm volatile("v_mfma_f32_16x16x32_f16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(b), "n"(4 * AQ), "n"(4 * AQ + 3), "n"(4 * RQ), "n"(4 * RQ + 3));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(b), "n"(4 * AQ), "n"(4 * AQ + 3), "n"(4 * RQ), "n"(4 * RQ + 3));
}
template <int AQ>
__device__ __forceinline__ void acc_zero() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0" ::"n"(4 * AQ),
                 "n"(4 * AQ + 1), "n"(4 * AQ + 2), "n"(4 * AQ + 3));
}
template <int AQ>
__device__ __forceinline__ f32x4_t acc_read() {
    float x, y, z, w;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(x), "=v"(y), "=v"(z), "=v"(w)
                 : "n"(4 * AQ), "n"(4 * AQ + 1), "n"(4 * AQ + 2), "n"(4 * AQ + 3));
    return f32x4_t{x, y, z, w};
}
// 16 bytes per lane VGPRs -> global, same addressing (hipcc's own stores take a 64-bit address pair each: 16 VGPRs in the epilogue)
template <int IMM>
__device__ __forceinline__ void gstore16_asm(const u32x4_t& v, unsigned off, unsigned char* base) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base), "n"(IMM) : "memory");
}
// 16 bytes per lane global -> ring quad RQ: SGPR base + 32-bit lane offset + immediate
template <int RQ, int IMM>
__device__ __forceinline__ void ring_load(unsigned off, const unsigned char* base) {
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
    asm volatile("global_load_dwordx4 a[%c2:%c3], %0, %1 offset:%4" ::"v"(off), "s"(base), "n"(4 * RQ), "n"(4 * RQ + 3), "n"(IMM) : "memory");
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// LDS-DMA with the source as SGPR base + 32-bit lane offset and the destination as SGPR base (the instruction's own offset
// field is NOT used: it is added to the global AND the LDS address)
__device__ __forceinline__ void dma16s(const unsigned char* sbase, unsigned voff, unsigned lds_wave_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_wave_addr)
        : "memory");
}

// ---- the vmcnt budget of every weight-fragment wait, from a simulation of the issue order over three consecutive chunks (the
// middle one is the steady state every chunk sees: the prologue issues what a previous chunk's tail would have).
//   allowed = operations issued AFTER the awaited fragment's load and before the wait.
constexpr int NWV = 4;     // waves per workgroup: one per SIMD, 512 registers each
constexpr int FW1 = 4;     // GEMM 1 channel fragments per wave: 64 of the chunk's 256 channels
constexpr int Y2P = 4;     // LDS-DMA pieces per wave and y2 k-tile (16 KB / 4 waves)
constexpr int S_OPS = (SEAM_ABL & 1) ? 0 : 32;  // per chunk epilogue: 16 out stores + 16 residual loads per lane
template <int KS1, int FW2, bool Y2RING, int RING>
struct Sched {
    static constexpr int G1F = FW1 * KS1, G2F = 8 * FW2, FPC = G1F + G2F;
    int allowed[FPC];
};
template <int KS1, int FW2, bool Y2RING, int RING>
constexpr Sched<KS1, FW2, Y2RING, RING> make_sched() {
    using S = Sched<KS1, FW2, Y2RING, RING>;
    S s{};
    int issued[4][S::FPC] = {};
    int seq = 0;
    for (int p = 0; p < RING; ++p) issued[0][p] = seq++;
    for (int k = 0; k < 3; ++k) {
        for (int p = 0; p < S::FPC; ++p) {
            if (p == S::G1F) seq += S_OPS;  // the chunk epilogue sits between the two GEMMs
            if (k == 1) s.allowed[p] = seq - 1 - issued[k][p];
            // the y2 ring's barrier + the k-tile AHEAD tiles down the stream: in front of the last fragment of every odd k-step
            if (Y2RING && p < S::G1F && (p % FW1) == FW1 - 1 && ((p / FW1) & 1)) seq += Y2P;
            const int q = p + RING;
            if (q >= S::FPC) issued[k + 1][q - S::FPC] = seq++;
            else issued[k][q] = seq++;
        }
    }
    return s;
}
template <int KS1, int FW2, bool Y2RING, int RING>
struct SchedOf {
    static constexpr Sched<KS1, FW2, Y2RING, RING> value = make_sched<KS1, FW2, Y2RING, RING>();
};

template <int K1, int N1, int N2>
__global__ __launch_bounds__(NWV * 64) void bottleneck_seam_kernel(const SeamParams p) {
    static_assert((K1 == 256 || K1 == 512) && N1 % SCH == 0 && (N2 == 256 || N2 == 512), "shapes");
    constexpr int NC = N1 / SCH;            // chunks
    constexpr int KS1 = K1 / 32;            // GEMM 1 k-steps per chunk
    constexpr int FW2 = N2 / 64;            // GEMM 2 channel fragments per wave (4 / 8): N2 / 4 channels
    constexpr bool Y2RING = K1 > 256;       // the y2 tile does not fit beside X: 16-KB k-tiles through five slots, three ahead
    constexpr int NKT = K1 / 64;            // y2 k-tiles
    constexpr int NSL = 5, AHEAD = 3;       // slots of the y2 ring, k-tiles requested ahead
    constexpr int XO = Y2RING ? NSL * 16384 : 64 * 1024;
    constexpr int BIAS_OFF = XO + 64 * 1024;  // b3 (N1 floats), resident
    // weight fragments in flight per wave (ring quads a[0 : 4 RING)): 16, or 8 where GEMM 2's accumulators need the AGPRs
    constexpr int RING = N2 == 256 ? 16 : 8;
    constexpr int NA2 = (64 - RING) < FW2 * 8 ? (64 - RING) : FW2 * 8;  // GEMM 2 accumulator quads in AGPRs (quad q = AGPR quad RING + q)
    constexpr int NV2 = FW2 * 8 - NA2;                                  // ... and in VGPRs
    using SCHED = SchedOf<KS1, FW2, Y2RING, RING>;
    constexpr int G1F = FW1 * KS1, G2F = 8 * FW2, FPC = G1F + G2F;  // weight fragments per chunk and wave
    static_assert(FPC % RING == 0, "the ring index of a fragment must not depend on the chunk");

    __shared__ __attribute__((aligned(16))) unsigned char smem_[BIAS_OFF + N1 * 4];
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int m0 = blockIdx.x * SBM;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- y2 staging: piece j of a k-tile covers tile rows wave*32 + 8j + (lane >> 3), 16-byte chunk lane & 7 (swizzled). Rows
    // 16 apart share the swizzle term, rows 8 apart differ by chunk ^ 4: two lane offsets + 16 rows on the scalar side.
    unsigned a_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave * 32 + j * 8 + (lane >> 3);
        a_off[j] = (unsigned)(m0 + row) * (K1 * 2) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
    auto issue_y2 = [&](int kt, int slot) {  // k-tile kt of the tile's y2 rows into slot
        const unsigned char* src = p.y2 + kt * 128;
        const unsigned dst = ldsA + slot * 16384;
        dma16s(src, a_off[0], dst);
        dma16s(src, a_off[1], dst + 1024);
        dma16s(src + 16 * (K1 * 2), a_off[0], dst + 2048);
        dma16s(src + 16 * (K1 * 2), a_off[1], dst + 3072);
    };
    // fragment reads: pixel fragment b = tile rows 16 b + frow; 128-byte rows, chunk ^ ((row >> 1) & 7)
    const int lo128 = frow * 128 + ((fchunk ^ ((frow >> 1) & 7)) << 4);
    // residual / out: lane (f = fchunk, pixel frow of fragment b) owns channels 64 wave + 32 j + 8 f .. + 7 of a chunk
    const unsigned r_off = (unsigned)(m0 + frow) * (N1 * 2) + (unsigned)(wave * 64 + 8 * fchunk) * 2;

    // LDS bases as opaque registers: every access below is base + immediate (< 64 KB). Left to itself hipcc folds XO into each of
    // the 64 hand-over addresses and keeps them in 64 VGPRs.
    const lds_u8_t* xptr = smem + XO + lane16;
    const lds_u8_t* sa0 = smem + lo128;          // y2 k-step parity 0 / 1 of a k-tile: chunk ^ 4
    const lds_u8_t* sa1 = smem + (lo128 ^ 64);
    asm volatile("" : "+v"(xptr), "+v"(sa0), "+v"(sa1));

    // ---- the weight stream of this wave: fragment position q of chunk c at wpk + ((c * NWV + wave) * FPC + q) KiB
    auto issue_w = [&](auto slot_c, const unsigned char* chunk_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        if constexpr (SEAM_ABL & 8) return;
        ring_load<SLOT, (POS & 3) * 1024>(lane16, chunk_base + (POS & ~3) * 1024);
    };
    auto chunk_stream = [&](int c) { return p.wpk + (size_t)(c * NWV + wave) * (FPC * 1024); };

    f32x4_t acc1[FW1][8];
    // GEMM 2's accumulator (a, b) = quad q = 8 a + b: AGPR quad RING + q for q < NA2, else acc2v[q - NA2]; the clobber makes the
    // kernel descriptor allocate all 256 AGPRs
    asm volatile("" ::: "a255");
    static_for<NA2>([&](auto qc) { acc_zero<RING + decltype(qc)::value>(); });
    f32x4_t acc2v[NV2 > 0 ? NV2 : 1];
#pragma unroll
    for (int i = 0; i < (NV2 > 0 ? NV2 : 1); ++i) acc2v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t rres[8][2];
    auto load_residual = [&](int c) {  // one lane offset; fragment / chunk offsets on the scalar side, the channel half as immediate
        const unsigned char* rc = p.res + (size_t)c * (SCH * 2);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if constexpr (SEAM_ABL & 1) { rres[b][0] = u32x4_t{(unsigned)c, r_off, 1u, 2u}; rres[b][1] = rres[b][0]; continue; }
            gload16_asm<0>(rres[b][0], r_off, rc + b * (16 * N1 * 2));
            gload16_asm<64>(rres[b][1], r_off, rc + b * (16 * N1 * 2));
        }
    };
    auto init_acc1 = [&](int c) {  // the chunk's accumulators start as residual + bias (landed: older than fragments already waited for)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float bias[8];
            {
                const lds_u8_t* bs = smem + BIAS_OFF + (c * SCH + wave * 64 + 32 * j + 8 * fchunk) * 4;
                const u32x4_t b0 = *reinterpret_cast<const lds_u32x4_t*>(bs), b1 = *reinterpret_cast<const lds_u32x4_t*>(bs + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bias[e] = __uint_as_float(b0[e]); bias[4 + e] = __uint_as_float(b1[e]); }
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t w4[4] = {rres[b][j][0], rres[b][j][1], rres[b][j][2], rres[b][j][3]};
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) unpack_lp16x2(w4[e], v[2 * e], v[2 * e + 1]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (SEAM_ABL & 1) ? bias[e] : v[e] + bias[e];
                acc1[2 * j][b] = f32x4_t{v[0], v[1], v[2], v[3]};
                acc1[2 * j + 1][b] = f32x4_t{v[4], v[5], v[6], v[7]};
                __builtin_amdgcn_sched_barrier(0);  // piece by piece: a residual quad dies where two accumulator quads are born
            }
        }
    };

    using std::integral_constant;

    // ---- prologue: residual of chunk 0, the conv3 bias, the y2 tile (or its first AHEAD k-tiles), then the first RING fragments
    load_residual(0);
#pragma unroll
    for (int i = 0; i < N1 / 1024; ++i)
        dma16s(reinterpret_cast<const unsigned char*>(p.b3) + (wave * (N1 / 1024) + i) * 1024, lane16, lds0 + BIAS_OFF + (wave * (N1 / 1024) + i) * 1024);
#pragma unroll
    for (int t = 0; t < (Y2RING ? AHEAD : NKT); ++t) issue_y2(t, t);
    {
        const unsigned char* s0 = chunk_stream(0);
        static_for<RING>([&](auto ic) { issue_w(ic, s0, ic); });
    }
    // everything older than the weight fragments has landed; the barrier publishes the DMA pieces of the other waves
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SEAM_ABL & 8) ? 0 : RING) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // the B fragments of the current k-step (8 pixel fragments of y2 in GEMM 1, of X in GEMM 2): held in registers, each replaced
    // by its successor of the NEXT k-step right behind its last reader
    u32x4_t bf[8];
    int kt_slot = 0;   // slot of the chunk's y2 k-tile 0 (Y2RING: global k-tile n sits in slot n % NSL)
    int slot_off = 0;  // byte offset of the slot the next y2 reads go to
    auto ld_y2 = [&](auto ks_c, int b) {
        constexpr int KS = decltype(ks_c)::value;
        const lds_u8_t* base = (KS & 1) ? sa1 : sa0;
        if constexpr (Y2RING) return lds_ld16(base + slot_off + b * 2048);
        else return lds_ld16(base + (KS >> 1) * 16384 + b * 2048);
    };

    for (int c = 0;; ++c) {
        const bool last = c + 1 == NC;
        // the last chunk requests chunk 0's residual and first fragments again (never used, drained at the end): the code stays
        // branch-free and the wait counts stay the same
        const int cn = last ? 0 : c + 1;
        const unsigned char* ws = chunk_stream(c);
        const unsigned char* wsn = chunk_stream(cn);

        // residual was requested one GEMM 2 ago and is older than fragments that have been waited for since; the empty statements
        // keep its (register-only) consumers behind those waits
#pragma unroll
        for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(rres[b][0]), "+v"(rres[b][1]));
        init_acc1(c);
        __builtin_amdgcn_sched_barrier(0);
        // the first y2 k-step (its k-tile was published in the prologue / during the previous chunk's GEMM 1)
#pragma unroll
        for (int b = 0; b < 8; ++b) bf[b] = ld_y2(integral_constant<int, 0>{}, b);
        asm volatile("s_nop 4" ::: "memory");  // VALU-written accumulators -> the first MFMA's C operand

        // ================= the chunk's FPC weight fragments, one at a time: wait, 8 MFMAs against the held B fragments, refill ===
        static_for<FPC>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            constexpr bool G1 = P < G1F;
            constexpr int F = G1 ? P : P - G1F;
            constexpr int FW = G1 ? FW1 : FW2;
            constexpr int KS = F / FW, A = F % FW;
            constexpr int SL = P % RING;
            constexpr bool LAST_A = A + 1 == FW;

            if constexpr (P == G1F) {
                // ============ chunk epilogue: ReLU, round once; the packed registers go to HBM (next block's residual) and to X ==
                if constexpr (!Y2RING) {
                    // every wave is past its reads of X (the previous chunk's GEMM 2) -- the y2 ring's barriers say the same
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                mfma_drain();  // the last GEMM 1 results are in their registers before the VALU reads them
                unsigned char* oc = p.out + (size_t)c * (SCH * 2);
                const lds_u8_t* xw = xptr + wave * 16384;
                static_for<2>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = relu_nan(acc1[2 * j][b][e]);
                            v[4 + e] = relu_nan(acc1[2 * j + 1][b][e]);
                        }
                        const u32x4_t pk = {pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7])};
                        if constexpr (!(SEAM_ABL & 1)) gstore16_asm<j * 64>(pk, r_off, oc + b * (16 * N1 * 2));
                        *reinterpret_cast<lds_u32x4_t*>(const_cast<lds_u8_t*>(xw) + (j * 8 + b) * 1024) = pk;
                    }
                });
                load_residual(cn);
                wg_barrier();  // X is complete
#pragma unroll
                for (int b = 0; b < 8; ++b) bf[b] = lds_ld16(xptr + b * 1024);
            }

            if constexpr (!(SEAM_ABL & 8)) wait_vmcnt<SCHED::value.allowed[P]>();

            if constexpr (Y2RING && G1 && LAST_A && (KS & 1)) {
                // The k-tile of the next k-step: this wave's pieces of it are older than fragments it has waited for, the barrier
                // covers the other waves'; it also frees the slot read two k-tiles ago for the k-tile AHEAD tiles down the stream.
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int nslot = (kt_slot + (KS >> 1) + 1) % NSL;
                issue_y2(((KS >> 1) + AHEAD) % NKT, (nslot + AHEAD - 1) % NSL);  // tiles t+1 .. t+AHEAD are then in flight / resident
                slot_off = nslot * 16384;
            }
            static_for<8>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                if constexpr (G1) {
                    if constexpr (SEAM_ABL & 2) asm volatile("" ::"v"(bf[b]));
                    else mfma_v<SL>(acc1[A][b], bf[b]);
                } else {
                    if constexpr (SEAM_ABL & 4) asm volatile("" ::"v"(bf[b]));
                    else if constexpr (A * 8 + b < NA2) mfma_a<RING + A * 8 + b, SL>(bf[b]);
                    else mfma_v<SL>(acc2v[A * 8 + b - NA2], bf[b]);
                }
                if constexpr (LAST_A) {  // the next k-step's B fragment replaces this one right behind its last reader
                    if constexpr (G1 && KS + 1 < KS1) {
                        __builtin_amdgcn_sched_barrier(0);
                        bf[b] = ld_y2(integral_constant<int, KS + 1>{}, b);
                    } else if constexpr (!G1 && KS + 1 < 8) {
                        __builtin_amdgcn_sched_barrier(0);
                        bf[b] = lds_ld16(xptr + ((KS + 1) * 8 + b) * 1024);
                    }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            constexpr int Q = P + RING;
            if constexpr (Q >= FPC) issue_w(integral_constant<int, SL>{}, wsn, integral_constant<int, Q - FPC>{});
            else issue_w(integral_constant<int, SL>{}, ws, integral_constant<int, Q>{});
            if constexpr (Y2RING && P + 1 == G1F) kt_slot = (kt_slot + NKT) % NSL;
        });
        if (last) break;
    }
    // the fragments requested for a chunk that does not exist are still landing
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mfma_drain();

    // ---- z = relu(acc2 + b1n), rounded, 16-byte stores (64 contiguous bytes per pixel row and instruction)
    // (the lane's coordinates are derived afresh: a lane constant kept alive across the chunk loop for this is one VGPR too many
    // with N2 = 512 -- hipcc parks it in an AGPR, i.e. in the weight ring)
    unsigned lane_z;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_z));
    const int frow_z = lane_z & 15, fchunk_z = lane_z >> 4;
    lp16_t* __restrict__ zp = reinterpret_cast<lp16_t*>(p.z);
    const int cb2 = wave * (N2 / NWV) + 8 * fchunk_z;
    static_for<FW2 / 2>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float4 b0 = *reinterpret_cast<const float4*>(p.b1n + cb2 + 32 * j);
        const float4 b1 = *reinterpret_cast<const float4*>(p.b1n + cb2 + 32 * j + 4);
        static_for<8>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const int gm = m0 + b * 16 + frow_z;
            f32x4_t lo, hi;
            if constexpr ((2 * j) * 8 + b < NA2) lo = acc_read<RING + (2 * j) * 8 + b>();
            else lo = acc2v[(2 * j) * 8 + b - NA2];
            if constexpr ((2 * j + 1) * 8 + b < NA2) hi = acc_read<RING + (2 * j + 1) * 8 + b>();
            else hi = acc2v[(2 * j + 1) * 8 + b - NA2];
            float v[8];
            v[0] = lo[0] + b0.x; v[1] = lo[1] + b0.y; v[2] = lo[2] + b0.z; v[3] = lo[3] + b0.w;
            v[4] = hi[0] + b1.x; v[5] = hi[1] + b1.y; v[6] = hi[2] + b1.z; v[7] = hi[3] + b1.w;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu_nan(v[e]);
            *reinterpret_cast<uint4*>(zp + (size_t)gm * N2 + cb2 + 32 * j) =
                make_uint4(pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7]));
        });
    });
}

// ---- one-off packing of the two weight matrices into the per-(chunk, wave) fragment streams.
//   GEMM 1 fragment (k-step ks, a < 4):   row i of the MFMA A operand = conv3 channel 256 c + 64 w + sigma(a, i)
//   GEMM 2 fragment (k-step ks, a < FW2): row i = conv1 channel (N2 / 4) w + sigma(a, i)
//   sigma(a, i) = 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3): the MFMA result rows 4 f + r of a fragment PAIR of a lane are
//   then 8 consecutive channels (one 16-byte piece; igemm_wide.hip)
// lane (i = lane & 15, f = lane >> 4) holds the row's k-elements 32 ks + 8 f .. + 7.
__global__ void seam_pack_kernel(const lp16_t* __restrict__ w3, const lp16_t* __restrict__ w1n, uint4* __restrict__ wpk, int K1, int N1, int N2) {
    const int KS1 = K1 / 32, FW2 = N2 / 64, G1F = FW1 * KS1, FPC = G1F + 8 * FW2;
    const long long total = (long long)(N1 / SCH) * NWV * FPC * 64;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        long long r = t >> 6;
        const int q = (int)(r % FPC);
        r /= FPC;
        const int w = (int)(r % NWV), c = (int)(r / NWV);
        const int i = lane & 15, f = lane >> 4;
        const lp16_t* src;
        if (q < G1F) {
            const int ks = q / FW1, a = q % FW1;
            const int ch = c * SCH + w * 64 + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
            src = w3 + (size_t)ch * K1 + ks * 32 + f * 8;
        } else {
            const int g = q - G1F, ks = g / FW2, a = g % FW2;
            const int och = w * (N2 / NWV) + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
            src = w1n + (size_t)och * N1 + c * SCH + ks * 32 + f * 8;
        }
        wpk[t] = *reinterpret_cast<const uint4*>(src);
    }
}

bool seam_shape_ok(int Cmid, int Cout, int Cnext) {
    return (Cmid == 256 && Cout == 1024 && Cnext == 256) || (Cmid == 512 && Cout == 2048 && Cnext == 512) ||
           (Cmid == 256 && Cout == 1024 && Cnext == 512);
}

}  // namespace

extern "C" long long agrl_bottleneck_seam_packed_bytes(int Cmid, int Cout, int Cnext) {
    if (!seam_shape_ok(Cmid, Cout, Cnext)) return 0;
    return ((long long)Cout * Cmid + (long long)Cnext * Cout) * 2;
}

extern "C" int agrl_bottleneck_seam_pack(const void* w3, const void* w1_next, void* packed, int Cmid, int Cout, int Cnext,
                                         agrl_stream_t stream) {
    AGRL_CHECK_ARG(w3 && w1_next && packed, "agrl_bottleneck_seam_pack: null pointer");
    AGRL_CHECK_ARG(seam_shape_ok(Cmid, Cout, Cnext), "agrl_bottleneck_seam_pack: built for 256/1024/256, 256/1024/512 and 512/2048/512 channels, got %d/%d/%d",
                   Cmid, Cout, Cnext);
    AGRL_CHECK_ARG((((uintptr_t)w3 | (uintptr_t)w1_next | (uintptr_t)packed) & 15) == 0, "agrl_bottleneck_seam_pack: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(seam_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const lp16_t*>(w3),
                       reinterpret_cast<const lp16_t*>(w1_next), reinterpret_cast<uint4*>(packed), Cmid, Cout, Cnext);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_seam_pack");
    return 0;
}

extern "C" int agrl_bottleneck_seam(const void* y2, const void* packed, const float* b3, const void* residual, void* out,
                                    const float* b1_next, void* z, int M, int Cmid, int Cout, int Cnext, agrl_stream_t stream) {
    AGRL_CHECK_ARG(y2 && packed && b3 && residual && out && b1_next && z, "agrl_bottleneck_seam: null pointer");
    AGRL_CHECK_ARG(seam_shape_ok(Cmid, Cout, Cnext), "agrl_bottleneck_seam: built for 256/1024/256, 256/1024/512 and 512/2048/512 channels, got %d/%d/%d",
                   Cmid, Cout, Cnext);
    AGRL_CHECK_ARG(M > 0 && M % SBM == 0, "agrl_bottleneck_seam: the pixel count must be a multiple of %d (whole 16 x 8 frames), got %d", SBM, M);
    AGRL_CHECK_ARG((size_t)M * Cout * 2 < (1ull << 32), "agrl_bottleneck_seam: maps beyond 4 GB are not addressed");
    const uintptr_t al = (uintptr_t)y2 | (uintptr_t)packed | (uintptr_t)b3 | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)b1_next | (uintptr_t)z;
    AGRL_CHECK_ARG((al & 15) == 0, "agrl_bottleneck_seam: pointers must be 16-byte aligned");
    SeamParams p;
    p.y2 = reinterpret_cast<const unsigned char*>(y2);
    p.wpk = reinterpret_cast<const unsigned char*>(packed);
    p.b3 = b3;
    p.res = reinterpret_cast<const unsigned char*>(residual);
    p.out = reinterpret_cast<unsigned char*>(out);
    p.b1n = b1_next;
    p.z = reinterpret_cast<unsigned char*>(z);
    p.M = M;
    const dim3 grid(M / SBM), block(NWV * 64);
    if (Cmid == 256 && Cnext == 256) hipLaunchKernelGGL((bottleneck_seam_kernel<256, 1024, 256>), grid, block, 0, (hipStream_t)stream, p);
    else if (Cmid == 256) hipLaunchKernelGGL((bottleneck_seam_kernel<256, 1024, 512>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((bottleneck_seam_kernel<512, 2048, 512>), grid, block, 0, (hipStream_t)stream, p);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_seam");
    return 0;
}
