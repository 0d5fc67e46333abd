// The seam between two Bottlenecks of the MFMA-bound stages (layers 3 / 4), back to back in ONE kernel (16-bit build type):
//   out (M, N1) = relu(y2 (M, K1) @ w3 (N1, K1)^T + b3 + residual)        conv3 / bn3 / += / relu of block i,   vmgn.py:56-64
//   z   (M, N2) = relu(out @ w1n (N2, N1)^T + b1n)                        conv1 / bn1 / relu    of block i + 1, vmgn.py:48-50
// `out` (the 1024- / 2048-channel map) is written once -- it is the next block's residual -- and never read back: as two
// launches it crosses HBM three times (written, read by conv1, read as residual). The two GEMMs run on the same 128-pixel
// tile: conv3's output is produced in 256-channel chunks, each chunk is finished (bias + residual + ReLU, rounded once, stored)
// and at once contracted against the matching 256-deep k-slice of the next conv1 into an accumulator set that lives across
// the chunks (128 x N2 fp32: 64 / 128 registers per lane).
//
// Round-4 history. The first form staged BOTH weight matrices through an LDS ring (48-KB k-tiles, LDS-DMA, one barrier per
// k-tile, two tiles ahead -- all that fits beside the 64-KB hand-over buffer): parity-green and no faster than the two
// launches (layer 4: 185 us against 190). Ablations said why: without a single MFMA it still took 171 us, and a lone tile on
// an idle chip 103 us -- 0.8 us per 32-MFMA step: a chain of barrier-gated LDS-DMA round trips (issue -> landed ~1.1 us) with
// at most two steps in flight. This form therefore takes the weights OUT of the LDS:
//
//   * The weights are static, so they are packed once (agrl_bottleneck_seam_pack) into the exact order the kernel consumes them:
//     per chunk and wave one contiguous stream of 1-KiB MFMA A-fragments (lane-linear: 16 rows x 64 B of one 32-deep k-step).
//     A wave streams ITS fragments global -> registers with plain 16-byte loads (one KiB per instruction, perfectly coalesced)
//     through a ring of 8 fragments that is refilled the moment a fragment's last MFMA has issued: no staging instruction, no
//     barrier, no slot that waits for seven other waves, and 64 KB in flight per CU in registers that cost nothing extra.
//   * 512 threads = 8 waves, each owning 32 conv3 channels of the chunk (GEMM 1: 2 fragments x 8 pixel fragments) and N2 / 8
//     conv1 channels (GEMM 2: 2 / 4 fragments x 8 pixel fragments) of ALL 128 pixels, so that no weight fragment is needed by
//     two waves. Wave w's 32 finished channels are exactly k-step w of GEMM 2.
//   * LDS holds activations only: the y2 tile (K1 = 256: resident, 64 KB; K1 = 512: 16-KB k-tiles through a 5-slot ring, three
//     tiles ahead, LDS-DMA), the conv3 bias and the hand-over buffer X (64 KB): a lane's packed epilogue registers (8 consecutive channels of
//     one pixel) ARE the B fragment of one k-step of GEMM 2, so X is an array of 1-KiB fragment blocks [k-step][pixel
//     fragment], written and read lane-linearly, and the same registers go to HBM as 16-byte stores.
//   * The residual never touches LDS: a chunk's accumulators START as residual + bias (8 loads per lane, issued one GEMM 2
//     earlier). Every load of the kernel is inline asm with hand-counted vmcnt waits (hipcc's own waits would drain the rings);
//     the counts come from a constexpr simulation of one chunk's issue order (make_sched).
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
// counted wait tied to the registers it guards: their consumers (register-only MFMAs) cannot be scheduled above it
template <int N>
__device__ __forceinline__ void wait_for(u32x4_t& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_for(u32x4_t& a, u32x4_t& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }

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
template <int KS1, int FW2, bool Y2RING, int RING>
struct Sched {
    static constexpr int G1F = 2 * KS1, G2F = 8 * FW2, FPC = G1F + G2F;
    int g1[KS1];   // wait in front of GEMM 1 k-step ks (both fragments)
    int g2[G2F];   // wait in front of GEMM 2 fragment f
};
constexpr int S_OPS = (SEAM_ABL & 1) ? 0 : 16;  // per chunk epilogue: 8 out stores + 8 residual loads per lane
template <int KS1, int FW2, bool Y2RING, int RING>
constexpr Sched<KS1, FW2, Y2RING, RING> make_sched() {
    using S = Sched<KS1, FW2, Y2RING, RING>;
    S s{};
    int issued[4][S::FPC] = {};
    int seq = 0;
    for (int p = 0; p < RING; ++p) issued[0][p] = seq++;
    for (int k = 0; k < 3; ++k) {
        for (int ks = 0; ks < KS1; ++ks) {
            if (k == 1) s.g1[ks] = seq - 1 - issued[k][2 * ks + 1];
            if (Y2RING && (ks & 1) == 0) seq += 2;  // the y2 k-tile three tiles ahead: 2 DMA pieces, right behind the wait + barrier
            for (int q = 2 * ks + RING; q < 2 * ks + 2 + RING; ++q) {
                if (q >= S::FPC) issued[k + 1][q - S::FPC] = seq++;
                else issued[k][q] = seq++;
            }
        }
        seq += S_OPS;
        for (int f = 0; f < S::G2F; ++f) {
            const int p = S::G1F + f;
            if (k == 1) s.g2[f] = seq - 1 - issued[k][p];
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
__global__ __launch_bounds__(512) void bottleneck_seam_kernel(const SeamParams p) {
    static_assert((K1 == 256 || K1 == 512) && N1 % SCH == 0 && (N2 == 256 || N2 == 512), "shapes");
    constexpr int NC = N1 / SCH;            // chunks
    constexpr int KS1 = K1 / 32;            // GEMM 1 k-steps per chunk
    constexpr int FW2 = N2 / 128;           // GEMM 2 channel fragments per wave (2 / 4): N2 / 8 channels
    constexpr bool Y2RING = K1 > 256;       // the y2 tile does not fit beside X: 16-KB k-tiles through five slots, three ahead
    constexpr int NKT = K1 / 64;            // y2 k-tiles
    constexpr int NSL = 5, AHEAD = 3;       // slots of the y2 ring, k-tiles requested ahead
    constexpr int XO = Y2RING ? NSL * 16384 : 64 * 1024;
    constexpr int BIAS_OFF = XO + 64 * 1024;  // b3 (N1 floats), resident
    // With both accumulator sets at their largest (K1 = N2 = 512) hipcc is one register quad short during GEMM 1 and spills a
    // GEMM 2 accumulator to scratch -- whose reload waits vmcnt(0) and drains the weight ring once per chunk. The last PARK pixel
    // fragments' accumulators of the last channel fragment are therefore parked in LDS (8 KB each, a lane's own 16 bytes) across
    // GEMM 1: an LDS round trip per chunk, counted on lgkmcnt.
    constexpr int PARK = (K1 == 512 && N2 == 512) ? 1 : 0;
    constexpr int PARK_OFF = BIAS_OFF + N1 * 4;
    constexpr int FPC = 2 * KS1 + 8 * FW2;  // weight fragments per chunk and wave
    // weight fragments in flight per wave: 16 where the accumulators leave room (N2 = 256), else 8. The ring is what keeps a wave
    // fed while the chunk's HBM accesses (residual in, out) sit in front of its younger fragment loads in the in-order return queue
    constexpr int RING = N2 == 256 ? 16 : 8;
    using SCHED = SchedOf<KS1, FW2, Y2RING, RING>;
    static_assert(FPC % RING == 0, "the ring index of a fragment must not depend on the chunk");

    __shared__ __attribute__((aligned(16))) unsigned char smem_[PARK_OFF + PARK * 8192];
    typedef __attribute__((address_space(3))) f32x4_t lds_f32x4_t;
    lds_u8_t* const smem = (lds_u8_t*)smem_;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int m0 = blockIdx.x * SBM;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- y2 staging: piece j of a k-tile covers tile rows wave*16 + 8j + (lane >> 3), 16-byte chunk lane & 7 (swizzled)
    unsigned a_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave * 16 + j * 8 + (lane >> 3);
        a_off[j] = (unsigned)(m0 + row) * (K1 * 2) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + wave * 2048);
    auto issue_y2 = [&](int kt, int slot) {  // k-tile kt of the tile's y2 rows into slot
        const unsigned char* src = p.y2 + kt * 128;
        const unsigned dst = ldsA + slot * 16384;
        dma16s(src, a_off[0], dst);
        dma16s(src, a_off[1], dst + 1024);
    };
    // fragment reads: pixel fragment b = tile rows 16 b + frow; 128-byte rows, chunk ^ ((row >> 1) & 7)
    const int lo128 = frow * 128 + ((fchunk ^ ((frow >> 1) & 7)) << 4);
    // residual / out: lane (f = fchunk, pixel frow of fragment b) owns channels 32 wave + 8 f .. + 7 of a chunk
    const unsigned r_off = (unsigned)(m0 + frow) * (N1 * 2) + (unsigned)(wave * 32 + 8 * fchunk) * 2;

    // LDS bases as opaque registers: every access below is base + immediate (< 64 KB). Left to itself hipcc folds XO into each of
    // the 64 hand-over addresses and keeps them in 64 VGPRs.
    const lds_u8_t* xptr = smem + XO + lane16;
    asm volatile("" : "+v"(xptr));

    // ---- the weight stream of this wave: fragment position q of chunk c at wpk + ((c * 8 + wave) * FPC + q) KiB
    u32x4_t wr[RING];
    auto issue_w = [&](auto slot_c, const unsigned char* chunk_base, auto pos_c) {
        constexpr int SLOT = decltype(slot_c)::value, POS = decltype(pos_c)::value;
        if constexpr (SEAM_ABL & 8) { wr[SLOT] = u32x4_t{lane16, 2u, 3u, 4u}; return; }
        gload16_asm<(POS & 3) * 1024>(wr[SLOT], lane16, chunk_base + (POS & ~3) * 1024);
    };
    auto chunk_stream = [&](int c) { return p.wpk + (size_t)(c * 8 + wave) * (FPC * 1024); };

    f32x4_t acc1[2][8];
    f32x4_t acc2[FW2][8];
#pragma unroll
    for (int a = 0; a < FW2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t rres[8];
    auto load_residual = [&](int c) {  // one lane offset; fragment / chunk offsets on the scalar side
        const unsigned char* rc = p.res + (size_t)c * (SCH * 2);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if constexpr (SEAM_ABL & 1) { rres[b] = u32x4_t{(unsigned)c, r_off, 1u, 2u}; continue; }
            gload16_asm<0>(rres[b], r_off, rc + b * (16 * N1 * 2));
        }
    };
    auto init_acc1 = [&](int c) {  // the chunk's accumulators start as residual + bias (landed: older than fragments already waited for)
        float bias[8];
        {
            const lds_u8_t* bs = smem + BIAS_OFF + (c * SCH + wave * 32 + 8 * fchunk) * 4;
            const u32x4_t b0 = *reinterpret_cast<const lds_u32x4_t*>(bs), b1 = *reinterpret_cast<const lds_u32x4_t*>(bs + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bias[e] = __uint_as_float(b0[e]); bias[4 + e] = __uint_as_float(b1[e]); }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t w4[4] = {rres[b][0], rres[b][1], rres[b][2], rres[b][3]};
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) unpack_lp16x2(w4[e], v[2 * e], v[2 * e + 1]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (SEAM_ABL & 1) ? bias[e] : v[e] + bias[e];
            acc1[0][b] = f32x4_t{v[0], v[1], v[2], v[3]};
            acc1[1][b] = f32x4_t{v[4], v[5], v[6], v[7]};
        }
    };

    using std::integral_constant;

    if constexpr (PARK > 0) {
#pragma unroll
        for (int k = 0; k < PARK; ++k) *reinterpret_cast<lds_f32x4_t*>(smem + PARK_OFF + k * 8192 + tid * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // ---- prologue: residual + bias of chunk 0, the y2 tile (or its first four k-tiles), then the first RING weight fragments
    load_residual(0);
    if (wave < N1 / 256) dma16s(reinterpret_cast<const unsigned char*>(p.b3) + wave * 1024, lane16, lds0 + BIAS_OFF + wave * 1024);
    if constexpr (Y2RING) {
#pragma unroll
        for (int t = 0; t < AHEAD; ++t) issue_y2(t, t);
    } else {
#pragma unroll
        for (int t = 0; t < NKT; ++t) issue_y2(t, t);
    }
    {
        const unsigned char* s0 = chunk_stream(0);
        static_for<RING>([&](auto ic) { issue_w(ic, s0, ic); });
    }
    // everything older than the weight fragments has landed; the barrier publishes the y2 pieces of the other waves
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SEAM_ABL & 8) ? 0 : RING) : "memory");
    static_assert(RING <= 16, "vmcnt is a 6-bit field: ring + epilogue accesses must stay below 64");
#pragma unroll
    for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(rres[b]));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int kt_slot = 0;  // slot of the chunk's y2 k-tile 0 (Y2RING: global k-tile n sits in slot n % NSL)
    for (int c = 0;; ++c) {
        const bool last = c + 1 == NC;
        // the last chunk requests chunk 0's residual and first fragments again (never used, drained at the end): the code stays
        // branch-free and the wait counts stay the same
        const int cn = last ? 0 : c + 1;
        const unsigned char* ws = chunk_stream(c);
        const unsigned char* wsn = chunk_stream(cn);

        // residual + bias were requested one GEMM 2 ago and are older than fragments that have been waited for since; the empty
        // statements keep their (register-only) consumers behind those waits
#pragma unroll
        for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(rres[b]));
        init_acc1(c);
        __builtin_amdgcn_sched_barrier(0);

        // ================= GEMM 1: acc1 (32 channels x 128 pixels) += w3 fragments x y2 fragments ===========================
        // One pixel-fragment read feeds two MFMAs; the reads run PD fragments ahead of their MFMAs through a ring of PD + 1
        // registers, in an order pinned by sched_barrier (left alone hipcc reads two fragments, waits, issues four MFMAs: the
        // two waves of a SIMD leave every barrier in lock step, so nothing overlaps the LDS round trips). With the y2 ring
        // the pipeline restarts at every k-tile: a k-tile may only be read behind its barrier.
        {
            constexpr int PD = N2 == 256 ? 4 : 2;  // reads ahead: what the registers beside the two accumulator sets allow
            constexpr int NR = KS1 * 8;  // pixel-fragment reads of the chunk
            u32x4_t px[PD + 1];
            const lds_u8_t* sa0 = smem + lo128;          // k-step parity 0 / 1 of a k-tile: chunk ^ 4
            const lds_u8_t* sa1 = smem + (lo128 ^ 64);
            asm volatile("" : "+v"(sa0), "+v"(sa1));
            int slot_off = 0;                            // byte offset of the current k-tile's slot (y2 ring)
            auto ldpx = [&](auto rc) {
                constexpr int R = decltype(rc)::value, KS = R / 8, B = R % 8;
                const lds_u8_t* base = (KS & 1) ? sa1 : sa0;
                if constexpr (Y2RING) return lds_ld16(base + slot_off + B * 2048);
                else return lds_ld16(base + (KS >> 1) * 16384 + B * 2048);
            };
            if constexpr (!Y2RING) static_for<PD>([&](auto rc) { px[decltype(rc)::value % (PD + 1)] = ldpx(rc); });
            static_for<NR>([&](auto rc) {
                constexpr int R = decltype(rc)::value, KS = R / 8, B = R % 8;
                constexpr int S0 = (2 * KS) % RING, S1 = (2 * KS + 1) % RING;
                if constexpr (B == 0) {
                    if constexpr (SEAM_ABL & 8) { asm volatile("" : "+v"(wr[S0]), "+v"(wr[S1])); }
                    else wait_for<SCHED::value.g1[KS]>(wr[S0], wr[S1]);
                    if constexpr (Y2RING && (KS & 1) == 0) {
                        // this wave's pieces of the k-tile are older than the fragments just waited for; the barrier covers the
                        // other waves' pieces and frees the slot read two k-tiles ago for the k-tile AHEAD tiles down the stream
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        const int slot = (kt_slot + (KS >> 1)) % NSL;
                        issue_y2(((KS >> 1) + AHEAD) % NKT, (slot + AHEAD) % NSL);
                        slot_off = slot * 16384;
                        static_for<PD>([&](auto dc) { px[(R + decltype(dc)::value) % (PD + 1)] = ldpx(integral_constant<int, R + decltype(dc)::value>{}); });
                    }
                }
                if constexpr (R + PD < NR && !(Y2RING && (R + PD) / 16 != R / 16))
                    px[(R + PD) % (PD + 1)] = ldpx(integral_constant<int, R + PD>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SEAM_ABL & 2) asm volatile("" ::"v"(wr[S0]), "v"(wr[S1]), "v"(px[R % (PD + 1)]));
                else {
                    acc1[0][B] = mfma_lp16_16x16x32(wr[S0], px[R % (PD + 1)], acc1[0][B]);
                    acc1[1][B] = mfma_lp16_16x16x32(wr[S1], px[R % (PD + 1)], acc1[1][B]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (B == 7) {  // both fragments are consumed: their successors RING positions down the stream
                    constexpr int Q0 = 2 * KS + RING, Q1 = Q0 + 1;
                    if constexpr (Q0 >= FPC) issue_w(integral_constant<int, S0>{}, wsn, integral_constant<int, Q0 - FPC>{});
                    else issue_w(integral_constant<int, S0>{}, ws, integral_constant<int, Q0>{});
                    if constexpr (Q1 >= FPC) issue_w(integral_constant<int, S1>{}, wsn, integral_constant<int, Q1 - FPC>{});
                    else issue_w(integral_constant<int, S1>{}, ws, integral_constant<int, Q1>{});
                }
            });
        }
        if constexpr (Y2RING) {
            kt_slot += NKT;
            while (kt_slot >= NSL) kt_slot -= NSL;
        } else {
            // every wave is past its reads of X (the previous chunk's GEMM 2) -- the y2 ring's barriers say the same where it exists
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }

        // ================= chunk epilogue: ReLU, round once; the packed registers go to HBM (next block's residual) and to X ====
        {
            unsigned char* oc = p.out + (size_t)c * (SCH * 2);
            const lds_u8_t* xw = xptr + wave * 8192;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = relu_nan(acc1[0][b][e]);
                    v[4 + e] = relu_nan(acc1[1][b][e]);
                }
                const u32x4_t pk = {pack_lp16x2(v[0], v[1]), pack_lp16x2(v[2], v[3]), pack_lp16x2(v[4], v[5]), pack_lp16x2(v[6], v[7])};
                if (!(SEAM_ABL & 1) || pk[0] == 0x12345678u) *reinterpret_cast<u32x4_t*>(oc + b * (16 * N1 * 2) + r_off) = pk;
                *reinterpret_cast<lds_u32x4_t*>(const_cast<lds_u8_t*>(xw) + b * 1024) = pk;
            }
        }
        load_residual(cn);
        wg_barrier();  // X is complete

        // ================= GEMM 2: acc2 (N2 / 8 channels x 128 pixels) += w1n fragments x X fragments ============================
        u32x4_t xf[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) xf[b] = lds_ld16(xptr + b * 1024);
        static_for<8 * FW2>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            constexpr int KS = F / FW2, A = F % FW2;
            constexpr int SL = (2 * KS1 + F) % RING;
            if constexpr (SEAM_ABL & 8) { asm volatile("" : "+v"(wr[SL])); }
            else wait_for<SCHED::value.g2[F]>(wr[SL]);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if constexpr (PARK > 0 && A + 1 == FW2 && KS == 0) {
                    if (b >= 8 - PARK) acc2[A][b] = *reinterpret_cast<const lds_f32x4_t*>(smem + PARK_OFF + (b - (8 - PARK)) * 8192 + tid * 16);
                }
                if constexpr (SEAM_ABL & 4) asm volatile("" ::"v"(wr[SL]), "v"(xf[b]));
                else acc2[A][b] = mfma_lp16_16x16x32(wr[SL], xf[b], acc2[A][b]);
                if constexpr (PARK > 0 && A + 1 == FW2 && KS == 7) {
                    if (b >= 8 - PARK) *reinterpret_cast<lds_f32x4_t*>(smem + PARK_OFF + (b - (8 - PARK)) * 8192 + tid * 16) = acc2[A][b];
                }
                if constexpr (A + 1 == FW2 && KS + 1 < 8) {  // the next k-step's fragment replaces this one right behind its last reader
                    __builtin_amdgcn_sched_barrier(0);
                    xf[b] = lds_ld16(xptr + ((KS + 1) * 8 + b) * 1024);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int Q = 2 * KS1 + F + RING;
            if constexpr (Q >= FPC) issue_w(integral_constant<int, SL>{}, wsn, integral_constant<int, Q - FPC>{});
            else issue_w(integral_constant<int, SL>{}, ws, integral_constant<int, Q>{});
        });
        if (last) break;
    }
    // the fragments requested for a chunk that does not exist are still landing in registers the compiler now considers free
    // (the empty statements keep the ring registers allocated up to the drain)
#pragma unroll
    for (int i = 0; i < RING; ++i) asm volatile("" : "+v"(wr[i]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < RING; ++i) asm volatile("" : "+v"(wr[i]));

    if constexpr (PARK > 0) {
#pragma unroll
        for (int k = 0; k < PARK; ++k) acc2[FW2 - 1][8 - PARK + k] = *reinterpret_cast<const lds_f32x4_t*>(smem + PARK_OFF + k * 8192 + tid * 16);
    }
    // ---- z = relu(acc2 + b1n), rounded, 16-byte stores (64 contiguous bytes per pixel row and instruction)
    lp16_t* __restrict__ zp = reinterpret_cast<lp16_t*>(p.z);
    const int cb2 = wave * (N2 / 8) + 8 * fchunk;
#pragma unroll
    for (int j = 0; j < FW2 / 2; ++j) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.b1n + cb2 + 32 * j);
        const float4 b1 = *reinterpret_cast<const float4*>(p.b1n + cb2 + 32 * j + 4);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int gm = m0 + b * 16 + frow;
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

// ---- one-off packing of the two weight matrices into the per-(chunk, wave) fragment streams.
//   GEMM 1 fragment (k-step ks, a in {0, 1}): row i of the MFMA A operand = conv3 channel 256 c + 32 w + 8 (i >> 2) + 4 a + (i & 3)
//   GEMM 2 fragment (k-step ks, a < FW2):      row i = conv1 channel (N2 / 8) w + 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3)
// (the MFMA result rows 4 f + r of a fragment PAIR of a lane are then 8 consecutive channels: one 16-byte piece; igemm_wide.hip)
// lane (i = lane & 15, f = lane >> 4) holds the row's k-elements 32 ks + 8 f .. + 7.
__global__ void seam_pack_kernel(const lp16_t* __restrict__ w3, const lp16_t* __restrict__ w1n, uint4* __restrict__ wpk, int K1, int N1, int N2) {
    const int KS1 = K1 / 32, FW2 = N2 / 128, FPC = 2 * KS1 + 8 * FW2;
    const long long total = (long long)(N1 / SCH) * 8 * FPC * 64;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        long long r = t >> 6;
        const int q = (int)(r % FPC);
        r /= FPC;
        const int w = (int)(r & 7), c = (int)(r >> 3);
        const int i = lane & 15, f = lane >> 4;
        const lp16_t* src;
        if (q < 2 * KS1) {
            const int ks = q >> 1, a = q & 1;
            const int ch = c * SCH + w * 32 + 8 * (i >> 2) + 4 * a + (i & 3);
            src = w3 + (size_t)ch * K1 + ks * 32 + f * 8;
        } else {
            const int g = q - 2 * KS1, ks = g / FW2, a = g % FW2;
            const int och = w * (N2 / 8) + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
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
    const dim3 grid(M / SBM), block(512);
    if (Cmid == 256 && Cnext == 256) hipLaunchKernelGGL((bottleneck_seam_kernel<256, 1024, 256>), grid, block, 0, (hipStream_t)stream, p);
    else if (Cmid == 256) hipLaunchKernelGGL((bottleneck_seam_kernel<256, 1024, 512>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((bottleneck_seam_kernel<512, 2048, 512>), grid, block, 0, (hipStream_t)stream, p);
    AGRL_CHECK_LAUNCH("agrl_bottleneck_seam");
    return 0;
}
