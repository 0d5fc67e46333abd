// Streaming form of the query x gallery distance matrix (torchreid/metrics/distance.py:59-89) for gfx950: a handful of
// queries (m <= 64: one eval batch) against a long resident gallery. The contraction is tiny (2 m n D flops); the
// kernel's job is to pull the gallery through the chip ONCE at HBM speed: bytes = (m + n) D e + 4 m n.
//
//   * grid = ceil(n / 32) workgroups of 256 threads; a workgroup owns 32 gallery rows over the FULL depth D -- no
//     split-K partials, no second pass, one launch (the tiled igemm form needed 8-way split-K + a reduce kernel here)
//   * 8-slot LDS ring of 128-byte-deep k-tiles (gallery 32 x 128 B + queries 16 QF x 128 B per slot), filled by
//     LDS-DMA with COUNTED vmcnt waits: 6 k-tiles stay in flight across every barrier, ~28 KB of gallery per
//     workgroup, 2-3 workgroups per CU -- enough outstanding bytes to cover the HBM latency at full rate
//   * the queries' k-tile rides along (L2-resident after its first touch)
//   * wave w multiplies k-half (w & 1) of gallery fragment (w >> 1) with every query fragment: one fragment read per
//     operand per k-tile; the two k-halves are added through LDS at the end in a fixed order (deterministic)
//   * MFMA 16x16x32 bf16 / 16x16x4 fp32 (exact); D[g][q]: a lane ends with 4 consecutive gallery columns of one
//     query row -> float4 stores
#include "igemm_dev.h"

namespace {

constexpr int DS_NS = 8;    // ring slots
constexpr int DS_BN = 32;   // gallery rows per workgroup

template <typename TIN, int QF>
__global__ __launch_bounds__(256) void distmat_stream_kernel(const IgemmParams p) {
    constexpr int ES = sizeof(TIN);
    constexpr int NPIECE = DS_BN / 8 + 2 * QF;    // 1-KiB DMA pieces per k-tile
    constexpr int PPW = (NPIECE + 3) / 4;         // per wave (waves with fewer real pieces issue dummies: equal vmcnt)
    constexpr int G_BYTES = DS_BN * 128, Q_BYTES = 16 * QF * 128, SLOT = G_BYTES + Q_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[DS_NS * SLOT + 1024];
    unsigned char* s_dummy = smem + DS_NS * SLOT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = blockIdx.x * DS_BN;
    const unsigned char* __restrict__ qg = reinterpret_cast<const unsigned char*>(p.x);   // queries  (M x K)
    const unsigned char* __restrict__ gg = reinterpret_cast<const unsigned char*>(p.w);   // gallery  (N x K)
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
    const size_t row_bytes = (size_t)p.K * ES;

    // this wave's pieces: piece pc = wave + 4 i; pc < 4: gallery rows 8 pc .. +7, else query rows 8 (pc - 4) .. +7
    const int lrow = lane >> 3, lchk = lane & 7;
    const unsigned char* src[PPW];
    int dst[PPW];
    bool adv[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + 4 * i;
        src[i] = zsrc;
        dst[i] = -1;
        adv[i] = false;
        if (pc < DS_BN / 8) {
            const int row = pc * 8 + lrow;
            dst[i] = pc * 1024;
            if (g0 + row < p.N) {
                src[i] = gg + (size_t)(g0 + row) * row_bytes + ((lchk ^ ((row >> 1) & 7)) << 4);
                adv[i] = true;
            }
        } else if (pc < NPIECE) {
            const int row = (pc - DS_BN / 8) * 8 + lrow;
            dst[i] = G_BYTES + (pc - DS_BN / 8) * 1024;
            if (row < p.M) {
                src[i] = qg + (size_t)row * row_bytes + ((lchk ^ ((row >> 1) & 7)) << 4);
                adv[i] = true;
            }
        }
    }
    auto stage = [&](int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            dma16(src[i], dst[i] >= 0 ? smem + slot * SLOT + dst[i] : s_dummy);
            if (adv[i]) src[i] += 128;
        }
    };

    const int nk = (int)(row_bytes >> 7);
#pragma unroll
    for (int s = 0; s < DS_NS - 1; ++s)
        if (s < nk) stage(s);

    const int kk = wave & 1, gf = wave >> 1;
    const int frow = lane & 15, fchunk = lane >> 4;
    f32x4_t acc[QF];
#pragma unroll
    for (int q = 0; q < QF; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tiles younger than kt still in flight: min(NS-2, nk-1-kt); full depth in steady state, drained at the tail
        if (kt + DS_NS - 2 < nk) wait_vmcnt<PPW*(DS_NS - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + DS_NS - 1 < nk) {
            int fill = cur + DS_NS - 1;
            fill = fill >= DS_NS ? fill - DS_NS : fill;  // the slot read in iteration kt-1
            stage(fill);
        }
        const unsigned char* sg = smem + cur * SLOT;
        const unsigned char* sq = sg + G_BYTES;
        const uint4 ga = *reinterpret_cast<const uint4*>(sg + lds_off(gf * 16 + frow, kk * 4 + fchunk));
        uint4 qa[QF];
#pragma unroll
        for (int q = 0; q < QF; ++q) qa[q] = *reinterpret_cast<const uint4*>(sq + lds_off(q * 16 + frow, kk * 4 + fchunk));
#pragma unroll
        for (int q = 0; q < QF; ++q) acc[q] = Frag<TIN>::mma(ga, qa[q], acc[q]);
        cur = cur + 1 == DS_NS ? 0 : cur + 1;
    }
    // add the two k-halves (wave w ^ 1) in a fixed order: odd waves park their partials, even waves finish
    wg_barrier();
    float* s_part = reinterpret_cast<float*>(smem);  // [2 gf][QF][64 lanes][4]
    if (kk == 1) {
#pragma unroll
        for (int q = 0; q < QF; ++q)
            *reinterpret_cast<float4*>(s_part + ((gf * QF + q) * 64 + lane) * 4) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    }
    wg_barrier();
    if (kk == 0) {
        const int gcol = g0 + gf * 16 + fchunk * 4;  // 4 consecutive gallery columns
        float cv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.colv) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gcol + r < p.N) cv[r] = p.colv[gcol + r];
        }
        float* outp = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int q = 0; q < QF; ++q) {
            const int qrow = q * 16 + frow;
            if (qrow >= p.M) continue;
            const float4 o4 = *reinterpret_cast<const float4*>(s_part + ((gf * QF + q) * 64 + lane) * 4);
            const float rv = p.rowv ? p.rowv[qrow] : p.rowc;
            float v[4];
            v[0] = fmaf(p.alpha, acc[q][0] + o4.x, rv + cv[0]);
            v[1] = fmaf(p.alpha, acc[q][1] + o4.y, rv + cv[1]);
            v[2] = fmaf(p.alpha, acc[q][2] + o4.z, rv + cv[2]);
            v[3] = fmaf(p.alpha, acc[q][3] + o4.w, rv + cv[3]);
            float* dst4 = outp + (size_t)qrow * p.ldo + gcol;
            if (p.vec_ok && gcol + 3 < p.N) {
                *reinterpret_cast<float4*>(dst4) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gcol + r < p.N) dst4[r] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 form with the QUERIES IN REGISTERS: no LDS staging, no barrier in the stream loop.
//   * 512 threads = 8 waves; wave w owns the k-slice [w D/8, (w+1) D/8) of every row. Its slice of the (<= 32) queries
//     is loaded once into 16 NS VGPRs per query fragment (MFMA B operands, NS = D/256 k-steps of 32) and stays there.
//   * the gallery streams HBM -> VGPR -> MFMA: one 16-byte load per lane per (16 rows x 32 k) A fragment, a 16-deep
//     register ring of inline-asm loads with hand-counted s_waitcnt (hipcc neither reorders nor drains them),
//     16 KB in flight per wave, 128 KB per CU
//   * a workgroup owns 16 GF gallery rows, GF chosen by the host so that the grid is about one workgroup per CU;
//     the 8 k-slice partials are added through LDS in a fixed order (deterministic) by the wave that writes the tile
__device__ inline f32x4_t ds_gload16(const void* p) {
    f32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline void ds_landed(int n, f32x4_t& v) {  // at most n younger loads in flight (n folds to a constant)
    switch (n) {
#define DS_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(v) : : "memory"); break
        DS_W(0); DS_W(1); DS_W(2); DS_W(3); DS_W(4); DS_W(5); DS_W(6); DS_W(7); DS_W(8); DS_W(9); DS_W(10); DS_W(11);
        DS_W(12); DS_W(13); DS_W(14); DS_W(15);
#undef DS_W
        default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); break;
    }
}

template <int QF, int GF, int NS>
__global__ __launch_bounds__(512) void distmat_regq_kernel(const IgemmParams p) {
    constexpr int RD = 16;        // ring depth (loads in flight per wave)
    constexpr int NL = GF * NS;   // gallery fragment loads per wave
    __shared__ __attribute__((aligned(16))) float s_part[8 * GF * QF * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    const int g0 = blockIdx.x * (16 * GF);
    const size_t row_bytes = (size_t)p.K * 2;
    const size_t koff = (size_t)wave * (NS * 64) + kg * 16;  // byte offset of this lane's 8 k values inside a row
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* gb = reinterpret_cast<const unsigned char*>(p.w);

    f32x4_t qreg[QF][NS];
#pragma unroll
    for (int q = 0; q < QF; ++q) {
        const unsigned char* src = qb + (size_t)min(q * 16 + i16, p.M - 1) * row_bytes + koff;
#pragma unroll
        for (int s = 0; s < NS; ++s) qreg[q][s] = ds_gload16(src + s * 64);
    }
    const unsigned char* grow[GF];
#pragma unroll
    for (int g = 0; g < GF; ++g) grow[g] = gb + (size_t)min(g0 + g * 16 + i16, p.N - 1) * row_bytes + koff;
    f32x4_t ring[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i)
        if (i < NL) ring[i] = ds_gload16(grow[i / NS] + (i % NS) * 64);

    f32x4_t acc[GF][QF];
#pragma unroll
    for (int g = 0; g < GF; ++g)
#pragma unroll
        for (int q = 0; q < QF; ++q) acc[g][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        ds_landed(NL - 1 - i < RD - 1 ? NL - 1 - i : RD - 1, ring[i % RD]);
        if (i == 0) {  // the queries were requested before the first gallery load: they are here too
#pragma unroll
            for (int q = 0; q < QF; ++q)
#pragma unroll
                for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(qreg[q][s]));
        }
        const f32x4_t a = ring[i % RD];
#pragma unroll
        for (int q = 0; q < QF; ++q) acc[i / NS][q] = mfma_lp16_16x16x32(a, qreg[q][i % NS], acc[i / NS][q]);
        if (i + RD < NL) {
            // (the register allocator orders the refill behind the MFMAs that read the old value)
            ring[i % RD] = ds_gload16(grow[(i + RD) / NS] + ((i + RD) % NS) * 64);
        }
    }
    // ---- add the 8 k-slice partials: s_part[wave][frag][lane] (float4)
#pragma unroll
    for (int g = 0; g < GF; ++g)
#pragma unroll
        for (int q = 0; q < QF; ++q)
            *reinterpret_cast<float4*>(s_part + (((wave * GF + g) * QF + q) * 64 + lane) * 4) =
                make_float4(acc[g][q][0], acc[g][q][1], acc[g][q][2], acc[g][q][3]);
    __syncthreads();
    float* outp = reinterpret_cast<float*>(p.out);
    for (int fr = wave; fr < GF * QF; fr += 8) {
        const int g = fr / QF, q = fr - g * QF;
        float4 t = *reinterpret_cast<const float4*>(s_part + (((0 * GF + g) * QF + q) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float4 o = *reinterpret_cast<const float4*>(s_part + (((w * GF + g) * QF + q) * 64 + lane) * 4);
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        const int qrow = q * 16 + i16;
        const int gcol = g0 + g * 16 + kg * 4;  // D[gallery row 4 kg + r][query i16]
        if (qrow >= p.M || gcol >= p.N) continue;
        const float rv = p.rowv ? p.rowv[qrow] : p.rowc;
        float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(p.alpha, v[r], rv + ((p.colv && gcol + r < p.N) ? p.colv[gcol + r] : 0.f));
        float* dst4 = outp + (size_t)qrow * p.ldo + gcol;
        if (p.vec_ok && gcol + 3 < p.N) {
            *reinterpret_cast<float4*>(dst4) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gcol + r < p.N) dst4[r] = v[r];
        }
    }
}

template <int QF, int NS>
int launch_regq(const IgemmParams& p, hipStream_t stream) {
    // gallery rows per workgroup: the smallest multiple of 16 (<= 64) that brings the grid down to about one
    // workgroup per CU
    int gf = cdiv(cdiv(p.N, 16), 256);
    gf = gf < 1 ? 1 : (gf > 4 ? 4 : gf);
    const int grid = cdiv(p.N, 16 * gf);
    switch (gf) {
        case 1: hipLaunchKernelGGL((distmat_regq_kernel<QF, 1, NS>), dim3(grid), dim3(512), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((distmat_regq_kernel<QF, 2, NS>), dim3(grid), dim3(512), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((distmat_regq_kernel<QF, 3, NS>), dim3(grid), dim3(512), 0, stream, p); break;
        default: hipLaunchKernelGGL((distmat_regq_kernel<QF, 4, NS>), dim3(grid), dim3(512), 0, stream, p); break;
    }
    AGRL_CHECK_LAUNCH("agrl_distmat(regq)");
    return 0;
}

template <typename TIN>
int launch_stream(const IgemmParams& p, hipStream_t stream) {
    const int grid = cdiv(p.N, DS_BN);
    if (p.M <= 16) hipLaunchKernelGGL((distmat_stream_kernel<TIN, 1>), dim3(grid), dim3(256), 0, stream, p);
    else if (p.M <= 32) hipLaunchKernelGGL((distmat_stream_kernel<TIN, 2>), dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((distmat_stream_kernel<TIN, 4>), dim3(grid), dim3(256), 0, stream, p);
    AGRL_CHECK_LAUNCH("agrl_distmat(stream)");
    return 0;
}

}  // namespace

bool distmat_stream_applicable(const IgemmParams& p, int elem_size) {
    if (p.M > 64 || p.N < 2048) return false;             // few queries, long gallery
    if (((size_t)p.K * elem_size) % 128 != 0) return false;  // whole 128-byte k-tiles
    return ((((uintptr_t)p.x) | ((uintptr_t)p.w)) & 15) == 0;
}

int launch_distmat_stream(const IgemmParams& p_in, int dtype, hipStream_t stream) {
    IgemmParams p = p_in;
    p.vec_ok = (p.ldo & 3) == 0 && (((uintptr_t)p.out) & 15) == 0;
    if (dtype == AGRL_F32) return launch_stream<float>(p, stream);
    if (p.M <= 32) {  // queries-in-registers form: D = 256 NS
        const int qf = p.M <= 16 ? 1 : 2;
#define REGQ(NS_) return qf == 1 ? launch_regq<1, NS_>(p, stream) : launch_regq<2, NS_>(p, stream)
        if (p.K == 4096) REGQ(16);
        if (p.K == 2048) REGQ(8);
        if (p.K == 1024) REGQ(4);
#undef REGQ
    }
    return launch_stream<lp16_t>(p, stream);
}
