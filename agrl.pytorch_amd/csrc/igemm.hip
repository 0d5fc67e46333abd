// Implicit-GEMM convolution / linear / distance-matrix kernel for gfx950 (MI355X).
//
// One kernel family covers every dense contraction of the hot path:
//   * Bottleneck convs (1x1, 3x3, stride 1|2) + folded BN + residual + ReLU   (vmgn.py:45-65)
//   * GraphLayer Linear(2048,2048,bias=False)                                  (vmgn.py:148)
//   * query x gallery distance matrix                                          (metrics/distance.py:59-89)
//
// GEMM view:  out[m][n] = epi( sum_k X[m][k] * Wt[n][k] ),  m = output pixel (n,oh,ow) of an NHWC
// tensor, n = output channel, k = (r, s, cin) of an OHWI weight. Both operands are "K-contiguous",
// so 16-byte chunks of K are the unit of staging: a k-tile is 128 bytes per row (64 bf16 / 32 fp32),
// which lies inside one filter tap because Cin % (128/sizeof(T)) == 0.
//
// MI355X mapping
//   * 256 threads = 4 wavefronts (2 x 2), block tile BM x BN, wave tile (BM/2) x (BN/2)
//   * MFMA 16x16: bf16 -> v_mfma_f32_16x16x32_bf16, fp32 -> v_mfma_f32_16x16x4_f32 (exact fp32 fma
//     chain); the WEIGHT fragment is the A operand and the PIXEL fragment the B operand, so a lane
//     ends up holding 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write), double-buffered LDS,
//     one barrier per k-tile; the DMA of tile t+1 is issued before the MFMAs of tile t; padded / out-of-range
//     chunks are DMA'd from a 16-byte zero block
//   * bf16 epilogue: the residual tile is DMA'd into the idle staging buffer during the last k-tile, combined
//     in fp32 in place, and the finished tile leaves as whole rows (16 B per lane, full cache lines)
//   * LDS rows are 128 B; 16-byte chunk c of row r lives at chunk c ^ ((r>>1)&7): ds_read_b128
//     fragment reads and ds_write_b128 staging writes are both bank-conflict free
//   * blockIdx -> tile map keeps the N-tiles of one M-tile on the same XCD (private L2) so the
//     gathered activation tile is fetched from HBM once
#include <stdlib.h>

#include "agrl_common.h"

#include "igemm_dev.h"


template <typename TIN, typename TOUT, int BM, int BN, bool LDS_EPI, int NS, int NW>  // NW waves: (NW/2) x 2 grid
__global__ __launch_bounds__(64 * NW) void igemm_kernel(const IgemmParams p) {
    constexpr int WM = NW / 2;          // wave rows
    constexpr int EPC = DT<TIN>::epc;   // elements per 16-byte chunk
    constexpr int BKE = 8 * EPC;        // elements per k-tile (128 bytes)
    constexpr int AJ = BM / (8 * NW);   // 8-row DMA pieces per wave for the pixel tile
    constexpr int BJ = BN / (8 * NW);   // ... for the weight tile
    constexpr int FM = BM / (16 * WM);  // 16-pixel fragments per wave
    constexpr int FN = BN / 32;         // 16-channel fragments per wave
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;

    constexpr int DPT = AJ + BJ;        // DMA instructions per thread per k-tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM;
    const int wn = wave / WM;

    // ---- XCD-aware block -> tile map: blocks b, b+8, b+16.. run on one XCD (observed b % 8);
    // give each XCD a contiguous range of tiles, N-tiles of an M-tile adjacent.
    const int nNt = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, within = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    int mt = bid / nNt;
    int nt = bid - mt * nNt;
    if (p.nmajor) {  // consecutive tile ids (= one XCD's range) walk the M-tiles of one N-tile
        const int nMt = nblk / nNt;
        nt = bid / nMt;
        mt = bid - nt * nMt;
    }
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* __restrict__ wg = reinterpret_cast<const unsigned char*>(p.w);
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);

    // ---- per-lane DMA coordinates. DMA piece j of this wave covers tile rows wave*(BM/4) + 8j .. +7:
    // lane L -> row + (L>>3), physical chunk L&7, which holds global chunk (L&7) ^ ((row>>1)&7).
    const int lrow = lane >> 3;
    const int lchk = lane & 7;
    int a_ih0[AJ], a_iw0[AJ], a_nbase[AJ], a_coff[AJ], a_gm[AJ];
    bool a_ok[AJ];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = wave * (BM / NW) + j * 8 + lrow;
        const int gm = m0 + row;
        a_ok[j] = gm < p.M;
        const int gmc = a_ok[j] ? gm : 0;
        const int n = gmc / ohw;
        const int rem = gmc - n * ohw;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        a_ih0[j] = oh * p.stride - p.pad;
        a_iw0[j] = ow * p.stride - p.pad;
        a_nbase[j] = n * p.H * p.W;
        a_gm[j] = gmc;
        a_coff[j] = (lchk ^ ((row >> 1) & 7)) * EPC;
    }
    size_t b_off[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = wave * (BN / NW) + j * 8 + lrow;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        b_off[j] = ((size_t)gn * p.K + (lchk ^ ((row >> 1) & 7)) * EPC) * sizeof(TIN);
    }

    int tap_r = 0, tap_s = 0, c0 = 0;  // filter tap and channel offset of the tile being STAGED
    size_t kbyte = 0;                  // byte offset of that tile along K in the weight rows
    int nk = p.K / BKE;
    [[maybe_unused]] int c_kt = 0;     // first k-tile of this workgroup
    if (p.ksplit > 1) {  // split-K (pointwise problems only): this workgroup owns k-tiles [z*nk/ks, (z+1)*nk/ks)
        const int per = nk / p.ksplit;
        c0 = blockIdx.y * per * BKE;
        kbyte = (size_t)blockIdx.y * per * 128;
        nk = per;
        c_kt = blockIdx.y * per;
    }

    // one 1-KiB DMA piece of the k-tile being staged: pieces 0..AJ-1 are pixel rows, AJ..AJ+BJ-1 weight rows
    auto stage_piece = [&](int buf, int idx) {
        if (idx < AJ) {
            const int j = idx;
            unsigned char* sa = smem + buf * BUF_BYTES + wave * (BM / NW) * 128;
            if (p.x2) {
                // two-source pointwise form (round 6; conv3 + the block's downsample conv as ONE GEMM over [x sampled at the stride | x2],
                // vmgn.py:56-64 -- the shortcut map is neither written nor read back): k-columns [0, K1) come from x (row length K1,
                // pixel (oh * stride, ow * stride)), [K1, K) from x2 (row length K - K1, the output pixel itself). Uniform branch.
                if (c0 < p.K1) {
                    const size_t off = ((size_t)(a_nbase[j] + a_ih0[j] * p.W + a_iw0[j]) * p.K1 + c0 + a_coff[j]) * sizeof(TIN);
                    dma16(a_ok[j] ? xg + off : zsrc, sa + j * 1024);
                } else {
                    const size_t off = ((size_t)a_gm[j] * (p.Cin - p.K1) + (c0 - p.K1) + a_coff[j]) * sizeof(TIN);
                    dma16(a_ok[j] ? reinterpret_cast<const unsigned char*>(p.x2) + off : zsrc, sa + j * 1024);
                }
                return;
            }
            const int ih = a_ih0[j] + tap_r;
            const int iw = a_iw0[j] + tap_s;
            const bool ok = a_ok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const size_t off = ((size_t)(a_nbase[j] + ih * p.W + iw) * p.Cin + c0 + a_coff[j]) * sizeof(TIN);
            dma16(ok ? xg + off : zsrc, sa + j * 1024);
        } else {
            const int j = idx - AJ;
            unsigned char* sb = smem + buf * BUF_BYTES + A_BYTES + wave * (BN / NW) * 128;
            dma16(wg + b_off[j] + kbyte, sb + j * 1024);
        }
    };
    auto stage_advance = [&]() {
        kbyte += 128;
        c0 += BKE;
        if (c0 == p.Cin) {
            c0 = 0;
            if (++tap_s == p.S) {
                tap_s = 0;
                ++tap_r;
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AJ + BJ; ++i) stage_piece(buf, i);
        stage_advance();
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- LDS-staged epilogue geometry (bf16 output): the out tile is BM rows of BN*2 bytes, 16-byte chunk c of
    // row r stored at chunk c ^ (r & (CPR-1)); the residual tile is DMA'd into the same image beforehand.
    constexpr int CPR = BN * 2 / 16;        // chunks per out-tile row (16 for BN=128, 8 for BN=64)
    constexpr int ROWB = BN * 2;            // bytes per out-tile row
    constexpr int RPI = 64 / CPR;           // rows per DMA piece
    constexpr int RJ = BM / (NW * RPI);     // residual DMA pieces per wave
    const TOUT* __restrict__ resp = reinterpret_cast<const TOUT*>(p.res);
    auto stage_residual = [&](int buf) {
        unsigned char* so = smem + buf * BUF_BYTES;
        const unsigned char* rg = reinterpret_cast<const unsigned char*>(p.res);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int row0 = (wave * RJ + j) * RPI;
            const int row = row0 + lane / CPR;
            const int gch = (lane % CPR) ^ (row & (CPR - 1));
            const int gm = m0 + row;
            const int gn = n0 + gch * 8;
            const bool ok = gm < p.M && gn < p.N;  // N % 8 == 0 is guaranteed on this path
            dma16(ok ? rg + ((size_t)gm * p.ldo + gn) * 2 : zsrc, so + row0 * ROWB);
        }
    };

    // per-channel vector (bias) of this lane's 4 channels per fragment: fetched now so its latency hides under the
    // main loop instead of stalling the epilogue (it was ~1/3 of the short-K layers' time)
    float cvr[FN][4];
    if constexpr (LDS_EPI) {
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int gn = n0 + wn * (BN / 2) + a * 16 + (lane >> 4) * 4;
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.colv && gn < p.N) c4 = *reinterpret_cast<const float4*>(p.colv + gn);
            cvr[a][0] = c4.x; cvr[a][1] = c4.y; cvr[a][2] = c4.z; cvr[a][3] = c4.w;
        }
    }

    // ---- main loop: NS-deep LDS ring, NS-1 k-tiles of DMA in flight, ONE raw barrier per k-tile and a COUNTED
    // vmcnt so the younger tiles' DMA stays in flight across the barrier (a plain __syncthreads() would drain it)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) stage(s);
    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    int cur = 0, last_fill = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tiles issued after tile kt so far: min(NS-2, nk-1-kt); everything older must have landed
        const int younger = min(NS - 2, nk - 1 - kt);
        if (AGRL_DBG_BITS(p) & 32) {
        } else if (NS >= 3 && younger == 1) wait_vmcnt<DPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the buffer read in iteration kt-1 is free now: refill it with tile kt+NS-1, or with the residual tile
        int fill = cur + NS - 1;
        fill = fill >= NS ? fill - NS : fill;
        last_fill = fill;
        // The DMA pieces of the next k-tile are interleaved with the MFMA groups (one piece per FM MFMAs): a burst of
        // 8 back-to-back pieces fills the memory pipeline's queue and stalls the wave for ~500 cycles, spread out they
        // ride under the matrix work. sched_barrier pins that order.
        bool do_stage = kt + NS - 1 < nk && !(AGRL_DBG_BITS(p) & 8);
        if (do_stage && (AGRL_DBG_BITS(p) & 64)) {  // A/B switch: burst-issue the whole k-tile up front
            stage(fill);
            do_stage = false;
        }
        if (!do_stage && LDS_EPI && resp && kt == nk - 1) stage_residual(fill);
        constexpr int GROUPS = 2 * FN;
        constexpr int PPG = (DPT + GROUPS - 1) / GROUPS;
        const unsigned char* sa = smem + cur * BUF_BYTES;
        const unsigned char* sb = sa + A_BYTES;
        if constexpr (DT<TIN>::code == AGRL_F32H3) {
            // split-fp16: both k-halves of the tile feed ONE K = 32 MFMA triple per fragment pair (Frag<f32h_t>::mma32)
            uint4 xh[FM], xl[FM], wh[FN], wl[FN];
            // (workgroup-uniform) the pixel rows of this k-tile were stored pre-split by the producing kernel's epilogue: operands as they stand
            const bool a_pre = p.x2 ? ((c_kt + kt) * BKE >= p.K1 ? (p.a_pre & 2) : (p.a_pre & 1)) : (p.a_pre & 1);
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int row = wm * (BM / WM) + b * 16 + frow;
                const uint4 c0_ = *reinterpret_cast<const uint4*>(sa + lds_off(row, fchunk));
                const uint4 c1_ = *reinterpret_cast<const uint4*>(sa + lds_off(row, 4 + fchunk));
                if (a_pre) { xh[b] = c0_; xl[b] = c1_; }
                else Frag<f32h_t>::split8(c0_, c1_, xh[b], xl[b]);
            }
            // the weights arrive PRE-SPLIT (agrl_split16_weights_inloop): chunk g of a k-tile holds the fp16 high halves of this lane's eight
            // k values, chunk 4 + g their low halves -- the same bytes as eight fp32, no VALU work per fragment
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int row = wn * (BN / 2) + a * 16 + frow;
                wh[a] = *reinterpret_cast<const uint4*>(sb + lds_off(row, fchunk));
                wl[a] = *reinterpret_cast<const uint4*>(sb + lds_off(row, 4 + fchunk));
            }
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                if (do_stage) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int q = 0; q < PPG; ++q) {
                            const int idx = (kk * FN + a) * PPG + q;
                            if (idx < DPT) stage_piece(fill, idx);
                        }
                }
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] = Frag<f32h_t>::mma32(wh[a], wl[a], xh[b], xl[b], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 xf[FM], wf[FN];
#pragma unroll
            for (int b = 0; b < FM; ++b)
                xf[b] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * (BM / WM) + b * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < FN; ++a)
                wf[a] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                if (do_stage) {
#pragma unroll
                    for (int q = 0; q < PPG; ++q) {
                        const int idx = (kk * FN + a) * PPG + q;
                        if (idx < DPT) stage_piece(fill, idx);
                    }
                }
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] = Frag<TIN>::mma(wf[a], xf[b], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (do_stage) stage_advance();
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    // `last_fill` = the slot freed in the last iteration: it holds the residual tile and will hold the out tile
    wait_vmcnt<0>();
    __syncthreads();

    // ---- epilogue: lane (g = lane>>4, j = lane&15) holds channels 4g..4g+3 of pixel j per fragment
    TOUT* __restrict__ outp = reinterpret_cast<TOUT*>(p.out);
    if (p.ksplit > 1) outp += (size_t)blockIdx.y * p.M * p.ldo;
    if constexpr (LDS_EPI) {
        // phase 1: combine in fp32, round once, park the bf16 tile in the free staging buffer (in place over the
        // residual image: every lane overwrites exactly the 8 bytes it just read)
        unsigned char* so = smem + last_fill * BUF_BYTES;
        if (!(AGRL_DBG_BITS(p) & 4))
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int prow = wm * (BM / WM) + b * 16 + frow;
            const int gm = m0 + prow;
            const float rv = p.rowv ? p.rowv[gm < p.M ? gm : 0] : p.rowc;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int c = wn * (BN / 2) + a * 16 + fchunk * 4;
                const float* cv = cvr[a];
                unsigned char* slot = so + prow * ROWB + (((c >> 3) ^ (prow & (CPR - 1))) << 4) + ((c & 4) << 1);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(p.alpha, acc[a][b][r], rv + cv[r]);
                if (resp) {
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
        __syncthreads();
        // phase 2: whole 16-byte chunks, a wavefront writes 4 (BN=128) / 8 (BN=64) complete rows per instruction
        constexpr int RPT = 64 * NW / CPR;  // rows covered by the workgroup per pass
        const int pch = tid % CPR;
        const int r0 = tid / CPR;
#pragma unroll
        for (int i = 0; i < BM / RPT; ++i) {
            const int row = r0 + i * RPT;
            const int gch = pch ^ (row & (CPR - 1));
            const int gm = m0 + row;
            const int gn = n0 + gch * 8;
            if (gm < p.M && gn < p.N && !(AGRL_DBG_BITS(p) & 1)) {
                const uint4 v = *reinterpret_cast<const uint4*>(so + row * ROWB + (pch << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + ((size_t)gm * p.ldo + gn) * 2) = v;
            }
        }
    } else {
        if constexpr (DT<TIN>::code == AGRL_F32H3 && sizeof(TOUT) == 4) {
            if (p.out_pre) {
                // pre-split output (conv1 / conv2 of a Bottleneck: read by ONE consumer, as a GEMM operand). A lane's two fragments 2t, 2t + 1
                // hold channels 4g..4g+3 and 16+4g..16+4g+3 of a 32-channel group: exactly lane group g's eight k values in the consumer
                // -> fp16 halves formed here, once (the consumer's taps would each redo it), 16 bytes of hi at chunk g, lo at chunk 4 + g
                unsigned char* ob = reinterpret_cast<unsigned char*>(p.out);
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int gm = m0 + wm * (BM / WM) + b * 16 + frow;
                    if (gm >= p.M) continue;
#pragma unroll
                    for (int t = 0; t < FN / 2; ++t) {
                        const int gn0 = n0 + wn * (BN / 2) + t * 32;
                        if (gn0 >= p.N) continue;   // (N % 32 == 0: whole groups)
                        uint32_t v[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float cv[4] = {0.f, 0.f, 0.f, 0.f};
                            if (p.colv) {
                                const float4 c4 = *reinterpret_cast<const float4*>(p.colv + gn0 + h * 16 + fchunk * 4);
                                cv[0] = c4.x; cv[1] = c4.y; cv[2] = c4.z; cv[3] = c4.w;
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float y = fmaf(p.alpha, acc[2 * t + h][b][r], cv[r]);
                                if (p.relu) y = relu_nan(y);
                                v[h * 4 + r] = __float_as_uint(y);
                            }
                        }
                        uint4 hi, lo;
                        Frag<f32h_t>::split8(make_uint4(v[0], v[1], v[2], v[3]), make_uint4(v[4], v[5], v[6], v[7]), hi, lo);
                        unsigned char* dst = ob + ((size_t)gm * p.ldo + gn0) * 4 + fchunk * 16;
                        *reinterpret_cast<uint4*>(dst) = hi;
                        *reinterpret_cast<uint4*>(dst + 64) = lo;
                    }
                }
                return;
            }
        }
        if constexpr (DT<TIN>::code == AGRL_F32H3 && sizeof(TOUT) == 4 && kLpF16) {
            if (p.out_planes) {
                // the seam to the plane kernels (conv1x1_duo.hip): the finished fp32 value leaves as fp16 planes [hi | lo 2^11 (| hi)] --
                // what agrl_split16_planes would make of the fp32 map, without the map (N % 4 == 0; no residual on this path)
                _Float16* ob = reinterpret_cast<_Float16*>(p.out);
                const size_t ldp = (size_t)p.out_planes * p.N;
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int gm = m0 + wm * (BM / WM) + b * 16 + frow;
                    if (gm >= p.M) continue;
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        const int gn = n0 + wn * (BN / 2) + a * 16 + fchunk * 4;
                        if (gn >= p.N) continue;
                        float cv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (p.colv) {
                            const float4 c4 = *reinterpret_cast<const float4*>(p.colv + gn);
                            cv[0] = c4.x; cv[1] = c4.y; cv[2] = c4.z; cv[3] = c4.w;
                        }
                        uint32_t h[2], l[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            float y0 = fmaf(p.alpha, acc[a][b][2 * e], cv[2 * e]), y1 = fmaf(p.alpha, acc[a][b][2 * e + 1], cv[2 * e + 1]);
                            if (p.relu) { y0 = relu_nan(y0); y1 = relu_nan(y1); }
                            h[e] = pack_lp16x2(y0, y1);
                            float a0, a1;
                            unpack_lp16x2(h[e], a0, a1);
                            l[e] = pack_lp16x2((y0 - a0) * 2048.f, (y1 - a1) * 2048.f);
                        }
                        _Float16* dst = ob + (size_t)gm * ldp + gn;
                        *reinterpret_cast<uint2*>(dst) = make_uint2(h[0], h[1]);
                        *reinterpret_cast<uint2*>(dst + p.N) = make_uint2(l[0], l[1]);
                        if (p.out_planes == 3) *reinterpret_cast<uint2*>(dst + 2 * (size_t)p.N) = make_uint2(h[0], h[1]);
                    }
                }
                return;
            }
        }
        if constexpr (sizeof(TIN) == 4 && sizeof(TOUT) == 4) {
            // fp32 result (+ fp32 residual) of the fp32-tensor convs (exact, split-bf16, split-fp16; measured on the split-fp16 ones) -- conv3 / conv3 + downsample of layers 1-2: two to twelve k-tiles, then a
            // result and a residual of 268-537 MB each. Straight from the accumulator layout a wave instruction touches 16 rows with 64
            // bytes each (half cache lines: with the ablation library the stores are 68 of 314 us in 64 -> 256 and, with the residual loads,
            // 94 of 185 in 128 -> 512, profiles/r06_split16_inloop_ablation.txt). So, as the 16-bit epilogue above: alpha acc + bias is parked
            // in the (now free) ring as a BM x BN fp32 image, 16-byte chunk c of row r at c ^ (r & (CPR - 1)); then every thread owns whole
            // chunks of whole rows -- residual load, ReLU, store: a wave covers 1 KiB of consecutive row bytes per instruction.
            constexpr int CPR32 = BN * 4 / 16, ROWB32 = BN * 4, RPT32 = 64 * NW / CPR32;
            static_assert(BM * ROWB32 <= NS * BUF_BYTES, "the fp32 out image fits the ring");
            // (a PRE-SPLIT result, conv1 / conv2 of a split-fp16 Bottleneck, was tried on the same road -- halves formed in the parking pass, whole
            // rows copied out -- and measured 25-30 us per step SLOWER than the register stores above, whose hi and lo chunks already complete
            // a 128-byte line per 16-lane group within two instructions: same-box A/B of tools/profile_layers.py, 10.31 / 10.27 against 10.28 / 10.24 ms)
            const bool full = p.vec_ok && !p.mix_f && !p.stats && !p.rowv && p.ksplit <= 1 && n0 + BN <= p.N && (p.ldo & 3) == 0 &&
                              ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.res)) & 15) == 0;   // workgroup-uniform
            if (full) {
                unsigned char* so = smem;
                const int pch = tid % CPR32, r0 = tid / CPR32;
                const float* __restrict__ resf = reinterpret_cast<const float*>(p.res);
                float* __restrict__ outf = reinterpret_cast<float*>(p.out);
                // the residual chunks this thread will add, requested now: their round trip runs under the parking of the tile and the barrier
                float4 rpf[BM / RPT32];
                const bool has_res = resf != nullptr && !(AGRL_DBG_BITS(p) & 2);
                if (has_res) {
#pragma unroll
                    for (int i = 0; i < BM / RPT32; ++i) {
                        const int row = r0 + i * RPT32;
                        const int gm = m0 + row < p.M ? m0 + row : p.M - 1;
                        rpf[i] = *reinterpret_cast<const float4*>(resf + (size_t)gm * p.ldo + n0 + ((pch ^ (row & (CPR32 - 1))) << 2));
                    }
                }
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int prow = wm * (BM / WM) + b * 16 + frow;
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        const int c = wn * (BN / 2) + a * 16 + fchunk * 4;
                        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.colv) cv = *reinterpret_cast<const float4*>(p.colv + n0 + c);
                        const float4 v = make_float4(fmaf(p.alpha, acc[a][b][0], p.rowc + cv.x), fmaf(p.alpha, acc[a][b][1], p.rowc + cv.y),
                                                     fmaf(p.alpha, acc[a][b][2], p.rowc + cv.z), fmaf(p.alpha, acc[a][b][3], p.rowc + cv.w));
                        *reinterpret_cast<float4*>(so + prow * ROWB32 + (((c >> 2) ^ (prow & (CPR32 - 1))) << 4)) = v;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < BM / RPT32; ++i) {
                    const int row = r0 + i * RPT32;
                    const int gm = m0 + row;
                    if (gm >= p.M) continue;
                    const int gch = pch ^ (row & (CPR32 - 1));
                    const size_t o = (size_t)gm * p.ldo + n0 + gch * 4;
                    float4 v = *reinterpret_cast<const float4*>(so + row * ROWB32 + (pch << 4));
                    if (has_res) { v.x += rpf[i].x; v.y += rpf[i].y; v.z += rpf[i].z; v.w += rpf[i].w; }
                    if (p.relu) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
                    if (!(AGRL_DBG_BITS(p) & 1)) *reinterpret_cast<float4*>(outf + o) = v;
                }
                return;
            }
        }
        const bool vec_ok = p.vec_ok != 0;
        const bool do_stats = p.stats != nullptr;   // workgroup-uniform
        float st1[FN][4], st2[FN][4];
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st1[a][r] = 0.f; st2[a][r] = 0.f; }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int gm = m0 + wm * (BM / WM) + b * 16 + frow;
            if (gm >= p.M) continue;
            const float rv = p.rowv ? p.rowv[gm] : p.rowc;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int gn = n0 + wn * (BN / 2) + a * 16 + fchunk * 4;
                if (gn >= p.N) continue;
                const size_t o = (size_t)gm * p.ldo + gn;
                float v[4];
                if (vec_ok && gn + 3 < p.N) {
                    float cv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.colv) {
                        const float4 c4 = *reinterpret_cast<const float4*>(p.colv + gn);
                        cv[0] = c4.x; cv[1] = c4.y; cv[2] = c4.z; cv[3] = c4.w;
                    }
                    if (p.mix_f) {   // GraphLayer: BatchNorm1d + LeakyReLU + residual mix on (G f) W^T (workgroup-uniform branch)
                        const float4 s4 = *reinterpret_cast<const float4*>(p.mix_scale + gn);
                        const float4 f4 = *reinterpret_cast<const float4*>(p.mix_f + o);
                        const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, fv[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float y = fmaf(acc[a][b][r], sv[r], cv[r]);
                            y = y > 0.f ? y : p.mix_slope * y;
                            v[r] = p.mix_keep * fv[r] + p.mix_gamma * y;
                        }
                    } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(p.alpha, acc[a][b][r], rv + cv[r]);
                    }
                    if (resp && !(AGRL_DBG_BITS(p) & 2)) {
                        float rr[4];
                        load4<TOUT>(resp + o, rr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rr[r];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = relu_nan(v[r]);
                    }
                    if (!(AGRL_DBG_BITS(p) & 1)) store4<TOUT>(outp + o, v);
                    if (do_stats) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { st1[a][r] += v[r]; st2[a][r] = fmaf(v[r], v[r], st2[a][r]); }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (gn + r < p.N) {
                            float t = fmaf(p.alpha, acc[a][b][r], rv + (p.colv ? p.colv[gn + r] : 0.f));
                            if (resp) t += DT<TOUT>::ld(resp + o + r);
                            if (p.relu) t = relu_nan(t);
                            DT<TOUT>::st(outp + o + r, t);
                        }
                    }
                }
            }
        }
        if (do_stats) {
            // per-channel sums of this tile's rows: over the 16 pixel lanes of a fragment by shuffles, over the WM wave rows
            // through the (now idle) staging buffer in wave order, one partial row per 64-row granule of M
            float* sred = reinterpret_cast<float*>(smem);   // [WM][BN][2]
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float u1 = st1[a][r], u2 = st2[a][r];
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) {
                        u1 += __shfl_xor(u1, off);
                        u2 += __shfl_xor(u2, off);
                    }
                    if (frow == 0) {
                        const int col = wn * (BN / 2) + a * 16 + fchunk * 4 + r;
                        sred[(wm * BN + col) * 2 + 0] = u1;
                        sred[(wm * BN + col) * 2 + 1] = u2;
                    }
                }
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                float u1 = 0.f, u2 = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < WM; ++w_) {
                    u1 += sred[(w_ * BN + tid) * 2 + 0];
                    u2 += sred[(w_ * BN + tid) * 2 + 1];
                }
                float* dst = p.stats + (size_t)(m0 >> 6) * 2 * p.N + n0 + tid;
                dst[0] = u1;
                dst[p.N] = u2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form (bf16 output, LDS epilogue). Same 4 waves / 2 x 2 MFMA layout, but each workgroup walks tiles
// T = vid, vid+G, ... and the waves take fixed side roles:
//   waves 0,1 ("loaders")  issue every LDS-DMA (k-tiles, residual, bias) and are the only ones that wait on vmcnt
//   waves 2,3 ("drainers") issue every global store of the finished tiles and never wait on them
// On gfx950 loads and stores retire in order on ONE per-wave counter, so a wave that stores and then waits for a
// DMA stalls until its stores have drained to HBM; with the roles split the loaders' counter only ever holds DMA
// and the store drain of tile T overlaps the whole of tile T+G. The first k-tile of tile T+G is issued while
// tile T is still in its epilogue, so no tile starts with an exposed L2/HBM round trip either.
// Barriers per tile: nk (ring) + 2 (residual landed / out tile complete); every wave executes every one.

template <typename TIN, int BM, int BN, int NW, bool POOL>  // NW waves, (NW/2) x 2 MFMA grid; POOL: fused frame pooling
__global__ __launch_bounds__(64 * NW) void igemm_persist_kernel(const IgemmParams p, int ntiles) {
    constexpr int NL = NW;
    constexpr int WM = NW / 2;
    constexpr int EPC = DT<TIN>::epc;
    constexpr int BKE = 8 * EPC;
    constexpr int AJ = BM / (8 * NL), BJ = BN / (8 * NL);  // 8-row DMA pieces per loader wave
    constexpr int FM = BM / (16 * WM), FN = BN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int CPR = BN * 2 / 16, ROWB = BN * 2, RPI = 64 / CPR;
    constexpr int RJ = BM / (NL * RPI);         // residual DMA pieces per loader wave
    static_assert(BM * ROWB <= BUF_BYTES, "out tile must fit one ring slot");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF_BYTES + 1024];
    unsigned char* s_bias = smem + 2 * BUF_BYTES;  // BN floats (<= 512 B), one DMA piece

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < NL;
    const bool drainer = true;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int nNt = (p.N + BN - 1) / BN;
    const int nk = p.K / BKE;
    const int G = gridDim.x;
    int vid = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7;
        const int xcd = vid & 7, within = vid >> 3;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int slot_step = (nk - 1) & 1;  // k-tile 0 of the next tile lands in the slot of this tile's last k-tile

    const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* __restrict__ wg = reinterpret_cast<const unsigned char*>(p.w);
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);
    const int lrow = lane >> 3;
    const int lchk = lane & 7;

    // ---- loader state. This kernel serves the pointwise case only (1x1, stride 1, no padding: 36 of the 52 convs,
    // the Linear and the distance matrix), where pixel m of the GEMM is row m of the input matrix and no
    // (n, oh, ow) decomposition has to be carried: the staging "coordinates" are just the tile origin.
    int sm0 = 0, sn0 = 0;
    auto make_coord = [&](int T) {
        const int mt = T / nNt, nt = T - mt * nNt;
        sm0 = mt * BM;
        sn0 = nt * BN;
    };
    const size_t a_row_bytes = (size_t)p.Cin * sizeof(TIN);
    unsigned kbyte = 0;  // byte offset of the k-tile being staged inside a row of X / W (K == Cin here)
    auto stage = [&](int buf) {  // loaders only
        unsigned char* sa = smem + buf * BUF_BYTES + wave * (BM / NL) * 128;
        unsigned char* sb = smem + buf * BUF_BYTES + A_BYTES + wave * (BN / NL) * 128;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int row = wave * (BM / NL) + j * 8 + lrow;
            const int gm = sm0 + row;
            const unsigned char* src = xg + (size_t)gm * a_row_bytes + kbyte + (lchk ^ ((row >> 1) & 7)) * 16;
            dma16(gm < p.M ? src : zsrc, sa + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = wave * (BN / NL) + j * 8 + lrow;
            int gn = sn0 + row;
            gn = gn < p.N ? gn : p.N - 1;
            dma16(wg + (size_t)gn * a_row_bytes + kbyte + (lchk ^ ((row >> 1) & 7)) * 16, sb + j * 1024);
        }
        kbyte += 128;
    };
    const bool has_res = p.res != nullptr;
    auto stage_residual_bias = [&](int buf, int cm0, int cn0) {  // loaders only, for the tile being computed
        unsigned char* so = smem + buf * BUF_BYTES;
        const unsigned char* rg = reinterpret_cast<const unsigned char*>(p.res);
        if (has_res) {
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int row0 = (wave * RJ + j) * RPI;
                const int row = row0 + lane / CPR;
                const int gch = (lane % CPR) ^ (row & (CPR - 1));
                const int gm = cm0 + row;
                const int gn = cn0 + gch * 8;
                const bool ok = gm < p.M && gn < p.N;
                dma16(ok ? rg + ((size_t)gm * p.ldo + gn) * 2 : zsrc, so + row0 * ROWB);
            }
        }
        if (wave == 0) {  // BN floats of bias: lanes 0 .. BN/4-1
            const int gn = cn0 + lane * 4;
            const bool ok = p.colv != nullptr && lane < BN / 4 && gn < p.N;
            dma16(ok ? reinterpret_cast<const unsigned char*>(p.colv + gn) : zsrc, s_bias);
        }
    };

    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    int s0 = 0;  // ring slot holding k-tile 0 of the current tile
    if (vid < ntiles) {
        make_coord(vid);
        if (loader) stage(0);
    }
    for (int T = vid; T < ntiles; T += G) {
        const int cm0 = sm0, cn0 = sn0;  // the tile being computed / written (sm0/sn0 move on to the next one)
        const bool has_next = T + G < ntiles;
        if (nk == 1 && has_next) {  // this tile's only k-tile is already in flight: retarget the staging coordinates
            make_coord(T + G);
            kbyte = 0;
        }

        f32x4_t acc[FN][FM];
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        int cur = s0;
        for (int kt = 0; kt < nk; ++kt) {
            if (loader) wait_vmcnt<0>();  // the loaders' queue holds nothing but DMA: k-tile kt has landed
            wg_barrier();                 // ... for everybody; and the other slot is no longer being read
            if (kt + 1 < nk) {
                if (loader) stage(cur ^ 1);
                if (kt + 2 == nk && has_next) {  // last k-tile issued: retarget the staging coordinates
                    make_coord(T + G);
                    kbyte = 0;
                }
            } else if (loader) {
                stage_residual_bias(cur ^ 1, cm0, cn0);
            }
            const unsigned char* sa = smem + cur * BUF_BYTES;
            const unsigned char* sb = sa + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 xf[FM], wf[FN];
#pragma unroll
                for (int b = 0; b < FM; ++b)
                    xf[b] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * (BM / WM) + b * 16 + frow, kk * 4 + fchunk));
#pragma unroll
                for (int a = 0; a < FN; ++a)
                    wf[a] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fchunk));
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b)
                        if (!(AGRL_DBG_BITS(p) & 2)) acc[a][b] = Frag<TIN>::mma(wf[a], xf[b], acc[a][b]);
            }
            cur ^= 1;
        }
        // cur = slot with the residual image (becomes the out tile); cur ^ 1 = slot of the last k-tile
        if (loader) wait_vmcnt<0>();
        wg_barrier();
        if (loader && has_next) stage(cur ^ 1);  // next tile's first k-tile flies under this tile's epilogue

        unsigned char* so = smem + cur * BUF_BYTES;
        if (!(AGRL_DBG_BITS(p) & 4)) {
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int prow = wm * (BM / WM) + b * 16 + frow;
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int c = wn * (BN / 2) + a * 16 + fchunk * 4;
                    const float4 cv = *reinterpret_cast<const float4*>(s_bias + c * 4);
                    unsigned char* slot = so + prow * ROWB + (((c >> 3) ^ (prow & (CPR - 1))) << 4) + ((c & 4) << 1);
                    float v[4];
                    v[0] = fmaf(p.alpha, acc[a][b][0], p.rowc + cv.x);
                    v[1] = fmaf(p.alpha, acc[a][b][1], p.rowc + cv.y);
                    v[2] = fmaf(p.alpha, acc[a][b][2], p.rowc + cv.z);
                    v[3] = fmaf(p.alpha, acc[a][b][3], p.rowc + cv.w);
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
        }
        wg_barrier();  // out tile complete
        if (drainer && (!POOL || p.pool_store_out)) {
            // drain: whole 16-byte chunks, full rows
            constexpr int ND = 64 * NW;  // drainer lanes
            const int dt = tid;
            const int pch = dt % CPR;
            const int r0 = dt / CPR;
            constexpr int RPP = ND / CPR;  // rows per pass
#pragma unroll
            for (int i = 0; i < BM / RPP; ++i) {
                const int row = r0 + i * RPP;
                const int gch = pch ^ (row & (CPR - 1));
                const int gm = cm0 + row;
                const int gn = cn0 + gch * 8;
                if (gm < p.M && gn < p.N && !(AGRL_DBG_BITS(p) & 1)) {
                    const uint4 v = *reinterpret_cast<const uint4*>(so + row * ROWB + (pch << 4));
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + ((size_t)gm * p.ldo + gn) * 2) = v;
                }
            }
        }
        if constexpr (POOL) {
            // fused part / global pooling of the finished frame (vmgn.py:298-308), two LDS stages:
            //  A: every thread sums 4 consecutive pixel rows of one 8-channel group        (512 threads, 128 rows)
            //  B: thread (group, bin) adds the 4-row partials that fall into its row bin   (16 x nparts threads)
            const int gch = tid % CPR;
            const int rg = tid / CPR;  // 0..31
            float part8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rg * 4 + i;
                const uint4 v = *reinterpret_cast<const uint4*>(so + row * ROWB + ((gch ^ (row & (CPR - 1))) << 4));
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo, hi;
                    unpack_lp16x2(w4[j], lo, hi);
                    part8[2 * j] += lo;
                    part8[2 * j + 1] += hi;
                }
            }
            wg_barrier();  // everybody holds its partial: the tile image may be overwritten
            float* sp = reinterpret_cast<float*>(so);  // [32 row groups][CPR][8] fp32 = 16 KB (BN = 128)
            *reinterpret_cast<float4*>(sp + (rg * CPR + gch) * 8) = make_float4(part8[0], part8[1], part8[2], part8[3]);
            *reinterpret_cast<float4*>(sp + (rg * CPR + gch) * 8 + 4) = make_float4(part8[4], part8[5], part8[6], part8[7]);
            wg_barrier();
            const int q = rg;
            if (q < p.pool_nparts && cn0 + gch * 8 < p.N) {
                const int g_lo = p.pool_start[q] * p.pool_w, g_hi = p.pool_end[q] * p.pool_w;  // pixel rows
                float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int g = 0; g < 32; ++g) {
                    const bool in = g * 4 >= g_lo && g * 4 < g_hi;  // bins are multiples of 4 pixels (checked by the host)
                    const float4 lo = *reinterpret_cast<const float4*>(sp + (g * CPR + gch) * 8);
                    const float4 hi = *reinterpret_cast<const float4*>(sp + (g * CPR + gch) * 8 + 4);
                    acc8[0] += in ? lo.x : 0.f; acc8[1] += in ? lo.y : 0.f; acc8[2] += in ? lo.z : 0.f; acc8[3] += in ? lo.w : 0.f;
                    acc8[4] += in ? hi.x : 0.f; acc8[5] += in ? hi.y : 0.f; acc8[6] += in ? hi.z : 0.f; acc8[7] += in ? hi.w : 0.f;
                }
                const float sc = p.pool_mean ? 1.f / (float)(g_hi - g_lo) : 1.f;
                const size_t o = ((size_t)(cm0 / BM) * p.pool_nparts + q) * p.N + cn0 + gch * 8;
                float4 lo4 = make_float4(acc8[0] * sc, acc8[1] * sc, acc8[2] * sc, acc8[3] * sc);
                float4 hi4 = make_float4(acc8[4] * sc, acc8[5] * sc, acc8[6] * sc, acc8[7] * sc);
                *reinterpret_cast<float4*>(p.pool_out + o) = lo4;
                *reinterpret_cast<float4*>(p.pool_out + o + 4) = hi4;
                if (p.pool_out_lp) {
                    uint4 pk;
                    pk.x = pack_lp16x2(lo4.x, lo4.y);
                    pk.y = pack_lp16x2(lo4.z, lo4.w);
                    pk.z = pack_lp16x2(hi4.x, hi4.y);
                    pk.w = pack_lp16x2(hi4.z, hi4.w);
                    *reinterpret_cast<uint4*>(reinterpret_cast<lp16_t*>(p.pool_out_lp) + o) = pk;
                }
            }
        }
        s0 ^= slot_step;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input patch (bf16). The generic kernel above re-stages
// the pixel tile for each of the 9 taps (9 x 16 KB of LDS-DMA per 64 input channels) although the taps only shift
// the same pixels; issuing that DMA is what bounds it (35 % of its time). Here a workgroup owns a 16x8 block of
// output pixels (one whole frame in layers 3/4), stages the 18x10 halo patch of a 64-channel slab ONCE (23 KB, zero
// padded by DMA from a zero block) and reads every tap's MFMA fragments straight out of it at a shifted pixel
// address -- the same trick as the stem. Per slab: 23 + 9 x 16 DMA pieces instead of 9 x 32.
//   LDS: patch 2 x 23 KB (double-buffered over slabs) + weight ring 2 x BN x 128 B; out tile overlays the patches.
template <int BN>
__global__ __launch_bounds__(512) void conv3x3_patch_kernel(const IgemmParams p) {
    constexpr int NW = 8, WM = 4, BM = 128;
    constexpr int FM = BM / (16 * WM), FN = BN / 32;
    constexpr int PW = 10;                       // patch width in pixels (8 + 2)
    constexpr int PPIX = 18 * PW;                // 180 patch pixels
    constexpr int PPIECES = (PPIX + 7) / 8;      // 23 DMA pieces of 8 pixel rows
    constexpr int PATCH_BYTES = PPIECES * 1024;  // 23552
    constexpr int B_BYTES = BN * 128;
    constexpr int BJ = BN / 64;                  // weight pieces per wave per tap
    constexpr int PJ = (PPIECES + NW - 1) / NW;  // patch pieces per wave per slab (3)
    constexpr int CPR = BN * 2 / 16, ROWB = BN * 2;
    static_assert(BM * ROWB <= 2 * PATCH_BYTES, "out tile must fit over the patch buffers");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PATCH_BYTES + 2 * B_BYTES];
    unsigned char* s_b = smem + 2 * PATCH_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int lrow = lane >> 3, lchk = lane & 7;

    const int nNt = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, within = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int mt = bid / nNt;
    const int nt = bid - mt * nNt;
    const int n0 = nt * BN;
    const int tw = p.W >> 3, th = p.H >> 4;
    const int img = mt / (tw * th);
    const int trem = mt - img * (tw * th);
    const int oy0 = (trem / tw) << 4, ox0 = (trem % tw) << 3;

    const unsigned char* __restrict__ xg = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* __restrict__ wg = reinterpret_cast<const unsigned char*>(p.w);
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(&g_zero16);

    // ---- DMA coordinates
    unsigned poff[PJ];
    bool pok[PJ];
#pragma unroll
    for (int i = 0; i < PJ; ++i) {
        const int piece = wave + NW * i;
        const int row = piece * 8 + lrow;  // patch pixel index
        const int py = row / PW, px = row - py * PW;
        const int iy = oy0 + py - 1, ix = ox0 + px - 1;
        pok[i] = piece < PPIECES && row < PPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        poff[i] = (unsigned)((((size_t)img * p.H + iy) * p.W + ix) * p.Cin * 2 + ((lchk ^ ((row >> 1) & 7)) << 4));
    }
    unsigned boff[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = wave * (BN / NW) + j * 8 + lrow;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        boff[j] = (unsigned)(((size_t)gn * p.K) * 2 + ((lchk ^ ((row >> 1) & 7)) << 4));
    }
    auto stage_patch_piece = [&](int slab, int buf, int i) {
        const int piece = wave + NW * i;
        if (piece < PPIECES) dma16(pok[i] ? xg + poff[i] + slab * 128 : zsrc, smem + buf * PATCH_BYTES + piece * 1024);
    };
    auto stage_b = [&](int slab, int tap, int slot) {
        const unsigned koff = (unsigned)(tap * p.Cin + slab * 64) * 2;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            dma16(wg + boff[j] + koff, s_b + slot * B_BYTES + (wave * (BN / NW) + j * 8) * 128);
    };

    // bias of this lane's channels, fetched now so the latency hides under the main loop
    float cvr[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int gn = n0 + wn * (BN / 2) + a * 16 + (lane >> 4) * 4;
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.colv && gn < p.N) c4 = *reinterpret_cast<const float4*>(p.colv + gn);
        cvr[a][0] = c4.x; cvr[a][1] = c4.y; cvr[a][2] = c4.z; cvr[a][3] = c4.w;
    }

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchunk = lane >> 4;
    int pr0[FM];  // patch pixel of this lane's output pixel at tap (0,0)
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = wm * (BM / WM) + b * 16 + frow;
        pr0[b] = (m >> 3) * PW + (m & 7);
    }

    const int nslab = p.Cin >> 6;
    const int nk = 9 * nslab;
#pragma unroll
    for (int i = 0; i < PJ; ++i) stage_patch_piece(0, 0, i);
    stage_b(0, 0, 0);
    int slab = 0, tap = 0;
    for (int it = 0; it < nk; ++it) {
        wait_vmcnt<0>();
        wg_barrier();
        // next step's weights; next slab's patch, one piece per tap step so the DMA queue never bursts
        int ntap = tap + 1, nsl = slab;
        if (ntap == 9) { ntap = 0; ++nsl; }
        if (it + 1 < nk) stage_b(nsl, ntap, (it + 1) & 1);
        if (slab + 1 < nslab && tap < PJ) stage_patch_piece(slab + 1, (slab + 1) & 1, tap);
        const int tr = tap / 3, ts = tap - tr * 3;
        const unsigned char* sp = smem + (slab & 1) * PATCH_BYTES;
        const unsigned char* sb = s_b + (it & 1) * B_BYTES;
        const int shift = tr * PW + ts;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 xf[FM], wf[FN];
#pragma unroll
            for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const uint4*>(sp + lds_off(pr0[b] + shift, kk * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < FN; ++a)
                wf[a] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] = Frag<lp16_t>::mma(wf[a], xf[b], acc[a][b]);
        }
        tap = ntap;
        slab = nsl;
    }
    wait_vmcnt<0>();
    wg_barrier();  // all fragment reads done: the patch buffers become the out tile

    unsigned char* so = smem;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int prow = wm * (BM / WM) + b * 16 + frow;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int c = wn * (BN / 2) + a * 16 + fchunk * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[a][b][r] + cvr[a][r];
                if (p.relu) v[r] = relu_nan(v[r]);
            }
            store4<lp16_t>(reinterpret_cast<lp16_t*>(so + prow * ROWB + (((c >> 3) ^ (prow & (CPR - 1))) << 4) + ((c & 4) << 1)), v);
        }
    }
    wg_barrier();
    constexpr int RPT = 64 * NW / CPR;
    const int pch = tid % CPR;
    const int r0 = tid / CPR;
#pragma unroll
    for (int i = 0; i < BM / RPT; ++i) {
        const int row = r0 + i * RPT;
        const int gch = pch ^ (row & (CPR - 1));
        const int gn = n0 + gch * 8;
        if (gn < p.N) {
            const size_t gm = ((size_t)img * p.H + oy0 + (row >> 3)) * p.W + ox0 + (row & 7);
            const uint4 v = *reinterpret_cast<const uint4*>(so + row * ROWB + (pch << 4));
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.out) + (gm * p.ldo + gn) * 2) = v;
        }
    }
}

template <typename TIN, typename TOUT>
static int launch_igemm(const IgemmParams& p_in, hipStream_t stream, const char* who) {
    constexpr int BKE = 8 * DT<TIN>::epc;
    IgemmParams p = p_in;
    AGRL_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "%s: empty problem (M=%d N=%d K=%d)", who, p.M, p.N, p.K);
    AGRL_CHECK_ARG(p.Cin % BKE == 0, "%s: Cin=%d must be a multiple of %d for this dtype", who, p.Cin, BKE);
    AGRL_CHECK_ARG(p.K == p.R * p.S * p.Cin, "%s: K mismatch", who);
    AGRL_CHECK_ARG((((uintptr_t)p.x) & 15) == 0 && (((uintptr_t)p.w) & 15) == 0,
                   "%s: operands must be 16-byte aligned", who);
    const AgrlOpts& opt = agrl_opts();
    p.dbg = opt.igemm_dbg;
    if (p.ksplit < 1) p.ksplit = 1;
    p.vec_ok = (p.ldo & 3) == 0 && (((uintptr_t)p.out) & 15) == 0 && (!p.colv || (((uintptr_t)p.colv) & 15) == 0) &&
               (!p.res || (((uintptr_t)p.res) & 15) == 0);
    // The GraphLayer epilogue (mix_f: BatchNorm + LeakyReLU + residual mix) exists only in the vectorised fp32 register epilogue of
    // igemm_kernel: the scalar tail, the wide and the persistent kernels would silently compute a plain Linear + shift.
    AGRL_CHECK_ARG(!p.mix_f || (sizeof(TOUT) == 4 && p.vec_ok && (p.N % 4) == 0 && p.mix_scale && p.ksplit == 1 && p.pool_nparts == 0),
                   "%s: the GraphLayer epilogue needs fp32 output, N %% 4 == 0 (got %d), 16-byte aligned out / shift and no split-K", who, p.N);
    // bf16 outputs whose rows are whole 16-byte chunks take the LDS-staged (fully coalesced) epilogue
    const bool lds_epi = sizeof(TOUT) == 2 && (p.N % 8) == 0 && (p.ldo % 8) == 0 && (((uintptr_t)p.out) & 15) == 0 &&
                         (!p.res || (((uintptr_t)p.res) & 15) == 0) && (!p.colv || (((uintptr_t)p.colv) & 15) == 0);
    const bool narrow = p.N <= 64;
    // ring depth 2 at two workgroups per CU beats deeper rings at one (measured: 6.4 vs 9.1 ms per forward)
    // (a 3-slot ring at one workgroup per CU and 4-wave workgroups were A/B switches until round 4: both measured slower)
    // short K loops (<= 2 k-tiles) are pure load->store latency chains: 64-row tiles halve the LDS footprint so
    // three workgroups fit a CU
    int bm = (p.K / BKE) <= 2 ? 64 : 128;
    // (split-fp16 conv3 of layer 2 -- four k-tiles, residual -- through 64-row tiles with the early residual request: 184-188 us against 167-170)
    if (cdiv(p.M, 128) * cdiv(p.N, narrow ? 64 : 128) < 400) bm = 64;  // too few 128-row tiles to fill 256 CUs twice
    const int grid = cdiv(p.M, bm) * cdiv(p.N, narrow ? 64 : 128);
    // 8-wave workgroups (4 x 2 wave grid) beat 4-wave ones by 3-14 % at equal tile size (A/B measured)
#define LAUNCH_IG(BM_, BN_, EPI_, NS_) \
    hipLaunchKernelGGL((igemm_kernel<TIN, TOUT, BM_, BN_, EPI_, NS_, 8>), dim3(grid, p.ksplit > 1 ? p.ksplit : 1), dim3(512), 0, stream, p)
#define LAUNCH_NS(BN_, EPI_)                                  \
    do {                                                      \
        if (bm == 64) LAUNCH_IG(64, BN_, EPI_, 2);            \
        else LAUNCH_IG(128, BN_, EPI_, 2);                    \
    } while (0)
    bool done = false;
    if constexpr (sizeof(TIN) == 2 && sizeof(TOUT) == 2) {
        // MFMA-bound pointwise layers (long K, enough 256 x 256 tiles to cover the chip): the wide-tile kernel
        int wide = -1;
        if (agrl_opt_set(opt.igemm_wide)) wide = opt.igemm_wide;
        if (wide != 0 && lds_epi && igemm_wide_applicable(p)) {
            const int wtiles = cdiv(p.M, 256) * (p.N / 256);
            if (wide >= 1 || (wtiles >= 224 && p.K >= 256)) return launch_igemm_wide(p, stream, who);
            // N = 256 layers with long K (layer 3's 1024 -> 256): 256 x 128 tiles, one per CU
            if (p.pool_nparts == 0 && p.N >= 256 && cdiv(p.M, 256) * (p.N / 128) >= 224 && p.K >= 512) return launch_igemm_wide(p, stream, who);
            if (p.pool_nparts > 0 && wide != 0 && wtiles >= 64 && !opt.pool_persist) return launch_igemm_wide(p, stream, who);
        }
    }
    if constexpr (sizeof(TIN) == 2 && sizeof(TOUT) == 4) {
        // the full query x gallery distance matrix (BASELINE configs[4]): 256 x 256 tiles once they cover the chip
        if (opt.igemm_wide != 0 && !agrl_opts().distmat_tiled && igemm_wide_f32out_applicable(p) && cdiv(p.M, 256) * cdiv(p.N, 256) >= 160 && p.K >= 512)
            return launch_igemm_wide_f32out(p, stream, who);
    }
    if constexpr (sizeof(TOUT) == 2) {
        // persistent tiles pay off where a tile is short (<= 8 k-tiles): its first DMA round trip and its store
        // drain are a large share of the tile; long K loops run better as independent workgroups (A/B measured)
        int persist = (p.K / BKE) <= 8 ? 1 : 0;
        if (p.pool_nparts > 0) persist = 1;  // the fused pooling epilogue lives in the persistent kernel
        const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;
        if (lds_epi && persist && !p.rowv && pointwise) {
            const int ntiles = cdiv(p.M, 128) * cdiv(p.N, narrow ? 64 : 128);
            const int wgs = 512;  // two resident workgroups per CU
            const int g = ntiles < wgs ? ntiles : wgs;
            if (p.pool_nparts > 0) {  // (the pooling epilogue exists for the 128 x 128 instantiation only)
                AGRL_CHECK_ARG(!narrow, "%s: fused pooling needs more than 64 output channels", who);
                hipLaunchKernelGGL((igemm_persist_kernel<TIN, 128, 128, 8, true>), dim3(g), dim3(512), 0, stream, p, ntiles);
            } else {
                if (narrow) hipLaunchKernelGGL((igemm_persist_kernel<TIN, 128, 64, 8, false>), dim3(g), dim3(512), 0, stream, p, ntiles);
                else hipLaunchKernelGGL((igemm_persist_kernel<TIN, 128, 128, 8, false>), dim3(g), dim3(512), 0, stream, p, ntiles);
            }
            AGRL_CHECK_LAUNCH(who);
            return 0;
        }
    }
    if constexpr (sizeof(TOUT) == 2) {
        if (lds_epi) {
            if (narrow) LAUNCH_NS(64, true); else LAUNCH_NS(128, true);
            done = true;
        }
    }
    if (!done) {
        if constexpr (DT<TIN>::code == AGRL_F32H3) {
            // split-fp16: the matrix work of a k-tile is ~5 x shorter than in the exact mode, so the k-tile in flight behind the one being
            // multiplied no longer covers the HBM / L2 round trip: a three-slot ring where it keeps two workgroups on a CU (round 6)
            // Measured per launch (profiles/r06_split16_ring_depth.txt): the 64-channel tiles (72 KB of LDS: still two workgroups per CU)
            // gain 4-10 % -- 3x3 64 -> 64 248 -> 238 us, 256 -> 64 164 -> 149 --, the 128-channel tiles (96 KB: ONE workgroup per CU) lose
            // 20-45 % (3x3 128 -> 128 171 -> 219, 128 -> 512 180 -> 258): three slots for the narrow tiles only. AGRL_SPLIT16_NS = 2 / 3 forces it.
            const int ns = agrl_opt_set(agrl_opts().split16_ns) ? agrl_opts().split16_ns : (narrow ? 3 : 2);
            if (ns == 3) {
                if (narrow) { if (bm == 64) LAUNCH_IG(64, 64, false, 3); else LAUNCH_IG(128, 64, false, 3); }
                else { if (bm == 64) LAUNCH_IG(64, 128, false, 3); else LAUNCH_IG(128, 128, false, 3); }
                done = true;
            }   // (64-row x 128-channel tiles with three slots -- 72 KB, two workgroups per CU again -- measured 10-40 % slower than the
                // 128 x 128 two-slot tile on every layer-2 / layer-3 shape: not kept)
        }
    }
    if (!done) {
        if (narrow) LAUNCH_NS(64, false); else LAUNCH_NS(128, false);
    }
#undef LAUNCH_NS
#undef LAUNCH_IG
    AGRL_CHECK_LAUNCH(who);
    return 0;
}

// The weight operand of the split-fp16 in-loop kernels: (rows, K) fp32, already scaled by the caller's power of two, -> the same
// rows x K x 4 bytes with every 32-value k-tile stored as [hi of k 4c..4c+3 and 16+4c..16+4c+3 (8 fp16) for c = 0..3 | the low halves in
// the same order]: the two 16-byte chunks lane group c reads of a k-tile ARE its operands of v_mfma_f32_16x16x32_f16.
// hi = fp16(w) to nearest, lo = fp16(w - hi): the values Frag<f32h_t>::split8 would produce in the loop, produced once.
namespace {
__global__ void __launch_bounds__(256) split16_weights_inloop_kernel(const float4* __restrict__ w, uint4* __restrict__ out, long long groups) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
        const long long tile = g >> 2;
        const int c = (int)(g & 3);
        const float4 a = w[tile * 8 + c], b = w[tile * 8 + 4 + c];
        const uint4 c0 = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w));
        const uint4 c1 = make_uint4(__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w));
        uint4 hi, lo;
        Frag<f32h_t>::split8(c0, c1, hi, lo);
        out[tile * 8 + c] = hi;
        out[tile * 8 + 4 + c] = lo;
    }
}
}  // namespace

extern "C" int agrl_split16_weights_inloop(const float* w_scaled, void* out, long long rows, int K, agrl_stream_t stream) {
    AGRL_CHECK_ARG(w_scaled && out && rows > 0 && K > 0, "agrl_split16_weights_inloop: null pointer or empty shape");
    AGRL_CHECK_ARG(K % 32 == 0, "agrl_split16_weights_inloop: K=%d must be a multiple of 32 (one k-tile)", K);
    AGRL_CHECK_ARG((const void*)w_scaled != (const void*)out, "agrl_split16_weights_inloop: not in place");
    AGRL_CHECK_ARG((((uintptr_t)w_scaled | (uintptr_t)out) & 15) == 0, "agrl_split16_weights_inloop: pointers must be 16-byte aligned");
    const long long groups = rows * (K / 32) * 4;
    const int grid = (int)((groups + 255) / 256 < 8192 ? (groups + 255) / 256 : 8192);
    hipLaunchKernelGGL(split16_weights_inloop_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(w_scaled),
                       reinterpret_cast<uint4*>(out), groups);
    AGRL_CHECK_LAUNCH("agrl_split16_weights_inloop");
    return 0;
}

extern "C" int agrl_conv2d_bn_act_split16(const void* x, const void* w_scaled, const float* bias, const void* residual, void* out, int N,
                                          int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int relu, float w_unscale,
                                          int x_presplit, int out_presplit, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w_scaled && out, "agrl_conv2d_bn_act_split16: null pointer");
    AGRL_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0,
                   "agrl_conv2d_bn_act_split16: bad shape");
    AGRL_CHECK_ARG(Cin % 32 == 0, "agrl_conv2d_bn_act_split16: Cin=%d must be a multiple of 32 (the pre-split weight k-tile)", Cin);
    AGRL_CHECK_ARG(w_unscale > 0.f && w_unscale == w_unscale && w_unscale <= 3.4e38f, "agrl_conv2d_bn_act_split16: w_unscale must be a positive finite power of two");
    {
        int e = 0;
        AGRL_CHECK_ARG(frexpf(w_unscale, &e) == 0.5f, "agrl_conv2d_bn_act_split16: w_unscale=%g is not a power of two (the un-scaling must be exact)", (double)w_unscale);
    }
    IgemmParams p{};
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = x; p.w = w_scaled; p.colv = bias; p.rowv = nullptr; p.res = residual; p.out = out;
    AGRL_CHECK_ARG(!out_presplit || (!residual && Cout % 32 == 0), "agrl_conv2d_bn_act_split16: a pre-split output takes no residual and Cout %% 32 == 0 (got %d)", Cout);
    p.a_pre = x_presplit ? 1 : 0; p.out_pre = out_presplit ? 1 : 0;
    p.alpha = w_unscale; p.rowc = 0.f; p.relu = relu; p.ksplit = 1; p.pool_nparts = 0;
    p.OH = (H + 2 * pad - R) / stride + 1;
    p.OW = (W + 2 * pad - S) / stride + 1;
    AGRL_CHECK_ARG(p.OH > 0 && p.OW > 0, "agrl_conv2d_bn_act_split16: empty output");
    p.M = N * p.OH * p.OW; p.N = Cout; p.K = R * S * Cin;
    p.Cin = Cin; p.H = H; p.W = W; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.ldo = Cout;
    return launch_igemm<f32h_t, float>(p, (hipStream_t)stream, "agrl_conv2d_bn_act_split16");
}

// conv3 + the 1x1 stride-s downsample conv of a first Bottleneck as ONE split-fp16 GEMM over [x sampled at the stride | x2] (vmgn.py:56-64
// with both BatchNorms folded: w_scaled = [w_downsample | w_conv3] 2^k per output channel, bias = b_downsample + b_conv3). fp32 tensors:
// x (N, H, W, K1) the block input, x2 (N, OH, OW, K2) conv2's output, out (N, OH, OW, Cout); OH = (H - 1) / stride + 1. The
// conforming mode's counterpart of agrl_conv1x1_packed_dual_strided: the fp32 shortcut map (537 MB in layer 1) no longer exists.
extern "C" int agrl_conv1x1_dual_split16(const void* x, const void* x2, const void* w_scaled, const float* bias, void* out, int N, int H,
                                         int W, int stride, int K1, int K2, int Cout, int relu, float w_unscale, int x2_presplit,
                                         int out_planes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(out_planes == 0 || ((out_planes == 2 || out_planes == 3) && agrl_lp16_is_f16() && Cout % 4 == 0),
                   "agrl_conv1x1_dual_split16: out_planes = %d (0: fp32 map; 2 / 3: split-fp16 planes, fp16 build, Cout %% 4 == 0)", out_planes);
    AGRL_CHECK_ARG(x && x2 && w_scaled && out, "agrl_conv1x1_dual_split16: null pointer");
    AGRL_CHECK_ARG(N > 0 && H > 0 && W > 0 && stride >= 1 && K1 > 0 && K2 > 0 && Cout > 0, "agrl_conv1x1_dual_split16: bad shape");
    AGRL_CHECK_ARG(K1 % 32 == 0 && K2 % 32 == 0, "agrl_conv1x1_dual_split16: K1 and K2 must be multiples of 32 (got %d, %d)", K1, K2);
    AGRL_CHECK_ARG((((uintptr_t)x2) & 15) == 0, "agrl_conv1x1_dual_split16: x2 must be 16-byte aligned");
    AGRL_CHECK_ARG(w_unscale > 0.f && w_unscale <= 3.4e38f, "agrl_conv1x1_dual_split16: w_unscale must be a positive finite power of two");
    {
        int e = 0;
        AGRL_CHECK_ARG(frexpf(w_unscale, &e) == 0.5f, "agrl_conv1x1_dual_split16: w_unscale=%g is not a power of two", (double)w_unscale);
    }
    IgemmParams p{};
    p.x = x; p.x2 = x2; p.K1 = K1; p.stats = nullptr; p.a_pre = x2_presplit ? 2 : 0; p.out_planes = out_planes;
    p.w = w_scaled; p.colv = bias; p.rowv = nullptr; p.res = nullptr; p.out = out;
    p.alpha = w_unscale; p.rowc = 0.f; p.relu = relu; p.ksplit = 1; p.pool_nparts = 0;
    p.OH = (H - 1) / stride + 1;
    p.OW = (W - 1) / stride + 1;
    p.M = N * p.OH * p.OW; p.N = Cout; p.K = K1 + K2;
    p.Cin = K1 + K2; p.H = H; p.W = W; p.R = 1; p.S = 1; p.stride = stride; p.pad = 0;
    p.ldo = Cout;
    return launch_igemm<f32h_t, float>(p, (hipStream_t)stream, "agrl_conv1x1_dual_split16");
}

extern "C" int agrl_conv2d_bn_act(const void* x, const void* w, const float* bias, const void* residual,
                                  void* out, int N, int H, int W, int Cin, int Cout, int R, int S,
                                  int stride, int pad, int relu, int dtype, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w && out, "agrl_conv2d_bn_act: null pointer");
    AGRL_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "agrl_conv2d_bn_act: bad shape");
    AGRL_CHECK_ARG(R > 0 && S > 0 && stride > 0 && pad >= 0, "agrl_conv2d_bn_act: bad filter geometry");
    AGRL_CHECK_ARG(dtype == AGRL_F32 || dtype == AGRL_LP16 || dtype == AGRL_F32X3, "agrl_conv2d_bn_act: bad dtype %d", dtype);
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = x; p.w = w; p.colv = bias; p.rowv = nullptr; p.res = residual; p.out = out;
    p.alpha = 1.f; p.rowc = 0.f; p.relu = relu; p.ksplit = 1; p.pool_nparts = 0;
    p.OH = (H + 2 * pad - R) / stride + 1;
    p.OW = (W + 2 * pad - S) / stride + 1;
    AGRL_CHECK_ARG(p.OH > 0 && p.OW > 0, "agrl_conv2d_bn_act: empty output");
    p.M = N * p.OH * p.OW; p.N = Cout; p.K = R * S * Cin;
    p.Cin = Cin; p.H = H; p.W = W; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.ldo = Cout;
    if (dtype == AGRL_F32) return launch_igemm<float, float>(p, (hipStream_t)stream, "agrl_conv2d_bn_act");
    if (dtype == AGRL_F32X3) return launch_igemm<f32s_t, float>(p, (hipStream_t)stream, "agrl_conv2d_bn_act");
    const bool patch_ok = R == 3 && S == 3 && stride == 1 && pad == 1 && !residual && (H % 16) == 0 && (W % 8) == 0 &&
                          (Cin % 64) == 0 && (Cout % 8) == 0 && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) == 0 &&
                          (!bias || (((uintptr_t)bias) & 15) == 0) && (size_t)N * H * W * Cin * 2 < (1ull << 32);
    if (patch_ok) {
        p.dbg = 0; p.vec_ok = 1;
        const int tiles = N * (H / 16) * (W / 8);
        int wide3 = -1;
        if (agrl_opt_set(agrl_opts().conv3x3_wide)) wide3 = agrl_opts().conv3x3_wide;
        // two pixel blocks per workgroup for the deep layers (3/4: measured +3..4 %; at Cin = 128 the one-block kernel
        // at two workgroups per CU is 5 % faster), where that still leaves >= one workgroup per CU
        if (Cout > 64 && wide3 != 0 && (wide3 == 1 || (Cin >= 256 && cdiv(tiles, 2) * cdiv(Cout, 128) >= 256))) {
            p.ksplit = 1;
            return launch_conv3x3_wide(p, (hipStream_t)stream);
        }
        int c64 = -1;
        if (agrl_opt_set(agrl_opts().conv3x3_c64)) c64 = agrl_opts().conv3x3_c64;
        if (Cin == 64 && Cout == 64 && p.ldo == 64 && c64 != 0 && (c64 == 1 || tiles >= 256)) {
            p.ksplit = 1;
            return launch_conv3x3_c64(p, (hipStream_t)stream);  // layer 1: weights resident, persistent over pixel blocks
        }
        if (Cout <= 64)
            hipLaunchKernelGGL(conv3x3_patch_kernel<64>, dim3(tiles * cdiv(Cout, 64)), dim3(512), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(conv3x3_patch_kernel<128>, dim3(tiles * cdiv(Cout, 128)), dim3(512), 0, (hipStream_t)stream, p);
        AGRL_CHECK_LAUNCH("agrl_conv2d_bn_act(3x3 patch)");
        return 0;
    }
    return launch_igemm<lp16_t, lp16_t>(p, (hipStream_t)stream, "agrl_conv2d_bn_act");
}

// conv without bias / residual / activation in fp32 (exact or split-bf16) whose epilogue also leaves the per-channel sum and sum
// of squares of every finished tile's rows in ``partial`` ([ceil(M / 64)][2][Cout] floats, zeroed here): the batch statistics
// of the train-mode BatchNorm behind the conv without a second pass over its output (agrl_bn_stats_from_partials finishes them)
extern "C" int agrl_conv2d_stats(const float* x, const float* w, float* out, float* partial, size_t partial_bytes, int N, int H, int W,
                                 int Cin, int Cout, int R, int S, int stride, int pad, int dtype, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w && out && partial, "agrl_conv2d_stats: null pointer");
    AGRL_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "agrl_conv2d_stats: bad shape");
    AGRL_CHECK_ARG(dtype == AGRL_F32 || dtype == AGRL_F32X3, "agrl_conv2d_stats: dtype must be fp32 (0) or split-bf16 fp32 (2), got %d", dtype);
    AGRL_CHECK_ARG((Cout % 4) == 0 && (((uintptr_t)out | (uintptr_t)partial) & 15) == 0, "agrl_conv2d_stats: Cout %% 4 == 0, out / partial 16-byte aligned");
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = partial;
    p.x = x; p.w = w; p.colv = nullptr; p.rowv = nullptr; p.res = nullptr; p.out = out;
    p.alpha = 1.f; p.rowc = 0.f; p.relu = 0; p.ksplit = 1; p.pool_nparts = 0;
    p.OH = (H + 2 * pad - R) / stride + 1;
    p.OW = (W + 2 * pad - S) / stride + 1;
    AGRL_CHECK_ARG(p.OH > 0 && p.OW > 0, "agrl_conv2d_stats: empty output");
    p.M = N * p.OH * p.OW; p.N = Cout; p.K = R * S * Cin;
    p.Cin = Cin; p.H = H; p.W = W; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.ldo = Cout;
    const size_t need = (size_t)cdiv(p.M, 64) * 2 * Cout * sizeof(float);
    AGRL_CHECK_ARG(partial_bytes >= need, "agrl_conv2d_stats: partial buffer too small (ceil(M / 64) * 2 * Cout floats)");
    if (hipMemsetAsync(partial, 0, need, (hipStream_t)stream) != hipSuccess) {   // 128-row tiles fill every other granule row
        agrl_set_error("agrl_conv2d_stats: hipMemsetAsync failed");
        return 1;
    }
    if (dtype == AGRL_F32) return launch_igemm<float, float>(p, (hipStream_t)stream, "agrl_conv2d_stats");
    return launch_igemm<f32s_t, float>(p, (hipStream_t)stream, "agrl_conv2d_stats");
}

extern "C" int agrl_conv1x1_dual_bn_act(const void* x1, const void* x2, const void* w, const float* bias, void* out, int M,
                                       int K1, int K2, int Cout, int relu, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x1 && x2 && w && out, "agrl_conv1x1_dual_bn_act: null pointer");
    AGRL_CHECK_ARG(M > 0 && K1 > 0 && K2 > 0 && Cout > 0, "agrl_conv1x1_dual_bn_act: bad shape");
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = x1; p.x2 = x2; p.K1 = K1; p.w = w; p.colv = bias; p.rowv = nullptr; p.res = nullptr; p.out = out;
    p.alpha = 1.f; p.rowc = 0.f; p.relu = relu; p.ksplit = 1; p.pool_nparts = 0; p.dbg = 0; p.vec_ok = 1;
    p.M = M; p.N = Cout; p.K = K1 + K2;
    p.Cin = p.K; p.H = 1; p.W = 1; p.OH = 1; p.OW = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.ldo = Cout;
    AGRL_CHECK_ARG(igemm_wide_applicable(p),
                   "agrl_conv1x1_dual_bn_act: needs K1 == 2 K2, K1 %% 64 == 0, Cout %% 256 == 0 and 16-byte aligned operands "
                   "(got K1=%d K2=%d Cout=%d)", K1, K2, Cout);
    return launch_igemm_wide(p, (hipStream_t)stream, "agrl_conv1x1_dual_bn_act");
}

extern "C" int agrl_conv1x1_bn_act_pool(const void* x, const void* w, const float* bias, const void* residual, void* out,
                                       float* pool_out, void* pool_out_lp, int N, int H, int W, int Cin, int Cout,
                                       int relu, const int* splits, int n_splits, int mean, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w && pool_out && splits, "agrl_conv1x1_bn_act_pool: null pointer");
    AGRL_CHECK_ARG(H * W == 128, "agrl_conv1x1_bn_act_pool: a frame must be exactly 128 pixels (got %dx%d)", H, W);
    AGRL_CHECK_ARG(Cout > 64 && Cout % 8 == 0 && Cin % 64 == 0, "agrl_conv1x1_bn_act_pool: unsupported channel counts");
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = x; p.w = w; p.colv = bias; p.rowv = nullptr; p.res = residual; p.out = out;
    p.alpha = 1.f; p.rowc = 0.f; p.relu = relu; p.ksplit = 1;
    p.OH = H; p.OW = W;
    p.M = N * H * W; p.N = Cout; p.K = Cin;
    p.Cin = Cin; p.H = H; p.W = W; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.ldo = Cout;
    int P = 0;
    for (int i = 0; i < n_splits; ++i) {
        const int n = splits[i];
        AGRL_CHECK_ARG(n > 0 && P + n <= 16, "agrl_conv1x1_bn_act_pool: at most 16 bins");
        for (int j = 0; j < n; ++j) {  // AdaptiveAvgPool2d bins over image rows
            p.pool_start[P] = (j * H) / n;
            p.pool_end[P] = ((j + 1) * H + n - 1) / n;
            ++P;
        }
    }
    for (int i = P; i < 16; ++i) p.pool_start[i] = p.pool_end[i] = 0;
    p.pool_nparts = P; p.pool_mean = mean; p.pool_store_out = out != nullptr; p.pool_w = W;
    p.pool_out = pool_out; p.pool_out_lp = pool_out_lp;
    if (!out) p.out = pool_out;  // never dereferenced as activations; keeps the alignment checks meaningful
    return launch_igemm<lp16_t, lp16_t>(p, (hipStream_t)stream, "agrl_conv1x1_bn_act_pool");
}

extern "C" int agrl_linear_nobias(const void* x, const void* w, float* y, int M, int K, int Nout,
                                  int in_dtype, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w && y, "agrl_linear_nobias: null pointer");
    AGRL_CHECK_ARG(in_dtype == AGRL_F32 || in_dtype == AGRL_LP16 || in_dtype == AGRL_F32X3, "agrl_linear_nobias: bad dtype %d", in_dtype);
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = x; p.w = w; p.colv = nullptr; p.rowv = nullptr; p.res = nullptr; p.out = y;
    p.alpha = 1.f; p.rowc = 0.f; p.relu = 0; p.ksplit = 1; p.pool_nparts = 0;
    p.M = M; p.N = Nout; p.K = K;
    p.Cin = K; p.H = 1; p.W = 1; p.OH = 1; p.OW = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.ldo = Nout;
    if (in_dtype == AGRL_F32) return launch_igemm<float, float>(p, (hipStream_t)stream, "agrl_linear_nobias");
    if (in_dtype == AGRL_F32X3) return launch_igemm<f32s_t, float>(p, (hipStream_t)stream, "agrl_linear_nobias");
    return launch_igemm<lp16_t, float>(p, (hipStream_t)stream, "agrl_linear_nobias");
}

// GraphLayer, second half, as ONE GEMM with a fused epilogue (vmgn.py:148, :168-172):
//     out = keep * f + gamma * LeakyReLU( BN( (G f) W^T ) )
// G (f W^T) = (G f) W^T: with the message pass applied to the layer INPUT (agrl_graph_apply: P = G f, written once in the GEMM's
// operand dtype) the Linear's output h never exists, and BatchNorm1d (folded scale / shift), LeakyReLU and the residual mix ride
// in the GEMM's register epilogue. p_op (M, K) in dtype, w (N, K) in dtype, f / out (M, N) fp32.
extern "C" int agrl_graph_linear_mix(const void* p_op, const void* w, const float* f, const float* bn_scale, const float* bn_shift,
                                     float keep, float gamma, float slope, float* out, int M, int K, int Nout, int in_dtype,
                                     agrl_stream_t stream) {
    AGRL_CHECK_ARG(p_op && w && f && bn_scale && bn_shift && out, "agrl_graph_linear_mix: null pointer");
    AGRL_CHECK_ARG(in_dtype == AGRL_F32 || in_dtype == AGRL_LP16 || in_dtype == AGRL_F32X3 || in_dtype == AGRL_F32H3 || in_dtype == AGRL_F32H3P,
                   "agrl_graph_linear_mix: bad dtype %d", in_dtype);
    const bool p_presplit = in_dtype == AGRL_F32H3P;
    if (p_presplit) in_dtype = AGRL_F32H3;
    AGRL_CHECK_ARG((Nout % 4) == 0 && ((((uintptr_t)f | (uintptr_t)out | (uintptr_t)bn_scale | (uintptr_t)bn_shift) & 15) == 0),
                   "agrl_graph_linear_mix: Nout %% 4 == 0 and 16-byte aligned f / out / scale / shift required");
    // 16-bit operands: the kernel shaped for this problem (graph_gemm.hip) where it applies
    if (in_dtype == AGRL_LP16 && graph_gemm_applicable(M, K, Nout))
        return launch_graph_gemm(p_op, w, f, bn_scale, bn_shift, keep, gamma, slope, out, M, K, Nout, (hipStream_t)stream);
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = p_op; p.w = w; p.colv = bn_shift; p.rowv = nullptr; p.res = nullptr; p.out = out;
    p.alpha = 1.f; p.rowc = 0.f; p.relu = 0; p.ksplit = 1; p.pool_nparts = 0;
    p.M = M; p.N = Nout; p.K = K;
    p.Cin = K; p.H = 1; p.W = 1; p.OH = 1; p.OW = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.ldo = Nout;
    p.mix_f = f; p.mix_scale = bn_scale; p.mix_keep = keep; p.mix_gamma = gamma; p.mix_slope = slope;
    // few pixel rows against a 2048 x 2048 weight matrix: every XCD keeps ITS slice of W in L2 and streams the operand rows
    // (with the conv map each XCD walked the whole 8-17 MB of W once per M-tile out of the memory-side cache)
    p.nmajor = agrl_opts().graph_linear_mmajor ? 0 : 1;   // AGRL_GRAPH_LINEAR_MMAJOR=1: A/B switch
    if (in_dtype == AGRL_F32) return launch_igemm<float, float>(p, (hipStream_t)stream, "agrl_graph_linear_mix");
    if (in_dtype == AGRL_F32X3) return launch_igemm<f32s_t, float>(p, (hipStream_t)stream, "agrl_graph_linear_mix");
    // split-fp16 (round 6): w holds the weight times a power of two 2^k (max |w| 2^k in [2^13, 2^14)), pre-split by agrl_split16_weights_inloop;
    // the caller folds 2^-k into bn_scale (exact)
    if (in_dtype == AGRL_F32H3) AGRL_CHECK_ARG(K % 32 == 0, "agrl_graph_linear_mix: split-fp16 needs K %% 32 == 0 (got %d)", K);
    if (in_dtype == AGRL_F32H3) {
        p.a_pre = p_presplit ? 1 : 0;
        return launch_igemm<f32h_t, float>(p, (hipStream_t)stream, "agrl_graph_linear_mix");
    }
    return launch_igemm<lp16_t, float>(p, (hipStream_t)stream, "agrl_graph_linear_mix");
}

// out[m][n] = alpha * sum_z ws[z][m][n] + rowv[m] (or rowc) + colv[n]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, float* __restrict__ out,
                                                            int M, int N, int ldo, float alpha, const float* __restrict__ rowv,
                                                            const float* __restrict__ colv, float rowc) {
    const size_t total = (size_t)M * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int m = (int)(e / N), n = (int)(e - (size_t)m * N);
        float s = 0.f;
        for (int z = 0; z < ksplit; ++z) s += ws[(size_t)z * total + e];
        out[(size_t)m * ldo + n] = fmaf(alpha, s, (rowv ? rowv[m] : rowc) + (colv ? colv[n] : 0.f));
    }
}

extern "C" int agrl_gemm_nt_splitk(const void* x, const void* w, float* y, int M, int K, int Nout, int in_dtype, void* workspace,
                                   size_t workspace_bytes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(x && w && y, "agrl_gemm_nt_splitk: null pointer");
    AGRL_CHECK_ARG(in_dtype == AGRL_F32 || in_dtype == AGRL_LP16 || in_dtype == AGRL_F32X3, "agrl_gemm_nt_splitk: bad dtype %d", in_dtype);
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = x; p.w = w; p.colv = nullptr; p.rowv = nullptr; p.res = nullptr; p.out = y;
    p.alpha = 1.f; p.rowc = 0.f; p.relu = 0; p.ksplit = 1; p.pool_nparts = 0;
    p.M = M; p.N = Nout; p.K = K;
    p.Cin = K; p.H = 1; p.W = 1; p.OH = 1; p.OW = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.ldo = Nout;
    // weight gradients: a handful of output tiles and a K axis of 10^4..10^6 pixels -> K slices over workgroups until the
    // chip is covered twice, fp32 partials in the caller's workspace, summed in slice order (deterministic)
    const int bke = in_dtype == AGRL_LP16 ? 64 : 32;
    AGRL_CHECK_ARG(K % bke == 0, "agrl_gemm_nt_splitk: K=%d must be a multiple of %d", K, bke);
    const int nk = K / bke;
    const int tiles = cdiv(M, 64) * cdiv(Nout, Nout <= 64 ? 64 : 128);
    int ks = 1;
    while (ks < 256 && tiles * ks < 1024 && nk % (ks * 2) == 0 && nk / (ks * 2) >= 8) ks *= 2;
    if (ks > 1 && workspace && workspace_bytes >= (size_t)ks * M * Nout * sizeof(float)) {
        IgemmParams ps = p;
        ps.out = workspace; ps.ksplit = ks;
        int rc;
        if (in_dtype == AGRL_F32) rc = launch_igemm<float, float>(ps, (hipStream_t)stream, "agrl_gemm_nt_splitk");
        else if (in_dtype == AGRL_F32X3) rc = launch_igemm<f32s_t, float>(ps, (hipStream_t)stream, "agrl_gemm_nt_splitk");
        else rc = launch_igemm<lp16_t, float>(ps, (hipStream_t)stream, "agrl_gemm_nt_splitk");
        if (rc) return rc;
        const size_t total = (size_t)M * Nout;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, ks, y, M, Nout,
                           Nout, 1.f, (const float*)nullptr, (const float*)nullptr, 0.f);
        AGRL_CHECK_LAUNCH("agrl_gemm_nt_splitk(reduce)");
        return 0;
    }
    if (in_dtype == AGRL_F32) return launch_igemm<float, float>(p, (hipStream_t)stream, "agrl_gemm_nt_splitk");
    if (in_dtype == AGRL_F32X3) return launch_igemm<f32s_t, float>(p, (hipStream_t)stream, "agrl_gemm_nt_splitk");
    return launch_igemm<lp16_t, float>(p, (hipStream_t)stream, "agrl_gemm_nt_splitk");
}

static int distmat_impl(const void* q, const void* g, const float* qn, const float* gn, float* dist, int m, int n, int D, int ldd, int metric,
                        int dtype, void* workspace, size_t workspace_bytes, float dot_scale, agrl_stream_t stream);

extern "C" int agrl_distmat(const void* q, const void* g, const float* qn, const float* gn, float* dist, int m, int n, int D,
                            int ldd, int metric, int dtype, void* workspace, size_t workspace_bytes, agrl_stream_t stream) {
    return distmat_impl(q, g, qn, gn, dist, m, n, D, ldd, metric, dtype, workspace, workspace_bytes, 1.f, stream);
}

// The distance matrix in the split-fp16 arithmetic (round 6, the conforming mode; torchreid/metrics/distance.py:59-89): q3 (m, D3) and g3
// (n, D3) are fp16 PLANE operands of D3 = 3 D columns -- queries [qh | ql 2^11 | qh] (agrl_split16_planes of the fp32 rows), gallery
// [gh | gh 2^-11 | gl] of g 2^k (hip_ops.split16_plane_weights) -- so that the 16-bit kernels' plain dot product over D3 columns IS
// qh gh + ql gh + qh gl, fp32-class (22 significand bits per operand), and g_unscale = 2^-k un-does the gallery's pre-scale inside the
// epilogue's alpha. qn / gn (euclidean) are the fp32 squared norms of the TRUE rows. Same kernels, same dispatch as agrl_distmat.
extern "C" int agrl_distmat_split16(const void* q3, const void* g3, const float* qn, const float* gn, float* dist, int m, int n, int D3,
                                    int ldd, int metric, float g_unscale, void* workspace, size_t workspace_bytes, agrl_stream_t stream) {
    AGRL_CHECK_ARG(agrl_lp16_is_f16(), "agrl_distmat_split16: the split planes are fp16 (load libagrl_hip.so, not the bf16 build)");
    AGRL_CHECK_ARG(D3 > 0 && D3 % 3 == 0 && g_unscale > 0.f && g_unscale <= 3.4e38f, "agrl_distmat_split16: D3 = 3 x columns, g_unscale a positive power of two");
    {
        int e = 0;
        AGRL_CHECK_ARG(frexpf(g_unscale, &e) == 0.5f, "agrl_distmat_split16: g_unscale=%g is not a power of two", (double)g_unscale);
    }
    return distmat_impl(q3, g3, qn, gn, dist, m, n, D3, ldd, metric, AGRL_LP16, workspace, workspace_bytes, g_unscale, stream);
}

static int distmat_impl(const void* q, const void* g, const float* qn, const float* gn, float* dist, int m, int n, int D, int ldd, int metric,
                        int dtype, void* workspace, size_t workspace_bytes, float dot_scale, agrl_stream_t stream) {
    AGRL_CHECK_ARG(q && g && dist, "agrl_distmat: null pointer");
    AGRL_CHECK_ARG(m > 0 && n > 0 && D > 0 && ldd >= n, "agrl_distmat: bad shape m=%d n=%d D=%d ldd=%d", m, n, D, ldd);
    AGRL_CHECK_ARG(dtype == AGRL_F32 || dtype == AGRL_LP16, "agrl_distmat: bad dtype %d", dtype);
    IgemmParams p;
    p.x2 = nullptr; p.K1 = 0; p.stats = nullptr;
    p.x = q; p.w = g; p.res = nullptr; p.out = dist; p.relu = 0; p.ksplit = 1; p.pool_nparts = 0;
    if (metric == AGRL_METRIC_EUCLIDEAN) {
        AGRL_CHECK_ARG(qn && gn, "agrl_distmat: euclidean needs the squared row norms");
        p.alpha = -2.f * dot_scale; p.rowv = qn; p.colv = gn; p.rowc = 0.f;   // (dot_scale: 1, or the power of two of agrl_distmat_split16 -- exact)
    } else if (metric == AGRL_METRIC_COSINE) {
        p.alpha = -1.f * dot_scale; p.rowv = nullptr; p.colv = nullptr; p.rowc = 1.f;
    } else {
        agrl_set_error("agrl_distmat: unknown metric %d", metric);
        return 1;
    }
    p.M = m; p.N = n; p.K = D;
    p.Cin = D; p.H = 1; p.W = 1; p.OH = 1; p.OW = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.ldo = ldd;
    // streaming form (one eval batch of queries against a long gallery): dedicated single-pass kernel
    if (!agrl_opts().distmat_tiled && distmat_stream_applicable(p, dtype == AGRL_F32 ? 4 : 2))
        return launch_distmat_stream(p, dtype, (hipStream_t)stream);
    // otherwise, when there are too few output tiles to fill 256 CUs -> split K over
    // workgroups, fp32 partials in the caller's workspace, deterministic reduce + epilogue afterwards
    const int bke = dtype == AGRL_F32 ? 32 : 64;
    const int nk = D / bke;
    const int tiles = cdiv(m, 64) * cdiv(n, 128);
    int ks = 1;
    while (ks < 8 && tiles * ks < 512 && nk % (ks * 2) == 0 && nk / (ks * 2) >= 4) ks *= 2;
    if (ks > 1 && workspace && workspace_bytes >= (size_t)ks * m * n * sizeof(float) && D % bke == 0) {
        IgemmParams ps = p;
        ps.out = workspace; ps.ldo = n; ps.alpha = 1.f; ps.rowv = nullptr; ps.colv = nullptr; ps.rowc = 0.f; ps.ksplit = ks;
        int rc = dtype == AGRL_F32 ? launch_igemm<float, float>(ps, (hipStream_t)stream, "agrl_distmat")
                                   : launch_igemm<lp16_t, float>(ps, (hipStream_t)stream, "agrl_distmat");
        if (rc) return rc;
        const size_t total = (size_t)m * n;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, ks,
                           dist, m, n, ldd, p.alpha, p.rowv, p.colv, p.rowc);
        AGRL_CHECK_LAUNCH("agrl_distmat(reduce)");
        return 0;
    }
    if (dtype == AGRL_F32) return launch_igemm<float, float>(p, (hipStream_t)stream, "agrl_distmat");
    return launch_igemm<lp16_t, float>(p, (hipStream_t)stream, "agrl_distmat");
}
