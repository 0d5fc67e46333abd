// Device-side building blocks shared by the implicit-GEMM kernels (igemm.hip, igemm_wide.hip): parameter block,
// MFMA fragment wrappers, the swizzled 128-byte-row LDS layout, LDS-DMA and counted waits.
#pragma once
#include "agrl_common.h"

struct IgemmParams {
    const void* x;
    const void* x2;  // second pixel-row operand of the two-source pointwise GEMM (igemm_wide DUAL form) or nullptr
    int K1;          // two-source form: k-columns [0, K1) come from x (row length K1), [K1, K) from x2 (row length K - K1)
    const void* w;
    const float* colv;  // per output channel (bias / gallery sq-norm) or nullptr
    const float* rowv;  // per output pixel (query sq-norm) or nullptr
    const void* res;    // residual, same layout/dtype as out, or nullptr
    void* out;
    float alpha;  // out = alpha*acc + rowv[m] + colv[n] + rowc (+ res) (relu)
    float rowc;   // constant added when rowv == nullptr (cosine: 1.0)
    int relu;
    int M, N, K;
    int Cin, H, W, OH, OW, R, S, stride, pad;
    int ldo;     // row stride of out / res in elements
    int vec_ok;  // 4-wide epilogue accesses are aligned
    int ksplit;  // > 1: blockIdx.y selects a K slice and raw fp32 partials go to out + z*M*ldo (pointwise only)
    // fused pooling epilogue (persistent kernel only; a tile must be exactly one frame: BM == OH*OW):
    int pool_nparts;      // 0 = off; else number of row bins summed per frame
    int pool_mean;        // 1: write bin means, 0: write bin sums
    int pool_store_out;   // 0: the activation tile itself is not written to HBM
    int pool_w;           // pixels per image row
    int pool_start[16], pool_end[16];  // bins in image rows [start, end)
    float* pool_out;      // fp32 (frames, nparts, N)
    void* pool_out_lp;    // optional bf16 copy
    float* stats;  // optional (fp32-output, non-LDS epilogue only): per-channel sum / sum of squares of the finished tile's rows, [M / 64 granule][2][N] (train: batch statistics without re-reading the conv output)
    // GraphLayer epilogue (fp32 output, register epilogue only): out = mix_keep * mix_f[m][n] + mix_gamma * lrelu(mix_scale[n] * acc + colv[n])
    // -- BatchNorm1d (folded) + LeakyReLU + the residual mix of vmgn.py:169-172 applied to (G f) W^T; nullptr = off
    const float* mix_f = nullptr;
    const float* mix_scale = nullptr;
    float mix_keep = 0.f, mix_gamma = 0.f, mix_slope = 0.f;
    int nmajor = 0;  // XCD map: 1 = each XCD owns a contiguous range of N-TILES for all M-tiles (few pixel rows, large weight
                     // matrix: the XCD's weight slice stays in its 4 MB L2 and the small activation matrix is streamed per
                     // XCD); 0 = the N-tiles of one M-tile share an XCD (convs: large activations, small weights)
    // split-fp16 in-loop kernels (AGRL_F32H3): activations that only ever feed a GEMM may be stored PRE-SPLIT, in the layout of
    // agrl_split16_weights_inloop (per 32-channel group [8 x fp16 hi | 8 x fp16 lo] x 4 lane groups: the bytes of 32 fp32) -- the k-loop
    // then has no VALU work for them. a_pre bit 0: x is pre-split, bit 1: x2 is; out_pre: write out in that layout (no residual)
    int a_pre = 0, out_pre = 0;
    int out_planes = 0;   // 2 / 3: write out as split-fp16 PLANES, (M, out_planes N) fp16 = [hi | lo 2^11 (| hi)] (the seam to the plane kernels)
    int dbg;     // ablation bits (AGRL_IGEMM_DBG, profiling only): 1 skip global stores, 4 skip epilogue phase 1, 8 skip steady-state DMA, 32 skip the DMA waits, 64 burst-issue DMA instead of interleaving
};

template <typename T>
struct Frag;

template <>
struct Frag<lp16_t> {
    // one 16-byte chunk = 8 elements = a quarter of the 32-deep k-step of v_mfma_f32_16x16x32_f16 / _bf16
    __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) { return mfma_lp16_16x16x32(a, b, c); }
};

template <>
struct Frag<float> {
    // one 16-byte chunk = 4 fp32; lane group g = lane>>4 feeds hardware-k g of MFMA j with actual
    // k = 4*chunk + j -- the same assignment for both operands, so the sum over k is complete
    __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
        return c;
    }
};

template <>
struct Frag<f32s_t> {
    // Same operand layout as Frag<float> (a 16-byte chunk = 4 fp32 = the four hardware-k values of this lane's k-group in
    // v_mfma_f32_16x16x16_bf16). x = xh + xl with xh = x truncated to bf16 (exact difference) and xl rounded to bf16:
    // x w = xh wh + xh wl + xl wh + O(2^-16 |x w|), three bf16 MFMAs (3 x 16 cycles) instead of four fp32 ones (4 x 32).
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    __device__ static inline void split(const uint4& v, s16x4_t& hi, s16x4_t& lo) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __uint_as_float(w[i]) - __uint_as_float(w[i] & 0xffff0000u);
        const uint32_t h0 = (w[0] >> 16) | (w[1] & 0xffff0000u), h1 = (w[2] >> 16) | (w[3] & 0xffff0000u);
        const uint32_t l0 = pack_bf16x2(d[0], d[1]), l1 = pack_bf16x2(d[2], d[3]);
        hi = __builtin_bit_cast(s16x4_t, make_uint2(h0, h1));
        lo = __builtin_bit_cast(s16x4_t, make_uint2(l0, l1));
    }
    __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
        s16x4_t ah, al, bh, bl;
        split(a, ah, al);
        split(b, bh, bl);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c, 0, 0, 0);
        return c;
    }
};

template <>
struct Frag<f32h_t> {
    // Same operand layout as Frag<float> / Frag<f32s_t> (a 16-byte chunk = 4 fp32 = this lane's four hardware-k values of
    // v_mfma_f32_16x16x16_f16). x = xh + xl, xh = fp16(x) rounded to nearest, xl = fp16(x - xh): the difference is exact in fp32
    // (xh keeps 11 of x's 24 significand bits), so xh + xl carries 22 bits wherever xl is a NORMAL fp16 (|x| >= ~2^-2; below that the
    // subnormal spacing 2^-24 bounds the ABSOLUTE error of the pair at 2^-25 -- the caller scales its small operand, the weights,
    // into range: agrl_conv2d_bn_act_split16). x w = xh wh + xh wl + xl wh + O(2^-22 |x w|): three fp16 MFMAs, fp32 accumulation,
    // the small terms first.
    typedef _Float16 h16x4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
    __device__ static inline void split(const uint4& v, h16x4_t& hi, h16x4_t& lo) {
        const float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        const h16x2_t h0 = {(_Float16)f[0], (_Float16)f[1]}, h1 = {(_Float16)f[2], (_Float16)f[3]};   // v_cvt_pk_f16_f32 (RNE)
        const float d0 = f[0] - (float)h0[0], d1 = f[1] - (float)h0[1], d2 = f[2] - (float)h1[0], d3 = f[3] - (float)h1[1];
        const h16x2_t l0 = {(_Float16)d0, (_Float16)d1}, l1 = {(_Float16)d2, (_Float16)d3};
        hi = h16x4_t{h0[0], h0[1], h1[0], h1[1]};
        lo = h16x4_t{l0[0], l0[1], l1[0], l1[1]};
    }
    __device__ static inline f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
        h16x4_t ah, al, bh, bl;
        split(a, ah, al);
        split(b, bh, bl);
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
        return c;
    }
    // The K = 32 form (igemm_kernel's k-loop for this type): the two 16-byte chunks a lane reads of a 128-byte k-tile (k-chunks g and
    // 4 + g, g = lane >> 4) make ONE 8-element operand of v_mfma_f32_16x16x32_f16 -- the lane's hardware-k slots 0-3 take the first
    // chunk, 4-7 the second, the same assignment for both operands -- so a k-tile costs three MFMAs per fragment pair instead of six
    // at the same cycles each (the K = 16 instructions of the gfx90a generation run at half the gfx950 rate).
    __device__ static inline void split8(const uint4& c0, const uint4& c1, uint4& hi, uint4& lo) {
        const float f[8] = {__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w),
                            __uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z), __uint_as_float(c1.w)};
        uint32_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const h16x2_t hh = {(_Float16)f[2 * e], (_Float16)f[2 * e + 1]};
            const h16x2_t ll = {(_Float16)(f[2 * e] - (float)hh[0]), (_Float16)(f[2 * e + 1] - (float)hh[1])};
            h[e] = __builtin_bit_cast(uint32_t, hh);
            l[e] = __builtin_bit_cast(uint32_t, ll);
        }
        hi = make_uint4(h[0], h[1], h[2], h[3]);
        lo = make_uint4(l[0], l[1], l[2], l[3]);
    }
    __device__ static inline f32x4_t mma32(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bl), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
        return c;
    }
};

__device__ inline int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename TOUT>
__device__ inline void store4(TOUT* p, const float v[4]);
template <>
__device__ inline void store4<float>(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ inline void store4<lp16_t>(lp16_t* p, const float v[4]) {
    uint2 u;
    u.x = pack_lp16x2(v[0], v[1]);
    u.y = pack_lp16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}
template <typename TOUT>
__device__ inline void load4(const TOUT* p, float v[4]);
template <>
__device__ inline void load4<float>(const float* p, float v[4]) {
    float4 f = *reinterpret_cast<const float4*>(p);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
template <>
__device__ inline void load4<lp16_t>(const lp16_t* p, float v[4]) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    unpack_lp16x2(u.x, v[0], v[1]);
    unpack_lp16x2(u.y, v[2], v[3]);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// 16 zero bytes in device memory: the DMA source of every padded / out-of-range chunk
static __device__ __attribute__((aligned(16))) uint4 g_zero16;

__device__ inline void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    // LDS-DMA: lane L's 16 bytes land at lds_wave_base + 16*L (wave-uniform base), no VGPR round trip.
    // Issued through inline asm on purpose: hipcc tracks the builtin as an LDS write and then puts
    // s_waitcnt vmcnt(0) in front of the next ds_read, which would drain the ring every k-tile; hidden in asm, the
    // counted waits below are the only ones. M0 (the LDS destination base) is saved/restored in the same statement.
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)lds_wave_base);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_addr)
        : "memory");
}

template <int N>
__device__ inline void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ inline void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Conflict-free shifted patch reads. A ds_read_b128 is served in groups of 16 lanes -- MFMA columns (lane & 15) 0-3 and
// 12-15 of one k-chunk together with columns 4-11 of the neighbouring k-chunk -- and 16 lanes are conflict-free when
// they hit 16 different 16-byte slots of the 256-byte bank window (= two 128-byte patch pixels, even / odd x). The
// pixel <-> MFMA column assignment is free, so columns 0-3, 12-15 take the eight pixels of one block row and columns
// 4-11 the eight of the next one; with the k-chunk of patch pixel (py, px) stored at chunk ^ g(py, px),
// g = ((px >> 1) & 3) ^ 4 (py & 1), the four same-parity pixels of a row land in four different slots for ANY tap shift
// (x >> 1 runs over four consecutive values), the neighbouring row differs in bit 2, and the k-chunk pair c, c ^ 1 of a
// group stays inside the row's own four-slot class. (With consecutive pixels on consecutive columns no swizzle of a
// 10-wide patch is conflict-free: exhaustive search.)
__device__ inline int frag_px(int col) { return col < 4 ? col : (col < 12 ? col + 4 : col - 8); }  // 8 * row + x inside the fragment
__device__ inline int patch_g(int py, int px) { return ((px >> 1) & 3) ^ ((py & 1) << 2); }
template <int PW_>
__device__ inline int patch_off(int py, int px, int chunk) { return (py * PW_ + px) * 128 + ((chunk ^ patch_g(py, px)) << 4); }

// 256 x 256 tile kernel for the MFMA-bound pointwise layers (igemm_wide.hip)
bool igemm_wide_applicable(const IgemmParams& p);
int launch_igemm_wide(const IgemmParams& p, hipStream_t stream, const char* who);
bool igemm_wide_f32out_applicable(const IgemmParams& p);   // fp32 output + distance epilogue (the full query x gallery matrix)
int launch_igemm_wide_f32out(const IgemmParams& p, hipStream_t stream, const char* who);

// streaming distance matrix (few queries x long gallery, distmat_stream.hip)
bool distmat_stream_applicable(const IgemmParams& p, int elem_size);
int launch_distmat_stream(const IgemmParams& p, int dtype, hipStream_t stream);

// GraphLayer Linear with the BatchNorm / LeakyReLU / residual-mix epilogue, bf16 operands (graph_gemm.hip)
bool graph_gemm_applicable(int M, int K, int Nout);
int launch_graph_gemm(const void* p_op, const void* w, const float* f, const float* bn_scale, const float* bn_shift, float keep,
                      float gamma, float slope, float* out, int M, int K, int Nout, hipStream_t stream);

// two-block 3x3 kernel (conv3x3_wide.hip)
int launch_conv3x3_wide(const IgemmParams& p, hipStream_t stream);
int launch_conv3x3_c64(const IgemmParams& p, hipStream_t stream);

