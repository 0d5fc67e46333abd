// Shared pieces of the four-wave, 512-register kernels (conv3x3_fat.hip, conv1x1_fat.hip): compile-time loops, MFMAs on named
// (asm-owned) AGPR accumulator quads, loads / waits / LDS-DMA hidden from hipcc's own wait insertion. See conv3x3_fat.hip's header.
#pragma once
#include <utility>

#include "igemm_dev.h"

namespace {

template <typename F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;

template <int AQ>  // AGPR quad AQ += a x b
__device__ __forceinline__ void fat_mfma(const u32x4_t& a, const u32x4_t& b) {
    if constexpr (kLpF16) asm volatile("v_mfma_f32_16x16x32_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * AQ), "n"(4 * AQ + 3));
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * AQ), "n"(4 * AQ + 3));
}
template <int AQ>
__device__ __forceinline__ void fat_zero() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0" ::"n"(4 * AQ),
                 "n"(4 * AQ + 1), "n"(4 * AQ + 2), "n"(4 * AQ + 3));
}
template <int AQ>
__device__ __forceinline__ f32x4_t fat_read() {
    float x, y, z, w;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(x), "=v"(y), "=v"(z), "=v"(w)
                 : "n"(4 * AQ), "n"(4 * AQ + 1), "n"(4 * AQ + 2), "n"(4 * AQ + 3));
    return f32x4_t{x, y, z, w};
}
// 16 bytes per lane global -> VGPRs behind hipcc's back: valid only after a counted wait that names the register
template <int IMM>
__device__ __forceinline__ void fat_gload(u32x4_t& dst, unsigned off, const unsigned char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(off), "s"(base), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void fat_wait(u32x4_t& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
__device__ __forceinline__ void fat_dma(const unsigned char* src, unsigned lds_wave_addr) {  // lane L's 16 bytes -> LDS lds_wave_addr + 16 L
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_wave_addr)
                 : "memory");
}

// the same with a scalar base and a 32-bit lane offset (one VGPR per request instead of two)
__device__ __forceinline__ void fat_dma_s(const unsigned char* sbase, unsigned voff, unsigned lds_wave_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_addr)
                 : "memory");
}

}  // namespace
