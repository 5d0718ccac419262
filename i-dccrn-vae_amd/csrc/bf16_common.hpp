// Helpers shared by the split-bf16 (bf16x3) kernels.
#pragma once
#include "cgemm.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_round(float a) { return (float)(__bf16)a; }

// W'(m, cc, kf, kt) of the block matrix [[Wr,-Wi],[Wi,Wr]] with the optional BN fold (same as pack.hip)
__device__ __forceinline__ float wprime(const float* w_re, const float* w_im, const float* fold, int Cout, int Cin_total,
                                        int Cin_used, int transposed, int m, int cc, int kf, int kt, int conj = 0) {
    const int co = m >> 1, ro = m & 1, ci = cc >> 1, ri = cc & 1;
    if (co >= Cout || ci >= Cin_used) return 0.f;
    const size_t off = transposed ? (((size_t)ci * Cout + co) * 5 + kf) * 2 + kt
                                  : (((size_t)co * Cin_total + ci) * 5 + kf) * 2 + kt;
    const float wr = w_re[off], wi = conj ? -w_im[off] : w_im[off];
    const float top = ri == 0 ? wr : -wi, bot = ri == 0 ? wi : wr;
    if (fold) {
        const float* z = fold + (size_t)co * 6;
        return ro == 0 ? z[0] * top + z[1] * bot : z[2] * top + z[3] * bot;
    }
    return ro == 0 ? top : bot;
}
