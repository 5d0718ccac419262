// Split-bf16 activation images (the inter-layer format of the bf16x3 path) <-> planar-J fp32.
//   image[hi|lo][octet o][f][j][8]  (bf16):  element e of octet o is planar channel cc = 8*o + e = 2*ci + ri,
//   hi = x truncated to bf16, lo = round-to-nearest bf16 of (x - hi); x is recovered as hi + lo to ~2^-17.
// The consumers (cgemm_bf16.hip / cgemm_c1.hip, IMGIN) copy 16-byte slots straight into their LDS patch.
#include "bf16_common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

// one thread: one (octet, f, j) slot = 8 planes
// rep > 1: every utterance (Tp columns) of x appears rep times in a row in the image (skip.repeat_interleave(rep, 0) of the
// two-phase decoder, reference pvae_module.py:2563-2567, fused into the conversion); x then has J / rep columns, pitch Jp_in
__global__ void planar_to_image_kernel(const float* __restrict__ x, int C, int F, int J, int Jp, unsigned short* __restrict__ img,
                                       long long lo_off, int Tp, int rep, int Jp_in) {
    const long long n = (long long)((2 * C + 7) / 8) * F * Jp;
    if (blockIdx.x == 0 && threadIdx.x < 2)
        *(uint4*)(img + IDV_IMG_ZSLOT * 8 + (threadIdx.x ? lo_off : 0)) = make_uint4(0u, 0u, 0u, 0u);
    // 8 zero slots behind each plane: a consumer with time taps (x[t], x[t+1]) (the data-gradient form, tshift 0) reads
    // column J of the last row, which is the slot behind the plane when the pitch has no padding (Jp == J)
    if (blockIdx.x == 0 && threadIdx.x >= 32 && threadIdx.x < 48)
        *(uint4*)(img + (n + (threadIdx.x & 7)) * 8 + ((threadIdx.x & 8) ? lo_off : 0)) = make_uint4(0u, 0u, 0u, 0u);
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Jp);
        const long long t = idx / Jp;
        const int f = (int)(t % F), o = (int)(t / F);
        unsigned hw[4], lw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int ci = 4 * o + w;
            float x0 = 0.f, x1 = 0.f;
            if (ci < C && j < J) {
                int jin = j;
                if (rep > 1) {
                    const int bo = j / Tp;
                    jin = (bo / rep) * Tp + (j - bo * Tp);
                }
                x0 = x[((size_t)ci * F + f) * Jp_in + jin];
                x1 = x[((size_t)(C + ci) * F + f) * Jp_in + jin];
            }
            const unsigned u0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u;
            const unsigned u1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
            hw[w] = (u0 >> 16) | u1;
            lw[w] = pack_bf16(x0 - __builtin_bit_cast(float, u0), x1 - __builtin_bit_cast(float, u1));
        }
        *(uint4*)(img + idx * 8) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *(uint4*)(img + lo_off + idx * 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

__global__ void image_to_planar_kernel(const unsigned short* __restrict__ img, long long lo_off, int C, int F, int J, int Jp,
                                       float* __restrict__ x) {
    const long long n = (long long)((2 * C + 7) / 8) * F * Jp;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Jp);
        const long long t = idx / Jp;
        const int f = (int)(t % F), o = (int)(t / F);
        if (j >= J) continue;
        const uint4 h = *(const uint4*)(img + idx * 8), l = *(const uint4*)(img + lo_off + idx * 8);
        const unsigned hh[4] = {h.x, h.y, h.z, h.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int ci = 4 * o + w;
            if (ci >= C) continue;
            const float r = __builtin_bit_cast(float, hh[w] << 16) + __builtin_bit_cast(float, ll[w] << 16);
            const float i = __builtin_bit_cast(float, hh[w] & 0xffff0000u) + __builtin_bit_cast(float, ll[w] & 0xffff0000u);
            x[((size_t)ci * F + f) * Jp + j] = r;
            x[((size_t)(C + ci) * F + f) * Jp + j] = i;
        }
    }
}

}  // namespace

extern "C" int idv_planar_to_image(const float* x, int C, int F, int J, int Jp, void* img, long long lo_off, void* stream) {
    if (!x || !img || C <= 0 || F <= 0 || J <= 0 || Jp < J || (lo_off % 8) || (reinterpret_cast<uintptr_t>(img) & 15)) return IDV_EINVAL;
    const long long n = (long long)((2 * C + 7) / 8) * F * Jp;
    long long g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(planar_to_image_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, C, F, J, Jp,
                       (unsigned short*)img, lo_off, J, 1, Jp);
    return idv_launch_status();
}

// planar [2][C][F][Jp_in] with B utterances of Tp columns -> split image with B * rep utterances, each input utterance rep times
// in a row (pitch Jp >= B * rep * Tp)
extern "C" int idv_planar_to_image_repeat(const float* x, int C, int F, int B, int Tp, int Jp_in, int rep, void* img, long long lo_off,
                                          int Jp, void* stream) {
    if (!x || !img || C <= 0 || F <= 0 || B <= 0 || Tp <= 1 || rep <= 0 || Jp_in < B * Tp || Jp < (long long)B * rep * Tp || (lo_off % 8) ||
        (reinterpret_cast<uintptr_t>(img) & 15))
        return IDV_EINVAL;
    const long long n = (long long)((2 * C + 7) / 8) * F * Jp;
    long long g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(planar_to_image_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, C, F, B * rep * Tp, Jp,
                       (unsigned short*)img, lo_off, Tp, rep, Jp_in);
    return idv_launch_status();
}

extern "C" int idv_image_to_planar(const void* img, long long lo_off, int C, int F, int J, int Jp, float* x, void* stream) {
    if (!x || !img || C <= 0 || F <= 0 || J <= 0 || Jp < J || (lo_off % 8) || (reinterpret_cast<uintptr_t>(img) & 15)) return IDV_EINVAL;
    const long long n = (long long)((2 * C + 7) / 8) * F * Jp;
    long long g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(image_to_planar_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)img, lo_off, C, F, J, Jp, x);
    return idv_launch_status();
}
