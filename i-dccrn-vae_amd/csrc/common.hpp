// Shared device helpers for the I-DCCRN-VAE MI355X (gfx950) kernels.
//
// Activation layout used by every kernel in this library ("planar-J"):
//   act[ri][C][F][Jp]   fp32, ri = 0 real / 1 imag plane
//   column j = b * Tp + tp,  Tp = T_stft + 1,  tp = t + 1
//   tp == 0 is a zero guard column in front of every utterance (it provides the causal
//   x[t-1] = 0 tap without a branch), columns tp > T_valid are zero as well.
//   Jp (row stride) = J rounded up to 4; buffers carry IDV_SLACK floats in front and behind.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IDV_SLACK 256

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define IDV_OK 0
#define IDV_EINVAL (-1)
#define IDV_ELAUNCH (-2)

static inline int idv_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IDV_OK : IDV_ELAUNCH;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// gate non-linearities on the hardware exp2 / rcp (about 1 ulp each): absolute error ~1e-7, which is what the
// LSTM cell needs; tanh(x) = 2 sigmoid(2x) - 1.  They sit on the per-step critical path of the recurrence.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.0f;
}

// Train-mode moment sums with less atomic contention: the conv epilogues add their per-wave partials to one of `rep` replicas
// [rep][C][5] of the sums (chosen by workgroup index) instead of all to the same C x 5 doubles -- at 32 or 64 channels and
// ~10^4 workgroups the same-address atomics cost more than the contraction (enc1, B = 32: 4.1 ms against 1.65) -- and this
// kernel folds the replicas into stats[C][5] (+=, fixed order).  elementwise.hip
int idv_launch_stats_collapse(const double* work, int rep, int n, double* stats, hipStream_t st);

