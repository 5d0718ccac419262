// Multi-tensor kernels of the data-parallel train step: the gradient bucket (gather before / scatter after the RCCL
// all-reduce) and the Adam update, each ONE launch over a pointer table instead of one torch kernel per parameter.
//
// The reference is single-GPU stock torch: `optimizer.step()` of torch.optim.Adam(lr, weight_decay = 0.001)
// (supervised_dccrn/train.py:109, 239-243; i_dccrn_vae/nsvae_dccrn/train_nsvae.py:200, 557-561;
// i_dccrn_vae/nsvae_dccrn/train_second_phase_decoder.py:420-433); the bucket is the MI355X-native addition north_star names
// ("a single RCCL all-reduce of gradients over xGMI per step").
//
// Layout: a bucket is a flat fp32 buffer in which tensor i occupies [off_i, off_i + numel_i), off_i a multiple of 4 floats
// (16 bytes) and the gap up to off_{i+1} padding (kept zero).  table[i] = {device pointer, off_i, numel_i} as three int64,
// sorted by off_i; `total` = off_n (multiple of 4).  A thread owns one 16-byte group of the flat buffer, which by the
// alignment of the offsets lies inside ONE tensor: float4 on the flat side always, float4 on the tensor side when its base
// pointer is 16-byte aligned (torch allocations are), scalar otherwise and for a tensor's last partial group.
// HBM-bound: 8 bytes moved per gradient element and direction (25 M parameters of the NSVAE encoder: 2 x 100 MB).
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

constexpr int BK_THREADS = 256;
constexpr int BK_CHUNK = BK_THREADS * 4;      // floats of the flat buffer per workgroup

// index of the tensor whose range holds flat element g (g < total): largest i with off_i <= g
__device__ __forceinline__ int bucket_find(const long long* __restrict__ table, int n, long long g0, long long g, int* s_idx) {
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[3 * mid + 1] <= g0) lo = mid; else hi = mid - 1;
        }
        *s_idx = lo;
    }
    __syncthreads();
    int idx = *s_idx;
    while (idx + 1 < n && table[3 * (idx + 1) + 1] <= g) ++idx;
    return idx;
}

// GATHER: flat <- tensors (a NULL pointer contributes zeros; padding is written as zero)
// !GATHER: tensors <- scale * flat
template <bool GATHER>
__global__ void __launch_bounds__(BK_THREADS) bucket_copy_kernel(const long long* __restrict__ table, int n, long long total,
                                                                 float* __restrict__ flat, float scale) {
    __shared__ int s_idx;
    const long long g0 = (long long)blockIdx.x * BK_CHUNK;
    const long long g = g0 + 4 * threadIdx.x;
    const int idx = bucket_find(table, n, g0, g < total ? g : g0, &s_idx);
    if (g >= total) return;
    float* tp = reinterpret_cast<float*>(table[3 * idx]);
    const long long local = g - table[3 * idx + 1];
    const long long left = table[3 * idx + 2] - local;          // valid elements from here (<= 0 inside the padding)
    float4* fl = reinterpret_cast<float4*>(flat + g);
    if (GATHER) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tp && left >= 4 && !(reinterpret_cast<uintptr_t>(tp) & 15)) {
            v = *reinterpret_cast<const float4*>(tp + local);
        } else if (tp && left > 0) {
            v.x = tp[local];
            if (left > 1) v.y = tp[local + 1];
            if (left > 2) v.z = tp[local + 2];
            if (left > 3) v.w = tp[local + 3];
        }
        *fl = v;
    } else {
        if (!tp || left <= 0) return;
        float4 v = *fl;
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (left >= 4 && !(reinterpret_cast<uintptr_t>(tp) & 15)) {
            *reinterpret_cast<float4*>(tp + local) = v;
        } else {
            tp[local] = v.x;
            if (left > 1) tp[local + 1] = v.y;
            if (left > 2) tp[local + 2] = v.z;
            if (left > 3) tp[local + 3] = v.w;
        }
    }
}

struct AdamArgs {
    const long long* ptable;      // parameters
    const long long* gtable;      // gradients (same offsets / sizes), or NULL: gflat
    const float* gflat;           // gradients in the bucket layout (e.g. the all-reduced bucket), or NULL: gtable
    float* m;                     // exp_avg, bucket layout
    float* v;                     // exp_avg_sq, bucket layout
    int n;
    long long total;
    float lr, beta1, beta2, omb1, omb2, eps, wd, bc1, bc2s, gscale;   // omb = 1 - beta rounded from double as torch does; bc1 = 1 - beta1^t, bc2s = sqrt(1 - beta2^t)
};

// torch.optim.Adam (no amsgrad, no maximize), in its own order of operations:
//   g += wd * p;  m = lerp(m, g, 1 - beta1);  v = beta2 * v + (1 - beta2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / bc2s + eps)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
    g = g * a.gscale + a.wd * p;
    m = m + (g - m) * a.omb1;
    v = a.beta2 * v + a.omb2 * g * g;
    const float denom = __fsqrt_rn(v) / a.bc2s + a.eps;
    p = p - (a.lr / a.bc1) * (m / denom);
}

__global__ void __launch_bounds__(BK_THREADS) bucket_adam_kernel(const AdamArgs a) {
    __shared__ int s_idx;
    const long long g0 = (long long)blockIdx.x * BK_CHUNK;
    const long long g = g0 + 4 * threadIdx.x;
    const int idx = bucket_find(a.ptable, a.n, g0, g < a.total ? g : g0, &s_idx);
    if (g >= a.total) return;
    float* pp = reinterpret_cast<float*>(a.ptable[3 * idx]);
    const long long local = g - a.ptable[3 * idx + 1];
    const long long left = a.ptable[3 * idx + 2] - local;
    if (!pp || left <= 0) return;
    const float* gp = a.gflat ? a.gflat + g : (reinterpret_cast<const float*>(a.gtable[3 * idx]) + local);
    if (!a.gflat && !a.gtable[3 * idx]) return;                 // parameter without a gradient: torch skips it
    const bool vec = left >= 4 && !(reinterpret_cast<uintptr_t>(pp) & 15) && !(reinterpret_cast<uintptr_t>(gp) & 15);
    float4 m4 = *reinterpret_cast<float4*>(a.m + g), v4 = *reinterpret_cast<float4*>(a.v + g);
    float pv[4], gv[4];
    float mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    const int cnt = left >= 4 ? 4 : (int)left;
    if (vec) {
        const float4 p4 = *reinterpret_cast<const float4*>(pp + local), g4 = *reinterpret_cast<const float4*>(gp);
        pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pv[e] = e < cnt ? pp[local + e] : 0.f;
            gv[e] = e < cnt ? gp[e] : 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (e < cnt) adam_one(pv[e], gv[e], mv[e], vv[e], a);
    if (vec) {
        *reinterpret_cast<float4*>(pp + local) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < cnt) pp[local + e] = pv[e];
    }
    *reinterpret_cast<float4*>(a.m + g) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *reinterpret_cast<float4*>(a.v + g) = make_float4(vv[0], vv[1], vv[2], vv[3]);
}

int check_bucket(const void* table, int n, long long total, const void* flat) {
    if (!table || n <= 0 || total <= 0 || (total & 3) || !flat || (reinterpret_cast<uintptr_t>(flat) & 15)) return IDV_EINVAL;
    if ((total + BK_CHUNK - 1) / BK_CHUNK > 0x7fffffffLL) return IDV_EINVAL;
    return IDV_OK;
}

}  // namespace

extern "C" int idv_bucket_gather(const long long* table, int n, long long total, float* flat, void* stream) {
    if (int rc = check_bucket(table, n, total, flat)) return rc;
    hipLaunchKernelGGL(bucket_copy_kernel<true>, dim3((unsigned)((total + BK_CHUNK - 1) / BK_CHUNK)), dim3(BK_THREADS), 0,
                       (hipStream_t)stream, table, n, total, flat, 1.0f);
    return idv_launch_status();
}

extern "C" int idv_bucket_scatter(const long long* table, int n, long long total, const float* flat, float scale, void* stream) {
    if (int rc = check_bucket(table, n, total, flat)) return rc;
    hipLaunchKernelGGL(bucket_copy_kernel<false>, dim3((unsigned)((total + BK_CHUNK - 1) / BK_CHUNK)), dim3(BK_THREADS), 0,
                       (hipStream_t)stream, table, n, total, const_cast<float*>(flat), scale);
    return idv_launch_status();
}

extern "C" int idv_bucket_adam(const long long* ptable, const long long* gtable, const float* gflat, float* exp_avg, float* exp_avg_sq,
                               int n, long long total, float lr, double beta1d, double beta2d, float eps, float weight_decay,
                               float bias_correction1, float bias_correction2_sqrt, float grad_scale, void* stream) {
    const float beta1 = (float)beta1d, beta2 = (float)beta2d;
    if (int rc = check_bucket(ptable, n, total, exp_avg)) return rc;
    if (!exp_avg_sq || (reinterpret_cast<uintptr_t>(exp_avg_sq) & 15) || (!gtable == !gflat)) return IDV_EINVAL;
    if (gflat && (reinterpret_cast<uintptr_t>(gflat) & 15)) return IDV_EINVAL;
    if (!(bias_correction1 > 0.f) || !(bias_correction2_sqrt > 0.f)) return IDV_EINVAL;
    // 1 - beta in double, THEN rounded: what torch hands its kernels (1 - 0.999f in float is off by 1.3e-5 relative)
    AdamArgs a{ptable, gtable, gflat, exp_avg, exp_avg_sq, n, total, lr, beta1, beta2, (float)(1.0 - (double)beta1d), (float)(1.0 - (double)beta2d),
               eps, weight_decay, bias_correction1, bias_correction2_sqrt, grad_scale};
    hipLaunchKernelGGL(bucket_adam_kernel, dim3((unsigned)((total + BK_CHUNK - 1) / BK_CHUNK)), dim3(BK_THREADS), 0,
                       (hipStream_t)stream, a);
    return idv_launch_status();
}
