// Loss reductions: SI-SNR, STFT reconstruction losses, closed-form complex-Gaussian KL.
// Every reduction accumulates block partials into double atomics (work buffers are zeroed by the
// entry point on the same stream) and a one-thread finalise kernel writes the fp32 scalars.
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

__device__ __forceinline__ void block_add3(double a, double b, double c, double* dst) {
    __shared__ double sh[3][4];
    a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[0][w] = a; sh[1][w] = b; sh[2][w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0, sb = 0, sc = 0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) { sa += sh[0][q]; sb += sh[1][q]; sc += sh[2][q]; }
        atomicAdd(dst + 0, sa); atomicAdd(dst + 1, sb); atomicAdd(dst + 2, sc);
    }
    __syncthreads();
}

// per utterance: E = <s,s>, D = <e,s>, Q = <e,e>
__global__ __launch_bounds__(256) void sisnr_partial_kernel(const float* __restrict__ src, int src_ld, int src_div,
                                                            const float* __restrict__ est, int est_ld, int L,
                                                            double* __restrict__ work) {
    const int b = blockIdx.y;
    const float* s = src + (size_t)(b / src_div) * src_ld;
    const float* e = est + (size_t)b * est_ld;
    double E = 0, D = 0, Q = 0;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < L; n += gridDim.x * blockDim.x) {
        const float sv = s[n], ev = e[n];
        E += (double)sv * sv; D += (double)ev * sv; Q += (double)ev * ev;
    }
    block_add3(E, D, Q, work + (size_t)b * 3);
}

// sum over (part, c < C, f, b, t <= t_valid) of (a[ca0 + c] - b[cb0 + c])^2 between two planar activations: the skip-matching
// ("residual") term of model/nsvae_loss.py:363-446, torch.mean((connct - connct2).pow(2)) per skip connection
__global__ __launch_bounds__(256) void msd_partial_kernel(const float* __restrict__ a, int Ca, int ca0, int JpA,
                                                          const float* __restrict__ b, int Cb, int cb0, int JpB, int C, int F,
                                                          int B, int Tp, int t_valid, double* __restrict__ work) {
    const int row = blockIdx.y;                       // (part, c, f)
    const int f = row % F, c = (row / F) % C, part = row / (F * C);
    const float* pa = a + ((size_t)(part * Ca + ca0 + c) * F + f) * JpA;
    const float* pb = b + ((size_t)(part * Cb + cb0 + c) * F + f) * JpB;
    const int J = B * Tp;
    double acc = 0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        if (tp < 1 || tp > t_valid) continue;
        const float d = pa[j] - pb[j];
        acc += (double)d * d;
    }
    block_add3(acc, 0.0, 0.0, work);
}

__global__ void msd_final_kernel(const double* __restrict__ work, double count, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(work[0] / count);
}

// s_target = (D/(E+eps)) s ; |s_t|^2 = a^2 E ; |e - s_t|^2 = Q - 2aD + a^2 E   (sisnr_loss.py:10-18)
__global__ void sisnr_final_kernel(const double* __restrict__ work, int B, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double eps = 1e-8;
    double acc = 0;
    for (int b = 0; b < B; ++b) {
        const double E = work[b * 3], D = work[b * 3 + 1], Q = work[b * 3 + 2];
        const double a = D / (E + eps);
        const double st = a * a * E;
        double en = Q - 2 * a * D + st;
        if (en < 0) en = 0;
        acc += 10.0 * log10(st / (en + eps) + eps);
    }
    out[0] = (float)(-acc / B);
}

// compute_sisdr (utils/eval_metrics.py:49-64), one value per utterance: a = (eps + <s,e>) / (<s,s> + eps),
// SI-SDR = 10 log10((eps + a^2 <s,s>) / (eps + |e - a s|^2)), eps = float32 machine epsilon
__global__ void sisdr_final_kernel(const double* __restrict__ work, int B, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double eps = 1.1920928955078125e-07;
    const double E = work[b * 3], D = work[b * 3 + 1], Q = work[b * 3 + 2];
    const double a = (eps + D) / (E + eps);
    const double sss = a * a * E;
    double snn = Q - 2 * a * D + sss;
    if (snn < 0) snn = 0;
    out[b] = (float)(10.0 * log10((eps + sss) / (eps + snn)));
}

// out[b][n] = mean_s x[b*ns + s][n]   (test_se_cvaefinetune.py:309-311: torch.mean over the sampled waveforms)
__global__ void mean_over_samples_kernel(const float* __restrict__ x, int ns, int B, int L, float* __restrict__ out) {
    const long long n = (long long)B * L;
    const float inv = 1.0f / ns;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / L, k = idx - b * L;
        float acc = 0.f;
        for (int s = 0; s < ns; ++s) acc += x[(b * ns + s) * L + k];
        out[idx] = acc * inv;
    }
}

__global__ __launch_bounds__(256) void recon_partial_kernel(const float* __restrict__ pred_c, const float* __restrict__ ori,
                                                            long long sb, long long sf, long long st, long long sr,
                                                            int ori_div, int B, int F, int T, double* __restrict__ work) {
    const long long n = (long long)B * F * T;
    double cpx = 0, mag = 0;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const float pr = pred_c[idx * 2], pi = pred_c[idx * 2 + 1];
        const long long o = (long long)(b / ori_div) * sb + f * sf + t * st;
        const float orr = ori[o], oi = ori[o + sr];
        const float dr = pr - orr, di = pi - oi;
        cpx += (double)(dr * dr) + (double)(di * di);
        const float pm = sqrtf(pr * pr + pi * pi + 1e-6f);
        const float om = sqrtf(orr * orr + orr * orr + 1e-6f);   // real part twice, nsvae_loss.py:783
        mag += (double)((pm - om) * (pm - om));
    }
    block_add3(cpx, mag, 0.0, work);
}

__global__ void recon_final_kernel(const double* __restrict__ work, double bt, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = (float)(work[0] / bt);
    out[1] = (float)(work[1] / bt);
}

struct LatRef {
    const float* q; int H, Jp, o_miu, o_ls, o_dl;
};

__device__ __forceinline__ void guard_delta(float sg, float& dr, float& di, float eps) {
    const float a = sqrtf(dr * dr + di * di + eps);
    const float sc = sg * 0.99f / (a + eps);
    if (a >= sg - 1e-3f) { dr *= sc; di *= sc; }
}

// sum over (b,t,h) of the per-element KL summand; mean and "- zdim" applied in the finalise step
__global__ __launch_bounds__(256) void ckl_partial_kernel(LatRef q1, LatRef q2, int zdim, float eps, int B, int T, int Tp,
                                                          double* __restrict__ work) {
    const long long n = (long long)B * zdim * T;
    double acc = 0;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int h = (int)((idx / T) % zdim);
        const int b = (int)(idx / ((long long)T * zdim));
        const size_t j = (size_t)b * Tp + t + 1;
        const float* r1 = q1.q; const float* i1 = q1.q + (size_t)q1.H * q1.Jp;
        const float m1r = r1[(size_t)(q1.o_miu + h) * q1.Jp + j], m1i = i1[(size_t)(q1.o_miu + h) * q1.Jp + j];
        const float s1 = expf(r1[(size_t)(q1.o_ls + h) * q1.Jp + j]);
        float d1r = r1[(size_t)(q1.o_dl + h) * q1.Jp + j], d1i = i1[(size_t)(q1.o_dl + h) * q1.Jp + j];
        float m2r = 0.f, m2i = 0.f, s2 = 1.f, d2r = 0.f, d2i = 0.f;
        if (q2.q) {
            const float* r2 = q2.q; const float* i2 = q2.q + (size_t)q2.H * q2.Jp;
            m2r = r2[(size_t)(q2.o_miu + h) * q2.Jp + j]; m2i = i2[(size_t)(q2.o_miu + h) * q2.Jp + j];
            s2 = expf(r2[(size_t)(q2.o_ls + h) * q2.Jp + j]);
            d2r = r2[(size_t)(q2.o_dl + h) * q2.Jp + j]; d2i = i2[(size_t)(q2.o_dl + h) * q2.Jp + j];
        }
        guard_delta(s1, d1r, d1i, eps);
        guard_delta(s2, d2r, d2i, eps);
        const float a1 = d1r * d1r + d1i * d1i, a2 = d2r * d2r + d2i * d2i;
        const float logdet1 = logf(0.25f * (s1 * s1 - a1) + eps);
        const float logdet2 = logf(0.25f * (s2 * s2 - a2) + eps);
        const float coeff = 2.0f / (s2 * s2 - a2 + eps);
        const float trace = s1 * s2 - d2r * d1r - d2i * d1i;
        const float dr = m2r - m1r, di = m2i - m1i;
        const float quad = dr * dr * (s2 - d2r) - 2.f * d2i * dr * di + di * di * (s2 + d2r);
        acc += (double)(coeff * (trace + quad) + logdet2 - logdet1);
    }
    block_add3(acc, 0.0, 0.0, work);
}

__global__ void ckl_final_kernel(const double* __restrict__ work, double bt, int zdim, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = (float)(0.5 * work[0] / bt - zdim);
}

// work[h*2 + ri] += sum_{b,t} (miu1 - miu2)^2 ; grid.y = zdim*2
__global__ __launch_bounds__(256) void miu_dist_partial_kernel(LatRef q1, LatRef q2, int B, int T, int Tp,
                                                               double* __restrict__ work) {
    const int h = blockIdx.y >> 1, ri = blockIdx.y & 1;
    const float* a = q1.q + ((size_t)ri * q1.H + q1.o_miu + h) * q1.Jp;
    const float* b2 = q2.q + ((size_t)ri * q2.H + q2.o_miu + h) * q2.Jp;
    double acc = 0;
    const int n = B * T;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const size_t j = (size_t)(idx / T) * Tp + (idx % T) + 1;
        const float dd = a[j] - b2[j];
        acc += (double)(dd * dd);
    }
    block_add3(acc, 0.0, 0.0, work + 3 * (size_t)blockIdx.y);
}

__global__ void miu_dist_final_kernel(const double* __restrict__ work, int n, double bt, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0;
    for (int q = 0; q < n; ++q) s += work[3 * q] / bt;
    out[0] = (float)sqrt(s);
}

inline int grid_for(long long n, int cap = 1024) {
    long long g = (n + 255) / 256;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int idv_sisnr(const float* source, int src_ld, int src_div, const float* est, int est_ld, int B, int L,
                         double* work, float* out, void* stream) {
    if (!source || !est || !work || !out || B <= 0 || L <= 0 || src_div < 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, sizeof(double) * 3 * B, st) != hipSuccess) return IDV_ELAUNCH;
    int gx = grid_for(L, 64);
    hipLaunchKernelGGL(sisnr_partial_kernel, dim3(gx, B), dim3(256), 0, st, source, src_ld, src_div, est, est_ld, L, work);
    hipLaunchKernelGGL(sisnr_final_kernel, dim3(1), dim3(64), 0, st, work, B, out);
    return idv_launch_status();
}

extern "C" int idv_sisdr(const float* ref, int ref_ld, const float* est, int est_ld, int B, int L, double* work, float* out,
                         void* stream) {
    if (!ref || !est || !work || !out || B <= 0 || L <= 0) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, sizeof(double) * 3 * B, st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(sisnr_partial_kernel, dim3(grid_for(L, 64), B), dim3(256), 0, st, ref, ref_ld, 1, est, est_ld, L, work);
    hipLaunchKernelGGL(sisdr_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, work, B, out);
    return idv_launch_status();
}

// mean squared difference between C channels of two planar activations (both parts): out[0] = mean over [B, C, F, T, 2] of
// (a[:, ca0:ca0+C] - b[:, cb0:cb0+C])^2 -- one term of residual_loss (model/nsvae_loss.py:363-446).  work: 3 doubles.
extern "C" int idv_msd(const float* a, int Ca, int ca0, int JpA, const float* b, int Cb, int cb0, int JpB, int C, int F, int B,
                       int Tp, int t_valid, double* work, float* out, void* stream) {
    if (!a || !b || !work || !out || C <= 0 || F <= 0 || B <= 0 || Tp <= 1 || t_valid < 1 || t_valid >= Tp || ca0 < 0 || cb0 < 0 ||
        ca0 + C > Ca || cb0 + C > Cb || JpA < B * Tp || JpB < B * Tp)
        return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, sizeof(double) * 3, st) != hipSuccess) return IDV_ELAUNCH;
    int gx = (B * Tp + 255) / 256;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(msd_partial_kernel, dim3(gx, 2 * C * F), dim3(256), 0, st, a, Ca, ca0, JpA, b, Cb, cb0, JpB, C, F, B, Tp,
                       t_valid, work);
    hipLaunchKernelGGL(msd_final_kernel, dim3(1), dim3(64), 0, st, work, 2.0 * C * F * (double)B * t_valid, out);
    return idv_launch_status();
}

extern "C" int idv_mean_over_samples(const float* x, int ns, int B, int L, float* out, void* stream) {
    if (!x || !out || ns < 1 || B <= 0 || L <= 0) return IDV_EINVAL;
    hipLaunchKernelGGL(mean_over_samples_kernel, dim3(grid_for((long long)B * L, 4096)), dim3(256), 0, (hipStream_t)stream, x, ns,
                       B, L, out);
    return idv_launch_status();
}

extern "C" int idv_recon_loss(const float* pred_c, const float* ori, long long sb, long long sf, long long st_, long long sr,
                              int ori_div, int B, int F, int T, double* work, float* out, void* stream) {
    if (!pred_c || !ori || !work || !out || B <= 0 || F <= 0 || T <= 0 || ori_div < 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, sizeof(double) * 3, st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(recon_partial_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, pred_c, ori, sb, sf, st_,
                       sr, ori_div, B, F, T, work);
    hipLaunchKernelGGL(recon_final_kernel, dim3(1), dim3(64), 0, st, work, (double)B * T, out);
    return idv_launch_status();
}

extern "C" int idv_ckl(const float* q1, int H1, int Jp1, int o1_miu, int o1_ls, int o1_dl, const float* q2, int H2, int Jp2,
                       int o2_miu, int o2_ls, int o2_dl, int zdim, float eps, int B, int T, int Tp, double* work, float* out,
                       void* stream) {
    if (!q1 || !work || !out || zdim <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, sizeof(double) * 3, st) != hipSuccess) return IDV_ELAUNCH;
    LatRef a{q1, H1, Jp1, o1_miu, o1_ls, o1_dl}, b{q2, H2, Jp2, o2_miu, o2_ls, o2_dl};
    hipLaunchKernelGGL(ckl_partial_kernel, dim3(grid_for((long long)B * zdim * T)), dim3(256), 0, st, a, b, zdim, eps, B, T,
                       Tp, work);
    hipLaunchKernelGGL(ckl_final_kernel, dim3(1), dim3(64), 0, st, work, (double)B * T, zdim, out);
    return idv_launch_status();
}

extern "C" int idv_miu_dist(const float* q1, int H1, int Jp1, int off1, const float* q2, int H2, int Jp2, int off2, int zdim,
                            int B, int T, int Tp, double* work, float* out, void* stream) {
    if (!q1 || !q2 || !work || !out || zdim <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, sizeof(double) * 3 * 2 * zdim, st) != hipSuccess) return IDV_ELAUNCH;
    LatRef a{q1, H1, Jp1, off1, 0, 0}, b{q2, H2, Jp2, off2, 0, 0};
    hipLaunchKernelGGL(miu_dist_partial_kernel, dim3(grid_for((long long)B * T, 16), 2 * zdim), dim3(256), 0, st, a, b, B, T,
                       Tp, work);
    hipLaunchKernelGGL(miu_dist_final_kernel, dim3(1), dim3(64), 0, st, work, 2 * zdim, (double)B * T, out);
    return idv_launch_status();
}
