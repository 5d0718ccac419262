// Minibatch mutual-information term of the CVAE / NVAE ELBO (complex_standard_vae_loss.mutual_information,
// model/pretrain_pvaes_loss.py:129-159, on cal_gaussian_prob :64-127):
//   lp(j; i, s, t) = log q(z[i, s, t] | x_j)        every sample against every posterior of the batch
//   MI = mean_{i,s,t} ( lp(i; i,s,t) - (logsumexp_j lp(j; i,s,t) - log B) )
// Off in the shipped recipe (mi_weight 0): small element-wise / reduction kernels, HBM-bound, no tiling.
// Buffers: posterior planar [2][H][Jp] with (miu, log_sigma, delta) at channel offsets (column b*Tp + t + 1); samples planar
// [2][zdim][Jpz] (column (b*ns + s)*Tp + t + 1); work = derived[4][zdim][T][B] (1/P, (R/P)_re, (R/P)_im, log terms) followed
// by rows[B*ns*T][B]: the log-probabilities, overwritten by d MI / d lp for the backward pass.
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

struct MiArgs {
    const float* lat; int H, Jp, o_miu, o_ls, o_dl;
    const float* z; int Jpz;
    int zdim, ns, B, T, Tp;
    float eps;
};

struct Derived { float s, dr, di, q, P, rp, D, E, m, Rr, Ri, L, a0, temp; bool guard; };

// cal_gaussian_prob :68-100 for one (b, t, h)
__device__ __forceinline__ Derived derive(float ls, float dr0, float di0, float e) {
    Derived d;
    d.s = expf(ls);
    d.a0 = sqrtf(dr0 * dr0 + di0 * di0 + e);
    d.temp = d.s * 0.90f / (d.a0 + e);
    d.guard = d.a0 >= d.s - 1e-3f;
    d.dr = d.guard ? dr0 * d.temp : dr0;
    d.di = d.guard ? di0 * d.temp : di0;
    d.q = d.dr * d.dr + d.di * d.di;
    d.P = d.s - d.q / (d.s + e);
    d.rp = 1.f / (d.P + e);
    d.D = d.s * d.P + e;
    d.Rr = d.dr / d.D;
    d.Ri = -d.di / d.D;
    d.E = d.s * d.P * d.s + e;
    d.m = d.rp - d.q / d.E;
    d.L = logf(d.m + e) + logf(d.rp + e);
    return d;
}

__device__ __forceinline__ void load_post(const MiArgs& a, int b, int t, int h, float& mr, float& mi, float& ls, float& dr, float& di) {
    const size_t col = (size_t)b * a.Tp + t + 1;
    const float* re = a.lat;
    const float* im = a.lat + (size_t)a.H * a.Jp;
    mr = re[(size_t)(a.o_miu + h) * a.Jp + col]; mi = im[(size_t)(a.o_miu + h) * a.Jp + col];
    ls = re[(size_t)(a.o_ls + h) * a.Jp + col];
    dr = re[(size_t)(a.o_dl + h) * a.Jp + col]; di = im[(size_t)(a.o_dl + h) * a.Jp + col];
}

__global__ __launch_bounds__(256) void mi_prep_kernel(MiArgs a, float* __restrict__ der) {
    const long long n = (long long)a.zdim * a.T * a.B;
    const size_t plane = (size_t)n;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx % a.B), t = (int)((idx / a.B) % a.T), h = (int)(idx / ((long long)a.B * a.T));
        float mr, mi, ls, dr, di;
        load_post(a, b, t, h, mr, mi, ls, dr, di);
        const Derived d = derive(ls, dr, di, a.eps);
        der[idx] = d.rp; der[plane + idx] = d.Rr; der[2 * plane + idx] = d.Ri; der[3 * plane + idx] = d.L;
    }
}

// rows[(i*ns + s)*T + t][j] = lp(j; i, s, t)      (:102-127)
__global__ __launch_bounds__(256) void mi_logp_kernel(MiArgs a, const float* __restrict__ der, float* __restrict__ rows) {
    const long long n = (long long)a.B * a.ns * a.T * a.B;
    const size_t plane = (size_t)a.zdim * a.T * a.B;
    const float* lre = a.lat;
    const float* lim = a.lat + (size_t)a.H * a.Jp;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % a.B);
        const long long row = idx / a.B;
        const int t = (int)(row % a.T);
        const int is = (int)(row / a.T);                                  // i*ns + s
        const size_t zc = (size_t)is * a.Tp + t + 1, pc = (size_t)j * a.Tp + t + 1;
        float quad = 0.f, logs = 0.f;
        for (int h = 0; h < a.zdim; ++h) {
            const size_t k = ((size_t)h * a.T + t) * a.B + j;
            const float rp = der[k], Rr = der[plane + k], Ri = der[2 * plane + k];
            logs += der[3 * plane + k];
            const float xr = a.z[(size_t)h * a.Jpz + zc] - lre[(size_t)(a.o_miu + h) * a.Jp + pc];
            const float xi = a.z[(size_t)(a.zdim + h) * a.Jpz + zc] - lim[(size_t)(a.o_miu + h) * a.Jp + pc];
            quad += (xr * xr - xi * xi) * Rr - 2.f * xr * xi * Ri - (xr * xr + xi * xi) * rp;
        }
        rows[idx] = 0.5f * logs + quad;
    }
}

// per row: log q(z|x_i) - (logsumexp_j - log B) summed into acc[0]; the row becomes d MI / d lp = (delta_ij - softmax_j) / rows
__global__ __launch_bounds__(256) void mi_rows_kernel(MiArgs a, float* __restrict__ rows, double* __restrict__ acc) {
    const long long n = (long long)a.B * a.ns * a.T;
    const float inv = 1.f / (float)n;
    double part = 0;
    for (long long row = blockIdx.x * (long long)blockDim.x + threadIdx.x; row < n; row += (long long)gridDim.x * blockDim.x) {
        float* r = rows + row * a.B;
        const int i = (int)(row / ((long long)a.ns * a.T));
        // (no loop vectorisation: it would emit packed-fp32 VALU instructions, which this library does not ship -- DESIGN.md 5.1)
        float mx = r[0];
#pragma clang loop vectorize(disable) interleave(disable)
        for (int j = 1; j < a.B; ++j) mx = fmaxf(mx, r[j]);
        float se = 0.f;
#pragma clang loop vectorize(disable) interleave(disable)
        for (int j = 0; j < a.B; ++j) se += expf(r[j] - mx);
        const float lse = mx + logf(se);
        part += (double)(r[i] - (lse - logf((float)a.B)));
#pragma clang loop vectorize(disable) interleave(disable)
        for (int j = 0; j < a.B; ++j) r[j] = ((j == i ? 1.f : 0.f) - expf(r[j] - lse)) * inv;
    }
    part = wave_sum_d(part);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, part);
}

__global__ void mi_final_kernel(const double* __restrict__ acc, double count, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(acc[0] / count);
}

// dz[i, s, t, h] = g * sum_j w[row][j] * d lp / d z
__global__ __launch_bounds__(256) void mi_bwd_z_kernel(MiArgs a, const float* __restrict__ der, const float* __restrict__ rows,
                                                       const float* __restrict__ gout, float* __restrict__ dz) {
    const long long n = (long long)a.zdim * a.B * a.ns * a.T;
    const size_t plane = (size_t)a.zdim * a.T * a.B;
    const float g = gout[0];
    const float* lre = a.lat;
    const float* lim = a.lat + (size_t)a.H * a.Jp;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % a.T);
        const int is = (int)((idx / a.T) % ((long long)a.B * a.ns));
        const int h = (int)(idx / ((long long)a.T * a.B * a.ns));
        const size_t zc = (size_t)is * a.Tp + t + 1;
        const float zr = a.z[(size_t)h * a.Jpz + zc], zi = a.z[(size_t)(a.zdim + h) * a.Jpz + zc];
        const float* w = rows + ((size_t)is * a.T + t) * a.B;
        float gr = 0.f, gi = 0.f;
        for (int j = 0; j < a.B; ++j) {
            const size_t k = ((size_t)h * a.T + t) * a.B + j, pc = (size_t)j * a.Tp + t + 1;
            const float rp = der[k], Rr = der[plane + k], Ri = der[2 * plane + k];
            const float xr = zr - lre[(size_t)(a.o_miu + h) * a.Jp + pc], xi = zi - lim[(size_t)(a.o_miu + h) * a.Jp + pc];
            gr += w[j] * 2.f * (xr * Rr - xi * Ri - xr * rp);
            gi += w[j] * 2.f * (-xi * Rr - xr * Ri - xi * rp);
        }
        dz[(size_t)h * a.Jpz + zc] = g * gr;
        dz[(size_t)(a.zdim + h) * a.Jpz + zc] = g * gi;
    }
}

// dlat(miu, log_sigma, delta)[j, t, h] += g * sum_{i,s} w[(i,s,t)][j] * d lp / d(.)   through the chain of derive()
__global__ __launch_bounds__(256) void mi_bwd_post_kernel(MiArgs a, const float* __restrict__ rows, const float* __restrict__ gout,
                                                          float* __restrict__ dlat) {
    const long long n = (long long)a.zdim * a.T * a.B;
    const float g = gout[0];
    float* dre = dlat;
    float* dim_ = dlat + (size_t)a.H * a.Jp;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % a.B), t = (int)((idx / a.B) % a.T), h = (int)(idx / ((long long)a.B * a.T));
        float mr, mi, ls, dr0, di0;
        load_post(a, j, t, h, mr, mi, ls, dr0, di0);
        const Derived d = derive(ls, dr0, di0, a.eps);
        float G_rp = 0.f, G_Rr = 0.f, G_Ri = 0.f, G_mr = 0.f, G_mi = 0.f, G_L = 0.f;
        for (int is = 0; is < a.B * a.ns; ++is) {
            const float w = rows[((size_t)is * a.T + t) * a.B + j];
            const size_t zc = (size_t)is * a.Tp + t + 1;
            const float xr = a.z[(size_t)h * a.Jpz + zc] - mr, xi = a.z[(size_t)(a.zdim + h) * a.Jpz + zc] - mi;
            G_rp -= w * (xr * xr + xi * xi);
            G_Rr += w * (xr * xr - xi * xi);
            G_Ri -= w * 2.f * xr * xi;
            G_mr -= w * 2.f * (xr * d.Rr - xi * d.Ri - xr * d.rp);
            G_mi -= w * 2.f * (-xi * d.Rr - xr * d.Ri - xi * d.rp);
            G_L += 0.5f * w;
        }
        const float e = a.eps;
        const float G_m = G_L / (d.m + e);
        G_rp += G_L / (d.rp + e) + G_m;
        float G_q = -G_m / d.E;
        const float G_E = G_m * d.q / (d.E * d.E);
        const float G_D = (-G_Rr * d.dr + G_Ri * d.di) / (d.D * d.D);
        float G_dr = G_Rr / d.D, G_di = -G_Ri / d.D;
        const float G_P = -G_rp * d.rp * d.rp + G_D * d.s + G_E * d.s * d.s;
        float G_s = G_D * d.P + G_E * 2.f * d.s * d.P + G_P * (1.f + d.q / ((d.s + e) * (d.s + e)));
        G_q -= G_P / (d.s + e);
        G_dr += 2.f * d.dr * G_q;
        G_di += 2.f * d.di * G_q;
        float G_dr0 = G_dr, G_di0 = G_di;
        if (d.guard) {
            const float G_t = G_dr * dr0 + G_di * di0;
            G_dr0 = G_dr * d.temp; G_di0 = G_di * d.temp;
            G_s += G_t * 0.90f / (d.a0 + e);
            const float G_a0 = -G_t * 0.90f * d.s / ((d.a0 + e) * (d.a0 + e));
            G_dr0 += G_a0 * dr0 / d.a0; G_di0 += G_a0 * di0 / d.a0;
        }
        const size_t col = (size_t)j * a.Tp + t + 1;
        dre[(size_t)(a.o_miu + h) * a.Jp + col] += g * G_mr;
        dim_[(size_t)(a.o_miu + h) * a.Jp + col] += g * G_mi;
        dre[(size_t)(a.o_ls + h) * a.Jp + col] += g * G_s * d.s;
        dre[(size_t)(a.o_dl + h) * a.Jp + col] += g * G_dr0;
        dim_[(size_t)(a.o_dl + h) * a.Jp + col] += g * G_di0;
    }
}

inline int grid_of(long long n) {
    long long gsz = (n + 255) / 256;
    return (int)(gsz > 8192 ? 8192 : (gsz < 1 ? 1 : gsz));
}

inline bool args_ok(const float* lat, int H, int Jp, int o_miu, int o_ls, int o_dl, const float* z, int Jpz, int zdim, int ns, int B,
                    int T, int Tp) {
    if (!lat || !z || zdim <= 0 || ns <= 0 || B <= 0 || T <= 0 || Tp <= T) return false;
    if (o_miu < 0 || o_ls < 0 || o_dl < 0 || o_miu + zdim > H || o_ls + zdim > H || o_dl + zdim > H) return false;
    return (long long)Jp >= (long long)B * Tp && (long long)Jpz >= (long long)B * ns * Tp;
}

}  // namespace

extern "C" long long idv_mi_work_floats(int B, int ns, int T, int zdim) {
    if (B <= 0 || ns <= 0 || T <= 0 || zdim <= 0) return 0;
    return 4LL * zdim * T * B + (long long)B * ns * T * B;
}

extern "C" int idv_mi_fwd(const float* lat, int H, int Jp, int o_miu, int o_ls, int o_dl, const float* z, int Jpz, int zdim, int ns,
                          int B, int T, int Tp, float eps, float* work, double* acc, float* out, void* stream) {
    if (!args_ok(lat, H, Jp, o_miu, o_ls, o_dl, z, Jpz, zdim, ns, B, T, Tp) || !work || !acc || !out) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    MiArgs a{lat, H, Jp, o_miu, o_ls, o_dl, z, Jpz, zdim, ns, B, T, Tp, eps};
    float* rows = work + 4LL * zdim * T * B;
    if (hipMemsetAsync(acc, 0, sizeof(double), st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(mi_prep_kernel, dim3(grid_of((long long)zdim * T * B)), dim3(256), 0, st, a, work);
    hipLaunchKernelGGL(mi_logp_kernel, dim3(grid_of((long long)B * ns * T * B)), dim3(256), 0, st, a, work, rows);
    hipLaunchKernelGGL(mi_rows_kernel, dim3(grid_of((long long)B * ns * T)), dim3(256), 0, st, a, rows, acc);
    hipLaunchKernelGGL(mi_final_kernel, dim3(1), dim3(64), 0, st, acc, (double)B * ns * T, out);
    return idv_launch_status();
}

extern "C" int idv_mi_bwd(const float* lat, int H, int Jp, int o_miu, int o_ls, int o_dl, const float* z, int Jpz, int zdim, int ns,
                          int B, int T, int Tp, float eps, const float* work, const float* gout, float* dlat, float* dz,
                          void* stream) {
    if (!args_ok(lat, H, Jp, o_miu, o_ls, o_dl, z, Jpz, zdim, ns, B, T, Tp) || !work || !gout || (!dlat && !dz)) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    MiArgs a{lat, H, Jp, o_miu, o_ls, o_dl, z, Jpz, zdim, ns, B, T, Tp, eps};
    const float* rows = work + 4LL * zdim * T * B;
    if (dz) hipLaunchKernelGGL(mi_bwd_z_kernel, dim3(grid_of((long long)zdim * B * ns * T)), dim3(256), 0, st, a, work, rows, gout, dz);
    if (dlat) hipLaunchKernelGGL(mi_bwd_post_kernel, dim3(grid_of((long long)zdim * T * B)), dim3(256), 0, st, a, rows, gout, dlat);
    return idv_launch_status();
}
