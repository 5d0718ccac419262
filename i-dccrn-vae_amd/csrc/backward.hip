// Backward (gradient) kernels of the bandwidth-bound operators: train-mode ComplexBatchNormal + PReLU, the mask,
// ISTFT / STFT framing adjoints, reparameterisation, and the loss reductions.  They are what torch.autograd runs behind
// `loss.backward()` in the reference's train steps (supervised_dccrn/train.py:239-243, pretrained_vaes/train.py:296-301,
// train_nsvae.py:557-561, train_second_phase_decoder.py:420-433); each kernel cites the forward lines it differentiates.
// All planar gradients keep the layout invariant of the activations: guard columns (tp == 0, tp > t_valid) are zero.
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

inline int grid_for(long long n, int cap = 8192) {
    long long g = (n + 255) / 256;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

// ------------------------------------------------------------------------------------------------------------------
// ComplexBatchNormal (train) + PReLU backward.  Forward (model/complex_progress.py:131-209, pvae_module.py:64-68):
//   mu = mean(y), V = cov(y - mu) + eps, Z = Gamma W(V), u = Z (y - mu) + beta, z = PReLU(u).
// Pass 1: per-channel sums of du = dz * PReLU'(u) and du (x) y;  pass 2 (one thread per channel): dGamma, dbeta, dV and
// the coefficients of  dy = Z^T du + A (y - mu) + c;  pass 3 applies them.
constexpr int NSUM = 8;

__global__ __launch_bounds__(256) void cbn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                             const float* __restrict__ fold, const float* __restrict__ slope_p,
                                                             int C, int F, int B, int Tp, int Jp, int t_valid,
                                                             double* __restrict__ sums) {
    const int row = blockIdx.y, c = row / F;
    const float* z = fold + (size_t)c * 6;
    const float Zrr = z[0], Zri = z[1], Zir = z[2], Zii = z[3], sr = z[4], si = z[5];
    const bool act = slope_p != nullptr;
    const float slope = act ? *slope_p : 1.0f;
    const size_t ro = (size_t)row * Jp, io = ((size_t)C * F + row) * Jp;
    const int J = B * Tp;
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        if (tp < 1 || tp > t_valid) continue;
        const float yr = y[ro + j], yi = y[io + j];
        float dr = dz[ro + j], di = dz[io + j];
        if (act) {
            const float ur = Zrr * yr + Zri * yi + sr, ui = Zir * yr + Zii * yi + si;
            if (!(ur > 0.f)) { s[6] += (double)dr * ur; dr *= slope; }
            if (!(ui > 0.f)) { s[6] += (double)di * ui; di *= slope; }
        }
        s[0] += dr; s[1] += di;
        s[2] += (double)dr * yr; s[3] += (double)dr * yi;
        s[4] += (double)di * yr; s[5] += (double)di * yi;
    }
    __shared__ double sh[7][4];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const double v = wave_sum_d(s[q]);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7)
        atomicAdd(&sums[(size_t)c * NSUM + threadIdx.x], sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// one block; thread c handles channel c (looping), then the shared PReLU slope gradient is reduced in a fixed order
__global__ __launch_bounds__(256) void cbn_bwd_finalize_kernel(const double* __restrict__ sums, double count,
                                                               const float* __restrict__ moments, const float* __restrict__ g_rr,
                                                               const float* __restrict__ g_ri, const float* __restrict__ g_ii,
                                                               int C, float* __restrict__ coef, float* __restrict__ dg_rr,
                                                               float* __restrict__ dg_ri, float* __restrict__ dg_ii,
                                                               float* __restrict__ db_r, float* __restrict__ db_i,
                                                               float* __restrict__ dslope, float pscale) {
    __shared__ double shs[256];
    double sl = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double* s = sums + (size_t)c * NSUM;
        const double N = count, eps = 1e-5;
        const double mur = moments[c], mui = moments[C + c];
        const double Vrr = moments[2 * C + c], Vri = moments[3 * C + c], Vii = moments[4 * C + c];
        const double grr = g_rr[c], gri = g_ri[c], gii = g_ii[c];
        // forward whitening matrix (complex_progress.py:178-190)
        const double delta = Vrr * Vii - Vri * Vri + eps;
        const bool clamped = delta < 1e-8;
        const double dc = clamped ? 1e-8 : delta;
        const double sq = sqrt(dc);
        const double ta = Vrr + Vii + 2.0 * sq + eps;
        const double tt = sqrt(ta);
        const double q = sq * tt + eps;
        const double inv = 1.0 / q;
        const double Wrr = (Vii + sq) * inv, Wii = (Vrr + sq) * inv, Wri = -Vri * inv;
        const double Zrr = grr * Wrr + gri * Wri, Zri = grr * Wri + gri * Wii;
        const double Zir = gri * Wrr + gii * Wri, Zii = gri * Wri + gii * Wii;
        // dZ = sum du (x) (y - mu)
        const double Sr = s[0], Si = s[1];
        const double dZrr = s[2] - mur * Sr, dZri = s[3] - mui * Sr, dZir = s[4] - mur * Si, dZii = s[5] - mui * Si;
        dg_rr[c] = (float)(pscale * (dZrr * Wrr + dZri * Wri));
        dg_ri[c] = (float)(pscale * (dZrr * Wri + dZri * Wii + dZir * Wrr + dZii * Wri));
        dg_ii[c] = (float)(pscale * (dZir * Wri + dZii * Wii));
        db_r[c] = (float)(pscale * Sr);
        db_i[c] = (float)(pscale * Si);
        const double dWrr = dZrr * grr + dZir * gri;
        const double dWri = dZrr * gri + dZri * grr + dZir * gii + dZii * gri;
        const double dWii = dZri * gri + dZii * gii;
        // W(V) in reverse
        double dVrr = 0, dVri = 0, dVii = 0, dsq = 0, dinv = 0;
        dVii += dWrr * inv; dsq += dWrr * inv; dinv += dWrr * (Vii + sq);
        dVrr += dWii * inv; dsq += dWii * inv; dinv += dWii * (Vrr + sq);
        dVri += -dWri * inv; dinv += -dWri * Vri;
        const double dq = -dinv * inv * inv;
        dsq += dq * tt;
        const double dtt = dq * sq;
        const double dta = dtt / (2.0 * tt);
        dVrr += dta; dVii += dta; dsq += 2.0 * dta;
        const double ddc = dsq / (2.0 * sq);
        const double ddelta = clamped ? 0.0 : ddc;
        dVrr += ddelta * Vii; dVii += ddelta * Vrr; dVri += -2.0 * ddelta * Vri;
        float* k = coef + (size_t)c * 12;
        k[0] = (float)Zrr; k[1] = (float)Zri; k[2] = (float)Zir; k[3] = (float)Zii;
        k[4] = (float)(2.0 * dVrr / N); k[5] = (float)(dVri / N); k[6] = (float)(2.0 * dVii / N);
        k[7] = (float)(-(Zrr * Sr + Zir * Si) / N);
        k[8] = (float)(-(Zri * Sr + Zii * Si) / N);
        k[9] = (float)mur; k[10] = (float)mui; k[11] = 0.f;
        sl += s[6];
    }
    shs[threadIdx.x] = sl;
    __syncthreads();
    if (threadIdx.x == 0 && dslope) {
        double t = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) t += shs[i];
        dslope[0] = (float)(pscale * t);
    }
}

__global__ void cbn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y, const float* __restrict__ fold,
                                     const float* __restrict__ coef, const float* __restrict__ slope_p, int C, int F, int B,
                                     int Tp, int Jp, int t_valid, float* __restrict__ dy) {
    const int row = blockIdx.y, c = row / F;
    const float* z = fold + (size_t)c * 6;
    const float Frr = z[0], Fri = z[1], Fir = z[2], Fii = z[3], sr = z[4], si = z[5];
    const float* k = coef + (size_t)c * 12;
    const float Zrr = k[0], Zri = k[1], Zir = k[2], Zii = k[3], Arr = k[4], Ari = k[5], Aii = k[6], cr = k[7], ci = k[8];
    const float mur = k[9], mui = k[10];
    const bool act = slope_p != nullptr;
    const float slope = act ? *slope_p : 1.0f;
    const size_t ro = (size_t)row * Jp, io = ((size_t)C * F + row) * Jp;
    const int J = B * Tp;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        float or_ = 0.f, oi = 0.f;
        if (tp >= 1 && tp <= t_valid) {
            const float yr = y[ro + j], yi = y[io + j];
            float dr = dz[ro + j], di = dz[io + j];
            if (act) {
                const float ur = Frr * yr + Fri * yi + sr, ui = Fir * yr + Fii * yi + si;
                if (!(ur > 0.f)) dr *= slope;
                if (!(ui > 0.f)) di *= slope;
            }
            const float cyr = yr - mur, cyi = yi - mui;
            or_ = Zrr * dr + Zir * di + Arr * cyr + Ari * cyi + cr;
            oi = Zri * dr + Zii * di + Aii * cyi + Ari * cyr + ci;
        }
        dy[ro + j] = or_;
        dy[io + j] = oi;
    }
}

// out-of-place normalise + PReLU (the in-place idv_cbn_apply_prelu keeps only the result; training keeps y for backward)
__global__ void cbn_apply_prelu_to_kernel(const float* __restrict__ y, const float* __restrict__ fold,
                                          const float* __restrict__ slope_p, int C, int F, int B, int Tp, int Jp, int t_valid,
                                          float* __restrict__ out) {
    const int row = blockIdx.y, c = row / F;
    const float* z = fold + (size_t)c * 6;
    const float Zrr = z[0], Zri = z[1], Zir = z[2], Zii = z[3], sr = z[4], si = z[5];
    const float slope = slope_p ? *slope_p : 1.0f;
    const size_t ro = (size_t)row * Jp, io = ((size_t)C * F + row) * Jp;
    const int J = B * Tp;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        float yr = 0.f, yi = 0.f;
        if (tp >= 1 && tp <= t_valid) {
            const float r = y[ro + j], im = y[io + j];
            yr = Zrr * r + Zri * im + sr;
            yi = Zir * r + Zii * im + si;
            yr = yr >= 0.f ? yr : slope * yr;
            yi = yi >= 0.f ? yi : slope * yi;
        }
        out[ro + j] = yr;
        out[io + j] = yi;
    }
}

// ---- the same two element-wise passes writing, besides the planar result, its split-bf16 IMAGE (what the bf16x3 training
// mode's conv kernels read): one workgroup row = (octet of 4 channels, frequency row), a thread owns a column and the 8
// planes of its slot, so every store is whole 16-byte slots (the separate planar -> image pass re-read the result: 1 of the
// 11 activation-sized memory passes of a training block).  Needs C % 4 == 0.
__device__ __forceinline__ void img_slot_store(unsigned short* __restrict__ img, long long lo_off, size_t slot, const float (&v)[8]) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float x0 = v[2 * w], x1 = v[2 * w + 1];
        const unsigned u0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u, u1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
        hw[w] = (u0 >> 16) | u1;
        const float r0 = x0 - __builtin_bit_cast(float, u0), r1 = x1 - __builtin_bit_cast(float, u1);
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 pk = {(__bf16)r0, (__bf16)r1};
        lw[w] = __builtin_bit_cast(unsigned, pk);
    }
    *(uint4*)(img + slot * 8) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *(uint4*)(img + lo_off + slot * 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

__device__ __forceinline__ void img_edge_zero(unsigned short* __restrict__ img, long long lo_off, long long nslots) {
    // the zero slot in front of each plane (rows outside [0, F)) and 8 zero slots behind it (tshift-0 consumers), as
    // planar_to_image_kernel (image.hip)
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        const int t = threadIdx.x;
        if (t < 2) *(uint4*)(img - 8 * 8 + (t ? lo_off : 0)) = make_uint4(0u, 0u, 0u, 0u);
        if (t >= 32 && t < 48) *(uint4*)(img + (nslots + (t & 7)) * 8 + ((t & 8) ? lo_off : 0)) = make_uint4(0u, 0u, 0u, 0u);
    }
}

__global__ void cbn_apply_prelu_to_img_kernel(const float* __restrict__ y, const float* __restrict__ fold,
                                              const float* __restrict__ slope_p, int C, int F, int B, int Tp, int Jp, int t_valid,
                                              float* __restrict__ out, unsigned short* __restrict__ img, long long lo_off) {
    const int o = blockIdx.y / F, f = blockIdx.y - o * F;
    const float slope = slope_p ? *slope_p : 1.0f;
    const int J = B * Tp;
    img_edge_zero(img, lo_off, (long long)(C / 4) * F * Jp);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Jp; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        const bool keep = j < J && tp >= 1 && tp <= t_valid;
        float v[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = 4 * o + w;
            const float* z = fold + (size_t)c * 6;
            const size_t ro = ((size_t)c * F + f) * Jp + j, io = ((size_t)(C + c) * F + f) * Jp + j;
            float yr = 0.f, yi = 0.f;
            if (keep) {
                const float r = y[ro], im = y[io];
                yr = z[0] * r + z[1] * im + z[4];
                yi = z[2] * r + z[3] * im + z[5];
                yr = yr >= 0.f ? yr : slope * yr;
                yi = yi >= 0.f ? yi : slope * yi;
            }
            if (j < J) {
                out[ro] = yr;
                out[io] = yi;
            }
            v[2 * w] = yr;
            v[2 * w + 1] = yi;
        }
        img_slot_store(img, lo_off, ((size_t)o * F + f) * Jp + j, v);
    }
}

__global__ void cbn_bwd_apply_img_kernel(const float* __restrict__ dz, const float* __restrict__ y, const float* __restrict__ fold,
                                         const float* __restrict__ coef, const float* __restrict__ slope_p, int C, int F, int B,
                                         int Tp, int Jp, int t_valid, float* __restrict__ dy, unsigned short* __restrict__ img,
                                         long long lo_off) {
    const int o = blockIdx.y / F, f = blockIdx.y - o * F;
    const bool act = slope_p != nullptr;
    const float slope = act ? *slope_p : 1.0f;
    const int J = B * Tp;
    img_edge_zero(img, lo_off, (long long)(C / 4) * F * Jp);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Jp; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        const bool keep = j < J && tp >= 1 && tp <= t_valid;
        float v[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = 4 * o + w;
            const float* z = fold + (size_t)c * 6;
            const float* k = coef + (size_t)c * 12;
            const size_t ro = ((size_t)c * F + f) * Jp + j, io = ((size_t)(C + c) * F + f) * Jp + j;
            float or_ = 0.f, oi = 0.f;
            if (keep) {
                const float yr = y[ro], yi = y[io];
                float dr = dz[ro], di = dz[io];
                if (act) {
                    const float ur = z[0] * yr + z[1] * yi + z[4], ui = z[2] * yr + z[3] * yi + z[5];
                    if (!(ur > 0.f)) dr *= slope;
                    if (!(ui > 0.f)) di *= slope;
                }
                const float cyr = yr - k[9], cyi = yi - k[10];
                or_ = k[0] * dr + k[2] * di + k[4] * cyr + k[5] * cyi + k[7];
                oi = k[1] * dr + k[3] * di + k[6] * cyi + k[5] * cyr + k[8];
            }
            if (j < J) {
                dy[ro] = or_;
                dy[io] = oi;
            }
            v[2 * w] = or_;
            v[2 * w + 1] = oi;
        }
        img_slot_store(img, lo_off, ((size_t)o * F + f) * Jp + j, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// mask branch (pvae_module.py:224-234): P = tanh|M| * X * M/|M|.  With G = dL/dP, u = M/|M|, q = X u, g = tanh|M|:
//   dL/dM = (g' - g/|M|) (G . q) u + (g/|M|) conj(X) G          (g' = 1 - g^2)
__global__ void mask_bwd_kernel(const float* __restrict__ mask, const float* __restrict__ X, int x_div, int JpX,
                                const float* __restrict__ dpred, const float* __restrict__ dpred_c, int F, int B, int T, int Tp,
                                int Jp, float* __restrict__ dmask, float* __restrict__ dX) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t jm = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        const size_t jx = (size_t)f * JpX + (size_t)(b / x_div) * Tp + t + 1;
        const float mr = mask[jm], mi = mask[(size_t)F * Jp + jm];
        const float xr = X[jx], xi = X[(size_t)F * JpX + jx];
        float Gr = 0.f, Gi = 0.f;
        if (dpred) { Gr += dpred[jm]; Gi += dpred[(size_t)F * Jp + jm]; }
        if (dpred_c) { Gr += dpred_c[idx * 2]; Gi += dpred_c[idx * 2 + 1]; }
        const float mm = sqrtf(mr * mr + mi * mi);
        const float g = tanhf(mm);
        float ur = 1.f, ui = 0.f, h = 1.f;
        if (mm > 0.f) {
            const float inv = 1.0f / mm;
            ur = mr * inv; ui = mi * inv; h = g * inv;
        }
        const float qr = xr * ur - xi * ui, qi = xr * ui + xi * ur;
        const float gq = Gr * qr + Gi * qi;
        const float k = (1.0f - g * g) - h;
        dmask[jm] = k * gq * ur + h * (Gr * xr + Gi * xi);
        dmask[(size_t)F * Jp + jm] = k * gq * ui + h * (Gi * xr - Gr * xi);
        if (dX) {       // P = (g u) X  ->  dL/dX = conj(g u) G     (x_div == 1)
            const float cr = g * ur, ci = g * ui;
            dX[jx] = Gr * cr + Gi * ci;
            dX[(size_t)F * JpX + jx] = Gi * cr - Gr * ci;
        }
    }
}

__global__ void zero_guard_cols_kernel(float* __restrict__ act, int planes, int B, int T, int Tp, int Jp) {
    // columns tp == 0 and tp > T of every utterance
    const int ng = Tp - T;
    const long long n = (long long)planes * B * ng;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % ng);
        const int b = (int)((idx / ng) % B);
        const long long pl = idx / ((long long)ng * B);
        const int tp = q == 0 ? 0 : T + q;
        act[(size_t)pl * Jp + (size_t)b * Tp + tp] = 0.f;
    }
}

// interleaved [B][F][T][2] -> planar [2][F][Jp]   (adjoint of planar_to_complex; also packs foreign complex input)
__global__ void complex_to_planar_kernel(const float* __restrict__ in_c, float* __restrict__ act, int F, int B, int T, int Tp,
                                         int Jp) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t j = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        act[j] = in_c[idx * 2];
        act[(size_t)F * Jp + j] = in_c[idx * 2 + 1];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// ISTFT.forward adjoint (pvae_module.py:38-42): y[s] = env_inv[s+half] * sum_t frames[s+half-hop*t-left][t]
//   ->  dframes[k][t] = dy[hop*t + left + k - half] * env_inv[hop*t + left + k]
__global__ void istft_ola_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ env_inv, int B, int n_fft, int win,
                                     int hop, int T, int Tp, int Jp, float* __restrict__ dframes) {
    const int Lout = hop * (T - 1);
    const int left = (n_fft - win) / 2, half = n_fft / 2;
    const int k = blockIdx.y;
    const int J = B * Tp;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int b = j / Tp, tp = j - b * Tp;
        float v = 0.f;
        if (tp >= 1 && tp <= T) {
            const int p = hop * (tp - 1) + left + k;      // position in the padded signal
            const int s = p - half;
            if (s >= 0 && s < Lout) v = dy[(size_t)b * Lout + s] * env_inv[p];
        }
        dframes[(size_t)k * Jp + j] = v;
    }
}

// STFT.forward framing adjoint (pvae_module.py:21-27): frames[k][t] = xp[hop*t + left + k], xp = reflect-pad(x, half)
//   ->  dx[s] = sum over padded positions p that read x[s] of  sum_t dframes[p - hop*t - left][t]
__device__ __forceinline__ float frames_ola(const float* __restrict__ dfr, int b, long long p, int win, int hop, int T, int Tp,
                                            int Jp, int left) {
    const long long q = p - left;               // relative to the window start of frame 0
    if (q < 0) return 0.f;
    long long t_hi = q / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    long long t_lo = (q - win + hop) / hop;
    if (q - win + 1 <= 0) t_lo = 0;
    if (t_lo < 0) t_lo = 0;
    float acc = 0.f;
    for (long long t = t_lo; t <= t_hi; ++t) {
        const long long k = q - hop * t;
        if (k >= 0 && k < win) acc += dfr[(size_t)k * Jp + (size_t)b * Tp + t + 1];
    }
    return acc;
}

__global__ void stft_frames_bwd_kernel(const float* __restrict__ dfr, int B, int L, int n_fft, int win, int hop, int T, int Tp,
                                       int Jp, float* __restrict__ dx) {
    const int left = (n_fft - win) / 2, half = n_fft / 2;
    const long long n = (long long)B * L;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / L), s = (int)(idx % L);
        float acc = frames_ola(dfr, b, (long long)s + half, win, hop, T, Tp, Jp, left);
        if (s >= 1 && s <= half) acc += frames_ola(dfr, b, (long long)half - s, win, hop, T, Tp, Jp, left);
        if (s <= L - 2 && s >= L - 1 - half) acc += frames_ola(dfr, b, (long long)half + 2LL * (L - 1) - s, win, hop, T, Tp, Jp, left);
        dx[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// reparameterization backward (pvae_module.py:1832-1886).  dz: planar [2][zdim][Jpz]; dlat: planar [2][Hl][Jp] (+=, the
// two latents of the NSVAE encoder write disjoint channels; the caller zeroes the buffer).
struct GuardOut { float dr, di, a0, scale; bool on; };

__device__ __forceinline__ GuardOut guard_fwd(float sg, float dr0, float di0, float e) {
    GuardOut g;
    g.a0 = sqrtf(dr0 * dr0 + di0 * di0 + e);
    g.scale = sg * 0.99f / (g.a0 + e);
    g.on = g.a0 >= sg - 1e-3f;
    g.dr = g.on ? dr0 * g.scale : dr0;
    g.di = g.on ? di0 * g.scale : di0;
    return g;
}
// (d dr, d di) of the guarded values -> (d dr0, d di0), and the part that reaches sigma
__device__ __forceinline__ void guard_bwd(const GuardOut& g, float dr0, float di0, float e, float ddr, float ddi, float& o_dr0,
                                          float& o_di0, float& o_dsg) {
    if (!g.on) { o_dr0 = ddr; o_di0 = ddi; o_dsg = 0.f; return; }
    const float ds = ddr * dr0 + ddi * di0;
    const float da0 = -ds * g.scale / (g.a0 + e);
    o_dr0 = ddr * g.scale + da0 * dr0 / g.a0;
    o_di0 = ddi * g.scale + da0 * di0 / g.a0;
    o_dsg = ds * 0.99f / (g.a0 + e);
}

__global__ void reparam_bwd_kernel(const float* __restrict__ lat, int Hl, int off_miu, int off_ls, int off_dl, int zdim,
                                   const float* __restrict__ eps_r, const float* __restrict__ eps_i, int ns, int B, int T,
                                   int Tp, int Jp, const float* __restrict__ dz, int Jpz, float* __restrict__ dlat) {
    const long long n = (long long)B * zdim * T;
    const float e = 1e-6f;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int h = (int)((idx / T) % zdim);
        const int b = (int)(idx / ((long long)T * zdim));
        const size_t j = (size_t)b * Tp + t + 1;
        const size_t im = (size_t)Hl * Jp;
        const float sg = expf(lat[(size_t)(off_ls + h) * Jp + j]);
        const float dr0 = lat[(size_t)(off_dl + h) * Jp + j], di0 = lat[im + (size_t)(off_dl + h) * Jp + j];
        const GuardOut g = guard_fwd(sg, dr0, di0, e);
        const float a = sqrtf(g.dr * g.dr + g.di * g.di + e);
        const float den = sqrtf(2.f * (sg + g.dr) + e);
        const float D = den + e;
        const float w = sg * sg - a * a + e;
        const float num_ii = sqrtf(w);
        float dmr = 0.f, dmi = 0.f, dk_rr = 0.f, dk_ir = 0.f, dk_ii = 0.f;
        for (int s = 0; s < ns; ++s) {
            const size_t ei = (((size_t)b * ns + s) * T + t) * zdim + h;
            const size_t jz = (size_t)(b * ns + s) * Tp + t + 1;
            const float gr = dz[(size_t)h * Jpz + jz], gi = dz[((size_t)zdim + h) * Jpz + jz];
            const float er = eps_r[ei], eim = eps_i[ei];
            dmr += gr; dmi += gi;
            dk_rr += gr * er; dk_ir += gi * er; dk_ii += gi * eim;
        }
        // k_rr = (sg + dr)/D, k_ir = di/D, k_ii = sqrt(sg^2 - a^2 + e)/D, D = sqrt(2(sg+dr)+e) + e
        const float dD = -(dk_rr * (sg + g.dr) + dk_ir * g.di + dk_ii * num_ii) / (D * D);
        float dsg = dk_rr / D, ddr = dk_rr / D, ddi = dk_ir / D;
        const float dw = dk_ii / D / (2.f * num_ii);
        dsg += dw * 2.f * sg;
        const float da = -2.f * a * dw;
        dsg += dD / den; ddr += dD / den;
        ddr += da * g.dr / a; ddi += da * g.di / a;
        float o_dr0, o_di0, o_dsg;
        guard_bwd(g, dr0, di0, e, ddr, ddi, o_dr0, o_di0, o_dsg);
        dsg += o_dsg;
        dlat[(size_t)(off_miu + h) * Jp + j] += dmr;
        dlat[im + (size_t)(off_miu + h) * Jp + j] += dmi;
        dlat[(size_t)(off_ls + h) * Jp + j] += dsg * sg;
        dlat[(size_t)(off_dl + h) * Jp + j] += o_dr0;
        dlat[im + (size_t)(off_dl + h) * Jp + j] += o_di0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// si_snr backward (model/sisnr_loss.py:7-19): per utterance the gradient is  cs_b * s + ce_b * e  (E, D, Q saved by forward)
__global__ void sisnr_bwd_kernel(const float* __restrict__ src, int src_ld, int src_div, const float* __restrict__ est,
                                 int est_ld, int B, int L, const double* __restrict__ work, const float* __restrict__ gout,
                                 float* __restrict__ dest) {
    const int b = blockIdx.y;
    const double eps = 1e-8;
    const double E = work[b * 3], D = work[b * 3 + 1], Q = work[b * 3 + 2];
    const double a = D / (E + eps);
    const double st = a * a * E;
    double en = Q - 2 * a * D + st;
    if (en < 0) en = 0;
    const double R = st / (en + eps);
    const double dR = -(10.0 / 2.302585092994046) / ((double)B * (R + eps)) * (double)gout[0];
    const double p = 2.0 * a * E / (E + eps);
    const double r2 = st / ((en + eps) * (en + eps));
    const float cs = (float)(dR * (p / (en + eps) + r2 * (2.0 * a + 2.0 * (D - a * E) / (E + eps))));
    const float ce = (float)(dR * (-2.0 * r2));
    const float* s = src + (size_t)(b / src_div) * src_ld;
    const float* e = est + (size_t)b * est_ld;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < L; n += gridDim.x * blockDim.x)
        dest[(size_t)b * L + n] = cs * s[n] + ce * e[n];
}

// multiple_recon_loss backward (model/nsvae_loss.py:775-797): loss_cpx and loss_mag terms
__global__ void recon_bwd_kernel(const float* __restrict__ pred_c, const float* __restrict__ ori, long long sb, long long sf,
                                 long long st, long long sr, int ori_div, int B, int F, int T, const float* __restrict__ g_cpx,
                                 const float* __restrict__ g_mag, float* __restrict__ dpred_c) {
    const long long n = (long long)B * F * T;
    const float gc = g_cpx ? g_cpx[0] : 0.f, gm = g_mag ? g_mag[0] : 0.f;
    const float inv_bt = 1.0f / ((float)B * (float)T);
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const float pr = pred_c[idx * 2], pi = pred_c[idx * 2 + 1];
        const long long o = (long long)(b / ori_div) * sb + f * sf + t * st;
        const float orr = ori[o], oi = ori[o + sr];
        const float pm = sqrtf(pr * pr + pi * pi + 1e-6f);
        const float om = sqrtf(orr * orr + orr * orr + 1e-6f);
        const float km = gm * 2.f * (pm - om) / pm;
        dpred_c[idx * 2] = inv_bt * (gc * 2.f * (pr - orr) + km * pr);
        dpred_c[idx * 2 + 1] = inv_bt * (gc * 2.f * (pi - oi) + km * pi);
    }
}

struct LatRef {
    const float* q; int H, Jp, o_miu, o_ls, o_dl;
};

// closed-form complex-Gaussian KL backward w.r.t. q1 (cal_kl, model/nsvae_loss.py:275-328; cal_kl_arbi_prior,
// model/pretrain_pvaes_loss.py:225-281): dq1 planar [2][H1][Jp1], += on the (miu, log_sigma, delta) channels.
__global__ void ckl_bwd_kernel(LatRef q1, LatRef q2, int zdim, float eps, int B, int T, int Tp, const float* __restrict__ gout,
                               float* __restrict__ dq1) {
    const long long n = (long long)B * zdim * T;
    const float gs = gout[0] * 0.5f / ((float)B * (float)T);
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int h = (int)((idx / T) % zdim);
        const int b = (int)(idx / ((long long)T * zdim));
        const size_t j = (size_t)b * Tp + t + 1;
        const float* r1 = q1.q; const float* i1 = q1.q + (size_t)q1.H * q1.Jp;
        const float m1r = r1[(size_t)(q1.o_miu + h) * q1.Jp + j], m1i = i1[(size_t)(q1.o_miu + h) * q1.Jp + j];
        const float s1 = expf(r1[(size_t)(q1.o_ls + h) * q1.Jp + j]);
        const float d1r0 = r1[(size_t)(q1.o_dl + h) * q1.Jp + j], d1i0 = i1[(size_t)(q1.o_dl + h) * q1.Jp + j];
        float m2r = 0.f, m2i = 0.f, s2 = 1.f, d2r = 0.f, d2i = 0.f;
        if (q2.q) {
            const float* r2 = q2.q; const float* i2 = q2.q + (size_t)q2.H * q2.Jp;
            m2r = r2[(size_t)(q2.o_miu + h) * q2.Jp + j]; m2i = i2[(size_t)(q2.o_miu + h) * q2.Jp + j];
            s2 = expf(r2[(size_t)(q2.o_ls + h) * q2.Jp + j]);
            d2r = r2[(size_t)(q2.o_dl + h) * q2.Jp + j]; d2i = i2[(size_t)(q2.o_dl + h) * q2.Jp + j];
        }
        const GuardOut g1 = guard_fwd(s1, d1r0, d1i0, eps);
        const GuardOut g2 = guard_fwd(s2, d2r, d2i, eps);
        d2r = g2.dr; d2i = g2.di;
        const float a1 = g1.dr * g1.dr + g1.di * g1.di, a2 = d2r * d2r + d2i * d2i;
        const float ld1 = 0.25f * (s1 * s1 - a1) + eps;            // argument of log_det_c1
        const float coeff = 2.0f / (s2 * s2 - a2 + eps);
        const float dr = m2r - m1r, di = m2i - m1i;
        // f = coeff (trace + quad) + logdet2 - logdet1
        const float dm1r = -coeff * (2.f * dr * (s2 - d2r) - 2.f * d2i * di);
        const float dm1i = -coeff * (-2.f * d2i * dr + 2.f * di * (s2 + d2r));
        float ds1 = coeff * s2 - 0.5f * s1 / ld1;
        const float dd1r = -coeff * d2r + 0.5f * g1.dr / ld1;
        const float dd1i = -coeff * d2i + 0.5f * g1.di / ld1;
        float o_dr0, o_di0, o_dsg;
        guard_bwd(g1, d1r0, d1i0, eps, dd1r, dd1i, o_dr0, o_di0, o_dsg);
        ds1 += o_dsg;
        float* dr1 = dq1; float* di1 = dq1 + (size_t)q1.H * q1.Jp;
        dr1[(size_t)(q1.o_miu + h) * q1.Jp + j] += gs * dm1r;
        di1[(size_t)(q1.o_miu + h) * q1.Jp + j] += gs * dm1i;
        dr1[(size_t)(q1.o_ls + h) * q1.Jp + j] += gs * ds1 * s1;
        dr1[(size_t)(q1.o_dl + h) * q1.Jp + j] += gs * o_dr0;
        di1[(size_t)(q1.o_dl + h) * q1.Jp + j] += gs * o_di0;
    }
}

// miu_dis_loss backward (model/nsvae_loss.py:349-360): L = sqrt(sum_{h,ri} mean_{b,t} (a - b)^2)
//   dL/da = (a - b) / (B T L), dL/db = -dL/da;  da / db planar (+=), either may be null
__global__ void miu_dist_bwd_kernel(LatRef q1, LatRef q2, int zdim, int B, int T, int Tp, const float* __restrict__ gout,
                                    const float* __restrict__ lval, float* __restrict__ d1, float* __restrict__ d2) {
    const long long n = 2LL * zdim * B * T;
    const float k = gout[0] / ((float)B * (float)T * lval[0]);
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int b = (int)((idx / T) % B);
        const int h = (int)((idx / ((long long)T * B)) % zdim);
        const int ri = (int)(idx / ((long long)T * B * zdim));
        const size_t j = (size_t)b * Tp + t + 1;
        const size_t o1 = ((size_t)ri * q1.H + q1.o_miu + h) * q1.Jp + j;
        const size_t o2 = ((size_t)ri * q2.H + q2.o_miu + h) * q2.Jp + j;
        const float v = k * (q1.q[o1] - q2.q[o2]);
        if (d1) d1[o1] += v;
        if (d2) d2[o2] -= v;
    }
}

// adjoint of the batch repeat of the skip connections (pvae_module.py:2563-2567, x.repeat over num_samples):
//   dx[row][b*Tp + tp] = sum_{s < n} drep[row][(b*n + s)*Tp + tp]
__global__ void repeat_sum_kernel(const float* __restrict__ drep, int n, int B, int Tp, int Jp_rep, int Jp,
                                  float* __restrict__ dx) {
    const int row = blockIdx.y;
    const int J = B * Tp;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int b = j / Tp, tp = j - b * Tp;
        float acc = 0.f;
        for (int s = 0; s < n; ++s) acc += drep[(size_t)row * Jp_rep + (size_t)(b * n + s) * Tp + tp];
        dx[(size_t)row * Jp + j] = acc;
    }
}

// out[m] (+)= sum_{j < J} x[m][j]   (bias gradients of the point-wise contractions)
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, int Jp, int J, int accumulate,
                                                     float* __restrict__ out) {
    const int m = blockIdx.x;
    const float* r = x + (size_t)m * Jp;
    double s = 0;
    for (int j = threadIdx.x; j < J; j += blockDim.x) s += r[j];
    __shared__ double sh[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
        out[m] = accumulate ? out[m] + v : v;
    }
}

}  // namespace

extern "C" int idv_cbn_apply_prelu_to(const float* y, const float* fold, const float* prelu_slope, int C, int F, int B, int Tp,
                                      int Jp, int t_valid, float* out, void* stream) {
    if (!y || !fold || !out || C <= 0 || F <= 0 || B <= 0) return IDV_EINVAL;
    const int J = B * Tp;
    int gx = (J + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(cbn_apply_prelu_to_kernel, dim3(gx, C * F), dim3(256), 0, (hipStream_t)stream, y, fold, prelu_slope, C,
                       F, B, Tp, Jp, t_valid, out);
    return idv_launch_status();
}

// idv_cbn_apply_prelu_to writing the split image of the result as well (img: hi plane, lo plane lo_off_elems bf16 further;
// C % 4 == 0; layout and edge slots as idv_planar_to_image)
extern "C" int idv_cbn_apply_prelu_to_img(const float* y, const float* fold, const float* prelu_slope, int C, int F, int B, int Tp,
                                          int Jp, int t_valid, float* out, void* img, long long lo_off_elems, void* stream) {
    if (!y || !fold || !out || !img || C <= 0 || (C % 4) || F <= 0 || B <= 0 || Jp < B * Tp || (lo_off_elems % 8) ||
        (reinterpret_cast<uintptr_t>(img) & 15))
        return IDV_EINVAL;
    int gx = (Jp + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(cbn_apply_prelu_to_img_kernel, dim3(gx, (C / 4) * F), dim3(256), 0, (hipStream_t)stream, y, fold, prelu_slope,
                       C, F, B, Tp, Jp, t_valid, out, (unsigned short*)img, lo_off_elems);
    return idv_launch_status();
}

extern "C" int idv_cbn_bwd_apply_img(const float* dz, const float* y, const float* fold, const float* coef, const float* prelu_slope,
                                     int C, int F, int B, int Tp, int Jp, int t_valid, float* dy, void* img, long long lo_off_elems,
                                     void* stream) {
    if (!dz || !y || !fold || !coef || !dy || !img || C <= 0 || (C % 4) || F <= 0 || B <= 0 || Jp < B * Tp || (lo_off_elems % 8) ||
        (reinterpret_cast<uintptr_t>(img) & 15))
        return IDV_EINVAL;
    int gx = (Jp + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(cbn_bwd_apply_img_kernel, dim3(gx, (C / 4) * F), dim3(256), 0, (hipStream_t)stream, dz, y, fold, coef,
                       prelu_slope, C, F, B, Tp, Jp, t_valid, dy, (unsigned short*)img, lo_off_elems);
    return idv_launch_status();
}

extern "C" int idv_cbn_bwd_reduce(const float* dz, const float* y, const float* fold, const float* prelu_slope, int C, int F,
                                  int B, int Tp, int Jp, int t_valid, double* sums, void* stream) {
    if (!dz || !y || !fold || !sums || C <= 0 || F <= 0 || B <= 0) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(double) * NSUM * C, st) != hipSuccess) return IDV_ELAUNCH;
    const int J = B * Tp;
    int gx = (J + 255) / 256;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(cbn_bwd_reduce_kernel, dim3(gx, C * F), dim3(256), 0, st, dz, y, fold, prelu_slope, C, F, B, Tp, Jp,
                       t_valid, sums);
    return idv_launch_status();
}

extern "C" int idv_cbn_bwd_finalize(const double* sums, double count, const float* moments, const float* gamma_rr,
                                    const float* gamma_ri, const float* gamma_ii, int C, float* coef, float* dgamma_rr,
                                    float* dgamma_ri, float* dgamma_ii, float* dbeta_r, float* dbeta_i, float* dslope,
                                    float param_grad_scale, void* stream) {
    if (!sums || count <= 0 || !moments || !gamma_rr || !gamma_ri || !gamma_ii || !coef || !dgamma_rr || !dgamma_ri ||
        !dgamma_ii || !dbeta_r || !dbeta_i || C <= 0)
        return IDV_EINVAL;
    hipLaunchKernelGGL(cbn_bwd_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, count, moments, gamma_rr,
                       gamma_ri, gamma_ii, C, coef, dgamma_rr, dgamma_ri, dgamma_ii, dbeta_r, dbeta_i, dslope, param_grad_scale);
    return idv_launch_status();
}

extern "C" int idv_cbn_bwd_apply(const float* dz, const float* y, const float* fold, const float* coef,
                                 const float* prelu_slope, int C, int F, int B, int Tp, int Jp, int t_valid, float* dy,
                                 void* stream) {
    if (!dz || !y || !fold || !coef || !dy || C <= 0 || F <= 0 || B <= 0) return IDV_EINVAL;
    const int J = B * Tp;
    int gx = (J + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(cbn_bwd_apply_kernel, dim3(gx, C * F), dim3(256), 0, (hipStream_t)stream, dz, y, fold, coef, prelu_slope,
                       C, F, B, Tp, Jp, t_valid, dy);
    return idv_launch_status();
}

extern "C" int idv_mask_apply_bwd(const float* mask, const float* X, int x_div, int JpX, const float* dpred,
                                  const float* dpred_c, int F, int B, int T, int Tp, int Jp, float* dmask, float* dX,
                                  void* stream) {
    if (!mask || !X || !dmask || (!dpred && !dpred_c) || x_div < 1 || F <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    if (dX && x_div != 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_guard_cols_kernel, dim3(grid_for(2LL * F * B * (Tp - T))), dim3(256), 0, st, dmask, 2 * F, B, T, Tp, Jp);
    if (dX)
        hipLaunchKernelGGL(zero_guard_cols_kernel, dim3(grid_for(2LL * F * B * (Tp - T))), dim3(256), 0, st, dX, 2 * F, B, T, Tp, JpX);
    hipLaunchKernelGGL(mask_bwd_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, mask, X, x_div, JpX, dpred,
                       dpred_c, F, B, T, Tp, Jp, dmask, dX);
    return idv_launch_status();
}

extern "C" int idv_complex_to_planar(const float* in_c, float* act, int F, int B, int T, int Tp, int Jp, void* stream) {
    if (!in_c || !act || F <= 0 || B <= 0 || T <= 0 || Tp < T + 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_guard_cols_kernel, dim3(grid_for(2LL * F * B * (Tp - T))), dim3(256), 0, st, act, 2 * F, B, T, Tp, Jp);
    hipLaunchKernelGGL(complex_to_planar_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, in_c, act, F, B, T, Tp, Jp);
    return idv_launch_status();
}

extern "C" int idv_istft_ola_bwd(const float* dy, const float* env_inv, int B, int n_fft, int win, int hop, int T, int Tp,
                                 int Jp, float* dframes, void* stream) {
    if (!dy || !env_inv || !dframes || B <= 0 || T < 2 || Tp < T + 1 || Jp < B * Tp) return IDV_EINVAL;
    const int J = B * Tp;
    int gx = (J + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(istft_ola_bwd_kernel, dim3(gx, win), dim3(256), 0, (hipStream_t)stream, dy, env_inv, B, n_fft, win, hop,
                       T, Tp, Jp, dframes);
    return idv_launch_status();
}

extern "C" int idv_stft_frames_bwd(const float* dframes, int B, int L, int n_fft, int win, int hop, int T, int Tp, int Jp,
                                   float* dx, void* stream) {
    if (!dframes || !dx || B <= 0 || L <= n_fft / 2 || T != 1 + L / hop || Tp < T + 1 || Jp < B * Tp) return IDV_EINVAL;
    hipLaunchKernelGGL(stft_frames_bwd_kernel, dim3(grid_for((long long)B * L)), dim3(256), 0, (hipStream_t)stream, dframes, B, L,
                       n_fft, win, hop, T, Tp, Jp, dx);
    return idv_launch_status();
}

extern "C" int idv_reparam_bwd(const float* lat, int Hl, int off_miu, int off_ls, int off_dl, int zdim, const float* eps_r,
                               const float* eps_i, int ns, int B, int T, int Tp, int Jp, const float* dz, int Jpz, float* dlat,
                               void* stream) {
    if (!lat || !eps_r || !eps_i || !dz || !dlat || zdim <= 0 || ns <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    if (off_miu + zdim > Hl || off_ls + zdim > Hl || off_dl + zdim > Hl || Jpz < B * ns * Tp) return IDV_EINVAL;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for((long long)B * zdim * T)), dim3(256), 0, (hipStream_t)stream, lat, Hl,
                       off_miu, off_ls, off_dl, zdim, eps_r, eps_i, ns, B, T, Tp, Jp, dz, Jpz, dlat);
    return idv_launch_status();
}

extern "C" int idv_sisnr_bwd(const float* source, int src_ld, int src_div, const float* est, int est_ld, int B, int L,
                             const double* work, const float* grad_out, float* dest, void* stream) {
    if (!source || !est || !work || !grad_out || !dest || B <= 0 || L <= 0 || src_div < 1) return IDV_EINVAL;
    int gx = (L + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(sisnr_bwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, source, src_ld, src_div, est, est_ld, B,
                       L, work, grad_out, dest);
    return idv_launch_status();
}

extern "C" int idv_recon_loss_bwd(const float* pred_c, const float* ori, long long sb, long long sf, long long st_,
                                  long long sr, int ori_div, int B, int F, int T, const float* g_cpx, const float* g_mag,
                                  float* dpred_c, void* stream) {
    if (!pred_c || !ori || !dpred_c || B <= 0 || F <= 0 || T <= 0 || ori_div < 1) return IDV_EINVAL;
    hipLaunchKernelGGL(recon_bwd_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, (hipStream_t)stream, pred_c, ori, sb,
                       sf, st_, sr, ori_div, B, F, T, g_cpx, g_mag, dpred_c);
    return idv_launch_status();
}

extern "C" int idv_ckl_bwd(const float* q1, int H1, int Jp1, int o1_miu, int o1_ls, int o1_dl, const float* q2, int H2,
                           int Jp2, int o2_miu, int o2_ls, int o2_dl, int zdim, float eps, int B, int T, int Tp,
                           const float* grad_out, float* dq1, void* stream) {
    if (!q1 || !grad_out || !dq1 || zdim <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    LatRef a{q1, H1, Jp1, o1_miu, o1_ls, o1_dl}, b{q2, H2, Jp2, o2_miu, o2_ls, o2_dl};
    hipLaunchKernelGGL(ckl_bwd_kernel, dim3(grid_for((long long)B * zdim * T)), dim3(256), 0, (hipStream_t)stream, a, b, zdim,
                       eps, B, T, Tp, grad_out, dq1);
    return idv_launch_status();
}

extern "C" int idv_miu_dist_bwd(const float* q1, int H1, int Jp1, int off1, const float* q2, int H2, int Jp2, int off2,
                                int zdim, int B, int T, int Tp, const float* grad_out, const float* loss_value, float* dq1,
                                float* dq2, void* stream) {
    if (!q1 || !q2 || !grad_out || !loss_value || (!dq1 && !dq2) || zdim <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    LatRef a{q1, H1, Jp1, off1, 0, 0}, b{q2, H2, Jp2, off2, 0, 0};
    hipLaunchKernelGGL(miu_dist_bwd_kernel, dim3(grid_for(2LL * zdim * B * T)), dim3(256), 0, (hipStream_t)stream, a, b, zdim, B,
                       T, Tp, grad_out, loss_value, dq1, dq2);
    return idv_launch_status();
}

extern "C" int idv_repeat_batch_bwd(const float* drep, int n, int rows, int B, int Tp, int Jp_rep, int Jp, float* dx,
                                    void* stream) {
    if (!drep || !dx || n < 1 || rows <= 0 || B <= 0 || Tp <= 0 || Jp < B * Tp || Jp_rep < B * n * Tp) return IDV_EINVAL;
    int gx = (B * Tp + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(repeat_sum_kernel, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, drep, n, B, Tp, Jp_rep, Jp, dx);
    return idv_launch_status();
}

extern "C" int idv_planar_rowsum(const float* x, int M, int Jp, int J, int accumulate, float* out, void* stream) {
    if (!x || !out || M <= 0 || J <= 0 || Jp < J) return IDV_EINVAL;
    hipLaunchKernelGGL(rowsum_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, Jp, J, accumulate, out);
    return idv_launch_status();
}
