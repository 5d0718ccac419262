// Bandwidth-bound kernels of the hot path: STFT framing, ISTFT overlap-add, mask application,
// train-mode complex batch-norm finalise/apply, reparameterisation.
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

// frames[k][j] = xp[b][hop*t + left + k],  xp = reflect-pad(x, n_fft/2)   (torch.stft, center=True)
// one block per (b, 32 frames); the signal segment is staged in LDS, writes are coalesced along j.
constexpr int FR_TT = 32;
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, int B, int L, int n_fft, int win,
                                                          int hop, int T, float* __restrict__ frames, int Tp, int Jp) {
    extern __shared__ float seg[];
    const int b = blockIdx.y, t0 = blockIdx.x * FR_TT;
    const int left = (n_fft - win) / 2, half = n_fft / 2;
    const int nt = min(FR_TT, T - t0);
    const int seglen = hop * (nt - 1) + win;
    const long long s0 = (long long)hop * t0 + left - half;      // original-signal index of seg[0]
    for (int e = threadIdx.x; e < seglen; e += blockDim.x) {
        long long s = s0 + e;
        if (s < 0) s = -s;
        if (s >= L) s = 2LL * (L - 1) - s;
        seg[e] = (s >= 0 && s < L) ? x[(size_t)b * L + s] : 0.f;
    }
    __syncthreads();
    const int tl = threadIdx.x & 31, kq = threadIdx.x >> 5;       // 32 frames x 8 k-lanes
    if (tl < nt) {
        const size_t col = (size_t)b * Tp + t0 + tl + 1;
        for (int k = kq; k < win; k += 8) frames[(size_t)k * Jp + col] = seg[hop * tl + k];
    }
    // guard column tp == 0 of this utterance
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < win; k += blockDim.x) frames[(size_t)k * Jp + (size_t)b * Tp] = 0.f;
}

// y[b][s] = env_inv[s + half] * sum_t frames[n = s + half - hop*t - left][b*Tp + t + 1]
__global__ void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ env_inv, int B, int n_fft,
                                 int win, int hop, int T, int Tp, int Jp, float* __restrict__ y) {
    const int Lout = hop * (T - 1);
    const int left = (n_fft - win) / 2, half = n_fft / 2;
    const long long n = (long long)B * Lout;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / Lout), s = (int)(idx % Lout);
        const int p = s + half - left;                 // position relative to the window start of frame 0
        int t_hi = p / hop;
        if (t_hi > T - 1) t_hi = T - 1;
        int t_lo = (p - win + hop) / hop;              // smallest t with p - hop*t < win
        if (p - win + 1 <= 0) t_lo = 0;
        if (t_lo < 0) t_lo = 0;
        float acc = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int k = p - hop * t;
            if (k >= 0 && k < win) acc += frames[(size_t)k * Jp + (size_t)b * Tp + t + 1];
        }
        y[idx] = acc * env_inv[s + half];
    }
}

// predict = X * M * tanh|M| / |M|   (== |X| tanh|M| exp(j(angle X + angle M)), pvae_module.py:224-234)
__global__ void mask_apply_kernel(const float* __restrict__ mask, const float* __restrict__ X, int x_div, int JpX,
                                  float* __restrict__ pred, float* __restrict__ pred_c, int F, int B, int T, int Tp, int Jp) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t jm = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        const size_t jx = (size_t)f * JpX + (size_t)(b / x_div) * Tp + t + 1;
        const float mr = mask[jm], mi = mask[(size_t)F * Jp + jm];
        const float xr = X[jx], xi = X[(size_t)F * JpX + jx];
        const float mm = sqrtf(mr * mr + mi * mi);
        const float g = tanhf(mm);
        // unit phasor of the mask exactly as the reference builds it: (m / (tanh|m| + 1e-8)) normalised
        float ur, ui;
        if (mm > 0.f) {
            const float inv = 1.0f / mm;
            ur = mr * inv; ui = mi * inv;
        } else {
            ur = 1.f; ui = 0.f;     // atan2(0, 0) = 0
        }
        const float pr = g * (xr * ur - xi * ui), pi = g * (xr * ui + xi * ur);
        pred[jm] = pr;
        pred[(size_t)F * Jp + jm] = pi;
        if (pred_c) {
            pred_c[idx * 2] = pr;
            pred_c[idx * 2 + 1] = pi;
        }
    }
}

// Enhancement estimators of the two-latent evaluation path (reference i_dccrn_vae/nsvae_dccrn/test_se_cvaefinetune.py:
// real_and_imag_mask :85-101, complex_mask :104-116, phase_sensitive_mask :119-135): S = mean over the ns sampled speech
// spectra, N = mean over the ns sampled noise spectra, X = the noisy spectrum; mode 0: (Sr^2 / (Sr^2 + Nr^2 + eps)) Xr and the
// same for the imaginary parts; mode 1: S / (S + N + eps) * X (complex); mode 2: |S| / (|S| + |N| + eps) * cos(angle S -
// angle X) * |X| * exp(j angle S) = S * Re(S conj X) / (|S| (|S| + |N| + eps)) (0 where S = 0; where X = 0 both forms are 0).
// speech / noise: interleaved complex [B*ns][F][T][2]; X: strided [B][F][T][2]; out: planar [2][F][Jp] (guard columns zeroed
// by the caller), out_c (optional): interleaved complex [B][F][T][2].
__global__ void outtype_kernel(const float* __restrict__ speech, const float* __restrict__ noise, const float* __restrict__ X,
                               long long sb, long long sf, long long st_, long long sr, int mode, int ns, int B, int F, int T,
                               int Tp, int Jp, float* __restrict__ out, float* __restrict__ out_c) {
    const long long n = (long long)B * F * T;
    const float eps = 1e-10f;
    const float inv = 1.0f / (float)ns;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        float s_r = 0.f, s_i = 0.f, n_r = 0.f, n_i = 0.f;
        for (int k = 0; k < ns; ++k) {
            const size_t o = ((((size_t)b * ns + k) * F + f) * T + t) * 2;
            s_r += speech[o]; s_i += speech[o + 1];
            n_r += noise[o]; n_i += noise[o + 1];
        }
        s_r *= inv; s_i *= inv; n_r *= inv; n_i *= inv;
        const float* xp = X + b * sb + f * sf + t * st_;
        const float xr = xp[0], xi = xp[sr];
        float er, ei;
        if (mode == 0) {
            er = s_r * s_r / (s_r * s_r + n_r * n_r + eps) * xr;
            ei = s_i * s_i / (s_i * s_i + n_i * n_i + eps) * xi;
        } else if (mode == 1) {
            const float dr = s_r + n_r + eps, di = s_i + n_i;          // complex + real eps
            const float dd = dr * dr + di * di;
            float mr = 0.f, mi = 0.f;
            if (dd > 0.f) {
                mr = (s_r * dr + s_i * di) / dd;
                mi = (s_i * dr - s_r * di) / dd;
            }
            er = mr * xr - mi * xi;
            ei = mr * xi + mi * xr;
        } else {
            const float sm = sqrtf(s_r * s_r + s_i * s_i), nm = sqrtf(n_r * n_r + n_i * n_i);
            float g = 0.f;
            if (sm > 0.f) g = (s_r * xr + s_i * xi) / (sm * (sm + nm + eps));
            er = g * s_r;
            ei = g * s_i;
        }
        const size_t j = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        out[j] = er;
        out[(size_t)F * Jp + j] = ei;
        if (out_c) {
            out_c[idx * 2] = er;
            out_c[idx * 2 + 1] = ei;
        }
    }
}

__global__ void planar_to_complex_kernel(const float* __restrict__ act, float* __restrict__ out_c, int F, int B, int T,
                                         int Tp, int Jp) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t j = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        out_c[idx * 2] = act[j];
        out_c[idx * 2 + 1] = act[(size_t)F * Jp + j];
    }
}

// sums (double) -> moments, running buffers, fold.  One thread per channel.
__global__ void cbn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ g_rr,
                                    const float* __restrict__ g_ri, const float* __restrict__ g_ii,
                                    const float* __restrict__ b_r, const float* __restrict__ b_i, int C, int first_call,
                                    float momentum, float* __restrict__ run_r, float* __restrict__ run_i,
                                    float* __restrict__ rVrr, float* __restrict__ rVri, float* __restrict__ rVii,
                                    float* __restrict__ moments, float* __restrict__ fold) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double* s = stats + (size_t)c * 5;
    const double mr = s[0] / count, mi = s[1] / count;
    const float eps = 1e-5f;
    const float mu_r = (float)mr, mu_i = (float)mi;
    const float Vrr = (float)(s[2] / count - mr * mr) + eps;
    const float Vii = (float)(s[3] / count - mi * mi) + eps;
    const float Vri = (float)(s[4] / count - mr * mi);
    if (moments) {
        moments[c] = mu_r; moments[C + c] = mu_i; moments[2 * C + c] = Vrr; moments[3 * C + c] = Vri; moments[4 * C + c] = Vii;
    }
    if (run_r) {
        if (first_call) {
            run_r[c] = mu_r; run_i[c] = mu_i; rVrr[c] = Vrr; rVri[c] = Vri; rVii[c] = Vii;
        } else {
            const float a = momentum, bq = 1.0f - momentum;
            run_r[c] = a * run_r[c] + bq * mu_r;
            run_i[c] = a * run_i[c] + bq * mu_i;
            rVrr[c] = a * rVrr[c] + bq * Vrr;
            rVri[c] = a * rVri[c] + bq * Vri;
            rVii[c] = a * rVii[c] + bq * Vii;
        }
    }
    float delta = Vrr * Vii - Vri * Vri + eps;
    delta = fmaxf(delta, 1e-8f);
    const float sq = sqrtf(delta);
    const float tt = sqrtf(Vrr + Vii + 2.f * sq + eps);
    const float inv = 1.0f / (sq * tt + eps);
    const float Wrr = (Vii + sq) * inv, Wii = (Vrr + sq) * inv, Wri = -Vri * inv;
    const float Zrr = g_rr[c] * Wrr + g_ri[c] * Wri;
    const float Zri = g_rr[c] * Wri + g_ri[c] * Wii;
    const float Zir = g_ri[c] * Wrr + g_ii[c] * Wri;
    const float Zii = g_ri[c] * Wri + g_ii[c] * Wii;
    float* z = fold + (size_t)c * 6;
    z[0] = Zrr; z[1] = Zri; z[2] = Zir; z[3] = Zii;
    z[4] = b_r[c] - (Zrr * mu_r + Zri * mu_i);
    z[5] = b_i[c] - (Zir * mu_r + Zii * mu_i);
}

// five moments of one channel of a planar activation (stand-alone ComplexBatchNormal.forward);
// grid = (chunks, C*F rows); double atomics into stats[c][5]
__global__ __launch_bounds__(256) void cbn_stats_kernel(const float* __restrict__ act, int C, int F, int B, int Tp, int Jp,
                                                        int t_valid, double* __restrict__ stats) {
    const int row = blockIdx.y, c = row / F;
    const float* pr = act + (size_t)row * Jp;
    const float* pi = act + ((size_t)C * F + row) * Jp;
    const int J = B * Tp;
    double s[5] = {0, 0, 0, 0, 0};
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        if (tp < 1 || tp > t_valid) continue;
        const double r = pr[j], im = pi[j];
        s[0] += r; s[1] += im; s[2] += r * r; s[3] += im * im; s[4] += r * im;
    }
    __shared__ double sh[5][4];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double v = wave_sum_d(s[q]);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&stats[(size_t)c * 5 + threadIdx.x], sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// in place: (r, i) <- PReLU(Z (r, i) + s) on kept columns; grid = (column tiles, C*F rows)
__global__ void cbn_apply_prelu_kernel(float* __restrict__ act, const float* __restrict__ fold,
                                       const float* __restrict__ slope_p, int C, int F, int B, int Tp, int Jp, int t_valid) {
    const int row = blockIdx.y;            // c*F + f
    const int c = row / F;
    const float* z = fold + (size_t)c * 6;
    const float Zrr = z[0], Zri = z[1], Zir = z[2], Zii = z[3], sr = z[4], si = z[5];
    const float slope = slope_p ? *slope_p : 1.0f;
    float* pr = act + (size_t)row * Jp;
    float* pi = act + ((size_t)C * F + row) * Jp;
    const int J = B * Tp;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
        const int tp = j % Tp;
        if (tp < 1 || tp > t_valid) continue;
        const float r = pr[j], im = pi[j];
        float yr = Zrr * r + Zri * im + sr;
        float yi = Zir * r + Zii * im + si;
        yr = yr >= 0.f ? yr : slope * yr;
        yi = yi >= 0.f ? yi : slope * yi;
        pr[j] = yr;
        pi[j] = yi;
    }
}

// reparameterisation, pvae_module.py:1832-1886.  One thread per (b, t, h); loops over the ns samples.
__global__ void reparam_kernel(const float* __restrict__ lat, int Hl, int off_miu, int off_ls, int off_dl, int zdim,
                               const float* __restrict__ eps_r, const float* __restrict__ eps_i, int ns, int B, int T,
                               int Tp, int Jp, float* __restrict__ z, int Jpz) {
    const long long n = (long long)B * zdim * T;
    const float e = 1e-6f;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int h = (int)((idx / T) % zdim);
        const int b = (int)(idx / ((long long)T * zdim));
        const size_t j = (size_t)b * Tp + t + 1;
        const float* re = lat;
        const float* im = lat + (size_t)Hl * Jp;
        const float mr = re[(size_t)(off_miu + h) * Jp + j], mi = im[(size_t)(off_miu + h) * Jp + j];
        const float sg = expf(re[(size_t)(off_ls + h) * Jp + j]);
        float dr = re[(size_t)(off_dl + h) * Jp + j], di = im[(size_t)(off_dl + h) * Jp + j];
        float a = sqrtf(dr * dr + di * di + e);
        const float scale = sg * 0.99f / (a + e);
        if (a >= sg - 1e-3f) { dr *= scale; di *= scale; }
        a = sqrtf(dr * dr + di * di + e);
        const float den = sqrtf(2.f * (sg + dr) + e);
        const float k_rr = (sg + dr) / (den + e);
        const float k_ir = di / (den + e);
        const float k_ii = sqrtf(sg * sg - a * a + e) / (den + e);
        for (int s = 0; s < ns; ++s) {
            const size_t ei = (((size_t)b * ns + s) * T + t) * zdim + h;
            const float er = eps_r[ei], eim = eps_i[ei];
            const size_t jz = (size_t)(b * ns + s) * Tp + t + 1;
            z[(size_t)h * Jpz + jz] = mr + k_rr * er;
            z[((size_t)zdim + h) * Jpz + jz] = mi + k_ir * er + k_ii * eim;
        }
    }
}

// Optional input normalisation of DCCRN_.forward (pvae_module.py:217-221):  (stft - mean) / (std + 1e-6) per (bin, part),
// imaginary part of the first and last bin zeroed;  inverse (:235-238): std * pred + mean, written planar and interleaved.
__global__ void datanorm_kernel(const float* __restrict__ X, const float* __restrict__ mean, const float* __restrict__ stdv,
                                int F, int B, int T, int Tp, int Jp, float* __restrict__ out) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t j = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        const float r = (X[j] - mean[2 * f]) / (stdv[2 * f] + 1e-6f);
        float im = (X[(size_t)F * Jp + j] - mean[2 * f + 1]) / (stdv[2 * f + 1] + 1e-6f);
        if (f == 0 || f == F - 1) im = 0.f;
        out[j] = r;
        out[(size_t)F * Jp + j] = im;
    }
}

__global__ void datadenorm_kernel(const float* __restrict__ P, const float* __restrict__ mean, const float* __restrict__ stdv,
                                  int F, int B, int T, int Tp, int Jp, float* __restrict__ out, float* __restrict__ out_c) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t j = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        const float r = stdv[2 * f] * P[j] + mean[2 * f];
        const float im = stdv[2 * f + 1] * P[(size_t)F * Jp + j] + mean[2 * f + 1];
        out[j] = r;
        out[(size_t)F * Jp + j] = im;
        out_c[idx * 2] = r;
        out_c[idx * 2 + 1] = im;
    }
}

// backward of datanorm_kernel: dX = dout / (std + 1e-6), the imaginary parts of the first and last bin carry no gradient
__global__ void datanorm_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ stdv, int F, int B, int T, int Tp,
                                    int Jp, float* __restrict__ dX) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t j = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        dX[j] = dout[j] / (stdv[2 * f] + 1e-6f);
        dX[(size_t)F * Jp + j] = (f == 0 || f == F - 1) ? 0.f : dout[(size_t)F * Jp + j] / (stdv[2 * f + 1] + 1e-6f);
    }
}

// backward of datadenorm_kernel: dP = std * (gradient of the planar output + gradient of the interleaved complex output)
__global__ void datadenorm_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ dout_c, const float* __restrict__ stdv,
                                      int F, int B, int T, int Tp, int Jp, float* __restrict__ dP) {
    const long long n = (long long)B * F * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const int f = (int)((idx / T) % F);
        const int b = (int)(idx / ((long long)T * F));
        const size_t j = (size_t)f * Jp + (size_t)b * Tp + t + 1;
        float gr = 0.f, gi = 0.f;
        if (dout) { gr += dout[j]; gi += dout[(size_t)F * Jp + j]; }
        if (dout_c) { gr += dout_c[idx * 2]; gi += dout_c[idx * 2 + 1]; }
        dP[j] = stdv[2 * f] * gr;
        dP[(size_t)F * Jp + j] = stdv[2 * f + 1] * gi;
    }
}

__global__ void zero_guard_kernel(float* __restrict__ act, int planes, int B, int Tp, int Jp) {
    const long long n = (long long)planes * B;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x)
        act[(size_t)(idx / B) * Jp + (size_t)(idx % B) * Tp] = 0.f;
}

inline int grid_for(long long n, int bs = 256) {
    long long g = (n + bs - 1) / bs;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int idv_stft_frames(const float* x, int B, int L, int n_fft, int win, int hop, int T, float* frames, int Tp,
                               int Jp, void* stream) {
    if (!x || !frames || B <= 0 || L <= n_fft / 2 || T != 1 + L / hop || Tp < T + 1 || Jp < B * Tp) return IDV_EINVAL;
    const size_t smem = (size_t)(hop * (FR_TT - 1) + win) * sizeof(float);
    hipLaunchKernelGGL(stft_frames_kernel, dim3((T + FR_TT - 1) / FR_TT, B), dim3(256), smem, (hipStream_t)stream, x, B, L,
                       n_fft, win, hop, T, frames, Tp, Jp);
    return idv_launch_status();
}

extern "C" int idv_istft_ola(const float* frames, const float* env_inv, int B, int n_fft, int win, int hop, int T, int Tp,
                             int Jp, float* y, void* stream) {
    if (!frames || !env_inv || !y || B <= 0 || T < 2 || Tp < T + 1) return IDV_EINVAL;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(grid_for((long long)B * hop * (T - 1))), dim3(256), 0, (hipStream_t)stream,
                       frames, env_inv, B, n_fft, win, hop, T, Tp, Jp, y);
    return idv_launch_status();
}

extern "C" int idv_mask_apply(const float* mask, const float* X, int x_div, int JpX, float* pred, float* pred_c, int F,
                              int B, int T, int Tp, int Jp, void* stream) {
    if (!mask || !X || !pred || x_div < 1 || F <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    hipLaunchKernelGGL(zero_guard_kernel, dim3(grid_for(2LL * F * B)), dim3(256), 0, (hipStream_t)stream, pred, 2 * F, B, Tp, Jp);
    hipLaunchKernelGGL(mask_apply_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, (hipStream_t)stream, mask, X,
                       x_div, JpX, pred, pred_c, F, B, T, Tp, Jp);
    return idv_launch_status();
}

extern "C" int idv_datanorm(const float* X, const float* mean, const float* stdv, int F, int B, int T, int Tp, int Jp, float* out,
                            void* stream) {
    if (!X || !mean || !stdv || !out || F <= 0 || B <= 0 || T <= 0 || Tp < T + 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * 2 * (size_t)F * Jp, st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(datanorm_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, X, mean, stdv, F, B, T, Tp, Jp, out);
    return idv_launch_status();
}

extern "C" int idv_datadenorm(const float* P, const float* mean, const float* stdv, int F, int B, int T, int Tp, int Jp, float* out,
                              float* out_c, void* stream) {
    if (!P || !mean || !stdv || !out || !out_c || F <= 0 || B <= 0 || T <= 0 || Tp < T + 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * 2 * (size_t)F * Jp, st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(datadenorm_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, P, mean, stdv, F, B, T, Tp, Jp,
                       out, out_c);
    return idv_launch_status();
}

// gradients of idv_datanorm / idv_datadenorm (training with the reference's --data_norm; model/pvae_module.py:217-221, :235-238)
extern "C" int idv_datanorm_bwd(const float* dout, const float* stdv, int F, int B, int T, int Tp, int Jp, float* dX, void* stream) {
    if (!dout || !stdv || !dX || F <= 0 || B <= 0 || T <= 0 || Tp < T + 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dX, 0, sizeof(float) * 2 * (size_t)F * Jp, st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(datanorm_bwd_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, dout, stdv, F, B, T, Tp, Jp, dX);
    return idv_launch_status();
}

extern "C" int idv_datadenorm_bwd(const float* dout, const float* dout_c, const float* stdv, int F, int B, int T, int Tp, int Jp,
                                  float* dP, void* stream) {
    if ((!dout && !dout_c) || !stdv || !dP || F <= 0 || B <= 0 || T <= 0 || Tp < T + 1) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dP, 0, sizeof(float) * 2 * (size_t)F * Jp, st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(datadenorm_bwd_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, dout, dout_c, stdv, F, B, T, Tp,
                       Jp, dP);
    return idv_launch_status();
}

// Two-latent enhancement estimators (test_se_cvaefinetune.py:85-135, used at :283-305): mode 0 real_imag_mask, 1 complex_mask,
// 2 phase_mask.  speech_c / noise_c: the decoders' `predict` outputs, interleaved complex [B*ns][F][T][2]; X: the noisy STFT with
// element strides (sb, sf, st, sr) for [b][f][t][re/im]; out: planar [2][F][Jp] spectrum for idv_pw_gemm + idv_istft_ola, out_c
// (may be NULL): interleaved complex [B][F][T][2].
extern "C" int idv_outtype_estimate(const float* speech_c, const float* noise_c, const float* X, long long sb, long long sf,
                                    long long st_, long long sr, int mode, int ns, int B, int F, int T, int Tp, int Jp, float* out,
                                    float* out_c, void* stream) {
    if (!speech_c || !noise_c || !X || !out || mode < 0 || mode > 2 || ns < 1 || B <= 0 || F <= 0 || T <= 0 || Tp < T + 1 ||
        Jp < B * Tp)
        return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * 2 * (size_t)F * Jp, st) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(outtype_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, st, speech_c, noise_c, X, sb, sf, st_, sr,
                       mode, ns, B, F, T, Tp, Jp, out, out_c);
    return idv_launch_status();
}

extern "C" int idv_planar_to_complex(const float* act, float* out_c, int F, int B, int T, int Tp, int Jp, void* stream) {
    if (!act || !out_c || F <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    hipLaunchKernelGGL(planar_to_complex_kernel, dim3(grid_for((long long)B * F * T)), dim3(256), 0, (hipStream_t)stream, act,
                       out_c, F, B, T, Tp, Jp);
    return idv_launch_status();
}

extern "C" int idv_cbn_finalize(const double* stats, double count, const float* gamma_rr, const float* gamma_ri,
                                const float* gamma_ii, const float* beta_r, const float* beta_i, int C, int first_call,
                                float momentum, float* running_mean_r, float* running_mean_i, float* Vrr, float* Vri,
                                float* Vii, float* moments, float* fold, void* stream) {
    if (!stats || count <= 0 || !gamma_rr || !gamma_ri || !gamma_ii || !beta_r || !beta_i || !fold || C <= 0) return IDV_EINVAL;
    if (running_mean_r && (!running_mean_i || !Vrr || !Vri || !Vii)) return IDV_EINVAL;
    hipLaunchKernelGGL(cbn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, count, gamma_rr,
                       gamma_ri, gamma_ii, beta_r, beta_i, C, first_call, momentum, running_mean_r, running_mean_i, Vrr, Vri,
                       Vii, moments, fold);
    return idv_launch_status();
}

extern "C" int idv_cbn_stats(const float* act, int C, int F, int B, int Tp, int Jp, int t_valid, double* stats, void* stream) {
    if (!act || !stats || C <= 0 || F <= 0 || B <= 0) return IDV_EINVAL;
    const int J = B * Tp;
    int gx = (J + 255) / 256;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(cbn_stats_kernel, dim3(gx, C * F), dim3(256), 0, (hipStream_t)stream, act, C, F, B, Tp, Jp, t_valid, stats);
    return idv_launch_status();
}

extern "C" int idv_cbn_apply_prelu(float* act, const float* fold, const float* prelu_slope, int C, int F, int B, int Tp,
                                   int Jp, int t_valid, void* stream) {
    if (!act || !fold || C <= 0 || F <= 0 || B <= 0) return IDV_EINVAL;
    const int J = B * Tp;
    int gx = (J + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(cbn_apply_prelu_kernel, dim3(gx, C * F), dim3(256), 0, (hipStream_t)stream, act, fold, prelu_slope, C,
                       F, B, Tp, Jp, t_valid);
    return idv_launch_status();
}

extern "C" int idv_reparam(const float* lat, int Hl, int off_miu, int off_ls, int off_dl, int zdim, const float* eps_r,
                           const float* eps_i, int ns, int B, int T, int Tp, int Jp, float* z, int Jpz, void* stream) {
    if (!lat || !eps_r || !eps_i || !z || zdim <= 0 || ns <= 0 || B <= 0 || T <= 0) return IDV_EINVAL;
    if (off_miu + zdim > Hl || off_ls + zdim > Hl || off_dl + zdim > Hl || Jpz < B * ns * Tp) return IDV_EINVAL;
    hipLaunchKernelGGL(zero_guard_kernel, dim3(grid_for(2LL * zdim * B * ns)), dim3(256), 0, (hipStream_t)stream, z, 2 * zdim,
                       B * ns, Tp, Jpz);
    hipLaunchKernelGGL(reparam_kernel, dim3(grid_for((long long)B * zdim * T)), dim3(256), 0, (hipStream_t)stream, lat, Hl,
                       off_miu, off_ls, off_dl, zdim, eps_r, eps_i, ns, B, T, Tp, Jp, z, Jpz);
    return idv_launch_status();
}

__global__ void stats_collapse_kernel(const double* __restrict__ work, int rep, int n, double* __restrict__ stats) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double acc = 0;
    for (int r = 0; r < rep; ++r) acc += work[(size_t)r * n + e];
    stats[e] += acc;
}

int idv_launch_stats_collapse(const double* work, int rep, int n, double* stats, hipStream_t st) {
    hipLaunchKernelGGL(stats_collapse_kernel, dim3((n + 255) / 256), dim3(256), 0, st, work, rep, n, stats);
    return idv_launch_status();
}

