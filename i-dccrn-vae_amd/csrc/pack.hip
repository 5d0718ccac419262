// Weight preparation kernels: reference-layout parameters -> MFMA fragment order.
#include "common.hpp"
#include "../../include/idccrn_hip.h"
#include <math.h>

namespace {

// fragment element (mt, ks, lane): lane = h*32 + ml supplies W'[mt*32+ml][k-pair ks][h]
__global__ void pack_cconv_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im,
                                  const float* __restrict__ b_re, const float* __restrict__ b_im,
                                  const float* __restrict__ fold, int Cout, int Cin_total, int Cin_used, int transposed,
                                  int KS, int Mtiles, float* __restrict__ wfrag, float* __restrict__ bias_out, int conj) {
    const long long n = (long long)Mtiles * KS * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long long t = idx >> 6;
        const int ks = (int)(t % KS), mt = (int)(t / KS);
        const int h = lane >> 5, m = mt * 32 + (lane & 31);
        const int co = m >> 1, ro = m & 1;
        const int cc = ks / 5, kf = ks % 5;
        const int ci = cc >> 1, ri = cc & 1;
        float v = 0.f;
        if (co < Cout && ci < Cin_used) {
            // conv: taps (x[t-1], x[t]) pair with kt = (0, 1); transposed conv: out[t] = W[..,0] x[t] + W[..,1] x[t-1]
            const int kt = transposed ? 1 - h : h;
            const size_t off = transposed ? (((size_t)ci * Cout + co) * 5 + kf) * 2 + kt
                                          : (((size_t)co * Cin_total + ci) * 5 + kf) * 2 + kt;
            const float wr = w_re[off], wi = conj ? -w_im[off] : w_im[off];
            // rows of [[Wr, -Wi], [Wi, Wr]] for this input plane
            const float top = ri == 0 ? wr : -wi;   // contributes to the real output
            const float bot = ri == 0 ? wi : wr;    // contributes to the imag output
            if (fold) {
                const float* z = fold + (size_t)co * 6;
                v = ro == 0 ? z[0] * top + z[1] * bot : z[2] * top + z[3] * bot;
            } else {
                v = ro == 0 ? top : bot;
            }
        }
        wfrag[idx] = v;
    }
    const int nb = Mtiles * 32;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < nb; m += gridDim.x * blockDim.x) {
        const int co = m >> 1, ro = m & 1;
        float v = 0.f;
        if (co < Cout && b_re) {
            const float top = b_re[co] - b_im[co], bot = b_re[co] + b_im[co];
            if (fold) {
                const float* z = fold + (size_t)co * 6;
                v = ro == 0 ? z[0] * top + z[1] * bot + z[4] : z[2] * top + z[3] * bot + z[5];
            } else {
                v = ro == 0 ? top : bot;
            }
        }
        bias_out[m] = v;
    }
}

// generic row-major W[M][K]; rowmap: 0 identity, 1 LSTM (two stacked [4H][K] matrices, gate permutation)
struct PwSrc {
    const float* w0; const float* w1;     // LSTM: lstm_re / lstm_im
    const float* b0a; const float* b0b;   // bias terms (summed); may be null
    const float* b1a; const float* b1b;
    int H;                                 // LSTM hidden size (rowmap 1)
};

__device__ __forceinline__ int lstm_src_row(int colp, int H) {
    // colp = ((u/16)*4 + g)*16 + u%16  ->  torch gate row g*H + u
    const int ub = colp >> 6, g = (colp >> 4) & 3, ul = colp & 15;
    return g * H + ub * 16 + ul;
}

__global__ void pack_pw_kernel(PwSrc s, int rowmap, int M, int K, int KS, int Mtiles, float* __restrict__ wfrag,
                               float* __restrict__ bias_out) {
    const long long n = (long long)Mtiles * KS * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long long t = idx >> 6;
        const int ks = (int)(t % KS), mt = (int)(t / KS);
        const int m = mt * 32 + (lane & 31), k = 2 * ks + (lane >> 5);
        float v = 0.f;
        if (m < M && k < K) {
            if (rowmap == 0) {
                v = s.w0[(size_t)m * K + k];
            } else {
                const int set = m / (4 * s.H), colp = m % (4 * s.H);
                const float* w = set ? s.w1 : s.w0;
                v = w[(size_t)lstm_src_row(colp, s.H) * K + k];
            }
        }
        wfrag[idx] = v;
    }
    const int nb = Mtiles * 32;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < nb; m += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (m < M) {
            if (rowmap == 0) {
                v = s.b0a ? s.b0a[m] : 0.f;
            } else {
                const int set = m / (4 * s.H), row = lstm_src_row(m % (4 * s.H), s.H);
                v = set ? s.b1a[row] + s.b1b[row] : s.b0a[row] + s.b0b[row];
            }
        }
        bias_out[m] = v;
    }
}

// recurrent weights: whh_frag[set][tile][kk][lane], tile = 16 gate columns (colp order), lane l supplies
// B[k = 4*kk + (l>>4)][n = l&15] = W_hh[row(colp = tile*16 + n)][k]
// vec4 (H % 64 == 0, H != 128: the sizes the per-step kernel serves): [set][tile][kk/4][lane][4], i.e. the four k-steps
// 4*kk4 + j of one lane side by side, so the step kernel loads 16 bytes per 4 k-steps
__global__ void pack_lstm_hh_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im, int H, int vec4,
                                    float* __restrict__ out) {
    const int NT = H / 4;   // tiles of 16 columns over 4H
    const int KK = H / 4;   // k-steps of 4
    const long long n = 2LL * NT * KK * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        int lane, kk;
        long long t;
        if (vec4) {
            const int j = (int)(idx & 3);
            lane = (int)((idx >> 2) & 63);
            t = idx >> 8;
            kk = 4 * (int)(t % (KK / 4)) + j; t /= (KK / 4);
        } else {
            lane = (int)(idx & 63);
            t = idx >> 6;
            kk = (int)(t % KK); t /= KK;
        }
        const int tile = (int)(t % NT);
        const int set = (int)(t / NT);
        const int colp = tile * 16 + (lane & 15), k = 4 * kk + (lane >> 4);
        const float* w = set ? w_im : w_re;
        out[idx] = w[(size_t)lstm_src_row(colp, H) * H + k];
    }
}

// split-bf16 recurrent weights appended to the fp32 fragments: [set][tile][kb][hi|lo][lane] x 16 B, lane l holds
// B[k = 32*kb + 8*(l>>4) + j][n = l&15] = W_hh[row(colp = tile*16 + n)][k], j = 0..7
__global__ void pack_lstm_hh_bf16_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im, int H,
                                         uint4* __restrict__ out) {
    const int NTl = H / 4, KB = H / 32;
    const long long n = 2LL * NTl * KB * 2 * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long long t = idx >> 6;
        const int sp = (int)(t & 1); t >>= 1;
        const int kb = (int)(t % KB); t /= KB;
        const int tile = (int)(t % NTl);
        const int set = (int)(t / NTl);
        const float* w = set ? w_im : w_re;
        const int row = lstm_src_row(tile * 16 + (lane & 15), H);
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float x = w[(size_t)row * H + 32 * kb + 8 * (lane >> 4) + 2 * q + e];
                const float hi = (float)(__bf16)x;
                v[e] = sp == 0 ? x : x - hi;
            }
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            bf2 pk = {(__bf16)v[0], (__bf16)v[1]};
            o[q] = __builtin_bit_cast(unsigned, pk);
        }
        out[idx] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__global__ void cbn_fold_kernel(const float* __restrict__ mom, const float* __restrict__ g_rr, const float* __restrict__ g_ri,
                                const float* __restrict__ g_ii, const float* __restrict__ b_r, const float* __restrict__ b_i,
                                int C, float* __restrict__ fold) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float eps = 1e-5f;
    const float mu_r = mom[c], mu_i = mom[C + c], Vrr = mom[2 * C + c], Vri = mom[3 * C + c], Vii = mom[4 * C + c];
    float delta = Vrr * Vii - Vri * Vri + eps;
    delta = fmaxf(delta, 1e-8f);
    const float s = sqrtf(delta);
    const float t = sqrtf(Vrr + Vii + 2.f * s + eps);
    const float inv = 1.0f / (s * t + eps);
    const float Wrr = (Vii + s) * inv, Wii = (Vrr + s) * inv, Wri = -Vri * inv;
    const float Zrr = g_rr[c] * Wrr + g_ri[c] * Wri;
    const float Zri = g_rr[c] * Wri + g_ri[c] * Wii;
    const float Zir = g_ri[c] * Wrr + g_ii[c] * Wri;
    const float Zii = g_ri[c] * Wri + g_ii[c] * Wii;
    float* z = fold + (size_t)c * 6;
    z[0] = Zrr; z[1] = Zri; z[2] = Zir; z[3] = Zii;
    z[4] = b_r[c] - (Zrr * mu_r + Zri * mu_i);
    z[5] = b_i[c] - (Zir * mu_r + Zii * mu_i);
}

__global__ void make_dft_kernel(int n_fft, int win, int hop, int T, float* __restrict__ w_fwd, float* __restrict__ w_inv,
                                float* __restrict__ env_inv) {
    const int F = n_fft / 2 + 1, left = (n_fft - win) / 2;
    const double two_pi = 6.283185307179586476925286766559;
    const long long n1 = 2LL * F * win;
    const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x, gsz = (long long)gridDim.x * blockDim.x;
    for (long long idx = gid; idx < n1; idx += gsz) {
        const int n = (int)(idx % win);
        const int m = (int)(idx / win);
        const int ri = m / F, f = m % F;
        const double wn = 0.5 - 0.5 * cos(two_pi * n / win);            // hann, periodic
        const long long ph = ((long long)f * (n + left)) % n_fft;       // exact phase reduction
        const double ang = two_pi * (double)ph / n_fft;
        w_fwd[idx] = (float)(ri == 0 ? wn * cos(ang) : -wn * sin(ang));
        // inverse: y[n'] = 1/N sum_f c_f (Xr cos - Xi sin), then * window
        const double cf = (f == 0 || f == n_fft / 2) ? 1.0 : 2.0;
        w_inv[(size_t)n * (2 * F) + m] = (float)((ri == 0 ? cos(ang) : -sin(ang)) * wn * cf / n_fft);
    }
    const int total = n_fft + hop * (T - 1);
    for (long long p = gid; p < total; p += gsz) {
        double e = 0.0;
        for (int t = 0; t < T; ++t) {
            const long long n = p - (long long)hop * t - left;
            if (n >= 0 && n < win) {
                const double wn = 0.5 - 0.5 * cos(two_pi * n / win);
                e += wn * wn;
            }
        }
        env_inv[p] = e > 1e-11 ? (float)(1.0 / e) : 0.f;
    }
}

inline int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int idv_abi_version(void) { return IDV_ABI_VERSION; }

extern "C" int idv_cbn_fold(const float* moments, const float* gamma_rr, const float* gamma_ri, const float* gamma_ii,
                            const float* beta_r, const float* beta_i, int C, float* fold, void* stream) {
    if (!moments || !gamma_rr || !gamma_ri || !gamma_ii || !beta_r || !beta_i || !fold || C <= 0) return IDV_EINVAL;
    hipLaunchKernelGGL(cbn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, moments, gamma_rr,
                       gamma_ri, gamma_ii, beta_r, beta_i, C, fold);
    return idv_launch_status();
}

extern "C" int idv_pack_cconv(const float* w_re, const float* w_im, const float* b_re, const float* b_im,
                              const float* fold, int Cout, int Cin_total, int Cin_used, int transposed, float* wfrag,
                              float* bias_out, void* stream) {
    if (!w_re || !w_im || !b_re || !b_im || !wfrag || !bias_out || Cout <= 0 || Cin_used <= 0 || Cin_used > Cin_total)
        return IDV_EINVAL;
    const int cck = idv_cconv_cck(Cin_used);
    const int CCp = ((2 * Cin_used + cck - 1) / cck) * cck;
    const int KS = CCp * 5;
    const int Mtiles = ((2 * Cout + 127) / 128) * 4;
    hipLaunchKernelGGL(pack_cconv_kernel, dim3(grid_for((long long)Mtiles * KS * 64)), dim3(256), 0, (hipStream_t)stream,
                       w_re, w_im, b_re, b_im, fold, Cout, Cin_total, Cin_used, transposed, KS, Mtiles, wfrag, bias_out, 0);
    return idv_launch_status();
}

// Weights of the ADJOINT operator (data gradient): the same parameter tensor read with the other layout
// (conv weights [Cout][Cin] as transposed-conv weights [Cin' = Cout][Cout' = Cin] and vice versa), imaginary part
// negated, no bias.  `transposed` is the mode of the adjoint operator itself.
extern "C" int idv_pack_cconv_adjoint(const float* w_re, const float* w_im, int Cout, int Cin_total, int Cin_used,
                                      int transposed, float* wfrag, float* bias_out, void* stream) {
    if (!w_re || !w_im || !wfrag || !bias_out || Cout <= 0 || Cin_used <= 0 || Cin_used > Cin_total) return IDV_EINVAL;
    const int cck = idv_cconv_cck(Cin_used);
    const int CCp = ((2 * Cin_used + cck - 1) / cck) * cck;
    const int KS = CCp * 5;
    const int Mtiles = ((2 * Cout + 127) / 128) * 4;
    hipLaunchKernelGGL(pack_cconv_kernel, dim3(grid_for((long long)Mtiles * KS * 64)), dim3(256), 0, (hipStream_t)stream,
                       w_re, w_im, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Cout, Cin_total, Cin_used,
                       transposed, KS, Mtiles, wfrag, bias_out, 1);
    return idv_launch_status();
}

extern "C" int idv_pack_pw(const float* w, const float* bias, int M, int K, float* wfrag, float* bias_out, void* stream) {
    if (!w || !wfrag || !bias_out || M <= 0 || K <= 0) return IDV_EINVAL;
    const int KS = ((K + 7) / 8) * 4;
    const int Mtiles = ((M + 127) / 128) * 4;
    PwSrc s{w, nullptr, bias, nullptr, nullptr, nullptr, 0};
    hipLaunchKernelGGL(pack_pw_kernel, dim3(grid_for((long long)Mtiles * KS * 64)), dim3(256), 0, (hipStream_t)stream, s, 0,
                       M, K, KS, Mtiles, wfrag, bias_out);
    return idv_launch_status();
}

extern "C" int idv_pack_lstm_ih(const float* w_ih_re, const float* b_ih_re, const float* b_hh_re, const float* w_ih_im,
                                const float* b_ih_im, const float* b_hh_im, int H, int K, float* wfrag, float* bias_out,
                                void* stream) {
    if (!w_ih_re || !w_ih_im || !b_ih_re || !b_hh_re || !b_ih_im || !b_hh_im || !wfrag || !bias_out || H <= 0 || (H % 16) || K <= 0)
        return IDV_EINVAL;
    const int M = 8 * H;
    const int KS = ((K + 7) / 8) * 4;
    const int Mtiles = ((M + 127) / 128) * 4;
    PwSrc s{w_ih_re, w_ih_im, b_ih_re, b_hh_re, b_ih_im, b_hh_im, H};
    hipLaunchKernelGGL(pack_pw_kernel, dim3(grid_for((long long)Mtiles * KS * 64)), dim3(256), 0, (hipStream_t)stream, s, 1,
                       M, K, KS, Mtiles, wfrag, bias_out);
    return idv_launch_status();
}

extern "C" int idv_pack_lstm_hh(const float* w_hh_re, const float* w_hh_im, int H, float* whh_frag, void* stream) {
    if (!w_hh_re || !w_hh_im || !whh_frag || H <= 0 || (H % 16)) return IDV_EINVAL;
    const int vec4 = (H % 64 == 0 && H != 128) ? 1 : 0;        // lstm.hip launch_rec: per-step kernel, VEC4 instantiation
    hipLaunchKernelGGL(pack_lstm_hh_kernel, dim3(grid_for(2LL * 4 * H * H)), dim3(256), 0, (hipStream_t)stream, w_hh_re,
                       w_hh_im, H, vec4, whh_frag);
    if (H % 32 == 0)
        hipLaunchKernelGGL(pack_lstm_hh_bf16_kernel, dim3(grid_for(2LL * H * H / 4)), dim3(256), 0, (hipStream_t)stream, w_hh_re,
                           w_hh_im, H, (uint4*)(whh_frag + (size_t)2 * 4 * H * H));
    return idv_launch_status();
}

extern "C" int idv_make_dft(int n_fft, int win, int hop, int T, float* w_fwd, float* w_inv, float* env_inv, void* stream) {
    if (!w_fwd || !w_inv || !env_inv || n_fft <= 0 || (n_fft & 1) || win <= 0 || win > n_fft || hop <= 0 || T <= 0)
        return IDV_EINVAL;
    hipLaunchKernelGGL(make_dft_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, n_fft, win, hop, T, w_fwd, w_inv, env_inv);
    return idv_launch_status();
}
