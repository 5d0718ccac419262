// Last decoder block in exact fp32: complex transposed conv with ONE output channel (Cout = 1, M = 2 output rows;
// reference: causal_ComplexConvTranspose2d.forward, model/complex_progress.py:244-250; Decoder.forward,
// model/pvae_module.py:88-93).  On the MFMA kernel such a layer fills 2 of 32 rows (8 TFLOP/s, 3.4 ms of the 123 ms
// headline step for 0.3 % of its flops) while its work is one pass over the 128 input planes: memory-shaped.  So it runs on
// the vector ALU with the contraction re-associated as in cgemm_c1.hip (the five frequency taps move from K into M):
//     P[kf, ro][m][j] = sum_{cc, h} W'[ro][cc][kf][h] * x[cc][m][j + h + tshift]             (10 values per input position)
//     out[ro][2m  ][j] = P[0][m+1] + P[2][m] + P[4][m-1] ,   out[ro][2m+1][j] = P[1][m+1] + P[3][m]
// A lane owns four columns (256 apart) and walks a segment of input rows m with a three-row window of P in registers; the weights
// are wave-uniform and come through the scalar cache straight from the MFMA fragment buffer (element (cc, kf, h, ro) of
// row tile 0 = wfrag[(cc*5 + kf)*64 + h*32 + ro], see pack.hip), i.e. every FMA is one SGPR x one VGPR.  Epilogue as
// cgemm_kernel: bias, PReLU, outputs outside tp in [1, t_valid] zeroed, optional train-mode moments.
#include "cgemm.hpp"
#include "../../include/idccrn_hip.h"

namespace {

constexpr int C1F_NC = 4;         // columns per lane (256 apart): each weight fetched through the scalar cache feeds 4 FMAs
constexpr int C1F_PR = 4;         // planes per round: their 2 x C1F_NC x C1F_PR loads are in flight together

template <bool STATS>
__global__ __launch_bounds__(256) void ctconv_c1_f32_kernel(const CgemmArgs a, int rseg) {     // rseg input rows (+ 2 halo rows) per workgroup
    const int tid = threadIdx.x;
    const int m_lo = blockIdx.y * rseg;
    int m_hi = m_lo + rseg;                       // rows [m_lo, m_hi) are emitted by this workgroup
    if (m_hi > a.Fin) m_hi = a.Fin;
    // per column: the two input columns of the time taps, and the same columns of the (possibly repeated) skip source
    int jcol[C1F_NC], c0[C1F_NC][2], c1[C1F_NC][2];
    bool inb[C1F_NC], keep[C1F_NC], cok[C1F_NC][2];
#pragma unroll
    for (int n = 0; n < C1F_NC; ++n) {
        const int j = (blockIdx.x * C1F_NC + n) * 256 + tid;
        jcol[n] = j;
        inb[n] = j < a.J;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = j + h + a.tshift;
            cok[n][h] = inb[n] && c >= 0 && c < a.J;
            c0[n][h] = cok[n][h] ? c : 0;
            const int b = c0[n][h] / a.Tp;
            c1[n][h] = (b / a.x1_div) * a.Tp + (c0[n][h] - b * a.Tp);
        }
        const int tp = j % a.Tp;
        keep[n] = inb[n] && tp >= 1 && tp <= a.t_valid;
    }
    const float slope = a.slope ? *a.slope : 1.0f;
    const bool has_act = a.slope != nullptr;
    const float bias0 = a.bias[0], bias1 = a.bias[1];
    const size_t plane0 = (size_t)a.Fin * a.Jp, plane1 = (size_t)a.Fin * a.Jp1;
    const int CC = 2 * (a.C0 + a.C1);

    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float pa[C1F_NC][10], pb[C1F_NC][10], pc[C1F_NC][10];       // P of rows m-1, m, m+1 as [kf*2 + ro]
#pragma unroll
    for (int n = 0; n < C1F_NC; ++n)
#pragma unroll
        for (int i = 0; i < 10; ++i) pa[n][i] = pb[n][i] = pc[n][i] = 0.f;

    auto contract = [&](int m, float (&p)[C1F_NC][10]) {
#pragma unroll
        for (int n = 0; n < C1F_NC; ++n)
#pragma unroll
            for (int i = 0; i < 10; ++i) p[n][i] = 0.f;
        if (m < 0 || m >= a.Fin) return;             // uniform
        // the loop count is a run-time value, which hipcc does not unroll: C1F_PR planes per round by hand, their loads
        // issued together (one plane per round left every wave waiting out a memory latency per 20 FMAs)
        for (int cc0 = 0; cc0 < CC; cc0 += C1F_PR) {
            float xv[C1F_PR][C1F_NC][2];
#pragma unroll
            for (int u = 0; u < C1F_PR; ++u) {
                const int cc = cc0 + u < CC ? cc0 + u : CC - 1;          // a ragged last round re-reads the last plane
                const int ci = cc >> 1, ri = cc & 1;
                const bool first = ci < a.C0;                            // uniform
                const float* r = first ? a.x0 + (size_t)(ri * a.C0 + ci) * plane0 + (size_t)m * a.Jp
                                       : a.x1 + (size_t)(ri * a.C1 + (ci - a.C0)) * plane1 + (size_t)m * a.Jp1;
#pragma unroll
                for (int n = 0; n < C1F_NC; ++n) {
                    xv[u][n][0] = r[first ? c0[n][0] : c1[n][0]];
                    xv[u][n][1] = r[first ? c0[n][1] : c1[n][1]];
                }
            }
#pragma unroll
            for (int u = 0; u < C1F_PR; ++u) {
                const bool live = cc0 + u < CC;                          // uniform
                const float* w = a.wfrag + (size_t)(live ? cc0 + u : 0) * 5 * 64;    // wave-uniform address: scalar loads
#pragma unroll
                for (int kf = 0; kf < 5; ++kf) {
                    const float w00 = live ? w[kf * 64] : 0.f, w01 = live ? w[kf * 64 + 1] : 0.f;
                    const float w10 = live ? w[kf * 64 + 32] : 0.f, w11 = live ? w[kf * 64 + 33] : 0.f;
#pragma unroll
                    for (int n = 0; n < C1F_NC; ++n) {
                        const float x0v = cok[n][0] ? xv[u][n][0] : 0.f, x1v = cok[n][1] ? xv[u][n][1] : 0.f;
                        p[n][kf * 2] += w00 * x0v;
                        p[n][kf * 2 + 1] += w01 * x0v;
                        p[n][kf * 2] += w10 * x1v;
                        p[n][kf * 2 + 1] += w11 * x1v;
                    }
                }
            }
        }
    };
    auto emit = [&](int n, int fo, float yr, float yi) {
        if (fo < 0 || fo >= a.Fout) return;
        yr += bias0;
        yi += bias1;
        if (has_act) {
            yr = yr >= 0.f ? yr : slope * yr;
            yi = yi >= 0.f ? yi : slope * yi;
        }
        yr = keep[n] ? yr : 0.f;
        yi = keep[n] ? yi : 0.f;
        if (inb[n]) {
            a.out[(size_t)fo * a.Jp + jcol[n]] = yr;                            // plane ro * Cout + co with Cout = 1
            a.out[((size_t)a.Fout + fo) * a.Jp + jcol[n]] = yi;
        }
        if (STATS && keep[n]) {
            st[0] += yr; st[1] += yi; st[2] += yr * yr; st[3] += yi * yi; st[4] += yr * yi;
        }
    };

    contract(m_lo - 1, pa);
    contract(m_lo, pb);
    for (int m = m_lo; m < m_hi; ++m) {
        contract(m + 1, pc);
        // out[2m] = P[0][m+1] + P[2][m] + P[4][m-1];  out[2m+1] = P[1][m+1] + P[3][m]
#pragma unroll
        for (int n = 0; n < C1F_NC; ++n) {
            emit(n, 2 * m, pc[n][0] + pb[n][4] + pa[n][8], pc[n][1] + pb[n][5] + pa[n][9]);
            emit(n, 2 * m + 1, pc[n][2] + pb[n][6], pc[n][3] + pb[n][7]);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                pa[n][i] = pb[n][i];
                pb[n][i] = pc[n][i];
            }
        }
    }
    if (STATS) {
        __shared__ float red[4][5];
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            float t = st[s];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) red[wave][s] = t;
        }
        __syncthreads();
        if (tid < 5) atomicAdd(&a.stats[tid], (double)red[0][tid] + (double)red[1][tid] + (double)red[2][tid] + (double)red[3][tid]);
    }
}

}  // namespace

// launcher used by idv_cconv2d_fwd (cgemm.hip) for transposed && Cout == 1
int idv_launch_ctconv_c1_f32(const CgemmArgs& a, hipStream_t st) {
    // 226 VGPRs: two workgroups per CU, 512 resident on the chip -> row segments sized for ONE round of at most 512
    // workgroups (fewer, longer segments also re-read fewer halo rows)
    const int colblocks = (a.J + 256 * C1F_NC - 1) / (256 * C1F_NC);
    int segs = 512 / colblocks;
    if (segs < 1) segs = 1;
    if (segs > a.Fin) segs = a.Fin;
    const int rseg = (a.Fin + segs - 1) / segs;
    dim3 grid((unsigned)colblocks, (unsigned)((a.Fin + rseg - 1) / rseg));
    if (a.stats)
        hipLaunchKernelGGL(ctconv_c1_f32_kernel<true>, grid, dim3(256), 0, st, a, rseg);
    else
        hipLaunchKernelGGL(ctconv_c1_f32_kernel<false>, grid, dim3(256), 0, st, a, rseg);
    return idv_launch_status();
}
