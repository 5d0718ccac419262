// Weight gradients of the complex conv / transposed conv / point-wise contractions (backward row):
// what torch.autograd computes behind `loss.backward()` in the reference's train steps
// (supervised_dccrn/train.py:239-243, pretrained_vaes/train.py:296-301, train_nsvae.py:557-561,
// train_second_phase_decoder.py:420-433) for nn.Conv2d / nn.ConvTranspose2d (model/complex_progress.py:8-36, :222-279),
// nn.Linear (:77-89) and the nn.LSTM input projections (:39-74).
//
// One contraction for all of them.  With S the "small" planar tensor (rows fs) and L the "large" one
// (rows fl = 2*fs + kf - 2; conv: S = dy, L = x; transposed conv: S = x, L = dy):
//
//   G[kf][kt][sp][lp] = sum_{fs, j} S[sp][fs][j] * L[lp][2*fs + kf - 2][j + kt + dt0]
//
// i.e. a GEMM with M = S planes, N = L planes x 10 taps and K = Fs x J (millions) -> split-K: every workgroup
// owns one (128 S planes x 32 L planes x 10 taps) output tile over a contiguous range of column tiles and
// writes its partial tile to a workspace; the unpack kernel sums the partials in a fixed order (deterministic,
// no atomics) and folds the four real products of a complex pair into (dW_re, dW_im).
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32).  As in the forward kernel the two k of one instruction are two
// adjacent columns: lanes 0-31 read LDS column c, lanes 32-63 column c+1, so the time tap is a free column
// offset and all LDS reads are conflict free (odd row and plane pitches against the 32 banks of ds_read_b32).
#include "common.hpp"
#include "wgrad_common.hpp"
#include "../../include/idccrn_hip.h"

namespace {


// WG_JT columns per step; LDS row pitch WG_JT + 3 words: odd, so the 32 planes one 32-lane half reads with ds_read_b32
// (bank = word address % 32, lanes l and l+32 never conflict) land on 32 distinct banks, for the S tile (plane pitch =
// row pitch) and for the L tile (plane pitch = KF * row pitch, KF odd)
// KF x KT taps; MT_W / NT_W 32-plane tiles per wave along S / L; WM x WN waves (4 in all).
// The accumulators (MT_W * NT_W * KF * KT tiles of 16 registers) must fit the 256 AGPRs: beyond that hipcc keeps the
// excess in VGPRs and swaps them through a[0:15] around every MFMA (32 v_accvgpr moves + 16 wait states each).  The conv
// instantiation therefore holds 10 tiles per wave (one 32 x 32 plane tile x 10 taps) and runs two workgroups per CU.
template <int KF, int KT, int MT_W, int NT_W, int WM, int WN, int WG_JT, int OCC>
__global__ __launch_bounds__(256, OCC) void wgrad_kernel(const WgradArgs a) {
    static_assert(WM * WN == 4 && MT_W * NT_W * KF * KT <= 16, "4 waves; accumulators within the AGPR file");
    constexpr int WG_PS = WG_JT + 3;
    constexpr int Q4 = WG_JT / 4;                             // float4 slots per row
    constexpr int KUNR = (KF * KT > 1) ? 2 : 4;
    static_assert((WG_PS & 1) == 1 && ((KF * WG_PS) & 1) == 1, "LDS pitches must be odd");
    constexpr int MS = WM * MT_W * 32, ML = WN * NT_W * 32;
    constexpr int TAPS = KF * KT;
    constexpr int S_SLOTS = MS * (WG_JT / 4);                 // float4 slots of the S tile
    constexpr int L_ROWS = ML * KF;
    constexpr int L_SLOTS = L_ROWS * (WG_JT / 4);
    constexpr int NS4 = (S_SLOTS + 255) / 256, NL4 = (L_SLOTS + 255) / 256;
    constexpr int NH = (L_ROWS * 2 + 255) / 256;              // halo elements (columns j0-1 and j0+32) per thread
    __shared__ float Ssm[MS * WG_PS];
    __shared__ float Lsm[L_ROWS * WG_PS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int split = blockIdx.x, ts = blockIdx.y;
    int tl = blockIdx.z, prod = 0;
    if (a.nprod > 1) {
        prod = tl / a.tilesL;
        tl -= prod * a.tilesL;
    }
    const float* __restrict__ Sg = prod == 0 ? a.S : (prod == 1 ? a.S1 : a.S2);
    const float* __restrict__ Lg = prod == 0 ? a.L : (prod == 1 ? a.L1 : a.L2);
    const int sp0 = ts * MS, lp0 = tl * ML;
    // this split's range of the flattened (column tile, frequency row) step sequence
    int g0, nsteps;
    if (a.nsplit_bal > 0) {
        const long long s0 = (long long)split * a.steps_total / a.nsplit_bal, s1 = (long long)(split + 1) * a.steps_total / a.nsplit_bal;
        g0 = (int)s0;
        nsteps = (int)(s1 - s0);
    } else {
        const int jt0 = split * a.jt_per_split;
        int jt1 = jt0 + a.jt_per_split;
        if (jt1 > a.jtiles) jt1 = a.jtiles;
        g0 = jt0 * a.Fs;
        nsteps = (jt1 > jt0) ? (jt1 - jt0) * a.Fs : 0;
    }

    f32x16 acc[MT_W][NT_W][TAPS];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int n = 0; n < NT_W; ++n)
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][n][t][r] = 0.f;

    f32x4 sreg[NS4], lreg[NL4];
    float hreg[NH];

    // Staging is branch-free: every slot loads unconditionally (invalid slots read element 0 of their tensor, which is
    // mapped memory) and the validity masks are applied when the registers are written to LDS one step later -- a
    // conditional load followed by its mask makes the compiler wait for each load in turn (vmcnt(0) after every one).
    auto load_step = [&](int step) {
        const int g = g0 + step;
        const int jt = g / a.Fs, fs = g - jt * a.Fs;
        const int j0 = jt * WG_JT;
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int sp = sp0 + row, j = j0 + 4 * q;
            const bool ok = (e < S_SLOTS) && (sp < a.Sp) && (j < a.J);
            const size_t off = ok ? ((size_t)sp * a.Fs + fs) * a.JpS + j : 0;
            sreg[i] = *(const f32x4*)(Sg + off);
        }
#pragma unroll
        for (int i = 0; i < NL4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int pl = row / KF, kf = row - pl * KF;
            const int lp = lp0 + pl, fl = (KF == 1) ? fs : 2 * fs + kf - 2, j = j0 + 4 * q;
            const bool ok = (e < L_SLOTS) && (lp < a.Lp) && (fl >= 0) && (fl < a.Fl) && (j < a.J);
            const size_t off = ok ? ((size_t)lp * a.Fl + fl) * a.JpL + j : 0;
            lreg[i] = *(const f32x4*)(Lg + off);
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int e = tid + i * 256;
            const int row = e >> 1, side = e & 1;
            const int pl = row / KF, kf = row - pl * KF;
            const int lp = lp0 + pl, fl = (KF == 1) ? fs : 2 * fs + kf - 2;
            const int j = side ? j0 + WG_JT : j0 - 1;
            const bool ok = (e < L_ROWS * 2) && (lp < a.Lp) && (fl >= 0) && (fl < a.Fl) && (j >= 0) && (j < a.J);
            const size_t off = ok ? ((size_t)lp * a.Fl + fl) * a.JpL + j : 0;
            hreg[i] = Lg[off];
        }
    };
    auto store_step = [&](int step) {
        const int g = g0 + step;
        const int jt = g / a.Fs, fs = g - jt * a.Fs;
        const int j0 = jt * WG_JT;
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int sp = sp0 + row, j = j0 + 4 * q;
            const bool ok = (sp < a.Sp);
            if (e < S_SLOTS) {
                float* d = Ssm + row * WG_PS + 4 * q;
#pragma unroll
                for (int c = 0; c < 4; ++c) d[c] = (ok && j + c < a.J) ? sreg[i][c] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NL4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int pl = row / KF, kf = row - pl * KF;
            const int lp = lp0 + pl, fl = (KF == 1) ? fs : 2 * fs + kf - 2, j = j0 + 4 * q;
            const bool ok = (lp < a.Lp) && (fl >= 0) && (fl < a.Fl);
            if (e < L_SLOTS) {
                float* d = Lsm + row * WG_PS + 1 + 4 * q;                // LDS column c = j - j0 + 1
#pragma unroll
                for (int c = 0; c < 4; ++c) d[c] = (ok && j + c < a.J) ? lreg[i][c] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int e = tid + i * 256;
            const int row = e >> 1, side = e & 1;
            const int pl = row / KF, kf = row - pl * KF;
            const int lp = lp0 + pl, fl = (KF == 1) ? fs : 2 * fs + kf - 2;
            const int j = side ? j0 + WG_JT : j0 - 1;
            const bool ok = (lp < a.Lp) && (fl >= 0) && (fl < a.Fl) && (j >= 0) && (j < a.J);
            if (e < L_ROWS * 2) Lsm[row * WG_PS + (side ? WG_JT + 1 : 0)] = ok ? hreg[i] : 0.f;
        }
    };

    if (nsteps > 0) load_step(0);
    for (int step = 0; step < nsteps; ++step) {
        store_step(step);
        __syncthreads();
        if (step + 1 < nsteps) load_step(step + 1);      // global loads fly under this step's MFMAs
        const float* As = Ssm + (wm * MT_W * 32 + l31) * WG_PS + half;
        const float* Bs = Lsm + ((wn * NT_W * 32 + l31) * KF) * WG_PS + half + 1 + a.dt0;
        // LDS operands one k-step ahead of the MFMAs: the ds_read latency hides behind the previous k-step's MFMAs
        float av[MT_W], bv[NT_W][TAPS], an[MT_W], bn[NT_W][TAPS];
        auto lds_load = [&](int ks, float (&ao)[MT_W], float (&bo)[NT_W][TAPS]) {
            const int col = 2 * ks;
#pragma unroll
            for (int i = 0; i < MT_W; ++i) ao[i] = As[i * 32 * WG_PS + col];
#pragma unroll
            for (int n = 0; n < NT_W; ++n)
#pragma unroll
                for (int kf = 0; kf < KF; ++kf)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) bo[n][kf * KT + kt] = Bs[(n * 32 * KF + kf) * WG_PS + col + kt];
        };
        lds_load(0, av, bv);
#pragma unroll KUNR
        for (int ks = 0; ks < WG_JT / 2; ++ks) {
            lds_load(ks + 1 < WG_JT / 2 ? ks + 1 : ks, an, bn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NT_W; ++n)
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int i = 0; i < MT_W; ++i)
                        acc[i][n][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[n][t], acc[i][n][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT_W; ++i) av[i] = an[i];
#pragma unroll
            for (int n = 0; n < NT_W; ++n)
#pragma unroll
                for (int t = 0; t < TAPS; ++t) bv[n][t] = bn[n][t];
        }
        __syncthreads();
    }

    // partial tile -> workspace (every slot of the padded tile is written, so the workspace needs no clearing)
    float* P = a.part + (size_t)prod * a.prod_stride + (size_t)split * TAPS * a.SpPad * a.LpPad;
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int n = 0; n < NT_W; ++n)
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sp = sp0 + (wm * MT_W + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int lp = lp0 + (wn * NT_W + n) * 32 + l31;
                    P[((size_t)t * a.SpPad + sp) * a.LpPad + lp] = acc[i][n][t][r];
                }
}

// (dW_re, dW_im)[.., kf, kt] from the partial real-block gradients; one thread per complex weight element.
// conv (transposed = 0): S = dy planes (ro*Cout + co), L = x planes (ri*Cx + cil); weight [Cout][Cin_total][5][2]
// tconv (transposed = 1): S = x planes (ri*Cx + cil), L = dy planes (ro*Cout + co); weight [Cin_total][Cout][5][2]
__global__ void wgrad_unpack_conv_kernel(const float* __restrict__ part, int nsplit, int SpPad, int LpPad, int Cout, int Cx,
                                         int Cin_total, int ci_off, int transposed, float* __restrict__ dw_re,
                                         float* __restrict__ dw_im) {
    // thread order (l channel fastest, then s channel, then tap): the partial reads of a wave are contiguous along l
    const int Cs = transposed ? Cx : Cout, Cl = transposed ? Cout : Cx;
    const long long n = (long long)Cs * Cl * 10;
    const size_t plane = (size_t)SpPad * LpPad;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lc = (int)(idx % Cl);
        const int sc = (int)((idx / Cl) % Cs);
        const int tap = (int)(idx / ((long long)Cl * Cs));
        const int s_re = sc, s_im = Cs + sc, l_re = lc, l_im = Cl + lc;
        double rr = 0, ii = 0, ri = 0, ir = 0;      // G[s_re][l_re], G[s_im][l_im], G[s_re][l_im], G[s_im][l_re]
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            const float* P = part + ((size_t)sidx * 10 + tap) * plane;
            rr += P[(size_t)s_re * LpPad + l_re];
            ii += P[(size_t)s_im * LpPad + l_im];
            ri += P[(size_t)s_re * LpPad + l_im];
            ir += P[(size_t)s_im * LpPad + l_re];
        }
        // y_r = Wr x_r - Wi x_i, y_i = Wi x_r + Wr x_i  ->  dWr = dy_r x_r + dy_i x_i,  dWi = -dy_r x_i + dy_i x_r
        // conv: S = dy, L = x;  transposed conv: S = x, L = dy
        const double dwr = rr + ii;
        const double dwi = transposed ? (ri - ir) : (ir - ri);
        const int co = transposed ? lc : sc, ci = ci_off + (transposed ? sc : lc);
        const size_t o = transposed ? (((size_t)ci * Cout + co) * 10 + tap) : (((size_t)co * Cin_total + ci) * 10 + tap);
        dw_re[o] = (float)dwr;
        dw_im[o] = (float)dwi;
    }
}

// ---- three-product (Gauss / Karatsuba) form of the complex weight gradient -------------------------------------------------
// With (p, q) the real / imaginary planes of an S channel and (u, v) those of an L channel, the complex weight gradient needs
//   dWr = p u + q v,   dWi = q u - p v  (conv; transposed conv: p v - q u)      -- four real contractions per channel pair.
// Three suffice:  P1 = p u,  P2 = q v,  P3 = (p - q)(u + v):   dWr = P1 + P2,  dWi = P1 - P2 - P3 (conv) / P3 + P2 - P1 (transposed)
// -- each a contraction between Cs x Cl REAL planes: the wgrad_kernel as it is, launched once with three (S, L) operand pairs.
// This variant needs ONE combined plane set per side, (p - q) and (u + v), written once per call by an elementwise pass
// (HBM-bound); the unpack kernel sums the split-K partials of the three products and applies the signs.
__global__ void wgrad_combine_kernel(const float* __restrict__ re, const float* __restrict__ im, long long n4, float sign,
                                     float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 a = ((const f32x4*)re)[i], b = ((const f32x4*)im)[i];
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = a[c] + sign * b[c];
        ((f32x4*)out)[i] = o;
    }
}

__global__ void wgrad_unpack_gauss_kernel(const float* __restrict__ part, long long prod_stride, int nsplit, int SpPad, int LpPad,
                                          int Cout, int Cx, int Cin_total, int ci_off, int transposed, float* __restrict__ dw_re,
                                          float* __restrict__ dw_im) {
    const int Cs = transposed ? Cx : Cout, Cl = transposed ? Cout : Cx;
    const long long n = (long long)Cs * Cl * 10;
    const size_t plane = (size_t)SpPad * LpPad;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lc = (int)(idx % Cl);
        const int sc = (int)((idx / Cl) % Cs);
        const int tap = (int)(idx / ((long long)Cl * Cs));
        double p1 = 0, p2 = 0, p3 = 0;
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            const float* P = part + ((size_t)sidx * 10 + tap) * plane + (size_t)sc * LpPad + lc;
            p1 += P[0];
            p2 += P[prod_stride];
            p3 += P[2 * prod_stride];
        }
        const double dwr = p1 + p2;
        const double dwi = transposed ? (p3 + p2 - p1) : (p1 - p2 - p3);
        const int co = transposed ? lc : sc, ci = ci_off + (transposed ? sc : lc);
        const size_t o = transposed ? (((size_t)ci * Cout + co) * 10 + tap) : (((size_t)co * Cin_total + ci) * 10 + tap);
        dw_re[o] = (float)dwr;
        dw_im[o] = (float)dwi;
    }
}


// ---- Winograd form of the frequency taps of the weight gradient (round 4) --------------------------------------------------------
// G[kf] = sum_fs S[fs] (x) L[2 fs + kf - 2] for kf = 0..4 spends ten products per PAIR of S rows (c0 = S[2 fp], c1 = S[2 fp + 1]; with
// r_k = L[4 fp - 2 + k]:  G0 += c0 r0 + c1 r2,  G2 += c0 r2 + c1 r4,  G4 += c0 r4 + c1 r6,  G1 += c0 r1 + c1 r3,  G3 += c0 r3 + c1 r5).
// The even taps are the gradient of an F(2,3) correlation with respect to its three taps, the odd taps that of an F(2,2) one; by
// the transposition principle they need the same 4 + 3 products as the forward transforms of cgemm_wino.hip:
//   even:  p1 = c0 (r0 - r4)   p2 = (c0 + c1)(r2 + r4)   p3 = (c0 - c1)(r4 - r2)   p4 = c1 (r6 - r2)
//          G0 = p1 + (p2 + p3)/2     G2 = (p2 - p3)/2     G4 = (p2 + p3)/2 + p4
//   odd:   q1 = c0 (r1 - r3)   q2 = (c0 + c1) r3         q3 = c1 (r5 - r3)           G1 = q1 + q2     G3 = q2 + q3
// -- 7 instead of 10 MFMA products per pair of rows and time tap.  The products are accumulated over all row pairs and columns;
// the tap combination is linear and happens ONCE, in the unpack kernel.  Two kernels (PHW 0: even taps, 4 variants x 2 time taps
// = 8 accumulator tiles per wave; PHW 1: odd taps, 6 tiles) so that both run at two workgroups per CU.  Both operands are
// transformed at the LDS write: S variants (c0, c0 + c1, c0 - c1, c1), L rows as above.  LDS layouts [variant][plane][col] /
// [row][plane][col] keep the odd plane pitch the conflict-free ds_read_b32 pattern needs.
constexpr int WW_MS = 128, WW_ML = 32;

template <int PHW> __device__ __forceinline__ int ww_la(int r) { return PHW == 0 ? (r == 0 ? 0 : (r == 1 ? 2 : (r == 2 ? 4 : 6))) : (r == 0 ? 1 : (r == 1 ? 3 : 5)); }
template <int PHW> __device__ __forceinline__ int ww_lb(int r) { return PHW == 0 ? (r == 0 ? 4 : (r == 1 ? 4 : 2)) : 3; }
template <int PHW> __device__ __forceinline__ float ww_cb(int r) { return PHW == 0 ? (r == 1 ? 1.f : -1.f) : (r == 1 ? 0.f : -1.f); }

// WW_JT columns per step (16 or 32).  The S tile is kept RAW in the LDS (the two rows of the pair); the variants c0 + c1, c0 - c1
// are formed in registers behind the LDS read (two VALU instructions per k-step instead of 16 instead of 8 LDS writes per staged
// slot and two more LDS reads per k-step); the L rows are transformed at the LDS write.
template <int PHW, int OCC, int WW_JT>
__global__ __launch_bounds__(256, OCC) void wgrad_wino_kernel(const WgradArgs a) {
    constexpr int WW_PS = WW_JT + 3;
    constexpr int NV = PHW == 0 ? 4 : 3;                      // S variants = transformed L rows = products per time tap
    constexpr int KT = 2;
    constexpr int Q4 = WW_JT / 4;
    constexpr int S_SLOTS = WW_MS * Q4;                       // raw float4 slots of the S tile (each loads both rows of the pair)
    constexpr int L_SLOTS = NV * WW_ML * Q4;                  // transformed float4 slots of the L tile
    constexpr int NS4 = (S_SLOTS + 255) / 256, NL4 = (L_SLOTS + 255) / 256;
    constexpr int L_HALO = NV * WW_ML * 2;                    // columns j0 - 1 and j0 + JT of every transformed row
    static_assert(L_HALO <= 256 && (WW_PS & 1) == 1, "one halo element per thread; odd LDS pitch");
    __shared__ float Ssm[2 * WW_MS * WW_PS];                  // [row of the pair][plane][col]
    __shared__ float Lsm[NV * WW_ML * WW_PS];                 // [transformed row][plane][col]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int split = blockIdx.x, ts = blockIdx.y;
    int tl = blockIdx.z, prod = 0;
    if (a.nprod > 1) {
        prod = tl / a.tilesL;
        tl -= prod * a.tilesL;
    }
    const float* __restrict__ Sg = prod == 0 ? a.S : (prod == 1 ? a.S1 : a.S2);
    const float* __restrict__ Lg = prod == 0 ? a.L : (prod == 1 ? a.L1 : a.L2);
    const int sp0 = ts * WW_MS, lp0 = tl * WW_ML;
    const int FP = (a.Fs + 1) >> 1;                           // row pairs
    const long long s0 = (long long)split * a.steps_total / a.nsplit_bal, s1 = (long long)(split + 1) * a.steps_total / a.nsplit_bal;
    const int g0 = (int)s0, nsteps = (int)(s1 - s0);

    f32x16 acc[NV][KT];
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][t][r] = 0.f;

    f32x4 s0r[NS4], s1r[NS4], lar[NL4], lbr[NL4];
    float har = 0.f, hbr = 0.f;

    // branch-free staging as wgrad_kernel: unconditional loads (invalid slots read element 0), masks at the LDS write
    auto load_step = [&](int step) {
        const int g = g0 + step;
        const int jt = g / FP, fp = g - jt * FP;
        const int j0 = jt * WW_JT;
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int sp = sp0 + row, j = j0 + 4 * q;
            const bool ok = (e < S_SLOTS) && (sp < a.Sp) && (j < a.J);
            const bool ok1 = ok && (2 * fp + 1 < a.Fs);
            const size_t o0 = ok ? ((size_t)sp * a.Fs + 2 * fp) * a.JpS + j : 0;
            const size_t o1 = ok1 ? o0 + a.JpS : 0;
            s0r[i] = *(const f32x4*)(Sg + o0);
            s1r[i] = *(const f32x4*)(Sg + o1);
        }
#pragma unroll
        for (int i = 0; i < NL4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int r = row / WW_ML, pl = row - r * WW_ML;
            const int lp = lp0 + pl, j = j0 + 4 * q;
            const int fa = 4 * fp - 2 + ww_la<PHW>(r), fb = 4 * fp - 2 + ww_lb<PHW>(r);
            const bool ok = (e < L_SLOTS) && (lp < a.Lp) && (j < a.J);
            const bool oka = ok && fa >= 0 && fa < a.Fl, okb = ok && ww_cb<PHW>(r) != 0.f && fb >= 0 && fb < a.Fl;
            lar[i] = *(const f32x4*)(Lg + (oka ? ((size_t)lp * a.Fl + fa) * a.JpL + j : 0));
            lbr[i] = *(const f32x4*)(Lg + (okb ? ((size_t)lp * a.Fl + fb) * a.JpL + j : 0));
        }
        {
            const int e = tid;
            const int row = e >> 1, side = e & 1;
            const int r = row / WW_ML, pl = row - r * WW_ML;
            const int lp = lp0 + pl;
            const int fa = 4 * fp - 2 + ww_la<PHW>(r < NV ? r : 0), fb = 4 * fp - 2 + ww_lb<PHW>(r < NV ? r : 0);
            const int j = side ? j0 + WW_JT : j0 - 1;
            const bool ok = (e < L_HALO) && (lp < a.Lp) && (j >= 0) && (j < a.J);
            const bool oka = ok && fa >= 0 && fa < a.Fl, okb = ok && ww_cb<PHW>(r < NV ? r : 0) != 0.f && fb >= 0 && fb < a.Fl;
            har = Lg[oka ? ((size_t)lp * a.Fl + fa) * a.JpL + j : 0];
            hbr = Lg[okb ? ((size_t)lp * a.Fl + fb) * a.JpL + j : 0];
        }
    };
    auto store_step = [&](int step) {
        const int g = g0 + step;
        const int jt = g / FP, fp = g - jt * FP;
        const int j0 = jt * WW_JT;
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int sp = sp0 + row, j = j0 + 4 * q;
            const bool ok = (sp < a.Sp), ok1 = ok && (2 * fp + 1 < a.Fs);
            if (e < S_SLOTS) {
                float* d = Ssm + row * WW_PS + 4 * q;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d[c] = (ok && j + c < a.J) ? s0r[i][c] : 0.f;
                    d[WW_MS * WW_PS + c] = (ok1 && j + c < a.J) ? s1r[i][c] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NL4; ++i) {
            const int e = tid + i * 256;
            const int row = e / Q4, q = e - row * Q4;
            const int r = row / WW_ML, pl = row - r * WW_ML;
            const int lp = lp0 + pl, j = j0 + 4 * q;
            const int fa = 4 * fp - 2 + ww_la<PHW>(r), fb = 4 * fp - 2 + ww_lb<PHW>(r);
            const bool ok = (lp < a.Lp);
            const bool oka = ok && fa >= 0 && fa < a.Fl, okb = ok && ww_cb<PHW>(r) != 0.f && fb >= 0 && fb < a.Fl;
            if (e < L_SLOTS) {
                float* d = Lsm + row * WW_PS + 1 + 4 * q;                // LDS column c = j - j0 + 1
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float va = (oka && j + c < a.J) ? lar[i][c] : 0.f, vb = (okb && j + c < a.J) ? lbr[i][c] : 0.f;
                    d[c] = va + ww_cb<PHW>(r) * vb;
                }
            }
        }
        {
            const int e = tid;
            const int row = e >> 1, side = e & 1;
            const int r = row / WW_ML, pl = row - r * WW_ML;
            const int lp = lp0 + pl;
            const int fa = 4 * fp - 2 + ww_la<PHW>(r < NV ? r : 0), fb = 4 * fp - 2 + ww_lb<PHW>(r < NV ? r : 0);
            const int j = side ? j0 + WW_JT : j0 - 1;
            const bool ok = (lp < a.Lp) && (j >= 0) && (j < a.J);
            const bool oka = ok && fa >= 0 && fa < a.Fl, okb = ok && ww_cb<PHW>(r < NV ? r : 0) != 0.f && fb >= 0 && fb < a.Fl;
            if (e < L_HALO) Lsm[row * WW_PS + (side ? WW_JT + 1 : 0)] = (oka ? har : 0.f) + ww_cb<PHW>(r < NV ? r : 0) * (okb ? hbr : 0.f);
        }
    };

    if (nsteps > 0) load_step(0);
    for (int step = 0; step < nsteps; ++step) {
        store_step(step);
        __syncthreads();
        if (step + 1 < nsteps) load_step(step + 1);
        const float* As = Ssm + (wm * 32 + l31) * WW_PS + half;
        const float* Bs = Lsm + l31 * WW_PS + half + 1 + a.dt0;
        float c0v, c1v, bv[NV][KT], c0n, c1n, bn[NV][KT];
        auto lds_load = [&](int ks, float& c0, float& c1, float (&bo)[NV][KT]) {
            const int col = 2 * ks;
            c0 = As[col];
            c1 = As[WW_MS * WW_PS + col];
#pragma unroll
            for (int q = 0; q < NV; ++q)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) bo[q][kt] = Bs[q * WW_ML * WW_PS + col + kt];
        };
        lds_load(0, c0v, c1v, bv);
#pragma unroll 2
        for (int ks = 0; ks < WW_JT / 2; ++ks) {
            lds_load(ks + 1 < WW_JT / 2 ? ks + 1 : ks, c0n, c1n, bn);
            __builtin_amdgcn_sched_barrier(0);
            float av[NV];
            av[0] = c0v;
            av[1] = c0v + c1v;
            if (PHW == 0) { av[2] = c0v - c1v; av[NV - 1] = c1v; } else { av[NV - 1] = c1v; }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int q = 0; q < NV; ++q) acc[q][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q][kt], acc[q][kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            c0v = c0n;
            c1v = c1n;
#pragma unroll
            for (int q = 0; q < NV; ++q)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) bv[q][kt] = bn[q][kt];
        }
        __syncthreads();
    }

    // partial product tiles -> workspace: [product][even: nsplit x 8 planes | odd: nsplit x 6 planes][SpPad][LpPad]
    const size_t plane = (size_t)a.SpPad * a.LpPad;
    float* P = a.part + (size_t)prod * a.prod_stride + (PHW == 0 ? 0 : (size_t)a.nsplit_bal * 8 * plane) + (size_t)split * (2 * NV) * plane;
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sp = sp0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int lp = lp0 + l31;
                P[((size_t)(q * KT + kt)) * plane + (size_t)sp * a.LpPad + lp] = acc[q][kt][r];
            }
}

// the tap (kf, kt) of product `P0` (one Gauss product's region) from the 8 + 6 partial product planes of the Winograd kernels
__device__ __forceinline__ double wgrad_wino_tap(const float* __restrict__ P0, int nsplit, size_t plane, size_t elem, int kf, int kt) {
    double e[4] = {0, 0, 0, 0};
    if ((kf & 1) == 0) {
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            const float* P = P0 + (size_t)sidx * 8 * plane + elem;
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] += P[(size_t)(q * 2 + kt) * plane];
        }
        return kf == 0 ? e[0] + 0.5 * (e[1] + e[2]) : (kf == 2 ? 0.5 * (e[1] - e[2]) : 0.5 * (e[1] + e[2]) + e[3]);
    }
    const float* Q0 = P0 + (size_t)nsplit * 8 * plane;
    for (int sidx = 0; sidx < nsplit; ++sidx) {
        const float* P = Q0 + (size_t)sidx * 6 * plane + elem;
#pragma unroll
        for (int q = 0; q < 3; ++q) e[q] += P[(size_t)(q * 2 + kt) * plane];
    }
    return kf == 1 ? e[0] + e[1] : e[1] + e[2];
}

__global__ void wgrad_unpack_gauss_wino_kernel(const float* __restrict__ part, long long prod_stride, int nsplit, int SpPad, int LpPad,
                                               int Cout, int Cx, int Cin_total, int ci_off, int transposed, float* __restrict__ dw_re,
                                               float* __restrict__ dw_im) {
    const int Cs = transposed ? Cx : Cout, Cl = transposed ? Cout : Cx;
    const long long n = (long long)Cs * Cl * 10;
    const size_t plane = (size_t)SpPad * LpPad;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lc = (int)(idx % Cl);
        const int sc = (int)((idx / Cl) % Cs);
        const int tap = (int)(idx / ((long long)Cl * Cs));
        const int kf = tap >> 1, kt = tap & 1;
        const size_t elem = (size_t)sc * LpPad + lc;
        const double p1 = wgrad_wino_tap(part, nsplit, plane, elem, kf, kt);
        const double p2 = wgrad_wino_tap(part + prod_stride, nsplit, plane, elem, kf, kt);
        const double p3 = wgrad_wino_tap(part + 2 * prod_stride, nsplit, plane, elem, kf, kt);
        const double dwr = p1 + p2;
        const double dwi = transposed ? (p3 + p2 - p1) : (p1 - p2 - p3);
        const int co = transposed ? lc : sc, ci = ci_off + (transposed ? sc : lc);
        const size_t o = transposed ? (((size_t)ci * Cout + co) * 10 + tap) : (((size_t)co * Cin_total + ci) * 10 + tap);
        dw_re[o] = (float)dwr;
        dw_im[o] = (float)dwi;
    }
}

// out[rowmap(m)][k] (+)= sum_splits part[split][0][m][k];  rowmap: 0 identity, 1 LSTM gate order (colp -> g*H + u)
__global__ void wgrad_unpack_plain_kernel(const float* __restrict__ part, int nsplit, int SpPad, int LpPad, int M, int K,
                                          int ldo, int rowmap, int H, int accumulate, float* __restrict__ out) {
    const long long n = (long long)M * K;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K), m = (int)(idx / K);
        double s = 0;
        for (int sidx = 0; sidx < nsplit; ++sidx) s += part[((size_t)sidx * SpPad + m) * LpPad + k];
        int row = m;
        if (rowmap == 1) {
            const int set = m / (4 * H), colp = m - set * 4 * H;
            const int ub = colp >> 6, g = (colp >> 4) & 3, ul = colp & 15;
            row = set * 4 * H + g * H + ub * 16 + ul;
        }
        float* d = out + (size_t)row * ldo + k;
        *d = accumulate ? *d + (float)s : (float)s;
    }
}

__global__ void cconv_bias_grad_kernel(const double* __restrict__ stats, int Cout, float* __restrict__ db_re,
                                       float* __restrict__ db_im) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cout) return;
    const double sr = stats[(size_t)c * 5], si = stats[(size_t)c * 5 + 1];
    db_re[c] = (float)(sr + si);      // out_r carries b_re - b_im, out_i carries b_re + b_im (complex_progress.py:16-18)
    db_im[c] = (float)(si - sr);
}


// ---- one-channel ends of the network (first encoder block: Cin = 1, last decoder block: Cout = 1): L has 2 planes --------
// Against the 128 x 32 plane MFMA tile such a layer fills 2 of 32 columns (3 TFLOP/s, 10 ms of the train step for 0.2 % of
// its flops).  The work is tiny and memory-shaped (one pass over S = 64 .. 128 planes), so it runs on the vector ALU: a lane
// owns a column j, a workgroup SK_SP planes of S x a column range x all frequency rows; per (fs, j) it reads the 5 x 2 x 2
// window of L once (sliding over fs: 3 of the 5 rows carry over) and does SK_SP x 20 FMAs into register accumulators,
// reduced over lanes (DPP), waves (LDS) and column splits (the same partial-tile workspace + unpack kernel as the MFMA path).
constexpr int SK_SP = 4, SK_LP = 2, SK_TAPS = 10;

__global__ __launch_bounds__(256) void wgrad_skinny_kernel(const WgradArgs a, int cols_per_split) {
    __shared__ float red[4][SK_SP * SK_LP * SK_TAPS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, sp0 = blockIdx.y * SK_SP;
    const int jbeg = split * cols_per_split;
    int jend = jbeg + cols_per_split;
    if (jend > a.J) jend = a.J;
    float acc[SK_SP][SK_LP][SK_TAPS];
#pragma unroll
    for (int s = 0; s < SK_SP; ++s)
#pragma unroll
        for (int l = 0; l < SK_LP; ++l)
#pragma unroll
            for (int t = 0; t < SK_TAPS; ++t) acc[s][l][t] = 0.f;
    for (int j = jbeg + tid; j < jend; j += 256) {
        const int jl = j + a.dt0;                    // L columns jl (kt = 0) and jl + 1 (kt = 1)
        const bool ok0 = jl >= 0 && jl < a.J, ok1 = jl + 1 >= 0 && jl + 1 < a.J;
        // window rows fl = 2 fs + kf - 2; w[l][kf][kt]
        float w[SK_LP][5][2];
        auto load_row = [&](int l, int fl, float (&o)[2]) {
            const bool rok = (l < a.Lp) && fl >= 0 && fl < a.Fl;
            const float* r = a.L + ((size_t)(rok ? l : 0) * a.Fl + (rok ? fl : 0)) * a.JpL;
            o[0] = (rok && ok0) ? r[jl] : 0.f;
            o[1] = (rok && ok1) ? r[jl + 1] : 0.f;
        };
#pragma unroll
        for (int l = 0; l < SK_LP; ++l)
#pragma unroll
            for (int kf = 0; kf < 5; ++kf) load_row(l, kf - 2, w[l][kf]);
        for (int fs = 0; fs < a.Fs; ++fs) {
            float sv[SK_SP];
#pragma unroll
            for (int s = 0; s < SK_SP; ++s) {
                const int sp = sp0 + s;
                sv[s] = sp < a.Sp ? a.S[((size_t)sp * a.Fs + fs) * a.JpS + j] : 0.f;
            }
            // next window's two new rows in flight under this row's FMAs
            float nw[SK_LP][2][2];
#pragma unroll
            for (int l = 0; l < SK_LP; ++l) {
                load_row(l, 2 * (fs + 1) + 1, nw[l][0]);
                load_row(l, 2 * (fs + 1) + 2, nw[l][1]);
            }
#pragma unroll
            for (int s = 0; s < SK_SP; ++s)
#pragma unroll
                for (int l = 0; l < SK_LP; ++l)
#pragma unroll
                    for (int kf = 0; kf < 5; ++kf) {
                        acc[s][l][kf * 2] += sv[s] * w[l][kf][0];
                        acc[s][l][kf * 2 + 1] += sv[s] * w[l][kf][1];
                    }
#pragma unroll
            for (int l = 0; l < SK_LP; ++l) {
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) {
                    w[l][kf][0] = w[l][kf + 2][0];
                    w[l][kf][1] = w[l][kf + 2][1];
                }
                w[l][3][0] = nw[l][0][0]; w[l][3][1] = nw[l][0][1];
                w[l][4][0] = nw[l][1][0]; w[l][4][1] = nw[l][1][1];
            }
        }
    }
    // lanes -> waves -> workgroup
#pragma unroll
    for (int s = 0; s < SK_SP; ++s)
#pragma unroll
        for (int l = 0; l < SK_LP; ++l)
#pragma unroll
            for (int t = 0; t < SK_TAPS; ++t) {
                float v = acc[s][l][t];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) red[wave][(s * SK_LP + l) * SK_TAPS + t] = v;
            }
    __syncthreads();
    if (tid < SK_SP * SK_LP * SK_TAPS) {
        const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        const int t = tid % SK_TAPS, l = (tid / SK_TAPS) % SK_LP, s = tid / (SK_TAPS * SK_LP);
        a.part[(((size_t)split * SK_TAPS + t) * a.SpPad + sp0 + s) * a.LpPad + l] = v;
    }
}

struct SkinnyPlan { int groups, nsplit, cols_per_split, SpPad, LpPad; };

inline SkinnyPlan skinny_plan(int Sp, int J) {
    SkinnyPlan p;
    p.groups = (Sp + SK_SP - 1) / SK_SP;
    p.SpPad = p.groups * SK_SP;
    p.LpPad = SK_LP;
    int want = (2048 + p.groups - 1) / p.groups;
    const int maxsplit = (J + 255) / 256;
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    p.cols_per_split = ((J + want - 1) / want + 255) / 256 * 256;
    p.nsplit = (J + p.cols_per_split - 1) / p.cols_per_split;
    return p;
}

constexpr int CONV_JT = 16, PW_JT = 32;
constexpr int CONV_MS = 128, CONV_ML = 32;      // wgrad_kernel<5, 2, 1, 1, 4, 1, ...>: 4 x 1 waves of one 32 x 32 tile x 10 taps


}  // namespace

void launch_wgrad_unpack_conv(const float* part, int nsplit, int SpPad, int LpPad, int Cout, int Cx, int Cin_total, int ci_off,
                              int transposed, float* dw_re, float* dw_im, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_unpack_conv_kernel, dim3(grid_for((long long)Cout * Cx * 10)), dim3(256), 0, st, part, nsplit, SpPad,
                       LpPad, Cout, Cx, Cin_total, ci_off, transposed, dw_re, dw_im);
}

void launch_wgrad_unpack_plain(const float* part, int nsplit, int SpPad, int LpPad, int M, int K, int ldw, int rowmap, int H,
                               int accumulate, float* dw, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_unpack_plain_kernel, dim3(grid_for((long long)M * K)), dim3(256), 0, st, part, nsplit, SpPad, LpPad, M,
                       K, ldw, rowmap, H, accumulate, dw);
}

extern "C" long long idv_cconv_wgrad_work_floats(int Cs, int Cl, int B, int Tp) {
    if (Cs <= 0 || Cl <= 0 || B <= 0 || Tp <= 0) return -1;
    if (2 * Cl <= SK_LP) {
        const SkinnyPlan k = skinny_plan(2 * Cs, B * Tp);
        return (long long)k.nsplit * 10 * k.SpPad * k.LpPad;
    }
    const Plan p = make_plan_rounds(2 * Cs, 2 * Cl, B * Tp, CONV_MS, CONV_ML, CONV_JT, 1, 2, 0);      // upper bound: Fs unknown here
    return (long long)p.nsplit * 10 * p.SpPad * p.LpPad;
}

extern "C" int idv_cconv2d_bwd_weight(const float* x, int Cx, int ci_off, const float* dy, int Cout, int Cin_total,
                                      int transposed, int tshift, int Fin, int B, int Tp, int Jp_x, int Jp_dy, float* work,
                                      long long work_floats, float* dw_re, float* dw_im, void* stream) {
    if (!x || !dy || !work || !dw_re || !dw_im || Cx <= 0 || Cout <= 0 || ci_off < 0 || ci_off + Cx > Cin_total || Fin <= 0 ||
        B <= 0 || Tp <= 1)
        return IDV_EINVAL;
    if ((tshift != 0 && tshift != -1) || (Jp_x % 4) || (Jp_dy % 4) || !aligned16(x) || !aligned16(dy) || Jp_x < B * Tp ||
        Jp_dy < B * Tp)
        return IDV_EINVAL;
    const int Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    WgradArgs a{};
    if (!transposed) {      // S = dy [2Cout][Fout], L = x [2Cx][Fin]
        if (2 * Fout - 1 != Fin) return IDV_EINVAL;
        a.S = dy; a.Sp = 2 * Cout; a.Fs = Fout; a.JpS = Jp_dy;
        a.L = x;  a.Lp = 2 * Cx;   a.Fl = Fin;  a.JpL = Jp_x;
        a.dt0 = tshift;
    } else {                // S = x [2Cx][Fin], L = dy [2Cout][Fout]
        a.S = x;  a.Sp = 2 * Cx;   a.Fs = Fin;  a.JpS = Jp_x;
        a.L = dy; a.Lp = 2 * Cout; a.Fl = Fout; a.JpL = Jp_dy;
        a.dt0 = 0;
    }
    a.J = B * Tp;
    hipStream_t st = (hipStream_t)stream;
    if (a.Lp <= SK_LP) {      // one-channel end of the network: vector-ALU kernel (see wgrad_skinny_kernel)
        const SkinnyPlan k = skinny_plan(a.Sp, a.J);
        if ((long long)k.nsplit * 10 * k.SpPad * k.LpPad > work_floats) return IDV_EINVAL;
        a.part = work; a.SpPad = k.SpPad; a.LpPad = k.LpPad;
        hipLaunchKernelGGL(wgrad_skinny_kernel, dim3(k.nsplit, k.groups), dim3(256), 0, st, a, k.cols_per_split);
        hipLaunchKernelGGL(wgrad_unpack_conv_kernel, dim3(grid_for((long long)Cout * Cx * 10)), dim3(256), 0, st, work, k.nsplit,
                           k.SpPad, k.LpPad, Cout, Cx, Cin_total, ci_off, transposed, dw_re, dw_im);
        return idv_launch_status();
    }
    const Plan p = make_plan_rounds(a.Sp, a.Lp, a.J, CONV_MS, CONV_ML, CONV_JT, 1, 2, a.Fs);
    if ((long long)p.nsplit * 10 * p.SpPad * p.LpPad > work_floats) return IDV_EINVAL;
    a.part = work; a.SpPad = p.SpPad; a.LpPad = p.LpPad; a.jtiles = p.jtiles;
    a.nsplit_bal = p.nsplit; a.steps_total = (long long)p.jtiles * a.Fs;
    hipLaunchKernelGGL((wgrad_kernel<5, 2, 1, 1, 4, 1, CONV_JT, 2>), dim3(p.nsplit, p.tilesS, p.tilesL), dim3(256), 0, st, a);
    hipLaunchKernelGGL(wgrad_unpack_conv_kernel, dim3(grid_for((long long)Cout * Cx * 10)), dim3(256), 0, st, work, p.nsplit,
                       p.SpPad, p.LpPad, Cout, Cx, Cin_total, ci_off, transposed, dw_re, dw_im);
    return idv_launch_status();
}

// ---- Gauss form: entry points -----------------------------------------------------------------------------------------------
namespace {
const bool WGRAD_GAUSS = [] { const char* e = getenv("IDV_WGRAD_GAUSS"); return !e || e[0] != '0'; }();
// Winograd form of the frequency taps (wgrad_wino_kernel): IDV_WGRAD_WINO=0 keeps the ten-product kernel.  Per layer at B = 32
// (tests/tools/wgrad_layers_probe.py, ten-product -> Winograd with 32-column steps): enc2 3.67 -> 3.08 ms, enc3 3.46 -> 3.02, enc4
// 3.51 -> 3.15, dec1 6.81 -> 6.47, dec2 6.68 -> 5.73, dec3 7.24 -> 5.77, dec4 3.94 -> 3.40; with five S rows (three pairs, the last
// half empty) it does not pay (enc5 3.85 -> 3.88, dec0 7.38 -> 7.45), so layers with fewer than 8 S rows keep the ten-product
// kernel.  All layers of the DCCRN-CL step: 49.9 -> 45.3 ms.  The gain stays below the 30 % fewer MFMAs because the kernel is
// bound as much by its staging (one LDS buffer between two barriers per step) as by the matrix pipe: the first version, with all
// four S variants written to the LDS and 16-column steps, gained 4 % only.
const bool WGRAD_WINO = [] { const char* e = getenv("IDV_WGRAD_WINO"); return !e || e[0] != '0'; }();
// columns per step of the Winograd kernels: 32 (half the barriers per MFMA; the even-tap kernel then sits at 256 registers with 11
// spilled: still faster -- all layers 47.05 -> 45.31 ms at B = 32); IDV_WGRAD_WINO_JT=16: 16-column steps
inline int wgrad_wino_jt() {
    static const int v = [] { const char* e = getenv("IDV_WGRAD_WINO_JT"); return (e && atoi(e) == 16) ? 16 : 32; }();
    return v;
}
inline bool wgrad_wino_for(int Fs) {
    static const int min_rows = [] { const char* e = getenv("IDV_WGRAD_WINO_MINF"); return e ? atoi(e) : 8; }();
    return WGRAD_WINO && Fs >= min_rows;
}
struct GaussPlan { Plan p; long long prod_stride, part_floats, s_comb, l_comb; };
inline GaussPlan gauss_plan(int Cs, int Cl, int Fs, int Fl, int J, int JpS, int JpL) {
    GaussPlan g;
    // Winograd form: a step is a PAIR of S rows, 8 + 6 partial product planes per split instead of 10 tap planes
    const bool wino = wgrad_wino_for(Fs);
    g.p = make_plan_rounds(Cs, Cl, J, CONV_MS, CONV_ML, wino ? wgrad_wino_jt() : CONV_JT, 3, 2, wino ? (Fs + 1) / 2 : Fs);
    g.prod_stride = (long long)g.p.nsplit * (wino ? 14 : 10) * g.p.SpPad * g.p.LpPad;
    g.part_floats = 3 * g.prod_stride;
    g.s_comb = ((long long)Cs * Fs * JpS + 63) / 64 * 64;
    g.l_comb = ((long long)Cl * Fl * JpL + 63) / 64 * 64;
    return g;
}
}  // namespace

// 1 if idv_cconv2d_bwd_weight_gauss serves the layer (IDV_WGRAD_GAUSS=0 turns it off)
// (Cs >= 128: the S tile of the contraction kernel is 128 planes; with the real and imaginary planes no longer stacked a
//  64-channel S side would leave half of it empty, and 3 products at 50 % lose against 4 at 100 %)
extern "C" int idv_cconv_wgrad_gauss_supported(int Cs, int Cl) { return WGRAD_GAUSS && Cs >= 128 && Cl >= 32; }

// work floats of idv_cconv2d_bwd_weight_gauss: split-K partials of the three products + the combined planes (p - q | u + v)
extern "C" long long idv_cconv_wgrad_gauss_work_floats(int Cx, int Cout, int transposed, int Fin, int B, int Tp, int Jp_x, int Jp_dy) {
    if (Cx <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 0) return -1;
    const int Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    const int Cs = transposed ? Cx : Cout, Cl = transposed ? Cout : Cx;
    const int Fs = transposed ? Fin : Fout, Fl = transposed ? Fout : Fin;
    const GaussPlan g = gauss_plan(Cs, Cl, Fs, Fl, B * Tp, transposed ? Jp_x : Jp_dy, transposed ? Jp_dy : Jp_x);
    return g.part_floats + g.s_comb + g.l_comb;
}

// idv_cconv2d_bwd_weight with three real contractions per complex channel pair instead of four (same arguments, same result up
// to fp32 rounding; reference: torch.autograd of nn.Conv2d / nn.ConvTranspose2d in model/complex_progress.py:8-36, :222-279).
// work: idv_cconv_wgrad_gauss_work_floats floats, 16-byte aligned.
extern "C" int idv_cconv2d_bwd_weight_gauss(const float* x, int Cx, int ci_off, const float* dy, int Cout, int Cin_total,
                                            int transposed, int tshift, int Fin, int B, int Tp, int Jp_x, int Jp_dy, float* work,
                                            long long work_floats, float* dw_re, float* dw_im, void* stream) {
    if (!x || !dy || !work || !dw_re || !dw_im || Cx <= 0 || Cout <= 0 || ci_off < 0 || ci_off + Cx > Cin_total || Fin <= 0 ||
        B <= 0 || Tp <= 1)
        return IDV_EINVAL;
    if ((tshift != 0 && tshift != -1) || (Jp_x % 4) || (Jp_dy % 4) || !aligned16(x) || !aligned16(dy) || !aligned16(work) ||
        Jp_x < B * Tp || Jp_dy < B * Tp)
        return IDV_EINVAL;
    const int Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    if (!transposed && 2 * Fout - 1 != Fin) return IDV_EINVAL;
    WgradArgs a{};
    const float* S; const float* L;
    int Cs, Cl;
    if (!transposed) {      // S = dy [2 Cout][Fout], L = x [2 Cx][Fin]
        S = dy; Cs = Cout; a.Fs = Fout; a.JpS = Jp_dy;
        L = x;  Cl = Cx;   a.Fl = Fin;  a.JpL = Jp_x;
        a.dt0 = tshift;
    } else {                // S = x [2 Cx][Fin], L = dy [2 Cout][Fout]
        S = x;  Cs = Cx;   a.Fs = Fin;  a.JpS = Jp_x;
        L = dy; Cl = Cout; a.Fl = Fout; a.JpL = Jp_dy;
        a.dt0 = 0;
    }
    if (!idv_cconv_wgrad_gauss_supported(Cs, Cl)) return IDV_EINVAL;
    a.J = B * Tp;
    a.Sp = Cs; a.Lp = Cl;
    const GaussPlan g = gauss_plan(Cs, Cl, a.Fs, a.Fl, a.J, a.JpS, a.JpL);
    if (g.part_floats + g.s_comb + g.l_comb > work_floats) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* s_dif = work + g.part_floats;
    float* l_sum = s_dif + g.s_comb;
    const long long ns4 = (long long)Cs * a.Fs * a.JpS / 4, nl4 = (long long)Cl * a.Fl * a.JpL / 4;
    const float* S_im = S + (size_t)Cs * a.Fs * a.JpS;
    const float* L_im = L + (size_t)Cl * a.Fl * a.JpL;
    hipLaunchKernelGGL(wgrad_combine_kernel, dim3(grid_for(ns4)), dim3(256), 0, st, S, S_im, ns4, -1.0f, s_dif);
    hipLaunchKernelGGL(wgrad_combine_kernel, dim3(grid_for(nl4)), dim3(256), 0, st, L, L_im, nl4, 1.0f, l_sum);
    a.S = S;      a.L = L;        // P1 = p u
    a.S1 = S_im;  a.L1 = L_im;    // P2 = q v
    a.S2 = s_dif; a.L2 = l_sum;   // P3 = (p - q)(u + v)
    a.nprod = 3; a.tilesL = g.p.tilesL; a.prod_stride = g.prod_stride;
    a.part = work; a.SpPad = g.p.SpPad; a.LpPad = g.p.LpPad; a.jtiles = g.p.jtiles;
    a.nsplit_bal = g.p.nsplit; a.steps_total = (long long)g.p.jtiles * a.Fs;
    const dim3 grid(g.p.nsplit, g.p.tilesS, 3 * g.p.tilesL);
    if (wgrad_wino_for(a.Fs)) {
        a.steps_total = (long long)g.p.jtiles * ((a.Fs + 1) / 2);
        if (wgrad_wino_jt() == 32) {
            hipLaunchKernelGGL((wgrad_wino_kernel<0, 2, 32>), grid, dim3(256), 0, st, a);
            hipLaunchKernelGGL((wgrad_wino_kernel<1, 2, 32>), grid, dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((wgrad_wino_kernel<0, 2, 16>), grid, dim3(256), 0, st, a);
            hipLaunchKernelGGL((wgrad_wino_kernel<1, 2, 16>), grid, dim3(256), 0, st, a);
        }
        hipLaunchKernelGGL(wgrad_unpack_gauss_wino_kernel, dim3(grid_for((long long)Cout * Cx * 10)), dim3(256), 0, st, work,
                           g.prod_stride, g.p.nsplit, g.p.SpPad, g.p.LpPad, Cout, Cx, Cin_total, ci_off, transposed, dw_re, dw_im);
        return idv_launch_status();
    }
    hipLaunchKernelGGL((wgrad_kernel<5, 2, 1, 1, 4, 1, CONV_JT, 2>), grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(wgrad_unpack_gauss_kernel, dim3(grid_for((long long)Cout * Cx * 10)), dim3(256), 0, st, work, g.prod_stride,
                       g.p.nsplit, g.p.SpPad, g.p.LpPad, Cout, Cx, Cin_total, ci_off, transposed, dw_re, dw_im);
    return idv_launch_status();
}

extern "C" int idv_cconv2d_bwd_bias(const double* stats_dy, int Cout, float* db_re, float* db_im, void* stream) {
    if (!stats_dy || !db_re || !db_im || Cout <= 0) return IDV_EINVAL;
    hipLaunchKernelGGL(cconv_bias_grad_kernel, dim3((Cout + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats_dy, Cout, db_re,
                       db_im);
    return idv_launch_status();
}

extern "C" long long idv_pw_wgrad_work_floats(int M, int K, int J) {
    if (M <= 0 || K <= 0 || J <= 0) return -1;
    const Plan p = make_plan_rounds(M, K, J, 128, 128, PW_JT, 1, 1, 1);
    const Plan q = make_plan(M, K, J, 128, 128, PW_JT);       // idv_pw_bwd_weight_bf16x3 (same tile, its own split) shares this size
    return (long long)(p.nsplit > q.nsplit ? p.nsplit : q.nsplit) * p.SpPad * p.LpPad;
}

extern "C" int idv_pw_bwd_weight(const float* dout, int M, int Jp_d, const float* x, int K, int Jp_x, int J, int shift,
                                 float* work, long long work_floats, float* dw, int ldw, int rowmap, int H, int accumulate,
                                 void* stream) {
    if (!dout || !x || !work || !dw || M <= 0 || K <= 0 || J <= 0 || ldw < K || (shift != 0 && shift != -1)) return IDV_EINVAL;
    if ((Jp_d % 4) || (Jp_x % 4) || !aligned16(dout) || !aligned16(x) || Jp_d < J || Jp_x < J) return IDV_EINVAL;
    if (rowmap == 1 && (H <= 0 || (H % 16) || M % (4 * H))) return IDV_EINVAL;
    WgradArgs a{};
    a.S = dout; a.Sp = M; a.Fs = 1; a.JpS = Jp_d;
    a.L = x;    a.Lp = K; a.Fl = 1; a.JpL = Jp_x;
    a.dt0 = shift; a.J = J;
    const Plan p = make_plan_rounds(M, K, J, 128, 128, PW_JT, 1, 1, 1);
    if ((long long)p.nsplit * p.SpPad * p.LpPad > work_floats) return IDV_EINVAL;
    a.part = work; a.SpPad = p.SpPad; a.LpPad = p.LpPad; a.jtiles = p.jtiles;
    a.nsplit_bal = p.nsplit; a.steps_total = p.jtiles;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((wgrad_kernel<1, 1, 2, 2, 2, 2, PW_JT, 1>), dim3(p.nsplit, p.tilesS, p.tilesL), dim3(256), 0, st, a);
    hipLaunchKernelGGL(wgrad_unpack_plain_kernel, dim3(grid_for((long long)M * K)), dim3(256), 0, st, work, p.nsplit, p.SpPad,
                       p.LpPad, M, K, ldw, rowmap, H, accumulate, dw);
    return idv_launch_status();
}
