// cgemm_tw: the complex ConvTranspose2d contraction of cgemm_wino.hip (three real products per complex product, frequency taps in
// Winograd form) with the TIME taps in Winograd form too: F(2,2) over pairs of output columns.
//
// The decoder block's kernel has two time taps (model/complex_progress.py:222-279: kernel (5, 2), causal).  cgemm / cgemm_gauss /
// cgemm_wino apply them with ONE v_mfma_f32_32x32x2_f32 whose two k are the two taps:  out[j] = Wa x[j + s] + Wb x[j + 1 + s]
// (s = tshift).  For the column pair (2c, 2c + 1) with  a = x[2c + s], b = x[2c + 1 + s], d = x[2c + 2 + s]:
//     m1 = Wa (a - b)      m2 = (Wa + Wb) b      m3 = Wb (b - d)          out[2c] = m1 + m2       out[2c + 1] = m2 - m3
// -- three products for two columns instead of four: the two k of an MFMA become two INPUT CHANNELS, and a (32 channels x 32
// column pairs) tile costs 3 MFMAs per channel pair where the 2-tap form costs 4 (2 column tiles x 2 channels).  On top of
// cgemm_wino's 7 / 10 frequency products and the three Gauss products: 3/4 x 7/10 x 3/4 = 0.39 of the reference's real products.
// All transform factors are +-1 (and the 1/2 of the frequency taps): exact in fp32 up to the rounding of the sums.
//
// Tiles.  One accumulator tile per (frequency product r, Gauss plane g, time product tau): the even-row phase (PH 0) has 4 x 3 x 3
// = 36 tiles, the odd-row phase (PH 1) 3 x 3 x 3 = 27, for ONE tile of 32 complex output channels x 64 columns x one pair of input
// rows.  No two tiles share an operand (each has its own transformed taps and its own transformed input row), so the tiles are
// dealt to the four waves of a workgroup round-robin (tile t -> wave t % 4: 9 / 7 tiles per wave, 144 accumulator registers, two
// workgroups per CU) and the output transforms (time, Gauss, frequency) run in the epilogue on tiles exchanged through the LDS.
// Staging forms the transformed input rows once per element at the LDS write: frequency (A + cb B), Gauss (s = r + i), time
// (a - b, b, b - d).  Weights: cgemm_wino's fragments re-ordered by idv_pack_cconv_tw (lane = channel parity x 32 + co).
#include <cstdint>
#include <cstdlib>
#include "cgemm.hpp"
#include "../../include/idccrn_hip.h"

namespace {

struct TwArgs {
    const float* x0;      // planar [2][C0][Fin][Jp]
    const float* x1;      // optional skip source, planar [2][C1][Fin][Jp] (same pitch)
    int C0, C1;
    int Fin, Fout;
    int J, Jp, Tp;
    const float* wfrag;   // [phase 0: cotiles][UP][36][64] then [phase 1: cotiles][UP][28][64]
    int UP;               // channel pairs per co tile as packed (Cin rounded up to the pack granularity, / 2)
    const float* epi;     // as cgemm_gauss: [cotiles * 32][8]
    int has_fold;
    const float* slope;
    float* out;           // planar [2][Cout][Fout][Jp]
    int Cout, cotiles;
    int tshift, t_valid;
    int jtiles, ftiles;
};

constexpr int TW_PACK_CI = 8;      // pack granularity in complex input channels (cgemm_wino's WCIK)

// frequency transforms: cgemm_wino.hip's tables (transformed row r = raw row ra + cb * raw row rb of d0..d3 = input rows m0 - 1 .. m0 + 2)
template <int PH> __device__ __forceinline__ int tw_ra(int r) {
    if (PH == 0) return r == 0 ? 0 : (r == 2 ? 2 : 1);
    return r == 0 ? 1 : 2;
}
template <int PH> __device__ __forceinline__ int tw_rb(int r) {
    if (PH == 0) return r == 2 ? 1 : (r == 3 ? 3 : 2);
    return r == 0 ? 2 : 3;
}
template <int PH> __device__ __forceinline__ float tw_cb(int r) {
    if (PH == 0) return r == 1 ? 1.f : -1.f;
    return r == 1 ? 0.f : -1.f;
}
template <int PH> constexpr int tw_nr() { return PH == 0 ? 4 : 3; }
template <int PH> constexpr int tw_nt() { return tw_nr<PH>() * 9; }            // tiles (r, g, tau): t = r * 9 + g * 3 + tau
template <int PH> constexpr int tw_ntp() { return PH == 0 ? 36 : 28; }         // tile slots per channel pair in the packed weights
template <int PH> constexpr int tw_ntw() { return (tw_nt<PH>() + 3) / 4; }     // tiles per wave

template <int PH, int CIK>
__global__ __launch_bounds__(256, 2) void cconv_tw_kernel(const TwArgs a) {
    constexpr int NR = tw_nr<PH>(), NT = tw_nt<PH>(), NTP = tw_ntp<PH>(), NTW = tw_ntw<PH>();
    constexpr int KS = CIK / 2;                  // MFMA k-steps (channel pairs) per chunk
    constexpr int BT = NT * 32;                  // floats per channel in a patch buffer: NT transformed rows of 32 column pairs
    constexpr int NE = CIK * BT;
    constexpr int NBUF = 3;
    constexpr int NITEM = CIK * NR * 16;         // staging items per chunk: (channel, frequency product, 2 column pairs)
    constexpr int NLD = (NITEM + 255) / 256;
    static_assert(NLD == 1, "one staging item per thread");
    static_assert(NT * 4 * 64 <= NBUF * NE, "the epilogue exchange fits the patch buffers");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    // block order: all (frequency tile, co tile) workgroups of a 64-column block on ONE XCD (block ids equal mod 8 share an XCD):
    // they read the same raw input rows
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int per = a.cotiles * a.ftiles;
    const int jt = (slot / per) * 8 + xcd;
    const int rem = slot - (slot / per) * per;
    const int ft = rem / a.cotiles, ct = rem - ft * a.cotiles;
    if (jt >= a.jtiles) return;
    const int j0 = jt * 64;
    const int m0 = 2 * ft;
    const int rbase = m0 - 1;

    const int Cin = a.C0 + a.C1;
    const int nchunk = (Cin + CIK - 1) / CIK;

    f32x16 acc[NTW];
#pragma unroll
    for (int k = 0; k < NTW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    // this wave's tiles t = wave + 4 k.  Two kinds of tile do work nobody reads, branch-free: the 28th slot of the odd-row phase (zero
    // taps) and, in a half tile, the products r = 3, which only feed the missing second output row.
    // ---- staging: one item = channel cl, frequency product r, column pairs 2 c8, 2 c8 + 1 (output columns j0 + 4 c8 .. + 3):
    // window columns w0..w4 = input columns jc + tshift .. jc + 4 + tshift of the two raw rows, real and imaginary
    f32x4 va_r, va_i, vb_r, vb_i;
    float ea_r, ea_i, eb_r, eb_i;
    unsigned offa_v, offa_e, offb_v, offb_e, ldsoff;
    unsigned okmask = 0;      // bits 0-4: window column valid; bit 5: row A valid; bit 6: row B used and valid; bit 7: item exists
    float cbv;
    int item_cl;
    {
        const int e = tid;
        const int c8 = e & 15;
        const int r = (e >> 4) % NR, cl = e / (16 * NR);
        item_cl = cl;
        const int fa = rbase + tw_ra<PH>(r), fb = rbase + tw_rb<PH>(r);
        cbv = tw_cb<PH>(r);
        const int jc = j0 + 4 * c8;
        const int je = a.tshift ? jc - 1 : jc + 4;
        const bool exists = e < NITEM;
        const bool oka = exists && fa >= 0 && fa < a.Fin;
        const bool okb = exists && cbv != 0.f && fb >= 0 && fb < a.Fin;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = jc + i + a.tshift;
            if (c >= 0 && c < a.J) okmask |= 1u << i;
        }
        if (oka) okmask |= 1u << 5;
        if (okb) okmask |= 1u << 6;
        if (exists) okmask |= 1u << 7;
        // addresses stay inside mapped memory whatever the masks say: vector slot clamped to the row, the extra column to [0, Jp)
        const int jcv = jc + 3 < a.Jp ? jc : 0;
        const int jev = (je >= 0 && je < a.Jp) ? je : 0;
        offa_v = oka ? (unsigned)((cl * a.Fin + fa) * a.Jp + jcv) : 0u;
        offa_e = oka ? (unsigned)((cl * a.Fin + fa) * a.Jp + jev) : 0u;
        offb_v = okb ? (unsigned)((cl * a.Fin + fb) * a.Jp + jcv) : 0u;
        offb_e = okb ? (unsigned)((cl * a.Fin + fb) * a.Jp + jev) : 0u;
        if (jc + 3 >= a.Jp) okmask &= ~0x1fu;                 // (a column block past the row pitch: nothing valid)
        if (!(je >= 0 && je < a.Jp)) okmask &= a.tshift ? ~1u : ~(1u << 4);
        ldsoff = (unsigned)((cl * NT + r * 9) * 32 + 2 * c8);
    }
    auto stage_load = [&](int chunk) {
        const int ci0 = chunk * CIK;
        const bool from0 = ci0 < a.C0;
        const float* br = from0 ? a.x0 + (size_t)ci0 * a.Fin * a.Jp : a.x1 + (size_t)(ci0 - a.C0) * a.Fin * a.Jp;
        const float* bi = from0 ? br + (size_t)a.C0 * a.Fin * a.Jp : br + (size_t)a.C1 * a.Fin * a.Jp;
        const int cvalid = (from0 ? a.C0 : Cin) - ci0;
        const bool dead = item_cl >= cvalid;
        const unsigned oav = dead ? 0u : offa_v, oae = dead ? 0u : offa_e, obv = dead ? 0u : offb_v, obe = dead ? 0u : offb_e;
        va_r = *(const f32x4*)(br + oav);
        va_i = *(const f32x4*)(bi + oav);
        ea_r = br[oae];
        ea_i = bi[oae];
        vb_r = *(const f32x4*)(br + obv);
        vb_i = *(const f32x4*)(bi + obv);
        eb_r = br[obe];
        eb_i = bi[obe];
    };
    auto stage_store = [&](float* dst, int chunk) {
        const int ci0s = chunk * CIK;
        const int cvalid = (ci0s < a.C0 ? a.C0 : Cin) - ci0s;
        unsigned m = okmask;
        if (item_cl >= cvalid) m &= ~0x7fu;
        const bool ra_ok = (m >> 5) & 1u, rb_ok = (m >> 6) & 1u;
        const bool left = a.tshift != 0;
        float fr[5], fi[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int vi_ = left ? i - 1 : i;                 // index into the vector slot; the extra column is w0 (left) or w4
            const bool ext = left ? i == 0 : i == 4;
            const float ar = ext ? ea_r : va_r[vi_ & 3], ai = ext ? ea_i : va_i[vi_ & 3];
            const float br_ = ext ? eb_r : vb_r[vi_ & 3], bi_ = ext ? eb_i : vb_i[vi_ & 3];
            const bool cok = (m >> i) & 1u;
            const float xa_r = (cok && ra_ok) ? ar : 0.f, xa_i = (cok && ra_ok) ? ai : 0.f;
            const float xb_r = (cok && rb_ok) ? br_ : 0.f, xb_i = (cok && rb_ok) ? bi_ : 0.f;
            fr[i] = xa_r + cbv * xb_r;
            fi[i] = xa_i + cbv * xb_i;
        }
        if ((m >> 7) & 1u) {
            float* d = dst + ldsoff;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float x[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) x[i] = g == 0 ? fr[i] + fi[i] : (g == 1 ? fr[i] : fi[i]);
                // pair 0: (a, b, d) = x0, x1, x2; pair 1: x2, x3, x4
                *(float2*)(d + (g * 3 + 0) * 32) = make_float2(x[0] - x[1], x[2] - x[3]);
                *(float2*)(d + (g * 3 + 1) * 32) = make_float2(x[1], x[3]);
                *(float2*)(d + (g * 3 + 2) * 32) = make_float2(x[1] - x[2], x[3] - x[4]);
            }
        }
    };

    // ---- weights: one 4-byte load per tile, channel pair and lane (lane = channel parity x 32 + co), in a ring of KS channel pairs
    const float* wbase = a.wfrag + ((size_t)(PH == 1 ? (size_t)a.cotiles * a.UP * tw_ntp<0>() : 0) + (size_t)ct * a.UP * NTP) * 64 +
                         (size_t)wave * 64 + lane;
    const int total_ks = nchunk * KS;
    float a_w[KS][NTW];
    auto load_w = [&](int g, float (&dst)[NTW]) {
        g = g < total_ks ? g : total_ks - 1;                  // (past the end of K: an unused re-fetch)
        const float* ws = wbase + (size_t)g * NTP * 64;
#pragma unroll
        for (int k = 0; k < NTW; ++k) dst[k] = ws[(wave + 4 * k < NTP ? 4 * k : 0) * 64];
    };
    auto load_b = [&](const float* P, int ul, float (&dst)[NTW]) {
        const float* row = P + (size_t)((2 * ul + half) * NT + wave) * 32 + l31;
#pragma unroll
        for (int k = 0; k < NTW; ++k) dst[k] = row[(wave + 4 * k < NT ? 4 * k : 0) * 32];
    };

    stage_load(0);
#pragma unroll
    for (int u = 0; u < KS; ++u) load_w(u, a_w[u]);
    stage_store(smem, 0);
    stage_load(nchunk > 1 ? 1 : 0);
    __syncthreads();

    float b_cur[NTW], b_nxt[NTW];
    load_b(smem, 0, b_cur);
    int ibuf = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const float* P = smem + ibuf * NE;
        const int i1 = ibuf + 1 == NBUF ? 0 : ibuf + 1;
        float* Pn = smem + i1 * NE;
#pragma unroll
        for (int ul = 0; ul < KS; ++ul) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[ul][0], b_cur[0], acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ul == KS - 1) __syncthreads();                // every wave has stored chunk + 1 (at ul == 0): Pn is complete
            if (ul + 1 < KS)
                load_b(P, ul + 1, b_nxt);
            else
                load_b(Pn, 0, b_nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (ul == 0) {
                // the registers hold chunk + 1 (loaded one chunk ago); its buffer was last read two chunks ago, a barrier since
                stage_store(Pn, chunk + 1 < nchunk ? chunk + 1 : chunk);
                stage_load(chunk + 2 < nchunk ? chunk + 2 : nchunk - 1);
            }
#pragma unroll
            for (int k = 1; k < NTW; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[ul][k], b_cur[k], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w((chunk + 1) * KS + ul, a_w[ul]);
#pragma unroll
            for (int k = 0; k < NTW; ++k) b_cur[k] = b_nxt[k];
        }
        ibuf = i1;
    }
    __syncthreads();                                          // all patch reads done: the buffers become the exchange area

    // ------------------------------------------------------------------ epilogue
    // four slices of four accumulator registers: every wave writes its tiles' slice, then thread (wave w, lane l) owns register
    // 4 s + w of lane l -- output channel co, column pair l31 -- reads ALL tiles there and runs the output transforms
    const float slope = a.slope ? *a.slope : 1.0f;
    const bool has_act = a.slope != nullptr;
    float* E = smem;
    const int jA = j0 + 2 * l31;                              // the pair's two output columns jA, jA + 1
    bool keep[2], inb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = jA + q;
        const int tp = j % a.Tp;
        inb[q] = j < a.J;
        keep[q] = inb[q] && tp >= 1 && tp <= a.t_valid;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s > 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < NTW; ++k) {
            const int t = wave + 4 * k;
            if (t < NT) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) E[(t * 4 + rr) * 64 + lane] = acc[k][4 * s + rr];
            }
        }
        __syncthreads();
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = E[(t * 4 + wave) * 64 + lane];
        // time, then Gauss: P[r][q][re / im]
        float pr[NR][2], pi[NR][2];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float y[3][2];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float m1 = v[r * 9 + g * 3], m2 = v[r * 9 + g * 3 + 1], m3 = v[r * 9 + g * 3 + 2];
                y[g][0] = m1 + m2;
                y[g][1] = m2 - m3;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pr[r][q] = y[0][q] - y[2][q];
                pi[r][q] = y[0][q] + y[1][q];
            }
        }
        const int rg = 4 * s + wave;
        const int co = ct * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
        const bool cok = co < a.Cout;
        const f32x4 e0 = *(const f32x4*)(a.epi + (size_t)co * 8);
        const float e4 = a.epi[(size_t)co * 8 + 4], e5 = a.epi[(size_t)co * 8 + 5];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int fo = 2 * m0 + PH + 2 * rt;
            if (fo >= a.Fout) continue;
            float yr[2], yi[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float re, im;
                if (PH == 0) {
                    re = rt == 0 ? pr[0][q] + pr[1][q] + pr[2][q] : pr[1][q] - pr[2][q] - pr[NR - 1][q];
                    im = rt == 0 ? pi[0][q] + pi[1][q] + pi[2][q] : pi[1][q] - pi[2][q] - pi[NR - 1][q];
                } else {
                    re = rt == 0 ? pr[0][q] + pr[1][q] : pr[1][q] - pr[2][q];
                    im = rt == 0 ? pi[0][q] + pi[1][q] : pi[1][q] - pi[2][q];
                }
                float r_, i_;
                if (a.has_fold) {
                    r_ = e0[0] * re + e0[1] * im + e4;
                    i_ = e0[2] * re + e0[3] * im + e5;
                } else {
                    r_ = re + e4;
                    i_ = im + e5;
                }
                if (has_act) {
                    r_ = r_ >= 0.f ? r_ : slope * r_;
                    i_ = i_ >= 0.f ? i_ : slope * i_;
                }
                yr[q] = keep[q] ? r_ : 0.f;
                yi[q] = keep[q] ? i_ : 0.f;
            }
            if (cok) {
                float* o_r = a.out + ((size_t)co * a.Fout + fo) * a.Jp + jA;
                float* o_i = a.out + ((size_t)(a.Cout + co) * a.Fout + fo) * a.Jp + jA;
                if (inb[1]) {
                    *(float2*)o_r = make_float2(yr[0], yr[1]);
                    *(float2*)o_i = make_float2(yi[0], yi[1]);
                } else if (inb[0]) {
                    o_r[0] = yr[0];
                    o_i[0] = yi[0];
                }
            }
        }
    }
}

// cgemm_wino's fragments [phase][ct][unit = ci * 3 + g][slot r (4)][lane = h * 32 + co] (h: the two time taps as the MFMA's two k,
// h = 0 multiplies column j + tshift) -> [phase][ct][pair u][tile t = r * 9 + g * 3 + tau (36 | 28 slots)][lane = parity * 32 + co]
// with the time-transformed taps  tau 0: W_h0,  tau 1: W_h0 + W_h1,  tau 2: W_h1.
__global__ void pack_cconv_tw_kernel(const float* __restrict__ wino, int cotiles, int UN, int UP, float* __restrict__ out) {
    const long long n0 = (long long)cotiles * UP * tw_ntp<0>() * 64, n1 = (long long)cotiles * UP * tw_ntp<1>() * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n0 + n1; idx += (long long)gridDim.x * blockDim.x) {
        const int ph = idx >= n0;
        const long long i = ph ? idx - n0 : idx;
        const int ntp = ph ? tw_ntp<1>() : tw_ntp<0>(), nt = ph ? tw_nt<1>() : tw_nt<0>();
        const int lane = (int)(i & 63);
        long long t_ = i >> 6;
        const int t = (int)(t_ % ntp); t_ /= ntp;
        const int u = (int)(t_ % UP);
        const int ct = (int)(t_ / UP);
        float val = 0.f;
        const int ci = 2 * u + (lane >> 5), co = lane & 31;
        if (t < nt && ci * 3 < UN) {
            const int r = t / 9, g = (t % 9) / 3, tau = t % 3;
            const float* src = wino + ((((size_t)ph * cotiles + ct) * UN + (size_t)ci * 3 + g) * 4 + r) * 64;
            const float w0 = src[co], w1 = src[32 + co];
            val = tau == 0 ? w0 : (tau == 1 ? w0 + w1 : w1);
        }
        out[idx] = val;
    }
}

template <int PH, int CIK>
int launch_tw_ph(const TwArgs& a, hipStream_t st) {
    constexpr int NE = CIK * tw_nt<PH>() * 32;
    constexpr size_t smem = 3 * NE * sizeof(float);
    static_assert(smem * 2 <= 160 * 1024, "the patch buffers of two workgroups must fit the 160 KB of LDS");
    TwArgs b = a;
    b.jtiles = (a.J + 63) / 64;
    b.ftiles = PH == 1 ? a.Fin / 2 : (a.Fin + 1) / 2;
    if (b.ftiles == 0) return IDV_OK;
    const long long nblk = (long long)((b.jtiles + 7) / 8) * 8 * b.ftiles * b.cotiles;
    if (nblk > 0x7fffffffLL) return IDV_EINVAL;
    auto k = cconv_tw_kernel<PH, CIK>;
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), smem, st, b);
    return idv_launch_status();
}

}  // namespace

// 1 if idv_ctconv2d_tw_fwd serves the layer: what cgemm_wino's transposed form serves, with C0 a multiple of 4 when there is a
// second source (a K chunk of 4 channels never straddles the sources)
extern "C" int idv_cconv_tw_supported(int C0, int C1, int Cout, int Fin) {
    if (!idv_cconv_wino_supported(1, C0, C1, Cout, Fin)) return 0;
    return (C1 == 0 || C0 % 4 == 0) ? 1 : 0;
}

extern "C" long long idv_cconv_tw_wfrag_floats(int Cout, int cin_used) {
    const long long cotiles = (Cout + 31) / 32, cpad = (cin_used + TW_PACK_CI - 1) / TW_PACK_CI * TW_PACK_CI;
    return cotiles * (cpad / 2) * (36 + 28) * 64;
}

// wino_frag: idv_pack_cconv_wino(transposed = 1) of the same weights; tw_frag: idv_cconv_tw_wfrag_floats floats
extern "C" int idv_pack_cconv_tw(const float* wino_frag, int Cout, int cin_used, float* tw_frag, void* stream) {
    if (!wino_frag || !tw_frag || Cout <= 0 || cin_used <= 0) return IDV_EINVAL;
    const int cotiles = (Cout + 31) / 32;
    const int cpad = (cin_used + TW_PACK_CI - 1) / TW_PACK_CI * TW_PACK_CI;
    const long long n = idv_cconv_tw_wfrag_floats(Cout, cin_used);
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_cconv_tw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, wino_frag, cotiles, cpad * 3, cpad / 2,
                       tw_frag);
    return idv_launch_status();
}

// idv_cconv2d_wino_fwd (transposed = 1, no statistics, no addend) on the time-Winograd kernels: same result up to the rounding of
// the transforms.  wfrag from idv_pack_cconv_tw, epi / has_fold from idv_pack_cconv_gauss.  Requires 16-byte aligned sources, Jp % 4
// == 0 and, with a second source, the same pitch.  Reference: model/complex_progress.py:222-279 (+ :161-209 and pvae_module.py:82
// for the epilogue).
extern "C" int idv_ctconv2d_tw_fwd(const float* x0, int C0, const float* x1, int C1, const float* wfrag, const float* epi, int has_fold,
                                   const float* prelu_slope, float* out, int tshift, int Cout, int Fin, int B, int Tp, int Jp,
                                   int t_valid_out, void* stream) {
    if (!x0 || !wfrag || !epi || !out || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (C1 > 0 && !x1) return IDV_EINVAL;
    if (tshift != 0 && tshift != -1) return IDV_EINVAL;
    if (!idv_cconv_tw_supported(C0, C1, Cout, Fin)) return IDV_EINVAL;
    if ((Jp & 3) || (reinterpret_cast<uintptr_t>(x0) & 15) || (C1 > 0 && (reinterpret_cast<uintptr_t>(x1) & 15)) ||
        (reinterpret_cast<uintptr_t>(out) & 7))
        return IDV_EINVAL;
    TwArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.Fin = Fin; a.Fout = 2 * Fin - 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp;
    a.wfrag = wfrag; a.UP = (C0 + C1 + TW_PACK_CI - 1) / TW_PACK_CI * TW_PACK_CI / 2;
    a.epi = epi; a.has_fold = has_fold; a.slope = prelu_slope; a.out = out;
    a.Cout = Cout; a.cotiles = (Cout + 31) / 32;
    a.tshift = tshift; a.t_valid = t_valid_out;
    if (Jp < a.J) return IDV_EINVAL;
    if ((long long)4 * Fin * (long long)Jp >= 0xffffffffLL) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = launch_tw_ph<0, 4>(a, st)) return rc;
    return launch_tw_ph<1, 4>(a, st);
}
