// cgemm_wino: the complex ConvTranspose2d contraction of cgemm_gauss.hip (three real products per complex product) with the
// FREQUENCY taps in Winograd minimal-filtering form: 7 instead of 10 real MFMA products per (input channel, pair of input rows).
//
// The reference's decoder block is a (5, 2)-tap transposed convolution with stride (2, 1) (model/complex_progress.py:222-279,
// model/pvae_module.py:72-93).  Per input row m it feeds two output rows,
//     out[2m]     = W4 x[m-1] + W2 x[m] + W0 x[m+1]          (even taps: a 3-tap stride-1 correlation over the input rows)
//     out[2m + 1] =             W3 x[m] + W1 x[m+1]          (odd taps:  a 2-tap one)
// (W_kf = the real 1 x 2 time-tap pair of frequency tap kf, applied by ONE v_mfma_f32_32x32x2_f32 whose two k are the two time
// taps, exactly as in cgemm.hpp).  cgemm_gauss.hip spends 5 MFMAs per input row = 10 per pair of rows.  For the pair (m, m+1)
// with d0..d3 = x[m-1..m+2]:
//     F(2,3)   M1 = (d0 - d2) W4            M2 = (d1 + d2) (W4 + W2 + W0)/2     M3 = (d2 - d1) (W4 - W2 + W0)/2    M4 = (d1 - d3) W0
//              out[2m] = M1 + M2 + M3       out[2m + 2] = M2 - M3 - M4
//     F(2,2)   N1 = (d2 - d1) (-W3)         N2 = d2 (W3 + W1)                   N3 = (d2 - d3) W1
//              out[2m + 1] = N1 + N2        out[2m + 3] = N2 - N3
// -- seven products on SIX transformed rows (d2 - d1 serves M3 and N1).  The transformed rows are formed once per staged element
// at the LDS write (where cgemm_gauss already forms s = x_r + x_i), the transformed taps once per parameter update by
// idv_pack_ctconv_wino, the output transform is register-local in the epilogue.  All factors are 1 or 1/2: the transforms are
// exact in fp32 up to the rounding of the sums (measured: DESIGN.md 3.1d).
//
// The encoder's stride-(2, 1) convolution (model/complex_progress.py:8-36) is the mirror image: with r0..r6 = x[2 fo - 2 .. 2 fo + 4]
// the even taps (W0, W2, W4) form a 3-tap correlation over the even-offset rows (r0, r2, r4, r6), the odd taps (W1, W3) a 2-tap
// one over (r1, r3, r5), and BOTH feed the same two outputs out[fo], out[fo + 1] -- so the seven products share FOUR accumulators
//     A0 = (r0 - r4) W0 + (r1 - r3) W1      A1 = (r2 + r4) (W0 + W2 + W4)/2 + r3 (W1 + W3)
//     A2 = (r4 - r2) (W0 - W2 + W4)/2       A3 = (r2 - r6) W4 + (r3 - r5) W3
//     out[fo] = A0 + A1 + A2                out[fo + 1] = A1 - A2 - A3
// (template PH = 2: 7 products on 7 transformed rows into 4 x 3 = 12 accumulator tiles per wave).
//
// Two kinds of workgroup tiles for the transposed conv (template PH = 0 / 1), because an MFMA's accumulators live in the 256 AGPRs = at most 16 tiles per wave
// (with more, hipcc swaps the rest through the AGPRs around every MFMA): PH = 0 computes the EVEN output rows of a pair of input
// rows (F(2,3): 4 products x 3 Gauss planes = 12 accumulator tiles per wave, 4 transformed patch rows), PH = 1 the ODD rows
// (F(2,2): 3 x 3 = 9 tiles, 3 transformed rows).  Per wave: 32 complex output channels x 2 output rows x ONE 32-column tile.
// Weights: 4 floats per (channel, plane) and lane as ONE 16-byte load -- one vector-memory instruction per 4 / 3 MFMAs
// (cgemm_gauss: five per ten).
#include <cstdlib>
#include "cgemm.hpp"
#include "../../include/idccrn_hip.h"

namespace {

struct WinoArgs {
    const float* x0;      // planar [2][C0][Fin][Jp]
    const float* x1;      // optional skip source, planar [2][C1][Fin][Jp] (same pitch, x1_div == 1)
    int C0, C1;
    int Fin, Fout;
    int J, Jp, Tp;
    const float* wfrag;   // transposed: [phase 2][cotiles][units = Cin_pad * 3][4][64] (the phase's 4 / 3 + pad transformed taps of unit
                          // (ci, p)); conv: [cotiles][units][8][64] (7 taps + pad)
    int UN;               // units per co tile as packed (Cin rounded up to the pack granularity, x 3)
    const float* epi;     // as cgemm_gauss: [cotiles * 32][8]
    int has_fold;
    const float* slope;
    float* out;           // planar [2][Cout][Fout][Jp]
    int Cout, cotiles;
    int tshift, t_valid;
    double* stats;        // train mode: [Cout][5] sums (r, i, rr, ii, ri) of conv + bias, or nullptr (as cgemm_gauss)
    int stats_rep;        // > 1: that many replicas [rep][Cout][5] (power of two), one chosen per workgroup
    const float* add;     // optional addend (see cgemm_gauss.hip)
    int add_div, add_Jp;
    int jtiles, ftiles, mblocks;
    int xcd_split;        // block order: co-tile blocks on different XCDs (see the kernel)
};

constexpr int WCIK = 8;          // pack granularity in complex input channels (= cgemm_gauss's: shared `supported` rule); the kernel's K
                                 // chunk CIK divides it (2: the weight ring of a chunk is 48 registers beside 336 accumulator registers)
// transformed row tq = A + cb B of the raw patch rows (transposed conv: d0..d3 = input rows m0 - 1 .. m0 + 2; conv: r0..r6 = input
// rows 2 fo0 - 2 .. 2 fo0 + 4); product q pairs tap q with row q and adds into accumulator wino_acc(q)
//   PH 0 (transposed, even rows):  d0 - d2,  d1 + d2,  d2 - d1,  d1 - d3     taps  W4, (W4 + W2 + W0)/2, (W4 - W2 + W0)/2, W0
//   PH 1 (transposed, odd rows):   d1 - d2,  d2,       d2 - d3               taps  W3, W3 + W1, W1
//   PH 2 (conv):  r0 - r4,  r2 + r4,  r4 - r2,  r2 - r6,  r1 - r3,  r3,  r3 - r5
//                 taps  W0, (W0 + W2 + W4)/2, (W0 - W2 + W4)/2, W4, W1, W1 + W3, W3      accumulators 0, 1, 2, 3, 0, 1, 3
template <int PH> __device__ __forceinline__ int wino_ra(int tq) {
    if (PH == 0) return tq == 0 ? 0 : (tq == 2 ? 2 : 1);
    if (PH == 1) return tq == 0 ? 1 : 2;
    return tq == 0 ? 0 : (tq == 1 ? 2 : (tq == 2 ? 4 : (tq == 3 ? 2 : (tq == 4 ? 1 : 3))));
}
template <int PH> __device__ __forceinline__ int wino_rb(int tq) {
    if (PH == 0) return tq == 2 ? 1 : (tq == 3 ? 3 : 2);
    if (PH == 1) return tq == 0 ? 2 : 3;
    return tq == 0 ? 4 : (tq == 1 ? 4 : (tq == 2 ? 2 : (tq == 3 ? 6 : (tq == 4 ? 3 : 5))));       // tq == 5: unused
}
template <int PH> __device__ __forceinline__ float wino_cb(int tq) {
    if (PH == 0) return tq == 1 ? 1.f : -1.f;
    if (PH == 1) return tq == 1 ? 0.f : -1.f;
    return tq == 1 ? 1.f : (tq == 5 ? 0.f : -1.f);
}
template <int PH> constexpr int wino_tr() { return PH == 0 ? 4 : (PH == 1 ? 3 : 7); }
template <int PH> constexpr int wino_nacc() { return PH == 1 ? 3 : 4; }
template <int PH> constexpr int wino_slots() { return PH == 2 ? 8 : 4; }          // weight slots per unit in the packed buffer
constexpr int wino_acc2(int q) { return q < 4 ? q : (q == 4 ? 0 : (q == 5 ? 1 : 3)); }

// OCC: workgroups per CU the kernel is built for (2: at most 256 registers); RD: depth of the weight ring in units (0: a whole chunk)
template <int PH, int WM, int WN, int CIK, int NBUF, bool STATS, int OCC = 1, int RDP = 0>
__global__ __launch_bounds__(WM* WN * 64, OCC) void cconv_wino_kernel(const WinoArgs a) {
    constexpr int NT = WM * WN * 64;
    constexpr int TR = wino_tr<PH>();             // transformed patch rows per (channel, plane) = products per (channel, plane)
    constexpr int NP = TR, NACC = wino_nacc<PH>(), SL = wino_slots<PH>();
    constexpr int JT = 32 * WN;
    constexpr int PS = JT + 8;                    // patch row: the 16-byte aligned span j0-4 .. j0+JT+3
    constexpr int COL0 = 4;
    constexpr int PS4 = PS / 4;
    constexpr int NS = CIK * TR * PS4;            // staging items per chunk: one float4 slot of one TRANSFORMED row of one channel
    constexpr int NLD = (NS + NT - 1) / NT;
    constexpr int NE = CIK * 3 * TR * PS;         // patch floats per chunk
    constexpr int UNITS = CIK * 3;                // pipeline units per chunk: (channel, plane)
    static_assert(NLD <= 8, "staging registers");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // block order as cgemm_gauss (map_ft form): all frequency tiles of a column block on ONE XCD, the MB co-tile blocks of a tile
    // on consecutive slots of it
    const int MB = a.mblocks, FTn = a.ftiles;
    const int bid = blockIdx.x;
    int jt, ft, mblk;
    if (a.xcd_split) {
        // the MB co-tile blocks go to DIFFERENT XCDs (block ids equal mod 8 share an XCD): an XCD then streams 1 / MB of the
        // layer's weights, which its L2 holds, instead of all of them (MB = 2, 4 or 8; a column block is read by MB XCDs)
        const int xcd = bid & 7, slot = bid >> 3;
        const int G = 8 / MB;                           // XCD groups per co-tile block
        mblk = xcd % MB;
        jt = (slot / FTn) * G + xcd / MB;
        ft = slot - (slot / FTn) * FTn;
    } else {
        const int per = 8 * MB * FTn;
        const int sg = bid / per, rem = bid - sg * per;
        const int v = rem >> 3;
        jt = sg * 8 + (rem & 7);
        ft = v / MB;
        mblk = v - (v / MB) * MB;
    }
    if (jt >= a.jtiles) return;
    const int j0 = jt * JT;
    const int ct = mblk * WM + wm;
    const bool ct_ok = ct < a.cotiles;
    const int m0 = 2 * ft;                        // transposed: first input row of the tile (raw patch rows m0 - 1 .. m0 + 2);
    const int rbase = PH == 2 ? 2 * m0 - 2 : m0 - 1;      // conv: first OUTPUT row (raw patch rows 2 m0 - 2 .. 2 m0 + 4)
    // last tile of an odd row count: its SECOND output row does not exist, the products that only feed it are skipped (PH 0: M4;
    // PH 2: the two into A3) -- a wave-uniform branch around 1 of 4 / 2 of 7 MFMAs per unit
    const bool half_tile = PH != 1 && m0 + 1 >= (PH == 2 ? a.Fout : a.Fin);

    const int Cin = a.C0 + a.C1;
    const int nchunk = (Cin + CIK - 1) / CIK;

    f32x16 acc[NACC][3];
#pragma unroll
    for (int w = 0; w < NACC; ++w)
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[w][p3][r] = 0.f;

    // ---- staging.  An item = 4 consecutive columns of one transformed row of one channel: it loads the real and imaginary
    // values of its two raw rows (A, B), forms T = A + cb B for the planes x_r and x_i and s = T_r + T_i, writes three float4.
    f32x4 sar[NLD], sai[NLD], sbr[NLD], sbi[NLD];
    unsigned voffa[NLD], voffb[NLD];  // float offsets of the two raw slots relative to the chunk's first real plane
    unsigned long long okbits = 0;    // 4 column bits (a) | 4 column bits (b) per item
    unsigned ldsoff[NLD];
    float cbv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * NT;
        const int row = e / PS4, c4 = e - row * PS4;
        const int cil = row / TR, tq = row - cil * TR;
        const int fa = rbase + wino_ra<PH>(tq), fb = rbase + wino_rb<PH>(tq);
        const int jv = j0 - 4 + 4 * c4;
        const bool oka = (e < NS) && fa >= 0 && fa < a.Fin;
        const bool okb = (e < NS) && wino_cb<PH>(tq) != 0.f && fb >= 0 && fb < a.Fin;
        unsigned ba = 0, bb = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool cok = jv + q >= 0 && jv + q < a.J;
            if (oka && cok) ba |= 1u << q;
            if (okb && cok) bb |= 1u << q;
        }
        okbits |= (unsigned long long)(ba | (bb << 4)) << (8 * i);
        voffa[i] = ba ? (unsigned)((cil * a.Fin + fa) * a.Jp + jv) : 0u;      // a dead slot loads mapped memory (offset 0)
        voffb[i] = bb ? (unsigned)((cil * a.Fin + fb) * a.Jp + jv) : 0u;
        cbv[i] = wino_cb<PH>(tq);
        ldsoff[i] = (unsigned)(((cil * 3) * TR + tq) * PS + 4 * c4);
    }
    auto stage_load = [&](int chunk) {
        const int ci0 = chunk * CIK;
        const bool from0 = ci0 < a.C0;
        const float* br = from0 ? a.x0 + (size_t)ci0 * a.Fin * a.Jp : a.x1 + (size_t)(ci0 - a.C0) * a.Fin * a.Jp;
        const float* bi = from0 ? br + (size_t)a.C0 * a.Fin * a.Jp : br + (size_t)a.C1 * a.Fin * a.Jp;
        const int cvalid = (from0 ? a.C0 : Cin) - ci0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool dead = (tid + i * NT) / (TR * PS4) >= cvalid;
            const unsigned oa = dead ? 0u : voffa[i], ob = dead ? 0u : voffb[i];
            sar[i] = *(const f32x4*)(br + oa);
            sai[i] = *(const f32x4*)(bi + oa);
            sbr[i] = *(const f32x4*)(br + ob);
            sbi[i] = *(const f32x4*)(bi + ob);
        }
    };
    auto stage_store = [&](float* dst, int chunk) {
        const int ci0s = chunk * CIK;
        const int cvalid = (ci0s < a.C0 ? a.C0 : Cin) - ci0s;
#pragma unroll
        for (int i = 0; i < NLD; ++i) asm volatile("" ::"v"(sar[i]), "v"(sai[i]), "v"(sbr[i]), "v"(sbi[i]));
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            unsigned bits = (unsigned)(okbits >> (8 * i)) & 255u;
            if (e / (TR * PS4) >= cvalid) bits = 0u;
            f32x4 vr, vi, vs;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float ar = (bits >> q) & 1u ? sar[i][q] : 0.f, ai = (bits >> q) & 1u ? sai[i][q] : 0.f;
                const float br_ = (bits >> (4 + q)) & 1u ? sbr[i][q] : 0.f, bi_ = (bits >> (4 + q)) & 1u ? sbi[i][q] : 0.f;
                vr[q] = ar + cbv[i] * br_;
                vi[q] = ai + cbv[i] * bi_;
                vs[q] = vr[q] + vi[q];
            }
            if (e < NS) {
                float* d = dst + ldsoff[i];
                *(f32x4*)d = vs;
                *(f32x4*)(d + TR * PS) = vr;
                *(f32x4*)(d + 2 * TR * PS) = vi;
            }
        }
    };

    // ---- weights: per unit NP 4-byte loads per lane (one coalesced 256-byte load per tap: measured 5 % faster than one 16-byte
    // load per lane), kept in a ring of RD units: the fragments of unit g are re-loaded with those of unit g + RD right after the
    // NEXT unit's first MFMA (rotated by one unit so that nothing loaded right before the loop's back edge is live across it, as
    // cgemm_gauss).  RD = a whole chunk with one workgroup per CU; with two per CU the other workgroup covers the latency.
    constexpr int RD = RDP ? RDP : UNITS;
    static_assert(UNITS % RD == 0, "ring slots are compile-time");
    const float* wbase = a.wfrag + (((size_t)(PH == 1 ? a.cotiles : 0) + (ct_ok ? ct : 0)) * a.UN) * SL * 64 + lane;
    const int total_units = nchunk * UNITS;
    float a_w[RD][NP];

    const int bcol = wn * 32 + (lane & 31) + (lane >> 5) + COL0 + a.tshift;
    auto load_b = [&](const float* P, int u, float (&dst)[TR]) {
#pragma unroll
        for (int tq = 0; tq < TR; ++tq) dst[tq] = P[(u * TR + tq) * PS + bcol];
    };

    stage_load(0);
#pragma unroll
    for (int u = 0; u < RD; ++u)
#pragma unroll
        for (int q = 0; q < NP; ++q) a_w[u][q] = wbase[(size_t)(u * SL + q) * 64];
    stage_store(smem, 0);
#pragma unroll
    for (int u = 0; u < RD; ++u)
#pragma unroll
        for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(a_w[u][q]));
    __syncthreads();

    constexpr int UMID = UNITS / 2;
    float b_cur[TR], b_nxt[TR];
    int ibuf = 0;
    if (NBUF == 3) load_b(smem, 0, b_cur);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const float* P = smem + ibuf * NE;
        const int inext = (ibuf + 1 == NBUF) ? 0 : ibuf + 1;
        float* Pn = smem + inext * NE;
        const int nxt = (chunk + 1 < nchunk) ? chunk + 1 : chunk;
        if (NBUF == 2) load_b(P, 0, b_cur);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int p3 = u % 3;
            // product q = tap q x transformed row q -> accumulator q (conv: 0, 1, 2, 3, 0, 1, 3)
            acc[0][p3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[u % RD][0], b_cur[0], acc[0][p3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (NBUF == 3 && u == UNITS - 1) __syncthreads();
            if (u + 1 < UNITS)
                load_b(P, u + 1, b_nxt);
            else if (NBUF == 3)
                load_b(Pn, 0, b_nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (u == 0) stage_load(nxt);
            if (u == (NBUF == 3 ? UMID : UNITS - 1)) stage_store(Pn, nxt);
#pragma unroll
            for (int q = 1; q < NP; ++q) {
                constexpr int dummy = 0;
                (void)dummy;
                const int ai = PH == 2 ? wino_acc2(q) : q;
                if (ai == 3 && half_tile) continue;       // feeds the missing second output row only
                acc[ai][p3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[u % RD][q], b_cur[q], acc[ai][p3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // the PREVIOUS unit's slot is free: fetch unit (g - 1) + RD into it (clamped at the end of K: unused re-fetch)
                int gl = chunk * UNITS + u - 1 + RD;
                gl = gl < total_units ? gl : total_units - 1;
                const float* ws = wbase + (size_t)gl * SL * 64;
#pragma unroll
                for (int q = 0; q < NP; ++q) a_w[(u + RD - 1) % RD][q] = ws[q * 64];
            }
            if (u + 1 < UNITS || NBUF == 3) {
#pragma unroll
                for (int tq = 0; tq < TR; ++tq) b_cur[tq] = b_nxt[tq];
            }
        }
        if (NBUF == 2) __syncthreads();
        ibuf = inext;
    }

    // ------------------------------------------------------------------ epilogue
    if (!ct_ok) return;
    const float slope = a.slope ? *a.slope : 1.0f;
    const bool has_act = a.slope != nullptr;
    const int half = lane >> 5, l31 = lane & 31;
    const int j = j0 + wn * 32 + l31;
    const int bj = j / a.Tp, tp = j - bj * a.Tp;
    const bool keep = (tp >= 1) && (tp <= a.t_valid);
    const bool inb = j < a.J;
    const int ja = j - (bj - bj / a.add_div) * a.Tp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const f32x4 e0 = *(const f32x4*)(a.epi + (size_t)co * 8);
        const float e4 = a.epi[(size_t)co * 8 + 4], e5 = a.epi[(size_t)co * 8 + 5];
        const bool cok = co < a.Cout;
        float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        float t[2][3];                       // this phase's two output rows x Gauss products
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
            if (PH != 1) {
                const float M1 = acc[0][p3][r], M2 = acc[1][p3][r], M3 = acc[2][p3][r], M4 = acc[NACC - 1][p3][r];
                t[0][p3] = M1 + M2 + M3;
                t[1][p3] = M2 - M3 - M4;
            } else {
                const float N1 = acc[0][p3][r], N2 = acc[1][p3][r], N3 = acc[2][p3][r];
                t[0][p3] = N1 + N2;
                t[1][p3] = N2 - N3;
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int fo = PH == 2 ? m0 + rt : 2 * m0 + PH + 2 * rt;
            if (fo >= a.Fout) continue;
            float re = t[rt][0] - t[rt][2], im = t[rt][0] + t[rt][1];
            if (a.add && cok && inb) {
                re += a.add[((size_t)co * a.Fout + fo) * a.add_Jp + ja];
                im += a.add[((size_t)(a.Cout + co) * a.Fout + fo) * a.add_Jp + ja];
            }
            float yr, yi;
            if (a.has_fold) {
                yr = e0[0] * re + e0[1] * im + e4;
                yi = e0[2] * re + e0[3] * im + e5;
            } else {
                yr = re + e4;
                yi = im + e5;
            }
            if (has_act) {
                yr = yr >= 0.f ? yr : slope * yr;
                yi = yi >= 0.f ? yi : slope * yi;
            }
            yr = keep ? yr : 0.f;
            yi = keep ? yi : 0.f;
            if (cok && inb) {
                a.out[((size_t)co * a.Fout + fo) * a.Jp + j] = yr;
                a.out[((size_t)(a.Cout + co) * a.Fout + fo) * a.Jp + j] = yi;
            }
            if (STATS && inb && keep) {
                st[0] += yr;
                st[1] += yi;
                st[2] += yr * yr;
                st[3] += yi * yi;
                st[4] += yr * yi;
            }
        }
        if (STATS) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                float tsum = st[q];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o, 64);
                if (l31 == 0 && cok)
                    atomicAdd(&a.stats[((size_t)(a.stats_rep > 1 ? (blockIdx.x & (a.stats_rep - 1)) : 0) * a.Cout + co) * 5 + q], (double)tsum);
            }
        }
    }
}

// fragment element (phase, ct, unit = ci * 3 + p, q, lane): tap q of the phase's transformed taps, lane = kt * 32 + col supplies
// co = ct * 32 + col.  Index conventions (transposed / conj, kt) as pack_cconv_gauss_kernel.  transposed operator: two phases of
// 4 slots per unit (even-row taps | odd-row taps + padding); conv operator: one block of 8 slots per unit (7 taps + padding).
__global__ void pack_cconv_wino_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im, int Cout, int Cin_total,
                                       int Cin_used, int transposed, int conj, int UN, int cotiles, float* __restrict__ wfrag) {
    const long long n = 2LL * cotiles * UN * 64;                  // one thread per (half, ct, unit, lane): 4 floats
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long long t = idx >> 6;
        int unit, ct, hf;
        if (transposed) {                                          // [phase][ct][unit]
            unit = (int)(t % UN); t /= UN;
            ct = (int)(t % cotiles);
            hf = (int)(t / cotiles);
        } else {                                                   // [ct][unit][half]
            hf = (int)(t & 1); t >>= 1;
            unit = (int)(t % UN);
            ct = (int)(t / UN);
        }
        const int h = lane >> 5, co = ct * 32 + (lane & 31);
        const int ci = unit / 3, p3 = unit % 3;
        float W[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (co < Cout && ci < Cin_used) {
            const int kt = transposed ? 1 - h : h;
#pragma unroll
            for (int kf = 0; kf < 5; ++kf) {
                const size_t off = transposed ? (((size_t)ci * Cout + co) * 5 + kf) * 2 + kt
                                              : (((size_t)co * Cin_total + ci) * 5 + kf) * 2 + kt;
                const float wr = w_re[off], wi = conj ? -w_im[off] : w_im[off];
                W[kf] = p3 == 0 ? wr : (p3 == 1 ? wi - wr : wr + wi);
            }
        }
        float o[4];
        if (transposed) {
            if (hf == 0) {
                o[0] = W[4]; o[1] = 0.5f * (W[4] + W[2] + W[0]); o[2] = 0.5f * (W[4] - W[2] + W[0]); o[3] = W[0];
            } else {
                o[0] = W[3]; o[1] = W[3] + W[1]; o[2] = W[1]; o[3] = 0.f;
            }
        } else {
            if (hf == 0) {
                o[0] = W[0]; o[1] = 0.5f * (W[0] + W[2] + W[4]); o[2] = 0.5f * (W[0] - W[2] + W[4]); o[3] = W[4];
            } else {
                o[0] = W[1]; o[1] = W[1] + W[3]; o[2] = W[3]; o[3] = 0.f;
            }
        }
        float* dst = wfrag + ((idx >> 6) * 4) * 64 + lane;          // 4 slots of 64 lanes per (.., half)
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * 64] = o[q];
    }
}

template <int PH, int WM, int WN, int CIK, bool STATS, int OCC = 1, int RD = 0>
int launch_wino_ph(const WinoArgs& a, hipStream_t st) {
    static_assert(WCIK % CIK == 0, "a K chunk never straddles the pack granularity (nor, with it, the two sources)");
    constexpr int TR = wino_tr<PH>();
    constexpr int JT = 32 * WN;
    constexpr int NE = CIK * 3 * TR * (JT + 8);
    constexpr int NBUF = (3 * NE * sizeof(float) * OCC <= 156 * 1024) ? 3 : 2;
    constexpr size_t smem = NBUF * NE * sizeof(float);
    static_assert(smem * OCC <= 160 * 1024, "the patch buffers of OCC workgroups must fit the 160 KB of LDS");
    WinoArgs b = a;
    b.jtiles = (a.J + JT - 1) / JT;
    // row tiles: pairs of input rows (transposed) / output rows (conv).  With an odd number of input rows the last pair's second
    // row does not exist and BOTH odd output rows of that tile (2 Fin - 1, 2 Fin + 1) lie outside the output: the odd-row phase
    // skips the tile (a third of its work on the 5-row dec0, a fifth on dec1); in the even-row phase and in the conv that tile
    // has ONE output row and skips the products that only feed the missing one (kernel: half_tile)
    b.ftiles = PH == 2 ? (a.Fout + 1) / 2 : (PH == 1 ? a.Fin / 2 : (a.Fin + 1) / 2);
    b.mblocks = (a.cotiles + WM - 1) / WM;
    if (b.ftiles == 0) return IDV_OK;
    // co-tile blocks on different XCDs where a layer has 2 / 4 / 8 of them: dec0 (8 co tiles = 2 blocks) 10.75 -> 10.45 ms, no
    // change elsewhere (the L2-miss traffic is not what bounds these kernels); IDV_WINO_XCD_SPLIT=0: all blocks of a tile on one XCD
    static const int xsplit = [] { const char* e = getenv("IDV_WINO_XCD_SPLIT"); return e ? atoi(e) : 1; }();
    b.xcd_split = (xsplit && (b.mblocks == 2 || b.mblocks == 4 || b.mblocks == 8)) ? 1 : 0;
    long long nblk = (long long)((b.jtiles + 7) / 8) * 8 * b.ftiles * b.mblocks;
    if (b.xcd_split) {
        const int G = 8 / b.mblocks;
        nblk = (long long)((b.jtiles + G - 1) / G) * b.ftiles * 8;
    }
    if (nblk > 0x7fffffffLL) return IDV_EINVAL;
    auto k = cconv_wino_kernel<PH, WM, WN, CIK, NBUF, STATS, OCC, RD>;
    // OCC 1: more than half a CU's LDS, i.e. one workgroup per CU whatever the register count says
    const size_t smem_req = OCC == 1 ? (smem > 84 * 1024 ? smem : 84 * 1024) : smem;
    if (smem_req > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_req) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(WM * WN * 64), smem_req, st, b);
    return idv_launch_status();
}

// transposed conv: even rows, then odd rows: two launches per layer (the second reads the same raw rows from the L2 / Infinity
// Cache).  IDV_WINO_PH1 (experiments): 0 = the odd-row phase as the even one (one workgroup per CU, ring = a chunk); default: two
// workgroups per CU with a ring of 8 units (9 accumulator tiles = 144 registers leave room for it).  conv: one launch, the weight
// ring half a chunk deep (7 fragments per unit).
template <int WM, int WN, int CIK, bool STATS>
int launch_wino_s(const WinoArgs& a, int transposed, hipStream_t st) {
    if (!transposed) {
        // two channels per chunk: the conv form at TWO workgroups per CU (one staging item per thread, ring of three units: 253
        // registers, accumulators in VGPRs)
        if constexpr (CIK == 2) return launch_wino_ph<2, WM, WN, 2, STATS, 2, 3>(a, st);
        else return launch_wino_ph<2, WM, WN, CIK, STATS, 1, (CIK * 3) / 2>(a, st);
    }
    if constexpr (CIK == 2) return IDV_EINVAL;                 // (the transposed form picks its own chunk sizes below)
    else {
    static const int ph1 = [] { const char* e = getenv("IDV_WINO_PH1"); return e ? atoi(e) : 1; }();
    // the even-row phase runs at two workgroups per CU too where it fits 256 registers (accumulators in VGPRs): the four-co-tile
    // form with four channels per chunk and a weight ring of four units (249 registers): dec0 11.09 -> 10.49 ms, dec1 10.13 -> 9.62,
    // dec2 9.67 -> 9.03 (B = 64); the 2 x 2 form with two channels per chunk (one staging item per thread; with four it needs two
    // and spills) and a ring of three: dec3 10.37 -> 9.99.  IDV_WINO_PH0=1: one workgroup per CU
    static const int ph0 = [] { const char* e = getenv("IDV_WINO_PH0"); return e ? atoi(e) : 2; }();
    if (ph0 == 2 && WM == 4 && WN == 1) {
        if (int rc = launch_wino_ph<0, 4, 1, 4, STATS, 2, 4>(a, st)) return rc;
    } else if (ph0 == 2 && WM == 2 && WN == 2) {
        if (int rc = launch_wino_ph<0, 2, 2, 2, STATS, 2, 3>(a, st)) return rc;
    } else if (ph0 == 2 && WM == 1 && WN == 4) {          // one co tile x four column groups: two channels per chunk (5 registers spilled)
        if (int rc = launch_wino_ph<0, 1, 4, 2, STATS, 2, 3>(a, st)) return rc;
    } else if (int rc = launch_wino_ph<0, WM, WN, CIK, STATS>(a, st)) return rc;
    if (ph1 == 0) return launch_wino_ph<1, WM, WN, CIK, STATS>(a, st);
    return launch_wino_ph<1, WM, WN, CIK, STATS, 2, (CIK * 3) % 8 == 0 ? 8 : 6>(a, st);
    }
}
template <int WM, int WN, int CIK>
int launch_wino(const WinoArgs& a, int transposed, hipStream_t st) {
    return a.stats ? launch_wino_s<WM, WN, CIK, true>(a, transposed, st) : launch_wino_s<WM, WN, CIK, false>(a, transposed, st);
}

const bool USE_WINO = [] { const char* e = getenv("IDV_WINO"); return !e || e[0] != '0'; }();
const bool USE_WINO_CONV = [] { const char* e = getenv("IDV_WINO_CONV"); return !e || e[0] != '0'; }();

}  // namespace

// 1 if the Winograd form serves this layer: what cgemm_gauss serves, where it measured faster (B = 64, tests/tools/
// wino_layers_probe.py): transposed conv with more than one tile of 32 complex output channels (one co tile x four column groups was
// 3 % SLOWER than cgemm_gauss's two-workgroup form on dec4, 128 -> 32: the staging transform is then amortised over one co tile
// only) and at least two input rows; conv with >= 64 output and >= 32 input channels (at two workgroups per CU: enc1 2.99 -> 2.61
// ms on the 2 x 2 form, enc2 5.62 -> 4.90, enc3 5.66 -> 5.04, enc4 6.03 -> 5.44, enc5 6.73 -> 6.30 on four co tiles) and >= 2 output rows.
extern "C" int idv_cconv_wino_supported(int transposed, int C0, int C1, int Cout, int Fin) {
    static const int min_cout = [] { const char* e = getenv("IDV_WINO_MIN_COUT"); return e ? atoi(e) : 33; }();
    static const int conv_min = [] { const char* e = getenv("IDV_WINO_CONV_MINC"); return e ? atoi(e) : 32; }();
    if (!USE_WINO || (!transposed && !USE_WINO_CONV) || Cout < min_cout) return 0;
    static const int conv_mincout = [] { const char* e = getenv("IDV_WINO_CONV_MINCOUT"); return e ? atoi(e) : 64; }();
    if (transposed ? Fin < 2 : ((Fin - 1) / 2 + 1 < 2 || Cout < conv_mincout || C0 + C1 < conv_min)) return 0;
    return idv_cconv_gauss_supported(C0, C1, Cout);
}

extern "C" long long idv_cconv_wino_wfrag_floats(int transposed, int Cout, int cin_used) {
    (void)transposed;                                     // both forms: 8 slots of 64 lanes per (co tile, unit)
    const long long cotiles = (Cout + 31) / 32, cpad = (cin_used + WCIK - 1) / WCIK * WCIK;
    return cotiles * cpad * 3 * 8 * 64;
}

// configuration id for bench.py / profiles: WM WN CIK as decimal digits (418 = four co tiles x one column group, 8 channels per
// K chunk).  Per layer at B = 64 (tests/tools/wino_layers_probe.py, against cgemm_gauss): dec0 13.15 -> 12.97 ms, dec1 11.73 ->
// 10.94, dec2 11.46 -> 10.09 (418); dec3 11.75 -> 10.59 (228)
extern "C" int idv_cconv_wino_config(int transposed, int Cin, int Cout) {
    (void)Cin;
    // conv: four co tiles x one column group at two workgroups per CU (enc3 5.35 -> 5.04 ms, enc4 5.66 -> 5.44 against the
    // one-workgroup form with eight channels per chunk)
    if (!transposed) return Cout >= 128 ? 412 : 222;
    return Cout >= 128 ? 418 : (Cout > 32 ? 228 : 144);
}

// Winograd-transformed Gauss planes of a complex conv / transposed conv weight, conventions (transposed = the OPERATOR's mode =
// the tensor's layout, conj for the adjoint / data-gradient operators) as idv_pack_cconv_gauss; the epilogue table is that
// function's.  wfrag: idv_cconv_wino_wfrag_floats(transposed, Cout, Cin_used) floats.
extern "C" int idv_pack_cconv_wino(const float* w_re, const float* w_im, int Cout, int Cin_total, int Cin_used, int transposed,
                                   int conj, float* wfrag, void* stream) {
    if (!w_re || !w_im || !wfrag || Cout <= 0 || Cin_used <= 0 || Cin_used > Cin_total) return IDV_EINVAL;
    const int cotiles = (Cout + 31) / 32;
    const int UN = (Cin_used + WCIK - 1) / WCIK * WCIK * 3;
    const long long n = 2LL * cotiles * UN * 64;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_cconv_wino_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_re, w_im, Cout, Cin_total, Cin_used,
                       transposed, conj, UN, cotiles, wfrag);
    return idv_launch_status();
}

// idv_cconv2d_gauss_fwd (x1_div == 1; statistics as there) on the Winograd kernels: same result up to the rounding of the
// transforms.  wfrag from idv_pack_cconv_wino, epi / has_fold from idv_pack_cconv_gauss.  Requires 16-byte aligned sources with
// Jp % 4 == 0 and, with a second source, the same pitch (the callers' planar buffers are).  Reference:
// model/complex_progress.py:8-36, :222-279 (+ :161-209 and pvae_module.py:58,82 for the epilogue).
extern "C" int idv_cconv2d_wino_fwd(const float* x0, int C0, const float* x1, int C1, const float* wfrag, const float* epi, int has_fold,
                                    const float* prelu_slope, float* out, double* stats, double* stats_work, int stats_rep,
                                    int transposed, int tshift, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out,
                                    const float* addend, int addend_div, int addend_Jp, void* stream) {
    if (!x0 || !wfrag || !epi || !out || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (stats && stats_work && (stats_rep < 2 || (stats_rep & (stats_rep - 1)))) return IDV_EINVAL;
    if (addend && (addend_div < 1 || B % addend_div || addend_Jp < (B / addend_div) * Tp)) return IDV_EINVAL;
    if (C1 > 0 && !x1) return IDV_EINVAL;
    if (tshift != 0 && tshift != -1) return IDV_EINVAL;
    if (!idv_cconv_wino_supported(transposed, C0, C1, Cout, Fin)) return IDV_EINVAL;
    if ((Jp & 3) || (reinterpret_cast<uintptr_t>(x0) & 15) || (C1 > 0 && (reinterpret_cast<uintptr_t>(x1) & 15))) return IDV_EINVAL;
    WinoArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.Fin = Fin; a.Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp;
    a.wfrag = wfrag; a.UN = (C0 + C1 + WCIK - 1) / WCIK * WCIK * 3; a.epi = epi; a.has_fold = has_fold; a.slope = prelu_slope; a.out = out;
    a.Cout = Cout; a.cotiles = (Cout + 31) / 32;
    a.tshift = tshift; a.t_valid = t_valid_out;
    a.add = addend; a.add_div = addend ? addend_div : 1; a.add_Jp = addend_Jp;
    if (Jp < a.J) return IDV_EINVAL;
    if ((long long)WCIK * Fin * (long long)Jp >= 0xffffffffLL) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    a.stats = stats;
    if (stats && stats_work) { a.stats = stats_work; a.stats_rep = stats_rep; }       // replicated sums, folded afterwards (common.hpp)
    int rc;
    // experiments: IDV_WINO_CFG / IDV_WINO_CCFG = WM WN CIK as decimal digits for the transposed conv / the conv
    static const int xcfg = [] { const char* e = getenv("IDV_WINO_CFG"); return e ? atoi(e) : 0; }();
    static const int xccfg = [] { const char* e = getenv("IDV_WINO_CCFG"); return e ? atoi(e) : 0; }();
    int cfg = transposed ? xcfg : xccfg;
    if (!cfg) cfg = idv_cconv_wino_config(transposed, C0 + C1, Cout);
    if (C1 > 0 && C0 % (cfg % 10)) cfg = cfg / 10 * 10 + 4;
    if (transposed && cfg % 10 == 2) return IDV_EINVAL;          // a K chunk must not straddle the two sources (C0 % 4 == 0 holds)
    switch (cfg) {
        case 412: rc = launch_wino<4, 1, 2>(a, transposed, st); break;
        case 414: rc = launch_wino<4, 1, 4>(a, transposed, st); break;
        case 418: rc = launch_wino<4, 1, 8>(a, transposed, st); break;
        case 222: rc = launch_wino<2, 2, 2>(a, transposed, st); break;
        case 224: rc = launch_wino<2, 2, 4>(a, transposed, st); break;
        case 228: rc = launch_wino<2, 2, 8>(a, transposed, st); break;
        case 144: rc = launch_wino<1, 4, 4>(a, transposed, st); break;
        default: return IDV_EINVAL;
    }
    if (rc || !(stats && stats_work)) return rc;
    return idv_launch_stats_collapse(stats_work, stats_rep, Cout * 5, stats, st);
}
