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
// dealt to the four waves of a workgroup (tw_tile: even rows wave = r with its nine planes; odd rows waves 0 .. 2 = r with planes
// 0 .. 5 and 8, wave 3 the planes 6, 7 of all three r: 9 / 7 tiles per wave, two workgroups per CU) and the output transforms (time,
// Gauss, frequency) run in the epilogue on tiles exchanged through the LDS.
// Staging writes the RAW input rows, each as nine planes (s = r + i | r | i) x (a - b | b | b - d) of 32 column pairs, planes (2 p,
// 2 p + 1) interleaved per pair (one 8-byte read fetches the operands of two tiles); the FREQUENCY transform A + cb B of cgemm_wino
// happens at the operand read (two reads + one add).  Weights: cgemm_wino's fragments re-ordered by idv_pack_cconv_tw (lane =
// channel parity x 32 + co), per (channel pair, wave).  Design notes and measurements: DESIGN.md 3.1e.
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
    const float* wfrag;   // [phase 0: cotiles][UP][4 waves x 9 slots][64] then [phase 1: cotiles][UP][4 x 8][64] (idv_pack_cconv_tw)
    int UP;               // channel pairs per co tile as packed (Cin rounded up to the pack granularity, / 2)
    const float* epi;     // as cgemm_gauss: [cotiles * 32][8]
    int has_fold;
    const float* slope;
    float* out;           // planar [2][Cout][Fout][Jp]
    int Cout, cotiles;
    int tshift, t_valid;
    double* stats;        // train mode: [Cout][5] sums (r, i, rr, ii, ri) of the outputs, or nullptr (as cgemm_gauss)
    int stats_rep;        // > 1: that many replicas [rep][Cout][5] (power of two), one chosen per workgroup
    const float* add;     // optional addend (see cgemm_gauss.hip)
    int add_div, add_Jp;
    int jtiles, ftiles;
    int xcd_split;        // block order: the co tiles of a column block on different XCDs (see the kernel)
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
template <int PH> constexpr int tw_ntp() { return PH == 0 ? 36 : 32; }         // weight slots per channel pair: 4 waves x 9 / 8
template <int PH> constexpr int tw_ntw() { return (tw_nt<PH>() + 3) / 4; }     // tiles per wave: 9 / 7
// tile k of wave w: t = r * 9 + plane, or -1 (the odd-row phase's 28th slot: zero taps, never read).  Tiles come as NPAIR pairs of
// planes (2 p, 2 p + 1) of ONE frequency product + one single tile (plane 8): even-row phase, wave = r: all nine planes of r.
// Odd-row phase: waves 0 .. 2 = r: planes 0 .. 5 and 8; wave 3: planes 6, 7 of r = 0, 1, 2.
__host__ __device__ inline int tw_tile(int ph, int w, int k) {
    if (ph == 0) return w * 9 + k;
    if (w < 3) return k < 6 ? w * 9 + k : w * 9 + 8;
    return k < 6 ? (k >> 1) * 9 + 6 + (k & 1) : -1;
}
template <int PH> constexpr int tw_wslots() { return PH == 0 ? 9 : 8; }        // packed weight slots per (channel pair, wave)

// DBG (timing experiments only, results wrong): 1 = no staging after the prologue, 2 = no weight re-loads, 4 = no epilogue exchange
// LEFT: the time taps read (x[t-1], x[t]) (tshift = -1: the extra window column is on the left), else (x[t], x[t+1])
template <int PH, int CIK, bool LEFT, int DBG = 0, int RDW = 2, bool WVEC = true, bool STATS = false>
__global__ __launch_bounds__(256, 2) void cconv_tw_kernel(const TwArgs a) {
    constexpr int NR = tw_nr<PH>(), NT = tw_nt<PH>(), NTP = tw_ntp<PH>(), NTW = tw_ntw<PH>();
    constexpr int NRAW = PH == 0 ? 4 : 3, ROW0 = PH == 0 ? 0 : 1;       // raw patch rows d(ROW0) .. : input rows m0 - 1 + ROW0 ..
    constexpr int KS = CIK / 2;                  // MFMA k-steps (channel pairs) per chunk
    constexpr int RT = NRAW * 9 * 32;            // floats per channel in a patch buffer: per raw row 9 (Gauss, time) planes of 32 pairs
    constexpr int NE = CIK * RT;
    constexpr int NBUF = 2;
    constexpr int NITEM = CIK * NRAW * 16;       // staging items per chunk: (channel, raw row, 2 column pairs)
    constexpr int NLD = (NITEM + 255) / 256;
    static_assert(KS % RDW == 0 && NLD <= 2, "weight ring slots are compile-time; at most two staging items per thread");
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
    int jt, ft, ct;
    if (a.xcd_split) {
        // 2 / 4 / 8 co tiles: co tile ct always on the XCDs = ct (mod cotiles), so that an XCD streams 1 / cotiles of the layer's taps
        // (2 - 5 MB: its L2 holds them) and the raw rows of a column block are read by cotiles XCDs instead of one
        const int G = 8 / a.cotiles;
        ct = xcd % a.cotiles;
        jt = (slot / a.ftiles) * G + xcd / a.cotiles;
        ft = slot - (slot / a.ftiles) * a.ftiles;
    } else {
        const int per = a.cotiles * a.ftiles;
        jt = (slot / per) * 8 + xcd;
        const int rem = slot - (slot / per) * per;
        ft = rem / a.cotiles;
        ct = rem - ft * a.cotiles;
    }
    if (jt >= a.jtiles) return;
    const int j0 = jt * 64;
    const int m0 = 2 * ft;
    const int rbase = m0 - 1 + ROW0;

    const int Cin = a.C0 + a.C1;
    const int nchunk = (Cin + CIK - 1) / CIK;

    f32x16 acc[NTW];
#pragma unroll
    for (int k = 0; k < NTW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    // this wave's tiles (tw_tile): NPAIR pairs of planes + one single; slot j of the wave reads raw rows ra(r_j), rb(r_j) at plane pair
    // pj_j -- the frequency transform  A + cb B  happens at the operand read.  In a half tile the products r = 3 do work nobody reads
    // (they only feed the missing output row), branch-free.
    constexpr int NPAIR = (NTW - 1) / 2;
    constexpr bool UNI = PH == 0;             // every slot of the wave belongs to ONE frequency product: one row offset, constant plane offsets
    int offA[UNI ? 1 : NPAIR + 1], offB[UNI ? 1 : NPAIR + 1];
    float cbj[UNI ? 1 : NPAIR + 1];
    if (UNI) {
        cbj[0] = tw_cb<PH>(wave);
        offA[0] = (tw_ra<PH>(wave) - ROW0) * 288;
        offB[0] = (tw_rb<PH>(wave) - ROW0) * 288;             // (cb is never 0 in this phase)
    } else {
#pragma unroll
        for (int j = 0; j <= NPAIR; ++j) {
            int t = tw_tile(PH, wave, j < NPAIR ? 2 * j : NTW - 1);
            t = t < 0 ? 0 : t;
            const int r = t / 9, plane = t - r * 9;
            cbj[j] = tw_cb<PH>(r);
            const int po = plane == 8 ? 256 : (plane >> 1) * 64;
            offA[j] = (tw_ra<PH>(r) - ROW0) * 288 + po;
            offB[j] = cbj[j] != 0.f ? (tw_rb<PH>(r) - ROW0) * 288 + po : offA[j];
        }
    }

    // ---- staging: one item = channel cl, raw row, column pairs 2 c8, 2 c8 + 1 (output columns j0 + 4 c8 .. + 3): window columns
    // w0..w4 = input columns jc + tshift .. jc + 4 + tshift, real and imaginary -> the 9 planes (s = r + i | r | i) x (a - b | b | b - d)
    f32x4 v_r[NLD], v_i[NLD];
    float e_r[NLD], e_i[NLD];
    unsigned off_v[NLD], off_e[NLD], ldsoff[NLD];
    unsigned okmask[NLD];     // bits 0-4: window column valid; bit 5: row valid; bit 7: item exists
    int item_cl[NLD];
    bool interior[NLD];       // wave-uniform: every lane's item has its row and all five window columns inside the tensor
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * 256;
        const int c8 = e & 15;
        const int rl = (e >> 4) % NRAW, cl = e / (16 * NRAW);
        item_cl[i] = cl;
        const int f = rbase + rl;
        const int jc = j0 + 4 * c8;
        const int je = LEFT ? jc - 1 : jc + 4;
        const bool exists = e < NITEM;
        const bool okr = exists && f >= 0 && f < a.Fin;
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int c = jc + q + (LEFT ? -1 : 0);
            if (c >= 0 && c < a.J) m |= 1u << q;
        }
        if (okr) m |= 1u << 5;
        if (exists) m |= 1u << 7;
        // addresses stay inside mapped memory whatever the masks say: vector slot clamped to the row, the extra column to [0, Jp)
        const int jcv = jc + 3 < a.Jp ? jc : 0;
        const int jev = (je >= 0 && je < a.Jp) ? je : 0;
        off_v[i] = okr ? (unsigned)((cl * a.Fin + f) * a.Jp + jcv) : 0u;
        off_e[i] = okr ? (unsigned)((cl * a.Fin + f) * a.Jp + jev) : 0u;
        if (jc + 3 >= a.Jp) m &= ~0x1fu;                      // (a column block past the row pitch: nothing valid)
        if (!(je >= 0 && je < a.Jp)) m &= LEFT ? ~1u : ~(1u << 4);
        okmask[i] = m;
        interior[i] = __builtin_amdgcn_ballot_w64((m & 0xbfu) == 0xbfu) == ~0ull;
        ldsoff[i] = (unsigned)((cl * NRAW + rl) * 288 + 4 * c8);
    }
    auto stage_load = [&](int chunk, int i) {
        const int ci0 = chunk * CIK;
        const bool from0 = ci0 < a.C0;
        const float* br = from0 ? a.x0 + (size_t)ci0 * a.Fin * a.Jp : a.x1 + (size_t)(ci0 - a.C0) * a.Fin * a.Jp;
        const float* bi = from0 ? br + (size_t)a.C0 * a.Fin * a.Jp : br + (size_t)a.C1 * a.Fin * a.Jp;
        const int cvalid = (from0 ? a.C0 : Cin) - ci0;
        const bool dead = item_cl[i] >= cvalid;
        const unsigned ov = dead ? 0u : off_v[i], oe = dead ? 0u : off_e[i];
        v_r[i] = *(const f32x4*)(br + ov);
        v_i[i] = *(const f32x4*)(bi + ov);
        e_r[i] = br[oe];
        e_i[i] = bi[oe];
    };
    // the store of an item in ten steps that the main loop deals out between MFMAs (step 0: masks and the window of five columns;
    // steps 1 .. 9: one (Gauss, time) plane each), so that the vector ALU work hides behind the matrix pipe
    float fr[5], fi[5];
    auto stage_window = [&](int chunk, int i) {
        const int ci0s = chunk * CIK;
        const int cvalid = (ci0s < a.C0 ? a.C0 : Cin) - ci0s;
        constexpr bool left = LEFT;
        if (interior[i] && cvalid >= CIK) {                   // (uniform) nothing to mask
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                fr[q] = left ? (q == 0 ? e_r[i] : v_r[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_r[i] : v_r[i][q == 4 ? 3 : q]);
                fi[q] = left ? (q == 0 ? e_i[i] : v_i[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_i[i] : v_i[i][q == 4 ? 3 : q]);
            }
            return;
        }
        unsigned m = okmask[i];
        if (item_cl[i] >= cvalid || !((m >> 5) & 1u)) m &= ~0x1fu;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            // window column q: the extra column is w0 (taps to the left) or w4; compile-time vector indices on either side
            const float xr = left ? (q == 0 ? e_r[i] : v_r[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_r[i] : v_r[i][q == 4 ? 3 : q]);
            const float xi = left ? (q == 0 ? e_i[i] : v_i[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_i[i] : v_i[i][q == 4 ? 3 : q]);
            const bool cok = (m >> q) & 1u;
            fr[q] = cok ? xr : 0.f;
            fi[q] = cok ? xi : 0.f;
        }
    };
    auto plane_val = [&](int gt, int pr) -> float {           // plane gt = g * 3 + tau of column pair pr (0 / 1) of the item
        const int g = gt / 3, tau = gt - g * 3;
        // pair 0: (a, b, d) = window columns 0, 1, 2; pair 1: 2, 3, 4
        const int q0 = 2 * pr;
        const float xa_ = g == 0 ? fr[q0] + fi[q0] : (g == 1 ? fr[q0] : fi[q0]);
        const float xb_ = g == 0 ? fr[q0 + 1] + fi[q0 + 1] : (g == 1 ? fr[q0 + 1] : fi[q0 + 1]);
        const float xd_ = g == 0 ? fr[q0 + 2] + fi[q0 + 2] : (g == 1 ? fr[q0 + 2] : fi[q0 + 2]);
        return tau == 0 ? xa_ - xb_ : (tau == 1 ? xb_ : xb_ - xd_);
    };
    // an item exists for every thread in the even-row phase (512 items); in the odd-row phase the second item only in waves 0, 1
    auto item_exists = [&](int i) -> bool { return NITEM >= (i + 1) * 256 || wave * 64 + i * 256 < NITEM; };
    // LDS layout of a raw row: planes (2 j, 2 j + 1) interleaved per column pair at [j][32 pairs][2], plane 8 at [256 + pair]:
    // a lane fetches the operands of two tiles with one 8-byte read, an item writes two planes x two pairs with one 16-byte write
    auto stage_plane2 = [&](float* dst, int i, int j) {
        if (!item_exists(i)) return;                          // (uniform)
        if (j < 4) {
            f32x4 o = {plane_val(2 * j, 0), plane_val(2 * j + 1, 0), plane_val(2 * j, 1), plane_val(2 * j + 1, 1)};
            *(f32x4*)(dst + ldsoff[i] + j * 64) = o;
        } else {
            *(float2*)(dst + ldsoff[i] - 2 * (int)(tid & 15) + 256) = make_float2(plane_val(8, 0), plane_val(8, 1));
        }
    };
    auto stage_store = [&](float* dst, int chunk, int i) {
        stage_window(chunk, i);
#pragma unroll
        for (int j = 0; j < 5; ++j) stage_plane2(dst, i, j);
    };

    // ---- weights in a ring of RDW k-steps: the wave's tiles of a channel pair are packed as groups of [64 lanes][4 tiles] (lane =
    // channel parity x 32 + co; idv_pack_cconv_tw): even-row phase two 16-byte loads + one 4-byte load ([64 lanes]) per k-step,
    // odd-row phase two 16-byte loads
    constexpr int WS = tw_wslots<PH>();
    const float* wbase = a.wfrag + ((size_t)(PH == 1 ? (size_t)a.cotiles * a.UP * tw_ntp<0>() : 0) + (size_t)ct * a.UP * NTP) * 64 +
                         (size_t)wave * WS * 64;
    const int total_ks = nchunk * KS;
    float a_w[RDW][NTW];
    auto load_w = [&](int g, float (&dst)[NTW]) {
        g = g < total_ks ? g : total_ks - 1;                  // (past the end of K: an unused re-fetch)
        const float* ws = wbase + (size_t)g * NTP * 64;
        if (WVEC) {
            const f32x4 w0 = *(const f32x4*)(ws + lane * 4), w1 = *(const f32x4*)(ws + 256 + lane * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dst[k] = w0[k];
                if (4 + k < NTW) dst[4 + k] = w1[k];
            }
            if (NTW == 9) dst[NTW - 1] = ws[512 + lane];
        } else {                                              // (experiments) slot-major [slot][64 lanes]: one 4-byte load per tile
#pragma unroll
            for (int k = 0; k < NTW; ++k) dst[k] = ws[k * 64 + lane];
        }
    };
    float xa[NTW], xb[NTW];
    // operands of tiles 2 j, 2 j + 1 (j < NPAIR: one 8-byte read per raw row) or of the single tile; base = buffer + channel of this half-wave
    auto load_raw2 = [&](const float* base, int j) {
        if (j < NPAIR) {
            const int oa = UNI ? offA[0] + j * 64 : offA[UNI ? 0 : j], ob = UNI ? offB[0] + j * 64 : offB[UNI ? 0 : j];
            const float2 pa_ = *(const float2*)(base + oa + 2 * l31), pb_ = *(const float2*)(base + ob + 2 * l31);
            xa[2 * j] = pa_.x; xa[2 * j + 1] = pa_.y;
            xb[2 * j] = pb_.x; xb[2 * j + 1] = pb_.y;
        } else {
            xa[NTW - 1] = base[(UNI ? offA[0] + 256 : offA[UNI ? 0 : NPAIR]) + l31];
            xb[NTW - 1] = base[(UNI ? offB[0] + 256 : offB[UNI ? 0 : NPAIR]) + l31];
        }
    };
    auto load_raw_all = [&](const float* base) {
#pragma unroll
        for (int j = 0; j <= NPAIR; ++j) load_raw2(base, j);
    };

#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_load(0, i);
#pragma unroll
    for (int u = 0; u < RDW; ++u) load_w(u, a_w[u]);
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_store(smem, 0, i);
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_load(nchunk > 1 ? 1 : 0, i);
    __syncthreads();

    load_raw_all(smem + (size_t)half * RT);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const float* P = smem + (chunk & 1) * NE;
        float* Pn = smem + ((chunk + 1) & 1) * NE;
        const int nxt = chunk + 1 < nchunk ? chunk + 1 : chunk, nxt2 = chunk + 2 < nchunk ? chunk + 2 : nchunk - 1;
#pragma unroll
        for (int ul = 0; ul < KS; ++ul) {
            // operands of the next k-step: of this chunk, or (last k-step, after the barrier below) of the next chunk
            const float* bnext = ul + 1 < KS ? P + (size_t)(2 * (ul + 1) + half) * RT : Pn + (size_t)half * RT;
            // staging of chunk + 1 rides on k-steps 1 (item 0) and 2 (item 1): the item's registers were loaded one chunk ago, the
            // other buffer was last read in the previous chunk (a barrier since); the steps of the store are dealt out between MFMAs
            const int si = ul - 1;
            const bool staging = !(DBG & 1) && si >= 0 && si < NLD;
#pragma unroll
            for (int k = 0; k < NTW; ++k) {
                const float b = xa[k] + cbj[UNI ? 0 : (k >> 1 < NPAIR ? k >> 1 : NPAIR)] * xb[k];
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[ul % RDW][k], b, acc[k], 0, 0, 0);
                if (ul == KS - 1 && k == 0) {
                    // every wave has stored chunk + 1 (k-steps 1, 2) and issued its last reads of P: after this barrier Pn is complete
                    // and P may be overwritten (from k-step 1 of the next chunk on)
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                }
                // the operand registers of the tiles done so far are free again
                if ((k & 1) && (k >> 1) < NPAIR) load_raw2(bnext, k >> 1);
                if (k == NTW - 1) load_raw2(bnext, NPAIR);
                if (staging) {
                    // the store's steps between the MFMAs: window at 0, plane pairs at 1, 3, 5 and (7 or, with seven tiles, 6), plane 8 last
                    if (k == 0) stage_window(nxt, si);
                    if ((k & 1) && k < 6) stage_plane2(Pn, si, k >> 1);
                    if (k == (NTW == 9 ? 7 : 6)) stage_plane2(Pn, si, 3);
                    if (k == NTW - 1) stage_plane2(Pn, si, 4);
                    if (k == NTW - 1) stage_load(nxt2, si);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(DBG & 2)) load_w(chunk * KS + ul + RDW, a_w[ul % RDW]);
        }
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
    int ja[2];                                                // the addend's column: utterance b / add_div of its own buffer
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = jA + q;
        const int bj = j / a.Tp, tp = j - bj * a.Tp;
        inb[q] = j < a.J;
        keep[q] = inb[q] && tp >= 1 && tp <= a.t_valid;
        ja[q] = j - (bj - bj / a.add_div) * a.Tp;
    }
#pragma unroll
    for (int s = 0; s < ((DBG & 4) ? 1 : 4); ++s) {
        if (s > 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < NTW; ++k) {
            const int t = tw_tile(PH, wave, k);
            if (t >= 0) {
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
        float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
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
                if (a.add && cok && inb[q]) {
                    re += a.add[((size_t)co * a.Fout + fo) * a.add_Jp + ja[q]];
                    im += a.add[((size_t)(a.Cout + co) * a.Fout + fo) * a.add_Jp + ja[q]];
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
                if (STATS && keep[q]) {
                    st[0] += yr[q];
                    st[1] += yi[q];
                    st[2] += yr[q] * yr[q];
                    st[3] += yi[q] * yi[q];
                    st[4] += yr[q] * yi[q];
                }
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
        if (STATS) {
            // the 32 lanes of a half-wave hold the 32 column pairs of ONE output channel
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

// cgemm_wino's fragments [phase][ct][unit = ci * 3 + g][slot r (4)][lane = h * 32 + co] (h: the two time taps as the MFMA's two k,
// h = 0 multiplies column j + tshift) -> [phase][ct][pair u][tile t = r * 9 + g * 3 + tau (36 | 28 slots)][lane = parity * 32 + co]
// with the time-transformed taps  tau 0: W_h0,  tau 1: W_h0 + W_h1,  tau 2: W_h1.
__global__ void pack_cconv_tw_kernel(const float* __restrict__ wino, int cotiles, int UN, int UP, int vec, float* __restrict__ out) {
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
        // per (channel pair u, wave w) WS = 9 / 8 slots: groups of [64 lanes][4 tiles] (+ one of [64 lanes] in the even-row phase)
        const int ws_ = ph ? tw_wslots<1>() : tw_wslots<0>();
        const int pos = t * 64 + lane, w = pos / (ws_ * 64), q = pos % (ws_ * 64);
        const bool vec_ = (vec >> ph) & 1;
        const int k = !vec_ ? q >> 6 : (q < 512 ? (q >> 8) * 4 + (q & 3) : 8);
        const int ln = !vec_ ? q & 63 : (q < 512 ? (q & 255) >> 2 : q - 512);
        const int tt = k < (ph ? 7 : 9) ? tw_tile(ph, w, k) : -1;
        const int ci = 2 * u + (ln >> 5), co = ln & 31;
        if (tt >= 0 && ci * 3 < UN) {
            const int r = tt / 9, g = (tt % 9) / 3, tau = tt % 3;
            const float* src = wino + ((((size_t)ph * cotiles + ct) * UN + (size_t)ci * 3 + g) * 4 + r) * 64;
            const float w0 = src[co], w1 = src[32 + co];
            val = tau == 0 ? w0 : (tau == 1 ? w0 + w1 : w1);
        }
        out[idx] = val;
    }
}

template <int PH, int CIK, bool LEFT, int DBG, int RDW, bool WVEC = true, bool STATS = false>
int launch_tw_ph_l(const TwArgs& a, hipStream_t st);
// weight fragments as [64 lanes][4 tiles] groups with 16-byte loads (bit PH set) or slot-major [slot][64 lanes] with 4-byte loads: measured
// (B = 64, dec0-3) even-row phase 22.0 -> 21.1 ms with the vector form, odd-row phase 14.1 -> 17.8 ms: default 1 = even-row phase only
// (IDV_TW_WVEC=0 .. 3 for experiments; read by the pack kernel and the launcher alike)
inline int tw_wvec_mask() {
    static const int v = [] { const char* e = getenv("IDV_TW_WVEC"); return e ? atoi(e) : 1; }();
    return v;
}
template <int PH, int CIK, int DBG = 0, int RDW = 2>
int launch_tw_ph(const TwArgs& a, hipStream_t st) {
    const bool wv = (tw_wvec_mask() >> PH) & 1;
    if (DBG == 0 && RDW == 2 && a.stats) {                    // the training forward: default weight layouts only
        constexpr bool WV = PH == 0;
        if (wv != WV) return IDV_EINVAL;
        return a.tshift ? launch_tw_ph_l<PH, CIK, true, 0, 2, WV, true>(a, st) : launch_tw_ph_l<PH, CIK, false, 0, 2, WV, true>(a, st);
    }
    if (DBG == 0 && RDW == 2 && !wv)
        return a.tshift ? launch_tw_ph_l<PH, CIK, true, 0, 2, false>(a, st) : launch_tw_ph_l<PH, CIK, false, 0, 2, false>(a, st);
    return a.tshift ? launch_tw_ph_l<PH, CIK, true, DBG, RDW>(a, st) : launch_tw_ph_l<PH, CIK, false, DBG, RDW>(a, st);
}
template <int PH, int CIK, bool LEFT, int DBG, int RDW, bool WVEC, bool STATS>
int launch_tw_ph_l(const TwArgs& a, hipStream_t st) {
    constexpr int NE = CIK * (PH == 0 ? 4 : 3) * 9 * 32;
    constexpr size_t smem = 2 * NE * sizeof(float);
    static_assert(smem * 2 <= 160 * 1024, "the patch buffers of two workgroups must fit the 160 KB of LDS");
    TwArgs b = a;
    b.jtiles = (a.J + 63) / 64;
    b.ftiles = PH == 1 ? a.Fin / 2 : (a.Fin + 1) / 2;
    if (b.ftiles == 0) return IDV_OK;
    // co tiles on different XCDs: dec0-3 at B = 64 34.7 -> 34.3 ms (dec0, eight co tiles: 9.81 -> 9.54); IDV_TW_XCD_SPLIT=0: one XCD per column block
    static const int xsplit = [] { const char* e = getenv("IDV_TW_XCD_SPLIT"); return e ? atoi(e) : 1; }();
    b.xcd_split = (xsplit && (b.cotiles == 2 || b.cotiles == 4 || b.cotiles == 8)) ? 1 : 0;
    long long nblk = (long long)((b.jtiles + 7) / 8) * 8 * b.ftiles * b.cotiles;
    if (b.xcd_split) {
        const int G = 8 / b.cotiles;
        nblk = (long long)((b.jtiles + G - 1) / G) * b.ftiles * 8;
    }
    if (nblk > 0x7fffffffLL) return IDV_EINVAL;
    auto k = cconv_tw_kernel<PH, CIK, LEFT, DBG, RDW, WVEC, STATS>;
    // (once per instantiation and device: setting it on every launch is host time, a lot of it under a profiler)
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return IDV_ELAUNCH;
    if (smem > 64 * 1024 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return IDV_ELAUNCH;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), smem, st, b);
    return idv_launch_status();
}

}  // namespace

// 1 if idv_ctconv2d_tw_fwd serves the layer: a transposed conv that cgemm_gauss serves, with at least two input rows, at least one
// FULL tile of 32 complex output channels (a workgroup is one co tile x 64 columns, so unlike cgemm_wino's four-co-tile form the
// 32-channel layer dec4 is served: 6.13 -> see DESIGN.md 3.1e) and, with a second source, C0 a multiple of 8 (a K chunk of 8 channels
// never straddles the sources).  IDV_TW_MIN_COUT (experiments): narrowest output served.
extern "C" int idv_cconv_tw_supported(int C0, int C1, int Cout, int Fin) {
    static const int min_cout = [] { const char* e = getenv("IDV_TW_MIN_COUT"); return e ? atoi(e) : 32; }();
    if (Cout < min_cout || Fin < 2) return 0;
    if (!idv_cconv_gauss_supported(C0, C1, Cout)) return 0;
    return (C1 == 0 || C0 % 8 == 0) ? 1 : 0;
}

extern "C" long long idv_cconv_tw_wfrag_floats(int Cout, int cin_used) {
    const long long cotiles = (Cout + 31) / 32, cpad = (cin_used + TW_PACK_CI - 1) / TW_PACK_CI * TW_PACK_CI;
    return cotiles * (cpad / 2) * (36 + 32) * 64;
}

// wino_frag: idv_pack_cconv_wino(transposed = 1) of the same weights; tw_frag: idv_cconv_tw_wfrag_floats floats
extern "C" int idv_pack_cconv_tw(const float* wino_frag, int Cout, int cin_used, float* tw_frag, void* stream) {
    if (!wino_frag || !tw_frag || Cout <= 0 || cin_used <= 0) return IDV_EINVAL;
    const int cotiles = (Cout + 31) / 32;
    const int cpad = (cin_used + TW_PACK_CI - 1) / TW_PACK_CI * TW_PACK_CI;
    const long long n = idv_cconv_tw_wfrag_floats(Cout, cin_used);
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_cconv_tw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, wino_frag, cotiles, cpad * 3, cpad / 2,
                       tw_wvec_mask(), tw_frag);
    return idv_launch_status();
}

// idv_cconv2d_wino_fwd (transposed = 1; statistics and addend as there) on the time-Winograd kernels: same result up to the rounding of
// the transforms.  wfrag from idv_pack_cconv_tw, epi / has_fold from idv_pack_cconv_gauss.  Requires 16-byte aligned sources, Jp % 4
// == 0 and, with a second source, the same pitch.  Reference: model/complex_progress.py:222-279 (+ :161-209 and pvae_module.py:82
// for the epilogue).
extern "C" int idv_ctconv2d_tw_fwd(const float* x0, int C0, const float* x1, int C1, const float* wfrag, const float* epi, int has_fold,
                                   const float* prelu_slope, float* out, double* stats, double* stats_work, int stats_rep, int tshift,
                                   int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out, const float* addend, int addend_div,
                                   int addend_Jp, void* stream) {
    if (!x0 || !wfrag || !epi || !out || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (stats && stats_work && (stats_rep < 2 || (stats_rep & (stats_rep - 1)))) return IDV_EINVAL;
    if (addend && (addend_div < 1 || B % addend_div || addend_Jp < (B / addend_div) * Tp)) return IDV_EINVAL;
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
    a.add = addend; a.add_div = addend ? addend_div : 1; a.add_Jp = addend_Jp;
    if (Jp < a.J) return IDV_EINVAL;
    if ((long long)8 * Fin * (long long)Jp >= 0xffffffffLL) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    a.stats = stats;
    if (stats && stats_work) { a.stats = stats_work; a.stats_rep = stats_rep; }       // replicated sums, folded afterwards (common.hpp)
    int rc = 0;
#ifdef IDV_TW_EXPERIMENTS
    // timing experiments (WRONG results by construction; compiled in only with -DIDV_TW_EXPERIMENTS): IDV_TW_DBG = the kernel's DBG bits
    // (1: no staging after the prologue, 2: no weight re-loads), IDV_TW_ONLY = 1 / 2: one phase only
    static const int dbg = [] { const char* e = getenv("IDV_TW_DBG"); return e ? atoi(e) : 0; }();
    static const int only = [] { const char* e = getenv("IDV_TW_ONLY"); return e ? atoi(e) : 0; }();
    if (dbg && !stats) {
        if (only != 2) rc = dbg == 1 ? launch_tw_ph<0, 8, 1>(a, st) : (dbg == 2 ? launch_tw_ph<0, 8, 2>(a, st) : launch_tw_ph<0, 8, 3>(a, st));
        if (rc) return rc;
        if (only != 1) rc = dbg == 1 ? launch_tw_ph<1, 8, 1>(a, st) : (dbg == 2 ? launch_tw_ph<1, 8, 2>(a, st) : launch_tw_ph<1, 8, 3>(a, st));
        return rc;
    }
    if (only) return only == 1 ? launch_tw_ph<0, 8>(a, st) : launch_tw_ph<1, 8>(a, st);
#endif
    rc = launch_tw_ph<0, 8>(a, st);
    if (!rc) rc = launch_tw_ph<1, 8>(a, st);
    if (rc || !(stats && stats_work)) return rc;
    return idv_launch_stats_collapse(stats_work, stats_rep, Cout * 5, stats, st);
}
