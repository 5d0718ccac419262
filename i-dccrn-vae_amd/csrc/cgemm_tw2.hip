// cgemm_tw2: the complex Conv2d contraction (encoder block: kernel (5, 2), stride (2, 1); model/complex_progress.py:8-36) in the form of
// cgemm_tw.hip -- three real products per complex product (Gauss), the FREQUENCY taps in Winograd form (cgemm_wino.hip's conv form: 7
// products on the raw rows r0..r6 = input rows 2 fo - 2 .. 2 fo + 4 of a pair of output rows, into FOUR accumulators
//     A0 = (r0 - r4) W0 + (r1 - r3) W1          A1 = (r2 + r4) (W0 + W2 + W4)/2 + r3 (W1 + W3)
//     A2 = (r4 - r2) (W0 - W2 + W4)/2           A3 = (r2 - r6) W4 + (r3 - r5) W3
//     out[fo] = A0 + A1 + A2                    out[fo + 1] = A1 - A2 - A3 ),
// and the two TIME taps as F(2,2) over pairs of output columns (cgemm_tw.hip: three products (a - b) Wa, b (Wa + Wb), (b - d) Wb per
// column pair, the MFMA's two k = two input channels).  Per (channel pair, co tile, 64 columns, pair of output rows): 7 x 3 x 3 = 63
// MFMAs into 4 x 9 = 36 accumulator tiles -- 0.39 of the reference's real products.
//
// The 63 MFMAs do not divide by four waves along any one axis; the deal that balances them with nine accumulators per wave:
//     waves 0, 1, 2 = accumulator A0, A1, A3 (two products each) on planes 0 .. 6      -> 14 MFMAs
//                   + accumulator A2 (one product) on planes 2 w, 2 w + 1              ->  2 MFMAs       (16 per k-step)
//     wave 3        = A0, A1, A3 on planes 7, 8 (12 MFMAs) + A2 on planes 6, 7, 8 (3)                     (15 per k-step)
// Staging, LDS layout (raw rows, planes paired per column pair), weight ring and the epilogue exchange are cgemm_tw.hip's; the
// frequency transform A + cb B happens at the operand read.  Weights: cgemm_wino's conv fragments re-ordered by idv_pack_cconv_tw2.
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include "cgemm.hpp"
#include "../../include/idccrn_hip.h"

namespace {

struct Tw2Args {
    const float* x0;      // planar [2][Cin][Fin][Jp]
    int Cin;
    int Fin, Fout;
    int J, Jp, Tp;
    const float* wfrag;   // [cotiles][UP][4 waves][4 groups][64 lanes][4 slots] (idv_pack_cconv_tw2)
    int UP;               // channel pairs per co tile as packed
    const float* epi;     // as cgemm_gauss: [cotiles * 32][8]
    int has_fold;
    const float* slope;
    float* out;           // planar [2][Cout][Fout][Jp]
    int Cout, cotiles;
    int tshift, t_valid;
    double* stats;
    int stats_rep;
    int jtiles, ftiles;
    int xcd_split;
};

constexpr int TW2_PACK_CI = 8;     // pack granularity in complex input channels (cgemm_wino's WCIK)

// product q of the conv form: raw rows (ra, rb), factor cb, accumulator
__host__ __device__ constexpr int tw2_ra(int q) { return q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 4 : (q == 3 ? 2 : (q == 4 ? 1 : 3)))); }
__host__ __device__ constexpr int tw2_rb(int q) { return q == 0 ? 4 : (q == 1 ? 4 : (q == 2 ? 2 : (q == 3 ? 6 : (q == 4 ? 3 : (q == 5 ? 3 : 5))))); }
__host__ __device__ constexpr float tw2_cb(int q) { return q == 1 ? 1.f : (q == 5 ? 0.f : -1.f); }
// the two products of accumulator a (a = 0, 1, 3) and the product of accumulator 2
__host__ __device__ constexpr int tw2_qx(int a) { return a == 0 ? 0 : (a == 1 ? 1 : 3); }
__host__ __device__ constexpr int tw2_qy(int a) { return a == 0 ? 4 : (a == 1 ? 5 : 6); }
__host__ __device__ constexpr int tw2_main_acc(int w) { return w == 2 ? 3 : w; }          // wave 0, 1, 2 -> A0, A1, A3
// MFMA slot k (0 .. 15) of wave w -> product q and plane, or q = -1 (wave 3's 16th slot)
__host__ __device__ inline void tw2_slot(int w, int k, int& q, int& plane) {
    if (w < 3) {
        if (k < 14) { q = (k & 1) ? tw2_qy(tw2_main_acc(w)) : tw2_qx(tw2_main_acc(w)); plane = k >> 1; }
        else { q = 2; plane = 2 * w + (k - 14); }
    } else {
        if (k < 12) { const int a = tw2_main_acc(k >> 2); q = (k & 1) ? tw2_qy(a) : tw2_qx(a); plane = 7 + ((k >> 1) & 1); }
        else if (k < 15) { q = 2; plane = 6 + (k - 12); }
        else { q = -1; plane = 0; }
    }
}
// accumulator i (0 .. 8) of wave w -> tile t = a * 9 + plane of the epilogue exchange
__host__ __device__ inline int tw2_acc_tile(int w, int i) {
    if (w < 3) return i < 7 ? tw2_main_acc(w) * 9 + i : 2 * 9 + 2 * w + (i - 7);
    return i < 6 ? tw2_main_acc(i >> 1) * 9 + 7 + (i & 1) : 2 * 9 + 6 + (i - 6);
}

// offset of plane p inside a raw row of the patch buffer (planes (2 j, 2 j + 1) interleaved per column pair, plane 8 apart), lane part
__device__ __forceinline__ int tw2_poff(int p, int l31) { return p < 8 ? (p >> 1) * 64 + 2 * l31 + (p & 1) : 256 + l31; }

template <bool LEFT, bool STATS, int DBG = 0>
__global__ __launch_bounds__(256, 2) void cconv_tw2_kernel(const Tw2Args a) {
    constexpr int NT = 36, NACC = 9, NSLOT = 16;
    constexpr int NRAW = 7, CIK = 4, KS = 2;
    constexpr int RT = NRAW * 288;               // floats per channel in a patch buffer
    constexpr int NE = CIK * RT;
    constexpr int NITEM = CIK * NRAW * 16;       // 448 staging items per chunk: (channel, raw row, 2 column pairs)
    constexpr int NLD = 2;
    static_assert(NT * 4 * 64 <= 2 * NE, "the epilogue exchange fits the patch buffers");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    int jt, ft, ct;
    if (a.xcd_split) {                            // co tile ct always on the XCDs = ct (mod cotiles): see cgemm_tw.hip
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
    const int m0 = 2 * ft;                        // first OUTPUT row of the pair
    const int rbase = 2 * m0 - 2;                 // raw row r0

    const int Cin = a.Cin;
    const int nchunk = (Cin + CIK - 1) / CIK;

    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // ---- staging (cgemm_tw.hip): item = channel cl, raw row, column pairs 2 c8, 2 c8 + 1
    f32x4 v_r[NLD], v_i[NLD];
    float e_r[NLD], e_i[NLD];
    unsigned off_v[NLD], ldsoff[NLD];
    unsigned okmask[NLD];     // bits 0-4: window column valid; bit 5: row valid
    bool interior[NLD];
    auto item_cl = [&](int i) -> int { return (tid + i * 256) / (16 * NRAW); };
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * 256;
        const int c8 = e & 15;
        const int rl = (e >> 4) % NRAW, cl = e / (16 * NRAW);
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
        // the extra column is loaded at the vector slot's offset - 1 / + 4: one float outside the row at the buffer's edges, inside the
        // slack every planar buffer has in front and behind (masked)
        const int jcv = jc + 3 < a.Jp ? jc : 0;
        off_v[i] = okr ? (unsigned)((cl * a.Fin + f) * a.Jp + jcv) : 4u;
        if (jc + 3 >= a.Jp) m &= ~0x1fu;
        if (!(je >= 0 && je < a.Jp)) m &= LEFT ? ~1u : ~(1u << 4);
        okmask[i] = m;
        interior[i] = __builtin_amdgcn_ballot_w64((m & 0x3fu) == 0x3fu) == ~0ull;
        ldsoff[i] = (unsigned)((cl * NRAW + rl) * 288 + 4 * c8);
    }
    // the second item exists in waves 0 .. 2 only (448 items)
    auto item_exists = [&](int i) -> bool { return i == 0 || wave * 64 + 256 < NITEM; };
    auto stage_load = [&](int chunk, int i) {
        if (!item_exists(i)) return;
        const int ci0 = chunk * CIK;
        const float* br = a.x0 + (size_t)ci0 * a.Fin * a.Jp;
        const float* bi = br + (size_t)Cin * a.Fin * a.Jp;
        const bool dead = item_cl(i) >= Cin - ci0;
        const unsigned ov = dead ? 4u : off_v[i];
        v_r[i] = *(const f32x4*)(br + ov);
        v_i[i] = *(const f32x4*)(bi + ov);
        e_r[i] = (br + ov)[LEFT ? -1 : 4];
        e_i[i] = (bi + ov)[LEFT ? -1 : 4];
    };
    float fr[5], fi[5];
    auto stage_window = [&](int chunk, int i) {
        if (!item_exists(i)) return;
        const int cvalid = Cin - chunk * CIK;
        constexpr bool left = LEFT;
        if (interior[i] && cvalid >= CIK) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                fr[q] = left ? (q == 0 ? e_r[i] : v_r[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_r[i] : v_r[i][q == 4 ? 3 : q]);
                fi[q] = left ? (q == 0 ? e_i[i] : v_i[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_i[i] : v_i[i][q == 4 ? 3 : q]);
            }
            return;
        }
        unsigned m = okmask[i];
        if (item_cl(i) >= cvalid || !((m >> 5) & 1u)) m &= ~0x1fu;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const float xr = left ? (q == 0 ? e_r[i] : v_r[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_r[i] : v_r[i][q == 4 ? 3 : q]);
            const float xi = left ? (q == 0 ? e_i[i] : v_i[i][q == 0 ? 0 : q - 1]) : (q == 4 ? e_i[i] : v_i[i][q == 4 ? 3 : q]);
            const bool cok = (m >> q) & 1u;
            fr[q] = cok ? xr : 0.f;
            fi[q] = cok ? xi : 0.f;
        }
    };
    auto plane_val = [&](int gt, int pr) -> float {
        const int g = gt / 3, tau = gt - g * 3;
        const int q0 = 2 * pr;
        const float xa_ = g == 0 ? fr[q0] + fi[q0] : (g == 1 ? fr[q0] : fi[q0]);
        const float xb_ = g == 0 ? fr[q0 + 1] + fi[q0 + 1] : (g == 1 ? fr[q0 + 1] : fi[q0 + 1]);
        const float xd_ = g == 0 ? fr[q0 + 2] + fi[q0 + 2] : (g == 1 ? fr[q0 + 2] : fi[q0 + 2]);
        return tau == 0 ? xa_ - xb_ : (tau == 1 ? xb_ : xb_ - xd_);
    };
    auto stage_plane2 = [&](float* dst, int i, int j) {
        if (!item_exists(i)) return;
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

    // ---- weights: 16 slots per k-step as four 16-byte loads ([group][64 lanes][4 slots]).  ONE set of registers: group g of the next
    // k-step is fetched right after the four MFMAs that use group g of this one (two sets do not fit beside 144 accumulator, 32
    // operand and 20 staging registers)
    const float* wbase = a.wfrag + (((size_t)ct * a.UP) * 4 + wave) * 1024 + lane * 4;
    const int total_ks = nchunk * KS;
    float a_w[NSLOT];
    auto load_wg = [&](int g, int grp) {
        g = g < total_ks ? g : total_ks - 1;
        const f32x4 w4 = *(const f32x4*)(wbase + (size_t)g * 4096 + grp * 256);
#pragma unroll
        for (int k = 0; k < 4; ++k) a_w[grp * 4 + k] = w4[k];
    };

    // ---- operands.  Waves 0 .. 2: products x, y of the wave's main accumulator on planes 0 .. 6 and product 2 on the plane pair w:
    // six row offsets at run time, plane offsets constant.  Wave 3: every row constant.
    const int am = tw2_main_acc(wave < 3 ? wave : 0);
    const int qxm = tw2_qx(am), qym = tw2_qy(am);
    const int rxa = tw2_ra(qxm) * 288, rxb = tw2_rb(qxm) * 288, rya = tw2_ra(qym) * 288, ryb = tw2_rb(qym) * 288;
    const float cbx = tw2_cb(qxm), cby = tw2_cb(qym);
    const int r2a = tw2_ra(2) * 288 + wave * 64, r2b = tw2_rb(2) * 288 + wave * 64;     // (+ plane pair j = w of product 2)
    // operands in a rolling window of two groups of four slots: group g of a k-step is fetched while group g - 1 runs.
    // Role A (waves 0 .. 2): group g < 3 = product x planes (2 g, 2 g + 1) on slots 0, 2 and product y on slots 1, 3 (two 8-byte reads per
    // row each); group 3 = plane 6 of x, y (slots 0, 1) and product 2's plane pair (slots 2, 3).  Role B (wave 3): slots 4 g .. 4 g + 3 of
    // tw2_slot(3, .), one 4-byte read per row.
    float xa[2][4], xb[2][4];
    auto load_grp_a = [&](const float* base, int g, float (&oa)[4], float (&ob)[4]) {
        if (g < 3) {
            const float2 xa_ = *(const float2*)(base + rxa + g * 64 + 2 * l31), xb_ = *(const float2*)(base + rxb + g * 64 + 2 * l31);
            const float2 ya_ = *(const float2*)(base + rya + g * 64 + 2 * l31), yb_ = *(const float2*)(base + ryb + g * 64 + 2 * l31);
            oa[0] = xa_.x; oa[2] = xa_.y; ob[0] = xb_.x; ob[2] = xb_.y;
            oa[1] = ya_.x; oa[3] = ya_.y; ob[1] = yb_.x; ob[3] = yb_.y;
        } else {
            oa[0] = base[rxa + 3 * 64 + 2 * l31]; ob[0] = base[rxb + 3 * 64 + 2 * l31];
            oa[1] = base[rya + 3 * 64 + 2 * l31]; ob[1] = base[ryb + 3 * 64 + 2 * l31];
            const float2 pa_ = *(const float2*)(base + r2a + 2 * l31), pb_ = *(const float2*)(base + r2b + 2 * l31);
            oa[2] = pa_.x; oa[3] = pa_.y; ob[2] = pb_.x; ob[3] = pb_.y;
        }
    };
    auto load_grp_b = [&](const float* base, int g, float (&oa)[4], float (&ob)[4]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            int q, plane;
            tw2_slot(3, 4 * g + s4, q, plane);
            if (q < 0) continue;
            oa[s4] = base[tw2_ra(q) * 288 + tw2_poff(plane, l31)];
            ob[s4] = base[tw2_rb(q) * 288 + tw2_poff(plane, l31)];
        }
    };

#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_load(0, i);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) load_wg(0, grp);
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_store(smem, 0, i);
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_load(nchunk > 1 ? 1 : 0, i);
    __syncthreads();
    if (wave < 3)
        load_grp_a(smem + (size_t)half * RT, 0, xa[0], xb[0]);
    else
        load_grp_b(smem + (size_t)half * RT, 0, xa[0], xb[0]);

    // the main loop, once per wave role (the branch is outside the loop: two straight-line loops, each with the workgroup's barriers)
    auto run = [&](auto role) {
        constexpr bool ROLE_B = decltype(role)::value;
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            const float* P = smem + (chunk & 1) * NE;
            float* Pn = smem + ((chunk + 1) & 1) * NE;
            const int nxt = chunk + 1 < nchunk ? chunk + 1 : chunk, nxt2 = chunk + 2 < nchunk ? chunk + 2 : nchunk - 1;
#pragma unroll
            for (int ul = 0; ul < KS; ++ul) {
                const float* bnext = ul + 1 < KS ? P + (size_t)(2 * (ul + 1) + half) * RT : Pn + (size_t)half * RT;
                const bool staging = !(DBG & 1) && ul == 0;  // both items ride on k-step 0; the barrier sits in k-step 1
#pragma unroll
                for (int k = 0; k < (ROLE_B ? NSLOT - 1 : NSLOT); ++k) {
                    const int g = k >> 2, s4 = k & 3;
                    float cb;
                    int ai;
                    if (ROLE_B) {
                        int q, plane;
                        tw2_slot(3, k, q, plane);
                        cb = tw2_cb(q);
                        ai = k < 12 ? k >> 1 : 6 + (k - 12);
                    } else {
                        cb = k < 14 ? ((k & 1) ? cby : cbx) : -1.f;
                        ai = k < 14 ? k >> 1 : 7 + (k - 14);
                    }
                    if (s4 == 0) {
                        // the next group's operands (of this k-step, or group 0 of the next one: after the barrier in the last k-step)
                        const float* src = g < 3 ? P + (size_t)(2 * ul + half) * RT : bnext;
                        if (!(ul == KS - 1 && g == 3)) {
                            if (ROLE_B) load_grp_b(src, (g + 1) & 3, xa[(g + 1) & 1], xb[(g + 1) & 1]);
                            else load_grp_a(src, (g + 1) & 3, xa[(g + 1) & 1], xb[(g + 1) & 1]);
                        }
                    }
                    const float b = xa[g & 1][s4] + cb * xb[g & 1][s4];
                    acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[k], b, acc[ai], 0, 0, 0);
                    if (ul == KS - 1 && k == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        __syncthreads();
                    }
                    if (ul == KS - 1 && k == 12) {
                        // (last k-step: group 0 of the next chunk comes from the other buffer, complete since the barrier above)
                        if (ROLE_B) load_grp_b(bnext, 0, xa[0], xb[0]);
                        else load_grp_a(bnext, 0, xa[0], xb[0]);
                    }
                    // weights of the next k-step, group by group
                    if (((k & 3) == 3 || (ROLE_B && k == NSLOT - 2)) && !(DBG & 2)) load_wg(chunk * KS + ul + 1, k >> 2);
                    if (staging) {
                        // item 0: window at slot 0, plane pairs at 1 .. 4, plane 8 at 5, reload at 5; item 1 (waves 0 .. 2): slots 8 .. 13
                        if (k == 0) stage_window(nxt, 0);
                        if (k >= 1 && k <= 5) stage_plane2(Pn, 0, k - 1);
                        if (k == 5) stage_load(nxt2, 0);
                        if (!ROLE_B) {
                            if (k == 8) stage_window(nxt, 1);
                            if (k >= 9 && k <= 13) stage_plane2(Pn, 1, k - 9);
                            if (k == 13) stage_load(nxt2, 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    if (wave < 3)
        run(std::false_type{});
    else
        run(std::true_type{});
    __syncthreads();                                          // all patch reads done: the buffers become the exchange area

    // ------------------------------------------------------------------ epilogue (cgemm_tw.hip's, 36 tiles)
    const float slope = a.slope ? *a.slope : 1.0f;
    const bool has_act = a.slope != nullptr;
    float* E = smem;
    const int jA = j0 + 2 * l31;
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
        for (int k = 0; k < NACC; ++k) {
            const int t = tw2_acc_tile(wave, k);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) E[(t * 4 + rr) * 64 + lane] = acc[k][4 * s + rr];
        }
        __syncthreads();
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = E[(t * 4 + wave) * 64 + lane];
        float pr[4][2], pi[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
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
            const int fo = m0 + rt;
            if (fo >= a.Fout) continue;
            float yr[2], yi[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float re = rt == 0 ? pr[0][q] + pr[1][q] + pr[2][q] : pr[1][q] - pr[2][q] - pr[3][q];
                const float im = rt == 0 ? pi[0][q] + pi[1][q] + pi[2][q] : pi[1][q] - pi[2][q] - pi[3][q];
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

// cgemm_wino's conv fragments [ct][unit = ci * 3 + g][8 slots q][lane = h * 32 + co] -> [ct][pair u][wave][group][lane = parity * 32 +
// co][4 slots] with the time-transformed taps (tau 0: W_h0, tau 1: W_h0 + W_h1, tau 2: W_h1) of the slot's (product, plane)
__global__ void pack_cconv_tw2_kernel(const float* __restrict__ wino, int cotiles, int UN, int UP, float* __restrict__ out) {
    const long long n = (long long)cotiles * UP * 4096;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(idx & 3), ln = (int)((idx >> 2) & 63), grp = (int)((idx >> 8) & 3), w = (int)((idx >> 10) & 3);
        const long long t_ = idx >> 12;
        const int u = (int)(t_ % UP), ct = (int)(t_ / UP);
        int q, plane;
        tw2_slot(w, grp * 4 + kk, q, plane);
        float val = 0.f;
        const int ci = 2 * u + (ln >> 5), co = ln & 31;
        if (q >= 0 && ci * 3 < UN) {
            const int g = plane / 3, tau = plane % 3;
            const float* src = wino + ((((size_t)ct * UN + (size_t)ci * 3 + g) * 8 + q)) * 64;
            const float w0 = src[co], w1 = src[32 + co];
            val = tau == 0 ? w0 : (tau == 1 ? w0 + w1 : w1);
        }
        out[idx] = val;
    }
}

template <bool LEFT, bool STATS, int DBG>
int launch_tw2(const Tw2Args& a, hipStream_t st) {
    constexpr size_t smem = 2 * 4 * 7 * 288 * sizeof(float);
    static_assert(smem * 2 <= 160 * 1024, "the patch buffers of two workgroups must fit the 160 KB of LDS");
    Tw2Args b = a;
    b.jtiles = (a.J + 63) / 64;
    b.ftiles = (a.Fout + 1) / 2;
    static const int xsplit = [] { const char* e = getenv("IDV_TW_XCD_SPLIT"); return e ? atoi(e) : 1; }();
    b.xcd_split = (xsplit && (b.cotiles == 2 || b.cotiles == 4 || b.cotiles == 8)) ? 1 : 0;
    long long nblk = (long long)((b.jtiles + 7) / 8) * 8 * b.ftiles * b.cotiles;
    if (b.xcd_split) {
        const int G = 8 / b.cotiles;
        nblk = (long long)((b.jtiles + G - 1) / G) * b.ftiles * 8;
    }
    if (nblk > 0x7fffffffLL) return IDV_EINVAL;
    auto k = cconv_tw2_kernel<LEFT, STATS, DBG>;
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

// 1 if idv_cconv2d_tw_fwd serves the layer: a conv cgemm_gauss serves (one source) with at least one full tile of 32 complex output
// channels, at least 64 input channels (measured at B = 64: enc1, 32 -> 64 channels, 2.63 -> 2.86 ms: eight K chunks do not amortise the
// epilogue exchange; enc2-5 19.6 -> 18.7 ms) and at least two output rows.  IDV_TW2_MIN_COUT / IDV_TW2_MIN_CIN (experiments).
extern "C" int idv_cconv_tw2_supported(int Cin, int Cout, int Fin) {
    static const int min_cout = [] { const char* e = getenv("IDV_TW2_MIN_COUT"); return e ? atoi(e) : 32; }();
    static const int min_cin = [] { const char* e = getenv("IDV_TW2_MIN_CIN"); return e ? atoi(e) : 64; }();
    if (Cout < min_cout || Cin < min_cin || (Fin - 1) / 2 + 1 < 2) return 0;
    return idv_cconv_gauss_supported(Cin, 0, Cout);
}

extern "C" long long idv_cconv_tw2_wfrag_floats(int Cout, int cin_used) {
    const long long cotiles = (Cout + 31) / 32, cpad = (cin_used + TW2_PACK_CI - 1) / TW2_PACK_CI * TW2_PACK_CI;
    return cotiles * (cpad / 2) * 4096;
}

// wino_frag: idv_pack_cconv_wino(transposed = 0) of the same weights; tw_frag: idv_cconv_tw2_wfrag_floats floats
extern "C" int idv_pack_cconv_tw2(const float* wino_frag, int Cout, int cin_used, float* tw_frag, void* stream) {
    if (!wino_frag || !tw_frag || Cout <= 0 || cin_used <= 0) return IDV_EINVAL;
    const int cotiles = (Cout + 31) / 32;
    const int cpad = (cin_used + TW2_PACK_CI - 1) / TW2_PACK_CI * TW2_PACK_CI;
    const long long n = idv_cconv_tw2_wfrag_floats(Cout, cin_used);
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_cconv_tw2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, wino_frag, cotiles, cpad * 3, cpad / 2,
                       tw_frag);
    return idv_launch_status();
}

// idv_cconv2d_wino_fwd (transposed = 0, one source, no addend) on the time-Winograd conv kernel: same result up to the rounding of the
// transforms.  wfrag from idv_pack_cconv_tw2, epi / has_fold from idv_pack_cconv_gauss.  16-byte aligned source, Jp % 4 == 0.
// Reference: model/complex_progress.py:8-36 (+ :161-209 and pvae_module.py:58 for the epilogue).
extern "C" int idv_cconv2d_tw_fwd(const float* x0, int Cin, const float* wfrag, const float* epi, int has_fold, const float* prelu_slope,
                                  float* out, double* stats, double* stats_work, int stats_rep, int tshift, int Cout, int Fin, int B, int Tp,
                                  int Jp, int t_valid_out, void* stream) {
    if (!x0 || !wfrag || !epi || !out || Cin <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (stats && stats_work && (stats_rep < 2 || (stats_rep & (stats_rep - 1)))) return IDV_EINVAL;
    if (tshift != 0 && tshift != -1) return IDV_EINVAL;
    if (!idv_cconv_tw2_supported(Cin, Cout, Fin)) return IDV_EINVAL;
    if ((Jp & 3) || (reinterpret_cast<uintptr_t>(x0) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return IDV_EINVAL;
    Tw2Args a{};
    a.x0 = x0; a.Cin = Cin;
    a.Fin = Fin; a.Fout = (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp;
    a.wfrag = wfrag; a.UP = (Cin + TW2_PACK_CI - 1) / TW2_PACK_CI * TW2_PACK_CI / 2;
    a.epi = epi; a.has_fold = has_fold; a.slope = prelu_slope; a.out = out;
    a.Cout = Cout; a.cotiles = (Cout + 31) / 32;
    a.tshift = tshift; a.t_valid = t_valid_out;
    if (Jp < a.J) return IDV_EINVAL;
    if ((long long)4 * Fin * (long long)Jp >= 0xffffffffLL) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    a.stats = stats;
    if (stats && stats_work) { a.stats = stats_work; a.stats_rep = stats_rep; }
    int rc;
#ifdef IDV_TW_EXPERIMENTS
    // timing experiments (WRONG results by construction; compiled in only with -DIDV_TW_EXPERIMENTS): IDV_TW_DBG = the kernel's DBG bits
    static const int dbg = [] { const char* e = getenv("IDV_TW_DBG"); return e ? atoi(e) : 0; }();
    if (!stats && dbg == 1) return tshift ? launch_tw2<true, false, 1>(a, st) : launch_tw2<false, false, 1>(a, st);
    if (!stats && dbg == 2) return tshift ? launch_tw2<true, false, 2>(a, st) : launch_tw2<false, false, 2>(a, st);
    if (!stats && dbg == 3) return tshift ? launch_tw2<true, false, 3>(a, st) : launch_tw2<false, false, 3>(a, st);
#endif
    if (stats)
        rc = tshift ? launch_tw2<true, true, 0>(a, st) : launch_tw2<false, true, 0>(a, st);
    else
        rc = tshift ? launch_tw2<true, false, 0>(a, st) : launch_tw2<false, false, 0>(a, st);
    if (rc || !(stats && stats_work)) return rc;
    return idv_launch_stats_collapse(stats_work, stats_rep, Cout * 5, stats, st);
}
