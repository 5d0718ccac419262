// cgemm_gauss: the complex Conv2d / ConvTranspose2d contraction with THREE real products per complex product.
//
// The reference computes a complex convolution as four real convolutions (model/complex_progress.py:32-36, :275-279):
//     re = conv_re(x_r) - conv_im(x_i),   im = conv_re(x_i) + conv_im(x_r).
// cgemm_kernel (cgemm.hpp) runs exactly those 4 * Cin * Cout * 10 MACs per output position as one real contraction.  This
// kernel uses the Gauss / Karatsuba identity:
//     t1 = W_r * (x_r + x_i),   t2 = (W_i - W_r) * x_r,   t3 = (W_r + W_i) * x_i      (* = the real 5x2 convolution)
//     re = t1 - t3,              im = t1 + t2
// i.e. 3 * Cin * Cout * 10 MACs: 25 % fewer fp32 MFMAs, which is what bounds the fp32 path.  Per output tile (32 complex
// output channels x 32 columns) a wave keeps THREE accumulator tiles (t1, t2, t3) and combines them register-locally in the
// epilogue; the s = x_r + x_i plane is formed once per staged element at the LDS write; the three weight planes
// (W_r, W_i - W_r, W_r + W_i) are built by idv_pack_cconv_gauss.  Because the three products are only equivalent to the
// complex product for a complex-linear weight, eval-mode ComplexBatchNormal (a real 2x2 map per channel) is NOT folded into
// the weights here: the epilogue applies y = Z (conv + b) + s from a per-channel table, then PReLU (pvae_module.py:58,82).
//
// Tiling: v_mfma_f32_32x32x2_f32, the two k of one instruction are the two time taps (lanes 0-31 column j, lanes 32-63
// column j+1 of the same LDS row) as in cgemm.hpp.  A wave owns 32 complex output channels x ROWS output frequency rows x
// JC_W column tiles x 3 products (12 accumulator tiles: transposed conv 2 rows x 2 column tiles, conv 1 x 4, 3 x 1 (9) or
// 5 x 1 (15)); WM x WN waves per workgroup.  Input patch per K chunk: CIK complex channels x 3 planes (s, x_r, x_i) x FR
// rows x JT+8 columns, staged global -> registers -> LDS, double buffered, one barrier per chunk.  Weights stream from L2 in
// fragment order, a chunk ahead.
#include <cstdlib>
#include "cgemm.hpp"
#include "../../include/idccrn_hip.h"

namespace {

struct GaussArgs {
    const float* x0;      // planar [2][C0][Fin][Jp]
    const float* x1;      // optional skip source, planar [2][C1][Fin][Jp1]
    int C0, C1;
    int Fin, Fout;
    int J, Jp, Tp;
    int Jp1, x1_div;
    const float* wfrag;   // [cotiles][KS = Cin_pad * 15][64]: k-step (ci, p, kf), lane = kt * 32 + co % 32
    int KS;               // k-steps per co tile as packed (Cin rounded up to the pack granularity)
    const float* epi;     // [cotiles * 32][8]: Zrr, Zri, Zir, Zii, sr, si, 0, 0   (no fold: 1, 0, 0, 1, b_re - b_im, b_re + b_im)
    int has_fold;
    const float* slope;
    float* out;           // planar [2][Cout][Fout][Jp]
    int Cout, cotiles;
    int tshift, t_valid;
    double* stats;        // train mode: [Cout][5] sums (r, i, rr, ii, ri) of conv + bias, or nullptr
    int stats_rep;        // > 1: stats holds that many replicas [rep][Cout][5] (power of two), one chosen per workgroup
    const float* add;     // optional addend, planar [2][Cout][Fout][add_Jp]: added to the contraction before bias / BN / PReLU;
    int add_div, add_Jp;  //   column j of the output reads the addend's utterance b / add_div (a conv of the skip connection
                          //   computed once per utterance and shared by its num_samples latent draws)
    int jtiles, ftiles, mblocks, map_ft;
};

template <int MODE, int FO_T>
struct GaussGeom {
    static constexpr int FR = (MODE == IDV_CONV) ? 2 * FO_T + 3 : FO_T + 2;
    static constexpr int ROWS = (MODE == IDV_TCONV) ? 2 * FO_T : FO_T;
};

// NBUF: LDS patch buffers.  2: the next chunk is written during the last unit, one barrier at the chunk's end, the first
// fragments of the next chunk are read behind it (their LDS latency + the store's vector work sit between two MFMAs).
// 3 (where 3 patches fit the 160 KB): the next chunk is written in the MIDDLE of the chunk, the barrier stands in front of
// the last unit and the next chunk's first fragments are read during the last unit's MFMAs -- nothing waits at the chunk
// boundary.  Three buffers because a wave that has passed barrier(c-1) may write chunk c+1's patch while a slower wave still
// reads chunk c-1's in its last unit: they must be different buffers.
template <int MODE, int WM, int WN, int FO_T, int JC_W, int CIK, bool STATS, bool VEC, int NBUF, int OCC = 1>
__global__ __launch_bounds__(WM* WN * 64, OCC) void cgemm_gauss_kernel(const GaussArgs a) {
    using G = GaussGeom<MODE, FO_T>;
    constexpr int NT = WM * WN * 64;
    constexpr int KF = 5, FR = G::FR, ROWS = G::ROWS;
    constexpr int JT = 32 * JC_W * WN;
    constexpr int PS = JT + 8;                    // patch row: the 16-byte aligned span j0-4 .. j0+JT+3
    constexpr int COL0 = 4;
    constexpr int PS4 = PS / 4;
    constexpr int NS = CIK * FR * PS4;            // float4 slots (of ONE of the three planes) per chunk
    constexpr int NLD = (NS + NT - 1) / NT;
    constexpr int NE = CIK * 3 * FR * PS;         // patch floats per chunk
    constexpr int UNITS = CIK * 3;                // pipeline units per chunk: (channel, plane)
    constexpr int KSC = UNITS * KF;               // MFMA k-steps per chunk
    static_assert(NLD <= 8, "ok-mask holds 8 slots x 4 bits");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // block order as in cgemm_kernel: the MB co-tile blocks that share one input patch sit on consecutive slots of ONE XCD
    const int MB = a.mblocks, FTn = a.ftiles;
    const int bid = blockIdx.x;
    int jt, ft, mblk;
    if (a.map_ft) {
        const int per = 8 * MB * FTn;
        const int sg = bid / per, rem = bid - sg * per;
        const int v = rem >> 3;
        jt = sg * 8 + (rem & 7);
        ft = v / MB;
        mblk = v - ft * MB;
        if (jt >= a.jtiles) return;
    } else {
        const int grp = bid / (8 * MB), rem = bid - grp * (8 * MB);
        const int tile = grp * 8 + (rem & 7);
        mblk = rem >> 3;
        if (tile >= a.jtiles * FTn) return;
        jt = tile / FTn;
        ft = tile - jt * FTn;
    }
    const int j0 = jt * JT;
    const int ct = mblk * WM + wm;                           // this wave's tile of 32 complex output channels
    const bool ct_ok = ct < a.cotiles;                       // (a wave without a tile still stages and meets the barriers)
    const int fo0 = ft * FO_T;
    const int fbase = (MODE == IDV_CONV) ? 2 * fo0 - 2 : fo0 - 1;

    const int Cin = a.C0 + a.C1;
    const int nchunk = (Cin + CIK - 1) / CIK;                // wfrag is zero padded to whole chunks (of the pack granularity)
    const int KS = a.KS;

    f32x16 acc[ROWS][JC_W][3];
#pragma unroll
    for (int rt = 0; rt < ROWS; ++rt)
#pragma unroll
        for (int jc = 0; jc < JC_W; ++jc)
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][jc][p3][r] = 0.f;

    // ---- staging.  A slot = 4 consecutive columns of one (channel, patch row); the thread loads the real and the imaginary
    // values of its slot and writes three float4 (s, x_r, x_i).  Offsets inside a chunk do not depend on the chunk.
    f32x4 sr[NLD], si[NLD];
    unsigned voff[NLD];               // VEC: float offset of the slot relative to the chunk's first real plane
    unsigned eoff0[VEC ? 1 : NLD][4]; // scalar path: per element, for the x0 source ...
    unsigned eoff1[VEC ? 1 : NLD][4]; // ... and for the x1 source (other row stride, repeated utterances)
    unsigned okbits = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * NT;
        const int row = e / PS4, c4 = e - row * PS4;
        const int cil = row / FR, fr = row - cil * FR;
        const int fi = fbase + fr;
        const int jv = j0 - 4 + 4 * c4;
        const bool rowok = (e < NS) && (fi >= 0) && (fi < a.Fin);
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (rowok && jv + q >= 0 && jv + q < a.J) bits |= 1u << q;
        okbits |= bits << (4 * i);
        if (VEC) {
            voff[i] = bits ? (unsigned)((cil * a.Fin + fi) * a.Jp + jv) : 0u;      // a dead slot loads mapped memory (offset 0)
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = (bits >> q) & 1u;
                const int j = jv + q;
                int js = j;
                if (a.x1_div > 1 && ok) {
                    const int b = j / a.Tp;
                    js = j - (b - b / a.x1_div) * a.Tp;
                }
                eoff0[i][q] = ok ? (unsigned)((cil * a.Fin + fi) * a.Jp + j) : 0u;
                eoff1[i][q] = ok ? (unsigned)((cil * a.Fin + fi) * a.Jp1 + js) : 0u;
            }
        }
    }
    auto stage_load = [&](int chunk) {
        const int ci0 = chunk * CIK;
        const bool from0 = ci0 < a.C0;
        const float* br = from0 ? a.x0 + (size_t)ci0 * a.Fin * a.Jp : a.x1 + (size_t)(ci0 - a.C0) * a.Fin * a.Jp1;
        const float* bi = from0 ? br + (size_t)a.C0 * a.Fin * a.Jp : br + (size_t)a.C1 * a.Fin * a.Jp1;
        const int cvalid = (from0 ? a.C0 : Cin) - ci0;       // channels of this chunk that exist in its source (ragged end)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool dead = (tid + i * NT) / (FR * PS4) >= cvalid;       // reads offset 0 (mapped memory), zeroed at the LDS write
            if (VEC) {
                const unsigned o = dead ? 0u : voff[i];
                sr[i] = *(const f32x4*)(br + o);
                si[i] = *(const f32x4*)(bi + o);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned o = dead ? 0u : (from0 ? eoff0[i][q] : eoff1[i][q]);
                    sr[i][q] = br[o];
                    si[i][q] = bi[o];
                }
            }
        }
    };
    auto stage_store = [&](float* dst, int chunk) {
        const int ci0s = chunk * CIK;
        const int cvalid = (ci0s < a.C0 ? a.C0 : Cin) - ci0s;      // channels of this chunk that exist (ragged last chunk)
        // every staged register is "used" here, outside the `e < NS` branch below: otherwise the threads that skip a slot
        // never wait for its load, and hipcc protects the register's next writer with a near-complete vmcnt drain at the loop
        // header -- which would expose the latency of the weight loads issued just before the back edge
#pragma unroll
        for (int i = 0; i < NLD; ++i) asm volatile("" ::"v"(sr[i]), "v"(si[i]));
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            const int row = e / PS4, c4 = e - row * PS4;
            const int cil = row / FR, fr = row - cil * FR;
            unsigned bits = (okbits >> (4 * i)) & 15u;
            if (cil >= cvalid) bits = 0u;
            f32x4 vr = sr[i], vi = si[i], vs;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vr[q] = (bits >> q) & 1u ? vr[q] : 0.f;
                vi[q] = (bits >> q) & 1u ? vi[q] : 0.f;
                vs[q] = vr[q] + vi[q];
            }
            if (e < NS) {
                float* d = dst + ((cil * 3) * FR + fr) * PS + 4 * c4;
                *(f32x4*)d = vs;
                *(f32x4*)(d + FR * PS) = vr;
                *(f32x4*)(d + 2 * FR * PS) = vi;
            }
        }
    };

    // ---- weight fragments: one coalesced 256 B load per k-step.  ONE register set: the five fragments of a unit are
    // re-loaded with their next use after the unit's MFMAs have consumed them, i.e. (almost) a whole chunk ahead of their use
    // and five loads per unit instead of a burst of KSC loads per chunk (the MFMA issue of a wave queues behind its own
    // vector-memory issue: a 30-load burst cost the first version ~5 % of every chunk)
    const float* wbase = a.wfrag + (size_t)(ct_ok ? ct : 0) * KS * 64 + lane;
    float a_w[KSC];

    // ---- activation fragments of one unit (one plane of one channel): FR rows x JC_W column tiles, one unit ahead
    const int bcol = wn * (JC_W * 32) + (lane & 31) + (lane >> 5) + COL0 + a.tshift;
    auto load_b = [&](const float* P, int u, float (&dst)[FR][JC_W]) {
#pragma unroll
        for (int fr = 0; fr < FR; ++fr)
#pragma unroll
            for (int jc = 0; jc < JC_W; ++jc) dst[fr][jc] = P[(u * FR + fr) * PS + bcol + jc * 32];
    };

    stage_load(0);
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) a_w[ks] = wbase[(size_t)ks * 64];
    stage_store(smem, 0);
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) asm volatile("" : "+v"(a_w[ks]));       // retire the prologue loads before the loop
    __syncthreads();

    constexpr int UMID = UNITS / 2;
    float b_cur[FR][JC_W], b_nxt[FR][JC_W];
    int ibuf = 0;                                                             // buffer of the current chunk
    if (NBUF == 3) load_b(smem, 0, b_cur);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const float* P = smem + ibuf * NE;
        const int inext = (ibuf + 1 == NBUF) ? 0 : ibuf + 1;
        float* Pn = smem + inext * NE;
        const int nxt = (chunk + 1 < nchunk) ? chunk + 1 : chunk;            // branch-free: the last chunk re-fetches itself
        const float* wnx = wbase + (size_t)nxt * KSC * 64;
        const float* wcu = wbase + (size_t)chunk * KSC * 64;
        if (NBUF == 2) load_b(P, 0, b_cur);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int p3 = u % 3;
            auto mfma_tap = [&](int kf) {
                const int ks = u * KF + kf;
#pragma unroll
                for (int rt = 0; rt < ROWS; ++rt) {
                    int fr;
                    if (MODE == IDV_CONV) {
                        fr = 2 * rt + kf;
                    } else {
                        if ((rt & 1) != (kf & 1)) continue;       // even rows take even taps, odd rows odd taps
                        fr = (rt >> 1) + 2 - (kf >> 1);
                    }
#pragma unroll
                    for (int jc = 0; jc < JC_W; ++jc)
                        acc[rt][jc][p3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[ks], b_cur[fr][jc], acc[rt][jc][p3], 0, 0, 0);
                }
            };
            // the unit's first tap goes BEFORE the next unit's LDS reads: at the loop header the fragments carried over the
            // back edge then need no wait (hipcc waits with lgkmcnt(0) there: behind freshly issued reads that is their latency)
            mfma_tap(0);
            __builtin_amdgcn_sched_barrier(0);
            if (NBUF == 3 && u == UNITS - 1) __syncthreads();                // every wave has written the next chunk's patch
            if (u + 1 < UNITS)
                load_b(P, u + 1, b_nxt);
            else if (NBUF == 3)
                load_b(Pn, 0, b_nxt);                                        // the next chunk's first unit
            __builtin_amdgcn_sched_barrier(0);
            if (u == 0) stage_load(nxt);
            if (u == (NBUF == 3 ? UMID : UNITS - 1)) stage_store(Pn, nxt);
#pragma unroll
            for (int kf = 1; kf < KF; ++kf) mfma_tap(kf);
            __builtin_amdgcn_sched_barrier(0);
            // the PREVIOUS unit's fragments are consumed: fetch its next use into the same registers (unit 0 fetches the last
            // unit of THIS chunk, every other unit the next chunk's: rotated by one so that nothing loaded right before the
            // loop's back edge is live across it -- the two values hipcc copies at the loop header are then old loads).
            // The five loads stay together behind the unit's MFMAs: one load behind each tap's MFMAs measured 2 % SLOWER on the
            // same box (659 -> 646.5 utt/s), five interruptions of the MFMA stream cost more than one
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                if (u == 0)
                    a_w[(UNITS - 1) * KF + kf] = wcu[(size_t)((UNITS - 1) * KF + kf) * 64];
                else
                    a_w[(u - 1) * KF + kf] = wnx[(size_t)((u - 1) * KF + kf) * 64];
            }
            if (u + 1 < UNITS || NBUF == 3) {
#pragma unroll
                for (int fr = 0; fr < FR; ++fr)
#pragma unroll
                    for (int jc = 0; jc < JC_W; ++jc) b_cur[fr][jc] = b_nxt[fr][jc];
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
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const f32x4 e0 = *(const f32x4*)(a.epi + (size_t)co * 8);           // table is allocated for every row of every tile
        const float e4 = a.epi[(size_t)co * 8 + 4], e5 = a.epi[(size_t)co * 8 + 5];
        const bool cok = co < a.Cout;
        float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jc = 0; jc < JC_W; ++jc) {
            const int j = j0 + wn * (JC_W * 32) + jc * 32 + l31;
            const int bj = j / a.Tp, tp = j - bj * a.Tp;
            const bool keep = (tp >= 1) && (tp <= a.t_valid);
            const bool inb = j < a.J;
            const int ja = j - (bj - bj / a.add_div) * a.Tp;          // column of utterance b / add_div in the addend
#pragma unroll
            for (int rt = 0; rt < ROWS; ++rt) {
                const int fo = (MODE == IDV_TCONV) ? 2 * (fo0 + (rt >> 1)) + (rt & 1) : fo0 + rt;
                if (fo >= a.Fout) continue;
                const float t1 = acc[rt][jc][0][r], t2 = acc[rt][jc][1][r], t3 = acc[rt][jc][2][r];
                float re = t1 - t3, im = t1 + t2;
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
        }
        if (STATS) {
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                float t = st[s];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                if (l31 == 0 && cok)
                    atomicAdd(&a.stats[((size_t)(a.stats_rep > 1 ? (blockIdx.x & (a.stats_rep - 1)) : 0) * a.Cout + co) * 5 + s], (double)t);
            }
        }
    }
}

// fragment element (ct, ks, lane): lane = kt * 32 + col supplies W_p[co = ct*32 + col][ci][kf][kt], ks = (ci * 3 + p) * 5 + kf
__global__ void pack_cconv_gauss_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im,
                                        const float* __restrict__ b_re, const float* __restrict__ b_im,
                                        const float* __restrict__ fold, int Cout, int Cin_total, int Cin_used, int transposed,
                                        int conj, int KS, int cotiles, float* __restrict__ wfrag, float* __restrict__ epi) {
    const long long n = (long long)cotiles * KS * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long long t = idx >> 6;
        const int ks = (int)(t % KS), ct = (int)(t / KS);
        const int h = lane >> 5, co = ct * 32 + (lane & 31);
        const int kf = ks % 5, cp = ks / 5;
        const int ci = cp / 3, p3 = cp % 3;
        float v = 0.f;
        if (co < Cout && ci < Cin_used) {
            // conv: taps (x[t-1], x[t]) pair with kt = (0, 1); transposed conv: out[t] = W[..,0] x[t] + W[..,1] x[t-1]
            const int kt = transposed ? 1 - h : h;
            const size_t off = transposed ? (((size_t)ci * Cout + co) * 5 + kf) * 2 + kt
                                          : (((size_t)co * Cin_total + ci) * 5 + kf) * 2 + kt;
            const float wr = w_re[off], wi = conj ? -w_im[off] : w_im[off];
            v = p3 == 0 ? wr : (p3 == 1 ? wi - wr : wr + wi);
        }
        wfrag[idx] = v;
    }
    const int nb = cotiles * 32;
    for (int co = blockIdx.x * blockDim.x + threadIdx.x; co < nb; co += gridDim.x * blockDim.x) {
        float e[8] = {1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f};
        if (co < Cout) {
            const float top = b_re ? b_re[co] - b_im[co] : 0.f, bot = b_re ? b_re[co] + b_im[co] : 0.f;
            if (fold) {
                const float* z = fold + (size_t)co * 6;
                e[0] = z[0]; e[1] = z[1]; e[2] = z[2]; e[3] = z[3];
                e[4] = z[0] * top + z[1] * bot + z[4];
                e[5] = z[2] * top + z[3] * bot + z[5];
            } else {
                e[4] = top;
                e[5] = bot;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) epi[(size_t)co * 8 + q] = e[q];
    }
}

template <int MODE, int WM, int WN, int FO_T, int JC_W, int CIK, bool STATS, bool VEC, int OCC = 1>
int launch_gauss_v(const GaussArgs& a, hipStream_t st) {
    using G = GaussGeom<MODE, FO_T>;
    constexpr int JT = 32 * JC_W * WN;
    constexpr int NE = CIK * 3 * G::FR * (JT + 8);
    constexpr int NBUF = (3 * NE * sizeof(float) * OCC <= 156 * 1024) ? 3 : 2;     // three patch buffers where the LDS holds them
    constexpr size_t smem = NBUF * NE * sizeof(float);
    static_assert(smem * OCC <= 160 * 1024, "the patch buffers of OCC workgroups must fit the 160 KB of LDS");
    const int rows = (MODE == IDV_TCONV) ? a.Fin : a.Fout;
    GaussArgs b = a;
    b.jtiles = (a.J + JT - 1) / JT;
    b.ftiles = (rows + FO_T - 1) / FO_T;
    b.mblocks = (a.cotiles + WM - 1) / WM;
    static const bool map_ft = [] { const char* e = getenv("IDV_MAP_FT"); return !e || e[0] != '0'; }();
    b.map_ft = (map_ft && MODE == IDV_TCONV) ? 1 : 0;
    const long long tiles = (long long)b.jtiles * b.ftiles;
    const long long nblk = b.map_ft ? (long long)((b.jtiles + 7) / 8) * 8 * b.ftiles * b.mblocks : ((tiles + 7) / 8) * 8 * b.mblocks;
    if (nblk > 0x7fffffffLL) return IDV_EINVAL;
    auto k = cgemm_gauss_kernel<MODE, WM, WN, FO_T, JC_W, CIK, STATS, VEC, NBUF, OCC>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(WM * WN * 64), smem, st, b);
    return idv_launch_status();
}

template <int MODE, int WM, int WN, int FO_T, int JC_W, int CIK, bool STATS, int OCC = 1>
int launch_gauss(const GaussArgs& a, hipStream_t st) {
    // vector staging: 16-byte aligned rows and the same column mapping for both sources
    const bool vec = (a.Jp % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x0) & 15) == 0) &&
                     (a.C1 == 0 || (a.x1_div == 1 && a.Jp1 == a.Jp && (reinterpret_cast<uintptr_t>(a.x1) & 15) == 0));
    if (vec) return launch_gauss_v<MODE, WM, WN, FO_T, JC_W, CIK, STATS, true, OCC>(a, st);
    return launch_gauss_v<MODE, WM, WN, FO_T, JC_W, CIK, STATS, false, OCC>(a, st);
}

inline int waste(int n, int t) { return ((n + t - 1) / t) * t - n; }

constexpr int CIK = 4;        // complex input channels per K chunk (wfrag / supported() granularity); the kernels' own chunk
constexpr int CIK5 = 2;       // the 5-row conv tile (15 accumulator tiles = 240 registers) and the 1-row x 4-column-tile conv
                              // (13 / 5 patch rows of 72 / 264 .. 520 columns per channel) stage 2 channels per chunk

// Two workgroups per CU (6 accumulator tiles per wave, < 256 registers, accumulators in VGPRs) so that one workgroup's prologue
// (first patch, weight ring) and epilogue overlap the other's MFMAs; per layer, tests/tools/gauss_layers_probe.py, B = 64:
//   * conv: one output row x two column tiles per wave beats the 12 / 9 / 15-tile one-workgroup forms on EVERY encoder layer
//     (enc1 3.44 -> 3.24 ms, enc2 6.32 -> 5.78, enc3 6.13 -> 5.84, enc4 6.40 -> 6.23, enc5 7.03 -> 6.94): no frequency-tile
//     waste, and the short-K layers (32 .. 128 input channels) gain most.  IDV_GAUSS_CCFG=0 restores the old forms,
//     IDV_GAUSS_CCFG=N limits the new one to <= N input channels;
//   * transposed conv: one column tile per wave wins at <= 128 input channels (dec4 6.78 -> 6.13 ms) and loses at 256 / 512
//     (dec3 11.75 -> 11.89 ms; all wide layers 659 -> 653 utt/s), where the 12-tile form amortises its patch reads over two
//     column tiles.  IDV_GAUSS_OCC2_MAXC moves the boundary.
inline int occ2_max_cin() {
    static const int v = [] { const char* e = getenv("IDV_GAUSS_OCC2_MAXC"); return e ? atoi(e) : 128; }();
    return v;
}
inline bool tconv_wm4() {
    static const bool v = [] { const char* e = getenv("IDV_GAUSS_TWM"); return !(e && atoi(e) == 2); }();
    return v;
}
inline bool conv_cik2() {
    static const bool v = [] { const char* e = getenv("IDV_GAUSS_CCIK"); return e && atoi(e) == 2; }();
    return v;
}
inline int conv_occ2_max_cin() {
    static const int v = [] { const char* e = getenv("IDV_GAUSS_CCFG"); return e ? atoi(e) : (1 << 30); }();
    return v;
}

// configuration id: 3 MODE WM WN FO_T JC_W OCC as decimal digits (leading 3 = the three-product kernel)
int gauss_config(int transposed, int Cin, int Cout, int rows) {
    const int wide = Cout > 32;                       // two co tiles per workgroup where the layer has them
    if (transposed) {
        const bool occ2 = Cin <= occ2_max_cin();
        // four co tiles x ONE column group per workgroup where the layer has four co tiles (a 72-column patch instead of 136
        // per workgroup: dec0 13.63 -> 13.16 ms, dec1 12.61 -> 11.81, dec2 11.96 -> 11.52 at B = 64; IDV_GAUSS_TWM=2 restores 2 x 2)
        if (!occ2 && Cout >= 128 && tconv_wm4()) return 3141121;
        return wide ? (occ2 ? 3122112 : 3122121) : (occ2 ? 3114112 : 3114121);
    }
    if (Cin <= conv_occ2_max_cin()) return wide ? 3022122 : 3014122;
    int fo = 5;
    if (waste(rows, 3) < waste(rows, fo)) fo = 3;
    if (waste(rows, 1) < waste(rows, fo)) fo = 1;
    const int jc = fo == 1 ? 4 : 1;
    return (300000 + (wide ? 2200 : 1400) + fo * 10 + jc) * 10 + 1;
}

template <bool STATS>
int launch_cfg(const GaussArgs& a, int transposed, hipStream_t st) {
    const int rows = transposed ? a.Fin : a.Fout;
    // experiments (IDV_GAUSS_TCFG): 1 = one co tile x four column groups per workgroup, 2 = two workgroups per CU whatever the K
    static const int tcfg = [] { const char* e = getenv("IDV_GAUSS_TCFG"); return e ? atoi(e) : 0; }();
    if (transposed && a.Cout > 32 && !STATS) {
        if (tcfg == 1) return launch_gauss<IDV_TCONV, 1, 4, 1, 2, CIK, STATS>(a, st);
        if (tcfg == 2) return launch_gauss<IDV_TCONV, 2, 2, 1, 1, CIK, STATS, 2>(a, st);
    }
    switch (gauss_config(transposed, a.C0 + a.C1, a.Cout, rows)) {
        // (eight channels per K chunk -- half the barriers -- change nothing here: three patch buffers already hide them)
        case 3141121: return launch_gauss<IDV_TCONV, 4, 1, 1, 2, CIK, STATS>(a, st);
        case 3122121: return launch_gauss<IDV_TCONV, 2, 2, 1, 2, CIK, STATS>(a, st);
        case 3122112: return launch_gauss<IDV_TCONV, 2, 2, 1, 1, CIK, STATS, 2>(a, st);
        case 3114121: return launch_gauss<IDV_TCONV, 1, 4, 1, 2, CIK, STATS>(a, st);
        case 3114112: return launch_gauss<IDV_TCONV, 1, 4, 1, 1, CIK, STATS, 2>(a, st);
        // (four channels per K chunk on two patch buffers beat two channels on three: 28.1 -> 27.1 ms over enc1-5, B = 64;
        //  IDV_GAUSS_CCIK=2 restores the latter)
        case 3022122: return conv_cik2() ? launch_gauss<IDV_CONV, 2, 2, 1, 2, CIK5, STATS, 2>(a, st)
                                         : launch_gauss<IDV_CONV, 2, 2, 1, 2, CIK, STATS, 2>(a, st);
        // (one co tile x four column groups: a 264-column patch -- four channels per chunk would not leave room for two workgroups)
        case 3014122: return launch_gauss<IDV_CONV, 1, 4, 1, 2, CIK5, STATS, 2>(a, st);
        case 3022511: return launch_gauss<IDV_CONV, 2, 2, 5, 1, CIK5, STATS>(a, st);
        case 3022311: return launch_gauss<IDV_CONV, 2, 2, 3, 1, CIK, STATS>(a, st);
        case 3022141: return launch_gauss<IDV_CONV, 2, 2, 1, 4, CIK5, STATS>(a, st);
        case 3014511: return launch_gauss<IDV_CONV, 1, 4, 5, 1, CIK5, STATS>(a, st);
        case 3014311: return launch_gauss<IDV_CONV, 1, 4, 3, 1, CIK, STATS>(a, st);
        case 3014141: return launch_gauss<IDV_CONV, 1, 4, 1, 4, CIK5, STATS>(a, st);
        default: return IDV_EINVAL;
    }
}

const bool USE_GAUSS = [] { const char* e = getenv("IDV_GAUSS"); return !e || e[0] != '0'; }();

}  // namespace

// 1 if the three-product kernel serves this layer shape (else idv_pack_cconv / idv_cconv2d_fwd): at least two complex input
// channels, more than one output channel (the one-channel ends are HBM-bound and have kernels of their own) and, with a
// second source, a first source of whole K chunks.  IDV_GAUSS=0 turns it off.
extern "C" int idv_cconv_gauss_supported(int C0, int C1, int Cout) {
    if (!USE_GAUSS || C0 + C1 < 2 || C0 < 1 || Cout < 2 || C1 < 0) return 0;
    return C1 == 0 || (C0 % CIK == 0);               // a K chunk never straddles the two sources
}

// floats of the fragment buffer / rows of the epilogue table (x 8 floats) for idv_pack_cconv_gauss
extern "C" long long idv_cconv_gauss_wfrag_floats(int Cout, int cin_used) {
    const long long cotiles = (Cout + 31) / 32, cpad = (cin_used + CIK - 1) / CIK * CIK;
    return cotiles * cpad * 15 * 64;
}
extern "C" int idv_cconv_gauss_epi_rows(int Cout) { return (Cout + 31) / 32 * 32; }

extern "C" int idv_cconv_gauss_config(int transposed, int Cin, int Cout, int Fin) {
    const int rows = transposed ? Fin : (Fin - 1) / 2 + 1;
    return gauss_config(transposed, Cin, Cout, rows);
}

// Pack ComplexConv2d / ComplexConvTranspose2d weights (layouts as idv_pack_cconv) into the three Gauss planes
// (W_r, W_i - W_r, W_r + W_i) in MFMA fragment order + the per-channel epilogue table [rows][8] =
// (Zrr, Zri, Zir, Zii, (Z b + s)_r, (Z b + s)_i, 0, 0) with b = (b_re - b_im, b_re + b_im); fold == NULL: Z = 1, s = 0.
// conj != 0 negates W_i (the adjoint / data-gradient operator, together with the swapped channel roles the caller passes).
extern "C" int idv_pack_cconv_gauss(const float* w_re, const float* w_im, const float* b_re, const float* b_im, const float* fold,
                                    int Cout, int Cin_total, int Cin_used, int transposed, int conj, float* wfrag, float* epi,
                                    void* stream) {
    if (!w_re || !w_im || !wfrag || !epi || Cout <= 0 || Cin_used <= 0 || Cin_used > Cin_total) return IDV_EINVAL;
    if ((b_re == nullptr) != (b_im == nullptr)) return IDV_EINVAL;
    const int cotiles = (Cout + 31) / 32;
    const int KS = (Cin_used + CIK - 1) / CIK * CIK * 15;
    const long long n = (long long)cotiles * KS * 64;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_cconv_gauss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_re, w_im, b_re, b_im, fold, Cout,
                       Cin_total, Cin_used, transposed, conj, KS, cotiles, wfrag, epi);
    return idv_launch_status();
}

// idv_cconv2d_fwd on the three-product kernel: same arguments, with (wfrag, epi) from idv_pack_cconv_gauss in place of
// (wfrag, bias); has_fold != 0 applies the table's 2x2 map (eval-mode ComplexBatchNormal) before PReLU.  addend (or NULL):
// planar [2][Cout][Fout][addend_Jp] with B / addend_div utterances, added to the contraction before bias / BN / PReLU --
// the convolution is linear in its input channels, so the skip half of a decoder block whose skips repeat per latent sample
// (pvae_module.py:2563-2567) is computed once per utterance and added to every sample's latent half.  Reference:
// model/complex_progress.py:16-22, :32-36, :244-250, :275-279 (+ :161-209 and pvae_module.py:58,82 for the epilogue).
extern "C" int idv_cconv2d_gauss_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, int x1_div, const float* wfrag,
                                     const float* epi, int has_fold, const float* prelu_slope, float* out, double* stats,
                                     double* stats_work, int stats_rep, int transposed, int tshift, int Cout, int Fin, int B, int Tp,
                                     int Jp, int t_valid_out, const float* addend, int addend_div, int addend_Jp, void* stream) {
    if (!x0 || !wfrag || !epi || !out || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (stats && stats_work && (stats_rep < 2 || (stats_rep & (stats_rep - 1)))) return IDV_EINVAL;
    if (addend && (addend_div < 1 || B % addend_div || addend_Jp < (B / addend_div) * Tp)) return IDV_EINVAL;
    if (C1 > 0 && (!x1 || x1_div < 1)) return IDV_EINVAL;
    if (tshift != 0 && tshift != -1) return IDV_EINVAL;
    if (!idv_cconv_gauss_supported(C0, C1, Cout)) return IDV_EINVAL;
    GaussArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.Fin = Fin;
    a.Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = C1 > 0 ? Jp1 : Jp; a.x1_div = x1_div < 1 ? 1 : x1_div;
    a.wfrag = wfrag; a.KS = (C0 + C1 + CIK - 1) / CIK * CIK * 15; a.epi = epi; a.has_fold = has_fold; a.slope = prelu_slope; a.out = out;
    a.Cout = Cout; a.cotiles = (Cout + 31) / 32;
    a.tshift = tshift; a.t_valid = t_valid_out; a.stats = stats;
    a.add = addend; a.add_div = addend ? addend_div : 1; a.add_Jp = addend_Jp;
    if (Jp < a.J) return IDV_EINVAL;
    // chunk-relative offsets are 32-bit: CIK channels of one source must stay below 2^32 floats (they do: 2 x 257 x Jp)
    if ((long long)CIK * Fin * (long long)(Jp > a.Jp1 ? Jp : a.Jp1) >= 0xffffffffLL) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!stats) return launch_cfg<false>(a, transposed, st);
    if (!stats_work) return launch_cfg<true>(a, transposed, st);
    a.stats = stats_work; a.stats_rep = stats_rep;       // replicated sums, folded into `stats` afterwards (common.hpp)
    const int rc = launch_cfg<true>(a, transposed, st);
    return rc ? rc : idv_launch_stats_collapse(stats_work, stats_rep, Cout * 5, stats, st);
}
