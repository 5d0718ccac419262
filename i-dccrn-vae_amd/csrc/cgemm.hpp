// cgemm: the one MFMA contraction kernel behind every dense op on the hot path.
//
//   out[m][fo][j] = bias[m] + sum_{cc,kf,kt} W'[m][cc][kf][kt] * x[cc][fi(fo,kf)][j + kt + tshift]
//
// A complex Conv2d / ConvTranspose2d of the reference (model/complex_progress.py:8-36,
// :222-279: four real convolutions + stack) is ONE such real contraction over the planar
// channel index cc = 2*ci + ri with the block weight [[Wr,-Wi],[Wi,Wr]] (rows m = 2*co + ro),
// built once by pack.hip; eval-mode ComplexBatchNormal (complex_progress.py:161-209) is
// folded into W'/bias there, PReLU (pvae_module.py:58,82) runs in the epilogue, and in train
// mode the epilogue emits the five per-channel moments the batch statistics need.
// The same kernel in PW mode (1x1, k-pairs = plane pairs) is the LSTM input projection,
// ComplexDense (complex_progress.py:77-89), and the DFT / inverse-DFT of STFT / ISTFT.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32).  The two k of one instruction are the two time
// taps (kt = 0,1) of one (cc, kf): lanes 0-31 read patch column j, lanes 32-63 column j+1 of
// the SAME LDS row, so the im2col expansion costs nothing and is bank-conflict free.
// Weights stream from L2 in pre-swizzled fragment order (one coalesced 256 B load per
// fragment, no LDS); the input patch (CCK planes x FR rows x JT+2 columns) is staged
// global -> registers -> LDS, double buffered, one barrier per K chunk.
#pragma once
#include "common.hpp"

#ifndef IDV_WAVES_PER_SIMD
#define IDV_WAVES_PER_SIMD 1
#endif

enum { IDV_CONV = 0, IDV_TCONV = 1, IDV_PW = 2 };

struct CgemmArgs {
    const float* x0;      // first source, planar [2][C0][Fin][Jp] (PW: K planes of stride Jp)
    const float* x1;      // optional second source (skip connection), planar [2][C1][Fin][Jp1]
    int C0, C1;           // complex channels taken from x0 / x1 (PW: C0 = K/2, C1 = 0)
    int Fin, Fout;        // input / output rows (PW: 1 / 1)
    int J, Jp, Tp;        // columns, row stride, columns per utterance
    int Jp1, x1_div;      // x1 row stride; x1 column j maps to utterance (b / x1_div)
    const float* wfrag;   // [Mtiles][KS][64] fragment-ordered weights
    const float* bias;    // [Mtiles*32]
    const float* slope;   // PReLU slope (device scalar) or nullptr
    float* out;           // planar [M planes][Fout][Jp] or, SWAP, row-major [pos][ldo]
    int M;                // valid rows
    int Mtiles;           // allocated 32-row tiles in wfrag (multiple of the block's tiles)
    int cplx_rows;        // 1: row m = 2*co + ro -> plane ro*Cout + co ; 0: plane = m
    int Cout;
    int tshift;           // -1: taps (x[t-1], x[t])   0: taps (x[t], x[t+1])
    int t_valid;          // outputs at tp in [1, t_valid] are kept, everything else is zero
    double* stats;        // train mode: [Cout][5] sums (r, i, rr, ii, ri) or nullptr
    int stats_rep;        // > 1: stats holds that many replicas [rep][Cout][5] (power of two), one chosen per workgroup
    int ldo;              // SWAP: row stride of out; rows are ordered (tp-1)*B + b
    int nB;               // SWAP: utterances (B)
    int jtiles, ftiles, mblocks;   // grid decomposition (filled by the launcher)
    int map_ft;                    // 1: an XCD takes ALL frequency tiles of its column blocks (they share halo rows in L2)
    // split-bf16 image sources / destination (cgemm_bf16.hip, cgemm_c1.hip); x0 / x1 then point at image data
    long long lo_off0, lo_off1;    // hi -> lo plane distance of the x0 / x1 image, in 16-byte slots
    void* out_img;                 // optional image destination (besides or instead of `out`)
    long long out_lo_off;          // its hi -> lo distance, in bf16 elements
};
// 16-byte slot, relative to the start of an image plane, that every producer of an image zeroes (both planes):
// consumers read it for frequency rows outside [0, Fin)
#define IDV_IMG_ZSLOT (-8)

typedef __bf16 idv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned idv_pack_bf16(float a, float b) {      // round-to-nearest-even pair
    idv_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

template <int MODE, int FO_T>
struct CgemmGeom {
    static constexpr int KF = (MODE == IDV_PW) ? 1 : 5;
    static constexpr int FR = (MODE == IDV_CONV) ? 2 * FO_T + 3 : (MODE == IDV_TCONV ? FO_T + 2 : 1);
    static constexpr int ROWS = (MODE == IDV_TCONV) ? 2 * FO_T : FO_T;   // output row tiles per j chunk
};

// WM x WN waves; each wave owns MT_W row tiles (32 rows) and ROWS*JC_W column tiles (32 cols).
template <int MODE, int WM, int WN, int MT_W, int FO_T, int JC_W, int CCK, bool SWAP, bool STATS, bool VEC>
__global__ __launch_bounds__(WM* WN * 64, IDV_WAVES_PER_SIMD) void cgemm_kernel(const CgemmArgs a) {
    using G = CgemmGeom<MODE, FO_T>;
    constexpr int NT = WM * WN * 64;
    constexpr int KF = G::KF, FR = G::FR, ROWS = G::ROWS;
    constexpr int JT = 32 * JC_W * WN;
    // patch row: scalar staging keeps columns j0-1 .. j0+JT; vector staging keeps the 16-byte aligned
    // span j0-4 .. j0+JT+3 so that every row is whole float4s
    constexpr int PS = VEC ? JT + 8 : JT + 2;
    constexpr int COL0 = VEC ? 4 : 1;             // patch column of j0
    constexpr int NE = CCK * FR * PS;             // patch elements per K chunk
    constexpr int NLD = VEC ? 1 : (NE + NT - 1) / NT;
    constexpr int PS4 = PS / 4;
    constexpr int NV = CCK * FR * PS4;            // float4 slots per chunk
    constexpr int NLD4 = VEC ? (NV + NT - 1) / NT : 1;
    static_assert(!VEC || NLD4 <= 8, "ok-mask holds 8 slots x 4 bits");
    constexpr int NCOL = ROWS * JC_W;
    constexpr int KSC = (MODE == IDV_PW) ? CCK / 2 : CCK * KF;   // MFMA k-steps per chunk
    static_assert(MODE != IDV_PW || (CCK % 2 == 0), "PW chunks are plane pairs");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // Block order: the MB row-tile blocks that share one input patch get consecutive slots on ONE XCD
    // (ids equal mod 8 share an XCD under round-robin placement), so the patch is fetched from HBM once
    // and re-read from that XCD's L2 (speed only, never correctness).
    const int MB = a.mblocks, FTn = a.ftiles;
    const int bid = blockIdx.x;
    int jt, ft, mblk;
    if (a.map_ft) {
        // super group = 8 column blocks (one per XCD) x FTn frequency tiles x MB row-tile blocks; on an XCD the MB blocks
        // of a frequency tile are consecutive, then the next frequency tile of the SAME column block
        const int per = 8 * MB * FTn;
        const int sg = bid / per, rem = bid - sg * per;
        const int v = rem >> 3;
        jt = sg * 8 + (rem & 7);
        ft = v / MB;
        mblk = v - ft * MB;
        if (jt >= a.jtiles) return;                         // padding blocks of the last super group
    } else {
        const int grp = bid / (8 * MB), rem = bid - grp * (8 * MB);
        const int tile = grp * 8 + (rem & 7);
        mblk = rem >> 3;
        if (tile >= a.jtiles * FTn) return;                 // padding blocks of the last group
        jt = tile / FTn;
        ft = tile - jt * FTn;
    }
    const int j0 = jt * JT;
    const int mt0 = (mblk * WM + wm) * MT_W;                // first 32-row tile of this wave
    const int fo0 = ft * FO_T;                              // CONV: first output row; TCONV: first m
    const int fbase = (MODE == IDV_CONV) ? 2 * fo0 - 2 : (MODE == IDV_TCONV ? fo0 - 1 : 0);

    const int CC = 2 * (a.C0 + a.C1);
    const int nchunk = (CC + CCK - 1) / CCK;                 // wfrag is zero padded to whole chunks
    const int KS = (MODE == IDV_PW) ? nchunk * CCK / 2 : nchunk * CCK * KF;   // k-steps per row tile

    f32x16 acc[MT_W][NCOL];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // ---- staging: global -> registers (issued before the chunk's MFMAs) -> LDS (after them) ----------
    // Scalar path (x1_div > 1, i.e. repeated skip connections): one element per load, address rebuilt per chunk.
    // Vector path: one aligned float4 per slot; a slot's offset inside its chunk and its validity bits do not
    // depend on the chunk, so the per-chunk work is one add per slot and the loads use a scalar chunk base.
    float stg[NLD];
    f32x4 stg4[NLD4];
    unsigned voff[NLD4];        // float offset of the slot relative to the chunk's first plane (real part)
    unsigned okbits = 0;        // 4 validity bits per slot
    unsigned ribits = 0;        // slot belongs to an imaginary plane
    if (VEC) {
#pragma unroll
        for (int i = 0; i < NLD4; ++i) {
            const int e = tid + i * NT;
            const int row = e / PS4, c4 = e - row * PS4;
            const int ccl = row / FR, fr = row - ccl * FR;
            const int fi = fbase + fr;
            const int jv = j0 - 4 + 4 * c4;
            const bool rowok = (e < NV) && (fi >= 0) && (fi < a.Fin);
            unsigned bits = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (rowok && jv + q >= 0 && jv + q < a.J) bits |= 1u << q;
            okbits |= bits << (4 * i);
            // a slot with no valid element loads from offset 0 of its chunk (always mapped memory)
            if (MODE == IDV_PW) {
                voff[i] = bits ? (unsigned)(ccl * a.Jp + jv) : 0u;
            } else {
                if (ccl & 1) ribits |= 1u << i;
                voff[i] = bits ? (unsigned)(((ccl >> 1) * a.Fin + fi) * a.Jp + jv) : 0u;
            }
        }
    }
    auto stage_load = [&](int chunk) {
        if (VEC) {
            const float* base;
            unsigned ristride;                           // distance from a real plane to its imaginary plane
            if (MODE == IDV_PW) {
                base = a.x0 + (size_t)chunk * CCK * a.Jp;
                ristride = 0;
            } else {
                const int ci0 = chunk * (CCK / 2);       // first complex channel of the chunk
                if (ci0 < a.C0) {
                    base = a.x0 + (size_t)ci0 * a.Fin * a.Jp;
                    ristride = (unsigned)a.C0 * a.Fin * a.Jp;
                } else {
                    base = a.x1 + (size_t)(ci0 - a.C0) * a.Fin * a.Jp1;
                    ristride = (unsigned)a.C1 * a.Fin * a.Jp1;
                }
            }
#pragma unroll
            for (int i = 0; i < NLD4; ++i) {
                unsigned o = voff[i];
                if (MODE == IDV_PW) {
                    // ragged last chunk (K not a multiple of CCK): planes beyond K are read as plane 0 and masked
                    const int e = tid + i * NT;
                    const int ccl = e / PS4;
                    if (chunk * CCK + ccl >= CC) o = 0u;
                } else {
                    o += ((ribits >> i) & 1u) ? (((okbits >> (4 * i)) & 15u) ? ristride : 0u) : 0u;
                }
                stg4[i] = *(const f32x4*)(base + o);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            const int row = e / PS, col = e - row * PS;
            const int ccl = row / FR, fr = row - ccl * FR;
            const int cc = chunk * CCK + ccl;
            const int fi = fbase + fr;
            const int j = j0 - 1 + col;
            const bool ok = (e < NE) && (cc < CC) && (fi >= 0) && (fi < a.Fin) && (j >= 0) && (j < a.J);
            const float* src = a.x0;                       // masked lanes: the always-zero guard element
            if (MODE == IDV_PW) {
                if (ok) src = a.x0 + (size_t)cc * a.Jp + j;
            } else {
                const int ci = cc >> 1, ri = cc & 1;
                if (ci < a.C0) {
                    if (ok) src = a.x0 + ((size_t)(ri * a.C0 + ci) * a.Fin + fi) * a.Jp + j;
                } else {
                    int js = j;
                    if (a.x1_div > 1) {
                        const int b = j / a.Tp;
                        js = j - (b - b / a.x1_div) * a.Tp;
                    }
                    if (ok) src = a.x1 + ((size_t)(ri * a.C1 + (ci - a.C0)) * a.Fin + fi) * a.Jp1 + js;
                }
            }
            stg[i] = *src;      // masked lanes read x0[0]: the tp == 0 guard column, zero by the layout invariant
        }
    };
    auto stage_store = [&](float* dst, int chunk) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NLD4; ++i) {
                const int e = tid + i * NT;
                unsigned bits = (okbits >> (4 * i)) & 15u;
                if (MODE == IDV_PW) {
                    if (chunk * CCK + e / PS4 >= CC) bits = 0u;
                }
                f32x4 v = stg4[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (bits >> q) & 1u ? v[q] : 0.f;
                if (e < NV) *(f32x4*)(dst + 4 * e) = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            if (e < NE) dst[e] = stg[i];
        }
    };

    // ---- weight fragments: one coalesced 256 B load per (k-step, row tile), a whole chunk ahead -------
    const float* wbase = a.wfrag + (size_t)mt0 * KS * 64 + lane;
    float a_cur[KSC][MT_W], a_nxt[KSC][MT_W];
    auto load_a = [&](int chunk, float (&dst)[KSC][MT_W]) {
        const float* wch = wbase + (size_t)chunk * KSC * 64;
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
            for (int i = 0; i < MT_W; ++i) dst[ks][i] = wch[((size_t)i * KS + ks) * 64];
    };

    // ---- activation fragments: all patch rows one plane (PW: one plane pair) needs, one unit ahead -----
    constexpr int UNITS = (MODE == IDV_PW) ? CCK / 2 : CCK;      // pipeline units per chunk
    const int bcol = wn * (JC_W * 32) + (lane & 31) + ((MODE == IDV_PW) ? COL0 : (lane >> 5) + COL0 + a.tshift);
    auto load_b = [&](const float* P, int u, float (&dst)[FR][JC_W]) {
#pragma unroll
        for (int fr = 0; fr < FR; ++fr)
#pragma unroll
            for (int jc = 0; jc < JC_W; ++jc) {
                if (MODE == IDV_PW)
                    dst[fr][jc] = P[(2 * u + (lane >> 5)) * PS + bcol + jc * 32];
                else
                    dst[fr][jc] = P[(u * FR + fr) * PS + bcol + jc * 32];
            }
    };

    stage_load(0);
    load_a(0, a_cur);
    stage_store(smem, 0);
    // Retire the prologue's weight loads before the loop: otherwise the loop header inherits them as
    // "pending" and hipcc puts counted vmcnt waits in front of the loop's MFMAs, which in steady state
    // wait for the NEXT chunk's prefetch (a false dependency worth ~1 us per chunk).
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
        for (int i = 0; i < MT_W; ++i) asm volatile("" : "+v"(a_cur[ks][i]));
    __syncthreads();

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const float* P = smem + (chunk & 1) * NE;
        // branch-free body: the last chunk re-fetches itself (unused) instead of branching, so each unit
        // is one scheduling region and the loads/stores below really interleave with the MFMAs
        const int nxt = (chunk + 1 < nchunk) ? chunk + 1 : chunk;
        float b_cur[FR][JC_W], b_nxt[FR][JC_W];
        load_b(P, 0, b_cur);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            if (u + 1 < UNITS) load_b(P, u + 1, b_nxt);
            __builtin_amdgcn_sched_barrier(0);        // keep the next unit's LDS reads ahead of this unit's MFMAs
            // the next chunk's global loads ride in the shadow of unit 0's MFMAs, its LDS writes in the
            // shadow of the last unit's (the other LDS buffer is idle: every wave passed the last barrier)
            if (u == 0) {
                stage_load(nxt);
                load_a(nxt, a_nxt);
            }
            if (u == UNITS - 1) stage_store(smem + ((chunk + 1) & 1) * NE, nxt);
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                const int ks = (MODE == IDV_PW) ? u : u * KF + kf;
#pragma unroll
                for (int rt = 0; rt < ROWS; ++rt) {
                    int fr;
                    if (MODE == IDV_CONV) {
                        fr = 2 * rt + kf;
                    } else if (MODE == IDV_TCONV) {
                        if ((rt & 1) != (kf & 1)) continue;       // even rows take even taps, odd rows odd taps
                        fr = (rt >> 1) + 2 - (kf >> 1);
                    } else {
                        fr = 0;
                    }
#pragma unroll
                    for (int jc = 0; jc < JC_W; ++jc)
#pragma unroll
                        for (int i = 0; i < MT_W; ++i) {
                            if (SWAP)
                                acc[i][rt * JC_W + jc] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[fr][jc], a_cur[ks][i], acc[i][rt * JC_W + jc], 0, 0, 0);
                            else
                                acc[i][rt * JC_W + jc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ks][i], b_cur[fr][jc], acc[i][rt * JC_W + jc], 0, 0, 0);
                        }
                }
            }
            if (u + 1 < UNITS) {
#pragma unroll
                for (int fr = 0; fr < FR; ++fr)
#pragma unroll
                    for (int jc = 0; jc < JC_W; ++jc) b_cur[fr][jc] = b_nxt[fr][jc];
            }
        }
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
            for (int i = 0; i < MT_W; ++i) a_cur[ks][i] = a_nxt[ks][i];
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    if (a.out_img && blockIdx.x == 0 && tid < 2) {           // the destination image's zero slot (hi and lo plane)
        unsigned short* z = (unsigned short*)a.out_img + IDV_IMG_ZSLOT * 8 + (tid ? a.out_lo_off : 0);
        *(uint4*)z = make_uint4(0u, 0u, 0u, 0u);
    }
    const float slope = a.slope ? *a.slope : 1.0f;
    const bool has_act = a.slope != nullptr;
    const int half = lane >> 5, l31 = lane & 31;

    if (!SWAP) {
        // rows of the tile live in registers, column j = lane & 31
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            const int mt = mt0 + i;
            float bia[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                bia[r] = a.bias[m];                 // bias is allocated (zero padded) for every row of every tile
            }
            // train-mode moments: registers (2q, 2q+1) hold (real, imag) of one channel
            float st[STATS ? 8 : 1][5];
            if (STATS) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int s = 0; s < 5; ++s) st[q][s] = 0.f;
            }
#pragma unroll
            for (int jc = 0; jc < JC_W; ++jc) {
                const int j = j0 + wn * (JC_W * 32) + jc * 32 + l31;
                const int tp = j % a.Tp;
                const bool keep = (tp >= 1) && (tp <= a.t_valid);
                const bool inb = j < a.J;
#pragma unroll
                for (int rt = 0; rt < ROWS; ++rt) {
                    const int fo = (MODE == IDV_TCONV) ? 2 * (fo0 + (rt >> 1)) + (rt & 1) : fo0 + rt;
                    if (fo >= a.Fout) continue;
                    const f32x16 v = acc[i][rt * JC_W + jc];
                    float y[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float t = v[r] + bia[r];
                        if (has_act) t = t >= 0.f ? t : slope * t;
                        y[r] = keep ? t : 0.f;
                        if (a.out && m < a.M && inb) {
                            const int plane = a.cplx_rows ? ((m & 1) * a.Cout + (m >> 1)) : m;
                            a.out[((size_t)plane * a.Fout + fo) * a.Jp + j] = y[r];
                        }
                    }
                    if (a.out_img && inb) {
                        // split-bf16 image for the bf16x3 consumers (layout: idccrn_hip.h "split image")
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (mt * 32 + 8 * g >= a.M) continue;
                            unsigned hw[2], lw[2];
#pragma unroll
                            for (int w = 0; w < 2; ++w) {
                                const float x0 = y[4 * g + 2 * w], x1 = y[4 * g + 2 * w + 1];
                                const unsigned u0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u;
                                const unsigned u1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
                                hw[w] = (u0 >> 16) | u1;
                                lw[w] = idv_pack_bf16(x0 - __builtin_bit_cast(float, u0), x1 - __builtin_bit_cast(float, u1));
                            }
                            unsigned short* d = (unsigned short*)a.out_img +
                                                (((size_t)(mt * 4 + g) * a.Fout + fo) * a.Jp + j) * 8 + 4 * half;
                            *(uint2*)d = make_uint2(hw[0], hw[1]);
                            *(uint2*)(d + a.out_lo_off) = make_uint2(lw[0], lw[1]);
                        }
                    }
                    if (STATS) {
                        if (inb && keep) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const float yr = y[2 * q], yi = y[2 * q + 1];
                                st[q][0] += yr;
                                st[q][1] += yi;
                                st[q][2] += yr * yr;
                                st[q][3] += yi * yi;
                                st[q][4] += yr * yi;
                            }
                        }
                    }
                }
            }
            if (STATS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int m = mt * 32 + ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * half;
#pragma unroll
                    for (int s = 0; s < 5; ++s) {
                        float t = st[q][s];
#pragma unroll
                        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                        if (l31 == 0 && m < a.M)
                            atomicAdd(&a.stats[((size_t)(a.stats_rep > 1 ? (blockIdx.x & (a.stats_rep - 1)) : 0) * a.Cout + (m >> 1)) * 5 + s],
                                      (double)t);
                    }
                }
            }
        }
    } else {
        // SWAP: rows of the tile (registers) are columns j, lane & 31 is the output row m
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            const int m = (mt0 + i) * 32 + l31;
            const float bm = (m < a.M) ? a.bias[m] : 0.f;
#pragma unroll
            for (int jc = 0; jc < JC_W; ++jc) {
                const f32x16 v = acc[i][jc];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + wn * (JC_W * 32) + jc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (j >= a.J || m >= a.M) continue;
                    const int b = j / a.Tp, tp = j - b * a.Tp;
                    if (tp < 1 || tp > a.t_valid) continue;
                    float y = v[r] + bm;
                    if (has_act) y = y >= 0.f ? y : slope * y;
                    a.out[((size_t)(tp - 1) * a.nB + b) * a.ldo + m] = y;
                }
            }
        }
    }
}
