// Split-precision variant of the complex conv / transposed-conv contraction: every fp32 operand is
// split as x = hi + lo (two bf16), and  w*x ~= w_hi*x_hi + w_hi*x_lo + w_lo*x_hi  is accumulated in fp32 on
// v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate; 3 MFMAs per 16-deep k block, dropped term 2^-16 relative).
// Same math, block-weight formulation, planar-J global layout, epilogue and C-ABI as cgemm.hpp; what changes
// is the K order inside the workgroup: bf16 MFMA wants 8 consecutive k per lane, so the LDS patch is
// channels-last ([freq row][column][16 planar channels], hi and lo images) and the 16-deep k block of one
// MFMA is 16 planar channels of ONE (freq tap, time tap).  Staging transposes on the fly: a thread loads
// 8 channel planes x 4 columns (8 aligned float4), converts, and writes 4 + 4 ds_write_b128.
// Weights: pre-split bf16 fragments [row tile][chunk][tap][hi|lo][lane] x 16 B, streamed from L2 through a
// 5-tap register ring (4 taps of prefetch distance).
#include <cstdlib>
#include "bf16_common.hpp"
#include "../../include/idccrn_hip.h"
#include <stdlib.h>

#ifndef IDV_AD
#define IDV_AD 3
#endif
#ifndef IDV_IMG_DMA
#define IDV_IMG_DMA 1
#endif

namespace {

// IMGIN: the sources are split-bf16 images (see idccrn_hip.h "split image"): staging is a plain 16-byte copy
// global -> LDS in linear slot order, no conversion and no transposition.
template <int K>
struct IntC { static constexpr int value = K; };

// AD: depth of the weight-fragment ring in phases (one phase = the 5 frequency taps of one time tap of one chunk);
// fragments are fetched AD-1 phases ahead of their use.
template <int MODE, int WM, int WN, int FO_T, int JC_W, bool STATS, int MT_W = 1, bool IMGIN = false, int AD = 2>
__global__ __launch_bounds__(WM* WN * 64, ((FO_T == 3 && JC_W == 1 && MT_W == 1) || (IMGIN && AD == 2 && JC_W == 1 && MT_W == 1)) ? 2 : 1) void cgemm_bf16_kernel(const CgemmArgs a) {
    using G = CgemmGeom<MODE, FO_T>;
    constexpr int NT = WM * WN * 64;
    constexpr int KF = 5, FR = G::FR, ROWS = G::ROWS;
    constexpr int JT = 32 * JC_W * WN;
    constexpr int PS = JT + 8;                       // columns j0-4 .. j0+JT+3
    constexpr int PS4 = PS / 4;
    constexpr int NCOL = ROWS * JC_W;
    constexpr int FRP_ = ((((FR * ((JT + 8) / 4) * 2) + NT - 1) / NT) * NT + ((JT + 8) / 4) * 2 - 1) / (((JT + 8) / 4) * 2);
    constexpr int IMG = FRP_ * PS * 16;              // bf16 elements of one image (hi or lo), padded rows included
    constexpr int BUF = 2 * IMG;                     // hi + lo
    constexpr int NTASK = FR * PS4 * 2;              // (row, 4-column group, channel octet)
    constexpr int NLD = (NTASK + NT - 1) / NT;
    // the LDS image is padded to FRP rows so that all NLD*NT staging slots are distinct in-bounds tasks: the
    // staging code then has no divergent branch (one basic block -> it interleaves with the MFMAs)
    constexpr int FRP = (NLD * NT + PS4 * 2 - 1) / (PS4 * 2);
    static_assert(FRP == FRP_, "padded row count");
    // image-source staging: one task = one 16-byte slot (fr, octet, column) of the LDS image, in linear order
    constexpr int NSLOT = FR * 2 * PS;
    constexpr int NLDI = (NSLOT + NT - 1) / NT;
    static_assert(NLDI * NT * 8 <= IMG, "padded image holds every staging slot");

    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int MB = a.mblocks, FTn = a.ftiles;
    const int bid = blockIdx.x;
    int jt, ft, mblk;
    if (a.map_ft) {                                          // all frequency tiles of a column block on one XCD (cgemm.hpp)
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
    const int mt0 = (mblk * WM + wm) * MT_W;                 // this wave's first 32-row tile
    const int fo0 = ft * FO_T;
    const int fbase = (MODE == IDV_CONV) ? 2 * fo0 - 2 : fo0 - 1;

    const int CC = 2 * (a.C0 + a.C1);
    const int nchunk = CC / 16;

    f32x16 acc[MT_W][NCOL];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // ---- staging ------------------------------------------------------------------------------------
    u32x4 sth[IMGIN ? NLDI : 1], stl[IMGIN ? NLDI : 1];
    int ioff[IMGIN ? NLDI : 1];
    unsigned iok = 0;
    if (IMGIN) {
#pragma unroll
        for (int i = 0; i < NLDI; ++i) {
            const int t = tid + i * NT;
            const int fr = t / (2 * PS), rem = t - fr * (2 * PS);
            const int oct = rem / PS, col = rem - oct * PS;
            const int fi = fbase + fr;
            const bool ok = (fr < FR) && (fi >= 0) && (fi < a.Fin);
            iok |= (ok ? 1u : 0u) << i;
            // columns outside [0, J) are read as they lie (neighbouring row / slack): they only reach outputs
            // that the epilogue forces to zero (guard columns) or drops (j >= J)
            ioff[i] = (oct * a.Fin + fi) * a.Jp + (j0 - 4 + col);
        }
    }
    auto stage_load_img = [&](int chunk) {
        const int ci0 = chunk * 8;
        const u32x4* xh;
        const u32x4* xl;
        int zoff;           // rows outside [0, Fin) read the image's zero slot (IDV_IMG_ZSLOT, both planes)
        if (ci0 < a.C0) {
            const int o0 = (ci0 / 4) * a.Fin * a.Jp;
            xh = (const u32x4*)a.x0 + o0;
            xl = xh + a.lo_off0;
            zoff = IDV_IMG_ZSLOT - o0;
        } else {
            const int o1 = ((ci0 - a.C0) / 4) * a.Fin * a.Jp;
            xh = (const u32x4*)a.x1 + o1;
            xl = xh + a.lo_off1;
            zoff = IDV_IMG_ZSLOT - o1;
        }
#pragma unroll
        for (int i = 0; i < NLDI; ++i) {
            const int o = ((iok >> i) & 1u) ? ioff[i] : zoff;
            sth[i] = xh[o];
            stl[i] = xl[o];
        }
    };
    // direct global -> LDS copies (global_load_lds_dwordx4): lane l of a wave fills slot (wave*64 + i*NT + l), i.e.
    // 1 KiB of contiguous LDS per instruction, no VGPR round trip and no ds_write
    auto stage_dma = [&](int chunk, unsigned short* dst, int i_lo, int i_hi) {
        const int ci0 = chunk * 8;
        const u32x4* xh;
        long long lo;
        int zoff;
        if (ci0 < a.C0) {
            const int o0 = (ci0 / 4) * a.Fin * a.Jp;
            xh = (const u32x4*)a.x0 + o0;
            lo = a.lo_off0;
            zoff = IDV_IMG_ZSLOT - o0;
        } else {
            const int o1 = ((ci0 - a.C0) / 4) * a.Fin * a.Jp;
            xh = (const u32x4*)a.x1 + o1;
            lo = a.lo_off1;
            zoff = IDV_IMG_ZSLOT - o1;
        }
        // issued through inline asm: the builtin makes the compiler drain vmcnt before the next ds_read (it cannot
        // prove the patch being read and the patch being filled are different buffers), which serialises the copy
        typedef __attribute__((address_space(3))) unsigned short lds_u16;
        const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u16*)(dst + (size_t)wave * 64 * 8));
#pragma unroll
        for (int i = 0; i < NLDI; ++i) {
            if (i < i_lo || i >= i_hi) continue;
            const int o = ((iok >> i) & 1u) ? ioff[i] : zoff;
            const u32x4* gh = xh + o;
            const u32x4* gl = gh + lo;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(gh), "s"(l0 + (unsigned)(i * NT * 16)) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(gl), "s"(l0 + (unsigned)(i * NT * 16 + IMG * 2)) : "memory", "m0");
        }
    };
    auto stage_store_img = [&](unsigned short* dst) {
#pragma unroll
        for (int i = 0; i < NLDI; ++i) {
            u32x4* d = (u32x4*)dst + (tid + i * NT);
            d[0] = sth[i];
            d[IMG / 8] = stl[i];
        }
    };
    f32x4 stg[IMGIN ? 1 : NLD][8];
    unsigned voff[NLD];
    unsigned okbits = 0;
#pragma unroll
    for (int i = 0; i < (IMGIN ? 0 : NLD); ++i) {
        const int e = tid + i * NT;
        const int oct = e & 1, rest = e >> 1;
        const int fr = rest / PS4, c4 = rest - fr * PS4;
        const int fi = fbase + fr;
        const int jv = j0 - 4 + 4 * c4;
        const bool rowok = (fr < FR) && (fi >= 0) && (fi < a.Fin);
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (rowok && jv + q >= 0 && jv + q < a.J) bits |= 1u << q;
        okbits |= bits << (4 * i);
        voff[i] = bits ? (unsigned)((4 * oct * a.Fin + fi) * a.Jp + jv) : 0u;
    }
    auto stage_load = [&](int chunk) {
        if (IMGIN) { stage_load_img(chunk); return; }
        const int ci0 = chunk * 8;
        const float* base;
        unsigned ristride, chstride;
        if (ci0 < a.C0) {
            base = a.x0 + (size_t)ci0 * a.Fin * a.Jp;
            ristride = (unsigned)a.C0 * a.Fin * a.Jp;
            chstride = (unsigned)a.Fin * a.Jp;
        } else {
            base = a.x1 + (size_t)(ci0 - a.C0) * a.Fin * a.Jp1;
            ristride = (unsigned)a.C1 * a.Fin * a.Jp1;
            chstride = (unsigned)a.Fin * a.Jp1;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool any = ((okbits >> (4 * i)) & 15u) != 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                // plane p of the octet: complex channel p>>1, part p&1
                const unsigned o = any ? voff[i] + (p >> 1) * chstride + (p & 1) * ristride : 0u;
                stg[i][p] = *(const f32x4*)(base + o);
            }
        }
    };
    auto stage_store = [&](unsigned short* dst) {
        if (IMGIN) { stage_store_img(dst); return; }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            const int oct = e & 1, rest = e >> 1;
            const unsigned bits = (okbits >> (4 * i)) & 15u;
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) stg[i][p][q] = ((bits >> q) & 1u) ? stg[i][p][q] : 0.f;
            const int frw = rest / PS4, c4w = rest - frw * PS4;
            // image[fr][octet][column][8 channels]: lanes of one MFMA operand half read consecutive 16-byte slots
            unsigned short* d0 = dst + ((size_t)((frw * 2 + oct) * PS + 4 * c4w) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hw[4], lw[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float x0 = stg[i][2 * w][q], x1 = stg[i][2 * w + 1][q];
                    // hi = truncation to bf16 (one AND), lo = round-to-nearest of the exact remainder
                    const unsigned u0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u;
                    const unsigned u1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
                    hw[w] = (u0 >> 16) | u1;
                    lw[w] = pack_bf16(x0 - __builtin_bit_cast(float, u0), x1 - __builtin_bit_cast(float, u1));
                }
                *(uint4*)(d0 + q * 8) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                *(uint4*)(d0 + q * 8 + IMG) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
            }
        }
    };

    // ---- weight fragments: taps are stored time-tap-major (g = chunk*10 + kt*5 + kf); one "phase" = the five
    // frequency taps of one time tap.  The phase after the current one is prefetched into the other half.
    const int NPH = nchunk * 2;
    const uint4* wstream = (const uint4*)a.wfrag + (size_t)mt0 * NPH * 5 * 128 + lane;
    static_assert(AD == 2 || AD == 3 || AD == 4, "ring depth");
    constexpr int UNR = (AD == 2) ? 1 : (AD == 3 ? 3 : 2);      // chunks per unrolled group: 2*UNR phases = k*AD
    uint4 a_hi[AD][5][MT_W], a_lo[AD][5][MT_W];
    auto load_a_tap = [&](int ph, int kf, uint4 (&dh)[5][MT_W], uint4 (&dl)[5][MT_W]) {
        const int pp = ph < NPH ? ph : NPH - 1;            // past the end: harmless re-read
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            dh[kf][i] = wstream[((size_t)i * NPH * 5 + pp * 5 + kf) * 128];
            dl[kf][i] = wstream[((size_t)i * NPH * 5 + pp * 5 + kf) * 128 + 64];
        }
    };
    auto load_a = [&](int ph, uint4 (&dh)[5][MT_W], uint4 (&dl)[5][MT_W]) {
#pragma unroll
        for (int kf = 0; kf < 5; ++kf) load_a_tap(ph, kf, dh, dl);
    };

    // ---- activation fragments: one LDS row (fr) serves every (output row, freq tap) pair that reads it ----
    const int half = lane >> 5, l31 = lane & 31;
    int cbase[2];                                           // column of this lane for time tap kt
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        // transposed conv: tshift -1 = taps (x[t], x[t-1]) (the forward block); 0 = taps (x[t+1], x[t]), the adjoint of
        // the causal conv
        const int toff = (MODE == IDV_CONV) ? kt + a.tshift : a.tshift + 1 - kt;
        cbase[kt] = wn * (JC_W * 32) + l31 + 4 + toff;
    }
    auto load_b = [&](const unsigned short* P, int fr, int kt, uint4 (&bh)[JC_W], uint4 (&bl)[JC_W]) {
#pragma unroll
        for (int jc = 0; jc < JC_W; ++jc) {
            const unsigned short* src = P + ((size_t)((fr * 2 + half) * PS + cbase[kt] + jc * 32) * 8);
            bh[jc] = *(const uint4*)src;
            bl[jc] = *(const uint4*)(src + IMG);
        }
    };

    if (IMGIN && IDV_IMG_DMA) {
        stage_dma(0, smem16, 0, NLDI);
    } else {
        stage_load(0);
    }
    load_a(0, a_hi[0], a_lo[0]);
    if (!(IMGIN && IDV_IMG_DMA)) stage_store(smem16);
#pragma unroll
    for (int d = 1; d < AD - 1; ++d) load_a(d, a_hi[d], a_lo[d]);
#pragma unroll
    for (int kf = 0; kf < 5; ++kf)
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            asm volatile("" : "+v"(a_hi[0][kf][i].x), "+v"(a_hi[0][kf][i].y), "+v"(a_hi[0][kf][i].z), "+v"(a_hi[0][kf][i].w));
            asm volatile("" : "+v"(a_lo[0][kf][i].x), "+v"(a_lo[0][kf][i].y), "+v"(a_lo[0][kf][i].z), "+v"(a_lo[0][kf][i].w));
        }
    if (IMGIN && IDV_IMG_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto chunk_body = [&](const int chunk, auto kc) {
        constexpr int KC = decltype(kc)::value;                 // chunk index modulo UNR (selects ring slots statically)
        const unsigned short* P = smem16 + (chunk & 1) * BUF;
        const int nxt = (chunk + 1 < nchunk) ? chunk + 1 : chunk;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            constexpr int dummy = 0; (void)dummy;
            const int slot = (2 * KC + kt) % AD, slot_pf = (2 * KC + kt + AD - 1) % AD;
            uint4 b_h[JC_W], b_l[JC_W], n_h[JC_W], n_l[JC_W];
            load_b(P, 0, kt, b_h, b_l);
#pragma unroll
            for (int fr = 0; fr < FR; ++fr) {
                if (fr + 1 < FR) load_b(P, fr + 1, kt, n_h, n_l);
                __builtin_amdgcn_sched_barrier(0);
                if (IMGIN && IDV_IMG_DMA) {
                    // vector-memory instructions are spread over the frequency rows of the phase: a burst of 30 of
                    // them per wave right after the barrier fills the CU's single address path and blocks the MFMA
                    // issue behind it (measured: the kernel time followed the count of these instructions, not where
                    // their data came from).  Time tap 0 carries the patch copy, both carry one weight tap per row.
                    if (kt == 0) stage_dma(nxt, smem16 + ((chunk + 1) & 1) * BUF, (fr * NLDI) / FR, ((fr + 1) * NLDI) / FR);
                    if (fr < 5) load_a_tap(chunk * 2 + kt + AD - 1, fr, a_hi[slot_pf], a_lo[slot_pf]);
                } else if (fr == 0) {
                    load_a(chunk * 2 + kt + AD - 1, a_hi[slot_pf], a_lo[slot_pf]);
                    if (kt == 0) stage_load(nxt);
                }
                if (!(IMGIN && IDV_IMG_DMA) && kt == 1 && fr == FR / 2) stage_store(smem16 + ((chunk + 1) & 1) * BUF);
#pragma unroll
                for (int rt = 0; rt < ROWS; ++rt) {
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        int frr;
                        if (MODE == IDV_CONV) {
                            frr = 2 * rt + kf;
                        } else {
                            if ((rt & 1) != (kf & 1)) continue;
                            frr = (rt >> 1) + 2 - (kf >> 1);
                        }
                        if (frr != fr) continue;
#pragma unroll
                        for (int i = 0; i < MT_W; ++i) {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, a_hi[slot][kf][i]);
                            const bf16x8 al = __builtin_bit_cast(bf16x8, a_lo[slot][kf][i]);
#pragma unroll
                            for (int jc = 0; jc < JC_W; ++jc) {
                                const bf16x8 bh = __builtin_bit_cast(bf16x8, b_h[jc]);
                                const bf16x8 bl = __builtin_bit_cast(bf16x8, b_l[jc]);
                                f32x16 c = acc[i][rt * JC_W + jc];
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                                acc[i][rt * JC_W + jc] = c;
                            }
                        }
                    }
                }
                if (fr + 1 < FR) {
#pragma unroll
                    for (int jc = 0; jc < JC_W; ++jc) { b_h[jc] = n_h[jc]; b_l[jc] = n_l[jc]; }
                }
            }
        }
        // the DMA of the next patch must have landed before anyone reads it; vmcnt retires in order
        if (IMGIN && IDV_IMG_DMA) {
            // every copy instruction sits in time tap 0, so the 10 * MT_W weight loads of time tap 1 are younger
            if (AD > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 * MT_W) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    };
    {
        int chunk = 0;
        for (; chunk + UNR <= nchunk; chunk += UNR) {
            chunk_body(chunk, IntC<0>{});
            if constexpr (UNR > 1) chunk_body(chunk + 1, IntC<1>{});
            if constexpr (UNR > 2) chunk_body(chunk + 2, IntC<2>{});
        }
        if constexpr (UNR > 1) {
            if (chunk < nchunk) chunk_body(chunk, IntC<0>{});
            if constexpr (UNR > 2) {
                if (chunk + 1 < nchunk) chunk_body(chunk + 1, IntC<1>{});
            }
        }
    }

    // ------------------------------------------------------------------ epilogue (as cgemm.hpp, !SWAP)
    if (a.out_img && bid == 0 && tid < 2) {                  // the destination image's zero slot (hi and lo plane)
        unsigned short* z = (unsigned short*)a.out_img + IDV_IMG_ZSLOT * 8 + (tid ? a.out_lo_off : 0);
        *(uint4*)z = make_uint4(0u, 0u, 0u, 0u);
    }
    const float slope = a.slope ? *a.slope : 1.0f;
    const bool has_act = a.slope != nullptr;
#pragma unroll
    for (int ei = 0; ei < MT_W; ++ei) {
        const int mt = mt0 + ei;
        float bia[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bia[r] = a.bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
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
                const f32x16 v = acc[ei][rt * JC_W + jc];
                float y[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float t = v[r] + bia[r];
                    if (has_act) t = t >= 0.f ? t : slope * t;
                    y[r] = keep ? t : 0.f;
                    if (a.out && m < a.M && inb) {
                        const int plane = (m & 1) * a.Cout + (m >> 1);
                        a.out[((size_t)plane * a.Fout + fo) * a.Jp + j] = y[r];
                    }
                }
                if (a.out_img && inb) {
                    // split image [hi|lo][octet = m/8][fo][j][8]: this lane owns elements 4*half .. 4*half+3 of 4 octets
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
                            lw[w] = pack_bf16(x0 - __builtin_bit_cast(float, u0), x1 - __builtin_bit_cast(float, u1));
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
                            st[q][0] += yr; st[q][1] += yi; st[q][2] += yr * yr; st[q][3] += yi * yi; st[q][4] += yr * yi;
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
}

template <int MODE, int WM, int WN, int FO_T, int JC_W, bool STATS, int MT_W = 1, bool IMGIN = false, int AD = 2>
int launch_bf16(const CgemmArgs& a, hipStream_t st) {
    using G = CgemmGeom<MODE, FO_T>;
    constexpr int JT = 32 * JC_W * WN;
    constexpr int PS4L = (JT + 8) / 4;
    constexpr int NTL = WM * WN * 64;
    constexpr int FRPL = ((((G::FR * PS4L * 2) + NTL - 1) / NTL) * NTL + PS4L * 2 - 1) / (PS4L * 2);
    constexpr size_t smem = (size_t)2 * 2 * FRPL * (JT + 8) * 16 * sizeof(unsigned short);
    static_assert(smem <= 160 * 1024, "LDS budget");
    const int rows = (MODE == IDV_TCONV) ? a.Fin : a.Fout;
    CgemmArgs b = a;
    b.jtiles = (a.J + JT - 1) / JT;
    b.ftiles = (rows + FO_T - 1) / FO_T;
    b.mblocks = ((a.M + 31) / 32 + WM * MT_W - 1) / (WM * MT_W);
    const long long tiles = (long long)b.jtiles * b.ftiles;
    // all frequency tiles of a column block on one XCD (cgemm.hpp): -25 % L2-miss traffic in both modes at unchanged speed
    static const int map_ft = [] { const char* e = getenv("IDV_MAP_FT_BF16"); return e ? atoi(e) : 2; }();   // 0 off, 1: TCONV, 2: both
    b.map_ft = ((map_ft == 1 && MODE == IDV_TCONV) || map_ft == 2) ? 1 : 0;
    const long long nblk = b.map_ft ? (long long)((b.jtiles + 7) / 8) * 8 * b.ftiles * b.mblocks : ((tiles + 7) / 8) * 8 * b.mblocks;
    if (nblk > 0x7fffffffLL) return IDV_EINVAL;
    auto k = cgemm_bf16_kernel<MODE, WM, WN, FO_T, JC_W, STATS, MT_W, IMGIN, AD>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(WM * WN * 64), smem, st, b);
    return idv_launch_status();
}

inline int waste(int n, int t) { return ((n + t - 1) / t) * t - n; }

// out[mt][g = chunk*10 + tap][split][lane] (uint4 = 8 bf16): lane l holds row mt*32 + (l&31), channels
// 16*chunk + 8*(l>>5) + 0..7 of tap (kt = tap/5, kf = tap%5: time-tap-major)
__global__ void pack_cconv_bf16_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im,
                                       const float* __restrict__ fold, int Cout, int Cin_total, int Cin_used,
                                       int transposed, int nchunk, int Mtiles, uint4* __restrict__ out, int conj) {
    const long long n = (long long)Mtiles * nchunk * 10 * 2 * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long long t = idx >> 6;
        const int split = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % 10); t /= 10;
        const int chunk = (int)(t % nchunk);
        const int mt = (int)(t / nchunk);
        const int m = mt * 32 + (lane & 31);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = wprime(w_re, w_im, fold, Cout, Cin_total, Cin_used, transposed, m,
                                   16 * chunk + 8 * (lane >> 5) + j, tap % 5, tap / 5, conj);
            v[j] = split == 0 ? w : w - bf16_round(w);
        }
        uint4 o;
        o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
        out[idx] = o;
    }
}

}  // namespace

extern "C" long long idv_cconv_bf16_wfrag_bytes(int Cout, int cin_used) {
    const long long mt = ((2LL * Cout + 127) / 128) * 4;
    const long long nchunk = (2LL * cin_used + 15) / 16;
    return mt * nchunk * 10 * 2 * 64 * 16;
}

extern "C" int idv_cconv_bf16_supported(int transposed, int C0, int C1, int x1_div, int Cout) {
    if (C0 % 8 || C1 % 8 || x1_div != 1) return 0;
    if (2 * Cout < 64) return 0;
    (void)transposed;
    return 1;
}

// template arguments <MODE, WM, WN, FO_T, JC_W> of the cgemm_bf16_kernel instantiation a layer shape uses, as digits
extern "C" int idv_cconv_bf16_config(int transposed, int Cout, int Fin) {
    // digits <MODE, WM, WN, FO_T, JC_W, MT_W> of the instantiation the eval-mode (no statistics) launch uses
    const int rows = transposed ? Fin : (Fin - 1) / 2 + 1;
    const bool fo5 = waste(rows, 5) <= waste(rows, 3);
    const bool wide = 2 * Cout >= 128;
    const int mode = transposed ? 1 : 0;
    if (!transposed && 2 * Cout >= 256) return fo5 ? 41512 : 41322;
    if (wide) return mode * 100000 + (fo5 ? 41511 : (transposed ? 41321 : 41311));
    return mode * 100000 + (fo5 ? 22511 : 22311);
}

extern "C" int idv_pack_cconv_bf16(const float* w_re, const float* w_im, const float* fold, int Cout, int Cin_total,
                                   int Cin_used, int transposed, void* wfrag, void* stream) {
    if (!w_re || !w_im || !wfrag || Cout <= 0 || Cin_used <= 0 || Cin_used > Cin_total || (Cin_used % 8)) return IDV_EINVAL;
    const int Mtiles = ((2 * Cout + 127) / 128) * 4;
    const int nchunk = (2 * Cin_used) / 16;
    long long n = (long long)Mtiles * nchunk * 10 * 2 * 64;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_cconv_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_re, w_im, fold, Cout,
                       Cin_total, Cin_used, transposed, nchunk, Mtiles, (uint4*)wfrag, 0);
    return idv_launch_status();
}

// split-bf16 fragments of the ADJOINT operator (data gradient in bf16x3 training): as idv_pack_cconv_adjoint
extern "C" int idv_pack_cconv_bf16_adjoint(const float* w_re, const float* w_im, int Cout, int Cin_total, int Cin_used,
                                           int transposed, void* wfrag, void* stream) {
    if (!w_re || !w_im || !wfrag || Cout <= 0 || Cin_used <= 0 || Cin_used > Cin_total || (Cin_used % 8)) return IDV_EINVAL;
    const int Mtiles = ((2 * Cout + 127) / 128) * 4;
    const int nchunk = (2 * Cin_used) / 16;
    long long n = (long long)Mtiles * nchunk * 10 * 2 * 64;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_cconv_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_re, w_im,
                       (const float*)nullptr, Cout, Cin_total, Cin_used, transposed, nchunk, Mtiles, (uint4*)wfrag, 1);
    return idv_launch_status();
}

extern "C" int idv_cconv2d_bf16x3_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, int x1_div,
                                      const void* wfrag_bf16, const float* bias, const float* prelu_slope, float* out,
                                      double* stats, double* stats_work, int stats_rep, int transposed, int tshift, int Cout, int Fin,
                                      int B, int Tp, int Jp, int t_valid_out, void* stream) {
    if (!x0 || !wfrag_bf16 || !bias || !out || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (stats && stats_work && (stats_rep < 2 || (stats_rep & (stats_rep - 1)))) return IDV_EINVAL;
    if (!idv_cconv_bf16_supported(transposed, C0, C1, x1_div < 1 ? 1 : x1_div, Cout)) return IDV_EINVAL;
    if (C1 > 0 && (!x1 || Jp1 != Jp || (reinterpret_cast<uintptr_t>(x1) & 15))) return IDV_EINVAL;
    if ((Jp % 4) || (reinterpret_cast<uintptr_t>(x0) & 15) || (tshift != 0 && tshift != -1)) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.Fin = Fin;
    a.Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = Jp1; a.x1_div = 1;
    a.wfrag = (const float*)wfrag_bf16; a.bias = bias; a.slope = prelu_slope; a.out = out;
    a.M = 2 * Cout; a.Mtiles = (a.M + 31) / 32; a.cplx_rows = 1; a.Cout = Cout;
    a.tshift = tshift; a.t_valid = t_valid_out; a.stats = stats; a.ldo = 0; a.nB = B;
    if (Jp < a.J) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rows = transposed ? Fin : a.Fout;
    const bool fo5 = waste(rows, 5) <= waste(rows, 3);
    const bool wide = a.M >= 128;                            // 4 row tiles per workgroup when the layer has them
    const bool repl = stats && stats_work;                   // replicated moment sums, folded into `stats` afterwards (common.hpp)
    if (repl) { a.stats = stats_work; a.stats_rep = stats_rep; }
    const int rc = [&]() -> int {
    // measured (B = 64): the conv with 3 row tiles runs best at 2 waves/SIMD with 32-column tiles (JC_W = 1),
    // the transposed conv (12 accumulator tiles) at 1 wave/SIMD with 64-column tiles
    // conv layers with >= 256 rows: two row tiles per wave (256-row workgroups) halve the staging per MFMA
    // (measured -10 % on enc5, -1.5 % on enc2..4; the transposed conv gets slower or spills, so it keeps one)
    if (a.M >= 256 && !stats && !transposed)
        return fo5 ? launch_bf16<IDV_CONV, 4, 1, 5, 1, false, 2>(a, st) : launch_bf16<IDV_CONV, 4, 1, 3, 2, false, 2>(a, st);
    if (wide && !fo5 && !transposed)
        return stats ? launch_bf16<IDV_CONV, 4, 1, 3, 1, true>(a, st) : launch_bf16<IDV_CONV, 4, 1, 3, 1, false>(a, st);
#define IDV_BF16_DISPATCH(MODE)                                                                              \
    if (stats) {                                                                                             \
        if (wide) return fo5 ? launch_bf16<MODE, 4, 1, 5, 1, true>(a, st) : launch_bf16<MODE, 4, 1, 3, 2, true>(a, st);   \
        return fo5 ? launch_bf16<MODE, 2, 2, 5, 1, true>(a, st) : launch_bf16<MODE, 2, 2, 3, 1, true>(a, st);             \
    } else {                                                                                                 \
        if (wide) return fo5 ? launch_bf16<MODE, 4, 1, 5, 1, false>(a, st) : launch_bf16<MODE, 4, 1, 3, 2, false>(a, st); \
        return fo5 ? launch_bf16<MODE, 2, 2, 5, 1, false>(a, st) : launch_bf16<MODE, 2, 2, 3, 1, false>(a, st);           \
    }
    if (transposed) {
        IDV_BF16_DISPATCH(IDV_TCONV)
    } else {
        IDV_BF16_DISPATCH(IDV_CONV)
    }
#undef IDV_BF16_DISPATCH
    }();
    return (rc || !repl) ? rc : idv_launch_stats_collapse(stats_work, stats_rep, Cout * 5, stats, st);
}

// <MODE, WM, WN, FO_T, JC_W, MT_W, IMGIN, AD> (as decimal digits) of the instantiation idv_cconv2d_img_fwd launches:
// the one place the tiling policy lives (measured at B = 64 on the DCCRN-CL layer shapes)
static int img_cfg(int src_is_image, int transposed, int M, int cin, int Fin) {
    const int rows = transposed ? Fin : (Fin - 1) / 2 + 1;
    const bool fo5 = waste(rows, 5) <= waste(rows, 3);
    const bool wide = M >= 128;
    const int img = src_is_image ? 1 : 0;
    int wm = 4, wn = 1, fo = fo5 ? 5 : 3, jc = 1, mt = 1, ad = img ? IDV_AD : 2;
    if (!transposed) {
        if (M >= 256) {                       // two row tiles per wave halve the staging and the B reads per MFMA
            jc = fo5 ? 1 : 2; mt = 2; ad = 2;
        } else if (wide) {
            // short K (few chunks): two workgroups per CU (a 2-deep weight ring keeps each under 256 registers)
            // overlap one workgroup's prologue / epilogue with the other's MFMAs
            if (img && fo5 && cin <= 64) ad = 2;
        } else {
            wm = 2; wn = 2;
        }
    } else if (wide) {
        jc = fo5 ? 1 : 2;
    } else {
        wm = 2; wn = 2;
    }
    return ((((((transposed ? 1 : 0) * 10 + wm) * 10 + wn) * 10 + fo) * 10 + jc) * 10 + mt) * 100 + img * 10 + ad;
}

extern "C" int idv_cconv_img_config(int src_is_image, int transposed, int Cin, int Cout, int Fin) {
    return img_cfg(src_is_image, transposed, 2 * Cout, Cin, Fin);
}

// Image-source form: x0 / x1 are split images (hi plane at the pointer, lo plane lo_off 16-byte slots further), the
// destination is a planar fp32 buffer, a split image, or both.  Eval mode only (no statistics).
extern "C" int idv_cconv2d_img_fwd(int src_is_image, const void* x0_img, long long lo_off0, int C0, const void* x1_img, long long lo_off1,
                                   int C1, const void* wfrag_bf16, const float* bias, const float* prelu_slope,
                                   float* out_planar, void* out_img, long long out_lo_off,
                                   int transposed, int tshift, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out,
                                   void* stream) {
    if (!x0_img || !wfrag_bf16 || !bias || (!out_planar && !out_img) || C0 <= 0 || Cout <= 0 || Fin <= 0 ||
        B <= 0 || Tp <= 1)
        return IDV_EINVAL;
    if (!idv_cconv_bf16_supported(transposed, C0, C1, 1, Cout) || (Cout % 4)) return IDV_EINVAL;
    if (C1 > 0 && (!x1_img || (reinterpret_cast<uintptr_t>(x1_img) & 15))) return IDV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x0_img) & 15) || (tshift != 0 && tshift != -1))
        return IDV_EINVAL;
    if (out_img && ((reinterpret_cast<uintptr_t>(out_img) & 15) || (out_lo_off % 8))) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = (const float*)x0_img; a.x1 = (const float*)x1_img; a.C0 = C0; a.C1 = C1;
    a.lo_off0 = lo_off0; a.lo_off1 = lo_off1;
    a.Fin = Fin;
    a.Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = Jp; a.x1_div = 1;
    a.wfrag = (const float*)wfrag_bf16; a.bias = bias; a.slope = prelu_slope; a.out = out_planar;
    a.out_img = out_img; a.out_lo_off = out_lo_off;
    a.M = 2 * Cout; a.Mtiles = (a.M + 31) / 32; a.cplx_rows = 1; a.Cout = Cout;
    a.tshift = tshift; a.t_valid = t_valid_out; a.stats = nullptr; a.ldo = 0; a.nB = B;
    if (Jp < a.J || (long long)((2 * (C0 > C1 ? C0 : C1) + 7) / 8) * Fin * Jp > 0x7fffff00LL) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!src_is_image && (Jp % 4)) return IDV_EINVAL;
    switch (img_cfg(src_is_image, transposed, a.M, C0 + C1, Fin)) {
#define IDV_CASE(MODE, WM, WN, FO, JC, MT, IMG, AD)                                                   \
    case ((((((MODE * 10 + WM) * 10 + WN) * 10 + FO) * 10 + JC) * 10 + MT) * 10 + IMG) * 10 + AD:    \
        return launch_bf16<MODE, WM, WN, FO, JC, false, MT, (IMG != 0), AD>(a, st);
        IDV_CASE(0, 4, 1, 5, 1, 2, 0, 2) IDV_CASE(0, 4, 1, 3, 2, 2, 0, 2) IDV_CASE(0, 4, 1, 5, 1, 1, 0, 2)
        IDV_CASE(0, 4, 1, 3, 1, 1, 0, 2) IDV_CASE(0, 2, 2, 5, 1, 1, 0, 2) IDV_CASE(0, 2, 2, 3, 1, 1, 0, 2)
        IDV_CASE(1, 4, 1, 5, 1, 1, 0, 2) IDV_CASE(1, 4, 1, 3, 2, 1, 0, 2) IDV_CASE(1, 2, 2, 5, 1, 1, 0, 2)
        IDV_CASE(1, 2, 2, 3, 1, 1, 0, 2)
        IDV_CASE(0, 4, 1, 5, 1, 2, 1, 2) IDV_CASE(0, 4, 1, 3, 2, 2, 1, 2) IDV_CASE(0, 4, 1, 5, 1, 1, 1, 2)
        IDV_CASE(0, 4, 1, 5, 1, 1, 1, IDV_AD) IDV_CASE(0, 4, 1, 3, 1, 1, 1, IDV_AD) IDV_CASE(0, 2, 2, 5, 1, 1, 1, IDV_AD)
        IDV_CASE(0, 2, 2, 3, 1, 1, 1, IDV_AD)
        IDV_CASE(1, 4, 1, 5, 1, 1, 1, IDV_AD) IDV_CASE(1, 4, 1, 3, 2, 1, 1, IDV_AD) IDV_CASE(1, 2, 2, 5, 1, 1, 1, IDV_AD)
        IDV_CASE(1, 2, 2, 3, 1, 1, 1, IDV_AD)
#undef IDV_CASE
    }
    return IDV_EINVAL;
}

// Training forward from split images: idv_cconv2d_img_fwd with the train-mode moments epilogue (planar fp32 output y, the
// five per-channel sums the batch statistics need) -- what the bf16x3 training mode runs instead of the planar-source
// idv_cconv2d_bf16x3_fwd, whose staging splits the fp32 patch in registers (1.5x slower).  One row tile per wave (the
// two-tile form keeps no registers for the moments).
extern "C" int idv_cconv2d_img_train_fwd(const void* x0_img, long long lo_off0, int C0, const void* x1_img, long long lo_off1, int C1,
                                         const void* wfrag_bf16, const float* bias, float* out_planar, double* stats,
                                         double* stats_work, int stats_rep, int transposed, int Cout, int Fin, int B, int Tp, int Jp,
                                         int t_valid_out, void* stream) {
    if (!x0_img || !wfrag_bf16 || !bias || !out_planar || !stats || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1)
        return IDV_EINVAL;
    if (stats_work && (stats_rep < 2 || (stats_rep & (stats_rep - 1)))) return IDV_EINVAL;
    if (!idv_cconv_bf16_supported(transposed, C0, C1, 1, Cout)) return IDV_EINVAL;
    if (C1 > 0 && (!x1_img || (reinterpret_cast<uintptr_t>(x1_img) & 15))) return IDV_EINVAL;
    if (reinterpret_cast<uintptr_t>(x0_img) & 15) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = (const float*)x0_img; a.x1 = (const float*)x1_img; a.C0 = C0; a.C1 = C1;
    a.lo_off0 = lo_off0; a.lo_off1 = lo_off1;
    a.Fin = Fin;
    a.Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = Jp; a.x1_div = 1;
    a.wfrag = (const float*)wfrag_bf16; a.bias = bias; a.slope = nullptr; a.out = out_planar;
    a.out_img = nullptr; a.out_lo_off = 0;
    a.M = 2 * Cout; a.Mtiles = (a.M + 31) / 32; a.cplx_rows = 1; a.Cout = Cout;
    a.tshift = -1; a.t_valid = t_valid_out; a.stats = stats; a.ldo = 0; a.nB = B;
    if (Jp < a.J || (long long)((2 * (C0 > C1 ? C0 : C1) + 7) / 8) * Fin * Jp > 0x7fffff00LL) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rows = transposed ? Fin : a.Fout;
    const bool fo5 = waste(rows, 5) <= waste(rows, 3);
    const bool wide = a.M >= 128;
    if (stats_work) { a.stats = stats_work; a.stats_rep = stats_rep; }
    const int rc = [&]() -> int {
        if (!transposed) {
            if (wide) return fo5 ? launch_bf16<IDV_CONV, 4, 1, 5, 1, true, 1, true, IDV_AD>(a, st) : launch_bf16<IDV_CONV, 4, 1, 3, 1, true, 1, true, IDV_AD>(a, st);
            return fo5 ? launch_bf16<IDV_CONV, 2, 2, 5, 1, true, 1, true, IDV_AD>(a, st) : launch_bf16<IDV_CONV, 2, 2, 3, 1, true, 1, true, IDV_AD>(a, st);
        }
        if (wide) return fo5 ? launch_bf16<IDV_TCONV, 4, 1, 5, 1, true, 1, true, IDV_AD>(a, st) : launch_bf16<IDV_TCONV, 4, 1, 3, 2, true, 1, true, IDV_AD>(a, st);
        return fo5 ? launch_bf16<IDV_TCONV, 2, 2, 5, 1, true, 1, true, IDV_AD>(a, st) : launch_bf16<IDV_TCONV, 2, 2, 3, 1, true, 1, true, IDV_AD>(a, st);
    }();
    return (rc || !stats_work) ? rc : idv_launch_stats_collapse(stats_work, stats_rep, Cout * 5, stats, st);
}
