// Last decoder block: complex transposed conv with ONE output channel (Cout = 1, M = 2 rows), fused with the
// folded eval BatchNorm + PReLU.  With only two output rows the general kernel wastes 30/32 of every MFMA, so the
// contraction is re-associated: the five frequency taps move from K into M,
//     P[kf, ro][fi][j] = sum_{cc, kt} W'[ro][cc][kf][kt] * x[cc][fi][j - kt]          (10 rows, K = 2*CC)
//     out[ro][2m  ][j] = P[0][m+1] + P[2][m] + P[4][m-1] ,   out[ro][2m+1][j] = P[1][m+1] + P[3][m]
// (reference: causal_ComplexConvTranspose2d.forward, model/complex_progress.py:244-250; Decoder.forward,
// model/pvae_module.py:88-93).  Split-bf16 arithmetic as cgemm_bf16.hip (3 bf16 MFMAs per k block, fp32 accumulate),
// same channels-last LDS patch and staging; the P tiles are combined through LDS in the epilogue.
#include "bf16_common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

constexpr int C1_FI = 8;        // input rows (m) per workgroup -> 16 output rows
constexpr int C1_ROWS = C1_FI + 2;
constexpr int C1_JC = 2;        // 32-column tiles per row
constexpr int C1_JT = 32 * C1_JC;
constexpr int C1_PS = C1_JT + 8;
constexpr int C1_PS4 = C1_PS / 4;
constexpr int C1_SLOTS = C1_ROWS * 2 * C1_PS;       // 16-byte slots (row, octet, column) of one image
constexpr int C1_NLDI = (C1_SLOTS + 255) / 256;     // image-source staging: slots per thread
constexpr int C1_IMG = C1_NLDI * 256 * 8;           // bf16 elements of one image (hi or lo), padded to whole staging rounds
constexpr int C1_BUF = 2 * C1_IMG;
constexpr int C1_NTASK = C1_ROWS * C1_PS4 * 2;
constexpr int C1_NLD = (C1_NTASK + 255) / 256;
constexpr int C1_TILES = C1_ROWS * C1_JC;           // 20
constexpr int C1_TPW = C1_TILES / 4;                // tiles per wave

// IMGIN: sources are split images, staged by LDS-DMA in linear slot order (as cgemm_bf16.hip)
template <bool IMGIN>
__global__ __launch_bounds__(256, 1) void ctconv_c1_bf16_kernel(const CgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int jt = blockIdx.x, ft = blockIdx.y;
    const int j0 = jt * C1_JT, m0 = ft * C1_FI;
    const int fbase = m0 - 1;
    const int CC = 2 * (a.C0 + a.C1);
    const int nchunk = CC / 16;

    f32x16 acc[C1_TPW];
#pragma unroll
    for (int c = 0; c < C1_TPW; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // ---- image-source staging ---------------------------------------------------------------------------
    int ioff[IMGIN ? C1_NLDI : 1];
    unsigned iok = 0;
    if (IMGIN) {
#pragma unroll
        for (int i = 0; i < C1_NLDI; ++i) {
            const int t = tid + i * 256;
            const int fr = t / (2 * C1_PS), rem = t - fr * (2 * C1_PS);
            const int oct = rem / C1_PS, col = rem - oct * C1_PS;
            const int fi = fbase + fr;
            iok |= ((fr < C1_ROWS && fi >= 0 && fi < a.Fin) ? 1u : 0u) << i;
            ioff[i] = (oct * a.Fin + fi) * a.Jp + (j0 - 4 + col);
        }
    }
    auto stage_dma = [&](int chunk, unsigned short* dst, int i_lo, int i_hi) {
        const int ci0 = chunk * 8;
        const u32x4* xh;
        long long lo;
        int zoff;
        if (ci0 < a.C0) {
            const int o0 = (ci0 / 4) * a.Fin * a.Jp;
            xh = (const u32x4*)a.x0 + o0; lo = a.lo_off0; zoff = IDV_IMG_ZSLOT - o0;
        } else {
            const int o1 = ((ci0 - a.C0) / 4) * a.Fin * a.Jp;
            xh = (const u32x4*)a.x1 + o1; lo = a.lo_off1; zoff = IDV_IMG_ZSLOT - o1;
        }
        typedef __attribute__((address_space(3))) unsigned short lds_u16;
        const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u16*)(dst + (size_t)wave * 64 * 8));
#pragma unroll
        for (int i = 0; i < C1_NLDI; ++i) {
            if (i < i_lo || i >= i_hi) continue;
            const int o = ((iok >> i) & 1u) ? ioff[i] : zoff;
            const u32x4* gh = xh + o;
            const u32x4* gl = gh + lo;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(gh), "s"(l0 + (unsigned)(i * 256 * 16)) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(gl), "s"(l0 + (unsigned)(i * 256 * 16 + C1_IMG * 2)) : "memory", "m0");
        }
    };

    // ---- planar staging (identical scheme to cgemm_bf16.hip) ---------------------------------------------
    f32x4 stg[IMGIN ? 1 : C1_NLD][8];
    unsigned voff[C1_NLD];
    unsigned okbits = 0;
#pragma unroll
    for (int i = 0; i < (IMGIN ? 0 : C1_NLD); ++i) {
        const int e = tid + i * 256;
        const int oct = e & 1, rest = e >> 1;
        const int fr = rest / C1_PS4, c4 = rest - fr * C1_PS4;
        const int fi = fbase + fr;
        const int jv = j0 - 4 + 4 * c4;
        const bool rowok = (e < C1_NTASK) && (fi >= 0) && (fi < a.Fin);
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (rowok && jv + q >= 0 && jv + q < a.J) bits |= 1u << q;
        okbits |= bits << (4 * i);
        voff[i] = bits ? (unsigned)((4 * oct * a.Fin + fi) * a.Jp + jv) : 0u;
    }
    auto stage_load = [&](int chunk) {
        if (IMGIN) return;
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
        for (int i = 0; i < C1_NLD; ++i) {
            const bool any = ((okbits >> (4 * i)) & 15u) != 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const unsigned o = any ? voff[i] + (p >> 1) * chstride + (p & 1) * ristride : 0u;
                stg[i][p] = *(const f32x4*)(base + o);
            }
        }
    };
    auto stage_store = [&](unsigned short* dst) {
        if (IMGIN) return;
#pragma unroll
        for (int i = 0; i < C1_NLD; ++i) {
            const int e = tid + i * 256;
            const int oct = e & 1, rest = e >> 1;
            const unsigned bits = (okbits >> (4 * i)) & 15u;
            if (e < C1_NTASK) {
                if (bits != 15u) {
#pragma unroll
                    for (int p = 0; p < 8; ++p)
#pragma unroll
                        for (int q = 0; q < 4; ++q) stg[i][p][q] = ((bits >> q) & 1u) ? stg[i][p][q] : 0.f;
                }
                const int frw = rest / C1_PS4, c4w = rest - frw * C1_PS4;
                unsigned short* d0 = dst + ((size_t)((frw * 2 + oct) * C1_PS + 4 * c4w) * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float x0 = stg[i][2 * w][q], x1 = stg[i][2 * w + 1][q];
                        const unsigned u0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u;
                        const unsigned u1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
                        hw[w] = (u0 >> 16) | u1;
                        lw[w] = pack_bf16(x0 - __builtin_bit_cast(float, u0), x1 - __builtin_bit_cast(float, u1));
                    }
                    *(uint4*)(d0 + q * 8) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    *(uint4*)(d0 + q * 8 + C1_IMG) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
        }
    };

    // ---- weights: [chunk][kt][hi|lo][lane] x 16 B, rows m = 2*kf + ro (10 of 32 used) --------------------
    const uint4* wstream = (const uint4*)a.wfrag + lane;
    uint4 a_cur[2][2], a_nxt[2][2];
    auto load_a = [&](int chunk, uint4 (&d)[2][2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) d[kt][sp] = wstream[(size_t)((chunk * 2 + kt) * 2 + sp) * 64];
    };

    int cb[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) cb[kt] = l31 + 4 - kt;          // transposed conv: tap kt reads x[t - kt]

    if (IMGIN) stage_dma(0, smem16, 0, C1_NLDI);
    stage_load(0);
    load_a(0, a_cur);
    stage_store(smem16);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
            asm volatile("" : "+v"(a_cur[kt][sp].x), "+v"(a_cur[kt][sp].y), "+v"(a_cur[kt][sp].z), "+v"(a_cur[kt][sp].w));
    if (IMGIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const unsigned short* P = smem16 + (chunk & 1) * C1_BUF;
        const int nxt = (chunk + 1 < nchunk) ? chunk + 1 : chunk;
        stage_load(nxt);
        load_a(nxt, a_nxt);
#pragma unroll
        for (int ti = 0; ti < C1_TPW; ++ti) {
            const int t = wave + 4 * ti;
            const int row = t / C1_JC, jc = t - row * C1_JC;
            if (IMGIN) stage_dma(nxt, smem16 + ((chunk + 1) & 1) * C1_BUF, (ti * C1_NLDI) / C1_TPW, ((ti + 1) * C1_NLDI) / C1_TPW);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const unsigned short* src = P + ((size_t)((row * 2 + half) * C1_PS + cb[kt] + jc * 32) * 8);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *(const uint4*)src);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)(src + C1_IMG));
                const bf16x8 ah = __builtin_bit_cast(bf16x8, a_cur[kt][0]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, a_cur[kt][1]);
                f32x16 c = acc[ti];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                acc[ti] = c;
            }
        }
        stage_store(smem16 + ((chunk + 1) & 1) * C1_BUF);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) a_cur[kt][sp] = a_nxt[kt][sp];
        if (IMGIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: P tiles -> LDS [row][10][JT], then each thread sums the taps of its outputs -------------
    float* Pl = (float*)smem16;
#pragma unroll
    for (int ti = 0; ti < C1_TPW; ++ti) {
        const int t = wave + 4 * ti;
        const int row = t / C1_JC, jc = t - row * C1_JC;
        float* dst = Pl + (size_t)row * 10 * C1_JT + jc * 32 + l31;
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * C1_JT] = acc[ti][r];                 // rows 0..3 : kf 0,1
            dst[8 * C1_JT] = acc[ti][4];                                             // rows 8,9 : kf 4
            dst[9 * C1_JT] = acc[ti][5];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(4 + r) * C1_JT] = acc[ti][r];           // rows 4..7 : kf 2,3
        }
    }
    __syncthreads();
    const float slope = a.slope ? *a.slope : 1.0f;
    const bool has_act = a.slope != nullptr;
    const float b0 = a.bias[0], b1 = a.bias[1];
    for (int e = tid; e < 2 * 2 * C1_FI * C1_JT; e += 256) {
        const int jl = e % C1_JT;
        const int fl = (e / C1_JT) % (2 * C1_FI);
        const int ro = e / (C1_JT * 2 * C1_FI);
        const int fo = 2 * m0 + fl, j = j0 + jl;
        if (fo >= a.Fout || j >= a.J) continue;
        const int rl = (fl >> 1) + 1;                                  // patch row of input row m = m0 + (fl>>1)
        const float* pp = Pl + jl;
        float v;
        if ((fl & 1) == 0)
            v = pp[((rl + 1) * 10 + 0 + ro) * C1_JT] + pp[(rl * 10 + 4 + ro) * C1_JT] + pp[((rl - 1) * 10 + 8 + ro) * C1_JT];
        else
            v = pp[((rl + 1) * 10 + 2 + ro) * C1_JT] + pp[(rl * 10 + 6 + ro) * C1_JT];
        v += ro ? b1 : b0;
        if (has_act) v = v >= 0.f ? v : slope * v;
        const int tp = j % a.Tp;
        if (tp < 1 || tp > a.t_valid) v = 0.f;
        a.out[((size_t)ro * a.Fout + fo) * a.Jp + j] = v;
    }
}

// [chunk][kt][split][lane]: lane l holds row (l&31) = 2*kf + ro, channels 16*chunk + 8*(l>>5) + 0..7
__global__ void pack_c1_bf16_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im, const float* __restrict__ fold,
                                    int Cin_total, int Cin_used, int nchunk, uint4* __restrict__ out) {
    const long long n = (long long)nchunk * 2 * 2 * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long long t = idx >> 6;
        const int split = (int)(t & 1); t >>= 1;
        const int kt = (int)(t & 1); t >>= 1;
        const int chunk = (int)t;
        const int row = lane & 31;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float w = 0.f;
            if (row < 10) w = wprime(w_re, w_im, fold, 1, Cin_total, Cin_used, 1, row & 1, 16 * chunk + 8 * (lane >> 5) + j, row >> 1, kt);
            v[j] = split == 0 ? w : w - bf16_round(w);
        }
        out[idx] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
    }
}

}  // namespace

template <bool IMGIN>
static int launch_c1(const CgemmArgs& a, hipStream_t st) {
    constexpr size_t smem = (size_t)2 * C1_BUF * sizeof(unsigned short);
    static_assert(smem >= (size_t)C1_ROWS * 10 * C1_JT * sizeof(float), "P exchange fits in the patch buffers");
    auto k = ctconv_c1_bf16_kernel<IMGIN>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    dim3 grid((a.J + C1_JT - 1) / C1_JT, (a.Fin + C1_FI - 1) / C1_FI);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a);
    return idv_launch_status();
}

extern "C" long long idv_ctconv_c1_wfrag_bytes(int cin_used) { return (2LL * cin_used / 16) * 2 * 2 * 64 * 16; }

extern "C" int idv_pack_ctconv_c1_bf16(const float* w_re, const float* w_im, const float* fold, int Cin_total, int Cin_used,
                                       void* wfrag, void* stream) {
    if (!w_re || !w_im || !wfrag || Cin_used <= 0 || Cin_used > Cin_total || (Cin_used % 8)) return IDV_EINVAL;
    const int nchunk = 2 * Cin_used / 16;
    hipLaunchKernelGGL(pack_c1_bf16_kernel, dim3((nchunk * 256 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_re, w_im, fold,
                       Cin_total, Cin_used, nchunk, (uint4*)wfrag);
    return idv_launch_status();
}

extern "C" int idv_ctconv_c1_bf16x3_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, const void* wfrag,
                                        const float* bias, const float* prelu_slope, float* out, int Fin, int B, int Tp,
                                        int Jp, int t_valid_out, void* stream) {
    if (!x0 || !wfrag || !bias || !out || C0 <= 0 || (C0 % 8) || (C1 % 8) || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (C1 > 0 && (!x1 || Jp1 != Jp || (reinterpret_cast<uintptr_t>(x1) & 15))) return IDV_EINVAL;
    if ((Jp % 4) || (reinterpret_cast<uintptr_t>(x0) & 15) || Jp < B * Tp) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.Fin = Fin; a.Fout = 2 * Fin - 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = Jp1; a.x1_div = 1;
    a.wfrag = (const float*)wfrag; a.bias = bias; a.slope = prelu_slope; a.out = out;
    a.M = 2; a.Cout = 1; a.t_valid = t_valid_out; a.tshift = -1; a.nB = B;
    return launch_c1<false>(a, (hipStream_t)stream);
}

// the same block with split-image sources (hi plane at the pointer, lo plane lo_off 16-byte slots further); planar output
extern "C" int idv_ctconv_c1_img_fwd(const void* x0_img, long long lo_off0, int C0, const void* x1_img, long long lo_off1,
                                     int C1, const void* wfrag, const float* bias, const float* prelu_slope, float* out,
                                     int Fin, int B, int Tp, int Jp, int t_valid_out, void* stream) {
    if (!x0_img || !wfrag || !bias || !out || C0 <= 0 || (C0 % 8) || (C1 % 8) || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (C1 > 0 && (!x1_img || (reinterpret_cast<uintptr_t>(x1_img) & 15))) return IDV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x0_img) & 15) || Jp < B * Tp) return IDV_EINVAL;
    if ((long long)((2 * (C0 > C1 ? C0 : C1) + 7) / 8) * Fin * Jp > 0x7fffff00LL) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = (const float*)x0_img; a.x1 = (const float*)x1_img; a.C0 = C0; a.C1 = C1;
    a.lo_off0 = lo_off0; a.lo_off1 = lo_off1;
    a.Fin = Fin; a.Fout = 2 * Fin - 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = Jp; a.x1_div = 1;
    a.wfrag = (const float*)wfrag; a.bias = bias; a.slope = prelu_slope; a.out = out;
    a.M = 2; a.Cout = 1; a.t_valid = t_valid_out; a.tshift = -1; a.nB = B;
    return launch_c1<true>(a, (hipStream_t)stream);
}
