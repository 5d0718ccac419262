// Weight gradient of the complex conv / transposed conv blocks in split-bf16 ("bf16x3") arithmetic: the contraction of
// wgrad.hip (same operands, same split-K partial tiles, same unpack kernel)
//
//   G[kf][kt][sp][lp] = sum_{fs, j} S[sp][fs][j] * L[lp][2*fs + kf - 2][j + kt + dt0]
//
// on v_mfma_f32_32x32x16_bf16 with both operands split on the fly, x = hi + lo (hi = the fp32 value truncated to bf16,
// lo = bf16(x - hi)), S*L ~ S_hi*L_hi + S_hi*L_lo + S_lo*L_hi, fp32 accumulate: 5.3x fewer MFMA cycles than the exact
// fp32 32x32x2 form.  What autograd computes for nn.Conv2d / nn.ConvTranspose2d weights behind `loss.backward()`
// (reference model/complex_progress.py:8-36, :222-279; train steps supervised_dccrn/train.py:239-243 etc.).
//
// The MFMA k index is the COLUMN j: a lane holds 8 consecutive columns of its plane (one ds_read_b128).  A time tap is a
// column offset of one, which would misalign that read by 2 bytes, so the L tile is staged twice: aligned and shifted by
// one column (the shift is made in registers from the aligned float4 + the neighbour lane's edge element).  The four waves
// are (2 halves of the 128 S planes) x (2 time taps): a wave holds 2 x 5 accumulator tiles (64 S planes x 32 L planes x 5
// frequency taps) and reads 14 LDS fragments per 30 MFMAs.
#include "common.hpp"
#include "bf16_common.hpp"
#include "wgrad_common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int BW_MS = 128, BW_ML = 32, BW_JT = 32, BW_KF = 5;
constexpr int BW_PITCH = BW_JT + 8;                 // bf16 elements per LDS row: 80 bytes = 5 x 16 (odd): conflict-free b128 reads
constexpr int BW_LROWS = BW_ML * BW_KF;             // 160

typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hi_bits(float x) { return __builtin_bit_cast(unsigned, x) & 0xffff0000u; }

// (hi, lo) bf16 quadruples of four floats: hi = the value truncated to bf16, lo = bf16(x - hi)
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo) {
    const unsigned h0 = hi_bits(x0), h1 = hi_bits(x1), h2 = hi_bits(x2), h3 = hi_bits(x3);
    hi = make_uint2((h0 >> 16) | h1, (h2 >> 16) | h3);
    lo = make_uint2(pack_bf16(x0 - __builtin_bit_cast(float, h0), x1 - __builtin_bit_cast(float, h1)),
                    pack_bf16(x2 - __builtin_bit_cast(float, h2), x3 - __builtin_bit_cast(float, h3)));
}

// D = +1 (dt0 == 0) or -1 (dt0 == -1): direction of the shifted copy of the L tile.
// Staging goes through raw buffer loads: an invalid slot (plane or frequency row outside the tensor, column >= J) gets the
// byte offset 0xffffff00 (beyond the tensor), which the hardware answers with zeros -- no per-element masks (they were 2/3 of the 620 vector
// instructions per step that bounded the first version at 190 TFLOP/s).  Needs tensors below 4 GB (host-checked; an offset of
// 0xffffffff wraps in the hardware's range check and reads memory: keep the marker a few hundred bytes below 2^32).
template <int D>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradArgs a) {
    constexpr int Q4 = BW_JT / 4;                                       // float4 slots per row
    constexpr int NS4 = BW_MS * Q4 / 256, NL4 = BW_LROWS * Q4 / 256;    // 4, 5
    constexpr unsigned OOB = 0xffffff00u;       // beyond num_records (< 4 GB - 512) and clear of the 32-bit wrap of offset + size
    __shared__ __attribute__((aligned(16))) unsigned short Ssm[2][BW_MS][BW_PITCH];          // [hi|lo]
    __shared__ __attribute__((aligned(16))) unsigned short Lsm[2][2][BW_LROWS][BW_PITCH];    // [aligned|shifted][hi|lo]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wt = wave & 1;                 // S half, time tap
    const int half = lane >> 5, l31 = lane & 31;
    const int split = blockIdx.x, ts = blockIdx.y, tl = blockIdx.z;
    const int sp0 = ts * BW_MS, lp0 = tl * BW_ML;
    const int jt0 = split * a.jt_per_split;
    int jt1 = jt0 + a.jt_per_split;
    if (jt1 > a.jtiles) jt1 = a.jtiles;
    const int nsteps = (jt1 > jt0) ? (jt1 - jt0) * a.Fs : 0;
    // L column = S column + kt + dt0: the aligned copy serves kt + dt0 == 0, the copy shifted by D the other tap
    const int my_copy = (wt + a.dt0 == 0) ? 0 : 1;
    const __amdgpu_buffer_rsrc_t Sr = __builtin_amdgcn_make_buffer_rsrc((void*)a.S, 0, (unsigned)((size_t)a.Sp * a.Fs * a.JpS * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t Lr = __builtin_amdgcn_make_buffer_rsrc((void*)a.L, 0, (unsigned)((size_t)a.Lp * a.Fl * a.JpL * 4), 0x00020000);

    f32x16 acc[2][BW_KF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < BW_KF; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    // staging slots: thread (r8 = tid / 8, q = tid % 8) owns float4 q of S rows r8 + 32 i (i < 4) and of the L rows
    // (plane r8, frequency tap kf = i) (i < 5): slot offsets are affine in i, so one base register each instead of nine
    // (with 160 accumulator registers that is what keeps the kernel under 256 and lets two workgroups share a CU: the
    // split / pack vector work of one overlaps the MFMAs of the other)
    const int q = tid & (Q4 - 1), r8 = tid / Q4;
    const unsigned sbase = (unsigned)(((size_t)(sp0 + r8) * a.Fs * a.JpS + 4 * q) * 4);
    const unsigned sstep = (unsigned)((size_t)32 * a.Fs * a.JpS * 4);
    const bool lok = lp0 + r8 < a.Lp;
    const unsigned lbase = (unsigned)(((size_t)(lp0 + r8) * a.Fl * a.JpL + 4 * q) * 4);
    const unsigned lstep = (unsigned)((size_t)a.JpL * 4);
    const bool edge_lane = D > 0 ? (q == Q4 - 1) : (q == 0);

    f32x4 sreg[NS4], lreg[NL4];
    float hreg[NL4];

    auto load_step = [&](int step) {
        const int jt = jt0 + step / a.Fs, fs = step - (step / a.Fs) * a.Fs;
        const int j0 = jt * BW_JT;
        const bool colok = j0 + 4 * q < a.J;
        const unsigned us = (unsigned)(((long long)fs * a.JpS + j0) * 4);
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            const unsigned off = (colok && sp0 + r8 + 32 * i < a.Sp) ? sbase + i * sstep + us : OOB;
            sreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(Sr, off, 0, 0));
        }
        const unsigned ul = (unsigned)(((long long)(2 * fs - 2) * a.JpL + j0) * 4);     // wraps for fs = 0: only used when valid
        const int je = D > 0 ? j0 + BW_JT : j0 - 1;                                     // edge column of the shifted copy
        const bool eok = edge_lane && je >= 0 && je < a.J;
        const unsigned ue = (unsigned)(((long long)(2 * fs - 2) * a.JpL + je - 4 * q) * 4);
#pragma unroll
        for (int i = 0; i < NL4; ++i) {
            const int fl = 2 * fs + i - 2;
            const bool rok = lok && fl >= 0 && fl < a.Fl;
            lreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(Lr, (rok && colok) ? lbase + i * lstep + ul : OOB, 0, 0));
            hreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Lr, (rok && eok) ? lbase + i * lstep + ue : OOB, 0, 0));
        }
    };
    auto store_step = [&](int step) {
        // J % 4 != 0: the one float4 per row that straddles J also holds columns of the pitch padding (arbitrary bytes);
        // only the last column tile takes this (uniform) branch
        const int j0s = (jt0 + step / a.Fs) * BW_JT;
        if (j0s + BW_JT > a.J && (a.J & 3)) {
            const int jq = j0s + 4 * q;
#pragma unroll
            for (int c = 1; c < 4; ++c) {
                const bool in = jq + c < a.J;
#pragma unroll
                for (int i = 0; i < NS4; ++i) sreg[i][c] = in ? sreg[i][c] : 0.f;
#pragma unroll
                for (int i = 0; i < NL4; ++i) lreg[i][c] = in ? lreg[i][c] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            const int row = r8 + 32 * i;
            uint2 hi, lo;
            split4(sreg[i][0], sreg[i][1], sreg[i][2], sreg[i][3], hi, lo);
            *(uint2*)&Ssm[0][row][4 * q] = hi;
            *(uint2*)&Ssm[1][row][4 * q] = lo;
        }
#pragma unroll
        for (int i = 0; i < NL4; ++i) {
            const int row = r8 * BW_KF + i;                      // LDS row = plane * 5 + kf (what the B-fragment reads expect)
            const f32x4 v = lreg[i];
            // shifted copy: columns j+D .. j+D+3 = own elements + the neighbour slot's edge element (lanes +-1 hold the
            // neighbouring float4 of the same row: DPP row shift; at the row ends the separately loaded edge element)
            // (float temporaries: __builtin_bit_cast applied directly to a vector element reads element 0 on this compiler)
            const float v_first = v[0], v_last = v[3];
            f32x4 w;
            if (D > 0) {
                const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v_first), 0x101, 0xf, 0xf, true));
                w[0] = v[1]; w[1] = v[2]; w[2] = v[3]; w[3] = edge_lane ? hreg[i] : nb;      // row_shl:1 = lane + 1
            } else {
                const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v_last), 0x111, 0xf, 0xf, true));
                w[0] = edge_lane ? hreg[i] : nb; w[1] = v[0]; w[2] = v[1]; w[3] = v[2];      // row_shr:1 = lane - 1
            }
            uint2 hi, lo;
            split4(v[0], v[1], v[2], v[3], hi, lo);
            *(uint2*)&Lsm[0][0][row][4 * q] = hi;
            *(uint2*)&Lsm[0][1][row][4 * q] = lo;
            split4(w[0], w[1], w[2], w[3], hi, lo);
            *(uint2*)&Lsm[1][0][row][4 * q] = hi;
            *(uint2*)&Lsm[1][1][row][4 * q] = lo;
        }
    };

    if (nsteps > 0) load_step(0);
    for (int step = 0; step < nsteps; ++step) {
        store_step(step);
        __syncthreads();
        if (step + 1 < nsteps) load_step(step + 1);      // global loads fly under this step's MFMAs
        const unsigned short* Ah = &Ssm[0][wm * 64 + l31][half * 8];
        const unsigned short* Al = &Ssm[1][wm * 64 + l31][half * 8];
        const unsigned short* Bh = &Lsm[my_copy][0][l31 * BW_KF][half * 8];
        const unsigned short* Bl = &Lsm[my_copy][1][l31 * BW_KF][half * 8];
        // B fragments one (kc, kf) ahead of their MFMAs (explicit double buffer: 16 instead of 40 registers)
        bf16x8_t ah[2], al[2], bh[2], bl[2];
        bh[0] = *(const bf16x8_t*)(Bh);
        bl[0] = *(const bf16x8_t*)(Bl);
#pragma unroll
        for (int kc = 0; kc < BW_JT / 16; ++kc) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *(const bf16x8_t*)(Ah + i * 32 * BW_PITCH + kc * 16);
                al[i] = *(const bf16x8_t*)(Al + i * 32 * BW_PITCH + kc * 16);
            }
#pragma unroll
            for (int k = 0; k < BW_KF; ++k) {
                const int cur = (kc * BW_KF + k) & 1, nxt = cur ^ 1;
                const int nk = (k + 1 < BW_KF) ? k + 1 : 0, nkc = (k + 1 < BW_KF) ? kc : kc + 1;
                if (nkc < BW_JT / 16) {
                    bh[nxt] = *(const bf16x8_t*)(Bh + nk * BW_PITCH + nkc * 16);
                    bl[nxt] = *(const bf16x8_t*)(Bl + nk * BW_PITCH + nkc * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x16 c = acc[i][k];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[cur], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[cur], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[cur], c, 0, 0, 0);
                    acc[i][k] = c;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // partial tile -> workspace (every slot of the padded tile is written, so the workspace needs no clearing)
    float* P = a.part + (size_t)split * 10 * a.SpPad * a.LpPad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < BW_KF; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sp = sp0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int lp = lp0 + l31;
                P[((size_t)(k * 2 + wt) * a.SpPad + sp) * a.LpPad + lp] = acc[i][k][r];
            }
}

// ---- point-wise form (LSTM input / recurrent projections, ComplexDense, DFT matrices): G[s][l] = sum_j S[s][j] L[l][j + shift]
// The weight gradients of the H = 384 / 768 LSTM projections were 16 % of the NSVAE train step on the fp32 kernel.
// 128 x 128 plane tile, 2 x 2 waves of 64 x 64 (four 32 x 32 accumulator tiles), 32 columns per step, both operands split
// on the fly as above; shift = -1 (h_{t-1} is one column to the left; the guard column supplies h_{-1} = 0) is a column
// offset of the L loads.
constexpr int PW_MS = 128, PW_ML = 128, PWJ = 32;

__global__ __launch_bounds__(256, 2) void wgrad_pw_bf16_kernel(const WgradArgs a) {
    constexpr int Q4 = PWJ / 4;
    constexpr unsigned OOB = 0xffffff00u;
    __shared__ __attribute__((aligned(16))) unsigned short Ssm[2][PW_MS][BW_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Lsm[2][PW_ML][BW_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int split = blockIdx.x, ts = blockIdx.y, tl = blockIdx.z;
    const int sp0 = ts * PW_MS, lp0 = tl * PW_ML;
    const int jt0 = split * a.jt_per_split;
    int jt1 = jt0 + a.jt_per_split;
    if (jt1 > a.jtiles) jt1 = a.jtiles;
    const int nsteps = jt1 > jt0 ? jt1 - jt0 : 0;
    const __amdgpu_buffer_rsrc_t Sr = __builtin_amdgcn_make_buffer_rsrc((void*)a.S, 0, (unsigned)((size_t)a.Sp * a.JpS * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t Lr = __builtin_amdgcn_make_buffer_rsrc((void*)a.L, 0, (unsigned)((size_t)a.Lp * a.JpL * 4), 0x00020000);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    const int q = tid & (Q4 - 1), r8 = tid / Q4;
    const unsigned sbase = (unsigned)(((size_t)(sp0 + r8) * a.JpS + 4 * q) * 4), sstep = (unsigned)((size_t)32 * a.JpS * 4);
    const unsigned lbase = (unsigned)(((size_t)(lp0 + r8) * a.JpL + 4 * q) * 4), lstep = (unsigned)((size_t)32 * a.JpL * 4);
    f32x4 sreg[4], lreg[4];

    auto load_step = [&](int step) {
        const int j0 = (jt0 + step) * PWJ;
        const int js = j0 + 4 * q;                   // first S column of the slot
        int jl = js + a.dt0;                         // first L column (dt0 = shift: 0 or -1)
        const bool neg = jl < 0;                     // only column -1 of the first slot: load columns 0 .. 3, rotate at the store
        if (neg) jl = 0;
        const unsigned us = (unsigned)((long long)j0 * 4), ul = (unsigned)(((long long)j0 + (neg ? 0 : a.dt0)) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                Sr, (js < a.J && sp0 + r8 + 32 * i < a.Sp) ? sbase + i * sstep + us : OOB, 0, 0));
            lreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                Lr, (jl < a.J && lp0 + r8 + 32 * i < a.Lp) ? lbase + i * lstep + ul : OOB, 0, 0));
        }
    };
    auto store_step = [&](int step) {
        const int j0 = (jt0 + step) * PWJ;
        const int js = j0 + 4 * q, jl = js + a.dt0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s0 = sreg[i][0], s1 = sreg[i][1], s2 = sreg[i][2], s3 = sreg[i][3];
            float l0 = lreg[i][0], l1 = lreg[i][1], l2 = lreg[i][2], l3 = lreg[i][3];
            if (jl < 0) { l3 = l2; l2 = l1; l1 = l0; l0 = 0.f; }          // the slot loaded columns 0 .. 3; it holds -1 .. 2
            if (j0 + PWJ > a.J) {                                         // last column tile: columns >= J are pitch padding
                if (js + 1 >= a.J) s1 = 0.f;
                if (js + 2 >= a.J) s2 = 0.f;
                if (js + 3 >= a.J) s3 = 0.f;
                if (jl + 1 >= a.J) l1 = 0.f;
                if (jl + 2 >= a.J) l2 = 0.f;
                if (jl + 3 >= a.J) l3 = 0.f;
            }
            const int row = r8 + 32 * i;
            uint2 hi, lo;
            split4(s0, s1, s2, s3, hi, lo);
            *(uint2*)&Ssm[0][row][4 * q] = hi;
            *(uint2*)&Ssm[1][row][4 * q] = lo;
            split4(l0, l1, l2, l3, hi, lo);
            *(uint2*)&Lsm[0][row][4 * q] = hi;
            *(uint2*)&Lsm[1][row][4 * q] = lo;
        }
    };

    if (nsteps > 0) load_step(0);
    for (int step = 0; step < nsteps; ++step) {
        store_step(step);
        __syncthreads();
        if (step + 1 < nsteps) load_step(step + 1);
        const unsigned short* Ah = &Ssm[0][wm * 64 + l31][half * 8];
        const unsigned short* Al = &Ssm[1][wm * 64 + l31][half * 8];
        const unsigned short* Bh = &Lsm[0][wn * 64 + l31][half * 8];
        const unsigned short* Bl = &Lsm[1][wn * 64 + l31][half * 8];
#pragma unroll
        for (int kc = 0; kc < PWJ / 16; ++kc) {
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *(const bf16x8_t*)(Ah + i * 32 * BW_PITCH + kc * 16);
                al[i] = *(const bf16x8_t*)(Al + i * 32 * BW_PITCH + kc * 16);
                bh[i] = *(const bf16x8_t*)(Bh + i * 32 * BW_PITCH + kc * 16);
                bl[i] = *(const bf16x8_t*)(Bl + i * 32 * BW_PITCH + kc * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x16 c = acc[i][n];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[n], c, 0, 0, 0);
                    acc[i][n] = c;
                }
        }
        __syncthreads();
    }
    float* P = a.part + (size_t)split * a.SpPad * a.LpPad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sp = sp0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int lp = lp0 + wn * 64 + n * 32 + l31;
                P[(size_t)sp * a.LpPad + lp] = acc[i][n][r];
            }
}

}  // namespace

extern "C" long long idv_cconv_wgrad_bf16_work_floats(int Cs, int Cl, int B, int Tp) {
    if (Cs <= 0 || Cl <= 0 || B <= 0 || Tp <= 0) return -1;
    const Plan p = make_plan(2 * Cs, 2 * Cl, B * Tp, BW_MS, BW_ML, BW_JT);
    const long long n = (long long)p.nsplit * 10 * p.SpPad * p.LpPad, nf = idv_cconv_wgrad_work_floats(Cs, Cl, B, Tp);
    return n > nf ? n : nf;          // covers the exact-fp32 fallback as well
}

// idv_cconv2d_bwd_weight in split-bf16 arithmetic (bf16x3 training mode); arguments as there, work:
// idv_cconv_wgrad_bf16_work_floats(Cs, Cl, B, Tp) floats with (Cs, Cl) = (Cout, Cx) for the conv, (Cx, Cout) transposed
extern "C" int idv_cconv2d_bwd_weight_bf16x3(const float* x, int Cx, int ci_off, const float* dy, int Cout, int Cin_total,
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
    // the buffer-load staging needs 32-bit byte offsets; larger tensors: the exact-fp32 kernel
    if ((size_t)a.Sp * a.Fs * a.JpS * 4 >= 0xfffffe00ull || (size_t)a.Lp * a.Fl * a.JpL * 4 >= 0xfffffe00ull) {
        if (idv_cconv_wgrad_work_floats(transposed ? Cx : Cout, transposed ? Cout : Cx, B, Tp) > work_floats) return IDV_EINVAL;
        return idv_cconv2d_bwd_weight(x, Cx, ci_off, dy, Cout, Cin_total, transposed, tshift, Fin, B, Tp, Jp_x, Jp_dy, work,
                                      work_floats, dw_re, dw_im, stream);
    }
    const Plan p = make_plan(a.Sp, a.Lp, a.J, BW_MS, BW_ML, BW_JT);
    if ((long long)p.nsplit * 10 * p.SpPad * p.LpPad > work_floats) return IDV_EINVAL;
    a.part = work; a.SpPad = p.SpPad; a.LpPad = p.LpPad; a.jtiles = p.jtiles; a.jt_per_split = p.jt_per_split;
    hipStream_t st = (hipStream_t)stream;
    if (a.dt0 == 0)
        hipLaunchKernelGGL(wgrad_bf16_kernel<1>, dim3(p.nsplit, p.tilesS, p.tilesL), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(wgrad_bf16_kernel<-1>, dim3(p.nsplit, p.tilesS, p.tilesL), dim3(256), 0, st, a);
    launch_wgrad_unpack_conv(work, p.nsplit, p.SpPad, p.LpPad, Cout, Cx, Cin_total, ci_off, transposed, dw_re, dw_im, st);
    return idv_launch_status();
}

// idv_pw_bwd_weight in split-bf16 arithmetic (bf16x3 training mode); arguments and work size as there
extern "C" int idv_pw_bwd_weight_bf16x3(const float* dout, int M, int Jp_d, const float* x, int K, int Jp_x, int J, int shift,
                                        float* work, long long work_floats, float* dw, int ldw, int rowmap, int H, int accumulate,
                                        void* stream) {
    if (!dout || !x || !work || !dw || M <= 0 || K <= 0 || J <= 0 || ldw < K || (shift != 0 && shift != -1)) return IDV_EINVAL;
    if ((Jp_d % 4) || (Jp_x % 4) || !aligned16(dout) || !aligned16(x) || Jp_d < J || Jp_x < J) return IDV_EINVAL;
    if (rowmap == 1 && (H <= 0 || (H % 16) || M % (4 * H))) return IDV_EINVAL;
    if ((size_t)M * Jp_d * 4 >= 0xfffffe00ull || (size_t)K * Jp_x * 4 >= 0xfffffe00ull)
        return idv_pw_bwd_weight(dout, M, Jp_d, x, K, Jp_x, J, shift, work, work_floats, dw, ldw, rowmap, H, accumulate, stream);
    WgradArgs a{};
    a.S = dout; a.Sp = M; a.Fs = 1; a.JpS = Jp_d;
    a.L = x;    a.Lp = K; a.Fl = 1; a.JpL = Jp_x;
    a.dt0 = shift; a.J = J;
    const Plan p = make_plan(M, K, J, PW_MS, PW_ML, PWJ);
    if ((long long)p.nsplit * p.SpPad * p.LpPad > work_floats) return IDV_EINVAL;
    a.part = work; a.SpPad = p.SpPad; a.LpPad = p.LpPad; a.jtiles = p.jtiles; a.jt_per_split = p.jt_per_split;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_pw_bf16_kernel, dim3(p.nsplit, p.tilesS, p.tilesL), dim3(256), 0, st, a);
    launch_wgrad_unpack_plain(work, p.nsplit, p.SpPad, p.LpPad, M, K, ldw, rowmap, H, accumulate, dw, st);
    return idv_launch_status();
}
