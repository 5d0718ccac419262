// ComplexLSTM.forward (model/complex_progress.py:50-74) on MI355X.
//
// The four real 2-layer LSTM passes (lstm_re/lstm_im applied to x_r/x_i) are four independent "runs"
// r = 2*z + s (z: 0 = real input, 1 = imag input; s: 0 = lstm_re weights, 1 = lstm_im weights):
//   real = run0 - run3,  imag = run2 + run1.
// Per layer: the input projection for all T is one MFMA GEMM (idv_pw_gemm for layer 0 straight from the
// planar encoder output; gemm_rm below for layer 1), then a persistent recurrent kernel walks the T steps.
// Recurrent kernel: one workgroup = 16 sequences of one run; gates[16 x 4H] = h[16 x H] * W_hh^T on
// v_mfma_f32_16x16x4_f32; h lives transposed in LDS ([unit][seq], conflict-free A-operand reads),
// the gate columns are ordered (unit-block, gate, unit) so that i,f,g,o of one (seq, unit) sit in the same
// lane/register and the cell update is register-local.  For H = 128 the whole W_hh slice of a wave
// (8 tiles x 32 k-steps = 256 VGPRs) stays in registers for all T steps; other H stream W_hh from L2.
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

struct RecArgs {
    const float* g;       // gate pre-activations
    long long g_run_z;    // element stride between z (input part) blocks
    long long g_run_s;    // element offset between weight sets
    int ldg;              // row stride of g
    const float* whh;     // [2 sets][4H/16 tiles][H/4][64]
    float* hout;          // [4 runs][T*B][H]
    int H, B, T;
    float* gsave;         // training: activated gates (i, f, g, o) written back over the pre-activations (same addressing as g)
    float* csave;         // training: cell state per step, [4 runs][T*B][H]
    void* kimg;           // persistent bf16x3 recurrence only: K-major split image of h instead of hout (see lstm_pers.hip)
    long long kimg_lo;
    int Tp, Jp;
};

template <bool WREG>
__global__ __launch_bounds__(256, 1) void lstm_rec_kernel(const RecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = WREG ? 128 : a.H;
    float* hT = sm;                 // [2][H][16]
    float* cS = sm + 2 * H * 16;    // [H][16]   (generic path only)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int run = blockIdx.y, z = run >> 1, s = run & 1;
    const int b0 = blockIdx.x * 16;
    const int col = lane & 15, rq = lane >> 4;          // C/D: col = lane&15, rows rq*4 + r
    const int KK = H / 4, NUB = H / 16;
    const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
    const float* whh = a.whh + (size_t)s * 4 * H * H;
    float* hout = a.hout + (size_t)run * a.T * a.B * H;
    float* gsv = a.gsave ? a.gsave + z * a.g_run_z + s * a.g_run_s : nullptr;
    float* csv = a.csave ? a.csave + (size_t)run * a.T * a.B * H : nullptr;

    for (int e = tid; e < 2 * H * 16 + (WREG ? 0 : H * 16); e += 256) sm[e] = 0.f;

    float breg[WREG ? 2 : 1][4][WREG ? 32 : 1];
    float creg[WREG ? 2 : 1][4];
    if (WREG) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ub = wave + 4 * u;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) breg[u][gg][kk] = whh[((size_t)(ub * 4 + gg) * KK + kk) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) creg[u][r] = 0.f;
        }
    }
    __syncthreads();

    // gate pre-activations of step t+1 are fetched while step t computes (they do not depend on h)
    f32x4 gcur[WREG ? 2 : 1][4], gnxt[WREG ? 2 : 1][4];
    auto load_g = [&](int t, f32x4 (&dst)[WREG ? 2 : 1][4]) {
        const size_t rb = (size_t)t * a.B + b0;
#pragma unroll
        for (int u = 0; u < (WREG ? 2 : 1); ++u) {
            const int ub = wave + 4 * u;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (b0 + rq * 4 + r < a.B) ? rq * 4 + r : 0;      // clamp: rows beyond B are never stored
                    dst[u][gg][r] = g[(rb + row) * a.ldg + (ub * 4 + gg) * 16 + col];
                }
        }
    };
    if (WREG) load_g(0, gcur);

    for (int t = 0; t < a.T; ++t) {
        const float* hc = hT + (t & 1) * H * 16;
        float* hn = hT + ((t + 1) & 1) * H * 16;
        const size_t rowbase = (size_t)t * a.B + b0;
        if (WREG) {
            if (t + 1 < a.T) load_g(t + 1, gnxt);
            f32x4 acc[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) acc[u][gg] = gcur[u][gg];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float av = hc[64 * kk + lane];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
                        acc[u][gg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, breg[u][gg][kk], acc[u][gg], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int unit = (wave + 4 * u) * 16 + col;
                f32x4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ig = sigmoidf_(acc[u][0][r]), fg = sigmoidf_(acc[u][1][r]);
                    const float gv = tanhf_(acc[u][2][r]), og = sigmoidf_(acc[u][3][r]);
                    const float cn = fg * creg[u][r] + ig * gv;
                    creg[u][r] = cn;
                    hv[r] = og * tanhf_(cn);
                    const int row = rq * 4 + r;
                    if (b0 + row < a.B) {
                        hout[(rowbase + row) * H + unit] = hv[r];
                        if (gsv) {
                            float* gp = gsv + (rowbase + row) * a.ldg + ((wave + 4 * u) * 4) * 16 + col;
                            gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                            csv[(rowbase + row) * H + unit] = cn;
                        }
                    }
                }
                *(f32x4*)&hn[unit * 16 + rq * 4] = hv;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) gcur[u][gg] = gnxt[u][gg];
        } else {
            for (int ub = wave; ub < NUB; ub += 4) {
                f32x4 acc[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rq * 4 + r;
                        acc[gg][r] = (b0 + row < a.B) ? g[(rowbase + row) * a.ldg + (ub * 4 + gg) * 16 + col] : 0.f;
                    }
                const float* wt = whh + (size_t)(ub * 4) * KK * 64 + lane;
#pragma unroll 4
                for (int kk = 0; kk < KK; ++kk) {
                    const float av = hc[64 * kk + lane];
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
                        acc[gg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wt[((size_t)gg * KK + kk) * 64], acc[gg], 0, 0, 0);
                }
                const int unit = ub * 16 + col;
                f32x4 cv = *(f32x4*)&cS[unit * 16 + rq * 4];
                f32x4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]);
                    const float gv = tanhf_(acc[2][r]), og = sigmoidf_(acc[3][r]);
                    const float cn = fg * cv[r] + ig * gv;
                    cv[r] = cn;
                    hv[r] = og * tanhf_(cn);
                    const int row = rq * 4 + r;
                    if (b0 + row < a.B) {
                        hout[(rowbase + row) * H + unit] = hv[r];
                        if (gsv) {
                            float* gp = gsv + (rowbase + row) * a.ldg + (ub * 4) * 16 + col;
                            gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                            csv[(rowbase + row) * H + unit] = cn;
                        }
                    }
                }
                *(f32x4*)&cS[unit * 16 + rq * 4] = cv;
                *(f32x4*)&hn[unit * 16 + rq * 4] = hv;
            }
        }
        __syncthreads();
    }
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// Split-precision recurrence for H = 128: h and W_hh as (hi, lo) bf16 pairs, gates = h*W on
// v_mfma_f32_16x16x32_bf16 with the three cross terms (fp32 accumulate) - 96 MFMAs of 16 cycles per step instead
// of 256 of 32.  W_hh (hi+lo: the same 256 VGPRs as the fp32 slice) stays in registers for all T steps; h lives
// in LDS as [seq][k] bf16 images with the 16-byte chunk index XOR-ed with the row (conflict-free b128 reads).
__global__ __launch_bounds__(256, 1) void lstm_rec_bf16_kernel(const RecArgs a) {
    constexpr int H = 128;
    __shared__ __attribute__((aligned(16))) unsigned short hs[2][2][16 * H];     // [buffer][hi|lo][seq][k]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int run = blockIdx.y, z = run >> 1, s = run & 1;
    const int b0 = blockIdx.x * 16;
    const int col = lane & 15, rq = lane >> 4;
    const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
    // bf16 fragments follow the fp32 ones in the packed blob: [set][tile][kb][split][lane] x 16 B
    const uint4* wb = (const uint4*)(a.whh + (size_t)2 * 4 * H * H) + (size_t)s * 32 * 4 * 2 * 64 + lane;
    float* hout = a.hout + (size_t)run * a.T * a.B * H;
    float* gsv = a.gsave ? a.gsave + z * a.g_run_z + s * a.g_run_s : nullptr;      // training: as lstm_rec_kernel
    float* csv = a.csave ? a.csave + (size_t)run * a.T * a.B * H : nullptr;

    for (int e = tid; e < 2 * 2 * 16 * H / 2; e += 256) ((unsigned*)hs)[e] = 0u;

    uint4 breg[2][4][4][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
                    breg[u][gg][kb][sp] = wb[((size_t)(((wave + 4 * u) * 4 + gg) * 4 + kb) * 2 + sp) * 64];
    float creg[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) creg[u][r] = 0.f;

    f32x4 gcur[2][4], gnxt[2][4];
    auto load_g = [&](int t, f32x4 (&dst)[2][4]) {
        const size_t rb = (size_t)t * a.B + b0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (b0 + rq * 4 + r < a.B) ? rq * 4 + r : 0;
                    dst[u][gg][r] = g[(rb + row) * a.ldg + ((wave + 4 * u) * 4 + gg) * 16 + col];
                }
    };
    load_g(0, gcur);
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
        const unsigned short* hc_hi = hs[t & 1][0];
        const unsigned short* hc_lo = hs[t & 1][1];
        unsigned short* hn_hi = hs[(t + 1) & 1][0];
        unsigned short* hn_lo = hs[(t + 1) & 1][1];
        const size_t rowbase = (size_t)t * a.B + b0;
        if (t + 1 < a.T) load_g(t + 1, gnxt);
        f32x4 acc[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) acc[u][gg] = gcur[u][gg];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            // A fragment: seq = lane&15, 8 consecutive k = chunk 4*kb + (lane>>4), swizzled with the row
            const int off = col * H + (((4 * kb + rq) ^ col) & 15) * 8;
            const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *(const uint4*)(hc_hi + off));
            const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *(const uint4*)(hc_lo + off));
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, breg[u][gg][kb][0]);
                    const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, breg[u][gg][kb][1]);
                    f32x4 c = acc[u][gg];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
                    acc[u][gg] = c;
                }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int unit = (wave + 4 * u) * 16 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = sigmoidf_(acc[u][0][r]), fg = sigmoidf_(acc[u][1][r]);
                const float gv = tanhf_(acc[u][2][r]), og = sigmoidf_(acc[u][3][r]);
                const float cn = fg * creg[u][r] + ig * gv;
                creg[u][r] = cn;
                const float hv = og * tanhf_(cn);
                const int row = rq * 4 + r;
                if (b0 + row < a.B) {
                    hout[(rowbase + row) * H + unit] = hv;
                    if (gsv) {
                        float* gp = gsv + (rowbase + row) * a.ldg + ((wave + 4 * u) * 4) * 16 + col;
                        gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                        csv[(rowbase + row) * H + unit] = cn;
                    }
                }
                const __bf16 hh = (__bf16)hv;
                const __bf16 hl = (__bf16)(hv - (float)hh);
                const int o = row * H + ((((unit >> 3) ^ row) & 15) << 3) + (unit & 7);
                hn_hi[o] = __builtin_bit_cast(unsigned short, hh);
                hn_lo[o] = __builtin_bit_cast(unsigned short, hl);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) gcur[u][gg] = gnxt[u][gg];
        __syncthreads();
    }
}

// One time step for hidden sizes whose W_hh does not fit a CU's registers (H = 384, 768 of the VAE encoders):
// the 4H gate columns are split over workgroups (32 units x 4 gates = 128 columns each), every workgroup takes
// 16 sequences of one run, its 4 waves split K, partial gate tiles are reduced through LDS and waves 0/1 do
// the cell update for one 16-unit block each.  h_{t-1} is read from the output of the previous launch
// (kernel boundary = the only synchronisation), c lives in global memory.  The 2*T launches per LSTM are
// queued back to back on the caller's stream; W_hh (2.4 / 9.4 MB per set) stays L2 resident across them.
struct StepArgs {
    RecArgs r;
    float* cstate;   // [4 runs][B][H]
    int t;
};

// VEC4 (H % 64 == 0; fragments packed by idv_pack_lstm_hh in the [tile][kk/4][lane][4] order): one 16-byte weight load and
// one ds_read_b128 of h per 4 k-steps instead of 4 + 4 scalar ones -- the step is bound by the latency of these loads
template <bool VEC4>
__global__ __launch_bounds__(256, 1) void lstm_step_kernel(const StepArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const RecArgs& a = sa.r;
    const int H = a.H, t = sa.t;
    float* hT = sm;                       // [H][16]
    float* red = sm + (size_t)H * 16;     // [4 waves][8 tiles][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = blockIdx.x, b0 = blockIdx.y * 16, run = blockIdx.z, z = run >> 1, s = run & 1;
    const int col = lane & 15, rq = lane >> 4;
    const int KK = H / 4;
    const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
    const float* whh = a.whh + (size_t)s * 4 * H * H;
    float* hout = a.hout + (size_t)run * a.T * a.B * H;
    float* cst = sa.cstate + (size_t)run * a.B * H;

    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;

    // inputs of the cell update do not depend on the MFMA loop: fetch them first (waves 0/1 finish one
    // 16-unit block each), so their latency hides behind the h staging and the contraction
    float gpre[4][4], cpre[4];
    if (wave < 2) {
        const int ubp = cs * 2 + wave;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (b0 + rq * 4 + r < a.B) ? rq * 4 + r : 0;
            const float* gp = g + ((size_t)t * a.B + b0 + row) * a.ldg + ubp * 64 + col;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) gpre[gg][r] = gp[16 * gg];
            cpre[r] = (t > 0) ? cst[(size_t)(b0 + row) * H + ubp * 16 + col] : 0.f;
        }
    }

    if (t > 0) {
        // h_{t-1}[16 seqs][H] -> LDS transposed [k][seq]
        const float* hp = hout + ((size_t)(t - 1) * a.B + b0) * H;
        for (int e = tid; e < 16 * (H / 4); e += 256) {
            const int row = e / (H / 4), k4 = e - row * (H / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b0 + row < a.B) v = *(const f32x4*)(hp + (size_t)row * H + 4 * k4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (VEC4) hT[(k4 >> 2) * 256 + (q * 16 + row) * 4 + (k4 & 3)] = v[q];      // k = 4*k4 + q: k-step k4, lane (q, row)
                else hT[(4 * k4 + q) * 16 + row] = v[q];
            }
        }
        __syncthreads();
        const int kw = KK / 4;             // k-steps per wave
        if (VEC4) {
            const f32x4* wt = (const f32x4*)whh + ((size_t)(cs * 8) * (KK / 4) + wave * (kw / 4)) * 64 + lane;
            const f32x4* hk = (const f32x4*)hT + (size_t)wave * (kw / 4) * 64 + lane;
#pragma unroll 2
            for (int k4 = 0; k4 < kw / 4; ++k4) {
                const f32x4 av = hk[64 * k4];
                f32x4 wv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) wv[q] = wt[((size_t)q * (KK / 4) + k4) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[q][j], acc[q], 0, 0, 0);
            }
        } else {
            const float* wt = whh + ((size_t)(cs * 8) * KK + wave * kw) * 64 + lane;
            const float* hk = hT + (size_t)wave * kw * 64 + lane;
#pragma unroll 8
            for (int kk = 0; kk < kw; ++kk) {
                const float av = hk[64 * kk];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wt[((size_t)q * KK + kk) * 64], acc[q], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) *(f32x4*)&red[((wave * 8 + q) * 64 + lane) * 4] = acc[q];
    __syncthreads();
    if (wave < 2) {
        const int ub = cs * 2 + wave;                  // 16-unit block finished by this wave
        const int unit = ub * 16 + col;
        const size_t rowbase = (size_t)t * a.B + b0;
        f32x4 gate[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 p = *(const f32x4*)&red[((w * 8 + wave * 4 + gg) * 64 + lane) * 4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p[r];
            }
            gate[gg] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rq * 4 + r;
            if (b0 + row >= a.B) continue;
            const float ig = sigmoidf_(gate[0][r] + gpre[0][r]), fg = sigmoidf_(gate[1][r] + gpre[1][r]);
            const float gv = tanhf_(gate[2][r] + gpre[2][r]), og = sigmoidf_(gate[3][r] + gpre[3][r]);
            const size_t ci = (size_t)(b0 + row) * H + unit;
            const float cn = fg * cpre[r] + ig * gv;
            cst[ci] = cn;
            hout[(rowbase + row) * H + unit] = og * tanhf_(cn);
            if (a.gsave) {
                float* gp = a.gsave + z * a.g_run_z + s * a.g_run_s + (rowbase + row) * a.ldg + ub * 64 + col;
                gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                a.csave[(size_t)run * a.T * a.B * H + (rowbase + row) * H + unit] = cn;
            }
        }
    }
}

// G1[run][pos][4H] = X[run][pos][H] * W^T + bias   (layer-1 input projection; X row-major)
// block: 32 positions x 4H columns of one run; wave w owns column tiles w, w+4, ...
template <int NCT>   // column tiles (32) per wave = 4H / 128
__global__ __launch_bounds__(256) void gemm_rm_kernel(const float* __restrict__ X, const float* __restrict__ wfrag,
                                                      const float* __restrict__ bias, float* __restrict__ G, int H,
                                                      long long TB, int KS) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int run = blockIdx.y, s = run & 1;
    const long long pos0 = (long long)blockIdx.x * 32;
    const float* x = X + (size_t)run * TB * H;
    const int tiles_per_set = (4 * H) / 32;
    f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const int half = lane >> 5, l31 = lane & 31;
    for (int k0 = 0; k0 < H; k0 += 64) {
        __syncthreads();
        for (int e = tid; e < 32 * 64; e += 256) {
            const int i = e >> 6, k = e & 63;
            tile[i][k] = (pos0 + i < TB && k0 + k < H) ? x[(size_t)(pos0 + i) * H + k0 + k] : 0.f;
        }
        __syncthreads();
        const int kmax = min(64, H - k0);
        for (int kp = 0; kp < kmax; kp += 2) {
            const float av = tile[l31][kp + half];
            const int ks = (k0 + kp) >> 1;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int mt = s * tiles_per_set + wave + 4 * c;
                const float bv = wfrag[((size_t)mt * KS + ks) * 64 + lane];
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int colg = (wave + 4 * c) * 32 + l31;                 // column within this set's 4H
        const float bm = bias[s * 4 * H + colg];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long pos = pos0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (pos < TB) G[((size_t)run * TB + pos) * (4 * H) + colg] = acc[c][r] + bm;
        }
    }
}

// generic (any H multiple of 32): column tiles looped at run time, 4 accumulators at a time
__global__ __launch_bounds__(256) void gemm_rm_generic_kernel(const float* __restrict__ X, const float* __restrict__ wfrag,
                                                              const float* __restrict__ bias, float* __restrict__ G, int H,
                                                              long long TB, int KS) {
    extern __shared__ float xt[];     // [32][H+1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int run = blockIdx.y, s = run & 1;
    const long long pos0 = (long long)blockIdx.x * 32;
    const float* x = X + (size_t)run * TB * H;
    const int ldx = H + 1;
    for (int e = tid; e < 32 * H; e += 256) {
        const int i = e / H, k = e - i * H;
        xt[i * ldx + k] = (pos0 + i < TB) ? x[(size_t)(pos0 + i) * H + k] : 0.f;
    }
    __syncthreads();
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_per_set = (4 * H) / 32;
    for (int ct = wave; ct < tiles_per_set; ct += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int mt = s * tiles_per_set + ct;
        for (int kp = 0; kp < H; kp += 2) {
            const float av = xt[l31 * ldx + kp + half];
            const float bv = wfrag[((size_t)mt * KS + (kp >> 1)) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        const int colg = ct * 32 + l31;
        const float bm = bias[s * 4 * H + colg];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long pos = pos0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (pos < TB) G[((size_t)run * TB + pos) * (4 * H) + colg] = acc[r] + bm;
        }
    }
}

// out[0][u][b*Tp+t+1] = h[run0] - h[run3];  out[1][u][...] = h[run2] + h[run1];  guard columns zeroed.
__global__ __launch_bounds__(256) void lstm_combine_kernel(const float* __restrict__ h, int H, int B, int T, int Tp, int Jp,
                                                           float* __restrict__ out) {
    __shared__ float tr[2][32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
    const size_t TB = (size_t)T * B;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int i = ty; i < 32; i += 8) {                            // i: t within tile, tx: unit
        const int t = t0 + i, u = u0 + tx;
        float re = 0.f, im = 0.f;
        if (t < T && u < H) {
            const size_t o = ((size_t)t * B + b) * H + u;
            re = h[o] - h[3 * TB * H + o];
            im = h[2 * TB * H + o] + h[1 * TB * H + o];
        }
        tr[0][i][tx] = re;
        tr[1][i][tx] = im;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {                            // i: unit within tile, tx: t
        const int t = t0 + tx, u = u0 + i;
        if (t < T && u < H) {
            const size_t j = (size_t)b * Tp + t + 1;
            out[(size_t)u * Jp + j] = tr[0][tx][i];
            out[((size_t)H + u) * Jp + j] = tr[1][tx][i];
        }
    }
    if (blockIdx.x == 0 && tx == 0)
        for (int i = ty; i < 32; i += 8) {
            const int u = u0 + i;
            if (u < H) {
                out[(size_t)u * Jp + (size_t)b * Tp] = 0.f;
                out[((size_t)H + u) * Jp + (size_t)b * Tp] = 0.f;
            }
        }
}

// h[run][t*B + b][u] -> hp[run][u][b*Tp + t + 1] (planar, one plane set per run) so that the layer-1 input
// projection can run on the tuned PW contraction kernel; guard columns zeroed.
__global__ __launch_bounds__(256) void lstm_to_planar_kernel(const float* __restrict__ h, int H, int B, int T, int Tp, int Jp,
                                                             float* __restrict__ hp) {
    __shared__ float tr[32][33];
    const int b = blockIdx.z % B, run = blockIdx.z / B, t0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
    const size_t TB = (size_t)T * B;
    const float* src = h + (size_t)run * TB * H;
    float* dst = hp + (size_t)run * H * Jp;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, u = u0 + tx;
        tr[i][tx] = (t < T && u < H) ? src[((size_t)t * B + b) * H + u] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + tx, u = u0 + i;
        if (t < T && u < H) dst[(size_t)u * Jp + (size_t)b * Tp + t + 1] = tr[tx][i];
    }
    if (blockIdx.x == 0 && tx == 0)
        for (int i = ty; i < 32; i += 8)
            if (u0 + i < H) dst[(size_t)(u0 + i) * Jp + (size_t)b * Tp] = 0.f;
}

__global__ void zero_tail_kernel(float* __restrict__ act, int planes, int B, int T, int Tp, int Jp) {
    // columns tp in (T, Tp) of every utterance
    const int tail = Tp - 1 - T;
    if (tail <= 0) return;
    const long long n = (long long)planes * B * tail;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % tail);
        const int b = (int)((idx / tail) % B);
        const long long pl = idx / ((long long)tail * B);
        act[(size_t)pl * Jp + (size_t)b * Tp + T + 1 + q] = 0.f;
    }
}

int launch_rec(const RecArgs& ra, float* cstate, int flags, hipStream_t st) {
    dim3 grid((ra.B + 15) / 16, 4);
    if ((flags & 1) && !(flags & 8) && cstate && idv_lstm_pers_supported(ra.H, ra.B)) {
        // H = 384 / 768 (VAE encoders), split-bf16 mode: one persistent cooperative launch per layer (lstm_pers.hip);
        // its exchange buffer lives in the scratch behind cstate (idv_clstm_work_floats covers it: 4*H*Jp floats)
        // the fp32 rows of h are skipped when the next projection reads the split image -- unless backward needs them (gsave)
        return idv_lstm_rec_pers(ra.g, ra.g_run_z, ra.g_run_s, ra.ldg, ra.whh, (ra.kimg && !ra.gsave) ? nullptr : ra.hout, ra.H, ra.B, ra.T,
                                 (void*)(cstate + 4LL * ra.B * ra.H), ra.kimg, ra.kimg_lo, ra.Tp, ra.Jp, ra.gsave, ra.csave, (void*)st);
    }
    if (!(flags & 1) && !(flags & 8) && cstate && idv_lstm_pers_f32_supported(ra.H, ra.B)) {
        // H = 384 / 768, exact fp32: one persistent cooperative launch per layer (lstm_pers_f32.hip) instead of T per-step
        // launches; flags bit 3 keeps the per-step kernel
        return idv_lstm_rec_pers_f32(ra.g, ra.g_run_z, ra.g_run_s, ra.ldg, ra.whh, ra.hout, ra.H, ra.B, ra.T,
                                     (void*)(cstate + 4LL * ra.B * ra.H), ra.gsave, ra.csave, (void*)st);
    }
    if (ra.H == 128 && (flags & 1)) {
        hipLaunchKernelGGL(lstm_rec_bf16_kernel, grid, dim3(256), 0, st, ra);
    } else if (ra.H == 128 && cstate && !(flags & 8) && idv_lstm_coop_f32_supported(ra.H, ra.B)) {
        // exact fp32 on four CUs per sequence tile (lstm_coop_f32.hip); flags bit 3 keeps the one-CU register-resident kernel
        return idv_lstm_rec_coop_f32(ra.g, ra.g_run_z, ra.g_run_s, ra.ldg, ra.whh, ra.hout, ra.H, ra.B, ra.T,
                                     (void*)(cstate + 4LL * ra.B * ra.H), ra.gsave, ra.csave, (void*)st);
    } else if (ra.H == 128) {
        const size_t smem = (size_t)2 * 128 * 16 * sizeof(float);
        hipLaunchKernelGGL(lstm_rec_kernel<true>, grid, dim3(256), smem, st, ra);
    } else if (ra.H % 32 == 0 && cstate) {
        const size_t smem = ((size_t)ra.H * 16 + 4 * 8 * 64 * 4) * sizeof(float);
        const bool vec4 = (ra.H % 64 == 0);            // must match idv_pack_lstm_hh's fragment order
        auto k = vec4 ? lstm_step_kernel<true> : lstm_step_kernel<false>;
        if (smem > 64 * 1024 &&
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return IDV_ELAUNCH;
        dim3 sgrid(ra.H / 32, (ra.B + 15) / 16, 4);
        for (int t = 0; t < ra.T; ++t) {
            StepArgs sa{ra, cstate, t};
            hipLaunchKernelGGL(k, sgrid, dim3(256), smem, st, sa);
        }
    } else {
        const size_t smem = (size_t)3 * ra.H * 16 * sizeof(float);
        if (smem > 64 * 1024 &&
            hipFuncSetAttribute((const void*)lstm_rec_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return IDV_ELAUNCH;
        hipLaunchKernelGGL(lstm_rec_kernel<false>, grid, dim3(256), smem, st, ra);
    }
    return idv_launch_status();
}

}  // namespace

// floats of the scratch region behind cstate: the planar transpose of h0 (per-step paths) or the exchange buffer + arrive
// counters of a cooperative recurrence (whose size does not shrink with T: for short inputs it is the larger one)
static long long hp_floats(int H, int B, int Jp) {
    long long n = 4LL * H * Jp;
    if (idv_lstm_pers_supported(H, B)) {
        const long long m = (idv_lstm_pers_work_bytes(H, B) + 3) / 4;
        if (m > n) n = m;
    }
    if (idv_lstm_coop_f32_supported(H, B)) {
        const long long m = (idv_lstm_coop_f32_work_bytes(H, B) + 3) / 4;
        if (m > n) n = m;
    }
    if (idv_lstm_pers_f32_supported(H, B)) {
        const long long m = (idv_lstm_pers_f32_work_bytes(H, B) + 3) / 4;
        if (m > n) n = m;
    }
    if (idv_lstm_stack2_f32_supported(H, B)) {
        const long long m = (idv_lstm_stack2_f32_work_bytes(H, B) + 3) / 4;
        if (m > n) n = m;
    }
    return (n + 63) / 64 * 64;
}

extern "C" long long idv_clstm_work_floats(int H, int B, int T, int Jp) {
    return 24LL * T * B * H + 4LL * B * H + hp_floats(H, B, Jp) + 4LL * H * Jp;   // [G | h0 | h1 | cstate | hp / exchange | split image of h0]
}
// training (flags bit 2): both layers' gate buffers and the cell states are kept for the backward pass
//   [G0 16TBH | G1 16TBH | h0 4TBH | h1 4TBH | c0 4TBH | c1 4TBH | cstate 4BH | hp 4*H*Jp]
extern "C" long long idv_clstm_train_work_floats(int H, int B, int T, int Jp) {
    return 48LL * T * B * H + 4LL * B * H + hp_floats(H, B, Jp) + 4LL * H * Jp;      // + split image of h0 (bf16x3 mode)
}

static int clstm_fwd_impl(const float* x, int K, const float* wih0, const float* bih0, const float* whh0,
                          const float* wih1, const float* bih1, const float* whh1, int H, int B, int T, int Tp, int Jp,
                          float* work, float* out, int flags, const void* wih1_bf16, const float* wih1_hh, void* stream) {
    if (!x || !wih0 || !bih0 || !whh0 || !wih1 || !bih1 || !whh1 || !work || !out) return IDV_EINVAL;
    if (H <= 0 || (H % 16) || K <= 0 || (K & 1) || B <= 0 || T <= 0 || Tp < T + 1 || Jp < B * Tp) return IDV_EINVAL;
    if ((size_t)3 * H * 16 * sizeof(float) > 160 * 1024) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long long TB = (long long)T * B;
    const bool save = (flags & 4) != 0;
    // training forward in split-bf16 mode: where a split-bf16 recurrence exists (H = 128: register-resident kernel;
    // H = 384 / 768: the persistent cooperative kernel); the generic sizes keep the exact-fp32 recurrence
    if (save && (flags & 1) && H != 128 && ((flags & 8) || !idv_lstm_pers_supported(H, B))) return IDV_EINVAL;
    float* G = work;                       // [2][TB][8H]  then (inference: same memory)  [4][TB][4H]
    float* G1 = save ? work + 16 * TB * H : work;
    float* h0 = (save ? G1 : work) + 16 * TB * H;        // [4][TB][H]
    float* h1 = h0 + 4 * TB * H;
    float* c0 = save ? h1 + 4 * TB * H : nullptr;
    float* c1 = save ? c0 + 4 * TB * H : nullptr;
    float* cstate = (save ? c1 : h1) + 4 * TB * H;       // [4][B][H], used by the per-step kernel only
    int rc;
    // layer 0 input projection: both weight sets at once (M = 8H), one call per input part z
    // (flags bit 1: the caller already filled G through idv_lstm_proj_bf16x3)
    for (int z = 0; z < 2 && !(flags & 2); ++z) {
        rc = idv_pw_gemm(x + (size_t)z * K * Jp, K, wih0, bih0, nullptr, G + (size_t)z * TB * 8 * H, 8 * H, B, Tp, Jp, T, 1,
                         8 * H, stream);
        if (rc) return rc;
    }
    // split-bf16 mode with the persistent recurrence and bf16 fragments of W_ih1: layer 0 writes h0 as the K-major split
    // image the bf16 point-wise kernel reads, no fp32 h0, no transpose (was: fp32 PW contraction, 14 % of the NSVAE step)
    const bool img1 = (flags & 1) && !(flags & 8) && wih1_bf16 && idv_lstm_pers_supported(H, B) &&
                      idv_lstm_proj_bf16_supported(H, H) && (4 * H) % 256 == 0;
    float* hp = cstate + 4LL * B * H;                  // [4 runs][H][Jp] (per-step path) / exchange buffer (persistent path)
    void* himg = (void*)(hp + hp_floats(H, B, Jp));    // [hi | lo][4 runs][H/8][Jp] x 16 B
    const long long himg_lo = 4LL * (H / 8) * Jp;
    // exact fp32 at H = 128 with W_ih of layer 1 in the recurrence's fragment order: both layers in one cooperative launch,
    // layer 1 one step behind layer 0, no hoisted layer-1 projection (lstm_stack2_f32.hip); the training forward keeps the
    // same buffers as the per-layer path (activated gates over G / in G1, cell states)
    if (wih1_hh && !(flags & 1) && !(flags & 8) && idv_lstm_stack2_f32_supported(H, B) && 16LL * TB * H < 0xfffffe00LL) {
        if ((rc = idv_lstm_stack2_f32(G, TB * 8 * H, 4LL * H, 8 * H, whh0, wih1_hh, whh1, bih1, h0, h1, H, B, T,
                                      (void*)(cstate + 4LL * B * H), save ? G1 : nullptr, c0, c1, stream)))
            return rc;
        hipLaunchKernelGGL(lstm_combine_kernel, dim3((T + 31) / 32, (H + 31) / 32, B), dim3(256), 0, st, h1, H, B, T, Tp, Jp, out);
        const long long ntail2 = 2LL * H * B * (Tp - 1 - T);
        if (ntail2 > 0)
            hipLaunchKernelGGL(zero_tail_kernel, dim3((unsigned)((ntail2 + 255) / 256 > 4096 ? 4096 : (ntail2 + 255) / 256)), dim3(256), 0,
                               st, out, 2 * H, B, T, Tp, Jp);
        return idv_launch_status();
    }
    RecArgs r0{G, TB * 8 * H, 4LL * H, 8 * H, whh0, h0, H, B, T, save ? G : nullptr, c0, img1 ? himg : nullptr, himg_lo, Tp, Jp};
    if ((rc = launch_rec(r0, cstate, flags, st))) return rc;
    // layer 1 input projection from h0 (row-major), per run
    const int KS = ((H + 7) / 8) * 4;
    dim3 ggrid((unsigned)((TB + 31) / 32), 4);
    if (img1) {
        if ((rc = idv_lstm_proj1_bf16x3(himg, himg_lo, wih1_bf16, bih1, G1, H, B, T, Tp, Jp, stream))) return rc;
    } else if (H == 128) {
        hipLaunchKernelGGL(gemm_rm_kernel<4>, ggrid, dim3(256), 0, st, h0, wih1, bih1, G1, H, TB, KS);
    } else if (H % 32 == 0) {
        // transpose h0 to planar and use the PW contraction: rows of weight set s start at tile s*(4H/32)
        hipLaunchKernelGGL(lstm_to_planar_kernel, dim3((T + 31) / 32, (H + 31) / 32, 4 * B), dim3(256), 0, st, h0, H, B, T, Tp, Jp, hp);
        if ((rc = idv_launch_status())) return rc;
        for (int run = 0; run < 4; ++run) {
            const int sset = run & 1;
            rc = idv_pw_gemm(hp + (size_t)run * H * Jp, H, wih1 + (size_t)sset * (4 * H / 32) * KS * 64, bih1 + sset * 4 * H, nullptr,
                             G1 + (size_t)run * TB * 4 * H, 4 * H, B, Tp, Jp, T, 1, 4 * H, stream);
            if (rc) return rc;
        }
    } else {
        const size_t smem = (size_t)32 * (H + 1) * sizeof(float);
        if (smem > 64 * 1024 &&
            hipFuncSetAttribute((const void*)gemm_rm_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return IDV_ELAUNCH;
        hipLaunchKernelGGL(gemm_rm_generic_kernel, ggrid, dim3(256), smem, st, h0, wih1, bih1, G1, H, TB, KS);
    }
    if ((rc = idv_launch_status())) return rc;
    // G1 is [run][TB][4H] with run = 2z + s
    RecArgs r1{G1, 2 * TB * 4 * H, TB * 4 * H, 4 * H, whh1, h1, H, B, T, save ? G1 : nullptr, c1, nullptr, 0, Tp, Jp};
    if ((rc = launch_rec(r1, cstate, flags, st))) return rc;
    hipLaunchKernelGGL(lstm_combine_kernel, dim3((T + 31) / 32, (H + 31) / 32, B), dim3(256), 0, st, h1, H, B, T, Tp, Jp, out);
    const long long ntail = 2LL * H * B * (Tp - 1 - T);
    if (ntail > 0)
        hipLaunchKernelGGL(zero_tail_kernel, dim3((unsigned)((ntail + 255) / 256 > 4096 ? 4096 : (ntail + 255) / 256)), dim3(256), 0,
                           st, out, 2 * H, B, T, Tp, Jp);
    return idv_launch_status();
}

extern "C" int idv_clstm_fwd(const float* x, int K, const float* wih0, const float* bih0, const float* whh0,
                             const float* wih1, const float* bih1, const float* whh1, int H, int B, int T, int Tp, int Jp,
                             float* work, float* out, int flags, const void* wih1_bf16, void* stream) {
    return clstm_fwd_impl(x, K, wih0, bih0, whh0, wih1, bih1, whh1, H, B, T, Tp, Jp, work, out, flags, wih1_bf16, nullptr, stream);
}

// idv_clstm_fwd with W_ih of layer 1 ALSO in the recurrence's fragment order (idv_pack_lstm_hh on weight_ih_l1; [4H][H] like
// W_hh): where idv_lstm_stack2_f32_supported(H, B) the exact-fp32 evaluation runs both layers in one cooperative launch;
// everything else is idv_clstm_fwd (wih1_hh may be NULL).
extern "C" int idv_clstm_fwd2(const float* x, int K, const float* wih0, const float* bih0, const float* whh0,
                              const float* wih1, const float* bih1, const float* whh1, const float* wih1_hh, int H, int B, int T,
                              int Tp, int Jp, float* work, float* out, int flags, const void* wih1_bf16, void* stream) {
    return clstm_fwd_impl(x, K, wih0, bih0, whh0, wih1, bih1, whh1, H, B, T, Tp, Jp, work, out, flags, wih1_bf16, wih1_hh, stream);
}
