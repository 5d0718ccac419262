// Backward of ComplexLSTM.forward (model/complex_progress.py:50-74): back-propagation through time of the four real
// 2-layer LSTM passes -- what torch.autograd runs for nn.LSTM behind `loss.backward()` in the reference's train steps
// (supervised_dccrn/train.py:239-243, train_nsvae.py:557-561).
//
// The training forward (idv_clstm_fwd, flags bit 2) leaves the activated gates (i, f, g, o) in the gate buffers and the
// cell states c_t in a side buffer.  Per layer, walking t = T-1 .. 0 with one launch per step (the kernel boundary is the
// only synchronisation, as in the forward's per-step kernel):
//   dh_t = dh_out[t] + dA_{t+1} W_hh          (16 sequences x 4H) x (4H x H) on v_mfma_f32_16x16x4_f32
//   cell backward -> dA_t (pre-activation gate gradients), written (a) over the saved gates, row-major, for the weight
//   gradients and (b) transposed [4H][16] into a ping-pong buffer that the next launch reads as coalesced MFMA operands.
// Weight gradients are contractions over (t, b) and run on the planar wgrad kernel (wgrad.hip) after a transposition
// to the planar-J layout, where h_{t-1} is simply "one column to the left" (the guard column supplies h_{-1} = 0).
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

__device__ __forceinline__ int lstm_src_row(int colp, int H) {
    const int ub = colp >> 6, g = (colp >> 4) & 3, ul = colp & 15;
    return g * H + ub * 16 + ul;
}

// whhT[set][unit tile][blk][lane][4]: 16 gate columns (k) per block; lane l = (q = l>>4, n = l&15) holds, for the four MFMA
// k-steps j of the block, B[k = 16*blk + 4*j + q][n] = W_hh[row(colp = k)][tile*16 + n] -- one 16-byte load per 4 k-steps
__global__ void pack_lstm_hh_bwd_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im, int H,
                                        float* __restrict__ out) {
    const int NTL = H / 16, KB4 = H / 4;      // 4H / 16 blocks
    const long long n = 2LL * NTL * KB4 * 64 * 4;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        long long t = idx >> 8;
        const int blk = (int)(t % KB4); t /= KB4;
        const int tile = (int)(t % NTL);
        const int set = (int)(t / NTL);
        const int colp = 16 * blk + 4 * j + (lane >> 4), unit = tile * 16 + (lane & 15);
        const float* w = set ? w_im : w_re;
        out[idx] = w[(size_t)lstm_src_row(colp, H) * H + unit];
    }
}

// dh[run][t*B + b][u] from the planar output gradient: real = run0 - run3, imag = run2 + run1
__global__ __launch_bounds__(256) void lstm_uncombine_kernel(const float* __restrict__ dout, int H, int B, int T, int Tp, int Jp,
                                                             float* __restrict__ dh) {
    __shared__ float tr[2][32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
    const size_t TB = (size_t)T * B;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {                            // i: unit within tile, tx: t
        const int t = t0 + tx, u = u0 + i;
        float re = 0.f, im = 0.f;
        if (t < T && u < H) {
            const size_t j = (size_t)b * Tp + t + 1;
            re = dout[(size_t)u * Jp + j];
            im = dout[((size_t)H + u) * Jp + j];
        }
        tr[0][i][tx] = re;
        tr[1][i][tx] = im;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {                            // i: t within tile, tx: unit
        const int t = t0 + i, u = u0 + tx;
        if (t < T && u < H) {
            const size_t o = ((size_t)t * B + b) * H + u;
            const float re = tr[0][tx][i], im = tr[1][tx][i];
            dh[o] = re;
            dh[3 * TB * H + o] = -re;
            dh[2 * TB * H + o] = im;
            dh[1 * TB * H + o] = im;
        }
    }
}

// src[(t*B + b)*ld + c0 + u] -> dst[u][b*Tp + t + 1]  (planar rows, guard and tail columns zeroed)
__global__ __launch_bounds__(256) void rows_to_planar_kernel(const float* __restrict__ src, long long ld, int c0, int ncol, int B,
                                                             int T, int Tp, int Jp, float* __restrict__ dst) {
    __shared__ float tr[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, u = u0 + tx;
        tr[i][tx] = (t < T && u < ncol) ? src[((size_t)t * B + b) * ld + c0 + u] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + tx, u = u0 + i;
        if (t < T && u < ncol) dst[(size_t)u * Jp + (size_t)b * Tp + t + 1] = tr[tx][i];
    }
    if (blockIdx.x == 0) {
        const int ng = Tp - T;                                    // tp == 0 and tp > T
        for (int e = threadIdx.x; e < 32 * ng; e += 256) {
            const int u = u0 + e / ng, q = e % ng;
            if (u < ncol) dst[(size_t)u * Jp + (size_t)b * Tp + (q == 0 ? 0 : T + q)] = 0.f;
        }
    }
}

struct BpttArgs {
    float* g;             // activated gates in, pre-activation gate gradients out (same addressing as the forward's g)
    long long g_run_z, g_run_s;
    int ldg;
    const float* c;       // [4][T*B][H] cell states
    const float* dhout;   // [4][T*B][H] gradient arriving at h_t from above
    const float* whhT;    // pack_lstm_hh_bwd
    float* dcs;           // [4][B][H] running dc
    float* dAT;           // [2][4 runs][b tiles][4H/16 blocks][64 lanes][4]: MFMA A operands of the next launch, 16 B per lane
    int H, B, T, t;
};

// NT: 16-unit tiles per workgroup (2 when H % 32 == 0).  A compile-time constant: with a run-time NT the k loop keeps
// branches, is not unrolled, and every 16-byte load is followed by its own vmcnt(0) (48 serial L2 round trips per step).
template <int NT>
__global__ __launch_bounds__(256, 1) void lstm_step_bwd_kernel(const BpttArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4 * 2 * 64 * 4];
    const int H = a.H, t = a.t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = blockIdx.x, bt = blockIdx.y, b0 = bt * 16, run = blockIdx.z, z = run >> 1, s = run & 1;
    const int col = lane & 15, rq = lane >> 4;
    const int ntb = gridDim.y;
    const size_t TBH = (size_t)a.T * a.B * H;
    float* g = a.g + z * a.g_run_z + s * a.g_run_s;
    const float* cst = a.c + (size_t)run * TBH;
    const float* dho = a.dhout + (size_t)run * TBH;
    float* dcs = a.dcs + (size_t)run * a.B * H;
    const size_t atile = (size_t)4 * H * 16;
    float* dAT_w = a.dAT + (((size_t)(t & 1) * 4 + run) * ntb + bt) * atile;
    const float* dAT_r = a.dAT + (((size_t)((t + 1) & 1) * 4 + run) * ntb + bt) * atile;
    const bool last = (t == a.T - 1);

    // inputs of the cell backward do not depend on the contraction: fetch them first (wave q finishes tile q)
    float gi[4], gf[4], gg_[4], go[4], cc[4], cp[4], dhv[4], dcv[4];
    const bool fin = wave < NT;
    const int ub = cs * NT + wave, unit = ub * 16 + col;
    if (fin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int seq = rq * 4 + r;
            const bool ok = b0 + seq < a.B;
            const size_t row = (size_t)t * a.B + b0 + (ok ? seq : 0);
            const float* gp = g + row * a.ldg + ub * 64 + col;
            gi[r] = gp[0]; gf[r] = gp[16]; gg_[r] = gp[32]; go[r] = gp[48];
            cc[r] = cst[row * H + unit];
            cp[r] = (t > 0) ? cst[(row - a.B) * H + unit] : 0.f;
            dhv[r] = ok ? dho[row * H + unit] : 0.f;
            dcv[r] = (!last && ok) ? dcs[(size_t)(b0 + seq) * H + unit] : 0.f;
        }
    }

    f32x4 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
    if (!last) {
        const int kw4 = H / 16;                     // blocks of 16 gate columns per wave: (4H / 16) / 4
        const f32x4* ak = (const f32x4*)dAT_r + (size_t)wave * kw4 * 64 + lane;
        const f32x4* wt = (const f32x4*)a.whhT + (((size_t)s * (H / 16) + cs * NT) * (H / 4) + (size_t)wave * kw4) * 64 + lane;
        // two interleaved partial sums per tile: consecutive MFMAs never wait on each other's result
        f32x4 acc2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[q][r] = 0.f;
        // groups of 4 blocks, the next group's 4 * (1 + NT) 16-byte loads in flight under the current group's MFMAs
        // (hand-pipelined: hipcc does not runtime-unroll this loop, and a load right before its MFMA costs an L2 round trip)
        constexpr int GB = 4;
        f32x4 A[GB], W[NT][GB], An[GB], Wn[NT][GB];
        auto load = [&](int b0, f32x4 (&a_)[GB], f32x4 (&w_)[NT][GB]) {
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int b = (b0 + u < kw4) ? b0 + u : kw4 - 1;          // ragged tail: reload the last block, not used
                a_[u] = ak[(size_t)b * 64];
#pragma unroll
                for (int q = 0; q < NT; ++q) w_[q][u] = wt[((size_t)q * (H / 4) + b) * 64];
            }
        };
        load(0, A, W);
        for (int b0 = 0; b0 < kw4; b0 += GB) {
            load(b0 + GB < kw4 ? b0 + GB : b0, An, Wn);
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                if (b0 + u < kw4) {
#pragma unroll
                    for (int j = 0; j < 4; j += 2)
#pragma unroll
                        for (int q = 0; q < NT; ++q) {
                            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u][j], W[q][u][j], acc[q], 0, 0, 0);
                            acc2[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u][j + 1], W[q][u][j + 1], acc2[q], 0, 0, 0);
                        }
                }
            }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                A[u] = An[u];
#pragma unroll
                for (int q = 0; q < NT; ++q) W[q][u] = Wn[q][u];
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] += acc2[q][r];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *(f32x4*)&red[((wave * 2 + q) * 64 + lane) * 4] = acc[q];
    __syncthreads();
    if (fin) {
        f32x4 dhr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4 p = *(const f32x4*)&red[((w * 2 + wave) * 64 + lane) * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dhr[r] += p[r];
        }
        f32x4 oi, of, og, oo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int seq = rq * 4 + r;
            const bool ok = b0 + seq < a.B;
            const float dh = dhv[r] + dhr[r];
            const float tc = tanhf_(cc[r]);
            const float d_o = dh * tc;
            const float dc = dcv[r] + dh * go[r] * (1.f - tc * tc);
            const float d_i = dc * gg_[r], d_g = dc * gi[r], d_f = dc * cp[r];
            const float ai = d_i * gi[r] * (1.f - gi[r]);
            const float af = d_f * gf[r] * (1.f - gf[r]);
            const float ag = d_g * (1.f - gg_[r] * gg_[r]);
            const float ao = d_o * go[r] * (1.f - go[r]);
            oi[r] = ok ? ai : 0.f; of[r] = ok ? af : 0.f; og[r] = ok ? ag : 0.f; oo[r] = ok ? ao : 0.f;
            if (ok) {
                const size_t row = (size_t)t * a.B + b0 + seq;
                float* gp = g + row * a.ldg + ub * 64 + col;
                gp[0] = ai; gp[16] = af; gp[32] = ag; gp[48] = ao;
                dcs[(size_t)(b0 + seq) * H + unit] = dc * gf[r];
            }
        }
        // transposed copy for the next launch: gate column colp = (ub*4 + g)*16 + col -> block ub*4 + g, k-step j = col>>2,
        // lane (q = col&3, seq): offset ((q*16 + seq)*4 + j) inside the 256-float block
        float* d = dAT_w + (size_t)(ub * 4) * 256 + (size_t)((col & 3) * 16 + rq * 4) * 4 + (col >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d[r * 4] = oi[r];
            d[256 + r * 4] = of[r];
            d[512 + r * 4] = og[r];
            d[768 + r * 4] = oo[r];
        }
    }
}

// db[torch gate row] (+)= sum_j dGp[colp][j]
__global__ __launch_bounds__(256) void lstm_bias_grad_kernel(const float* __restrict__ dGp, int H, int Jp, int J, int accumulate,
                                                             float* __restrict__ db) {
    const int colp = blockIdx.x;
    const float* r = dGp + (size_t)colp * Jp;
    double s = 0;
    for (int j = threadIdx.x; j < J; j += blockDim.x) s += r[j];
    __shared__ double sh[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
        const int row = lstm_src_row(colp, H);
        db[row] = accumulate ? db[row] + v : v;
    }
}

inline int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int idv_pack_lstm_hh_bwd(const float* w_hh_re, const float* w_hh_im, int H, float* whhT, void* stream) {
    if (!w_hh_re || !w_hh_im || !whhT || H <= 0 || (H % 16)) return IDV_EINVAL;
    hipLaunchKernelGGL(pack_lstm_hh_bwd_kernel, dim3(grid_for(2LL * 4 * H * H)), dim3(256), 0, (hipStream_t)stream, w_hh_re,
                       w_hh_im, H, whhT);
    return idv_launch_status();
}

extern "C" int idv_lstm_uncombine(const float* dout, int H, int B, int T, int Tp, int Jp, float* dh, void* stream) {
    if (!dout || !dh || H <= 0 || B <= 0 || T <= 0 || Tp < T + 1 || Jp < B * Tp) return IDV_EINVAL;
    hipLaunchKernelGGL(lstm_uncombine_kernel, dim3((T + 31) / 32, (H + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, dout, H, B,
                       T, Tp, Jp, dh);
    return idv_launch_status();
}

extern "C" int idv_rows_to_planar(const float* src, long long ld, int c0, int ncol, int B, int T, int Tp, int Jp, float* dst,
                                  void* stream) {
    if (!src || !dst || ld <= 0 || c0 < 0 || ncol <= 0 || B <= 0 || T <= 0 || Tp < T + 1 || Jp < B * Tp) return IDV_EINVAL;
    hipLaunchKernelGGL(rows_to_planar_kernel, dim3((T + 31) / 32, (ncol + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, src, ld,
                       c0, ncol, B, T, Tp, Jp, dst);
    return idv_launch_status();
}

extern "C" long long idv_lstm_bptt_work_floats(int H, int B) {
    const long long ntb = (B + 15) / 16;
    long long n = 4LL * B * H + 2LL * 4 * ntb * 4 * H * 16;
    if (idv_lstm_bptt_coop_supported(H, B)) {                   // the cooperative form's exchange buffer + counters
        const long long m = (idv_lstm_bptt_coop_work_bytes(H, B) + 3) / 4;
        if (m > n) n = m;
    }
    return n;
}

extern "C" int idv_lstm_bptt(float* gates, long long g_run_z, long long g_run_s, int ldg, const float* cstates,
                             const float* dhout, const float* whhT, int H, int B, int T, float* work, void* stream) {
    if (!gates || !cstates || !dhout || !whhT || !work || H <= 0 || (H % 16) || B <= 0 || T <= 0 || ldg < 4 * H) return IDV_EINVAL;
    // H = 128: one cooperative launch per layer (lstm_bptt_coop_f32.hip) instead of one launch per step
    if (idv_lstm_bptt_coop_supported(H, B) && !(reinterpret_cast<uintptr_t>(work) & 15))
        return idv_lstm_bptt_coop(gates, g_run_z, g_run_s, ldg, cstates, dhout, whhT, H, B, T, (void*)work, stream);
    BpttArgs a{};
    a.g = gates; a.g_run_z = g_run_z; a.g_run_s = g_run_s; a.ldg = ldg;
    a.c = cstates; a.dhout = dhout; a.whhT = whhT;
    a.dcs = work; a.dAT = work + 4LL * B * H;
    a.H = H; a.B = B; a.T = T;
    const int NT = (H % 32 == 0) ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(H / (16 * NT), (B + 15) / 16, 4);
    for (int t = T - 1; t >= 0; --t) {
        a.t = t;
        if (NT == 2) hipLaunchKernelGGL(lstm_step_bwd_kernel<2>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(lstm_step_bwd_kernel<1>, grid, dim3(256), 0, st, a);
    }
    return idv_launch_status();
}

extern "C" int idv_lstm_bias_grad(const float* dGp, int H, int Jp, int J, int accumulate, float* db, void* stream) {
    if (!dGp || !db || H <= 0 || (H % 16) || J <= 0 || Jp < J) return IDV_EINVAL;
    hipLaunchKernelGGL(lstm_bias_grad_kernel, dim3(4 * H), dim3(256), 0, (hipStream_t)stream, dGp, H, Jp, J, accumulate, db);
    return idv_launch_status();
}
