// Split-precision (bf16x3) form of the point-wise contraction with the transposed ("swap") store: the hoisted LSTM
// input projection  G[part][(t, b)][m] = sum_k W[m][k] * x[part][k][j] + bias[m]   (reference: nn.LSTM inside
// ComplexLSTM.forward, model/complex_progress.py:50-74 -- lstm_re / lstm_im applied to the real and the imaginary
// input; the two weight sets are stacked along m, the two inputs are the two `parts`, so one weight fragment
// serves both).  Same arithmetic as cgemm_bf16.hip: x = hi + lo, w = hi + lo, three bf16 MFMAs per product, fp32
// accumulate.
//
// The activation comes as a K-major split image  ximg[hi|lo][k octet][Jp][8]  (idv_planar_to_kimage: octet o =
// planes 8o .. 8o+7 of the planar source), so staging is the same verbatim LDS-DMA copy as the conv path.  MFMA
// roles are swapped (A = activation: 32 columns j, B = weights: 32 features m) so that a lane's accumulator
// registers run along j and lanes run along m: the store to the row-major G is 128-byte coalesced.
//
// Workgroup: 4 waves x (2 feature tiles of 32) = 256 features x JT = 32*JC columns x NP parts; K stage = 64 k.
#include "bf16_common.hpp"
#include "../../include/idccrn_hip.h"

namespace {

struct PwBf16Args {
    const u32x4* ximg;        // hi plane of the K-major image
    long long lo_off;         // hi -> lo distance, 16-byte slots
    long long part_stride;    // slots between the parts' first octets (KO * Jp when the parts are contiguous)
    int KO;                   // k octets per part (K / 8), a multiple of 8
    int J, Jp, Tp, t_valid, nB;
    const uint4* wfrag;       // [m tile][k block of 16][hi|lo][lane] x 16 B
    const float* bias;        // [M]
    float* out;               // [part][T*B][ldo]
    long long out_part_stride;
    int ldo, M;
    int jtiles, mblocks;
};

constexpr int PW_SO = 8;      // octets (of 8 k) per stage
constexpr int PW_MW = 2;      // feature tiles per wave
constexpr int PW_RING = 4;    // weight ring depth in k blocks (= k blocks per stage)

// SWAP: transposed store (LSTM gates, see above).  !SWAP: MFMA roles as in the conv kernels (A = weights), planar
// output out[m][Jp] with guard columns zeroed -- the DFT / inverse DFT of STFT / ISTFT and ComplexDense.
template <int JC, int NP, bool SWAP>
__global__ __launch_bounds__(256, 1) void pw_bf16_kernel(const PwBf16Args a) {
    constexpr int JT = 32 * JC;
    constexpr int NSLOT = NP * PW_SO * JT;           // 16-byte slots per stage image (hi or lo)
    constexpr int NLD = NSLOT / 256;
    static_assert(NSLOT % 256 == 0, "whole staging rounds");
    constexpr int IMG = NSLOT * 8;                   // bf16 elements of one image
    constexpr int BUF = 2 * IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware order: the feature blocks of one column tile are consecutive on one XCD (they share the patch)
    const int MB = a.mblocks;
    const int bid = blockIdx.x;
    const int grp = bid / (8 * MB), rem = bid - grp * (8 * MB);
    const int jt = grp * 8 + (rem & 7);
    const int mblk = rem >> 3;
    if (jt >= a.jtiles) return;
    const int j0 = jt * JT;
    const int mt0 = (mblk * 4 + wave) * PW_MW;

    f32x16 acc[NP][JC][PW_MW];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int jc = 0; jc < JC; ++jc)
#pragma unroll
            for (int i = 0; i < PW_MW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][jc][i][r] = 0.f;

    // staging slot t = ((part * SO + octet) * JT + column); columns past J read whatever lies there (dropped outputs)
    long long soff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int t = tid + i * 256;
        const int col = t % JT, po = t / JT;
        const int o = po % PW_SO, p = po / PW_SO;
        soff[i] = (long long)p * a.part_stride + (long long)o * a.Jp + (j0 + col);
    }
    typedef __attribute__((address_space(3))) unsigned short lds_u16;
    auto stage_dma = [&](int stage, unsigned short* dst, int i_lo, int i_hi) {
        const u32x4* xh = a.ximg + (long long)stage * PW_SO * a.Jp;
        const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u16*)(dst + (size_t)wave * 64 * 8));
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (i < i_lo || i >= i_hi) continue;
            const u32x4* gh = xh + soff[i];
            const u32x4* gl = gh + a.lo_off;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(gh), "s"(l0 + (unsigned)(i * 256 * 16)) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(gl), "s"(l0 + (unsigned)(i * 256 * 16 + IMG * 2)) : "memory", "m0");
        }
    };

    const int NKB = a.KO / 2;                         // k blocks of 16
    const int nstage = a.KO / PW_SO;
    const uint4* wstream = a.wfrag + (size_t)mt0 * NKB * 128 + lane;
    uint4 w_hi[PW_RING][PW_MW], w_lo[PW_RING][PW_MW];
    auto load_w = [&](int kb, uint4 (&dh)[PW_MW], uint4 (&dl)[PW_MW]) {
        const int kk = kb < NKB ? kb : NKB - 1;       // past the end: harmless re-read
#pragma unroll
        for (int i = 0; i < PW_MW; ++i) {
            dh[i] = wstream[((size_t)i * NKB + kk) * 128];
            dl[i] = wstream[((size_t)i * NKB + kk) * 128 + 64];
        }
    };

    const bool live = mt0 * 32 < a.M;                  // waves whose feature tiles are all padding only stage and sync
    stage_dma(0, smem16, 0, NLD);
#pragma unroll
    for (int d = 0; d < PW_RING - 1; ++d) load_w(d, w_hi[d], w_lo[d]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int stage = 0; stage < nstage; ++stage) {
        const unsigned short* P = smem16 + (stage & 1) * BUF;
        unsigned short* Pn = smem16 + ((stage + 1) & 1) * BUF;
        const int nxt = stage + 1 < nstage ? stage + 1 : stage;
#pragma unroll
        for (int q = 0; q < PW_SO / 2; ++q) {         // k block = octets 2q, 2q+1 of the stage
            // spread the next patch's copy and the weight prefetch over the stage (see cgemm_bf16.hip)
            stage_dma(nxt, Pn, (q * NLD) / (PW_SO / 2), ((q + 1) * NLD) / (PW_SO / 2));
            load_w(stage * (PW_SO / 2) + q + PW_RING - 1, w_hi[(q + PW_RING - 1) % PW_RING], w_lo[(q + PW_RING - 1) % PW_RING]);
            if (live)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int jc = 0; jc < JC; ++jc) {
                    const unsigned short* src = P + ((size_t)((p * PW_SO + 2 * q + half) * JT + jc * 32 + l31)) * 8;
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, *(const uint4*)src);
                    const bf16x8 xl = __builtin_bit_cast(bf16x8, *(const uint4*)(src + IMG));
#pragma unroll
                    for (int i = 0; i < PW_MW; ++i) {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, w_hi[q % PW_RING][i]);
                        const bf16x8 wl = __builtin_bit_cast(bf16x8, w_lo[q % PW_RING][i]);
                        f32x16 c = acc[p][jc][i];
                        if (SWAP) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, c, 0, 0, 0);
                        } else {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, c, 0, 0, 0);
                        }
                        acc[p][jc][i] = c;
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // every copy instruction is older than the 2 * MW weight loads of the stage's last k block (vmcnt retires
        // in order), so those may stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW_MW) : "memory");
        __syncthreads();
    }

    if (!SWAP) {
        // ---- epilogue, planar: lanes run along j, registers along m -> out[part][m][Jp], guard columns zero
#pragma unroll
        for (int i = 0; i < PW_MW; ++i)
#pragma unroll
            for (int jc = 0; jc < JC; ++jc) {
                const int j = j0 + jc * 32 + l31;
                if (j >= a.J) continue;
                const int tp = j % a.Tp;
                const bool keep = (tp >= 1) && (tp <= a.t_valid);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m >= a.M) continue;
                    const float bm = a.bias ? a.bias[m] : 0.f;
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        a.out[(size_t)p * a.out_part_stride + (size_t)m * a.Jp + j] = keep ? acc[p][jc][i][r] + bm : 0.f;
                }
            }
        return;
    }
    // ---- epilogue: registers run along j, lanes along m -> out[part][(tp-1)*B + b][m]
#pragma unroll
    for (int i = 0; i < PW_MW; ++i) {
        const int m = (mt0 + i) * 32 + l31;
        if (m >= a.M) continue;
        const float bm = a.bias[m];
#pragma unroll
        for (int jc = 0; jc < JC; ++jc)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + jc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (j >= a.J) continue;
                const int b = j / a.Tp, tp = j - b * a.Tp;
                if (tp < 1 || tp > a.t_valid) continue;
                const size_t row = (size_t)(tp - 1) * a.nB + b;
#pragma unroll
                for (int p = 0; p < NP; ++p) a.out[(size_t)p * a.out_part_stride + row * a.ldo + m] = acc[p][jc][i][r] + bm;
            }
    }
}

template <int JC, int NP, bool SWAP>
int launch_pw(const PwBf16Args& a0, hipStream_t st) {
    constexpr int JT = 32 * JC;
    constexpr size_t smem = (size_t)2 * 2 * NP * PW_SO * JT * 16;
    static_assert(smem <= 160 * 1024, "LDS budget");
    PwBf16Args a = a0;
    a.jtiles = (a.J + JT - 1) / JT;
    a.mblocks = (a.M + 255) / 256;
    const long long nblk = (long long)((a.jtiles + 7) / 8) * 8 * a.mblocks;
    auto k = pw_bf16_kernel<JC, NP, SWAP>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return IDV_ELAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), smem, st, a);
    return idv_launch_status();
}

// one thread: one (octet, j) slot = 8 consecutive planes
__global__ void planar_to_kimage_kernel(const float* __restrict__ x, int nvalid, int nplanes, int J, int Jp,
                                        unsigned short* __restrict__ img, long long lo_off) {
    const long long n = (long long)(nplanes / 8) * Jp;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Jp);
        const long long o = idx / Jp;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float x0 = 0.f, x1 = 0.f;
            if (j < J) {
                if (8 * o + 2 * w < nvalid) x0 = x[(size_t)(8 * o + 2 * w) * Jp + j];
                if (8 * o + 2 * w + 1 < nvalid) x1 = x[(size_t)(8 * o + 2 * w + 1) * Jp + j];
            }
            const unsigned u0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u;
            const unsigned u1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
            hw[w] = (u0 >> 16) | u1;
            lw[w] = pack_bf16(x0 - __builtin_bit_cast(float, u0), x1 - __builtin_bit_cast(float, u1));
        }
        *(uint4*)(img + idx * 8) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *(uint4*)(img + lo_off + idx * 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

// wfrag[m tile][k block][hi|lo][lane]: lane l holds feature m = 32*mt + (l & 31), k = 16*kb + 8*(l >> 5) .. +7
__global__ void pack_lstm_ih_bf16_kernel(const float* __restrict__ w_re, const float* __restrict__ w_im, int H, int K,
                                         uint4* __restrict__ out) {
    const int M = 8 * H, NKB = K / 16;
    const long long n = (long long)(M / 32) * NKB * 2 * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long long t = idx >> 6;
        const int split = (int)(t & 1); t >>= 1;
        const int kb = (int)(t % NKB);
        const int mt = (int)(t / NKB);
        const int m = mt * 32 + (lane & 31);
        const int set = m / (4 * H), colp = m % (4 * H);
        // gate column permutation of the recurrence kernels: colp = ((u/16)*4 + g)*16 + u%16 -> torch row g*H + u
        const int row = ((colp >> 4) & 3) * H + (colp >> 6) * 16 + (colp & 15);
        const float* w = (set ? w_im : w_re) + (size_t)row * K + 16 * kb + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = split == 0 ? w[j] : w[j] - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, w[j]) & 0xffff0000u);
        uint4 o;
        if (split == 0) {
            unsigned u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = __builtin_bit_cast(unsigned, v[j]) >> 16;
            o = make_uint4(u[0] | (u[1] << 16), u[2] | (u[3] << 16), u[4] | (u[5] << 16), u[6] | (u[7] << 16));
        } else {
            o = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        }
        out[idx] = o;
    }
}

}  // namespace

extern "C" long long idv_lstm_ih_bf16_bytes(int H, int K) { return (long long)(8 * H / 32) * (K / 16) * 2 * 64 * 16; }

extern "C" int idv_lstm_proj_bf16_supported(int H, int K) { return (H > 0 && K > 0 && (8 * H) % 256 == 0 && K % 64 == 0) ? 1 : 0; }

extern "C" int idv_pack_lstm_ih_bf16(const float* w_ih_re, const float* w_ih_im, int H, int K, void* wfrag, void* stream) {
    if (!w_ih_re || !w_ih_im || !wfrag || !idv_lstm_proj_bf16_supported(H, K)) return IDV_EINVAL;
    const long long n = (long long)(8 * H / 32) * (K / 16) * 2 * 64;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_lstm_ih_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_ih_re, w_ih_im, H, K,
                       (uint4*)wfrag);
    return idv_launch_status();
}

// nplanes: planes of the image (a multiple of 8); planes >= nvalid are written as zeros (K padding)
extern "C" int idv_planar_to_kimage(const float* x, int nvalid, int nplanes, int J, int Jp, void* img, long long lo_off, void* stream) {
    if (!x || !img || nplanes <= 0 || (nplanes % 8) || nvalid <= 0 || nvalid > nplanes || J <= 0 || Jp < J || (lo_off % 8) || (reinterpret_cast<uintptr_t>(img) & 15))
        return IDV_EINVAL;
    const long long n = (long long)(nplanes / 8) * Jp;
    long long g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(planar_to_kimage_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, nvalid, nplanes, J, Jp,
                       (unsigned short*)img, lo_off);
    return idv_launch_status();
}

// G[part][(t, b)][8H] for part = real / imaginary input: the layer-0 input projection of idv_clstm_fwd (pass
// flags bit 1 there to use it).  ximg: K-major image of the 2*K input planes (idv_planar_to_kimage).
extern "C" int idv_lstm_proj_bf16x3(const void* ximg, long long lo_off_slots, int K, const void* wfrag_bf16, const float* bias,
                                    float* G, int H, int B, int T, int Tp, int Jp, void* stream) {
    if (!ximg || !wfrag_bf16 || !bias || !G || !idv_lstm_proj_bf16_supported(H, K) || B <= 0 || T <= 0 || Tp < T + 1 || Jp < B * Tp)
        return IDV_EINVAL;
    if (reinterpret_cast<uintptr_t>(ximg) & 15) return IDV_EINVAL;
    PwBf16Args a{};
    a.ximg = (const u32x4*)ximg; a.lo_off = lo_off_slots; a.KO = K / 8; a.part_stride = (long long)(K / 8) * Jp;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.t_valid = T; a.nB = B;
    a.wfrag = (const uint4*)wfrag_bf16; a.bias = bias; a.out = G;
    a.out_part_stride = (long long)T * B * 8 * H; a.ldo = 8 * H; a.M = 8 * H;
    // fewest idle workgroup slots in the last round of 256
    const long long mb = (a.M + 255) / 256;
    auto eff = [&](int jt) { const long long n = ((a.J + jt - 1) / jt) * mb; return (double)n / (double)(((n + 255) / 256) * 256); };
    hipStream_t st = (hipStream_t)stream;
    return eff(64) > eff(128) + 0.02 ? launch_pw<2, 2, true>(a, st) : launch_pw<4, 2, true>(a, st);
}

// Layer-1 input projection of the complex LSTM from the K-major split image of h0 that idv_lstm_rec_pers wrote (4 runs x
// H/8 octets): G1[run = 2z + s][(t, b)][4H] = W_ih1(set s) h0[run] + bias.  wfrag_bf16: idv_pack_lstm_ih_bf16(w_ih_l1 of
// both sets, H, K = H); bias: [2 sets][4H] in the recurrence's gate-column order (idv_pack_lstm_ih).
extern "C" int idv_lstm_proj1_bf16x3(const void* himg, long long lo_off_slots, const void* wfrag_bf16, const float* bias, float* G1,
                                     int H, int B, int T, int Tp, int Jp, void* stream) {
    if (!himg || !wfrag_bf16 || !bias || !G1 || !idv_lstm_proj_bf16_supported(H, H) || (4 * H) % 256 || B <= 0 || T <= 0 ||
        Tp < T + 1 || Jp < B * Tp)
        return IDV_EINVAL;
    if (reinterpret_cast<uintptr_t>(himg) & 15) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int KO = H / 8;
    for (int s = 0; s < 2; ++s) {
        PwBf16Args a{};
        a.ximg = (const u32x4*)himg + (long long)s * KO * Jp;       // runs s and 2 + s are this weight set's two parts
        a.lo_off = lo_off_slots; a.KO = KO; a.part_stride = 2LL * KO * Jp;
        a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.t_valid = T; a.nB = B;
        a.wfrag = (const uint4*)wfrag_bf16 + (long long)s * (4 * H / 32) * (H / 16) * 2 * 64;
        a.bias = bias + s * 4 * H; a.out = G1 + (long long)s * T * B * 4 * H;
        a.out_part_stride = 2LL * T * B * 4 * H; a.ldo = 4 * H; a.M = 4 * H;
        const long long mb = (a.M + 255) / 256;
        auto eff = [&](int jt) { const long long n = ((a.J + jt - 1) / jt) * mb; return (double)n / (double)(((n + 255) / 256) * 256); };
        const int rc = eff(64) > eff(128) + 0.02 ? launch_pw<2, 2, true>(a, st) : launch_pw<4, 2, true>(a, st);
        if (rc) return rc;
    }
    return IDV_OK;
}

namespace {

// generic [M][K] row-major fp32 -> fragments, K zero-padded to Kp (a multiple of 64), rows padded to 256
__global__ void pack_pw_bf16_kernel(const float* __restrict__ w, int M, int K, int Kp, int Mtiles, uint4* __restrict__ out) {
    const int NKB = Kp / 16;
    const long long n = (long long)Mtiles * NKB * 2 * 64;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long long t = idx >> 6;
        const int split = (int)(t & 1); t >>= 1;
        const int kb = (int)(t % NKB);
        const int mt = (int)(t / NKB);
        const int m = mt * 32 + (lane & 31);
        unsigned u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * kb + 8 * (lane >> 5) + j;
            const float v = (m < M && k < K) ? w[(size_t)m * K + k] : 0.f;
            const unsigned hi = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
            u[j] = split == 0 ? hi >> 16 : (pack_bf16(v - __builtin_bit_cast(float, hi), 0.f) & 0xffffu);
        }
        out[idx] = make_uint4(u[0] | (u[1] << 16), u[2] | (u[3] << 16), u[4] | (u[5] << 16), u[6] | (u[7] << 16));
    }
}

// STFT framing straight into the K-major image: slot (octet o, column j) = samples 8o..8o+7 of frame j
__global__ __launch_bounds__(256) void stft_frames_kimg_kernel(const float* __restrict__ x, int B, int L, int n_fft, int win,
                                                               int hop, int T, unsigned short* __restrict__ img, long long lo_off,
                                                               int KO, int Tp, int Jp) {
    extern __shared__ float seg[];
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int left = (n_fft - win) / 2, half = n_fft / 2;
    const int nt = min(32, T - t0);
    const int seglen = hop * (nt - 1) + win;
    const long long s0 = (long long)hop * t0 + left - half;
    for (int e = threadIdx.x; e < seglen; e += blockDim.x) {
        long long s = s0 + e;
        if (s < 0) s = -s;
        if (s >= L) s = 2LL * (L - 1) - s;
        seg[e] = (s >= 0 && s < L) ? x[(size_t)b * L + s] : 0.f;
    }
    __syncthreads();
    const int tl = threadIdx.x & 31;
    // 33 columns per block: the 32 frames and, in the first block of an utterance, its guard column
    for (int o = threadIdx.x >> 5; o < KO; o += 8) {
        for (int pass = 0; pass < 2; ++pass) {
            const bool guard = pass == 1;
            if (guard && (blockIdx.x != 0 || tl != 0)) continue;
            if (!guard && tl >= nt) continue;
            const size_t col = (size_t)b * Tp + (guard ? 0 : t0 + tl + 1);
            unsigned hw[4], lw[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int k0 = 8 * o + 2 * w;
                const float x0 = (!guard && k0 < win) ? seg[hop * tl + k0] : 0.f;
                const float x1 = (!guard && k0 + 1 < win) ? seg[hop * tl + k0 + 1] : 0.f;
                const unsigned u0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u;
                const unsigned u1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
                hw[w] = (u0 >> 16) | u1;
                lw[w] = pack_bf16(x0 - __builtin_bit_cast(float, u0), x1 - __builtin_bit_cast(float, u1));
            }
            unsigned short* d = img + ((size_t)o * Jp + col) * 8;
            *(uint4*)d = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *(uint4*)(d + lo_off) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    }
}

}  // namespace

extern "C" long long idv_pw_bf16_wfrag_bytes(int M, int K) {
    return (long long)((M + 255) / 256 * 8) * ((K + 63) / 64 * 4) * 2 * 64 * 16;
}

extern "C" int idv_pack_pw_bf16(const float* w, int M, int K, void* wfrag, void* stream) {
    if (!w || !wfrag || M <= 0 || K <= 0) return IDV_EINVAL;
    const int Kp = (K + 63) / 64 * 64, Mtiles = (M + 255) / 256 * 8;
    const long long n = (long long)Mtiles * (Kp / 16) * 2 * 64;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_pw_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, M, K, Kp, Mtiles, (uint4*)wfrag);
    return idv_launch_status();
}

// out[m][Jp] (planar, guard columns zero) = sum_k W[m][k] * x[k][j] + bias[m]; ximg: K-major image with Kp/8 octets
// (Kp = K rounded up to 64, the padding octets hold zeros or anything finite -- their weights are zero)
extern "C" int idv_pw_bf16x3(const void* ximg, long long lo_off_slots, int K, const void* wfrag_bf16, const float* bias, float* out,
                             int M, int B, int Tp, int Jp, int t_valid, void* stream) {
    if (!ximg || !wfrag_bf16 || !out || K <= 0 || M <= 0 || B <= 0 || Tp <= 1 || Jp < B * Tp || (reinterpret_cast<uintptr_t>(ximg) & 15))
        return IDV_EINVAL;
    PwBf16Args a{};
    const int Kp = (K + 63) / 64 * 64;
    a.ximg = (const u32x4*)ximg; a.lo_off = lo_off_slots; a.KO = Kp / 8; a.part_stride = 0;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.t_valid = t_valid; a.nB = B;
    a.wfrag = (const uint4*)wfrag_bf16; a.bias = bias; a.out = out; a.out_part_stride = 0; a.ldo = 0; a.M = M;
    const long long mb = (M + 255) / 256;
    auto eff = [&](int jt) { const long long n = ((a.J + jt - 1) / jt) * mb; return (double)n / (double)(((n + 255) / 256) * 256); };
    hipStream_t st = (hipStream_t)stream;
    return eff(64) > eff(128) + 0.02 ? launch_pw<2, 1, false>(a, st) : launch_pw<4, 1, false>(a, st);
}

// idv_pw_bf16x3 with the transposed store of the LSTM gate layout: out[(t * B + b) * ldo + m] for the valid (t, b) -- the
// gradient arriving at a layer's output, dh = W^T dG, in the row-major form idv_lstm_bptt reads (bf16x3 training mode)
extern "C" int idv_pw_bf16x3_rows(const void* ximg, long long lo_off_slots, int K, const void* wfrag_bf16, const float* bias,
                                  float* out_rows, int M, int ldo, int B, int T, int Tp, int Jp, void* stream) {
    if (!ximg || !wfrag_bf16 || !bias || !out_rows || K <= 0 || M <= 0 || ldo < M || B <= 0 || T <= 0 || Tp < T + 1 || Jp < B * Tp ||
        (reinterpret_cast<uintptr_t>(ximg) & 15))
        return IDV_EINVAL;
    PwBf16Args a{};
    const int Kp = (K + 63) / 64 * 64;
    a.ximg = (const u32x4*)ximg; a.lo_off = lo_off_slots; a.KO = Kp / 8; a.part_stride = 0;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.t_valid = T; a.nB = B;
    a.wfrag = (const uint4*)wfrag_bf16; a.bias = bias; a.out = out_rows; a.out_part_stride = 0; a.ldo = ldo; a.M = M;
    const long long mb = (M + 255) / 256;
    auto eff = [&](int jt) { const long long n = ((a.J + jt - 1) / jt) * mb; return (double)n / (double)(((n + 255) / 256) * 256); };
    hipStream_t st = (hipStream_t)stream;
    return eff(64) > eff(128) + 0.02 ? launch_pw<2, 1, true>(a, st) : launch_pw<4, 1, true>(a, st);
}

extern "C" int idv_stft_frames_kimage(const float* x, int B, int L, int n_fft, int win, int hop, int T, void* img, long long lo_off,
                                      int Tp, int Jp, void* stream) {
    if (!x || !img || B <= 0 || L <= n_fft / 2 || T != 1 + L / hop || Tp < T + 1 || Jp < B * Tp || (lo_off % 8)) return IDV_EINVAL;
    const int KO = (win + 63) / 64 * 8;
    const size_t smem = (size_t)(hop * 31 + win) * sizeof(float);
    hipLaunchKernelGGL(stft_frames_kimg_kernel, dim3((T + 31) / 32, B), dim3(256), smem, (hipStream_t)stream, x, B, L, n_fft, win,
                       hop, T, (unsigned short*)img, lo_off, KO, Tp, Jp);
    return idv_launch_status();
}
