// Persistent cooperative recurrence for the hidden sizes of the VAE encoders (H = 384 = 3*zdim, H = 768 = 6*zdim;
// reference model/pvae_module.py:1819, :2160-2163 feeding ComplexLSTM.forward, model/complex_progress.py:50-74).
//
// W_hh of one weight set (2.4 / 9.4 MB) does not fit one CU, so the 4H gate columns are spread over H/16 workgroups per
// weight set, each holding the (i, f, g, o) columns of 16 hidden units for ALL T steps in registers as split-bf16 MFMA
// fragments (96 / 192 VGPRs per lane) -- one launch per layer instead of one launch per time step.  Per step a workgroup
//   1. waits until every workgroup of its group (same weight set, same batch chunk) has published h_{t-1}
//      (monotonic arrive counter in global memory, release / acquire fences at agent scope: the 8 XCDs have private L2s),
//   2. reads h_{t-1} (all H units of its 2 runs x 16*RTR sequences, split-bf16, straight into MFMA A fragments; the 4
//      waves split K), contracts with its W_hh slice on v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate),
//   3. reduces the 4 K-partials through LDS, adds the hoisted input projection, runs the cell update with c kept in
//      registers, and publishes its 16 units of h_t (fp32 for the next layer, split-bf16 for the next step).
// Every spin is bounded: a workgroup that waits longer than ~0.4 s raises the abort flag, all workgroups drain, and the
// outputs are poisoned with NaN (no hang; results never silently wrong).
#include "common.hpp"
#include "../../include/idccrn_hip.h"

namespace idv_pers {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct PersArgs {
    const float* g;           // gate pre-activations (hoisted input projection)
    long long g_run_z, g_run_s;
    int ldg;
    const uint4* whh16;       // split-bf16 fragments of idv_pack_lstm_hh: [set][tile = ub*4 + gate][kb][hi|lo][lane]
    float* hout;              // [4 runs][T*B][H]
    unsigned short* hx;       // exchange [2 parity][hi|lo][4 runs][Bpad][H] bf16
    unsigned* sync;           // [2 sets * chunks] arrive counters, then [1] abort flag
    int H, B, T, Bpad, nchunks;
};

constexpr unsigned long long SPIN_LIMIT_CYCLES = 1000000000ull;     // ~0.4 s at 2.4 GHz

template <int KBW, int RTR>     // k-blocks (32) per wave = H/128; 16-row tiles per run per workgroup
__global__ __launch_bounds__(256, 1) void lstm_pers_kernel(const PersArgs a) {
    constexpr int NRT = 2 * RTR;                 // row tiles per workgroup (2 runs share a weight set)
    constexpr int RPW = (NRT + 3) / 4;           // row tiles a wave finalises
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][NRT][4 gates][64 lanes][4]
    __shared__ int abort_sh;
    const int H = a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = blockIdx.x, s = blockIdx.y, ch = blockIdx.z;
    const int nslice = gridDim.x;
    const int col = lane & 15, rq = lane >> 4;
    const int KB = H / 32;
    unsigned* counter = a.sync + s * a.nchunks + ch;
    unsigned* abortf = a.sync + 2 * a.nchunks;
    const size_t TBH = (size_t)a.T * a.B * H;
    const int b_base = ch * 16 * RTR;

    // W_hh slice: gate tiles (sl*4 + g), this wave's k-blocks
    uint4 wreg[4][KBW][2];
    {
        const uint4* wb = a.whh16 + (size_t)s * (H / 4) * KB * 2 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < KBW; ++k)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
                    wreg[g][k][sp] = wb[((size_t)((sl * 4 + g) * KB + wave * KBW + k) * 2 + sp) * 64];
    }
    float creg[RPW][4];
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) creg[q][r] = 0.f;

    bool aborted = false;
    if (tid == 0) abort_sh = 0;
    for (int t = 0; t < a.T; ++t) {
        // ---- inputs of the cell update (independent of h): issue first
        float gpre[RPW][4][4];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int rt = wave + 4 * q;
            if (rt < NRT) {
                const int z = rt / RTR, bt = rt - z * RTR;
                const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int b = b_base + bt * 16 + rq * 4 + r;
                    if (b >= a.B) b = a.B - 1;
                    const float* gp = g + ((size_t)t * a.B + b) * a.ldg + sl * 64 + col;
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) gpre[q][gg][r] = gp[16 * gg];
                }
            }
        }
        f32x4 acc[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][g][r] = 0.f;

        if (t > 0) {
            // ---- wait for h_{t-1} of the whole group
            if (tid == 0) {
                const unsigned want = (unsigned)t * (unsigned)nslice;
                const unsigned long long t0 = wall_clock64();
                unsigned long long spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023) == 0) {
                        if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abort_sh = 1; break; }
                        if (wall_clock64() - t0 > SPIN_LIMIT_CYCLES / 24) {     // wall_clock64 ticks at 100 MHz
                            __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            abort_sh = 1;
                            break;
                        }
                    }
                }
                // agent-scope acquire: invalidates this CU's vector cache and the stale lines of this XCD's L2, for every
                // wave of the workgroup (they are behind the barrier below) -- one thread, as in a cooperative grid sync
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (abort_sh) { aborted = true; break; }
            // ---- gates += h_{t-1} W_hh^T : A fragments straight from the exchange buffer (row = lane & 15, 8 k per lane)
            const unsigned short* hx = a.hx + (size_t)((t - 1) & 1) * 2 * 4 * a.Bpad * H;
            // A fragments of row tile rt+1 are in flight while row tile rt multiplies (explicit double buffer: left to
            // itself hipcc issues each (hi, lo) pair right before its 12 MFMAs, i.e. 4 * KBW serial memory round trips)
            uint4 ah[2][KBW], al[2][KBW];
            auto load_a = [&](int rt, uint4 (&h_)[KBW], uint4 (&l_)[KBW]) {
                const int z = rt / RTR, bt = rt - z * RTR;
                const int run = 2 * z + s;
                const size_t rowoff = ((size_t)run * a.Bpad + b_base + bt * 16 + col) * H;
#pragma unroll
                for (int k = 0; k < KBW; ++k) {
                    const size_t ko = rowoff + 32 * (wave * KBW + k) + 8 * rq;
                    h_[k] = *(const uint4*)(hx + ko);
                    l_[k] = *(const uint4*)(hx + (size_t)4 * a.Bpad * H + ko);
                }
            };
            load_a(0, ah[0], al[0]);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                if (rt + 1 < NRT) load_a(rt + 1, ah[(rt + 1) & 1], al[(rt + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);          // the prefetch is issued BEFORE this row tile's MFMAs
#pragma unroll
                for (int k = 0; k < KBW; ++k) {
                    const bf16x8_t vh = __builtin_bit_cast(bf16x8_t, ah[rt & 1][k]);
                    const bf16x8_t vl = __builtin_bit_cast(bf16x8_t, al[rt & 1][k]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, wreg[g][k][0]);
                        const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, wreg[g][k][1]);
                        f32x4 c = acc[rt][g];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, bh, c, 0, 0, 0);
                        acc[rt][g] = c;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- reduce the 4 K-partials through LDS
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) *(f32x4*)&red[(((wave * NRT + rt) * 4 + g) * 64 + lane) * 4] = acc[rt][g];
        __syncthreads();
        unsigned short* hxw = a.hx + (size_t)(t & 1) * 2 * 4 * a.Bpad * H;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int rt = wave + 4 * q;
            if (rt < NRT) {
                const int z = rt / RTR, bt = rt - z * RTR;
                const int run = 2 * z + s;
                f32x4 gate[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const f32x4 p = *(const f32x4*)&red[(((w * NRT + rt) * 4 + g) * 64 + lane) * 4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += p[r];
                    }
                    gate[g] = v;
                }
                const int unit = sl * 16 + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = b_base + bt * 16 + rq * 4 + r;
                    const float ig = sigmoidf_(gate[0][r] + gpre[q][0][r]), fg = sigmoidf_(gate[1][r] + gpre[q][1][r]);
                    const float gv = tanhf_(gate[2][r] + gpre[q][2][r]), og = sigmoidf_(gate[3][r] + gpre[q][3][r]);
                    const float cn = fg * creg[q][r] + ig * gv;
                    creg[q][r] = cn;
                    const float hv = og * tanhf_(cn);
                    if (b < a.B) a.hout[(size_t)run * TBH + ((size_t)t * a.B + b) * H + unit] = hv;
                    const __bf16 hh = (__bf16)hv;
                    const __bf16 hl = (__bf16)(hv - (float)hh);
                    const size_t o = ((size_t)run * a.Bpad + b) * H + unit;          // b < Bpad by construction
                    hxw[o] = __builtin_bit_cast(unsigned short, hh);
                    hxw[(size_t)4 * a.Bpad * H + o] = __builtin_bit_cast(unsigned short, hl);
                }
            }
        }
        // ---- publish: all stores of this workgroup, then release + arrive
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (aborted) {
        // poison this workgroup's outputs: a timed-out recurrence must never look like a result
        const float qnan = __builtin_nanf("");
        for (int z = 0; z < 2; ++z)
            for (long long e = tid; e < (long long)a.T * 16 * RTR * 16; e += 256) {
                const int u = (int)(e & 15);
                const long long rest = e >> 4;
                const int br = (int)(rest % (16 * RTR));
                const long long t = rest / (16 * RTR);
                const int b = b_base + br;
                if (b < a.B) a.hout[(size_t)(2 * z + s) * TBH + ((size_t)t * a.B + b) * H + sl * 16 + u] = qnan;
            }
    }
}

// 16-row tiles per run per workgroup: 2 (32 sequences per batch chunk; 1 for B <= 16).  More rows per workgroup would
// halve the workgroup count but not the per-step latency, which is what bounds the step (measured: CVAE, B = 64, H = 384:
// one chunk of 64 rows 675 utt/s, two chunks of 32 rows faster), and at H = 768 the registers do not allow more.
inline int rtr_for(int H, int B) { (void)H; return B <= 16 ? 1 : 2; }

}  // namespace idv_pers

extern "C" int idv_lstm_pers_supported(int H, int B) {
    if (H != 384 && H != 768) return 0;
    if (B <= 0) return 0;
    const int rtr = idv_pers::rtr_for(H, B);
    const int chunks = (B + 16 * rtr - 1) / (16 * rtr);
    return 2 * (H / 16) * chunks <= 240;          // every workgroup must be resident at once (256 CUs, one each)
}

extern "C" long long idv_lstm_pers_work_bytes(int H, int B) {
    const int rtr = idv_pers::rtr_for(H, B);
    const int chunks = (B + 16 * rtr - 1) / (16 * rtr);
    const long long Bpad = (long long)chunks * 16 * rtr;
    return 2LL * 2 * 4 * Bpad * H * 2 + (2LL * chunks + 2) * 4 + 64;
}

// one layer of the recurrence; work: idv_lstm_pers_work_bytes(H, B) bytes (16-byte aligned), contents arbitrary
extern "C" int idv_lstm_rec_pers(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh_frag, float* hout,
                                 int H, int B, int T, void* work, void* stream) {
    using namespace idv_pers;
    if (!g || !whh_frag || !hout || !work || T <= 0 || !idv_lstm_pers_supported(H, B)) return IDV_EINVAL;
    if (reinterpret_cast<uintptr_t>(work) & 15) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rtr = rtr_for(H, B);
    const int chunks = (B + 16 * rtr - 1) / (16 * rtr);
    const long long Bpad = (long long)chunks * 16 * rtr;
    const size_t hx_bytes = (size_t)2 * 2 * 4 * Bpad * H * 2;
    if (hipMemsetAsync(work, 0, (size_t)idv_lstm_pers_work_bytes(H, B), st) != hipSuccess) return IDV_ELAUNCH;
    PersArgs a{};
    a.g = g; a.g_run_z = g_run_z; a.g_run_s = g_run_s; a.ldg = ldg;
    a.whh16 = (const uint4*)(whh_frag + (size_t)2 * 4 * H * H);
    a.hout = hout;
    a.hx = (unsigned short*)work;
    a.sync = (unsigned*)((char*)work + hx_bytes);
    a.H = H; a.B = B; a.T = T; a.Bpad = (int)Bpad; a.nchunks = chunks;
    dim3 grid(H / 16, 2, chunks);
    const size_t smem = (size_t)4 * 2 * rtr * 4 * 64 * 4 * sizeof(float);
#define IDV_PERS_LAUNCH(KBW, RTR)                                                                                         \
    do {                                                                                                                  \
        auto k = lstm_pers_kernel<KBW, RTR>;                                                                              \
        if (smem > 64 * 1024 &&                                                                                           \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)     \
            return IDV_ELAUNCH;                                                                                           \
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a);                                                              \
    } while (0)
    if (H == 384) {
        if (rtr == 1) IDV_PERS_LAUNCH(3, 1); else IDV_PERS_LAUNCH(3, 2);
    } else {
        if (rtr == 1) IDV_PERS_LAUNCH(6, 1); else IDV_PERS_LAUNCH(6, 2);
    }
#undef IDV_PERS_LAUNCH
    return idv_launch_status();
}
