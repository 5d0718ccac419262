// Back-propagation through time of one H = 128 LSTM layer (DCCRN-CL bottleneck) as ONE cooperative launch: the per-step
// kernels of lstm_bwd.hip (same decomposition: a workgroup owns 32 hidden units and contracts over all 4H gate columns,
//   dh_t = dh_out[t] + dA_{t+1} W_hh ,   cell backward -> dA_t written over the saved gates)
// with W_hh^T resident in registers (64 VGPRs per lane), the running dc in registers, and dA_t exchanged between the four
// workgroups of a (run, 16-sequence tile) through global memory with the fence-free hand-off of lstm_pers.hip /
// lstm_coop_f32.hip instead of a kernel boundary per step (6.2 us per step -> see DESIGN 3.5).  What torch.autograd runs for
// nn.LSTM behind `loss.backward()` in the reference (model/complex_progress.py:50-74; supervised_dccrn/train.py:239-243).
#include <cstdlib>
#include "common.hpp"
#include "coop.hpp"
#include "../../include/idccrn_hip.h"


namespace idv_bcoop {

typedef int v4i __attribute__((ext_vector_type(4)));

struct BCoopArgs {
    float* g;             // activated gates in, pre-activation gate gradients out (same addressing as the forward's g)
    long long g_run_z, g_run_s;
    int ldg;
    const float* c;       // [4][T*B][H] cell states
    const float* dhout;   // [4][T*B][H] gradient arriving at h_t from above
    const float* whhT;    // idv_pack_lstm_hh_bwd: [set][unit tile][blk][lane][4]
    float* hx;            // exchange [2 parity][4 runs][Bpad][4H] fp32: dA_t, row-major in the gate-column order colp
    unsigned hx_bytes;
    unsigned* sync;       // [abort flag: 256 B][group = run * tiles + tile][replica][256 B]
    int nrep;
    int B, T, Bpad, tiles;
    unsigned* status;     // host-mapped sticky status word (coop.hpp) or nullptr
    int fault;
};

constexpr int H = 128, NSL = 4;
constexpr unsigned long long SPIN_LIMIT_TICKS = 40000000ull;     // 0.4 s of the 100 MHz wall clock

__global__ __launch_bounds__(256, 1) void lstm_bptt_coop_f32_kernel(const BCoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];                // [4 waves][2 tiles][4 r][64 lanes]
    __shared__ int abort_sh;
    __shared__ __attribute__((aligned(16))) float stage[16][128];              // dA_t of this workgroup: [row][its 128 colp]
    const __amdgpu_buffer_rsrc_t hxr = __builtin_amdgcn_make_buffer_rsrc((void*)a.hx, 0, a.hx_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = blockIdx.x, run = blockIdx.y, tile = blockIdx.z;
    const int z = run >> 1, s = run & 1;
    const int col = lane & 15, rq = lane >> 4;
    const int b0 = tile * 16;
    unsigned* abortf = a.sync;
    unsigned* counter0 = a.sync + 64 + (size_t)((run * a.tiles + tile) * a.nrep) * 64;
    unsigned* counter = counter0 + (size_t)(sl & (a.nrep - 1)) * 64;
    const size_t TBH = (size_t)a.T * a.B * H;
    float* g = a.g + z * a.g_run_z + s * a.g_run_s;
    const float* cst = a.c + (size_t)run * TBH;
    const float* dho = a.dhout + (size_t)run * TBH;

    // W_hh^T slice: output units of tiles 2 sl, 2 sl + 1; this wave's 128 gate columns k in four chunks of 32; lane (row, kq)
    // holds k = 128 w + 32 c + 8 kq + j as MFMA k-step j of chunk c (A operands are two 16-byte loads per chunk); gathered
    // from the packed blob, where element (unit tile, k, n) sits at (((tile * 32 + k / 16) * 64 + (k % 4) * 16 + n) * 4 + (k / 4) % 4
    float breg[2][4][8];
    {
        const float* wb = a.whhT + (size_t)s * (H / 16) * (H / 4) * 64 * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 128 * wave + 32 * c + 8 * rq + j;
                    breg[u][c][j] = wb[((((size_t)(2 * sl + u) * (H / 4) + (k >> 4)) * 64 + (k & 3) * 16 + col) << 2) + ((k >> 2) & 3)];
                }
    }
    // cell backward split by row over the waves (as the forward): wave w owns row rq * 4 + w, units 16 ub + (lane & 15)
    const int myrow = rq * 4 + wave;
    const int brow = b0 + myrow;
    const bool ok = brow < a.B;
    const int bclamp = ok ? brow : a.B - 1;
    float dcreg[2] = {0.f, 0.f};

    bool aborted = false;
    if (tid == 0) abort_sh = 0;
    if (a.fault && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) return;      // injected failure (tests only)
    for (int t = a.T - 1; t >= 0; --t) {
        const int step = a.T - 1 - t;                // 0, 1, ...
        // inputs of the cell backward (independent of the contraction): issue first
        float gi[2], gf[2], gg_[2], go[2], cc[2], cp[2], dhv[2];
        const size_t row = (size_t)t * a.B + bclamp;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ub = 2 * sl + u, unit = ub * 16 + col;
            const float* gp = g + row * a.ldg + ub * 64 + col;
            gi[u] = gp[0]; gf[u] = gp[16]; gg_[u] = gp[32]; go[u] = gp[48];
            cc[u] = cst[row * H + unit];
            cp[u] = (t > 0) ? cst[(row - a.B) * H + unit] : 0.f;
            dhv[u] = ok ? dho[row * H + unit] : 0.f;
        }
        f32x4 acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[u][r] = 0.f;
        if (step > 0) {
            if (tid == 0) {
                const unsigned want = (unsigned)step * (unsigned)NSL;
                const unsigned long long t0 = wall_clock64();
                unsigned long long spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023) == 0) {
                        if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abort_sh = 1; break; }
                        if (wall_clock64() - t0 > SPIN_LIMIT_TICKS) {
                            __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            abort_sh = 1;
                            break;
                        }
                    }
                }
            }
            __syncthreads();                 // the polling wave joins after its match; every load below is sc1
            if (abort_sh) { aborted = true; break; }
            // dA_{t+1} of the whole group: row = lane & 15, 8 consecutive gate columns per lane and chunk
            const unsigned par_r = (unsigned)((step - 1) & 1) * 4u * (unsigned)a.Bpad * 512u * 4u;
            const unsigned off = (((unsigned)run * a.Bpad + b0 + col) * 512u + 128 * wave + 8 * rq) * 4u;
            f32x4 av[4][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                av[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off + c * 128u, par_r, 16));
                av[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off + c * 128u + 16u, par_r, 16));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 h4 = av[c][j >> 2];
                    const float aj = (j & 3) == 0 ? h4[0] : ((j & 3) == 1 ? h4[1] : ((j & 3) == 2 ? h4[2] : h4[3]));
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, breg[u][c][j], acc[u], 0, 0, 0);
                }
        }
        // ---- reduce the 4 K-partials through LDS: [wave][tile][r][lane]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * 2 + u) * 4 + r) << 6) + lane] = acc[u][r];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float dhr = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) dhr += red[(((w * 2 + u) * 4 + wave) << 6) + lane];
            const float dh = dhv[u] + dhr;
            const float tc = tanhf_(cc[u]);
            const float d_o = dh * tc;
            const float dc = dcreg[u] + dh * go[u] * (1.f - tc * tc);
            const float d_i = dc * gg_[u], d_g = dc * gi[u], d_f = dc * cp[u];
            float ai = d_i * gi[u] * (1.f - gi[u]);
            float af = d_f * gf[u] * (1.f - gf[u]);
            float ag = d_g * (1.f - gg_[u] * gg_[u]);
            float ao = d_o * go[u] * (1.f - go[u]);
            dcreg[u] = dc * gf[u];
            if (ok) {
                float* gp = g + ((size_t)t * a.B + brow) * a.ldg + (2 * sl + u) * 64 + col;
                gp[0] = ai; gp[16] = af; gp[32] = ag; gp[48] = ao;
            } else {
                ai = af = ag = ao = 0.f;            // padded rows publish zeros
            }
            float* sp = &stage[myrow][u * 64 + col];
            sp[0] = ai; sp[16] = af; sp[32] = ag; sp[48] = ao;
        }
        __syncthreads();
        {
            // 16 rows x 32 float4 = 512 stores, two per thread: write-through (sc1) to the exchange buffer
            const unsigned par_w = (unsigned)(step & 1) * 4u * (unsigned)a.Bpad * 512u * 4u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = tid + 256 * h, r = e >> 5, c4 = e & 31;
                const v4i pk = *(const v4i*)&stage[r][c4 * 4];
                const unsigned off = (((unsigned)run * a.Bpad + b0 + r) * 512u + sl * 128 + c4 * 4) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(pk, hxr, off, par_w, 16);       // aux 16 = sc1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < a.nrep) __hip_atomic_fetch_add(counter0 + (size_t)tid * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (aborted) {
        // poison this workgroup's gate gradients: a timed-out BPTT must never look like a result
        if (tid == 0) idv_coop_raise(a.status);
        const float qnan = __builtin_nanf("");
        for (long long e = tid; e < (long long)a.T * 16 * 128; e += 256) {
            const int cidx = (int)(e & 127), br = (int)((e >> 7) & 15);
            const long long t = e >> 11;
            if (b0 + br < a.B) g[((size_t)t * a.B + b0 + br) * a.ldg + sl * 128 + cidx] = qnan;
        }
    }
}

constexpr int SYNC_BYTES = 256 + 64 * 8 * 256;

}  // namespace idv_bcoop

extern "C" int idv_lstm_bptt_coop_supported(int H, int B) {
    static const bool on = [] { const char* e = getenv("IDV_LSTM_BPTT_COOP"); return !e || e[0] != '0'; }();
    if (!on || H != 128 || B <= 0) return 0;
    return 4 * 4 * ((B + 15) / 16) <= idv_coop_max_workgroups();      // every workgroup resident at once, one per CU
}

extern "C" long long idv_lstm_bptt_coop_work_bytes(int H, int B) {
    const long long Bpad = (B + 15) / 16 * 16;
    return idv_bcoop::SYNC_BYTES + 2LL * 4 * Bpad * 4 * H * 4;
}

// idv_lstm_bptt as one cooperative launch (H = 128); work: idv_lstm_bptt_coop_work_bytes(H, B) bytes, 16-byte aligned
extern "C" int idv_lstm_bptt_coop(float* gates, long long g_run_z, long long g_run_s, int ldg, const float* cstates,
                                  const float* dhout, const float* whhT, int H, int B, int T, void* work, void* stream) {
    using namespace idv_bcoop;
    if (!gates || !cstates || !dhout || !whhT || !work || T <= 0 || ldg < 4 * H || !idv_lstm_bptt_coop_supported(H, B)) return IDV_EINVAL;
    if (reinterpret_cast<uintptr_t>(work) & 15) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (B + 15) / 16;
    const long long Bpad = 16LL * tiles;
    if (hipMemsetAsync(work, 0, SYNC_BYTES, st) != hipSuccess) return IDV_ELAUNCH;
    BCoopArgs a{};
    a.g = gates; a.g_run_z = g_run_z; a.g_run_s = g_run_s; a.ldg = ldg;
    a.c = cstates; a.dhout = dhout; a.whhT = whhT;
    a.sync = (unsigned*)work;
    a.hx = (float*)((char*)work + SYNC_BYTES);
    a.hx_bytes = (unsigned)(2LL * 4 * Bpad * 4 * H * 4);
    a.nrep = 4;
    a.B = B; a.T = T; a.Bpad = (int)Bpad; a.tiles = tiles;
    { const char* e = getenv("IDV_COOP_FAULT"); a.fault = (e && e[0] == '1') ? 1 : 0; }
    a.status = idv_coop_status_word();
    const size_t smem = 84 * 1024;                   // one workgroup per CU
    if (hipFuncSetAttribute((const void*)lstm_bptt_coop_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    int rc = idv_coop_chain_begin(st);
    if (rc) return rc;
    hipLaunchKernelGGL(lstm_bptt_coop_f32_kernel, dim3(NSL, 4, tiles), dim3(256), smem, st, a);
    if ((rc = idv_coop_chain_end(st))) return rc;
    return idv_launch_status();
}
