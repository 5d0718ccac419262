// Persistent cooperative recurrence for the VAE encoders' hidden sizes (H = 384 = 3*zdim, H = 768 = 6*zdim; reference
// model/pvae_module.py:1819, :2160-2163 feeding ComplexLSTM.forward, model/complex_progress.py:50-74) in EXACT fp32 -- the
// reference's precision.  Twin of lstm_pers.hip (split-bf16): one launch per layer instead of one launch per time step.
//
// H/16 workgroups per weight set x batch chunks, one per CU (<= idv_coop_max_workgroups()), each owning the (i, f, g, o) columns
// of 16 hidden units: its W_hh slice (64 columns x H fp32 = 96 / 192 VGPRs per lane, the 4 waves split K) stays in registers
// for all T steps; gates += h_{t-1} W_hh^T on v_mfma_f32_16x16x4_f32 (the A operand loaded as 8 consecutive k per lane, two
// 16-byte loads, the B fragments gathered once to match that k order).  h_t is exchanged as fp32 in 16-byte granules with the
// fence-free hand-off of lstm_pers.hip: write-through `sc1` stores drained by every storing wave, one agent-scope add per
// workgroup to all replicas of its group's arrive counter behind the workgroup barrier, an `sc1` poll of one replica, `sc1`
// buffer loads behind the poll and the barrier.  Every spin is bounded; on a time-out all workgroups drain, the outputs are
// poisoned with NaN and the sticky status word is raised (coop.hpp).
//
// What bounds it: the fp32 MFMA rate.  H = 768, B = 32: 16 x 2 tiles x 64 columns x 768 MACs per workgroup per step = 384
// MFMAs of 32 cycles per wave = 5.1 us, plus ~2 us of hand-off latencies (the split-bf16 twin: 5.5 us in total); H = 384,
// B = 32: 1.3 us of MFMA.  The per-step kernels it replaces (lstm_step_kernel) take 11 us per step.
#include <cstdlib>
#include "common.hpp"
#include "coop.hpp"
#include "../../include/idccrn_hip.h"

namespace idv_pers32 {

typedef int v4i __attribute__((ext_vector_type(4)));

struct Pers32Args {
    const float* g;           // gate pre-activations (hoisted input projection), addressing as lstm.hip RecArgs
    long long g_run_z, g_run_s;
    int ldg;
    const float* whh;         // idv_pack_lstm_hh fp32 fragments, vec4 order: [set][tile = ub*4 + gate][kk/4][lane][4]
    float* hout;              // [4 runs][T*B][H]
    float* gsave;             // training: activated gates (i, f, g, o) over the pre-activations (== g), or nullptr
    float* csave;             // training: cell state per step [4 runs][T*B][H], or nullptr
    float* hx;                // exchange [2 parity][4 runs][Bpad][H] fp32
    unsigned hx_bytes;
    unsigned* sync;           // [abort flag: 256 B][group = set * chunks + chunk][replica][256 B] arrive counters
    int nrep;
    int H, B, T, Bpad, nchunks;
    int fault;                // test hook (IDV_COOP_FAULT=1): workgroup (0, 0, 0) never arrives -> the bounded spins must abort
    unsigned* status;         // host-mapped sticky status word (coop.hpp) or nullptr
};

constexpr unsigned long long SPIN_LIMIT_TICKS = 40000000ull;     // 0.4 s of the 100 MHz wall clock

template <int KBW, int NRT>     // 32-k blocks per wave = H/128; 16-row tiles per workgroup
__global__ __launch_bounds__(256, 1) void lstm_pers_f32_kernel(const Pers32Args a) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][NRT][4 gates][4 rows r][64 lanes]
    __shared__ int abort_sh;
    __shared__ __attribute__((aligned(16))) float stage[NRT][16][16];          // h_t of this workgroup: [tile][row][unit]
    const __amdgpu_buffer_rsrc_t hxr = __builtin_amdgcn_make_buffer_rsrc((void*)a.hx, 0, a.hx_bytes, 0x00020000);
    const int H = a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = blockIdx.x, s = blockIdx.y, ch = blockIdx.z;
    const int nslice = gridDim.x;
    const int col = lane & 15, rq = lane >> 4;
    const int TPR = a.Bpad / 16, NT = 2 * TPR;
    unsigned* abortf = a.sync;
    unsigned* counter0 = a.sync + 64 + (size_t)((s * a.nchunks + ch) * a.nrep) * 64;
    unsigned* counter = counter0 + (size_t)(sl & (a.nrep - 1)) * 64;
    const size_t TBH = (size_t)a.T * a.B * H;

    // this workgroup's tiles of the (2 runs of the weight set) x ceil(B/16) tile space: run (2 z + s), first row, validity
    int t_run[NRT], t_b0[NRT];
    bool t_ok[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        int tile = ch * NRT + rt;
        t_ok[rt] = tile < NT;
        if (tile >= NT) tile = NT - 1;
        const int z = tile / TPR;
        t_run[rt] = 2 * z + s;
        t_b0[rt] = (tile - z * TPR) * 16;
    }

    // W_hh slice: gate tiles (sl*4 + g), this wave's k-blocks; MFMA k-step j of block kb is k = 32 (wave KBW + kb) + 8 rq + j
    // for this lane (so that the A operand is 8 consecutive floats): gathered once from the vec4 fragment order
    float breg[4][KBW][8];
    {
        const float* wb = a.whh + (size_t)s * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * (wave * KBW + kb) + 8 * rq + j;
                    const int kk = k >> 2, fl = (k & 3) * 16 + col;          // fragment k-step and lane that hold (k, col)
                    breg[g][kb][j] = wb[(((size_t)(sl * 4 + g) * (H / 16) + (kk >> 2)) * 64 + fl) * 4 + (kk & 3)];
                }
    }
    // cell update split by ROW over the waves: wave w owns rows rq * 4 + w of every tile (unit = lane & 15)
    float creg[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) creg[rt] = 0.f;
    const int myrow = rq * 4 + wave;

    bool aborted = false;
    if (tid == 0) abort_sh = 0;
    if (a.fault && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) return;      // injected failure (tests only)
    for (int t = 0; t < a.T; ++t) {
        // ---- inputs of the cell update (independent of h): issue first
        float gpre[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int z = t_run[rt] >> 1;
            const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
            int b = t_b0[rt] + myrow;
            if (b >= a.B) b = a.B - 1;
            const float* gp = g + ((size_t)t * a.B + b) * a.ldg + sl * 64 + col;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) gpre[rt][gg] = gp[16 * gg];
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
                        if (wall_clock64() - t0 > SPIN_LIMIT_TICKS) {
                            __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            abort_sh = 1;
                            break;
                        }
                    }
                }
            }
            // no acquire fence: every byte of the exchange buffer was stored sc1 and drained before the arrive, and every
            // load of it below is an sc1 buffer load issued after this barrier, which the polling wave joins after its match
            __syncthreads();
            if (abort_sh) { aborted = true; break; }
            const unsigned par_r = (unsigned)((t - 1) & 1) * 4u * (unsigned)a.Bpad * (unsigned)H * 4u;
            // every row tile's A fragments in flight at once where the registers allow, else tile rt+1 loads while rt multiplies
            constexpr bool ALL_UP = (NRT * KBW <= 12);
            constexpr int NB = ALL_UP ? NRT : 2;
            f32x4 av[NB][KBW][2];
            auto load_a = [&](int rt, f32x4 (&d)[KBW][2]) {
                const unsigned rowoff = (((unsigned)t_run[rt] * a.Bpad + t_b0[rt] + col) * (unsigned)H + 8 * rq) * 4u;
#pragma unroll
                for (int kb = 0; kb < KBW; ++kb) {
                    const unsigned ko = rowoff + 128u * (unsigned)(wave * KBW + kb);
                    d[kb][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, ko, par_r, 16));
                    d[kb][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, ko + 16u, par_r, 16));
                }
            };
            if (ALL_UP) {
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) load_a(rt, av[rt % NB]);
            } else {
                load_a(0, av[0]);
            }
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                if (!ALL_UP && rt + 1 < NRT) load_a(rt + 1, av[(rt + 1) % NB]);
                __builtin_amdgcn_sched_barrier(0);          // the prefetch is issued BEFORE this row tile's MFMAs
#pragma unroll
                for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float ak = av[rt % NB][kb][j >> 2][j & 3];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak, breg[g][kb][j], acc[rt][g], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- reduce the 4 K-partials through LDS; layout [wave][tile][gate][r][lane]: conflict-free dword writes and reads
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((((wave * NRT + rt) * 4 + g) * 4 + r) << 6) + lane] = acc[rt][g][r];
        __syncthreads();
        const unsigned par_w = (unsigned)(t & 1) * 4u * (unsigned)a.Bpad * (unsigned)H * 4u;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float gate[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = gpre[rt][g];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[((((w * NRT + rt) * 4 + g) * 4 + wave) << 6) + lane];
                gate[g] = v;
            }
            const float ig = sigmoidf_(gate[0]), fg = sigmoidf_(gate[1]);
            const float gv = tanhf_(gate[2]), og = sigmoidf_(gate[3]);
            const float cn = fg * creg[rt] + ig * gv;
            creg[rt] = cn;
            const float hv = og * tanhf_(cn);
            const int b = t_b0[rt] + myrow;
            if (a.gsave && t_ok[rt] && b < a.B) {          // what idv_lstm_bptt reads: the layout the per-step kernels leave
                const int z = t_run[rt] >> 1;
                float* gp = a.gsave + z * a.g_run_z + s * a.g_run_s + ((size_t)t * a.B + b) * a.ldg + sl * 64 + col;
                gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                a.csave[(size_t)t_run[rt] * TBH + ((size_t)t * a.B + b) * H + sl * 16 + col] = cn;
            }
            stage[rt][myrow][col] = hv;                    // transpose through LDS: a 16-byte store wants 4 units of a row
        }
        __syncthreads();
        if (wave < NRT && t_ok[wave < NRT ? wave : 0]) {
            // (row, 4-unit granule) = (lane >> 2, lane & 3); write-through (sc1) to the exchange buffer, plain to hout
            const int rt = wave;
            const int row = lane >> 2, c4 = lane & 3;
            const v4i pk = *(const v4i*)&stage[rt][row][c4 * 4];
            const unsigned off = (((unsigned)t_run[rt] * a.Bpad + t_b0[rt] + row) * (unsigned)H + sl * 16 + c4 * 4) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(pk, hxr, off, par_w, 16);       // aux 16 = sc1
            const int b = t_b0[rt] + row;
            if (b < a.B) *(v4i*)&a.hout[(size_t)t_run[rt] * TBH + ((size_t)t * a.B + b) * H + sl * 16 + c4 * 4] = pk;
        }
        // ---- publish: every storing wave drains its stores, the workgroup meets, ONE wave instruction arrives
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < a.nrep) __hip_atomic_fetch_add(counter0 + (size_t)tid * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (aborted) {
        // poison this workgroup's outputs: a timed-out recurrence must never look like a result
        if (tid == 0) idv_coop_raise(a.status);
        const float qnan = __builtin_nanf("");
        for (int rt = 0; rt < NRT; ++rt) {
            if (!t_ok[rt]) continue;
            for (long long e = tid; e < (long long)a.T * 16 * 16; e += 256) {
                const int u = (int)(e & 15), br = (int)((e >> 4) & 15);
                const long long t = e >> 8;
                const int b = t_b0[rt] + br;
                if (b < a.B) a.hout[(size_t)t_run[rt] * TBH + ((size_t)t * a.B + b) * H + sl * 16 + u] = qnan;
            }
        }
    }
}

// 16-row tiles per workgroup: as few as the residency bound allows (more CUs share the fp32 MFMA work)
inline int nrt_for(int H, int B) {
    const int NT = 2 * ((B + 15) / 16);
    for (int nrt = 1; nrt <= 4; nrt *= 2)
        if (2 * (H / 16) * ((NT + nrt - 1) / nrt) <= idv_coop_max_workgroups()) return nrt;
    return 0;
}

constexpr int SYNC_BYTES = 256 + 20 * 8 * 256;     // as lstm_pers.hip: abort flag + (<= 20 groups) x (<= 8 replicas) x 256 B

}  // namespace idv_pers32

extern "C" int idv_lstm_pers_f32_supported(int H, int B) {
    static const bool on = [] { const char* e = getenv("IDV_LSTM_PERS_F32"); return !e || e[0] != '0'; }();
    if (!on || (H != 384 && H != 768) || B <= 0) return 0;
    const int nrt = idv_pers32::nrt_for(H, B);
    if (nrt <= 0) return 0;
    const int NT = 2 * ((B + 15) / 16);
    return 2 * ((NT + nrt - 1) / nrt) <= 20;
}

extern "C" long long idv_lstm_pers_f32_work_bytes(int H, int B) {
    const long long Bpad = (B + 15) / 16 * 16;
    return idv_pers32::SYNC_BYTES + 2LL * 4 * Bpad * H * 4;       // [abort flag + arrive counters, zeroed per call][exchange]
}

// one layer of the recurrence in exact fp32; arguments as idv_lstm_rec_coop_f32 (hout required; gsave == g / csave for the
// training forward or both NULL); whh_frag: idv_pack_lstm_hh; work: idv_lstm_pers_f32_work_bytes(H, B) bytes, 16-byte aligned
extern "C" int idv_lstm_rec_pers_f32(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh_frag, float* hout,
                                     int H, int B, int T, void* work, float* gsave, float* csave, void* stream) {
    using namespace idv_pers32;
    if (!g || !whh_frag || !hout || !work || T <= 0 || !idv_lstm_pers_f32_supported(H, B)) return IDV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(work) & 15) || (gsave != nullptr) != (csave != nullptr) || (gsave && gsave != g)) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nrt = nrt_for(H, B);
    const int TPR = (B + 15) / 16, NT = 2 * TPR;
    const int chunks = (NT + nrt - 1) / nrt;
    const long long Bpad = 16LL * TPR;
    Pers32Args a{};
    a.g = g; a.g_run_z = g_run_z; a.g_run_s = g_run_s; a.ldg = ldg;
    a.whh = whh_frag; a.hout = hout; a.gsave = gsave; a.csave = csave;
    a.sync = (unsigned*)work;
    a.hx = (float*)((char*)work + SYNC_BYTES);
    a.hx_bytes = (unsigned)(2LL * 4 * Bpad * H * 4);
    a.nrep = 8;
    a.H = H; a.B = B; a.T = T; a.Bpad = (int)Bpad; a.nchunks = chunks;
    { const char* e = getenv("IDV_COOP_FAULT"); a.fault = (e && e[0] == '1') ? 1 : 0; }
    a.status = idv_coop_status_word();
    typedef void (*kern_t)(const Pers32Args);
    kern_t k;
    if (H == 384) k = nrt == 1 ? (kern_t)lstm_pers_f32_kernel<3, 1> : (nrt == 2 ? (kern_t)lstm_pers_f32_kernel<3, 2> : (kern_t)lstm_pers_f32_kernel<3, 4>);
    else          k = nrt == 1 ? (kern_t)lstm_pers_f32_kernel<6, 1> : (nrt == 2 ? (kern_t)lstm_pers_f32_kernel<6, 2> : (kern_t)lstm_pers_f32_kernel<6, 4>);
    // at least 84 KB of LDS per workgroup: ONE workgroup per CU whatever the register count
    size_t smem = (size_t)4 * nrt * 4 * 4 * 64 * sizeof(float);
    if (smem < 84 * 1024) smem = 84 * 1024;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return IDV_ELAUNCH;
    if (hipMemsetAsync(work, 0, SYNC_BYTES, st) != hipSuccess) return IDV_ELAUNCH;
    int rc = idv_coop_chain_begin(st);
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(H / 16, 2, chunks), dim3(256), smem, st, a);
    if ((rc = idv_coop_chain_end(st))) return rc;
    return idv_launch_status();
}
