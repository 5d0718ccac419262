// Exact-fp32 recurrence of the DCCRN-CL bottleneck LSTM (H = 128; reference ComplexLSTM.forward,
// model/complex_progress.py:50-74) spread over FOUR CUs per 16-sequence tile.
//
// The register-resident kernel (lstm_rec_kernel<true>, lstm.hip) keeps a tile on ONE CU: 16 x 512 x 128 MACs per step on
// v_mfma_f32_16x16x4_f32 are 8192 MFMA cycles = 3.4 us, 6.0 us per step measured, 16 of 256 CUs busy at B = 64 -- 6.4 % of the
// fp32 headline step.  Here the 512 gate columns of a (run, tile) are split over 4 workgroups (32 hidden units x 4 gates
// each; the W_hh slice of 128 columns x 128 k is 64 VGPRs per lane, the 4 waves split K), and h_t is exchanged through
// global memory with the fence-free hand-off of lstm_pers.hip: write-through `sc1` 16-byte stores, drained by every storing
// wave, one agent-scope atomic add per workgroup behind the workgroup barrier, an `sc1` poll of one counter replica, `sc1`
// loads behind the poll and the barrier; one workgroup per CU (84 KB of LDS requested); bounded spins, NaN poison on time-out.
// Per step: 0.85 us of MFMA + ~2 us of hand-off latencies instead of 6 us.
#include <cstdlib>
#include <mutex>
#include "common.hpp"
#include "coop.hpp"
#include "../../include/idccrn_hip.h"

namespace idv_coop {

typedef int v4i __attribute__((ext_vector_type(4)));

struct CoopArgs {
    const float* g;           // gate pre-activations (hoisted input projection), addressing as lstm.hip RecArgs
    long long g_run_z, g_run_s;
    int ldg;
    const float* whh;         // idv_pack_lstm_hh fp32 fragments: [set][tile = ub*4 + gate][kk][lane]
    float* hout;              // [4 runs][T*B][H]
    float* gsave;             // training: activated gates over the pre-activations (== g), or nullptr
    float* csave;             // training: cell states [4 runs][T*B][H], or nullptr
    float* hx;                // exchange [2 parity][4 runs][Bpad][H] fp32
    unsigned hx_bytes;
    unsigned* sync;           // [abort flag: 256 B][group = run * tiles + tile][replica][256 B]
    int nrep;
    int B, T, Bpad, tiles;
    unsigned* status;         // host-mapped sticky status word (coop.hpp) or nullptr
    int fault;                // test hook (IDV_COOP_FAULT=1): workgroup (0, 0, 0) never arrives -> the bounded spins must abort
};

constexpr int H = 128, NSL = 4, UPW = 32;          // hidden size, workgroups per group, units per workgroup
constexpr unsigned long long SPIN_LIMIT_TICKS = 40000000ull;     // 0.4 s of the 100 MHz wall clock

__global__ __launch_bounds__(256, 1) void lstm_coop_f32_kernel(const CoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];                // [4 waves][8 tiles][4 r][64 lanes]
    __shared__ int abort_sh;
    __shared__ __attribute__((aligned(16))) float stage[16][UPW];              // h_t of this workgroup: [row][unit]
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
    const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
    float* gsv = a.gsave ? a.gsave + z * a.g_run_z + s * a.g_run_s : nullptr;

    // W_hh slice: this workgroup's 8 column tiles (unit blocks 2 sl, 2 sl + 1 x 4 gates), this wave's 32 k.  The A operand
    // is loaded as 8 CONSECUTIVE k per lane (two 16-byte loads), i.e. MFMA k-step j of lane (row, kq) is k = 32 w + 8 kq + j;
    // the B fragments are gathered from the packed blob to match (one-time)
    float breg[8][8];
    {
        const float* wb = a.whh + (size_t)s * 4 * H * H;
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const int ctile = (2 * sl + (t8 >> 2)) * 4 + (t8 & 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * wave + 8 * rq + j;
                breg[t8][j] = wb[((size_t)ctile * (H / 4) + (k >> 2)) * 64 + (k & 3) * 16 + col];
            }
        }
    }
    // cell update split by row over the waves: wave w owns rows rq * 4 + w of both unit blocks (unit = 16 ub + lane & 15)
    const int myrow = rq * 4 + wave;
    const int brow = b0 + myrow;
    const bool rowok = brow < a.B;
    const int bclamp = rowok ? brow : a.B - 1;
    float creg[2] = {0.f, 0.f};

    bool aborted = false;
    if (tid == 0) abort_sh = 0;
    if (a.fault && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) return;      // injected failure (tests only)
    for (int t = 0; t < a.T; ++t) {
        float gpre[2][4];
        {
            const float* gp = g + ((size_t)t * a.B + bclamp) * a.ldg + (2 * sl) * 64 + col;
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) gpre[ub][gg] = gp[ub * 64 + 16 * gg];
        }
        f32x4 acc[8];
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t8][r] = 0.f;
        if (t > 0) {
            if (tid == 0) {
                const unsigned want = (unsigned)t * (unsigned)NSL;
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
            const unsigned par_r = (unsigned)((t - 1) & 1) * 4u * (unsigned)a.Bpad * (unsigned)H * 4u;
            const unsigned off = (((unsigned)run * a.Bpad + b0 + col) * (unsigned)H + 32 * wave + 8 * rq) * 4u;
            const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off, par_r, 16));
            const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off + 16u, par_r, 16));
            float av[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[j] = a0[j];
                av[4 + j] = a1[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int t8 = 0; t8 < 8; ++t8)
                    acc[t8] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], breg[t8][j], acc[t8], 0, 0, 0);
        }
        // ---- reduce the 4 K-partials through LDS: [wave][tile][r][lane], conflict-free dword writes and reads
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * 8 + t8) * 4 + r) << 6) + lane] = acc[t8][r];
        __syncthreads();
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            float gate[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                float v = gpre[ub][gg];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[(((w * 8 + ub * 4 + gg) * 4 + wave) << 6) + lane];
                gate[gg] = v;
            }
            const float ig = sigmoidf_(gate[0]), fg = sigmoidf_(gate[1]);
            const float gv = tanhf_(gate[2]), og = sigmoidf_(gate[3]);
            const float cn = fg * creg[ub] + ig * gv;
            creg[ub] = cn;
            const float hv = og * tanhf_(cn);
            stage[myrow][ub * 16 + col] = hv;
            if (gsv && rowok) {
                float* gp = gsv + ((size_t)t * a.B + brow) * a.ldg + (2 * sl + ub) * 64 + col;
                gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                a.csave[(size_t)run * TBH + ((size_t)t * a.B + brow) * H + sl * UPW + ub * 16 + col] = cn;
            }
        }
        __syncthreads();
        if (tid < 128) {
            // 16 rows x 8 float4: write-through (sc1) to the exchange buffer, plain to hout
            const int row = tid >> 3, c4 = tid & 7;
            const v4i pk = *(const v4i*)&stage[row][c4 * 4];
            const unsigned par_w = (unsigned)(t & 1) * 4u * (unsigned)a.Bpad * (unsigned)H * 4u;
            const unsigned off = (((unsigned)run * a.Bpad + b0 + row) * (unsigned)H + sl * UPW + c4 * 4) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(pk, hxr, off, par_w, 16);       // aux 16 = sc1
            if (b0 + row < a.B)
                *(v4i*)&a.hout[(size_t)run * TBH + ((size_t)t * a.B + b0 + row) * H + sl * UPW + c4 * 4] = pk;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < a.nrep) __hip_atomic_fetch_add(counter0 + (size_t)tid * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (aborted) {
        if (tid == 0) idv_coop_raise(a.status);
        const float qnan = __builtin_nanf("");
        for (long long e = tid; e < (long long)a.T * 16 * UPW; e += 256) {
            const int u = (int)(e % UPW), br = (int)((e / UPW) & 15);
            const long long t = e / (UPW * 16);
            if (b0 + br < a.B) a.hout[(size_t)run * TBH + ((size_t)t * a.B + b0 + br) * H + sl * UPW + u] = qnan;
        }
    }
}

constexpr int SYNC_BYTES = 256 + 64 * 8 * 256;      // abort flag + (<= 64 groups) x 8 replicas x 256 B

}  // namespace idv_coop


extern "C" int idv_lstm_coop_f32_supported(int H, int B) {
    static const bool on = [] { const char* e = getenv("IDV_LSTM_COOP_F32"); return !e || e[0] != '0'; }();
    if (!on || H != 128 || B <= 0) return 0;
    const int tiles = (B + 15) / 16;
    return 4 * 4 * tiles <= idv_coop_max_workgroups();      // every workgroup resident at once, one per CU
}

extern "C" long long idv_lstm_coop_f32_work_bytes(int H, int B) {
    const long long Bpad = (B + 15) / 16 * 16;
    return idv_coop::SYNC_BYTES + 2LL * 4 * Bpad * H * 4;
}

// one layer of the H = 128 recurrence in exact fp32 on 4 CUs per (run, 16-sequence tile); arguments as idv_lstm_rec_pers
// (hout required; gsave == g / csave for the training forward or both NULL); work: idv_lstm_coop_f32_work_bytes bytes
extern "C" int idv_lstm_rec_coop_f32(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh_frag, float* hout,
                                     int H, int B, int T, void* work, float* gsave, float* csave, void* stream) {
    using namespace idv_coop;
    if (!g || !whh_frag || !hout || !work || T <= 0 || !idv_lstm_coop_f32_supported(H, B)) return IDV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(work) & 15) || (gsave != nullptr) != (csave != nullptr) || (gsave && gsave != g)) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (B + 15) / 16;
    const long long Bpad = 16LL * tiles;
    if (hipMemsetAsync(work, 0, SYNC_BYTES, st) != hipSuccess) return IDV_ELAUNCH;
    CoopArgs a{};
    a.g = g; a.g_run_z = g_run_z; a.g_run_s = g_run_s; a.ldg = ldg;
    a.whh = whh_frag; a.hout = hout; a.gsave = gsave; a.csave = csave;
    a.sync = (unsigned*)work;
    a.hx = (float*)((char*)work + SYNC_BYTES);
    a.hx_bytes = (unsigned)(2LL * 4 * Bpad * H * 4);
    a.nrep = 4;
    a.B = B; a.T = T; a.Bpad = (int)Bpad; a.tiles = tiles;
    { const char* e = getenv("IDV_COOP_FAULT"); a.fault = (e && e[0] == '1') ? 1 : 0; }
    a.status = idv_coop_status_word();
    const size_t smem = 84 * 1024;                   // > half a CU's LDS: one workgroup per CU (red[] needs 32 KB)
    if (hipFuncSetAttribute((const void*)lstm_coop_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    int rc = idv_coop_chain_begin(st);
    if (rc) return rc;
    hipLaunchKernelGGL(lstm_coop_f32_kernel, dim3(NSL, 4, tiles), dim3(256), smem, st, a);
    if ((rc = idv_coop_chain_end(st))) return rc;
    return idv_launch_status();
}
