// Weight-stationary bidirectional LSTM recurrence over a cluster of CUs (forward and backward through time)
// for the Improved Dual-Path modules (reference: src/model/DTTNet/modules/latent.py:87-92,
// nn.LSTM(C', 2C', bidirectional); gate order i, f, g, o).
//
// Why: at the latent widths of the audio-visual models (H = 512 / 1024) W_hh is 2 / 8 MB per direction.
// A workgroup that owns 32 sequences for all L steps (lstm.hip) has to re-stream all of it from L2 every step:
// the recurrence ran at 0.08-0.13 of the MFMA peak, bound by the 64 B/clk vector-memory path of a CU.  Here the
// 4H gate columns of a direction are split over G workgroups (one per CU): workgroup c keeps the rows of W_hh for
// its U = H / G hidden units (all four gates: 4U rows, <= 133 KB) RESIDENT in LDS for the whole launch, and the
// chip is cut into 2 directions x S sequence groups x G column groups.  Per step a workgroup computes
//     G[n, its 4U columns] = xg[n, ...] + h_{t-1}[n, :] . W_hh[its rows, :]^T      for every sequence n of its group
// with 32x32 MFMA tiles (a wave owns a 32-sequence row tile: the four gates of a hidden unit meet in one lane),
// applies the gates, and publishes its U columns of h_t.  What crosses CUs per step is h only (N x H elements per
// direction), through the y output buffer itself:
//   producer: write-through (sc1) 16-byte stores of the h_t slice -> every wave drains vmcnt -> workgroup barrier
//             -> ONE lane adds 1 to the group's counter (agent-scope atomic);
//   consumer: ONE lane polls the counter (sc1 load) until it reaches G * step -> workgroup barrier -> every load of
//             the handed-off bytes is an sc1 buffer load straight into MFMA A fragments.
// (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table; hipMalloc memory, one workgroup per CU.)
// All workgroups of a launch must be co-resident: the host sizes the grid to the CU count (1 workgroup per CU by
// LDS) and every spin is bounded -- on a timeout the kernel raises a status word and runs to completion.
//
// Activations saved for the backward pass use the blocked layout of lstm.hip's fused kernels (16-byte vectors in
// MFMA accumulator order):  gates [2][NT][L][4][H][2][16] (T),  cst [2][NT][L][H][2][16] (fp32),  NT = ceil(N/32).
#include <algorithm>
#include "common.h"
unsigned dtt_handoff_spin_limit_value();      // lstm_rs.hip: polls per hand-off wait (dtt_lstm_handoff_spin_limit)
#include "../../include/dtt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ float sigm_c(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_c(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
template <typename T> __device__ __forceinline__ float sigm_t(float x) {
    if constexpr (sizeof(T) == 4) return 1.0f / (1.0f + __expf(-x)); else return sigm_c(x);
}
template <typename T> __device__ __forceinline__ float tanh_t(float x) {
    if constexpr (sizeof(T) == 4) return 1.0f - 2.0f / (1.0f + __expf(2.0f * x)); else return tanh_c(x);
}

constexpr unsigned SYNC_STRIDE = 64;          // uints between counters (256 B: one counter per line pair)
// polls per wait before it raises the status word and stops waiting: LstmCP::spin (dtt_lstm_handoff_spin_limit, lstm_rs.hip)
constexpr unsigned OOB = 0xfffffff0u;         // buffer offset beyond num_records: loads return 0, stores are dropped

struct LstmCP {
    const void* xg;      // [rows][2][4H] (T) input projection + biases
    const void* whh;     // [2][4H][H] (T) recurrent weights, PyTorch layout
    void* y;             // [rows][2H] (T): output and the medium of the per-step h exchange
    void* gates;         // blocked, may be NULL (inference)
    float* cst;          // blocked, required (the cell state round-trips through it)
    unsigned* sync;      // [2*S] counters SYNC_STRIDE apart, then one status word; zeroed by the entry point
    int N, L, heads, H, U, G, S, NT, TPG;
    unsigned spin;
};

// 16-byte fragment pieces through buffer instructions with sc1 (bypass this CU's L1 / write through)
__device__ __forceinline__ u32x4_t ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}
template <typename T> struct AFrag;
template <> struct AFrag<bf16> {
    u32x4_t v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned off) { v = ld_sc1(r, off); }
    __device__ __forceinline__ MM<bf16>::Frag frag() const { return __builtin_bit_cast(s16x8_t, v); }
};
template <> struct AFrag<float> {
    u32x4_t lo, hi;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        lo = ld_sc1(r, off);
        hi = ld_sc1(r, off == OOB ? OOB : off + 16);
    }
    __device__ __forceinline__ MM<float>::Frag frag() const {
        MM<float>::Frag f;
        f.lo = __builtin_bit_cast(float4, lo);
        f.hi = __builtin_bit_cast(float4, hi);
        return f;
    }
};

// Sticky, one per GPU: raised (never cleared by a kernel) by any launch of this file whose poll timed out; read
// through dtt_lstm_cluster_status_fold (the per-launch status word is reset by the next call on its buffer).
static __device__ unsigned g_cluster_timeouts;

// One lane polls, the workgroup barrier releases the others (sc1 table, first row).
// `dead` (kept by the polling lane): a wait of this workgroup has timed out once -- a workgroup of the launch is not
// resident or the launch is broken -- so every later wait is skipped and the kernel drains (results are garbage;
// the status word says so).
__device__ __forceinline__ void wait_counter(unsigned* cnt, unsigned target, unsigned* status, bool& dead, unsigned spin_limit) {
    if (threadIdx.x == 0 && !dead) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) {
                __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g_cluster_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
            }
        }
    }
    __syncthreads();
}
// EVERY storing wave drains its sc1 stores, then the workgroup barrier, then ONE lane signals for all of them.
// The exchanged slice is stored FIRST and the wave's private stores (saved activations, running state) behind
// it: vmcnt retires in issue order, so waiting until only those AFTER younger operations are outstanding drains
// exactly the exchanged bytes and keeps the private stores' write-back off the critical path of the step.
template <int AFTER>
__device__ __forceinline__ void signal_counter(unsigned* cnt) {
    static_assert(AFTER == 0 || AFTER == 4 || AFTER == 12 || AFTER == 20, "counted drain");
    if (AFTER == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (AFTER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (AFTER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if (AFTER == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Role of a workgroup.  Workgroups b and b + 8 land on the same XCD (observed dispatch order; speed only, nothing
// depends on it): when the (direction, sequence group) pairs divide evenly over the 8 XCDs, the G workgroups of a
// group are given ids of one residue class, so that the slice one of them publishes is fetched into that XCD's L2
// once and the other G - 1 readers hit it there.
__device__ __forceinline__ void cluster_role(int b, int G, int S, int& dir, int& sg, int& cg) {
    const int P = 2 * S;
    int q;
    if (P % 8 == 0) {
        const int x = b & 7, i = b >> 3;
        q = x * (P >> 3) + i / G;
        cg = i % G;
    } else {
        q = b / G;
        cg = b % G;
    }
    dir = q / S;
    sg = q - dir * S;
}

// UT = U / 32 unit tiles per workgroup (1 or 2); WLDS: the W_hh slice lives in LDS (else it is streamed from L2
// every step: the fp32 parity mode and widths whose slice does not fit).
template <typename T, int AB, bool WLDS>
__global__ __launch_bounds__(512) void lstm_cluster_fwd_kernel(LstmCP p) {
    constexpr int UT = 1;                                   // unit tiles per workgroup (U = 32)
    constexpr int ES = sizeof(T), EPC = 16 / ES;            // elements per 16-byte chunk
    constexpr int CPR = 32 / EPC;                           // chunks per 32-unit row segment (4 / 8)
    constexpr int CPL = 32 * CPR / 64;                      // chunks of a 32 x 32 tile per lane (2 / 4)
    constexpr int LDS_ROW = 32 + EPC;                       // staging row stride (elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, U = UT * 32, LDW = H + MM<T>::PAD, KS = H / 16, G4 = 4 * H;
    T* wbuf = reinterpret_cast<T*>(smem);                                   // [4U][LDW]           (WLDS)
    T* stg_all = wbuf + (WLDS ? (size_t)4 * U * LDW : 0);                   // [8 waves][32][LDS_ROW]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    T* stg = stg_all + wave * 32 * LDS_ROW;
    int cg, sg, dir;
    cluster_role(blockIdx.x, p.G, p.S, dir, sg, cg);
    const int u0 = cg * U;
    const int tile0 = sg * p.TPG, ntile = min(p.TPG, p.NT - tile0);
    unsigned* cnt = p.sync + (size_t)(dir * p.S + sg) * SYNC_STRIDE;
    unsigned* status = p.sync + (size_t)2 * p.S * SYNC_STRIDE;
    const T* whh = reinterpret_cast<const T*>(p.whh) + (size_t)dir * G4 * H;
    const size_t rows = (size_t)p.N * p.L;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(rows * 2 * H * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xg), 0, (int)std::min<size_t>(rows * 2 * G4 * ES, 0xffffffe0u), 0x00020000);

    if (WLDS) {                                            // resident slice: rows (g, u0 + u) of W_hh, k contiguous
        const int cpr = H / EPC;
        for (int i = tid; i < 4 * U * cpr; i += 512) {
            const int rw = i / cpr, ch = i - rw * cpr, g = rw / U, u = rw - g * U;
            st16<T>(wbuf + (size_t)rw * LDW + ch * EPC, ld16<T>(whh + ((size_t)g * H + u0 + u) * H + ch * EPC));
        }
    }
    __syncthreads();
    if (ntile <= 0) return;                                // (the host never launches empty groups)
    bool dead = false;
    // every wave owns at most one row tile: its cell state stays in registers (cst is still written for backward)
    const bool inreg = ntile <= 8;
    float creg[UT][16];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int r = 0; r < 16; ++r) creg[ut][r] = 0.f;

    for (int s = 0; s < p.L; ++s) {
        const int t = dir ? p.L - 1 - s : s;
        const int tprev = dir ? t + 1 : t - 1;
        // the input projection of this wave's first tile does not depend on the exchange: its loads are issued ahead of
        // the wait (their latency hides behind it)
        u32x4_t xq[UT][4][CPL];
        auto x_offsets = [&](int tl, unsigned (&offX)[CPL]) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR, n = (tile0 + tl) * 32 + row;
                const bool ok = n < p.N;
                const unsigned rb = ok ? (unsigned)((n / p.heads) * p.L * p.heads + (n % p.heads) + t * p.heads) : 0u;
                offX[c] = ok ? ((rb * 2u + dir) * (unsigned)G4 + u0 + part * EPC) * ES : OOB;
            }
        };
        auto x_issue = [&](const unsigned (&offX)[CPL]) {
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
                        xq[ut][g][c] = __builtin_amdgcn_raw_buffer_load_b128(rx, offX[c] == OOB ? OOB : offX[c] + (unsigned)(g * H + ut * 32) * ES, 0, 0);
        };
        if (wave < ntile) {
            unsigned offX[CPL];
            x_offsets(wave, offX);
            x_issue(offX);
        }
        if (s > 0) wait_counter(cnt, (unsigned)(p.G * s), status, dead, p.spin);
        for (int tl = wave; tl < ntile; tl += 8) {
            const int gt = tile0 + tl, n0 = gt * 32;
            // ---- row bookkeeping: A-fragment row (lr), staging-chunk rows (2 / 4 per lane)
            const int nA = n0 + lr;
            const bool okA = nA < p.N;
            const unsigned rbA = okA ? (unsigned)((nA / p.heads) * p.L * p.heads + (nA % p.heads)) : 0u;
            unsigned offC[CPL];                            // byte offsets of this lane's chunks in y at step t
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR, n = n0 + row;
                const bool ok = n < p.N;
                const unsigned rb = ok ? (unsigned)((n / p.heads) * p.L * p.heads + (n % p.heads) + t * p.heads) : 0u;
                offC[c] = ok ? ((rb * 2u + dir) * (unsigned)H + u0 + part * EPC) * ES : OOB;
            }
            if (tl != wave) {
                unsigned offX[CPL];
                x_offsets(tl, offX);
                x_issue(offX);
            }
            f32x16_t acc[UT][4];
            // ---- accumulators <- input projection, through the staging tile (transposed LDS reads)
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR;
                        *reinterpret_cast<u32x4_t*>(stg + row * LDS_ROW + part * EPC) = xq[ut][g][c];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ut][g][r] = ldf(stg + ((r & 3) + 8 * (r >> 2) + 4 * lh) * LDS_ROW + lr);
                }
            // ---- previous cell state: it lives in creg across steps when the wave owns one tile; with several tiles per
            // wave it comes back from the cst buffer this same lane wrote one step ago (issued ahead of the A fragments
            // so that the two latencies overlap)
            const size_t slab = ((size_t)dir * p.NT + gt) * p.L + t;
            if (!inreg) {
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) {
                    const size_t slabp = ((size_t)dir * p.NT + gt) * p.L + (s > 0 ? tprev : t);
                    const float* cs = p.cst + (slabp * H + u0 + ut * 32 + lr) * 32 + lh * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(cs + q * 4);
                        creg[ut][q * 4] = s > 0 ? v.x : 0.f; creg[ut][q * 4 + 1] = s > 0 ? v.y : 0.f;
                        creg[ut][q * 4 + 2] = s > 0 ? v.z : 0.f; creg[ut][q * 4 + 3] = s > 0 ? v.w : 0.f;
                    }
                }
            }
            // ---- recurrent term: A fragments of h_{t-1} straight from the exchange buffer (sc1).  The loop is bound by
            // the latency of these loads, not by their bandwidth: a whole batch of AB k-steps is put in flight at once
            if (s > 0) {
                const unsigned baseA = okA ? (((rbA + (unsigned)(tprev * p.heads)) * 2u + dir) * (unsigned)H + lh * 8) * ES : OOB;
                // a rolling window of AB fragments: a slot is re-issued as soon as its fragment has been consumed.  NO
                // branch inside the main loop (hipcc merges wait counts across branches into vmcnt(0) per MFMA, which
                // serialises the window): the refill past the end selects an out-of-range offset instead
                AFrag<T> aq[AB];
                const int KM = KS;                         // the host picks AB as a divisor of KS
#pragma unroll
                for (int d = 0; d < AB; ++d) aq[d].load(ry, (baseA == OOB || d >= KS) ? OOB : baseA + d * 16 * ES);
                for (int kb = 0; kb < KM; kb += AB) {
#pragma unroll
                    for (int d = 0; d < AB; ++d) {
                        const int k = kb + d;
                        const typename MM<T>::Frag af = aq[d].frag();
                        aq[d].load(ry, (baseA == OOB || k + AB >= KS) ? OOB : baseA + (k + AB) * 16 * ES);
#pragma unroll
                        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                typename MM<T>::Frag bq;
                                if (WLDS) bq = MM<T>::ldfrag(wbuf + (size_t)(g * U + ut * 32 + lr) * LDW + k * 16 + lh * 8);
                                else bq = MM<T>::ldfrag(whh + ((size_t)g * H + u0 + ut * 32 + lr) * H + k * 16 + lh * 8);
                                MM<T>::mma(af, bq, acc[ut][g]);
                            }
                    }
                }
            }
            // ---- gates, state update.  h_t goes out FIRST (staging -> 16-byte write-through stores: what the other CUs
            // of the group wait for), the saved activations behind it.
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                const int j = u0 + ut * 32 + lr;
                Vec16<T> gpk[4][16 / EPC];                 // gates packed as they are computed (register pressure)
#pragma unroll
                for (int q = 0; q < 16 / EPC; ++q) {
                    float gv[4][EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const int r = q * EPC + e;
                        gv[0][e] = sigm_t<T>(acc[ut][0][r]); gv[1][e] = sigm_t<T>(acc[ut][1][r]);
                        gv[2][e] = tanh_t<T>(acc[ut][2][r]); gv[3][e] = sigm_t<T>(acc[ut][3][r]);
                        const float c = gv[1][e] * creg[ut][r] + gv[0][e] * gv[2][e];
                        creg[ut][r] = c;
                        stf(stg + ((r & 3) + 8 * (r >> 2) + 4 * lh) * LDS_ROW + lr, gv[3][e] * tanh_t<T>(c));
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (ES == 2) {
#pragma unroll
                            for (int e = 0; e < EPC; e += 2) gpk[g][q].set2(e, gv[g][e], gv[g][e + 1]);
                        } else {
#pragma unroll
                            for (int e = 0; e < EPC; ++e) gpk[g][q].set(e, gv[g][e]);
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR;
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stg + row * LDS_ROW + part * EPC);
                    st_sc1(ry, offC[c] == OOB ? OOB : offC[c] + (unsigned)(ut * 32) * ES, v);
                }
                asm volatile("" ::: "memory");             // the private stores stay behind the exchanged ones
                float* cd = p.cst + (slab * H + j) * 32 + lh * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(cd + q * 4) = make_float4(creg[ut][q * 4], creg[ut][q * 4 + 1], creg[ut][q * 4 + 2], creg[ut][q * 4 + 3]);
                if (p.gates) {
                    T* gd = reinterpret_cast<T*>(p.gates) + ((slab * 4) * H + j) * 32 + lh * 16;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int q = 0; q < 16 / EPC; ++q) st16<T>(gd + (size_t)g * H * 32 + q * EPC, gpk[g][q]);
                }
            }
        }
        if (s + 1 < p.L) {
            if (p.gates) signal_counter<4 + 4 * (16 / EPC)>(cnt); else signal_counter<4>(cnt);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward through time, same decomposition: workgroup c owns the hidden units [u0, u0 + U) of its direction and
// keeps W_hh[:, u0 : u0 + U] (all 4H rows, transposed: [U][4H], <= 133 KB) resident in LDS.  Per step
//     dh_t[n, its units] = dy_t[n, its units] + dG_{t+1}[n, :] . W_hh[:, its units]        (K = 4H)
// then the gate gradients of its units, published as its 4U columns of the row-major dG buffer, which doubles as
// the exchange medium (the weight-gradient GEMMs read it afterwards anyway).  The running dc lives in a blocked
// fp32 scratch (written and re-read by the same lane).
struct LstmCBP {
    const void* dy;      // [rows][2H] (T)
    const void* gates;   // blocked (T)
    const float* cst;    // blocked
    const void* whhT;    // [2][H][4H] (T): W_hh^T per direction
    void* dG;            // [rows][2][4H] (T): output and exchange medium
    float* dcbuf;        // [2][NT][H][2][16] fp32 scratch
    float* dbias;        // [2][4H] fp32, zeroed by the caller (may be NULL)
    unsigned* sync;
    int N, L, heads, H, U, G, S, NT, TPG;
    unsigned spin;
};

template <typename T, int AB, bool WLDS>
__global__ __launch_bounds__(512) void lstm_cluster_bwd_kernel(LstmCBP p) {
    constexpr int UT = 1;
    constexpr int ES = sizeof(T), EPC = 16 / ES, CPR = 32 / EPC, CPL = 32 * CPR / 64, LDS_ROW = 32 + EPC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, U = UT * 32, G4 = 4 * H, LDW = G4 + MM<T>::PAD, KS = G4 / 16;
    T* wbuf = reinterpret_cast<T*>(smem);                                   // [U][LDW]   (WLDS)
    T* stg_all = wbuf + (WLDS ? (size_t)U * LDW : 0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    T* stg = stg_all + wave * 32 * LDS_ROW;
    int cg, sg, dir;
    cluster_role(blockIdx.x, p.G, p.S, dir, sg, cg);
    const int u0 = cg * U;
    const int tile0 = sg * p.TPG, ntile = min(p.TPG, p.NT - tile0);
    unsigned* cnt = p.sync + (size_t)(dir * p.S + sg) * SYNC_STRIDE;
    unsigned* status = p.sync + (size_t)2 * p.S * SYNC_STRIDE;
    const T* whhT = reinterpret_cast<const T*>(p.whhT) + (size_t)dir * G4 * H;
    const size_t rows = (size_t)p.N * p.L;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(p.dG, 0, (int)std::min<size_t>(rows * 2 * G4 * ES, 0xffffffe0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)(rows * 2 * H * ES), 0x00020000);

    if (WLDS) {
        const int cpr = G4 / EPC;
        for (int i = tid; i < U * cpr; i += 512) {
            const int u = i / cpr, ch = i - u * cpr;
            st16<T>(wbuf + (size_t)u * LDW + ch * EPC, ld16<T>(whhT + (size_t)(u0 + u) * G4 + ch * EPC));
        }
    }
    __syncthreads();
    if (ntile <= 0) return;
    bool dead = false;
    float dbs[UT][4];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int g = 0; g < 4; ++g) dbs[ut][g] = 0.f;

    // every wave owns at most one row tile: c_t and the running dc stay in registers across steps
    const bool inreg = ntile <= 8;
    float cc[UT][16], dc[UT][16];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int r = 0; r < 16; ++r) { cc[ut][r] = 0.f; dc[ut][r] = 0.f; }

    for (int s = p.L - 1; s >= 0; --s) {
        const int t = dir ? p.L - 1 - s : s;
        const int tprev = dir ? t + 1 : t - 1;            // forward-pass predecessor (its c is c_{t-1})
        const int tnext = dir ? t - 1 : t + 1;            // step processed in the previous iteration
        const bool first = s == p.L - 1;
        // what the gate math of a tile needs besides dh does not depend on the exchange: for the wave's first tile it
        // is requested ahead of the wait (saved gates, c_{t-1}, the dy tile)
        Vec16<T> gq[UT][4][16 / EPC];
        float4 cpq[UT][4];
        u32x4_t dyq[UT][CPL];
        auto prefetch = [&](int tl) {
            const int gt = tile0 + tl;
            const size_t slab = ((size_t)dir * p.NT + gt) * p.L + t;
            const size_t slabp = ((size_t)dir * p.NT + gt) * p.L + (s > 0 ? tprev : t);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                const int j = u0 + ut * 32 + lr;
                const T* gs = reinterpret_cast<const T*>(p.gates) + ((slab * 4) * H + j) * 32 + lh * 16;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int q = 0; q < 16 / EPC; ++q) gq[ut][g][q] = ld16<T>(gs + (size_t)g * H * 32 + q * EPC);
                const float* cps = p.cst + (slabp * H + j) * 32 + lh * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) cpq[ut][q] = *reinterpret_cast<const float4*>(cps + q * 4);
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR, n = gt * 32 + row;
                    const bool ok = n < p.N;
                    const unsigned rb = ok ? (unsigned)((n / p.heads) * p.L * p.heads + (n % p.heads) + t * p.heads) : 0u;
                    dyq[ut][c] = __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? ((rb * 2u + dir) * (unsigned)H + u0 + ut * 32 + part * EPC) * ES : OOB, 0, 0);
                }
            }
        };
        auto load_state = [&](int tl) {                    // c_t (and the running dc) of a tile from memory
            const int gt = tile0 + tl;
            const size_t slab = ((size_t)dir * p.NT + gt) * p.L + t;
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                const int j = u0 + ut * 32 + lr;
                const float* cs = p.cst + (slab * H + j) * 32 + lh * 16;
                const float* dcs = p.dcbuf + ((((size_t)dir * p.NT + gt) * H + j) * 32 + lh * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(cs + q * 4);
                    cc[ut][q * 4] = a.x; cc[ut][q * 4 + 1] = a.y; cc[ut][q * 4 + 2] = a.z; cc[ut][q * 4 + 3] = a.w;
                    if (!first && !inreg) {
                        const float4 d = *reinterpret_cast<const float4*>(dcs + q * 4);
                        dc[ut][q * 4] = d.x; dc[ut][q * 4 + 1] = d.y; dc[ut][q * 4 + 2] = d.z; dc[ut][q * 4 + 3] = d.w;
                    }
                }
            }
        };
        if (wave < ntile) {
            prefetch(wave);
            if (first && inreg) load_state(wave);
        }
        if (!first) wait_counter(cnt, (unsigned)(p.G * (p.L - 1 - s)), status, dead, p.spin);
        for (int tl = wave; tl < ntile; tl += 8) {
            const int gt = tile0 + tl, n0 = gt * 32;
            const int nA = n0 + lr;
            const bool okA = nA < p.N;
            const unsigned rbA = okA ? (unsigned)((nA / p.heads) * p.L * p.heads + (nA % p.heads)) : 0u;
            unsigned offG[CPL];                            // this lane's chunks of the dG tile at step t
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR, n = n0 + row;
                const bool ok = n < p.N;
                const unsigned rb = ok ? (unsigned)((n / p.heads) * p.L * p.heads + (n % p.heads) + t * p.heads) : 0u;
                offG[c] = ok ? ((rb * 2u + dir) * (unsigned)G4 + u0 + part * EPC) * ES : OOB;
            }
            if (tl != wave) prefetch(tl);
            if (!inreg) {
                if (first) {
#pragma unroll
                    for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dc[ut][r] = 0.f;
                }
                load_state(tl);
            }
            // ---- dh_t <- dG_{t+1} . W_hh[:, units]: A fragments straight from the exchange buffer
            f32x16_t dh[UT];
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[ut][r] = 0.f;
            if (!first) {
                const unsigned baseA = okA ? (((rbA + (unsigned)(tnext * p.heads)) * 2u + dir) * (unsigned)G4 + lh * 8) * ES : OOB;
                // latency-bound: a rolling window of AB fragments in flight, a slot re-issued once consumed; no branch in
                // the main loop (see the forward kernel)
                AFrag<T> aq[AB];
                const int KM = KS;                         // the host picks AB as a divisor of KS
#pragma unroll
                for (int d = 0; d < AB; ++d) aq[d].load(rg, (baseA == OOB || d >= KS) ? OOB : baseA + d * 16 * ES);
                for (int kb = 0; kb < KM; kb += AB) {
#pragma unroll
                    for (int d = 0; d < AB; ++d) {
                        const int k = kb + d;
                        const typename MM<T>::Frag af = aq[d].frag();
                        aq[d].load(rg, (baseA == OOB || k + AB >= KS) ? OOB : baseA + (k + AB) * 16 * ES);
#pragma unroll
                        for (int ut = 0; ut < UT; ++ut) {
                            typename MM<T>::Frag bq;
                            if (WLDS) bq = MM<T>::ldfrag(wbuf + (size_t)(ut * 32 + lr) * LDW + k * 16 + lh * 8);
                            else bq = MM<T>::ldfrag(whhT + (size_t)(u0 + ut * 32 + lr) * G4 + k * 16 + lh * 8);
                            MM<T>::mma(af, bq, dh[ut]);
                        }
                    }
                }
            }
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                const int j = u0 + ut * 32 + lr;
                // dy tile through the staging buffer
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR;
                    *reinterpret_cast<u32x4_t*>(stg + row * LDS_ROW + part * EPC) = dyq[ut][c];
                }
                float di[16], df[16], dg[16], dox[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const bool ok = n0 + nl < p.N;
                    const float gi = gq[ut][0][r / EPC].get(r % EPC), gf = gq[ut][1][r / EPC].get(r % EPC);
                    const float gg = gq[ut][2][r / EPC].get(r % EPC), go = gq[ut][3][r / EPC].get(r % EPC);
                    const float cpv = s > 0 ? (&cpq[ut][r >> 2].x)[r & 3] : 0.f;
                    const float dhv = ldf(stg + nl * LDS_ROW + lr) + dh[ut][r];
                    const float tc = tanh_t<T>(cc[ut][r]);
                    const float dcv = dc[ut][r] + dhv * go * (1.f - tc * tc);
                    dox[r] = ok ? dhv * tc * go * (1.f - go) : 0.f;
                    di[r] = ok ? dcv * gg * gi * (1.f - gi) : 0.f;
                    df[r] = ok ? dcv * cpv * gf * (1.f - gf) : 0.f;
                    dg[r] = ok ? dcv * gi * (1.f - gg * gg) : 0.f;
                    dc[ut][r] = dcv * gf;
                    cc[ut][r] = cpv;                       // c_{t-1} is the next iteration's c_t
                }
                // gate gradients FIRST (the other CUs wait for them): staging -> row-major 16-byte sc1 stores
                const float* gsrc[4] = {di, df, dg, dox};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        stf(stg + ((r & 3) + 8 * (r >> 2) + 4 * lh) * LDS_ROW + lr, gsrc[g][r]);
                        sum += gsrc[g][r];
                    }
                    dbs[ut][g] += sum;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const int ci = lane + 64 * c, row = ci / CPR, part = ci - row * CPR;
                        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stg + row * LDS_ROW + part * EPC);
                        st_sc1(rg, offG[c] == OOB ? OOB : offG[c] + (unsigned)(g * H + ut * 32) * ES, v);
                    }
                }
                asm volatile("" ::: "memory");             // the private store stays behind the exchanged ones
                if (!inreg && s > 0) {
                    float* dcd = p.dcbuf + ((((size_t)dir * p.NT + gt) * H + j) * 32 + lh * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(dcd + q * 4) = make_float4(dc[ut][q * 4], dc[ut][q * 4 + 1], dc[ut][q * 4 + 2], dc[ut][q * 4 + 3]);
                }
            }
        }
        if (s > 0) {
            if (inreg) signal_counter<0>(cnt); else signal_counter<4>(cnt);
        }
    }
    if (p.dbias) {
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v = dbs[ut][g] + __shfl_xor(dbs[ut][g], 32, 64);
                if (lh == 0) atomicAdd(p.dbias + (size_t)dir * G4 + g * H + u0 + ut * 32 + lr, v);
            }
    }
}

size_t cluster_bwd_lds(int H, int U, int es, bool wlds) {
    const int pad = es == 2 ? 8 : 4, epc = 16 / es;
    return (wlds ? (size_t)U * (4 * H + pad) * es : 0) + (size_t)8 * 32 * (32 + epc) * es;
}

size_t cluster_fwd_lds(int H, int U, int es, bool wlds) {
    const int pad = es == 2 ? 8 : 4, epc = 16 / es;
    return (wlds ? (size_t)4 * U * (H + pad) * es : 0) + (size_t)8 * 32 * (32 + epc) * es;
}

constexpr size_t LDS_CAP = 160 * 1024;
int g_num_cus = 0;
int num_cus() {
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        g_num_cus = pr.multiProcessorCount;
    }
    return g_num_cus;
}

}  // namespace

// Geometry the host needs to size the buffers: U (hidden units per workgroup), G = H / U column groups, S sequence
// groups.  Returns 0 when the cluster kernels cover (H, dtype) on this GPU, 1 otherwise (use dtt_lstm_fwd / _bwd).
extern "C" int dtt_lstm_cluster_geometry(int N, int H, int dtype, int* U, int* G, int* S) {
    const int ncu = num_cus();
    if (H < 32 || H % 32 != 0 || N <= 0 || ncu <= 0) return 1;
    int u = 32;
    (void)dtype;                                            // (a 64-unit variant exists in the templates; it spills and is not used)
    const int g = H / u;
    if (2 * g > ncu) return 1;
    const int nt = (N + 31) / 32;
    int s = std::min(nt, ncu / (2 * g));
    const int tpg = (nt + s - 1) / s;
    s = (nt + tpg - 1) / tpg;                               // no empty groups
    if (U) *U = u;
    if (G) *G = g;
    if (S) *S = s;
    return 0;
}

// The hand-off counters are zeroed by a KERNEL, not by hipMemsetAsync: recorded into a HIP graph, the memset nodes
// were the one thing these entry points had that the streaming kernels had not, and replayed after a host-side
// stream sync the step then returned wrong gradients (DESIGN.md finding 34).
static __global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

static __global__ void cluster_status_fold_kernel(unsigned* __restrict__ flag, unsigned bit, int clear) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (g_cluster_timeouts && flag) atomicOr(flag, bit);
        if (clear) g_cluster_timeouts = 0u;
    }
}

// flag |= 2 when a launch of this file has timed out on this GPU since the last clear (clear != 0 re-arms it)
extern "C" int dtt_lstm_cluster_status_fold(unsigned* flag, int clear, void* stream) {
    hipLaunchKernelGGL(cluster_status_fold_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), flag, 2u, clear);
    DTT_LAUNCH_CHECK();
    return 0;
}

// Diagnostic, as dtt_lstm_rs_diag_drop_workgroups: the next cluster launch starts n workgroups short.
static int g_cluster_diag_drop = 0;
extern "C" void dtt_lstm_cluster_diag_drop_workgroups(int n) { g_cluster_diag_drop = n > 0 ? n : 0; }
static unsigned cluster_grid(int full) {
    const int drop = std::min(g_cluster_diag_drop, full - 1);
    g_cluster_diag_drop = 0;
    return (unsigned)(full - drop);
}

extern "C" int dtt_lstm_cluster_sync_words(int S) { return (2 * S + 1) * (int)SYNC_STRIDE; }

extern "C" int dtt_lstm_cluster_fwd(const void* xg, const void* whh, void* y, void* gates_blk, float* cst_blk,
                                    unsigned* sync, int N, int L, int heads, int H, int dtype, void* stream) {
    int U, G, S;
    DTT_CHECK(dtt_lstm_cluster_geometry(N, H, dtype, &U, &G, &S) == 0, "dtt_lstm_cluster_fwd: unsupported shape H=%d N=%d", H, N);
    DTT_CHECK(L > 0 && heads > 0 && N % heads == 0 && xg && whh && y && cst_blk && sync, "dtt_lstm_cluster_fwd: bad arguments");
    const int es = dtype ? 2 : 4;
    DTT_CHECK((int64_t)N * L * 2 * 4 * H * es < (1LL << 32) - 64, "dtt_lstm_cluster_fwd: xg exceeds 32-bit byte offsets");
    const int NT = (N + 31) / 32, TPG = (NT + S - 1) / S;
    LstmCP p{xg, whh, y, gates_blk, cst_blk, sync, N, L, heads, H, U, G, S, NT, TPG, dtt_handoff_spin_limit_value()};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool wlds = dtype == 1 && cluster_fwd_lds(H, U, es, true) <= LDS_CAP;
    const size_t sm = cluster_fwd_lds(H, U, es, wlds);
    { const int nw = dtt_lstm_cluster_sync_words(S); hipLaunchKernelGGL(zero_words_kernel, dim3((nw + 255) / 256), dim3(256), 0, st, sync, nw); }
    const dim3 grid(cluster_grid(2 * S * G)), block(512);
#define LAUNCH_CF(TT, UTV, WL) do { auto k = lstm_cluster_fwd_kernel<TT, UTV, WL>; \
        dtt_want_lds((const void*)k, (int)sm); \
        hipLaunchKernelGGL(k, grid, block, sm, st, p); } while (0)
    const int KS = H / 16;                                  // k-steps: the window depth AB must divide it
    if (dtype) {
        if (wlds) { if (KS % 16 == 0) LAUNCH_CF(bf16, 16, true); else if (KS % 8 == 0) LAUNCH_CF(bf16, 8, true); else if (KS % 4 == 0) LAUNCH_CF(bf16, 4, true); else LAUNCH_CF(bf16, 2, true); }
        else { if (KS % 16 == 0) LAUNCH_CF(bf16, 16, false); else if (KS % 4 == 0) LAUNCH_CF(bf16, 4, false); else LAUNCH_CF(bf16, 2, false); }
    } else {
        if (KS % 4 == 0) LAUNCH_CF(float, 4, false); else LAUNCH_CF(float, 2, false);
    }
#undef LAUNCH_CF
    DTT_LAUNCH_CHECK();
    return 0;
}

// dy [rows][2H], gates / cst blocked as written by dtt_lstm_cluster_fwd, whhT [2][H][4H] = W_hh^T per direction,
// dG [rows][2][4H] row-major (output), dcbuf [2][NT][H][32] fp32 scratch, dbias [2][4H] fp32 zeroed by the caller
// (may be NULL), sync as for the forward call.
extern "C" int dtt_lstm_cluster_bwd(const void* dy, const void* gates_blk, const float* cst_blk, const void* whhT, void* dG,
                                    float* dcbuf, float* dbias, unsigned* sync, int N, int L, int heads, int H, int dtype,
                                    void* stream) {
    int U, G, S;
    DTT_CHECK(dtt_lstm_cluster_geometry(N, H, dtype, &U, &G, &S) == 0, "dtt_lstm_cluster_bwd: unsupported shape H=%d N=%d", H, N);
    DTT_CHECK(L > 0 && heads > 0 && N % heads == 0 && dy && gates_blk && cst_blk && whhT && dG && dcbuf && sync,
              "dtt_lstm_cluster_bwd: bad arguments");
    const int es = dtype ? 2 : 4;
    DTT_CHECK((int64_t)N * L * 2 * 4 * H * es < (1LL << 32) - 64, "dtt_lstm_cluster_bwd: dG exceeds 32-bit byte offsets");
    const int NT = (N + 31) / 32, TPG = (NT + S - 1) / S;
    LstmCBP p{dy, gates_blk, cst_blk, whhT, dG, dcbuf, dbias, sync, N, L, heads, H, U, G, S, NT, TPG, dtt_handoff_spin_limit_value()};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool wlds = dtype == 1 && cluster_bwd_lds(H, U, es, true) <= LDS_CAP;
    const size_t sm = cluster_bwd_lds(H, U, es, wlds);
    { const int nw = dtt_lstm_cluster_sync_words(S); hipLaunchKernelGGL(zero_words_kernel, dim3((nw + 255) / 256), dim3(256), 0, st, sync, nw); }
    const dim3 grid(cluster_grid(2 * S * G)), block(512);
#define LAUNCH_CB(TT, UTV, WL) do { auto k = lstm_cluster_bwd_kernel<TT, UTV, WL>; \
        dtt_want_lds((const void*)k, (int)sm); \
        hipLaunchKernelGGL(k, grid, block, sm, st, p); } while (0)
    const int KS = 4 * H / 16;
    if (dtype) {
        if (wlds) { if (KS % 16 == 0) LAUNCH_CB(bf16, 16, true); else LAUNCH_CB(bf16, 8, true); }
        else { if (KS % 16 == 0) LAUNCH_CB(bf16, 16, false); else LAUNCH_CB(bf16, 8, false); }
    } else {
        LAUNCH_CB(float, 4, false);
    }
#undef LAUNCH_CB
    DTT_LAUNCH_CHECK();
    return 0;
}
