// Dedicated 3x3 stride-1 "same" convolution kernels on channels-last tensors
// (the 49 TFC convs + 49 data-gradient convs + 49 weight-gradient reductions of a
// training step: block_tfc_tdf.py:12-32).  Input patches (16x16 pixels + halo) are
// staged once in LDS and re-used by all nine taps; the forward kernel also emits the
// per-tile InstanceNorm statistics from its fp32 accumulators.
//
//   conv3x3_s1_kernel : y[b,f,t,co] = bias[co] + sum_{tap,ci} x[b,f+kh-1,t+kw-1,ci] w[co][tap][ci] (+ res)
//   conv3x3_wgrad_kernel (bf16): dw[co][tap][ci] += sum_px dy[px][co] * x[px+tap][ci]
//       pixels are the MFMA K dimension; operands are read from the natural
//       [pixel][channel] LDS image with the transposing ds_read_b64_tr_b16.
#include <stdlib.h>
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

constexpr int TH = 16, TW = 16, PH = TH + 2, PW = TW + 2, KC = 32;
// Patch rows start on multiples of 256 bytes (= all 64 LDS banks).  An A-fragment read (ds_read_b128) takes, per 16-lane
// group, tile columns 0-3 and 12-15 of one patch row and 4-11 of the next: with 80-byte (144-byte fp32) pixels the sixteen
// columns of one row fall on sixteen different 4-bank slots, and with the next row a whole number of bank rounds further
// its columns keep theirs -- no conflict.  At the natural pitch of 18 pixels two of the eight collided (35 % of the
// LDS-array cycles of conv3x3_s1_kernel were bank-conflict cycles, profiles/r02z_pmc_sq_lds_mfma.txt).
template <typename T, int LDC> struct PatchPitch { static constexpr int ROWP = (PW * LDC * (int)sizeof(T) + 255) / 256 * 256 / (int)sizeof(T); };

struct Conv3P {
    const void* x; const void* wp; const float* bias; void* y; const void* res; float* partial;
    int B, F, T, Ci, Co, tiles_f, tiles_t;
};

// Shared epilogue.  A wave owns 64 pixels (tile rows 4*wave .. +3) x BN channels in acc[a][j]
// (lane: channel j*32 + (lane&31), pixels i = (r&3) + 8*(r>>2) + 4*(lane>>5) of half a).  The values
// (+bias) are parked in a wave-private fp32 LDS slab and leave as 16-byte chunks -- a tile row of 16
// pixels is one contiguous run of 16*Co elements in the channels-last tensor -- with the residual
// read the same way.  Also the per-tile InstanceNorm statistics {mean, M2}; all four waves must call.
// Per-lane running InstanceNorm sums of a persistent workgroup (conv3x3_c32_kernel): accumulated over
// all tiles of one batch item about a fixed pivot, flushed once per (workgroup, wave, batch item).
struct RunStat { float piv, s1, s2, n; bool open; };

// bias of the lane's output channels (co0 + 32 j + lane % 32), 0 without a bias; loaded by the kernels ahead
// of their main loop so that no wait for it sits behind the prefetch loads
template <int TN>
__device__ __forceinline__ void conv3_bias(const Conv3P& p, int co0, float (&bsv)[TN]) {
    const float* bp = p.bias ? p.bias : reinterpret_cast<const float*>(&dtt_zero16);
#pragma unroll
    for (int j = 0; j < TN; ++j) bsv[j] = bp[p.bias ? co0 + j * 32 + (threadIdx.x & 31) : 0];
}

// RES (fused residual) is a compile-time flag: with a run-time `if (res)` around the loads and another
// around their use, the compiler sees a path on which the loads stay pending and puts vmcnt(0) in front
// of the next tile's MFMAs in the persistent kernel -- which serialised its register prefetch.
template <typename T, int TN, bool RES>
__device__ __forceinline__ void conv3_epilogue(const Conv3P& p, f32x16_t (&acc)[2][TN], float* stage, const float (&bsv)[TN],
                                               int b, int tf, int tt, int co0, RunStat* run = nullptr) {
    constexpr int BN = 32 * TN, SW = BN + 4, E = Vec16<T>::N, CPP = BN / E, NIT = CPP;   // 64 px * CPP chunks / 64 lanes
    // (wave: the wave's 4-row slab of the tile; kernels with eight waves give waves 4-7 the next 32 output channels)
    const int tid = threadIdx.x, wave = (tid >> 6) & 3, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int f0 = tf * TH, t0 = tt * TW;
    T* y = reinterpret_cast<T*>(p.y) + (int64_t)b * p.F * p.T * p.Co;
    const T* res = RES ? reinterpret_cast<const T*>(p.res) + (int64_t)b * p.F * p.T * p.Co : nullptr;
    // statistics are accumulated about a pivot (the slab's first pixel, always inside the image when
    // the slab has any valid pixel) so that sum-of-squares minus square-of-sum does not cancel
    float s1[TN], s2[TN], piv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
        piv[j] = __shfl(acc[0][j][0] + bsv[j], lr, 64);
    }
    if (run) {                                             // TN == 1: keep the pivot of the run
        if (!run->open) { run->piv = piv[0]; run->s1 = 0.f; run->s2 = 0.f; run->n = 0.f; run->open = true; }
        piv[0] = run->piv;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int f = f0 + wave * 4 + a * 2 + (i >> 4), t = t0 + (i & 15);
            const bool ok = f < p.F && t < p.T;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float v = acc[a][j][r] + bsv[j];
                const float dv = ok ? v - piv[j] : 0.f;
                s1[j] += dv;
                s2[j] += dv * dv;
                stage[(a * 32 + i) * SW + j * 32 + lr] = v;
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    Vec16<T> rv[NIT];
    if constexpr (RES) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = lane + it * 64, px = id / CPP, ch = id % CPP;
            const int f = f0 + wave * 4 + (px >> 4), t = t0 + (px & 15);
            rv[it] = ld16_pred<T>(res + ((int64_t)f * p.T + t) * p.Co + co0 + ch * E, res, f < p.F && t < p.T);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int id = lane + it * 64, px = id / CPP, ch = id % CPP;
        const int f = f0 + wave * 4 + (px >> 4), t = t0 + (px & 15);
        float v[E];
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const float4 s = *reinterpret_cast<const float4*>(stage + px * SW + ch * E + q * 4);
            v[q * 4] = s.x; v[q * 4 + 1] = s.y; v[q * 4 + 2] = s.z; v[q * 4 + 3] = s.w;
        }
        if constexpr (RES) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += rv[it].get(e);
        }
        Vec16<T> ov;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < E; e += 2) ov.set2(e, v[e], v[e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) ov.set(e, v[e]);
        }
        if (f < p.F && t < p.T) st16<T>(y + ((int64_t)f * p.T + t) * p.Co + co0 + ch * E, ov);
    }
    if (!p.partial) return;
    // ---- InstanceNorm statistics {mean, M2} of this wave's slab (4 tile rows), no workgroup barrier:
    // fp32 sum and sum of squares over <= 64 pixels; the slabs are merged (Chan) by conv_stats_finalize.
    const int nr = min(4, max(0, p.F - (f0 + wave * 4))), ntv = min(TW, p.T - t0);
    const float n = (float)(nr * ntv), inv_n = n > 0.f ? 1.f / n : 0.f;
    if (run) { run->s1 += s1[0]; run->s2 += s2[0]; run->n += n; return; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float a1 = s1[j] + __shfl_xor(s1[j], 32, 64);
        const float a2 = s2[j] + __shfl_xor(s2[j], 32, 64);
        if (lh == 0) {
            const float dm = a1 * inv_n;
            float2 o = make_float2(piv[j] + dm, fmaxf(a2 - a1 * dm, 0.f));
            *reinterpret_cast<float2*>(p.partial + (((((int64_t)b * p.tiles_f + tf) * 4 + wave) * p.tiles_t + tt) * p.Co +
                                                    co0 + j * 32 + lr) * 2) = o;
        }
    }
}

// waves: 4 along M (each 4 tile rows = 64 px, TM = 2), all of BN = 32*TN output channels.
template <typename T, int TN>
__global__ __launch_bounds__(256) void conv3x3_s1_kernel(Conv3P p) {
    constexpr int EPT = MM<T>::EPT, LDC = KC + MM<T>::PAD, BN = 32 * TN, CPR = KC / EPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWP = PatchPitch<T, LDC>::ROWP;          // elements between patch rows (256-byte multiples)
    T* patch = reinterpret_cast<T*>(smem);                  // [PH][ROWP >= PW*LDC]
    T* wl = patch + PH * ROWP;                              // [9*BN][LDC]

    const int ntile_n = p.Co / BN;
    int bid = blockIdx.x;
    const int nt = bid % ntile_n; bid /= ntile_n;
    const int tt = bid % p.tiles_t, tf = bid / p.tiles_t;
    const int b = blockIdx.y;
    const int f0 = tf * TH, t0 = tt * TW, co0 = nt * BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const T* x = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.F * p.T * p.Ci;
    const T* wp = reinterpret_cast<const T*>(p.wp);

    float bsv[TN];
    conv3_bias<TN>(p, co0, bsv);
    f32x16_t acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;

    // 32-channel chunks of Ci; the next chunk's patch and weights are fetched into registers while
    // the current one is in the MFMAs (16-byte pieces: NP of the patch, NW of the weights per thread)
    constexpr int NPC = PH * PW * CPR, NP = (NPC + 255) / 256, NWC = 9 * BN * CPR, NW = (NWC + 255) / 256;
    Vec16<T> pre_p[NP], pre_w[NW];
    auto fetch = [&](int c) {
        const int ci0 = c * KC;
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int idx = tid + it * 256, pp = idx / CPR, ch = idx % CPR;
            const int pf = pp / PW, pt = pp - pf * PW;
            const int f = f0 - 1 + pf, t = t0 - 1 + pt, ci = ci0 + ch * EPT;
            pre_p[it] = ld16_pred<T>(x + ((int64_t)f * p.T + t) * p.Ci + ci, x,
                                     idx < NPC && f >= 0 && f < p.F && t >= 0 && t < p.T && ci < p.Ci);
        }
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const int idx = tid + it * 256, row = idx / CPR, ch = idx % CPR;
            const int tap = row / BN, col = row - tap * BN, ci = ci0 + ch * EPT;
            pre_w[it] = ld16_pred<T>(wp + ((int64_t)(co0 + col) * 9 + tap) * p.Ci + ci, wp, idx < NWC && ci < p.Ci);
        }
    };
    const int nchunks = (p.Ci + KC - 1) / KC;
    fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int idx = tid + it * 256;
            if (idx < NPC) { const int pix = idx / CPR; st16<T>(patch + (pix / PW) * ROWP + (pix % PW) * LDC + (idx % CPR) * EPT, pre_p[it]); }
        }
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const int idx = tid + it * 256;
            if (idx < NWC) st16<T>(wl + (idx / CPR) * LDC + (idx % CPR) * EPT, pre_w[it]);
        }
        __syncthreads();
        if (c + 1 < nchunks) fetch(c + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kofs = s * 16 + lh * 8;
                typename MM<T>::Frag af[2], bf[TN];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    int prow = wave * 4 + a * 2 + (lr >> 4) + kh, pcol = (lr & 15) + kw;
                    af[a] = MM<T>::ldfrag(patch + prow * ROWP + pcol * LDC + kofs);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = MM<T>::ldfrag(wl + (tap * BN + j * 32 + lr) * LDC + kofs);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < TN; ++j) MM<T>::mma(af[a], bf[j], acc[a][j]);
            }
        }
    }

    __syncthreads();                                        // operand LDS becomes the output staging area
    float* stage = reinterpret_cast<float*>(smem) + wave * 64 * (BN + 4);
    if (p.res) conv3_epilogue<T, TN, true>(p, acc, stage, bsv, b, tf, tt, co0);
    else conv3_epilogue<T, TN, false>(p, acc, stage, bsv, b, tf, tt, co0);
}

// Specialisation for Ci = Co = 32, bf16 (the full-resolution TFC convs: HBM-bound at 144 flop/byte):
// persistent workgroups; the 18 KB of weights live in registers for the whole launch; the next tile's
// patch is prefetched into registers while the current one is in the MFMAs.
template <bool RES>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_kernel(Conv3P p, int tiles_per_wg) {
    constexpr int LDC = 40, NPC = PH * PW * 4, NIT = (NPC + 255) / 256, SW = 36, ROWP = PatchPitch<bf16, LDC>::ROWP;
    __shared__ __attribute__((aligned(256))) bf16 patch[PH * ROWP];
    __shared__ __attribute__((aligned(16))) float stage_all[4 * 64 * SW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const bf16* wp = reinterpret_cast<const bf16*>(p.wp);
    s16x8_t wf[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int s = 0; s < 2; ++s) wf[tap][s] = MM<bf16>::ldfrag(wp + (lr * 9 + tap) * 32 + s * 16 + lh * 8);

    const int per_b = p.tiles_f * p.tiles_t, total = p.B * per_b;
    const int tile0 = blockIdx.x * tiles_per_wg, tile1 = min(total, tile0 + tiles_per_wg);
    if (tile0 >= tile1) return;
    Vec16<bf16> pre[NIT];
    auto prefetch = [&](int tile) {
        const int b = tile / per_b, rr = tile - b * per_b, tf = rr / p.tiles_t, tt = rr - tf * p.tiles_t;
        const bf16* x = reinterpret_cast<const bf16*>(p.x) + (int64_t)b * p.F * p.T * 32;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, pp = idx >> 2, ch = idx & 3;
            const int pf = pp / PW, pt = pp - pf * PW;
            const int f = tf * TH - 1 + pf, t = tt * TW - 1 + pt;
            pre[it] = ld16_pred<bf16>(x + ((int64_t)f * p.T + t) * 32 + ch * 8, x,
                                      idx < NPC && f >= 0 && f < p.F && t >= 0 && t < p.T);
        }
    };
    float bsv[1];
    conv3_bias<1>(p, 0, bsv);
    prefetch(tile0);
    // InstanceNorm partials: one {mean, M2, n} per (batch item, workgroup slot, wave, channel), the slot
    // being the workgroup's index among those that touch the batch item (see conv_c32_stats_finalize)
    RunStat run{0.f, 0.f, 0.f, 0.f, false};
    int run_b = -1;
    const int nslot = (per_b + tiles_per_wg - 1) / tiles_per_wg + 1;
    auto flush = [&]() {
        if (!run.open || !p.partial) return;
        const float a1 = run.s1 + __shfl_xor(run.s1, 32, 64), a2 = run.s2 + __shfl_xor(run.s2, 32, 64);
        if (lh == 0) {
            const float dm = run.n > 0.f ? a1 / run.n : 0.f;
            const int slot = blockIdx.x - (run_b * per_b) / tiles_per_wg;
            float* d = p.partial + ((((int64_t)run_b * nslot + slot) * 4 + wave) * 32 + lr) * 3;
            d[0] = run.piv + dm; d[1] = fmaxf(a2 - a1 * dm, 0.f); d[2] = run.n;
        }
        run.open = false;
    };
    auto park = [&]() {                                     // prefetched patch: registers -> LDS
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            if (idx < NPC) { const int pix = idx >> 2; st16<bf16>(patch + (pix / PW) * ROWP + (pix % PW) * LDC + (idx & 3) * 8, pre[it]); }
        }
    };
    park();
    __syncthreads();
    if (tile0 + 1 < tile1) prefetch(tile0 + 1);
    // Per tile: MFMAs on patch(i) | barrier | patch(i+1) registers -> LDS, prefetch(i+2) issued | epilogue(i)
    // | barrier.  The wait for the loads of patch(i+1) thus comes a whole MFMA phase after the output
    // stores of tile i-1 were issued (vmcnt retires in order: a wait placed right after an epilogue
    // would also wait for that epilogue's stores to be acknowledged).
    for (int tile = tile0; tile < tile1; ++tile) {
        const int b = tile / per_b, rr = tile - b * per_b, tf = rr / p.tiles_t, tt = rr - tf * p.tiles_t;
        if (b != run_b) { flush(); run_b = b; }
        f32x16_t acc[2][1];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int prow = wave * 4 + a * 2 + (lr >> 4) + kh, pcol = (lr & 15) + kw;
                    const s16x8_t af = MM<bf16>::ldfrag(patch + prow * ROWP + pcol * LDC + s * 16 + lh * 8);
                    MM<bf16>::mma(af, wf[tap][s], acc[a][0]);
                }
            }
        }
        __syncthreads();                                    // every wave is done reading patch(i)
        if (tile + 1 < tile1) park();
        if (tile + 2 < tile1) prefetch(tile + 2);
        conv3_epilogue<bf16, 1, RES>(p, acc, stage_all + wave * 64 * SW, bsv, b, tf, tt, 0, p.partial ? &run : nullptr);
        __syncthreads();                                    // patch(i+1) is visible
    }
    flush();
}

// Ci = Co = 64, bf16 (the second U-Net level: 257 x 63 planes).  conv3x3_s1_kernel spends its time at B = 16 on the
// per-chunk chain load -> LDS -> barrier -> MFMA of workgroups that live for one tile (~12 us per round of resident
// workgroups against 1.9 us of MFMA, DESIGN finding 35).  Here, as in the 32-channel kernel, ONE persistent workgroup
// per CU walks its tiles: eight waves = 4 row slabs x 2 halves of the output channels; all 9 x 64 x 64 weights stay in
// LDS for the whole launch (83 KB; as register fragments they would be 144 registers per wave: spills at two waves
// per SIMD); the whole 64-channel patch of a tile is one LDS image (no channel chunks, no barrier inside a tile); the
// next tile's patch travels through registers while the current one is in the MFMAs.  The fp32 staging slabs of the
// epilogue (74 KB) overlay the patch (51 KB): weights + max(patch, slabs) = 157 KB of the CU's 160.
// Per tile: MFMAs(i) | barrier | epilogue(i) | barrier | patch(i+1) registers -> LDS, prefetch(i+2) issued | barrier.
constexpr int C64_LDC = 72, C64_NPC = PH * PW * 8, C64_NIT = (C64_NPC + 511) / 512, C64_SW = 36;
constexpr int C64_ROWP = PatchPitch<bf16, C64_LDC>::ROWP;
constexpr size_t C64_WBYTES = (size_t)9 * 64 * C64_LDC * sizeof(bf16);
constexpr size_t C64_PBYTES = (size_t)PH * C64_ROWP * sizeof(bf16), C64_SBYTES = (size_t)8 * 64 * C64_SW * sizeof(float);
constexpr size_t C64_SMEM = C64_WBYTES + (C64_PBYTES > C64_SBYTES ? C64_PBYTES : C64_SBYTES);
// The MFMA phase of a tile as 36 (tap, k-step) items of two MFMAs, with the three fragment reads of item i + 2 issued
// ahead of the MFMAs of item i.  In asm, with counted waits: left to the compiler every ds_read sinks to its MFMA behind
// s_waitcnt lgkmcnt(0..1) (also from source-level double buffering with scheduling barriers), and the phase takes 3.5 us
// per tile instead of the 1.9 us of its MFMAs.
typedef __attribute__((address_space(3))) void* c64_lptr_t;
struct C64Frag { s16x8_t b, a0, a1; };
constexpr int C64_WREG = 2;            // k-steps (of 4) whose weight fragments a wave keeps in registers: 18 x 4 registers
template <int IT> constexpr int c64_nreads() { return (IT & 3) < C64_WREG ? 2 : 3; }
template <int IT>
__device__ __forceinline__ void c64_read(C64Frag& f, unsigned wb0, unsigned wb1, unsigned a0, unsigned a1) {
    constexpr int tap = IT >> 2, s = IT & 3, kh = tap / 3, kw = tap % 3;
    constexpr int ob = ((tap % 5) * 64 * C64_LDC + s * 16) * 2;          // taps 5-8 from a second base: 16-bit offsets
    constexpr int oa = (kh * C64_ROWP + kw * C64_LDC + s * 16) * 2;
    if constexpr (s < C64_WREG)
        asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"
                     : "=&v"(f.a0), "=&v"(f.a1) : "v"(a0), "v"(a1), "n"(oa));
    else
        asm volatile("ds_read_b128 %0, %3 offset:%6\n\tds_read_b128 %1, %4 offset:%7\n\tds_read_b128 %2, %5 offset:%7"
                     : "=&v"(f.b), "=&v"(f.a0), "=&v"(f.a1)
                     : "v"(tap < 5 ? wb0 : wb1), "v"(a0), "v"(a1), "n"(ob), "n"(oa));
}
template <int N>
__device__ __forceinline__ void c64_fence(C64Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.b), "+v"(f.a0), "+v"(f.a1) : "n"(N));
}
template <int IT>
__device__ __forceinline__ void c64_items(C64Frag (&f)[3], const s16x8_t (&wreg)[9][C64_WREG], unsigned wb0, unsigned wb1,
                                          unsigned a0, unsigned a1, f32x16_t& acc0, f32x16_t& acc1) {
    if constexpr (IT + 2 < 36) c64_read<IT + 2>(f[(IT + 2) % 3], wb0, wb1, a0, a1);
    // reads still in flight behind item IT's: those of items IT + 1 and IT + 2
    c64_fence<(IT + 1 < 36 ? c64_nreads<IT + 1>() : 0) + (IT + 2 < 36 ? c64_nreads<IT + 2>() : 0)>(f[IT % 3]);
    if constexpr ((IT & 3) < C64_WREG) {
        MM<bf16>::mma(f[IT % 3].a0, wreg[IT >> 2][IT & 3], acc0);
        MM<bf16>::mma(f[IT % 3].a1, wreg[IT >> 2][IT & 3], acc1);
    } else {
        MM<bf16>::mma(f[IT % 3].a0, f[IT % 3].b, acc0);
        MM<bf16>::mma(f[IT % 3].a1, f[IT % 3].b, acc1);
    }
    if constexpr (IT + 1 < 36) c64_items<IT + 1>(f, wreg, wb0, wb1, a0, a1, acc0, acc1);
}

template <bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_c64_kernel(Conv3P p, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem64[];
    bf16* wl = reinterpret_cast<bf16*>(smem64);                           // [9][64 co][72]
    bf16* patch = reinterpret_cast<bf16*>(smem64 + C64_WBYTES);           // [PH][ROWP]
    float* stage_all = reinterpret_cast<float*>(smem64 + C64_WBYTES);     // overlays the patch between two barriers
    const int tid = threadIdx.x, wave = tid >> 6, wm = wave & 3, wn = wave >> 2, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const bf16* wp = reinterpret_cast<const bf16*>(p.wp);                 // [Co][9][Ci]

    const int per_b = p.tiles_f * p.tiles_t, total = p.B * per_b;
    const int tile0 = blockIdx.x * tiles_per_wg, tile1 = min(total, tile0 + tiles_per_wg);
    if (tile0 >= tile1) return;
    Vec16<bf16> pre[C64_NIT];
    auto prefetch = [&](int tile) {
        const int b = tile / per_b, rr = tile - b * per_b, tf = rr / p.tiles_t, tt = rr - tf * p.tiles_t;
        const bf16* x = reinterpret_cast<const bf16*>(p.x) + (int64_t)b * p.F * p.T * 64;
#pragma unroll
        for (int it = 0; it < C64_NIT; ++it) {
            const int idx = tid + it * 512, pp = idx >> 3, ch = idx & 7;
            const int pf = pp / PW, pt = pp - pf * PW;
            const int f = tf * TH - 1 + pf, t = tt * TW - 1 + pt;
            pre[it] = ld16_pred<bf16>(x + ((int64_t)f * p.T + t) * 64 + ch * 8, x,
                                      idx < C64_NPC && f >= 0 && f < p.F && t >= 0 && t < p.T);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int it = 0; it < C64_NIT; ++it) {
            const int idx = tid + it * 512;
            if (idx < C64_NPC) { const int pix = idx >> 3; st16<bf16>(patch + (pix / PW) * C64_ROWP + (pix % PW) * C64_LDC + (idx & 7) * 8, pre[it]); }
        }
    };
    prefetch(tile0);
    // weights: 9 * 64 rows of 64 channels = 4608 16-byte pieces, [co][tap][ci] in memory -> [tap][co][72] in LDS
    {   // (all nine loads of a thread in flight before the first store: as a rolled load -> store loop this prologue
        // was nine serial round trips to L2, a third of the launch at B = 16)
        Vec16<bf16> wv[9];
#pragma unroll
        for (int it = 0; it < 9; ++it) wv[it] = ld16<bf16>(wp + (int64_t)(tid + it * 512) * 8);
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int idx = tid + it * 512, row = idx >> 3, ch = idx & 7, co = row / 9, tap = row - co * 9;
            st16<bf16>(wl + (tap * 64 + co) * C64_LDC + ch * 8, wv[it]);
        }
    }
    float bsv[1];
    conv3_bias<1>(p, wn * 32, bsv);
    RunStat run{0.f, 0.f, 0.f, 0.f, false};
    int run_b = -1;
    const int nslot = (per_b + tiles_per_wg - 1) / tiles_per_wg + 1;
    auto flush = [&]() {
        if (!run.open || !p.partial) return;
        const float a1 = run.s1 + __shfl_xor(run.s1, 32, 64), a2 = run.s2 + __shfl_xor(run.s2, 32, 64);
        if (lh == 0) {
            const float dm = run.n > 0.f ? a1 / run.n : 0.f;
            const int slot = blockIdx.x - (run_b * per_b) / tiles_per_wg;
            float* d = p.partial + ((((int64_t)run_b * nslot + slot) * 4 + wm) * 64 + wn * 32 + lr) * 3;
            d[0] = run.piv + dm; d[1] = fmaxf(a2 - a1 * dm, 0.f); d[2] = run.n;
        }
        run.open = false;
    };
    park();
    __syncthreads();
    if (tile0 + 1 < tile1) prefetch(tile0 + 1);
    s16x8_t wreg[9][C64_WREG];                              // the weight fragments of the first k-steps stay in registers:
#pragma unroll                                              // the MFMA phase is bound by LDS bandwidth (3 reads per 2 MFMAs)
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int s = 0; s < C64_WREG; ++s) wreg[tap][s] = MM<bf16>::ldfrag(wl + (tap * 64 + wn * 32 + lr) * C64_LDC + s * 16 + lh * 8);
    // LDS byte addresses of the lane's fragment pieces: weights (two bases, see c64_read), patch rows of its two halves
    const unsigned wb0 = (unsigned)(uintptr_t)(c64_lptr_t)wl + ((wn * 32 + lr) * C64_LDC + lh * 8) * 2;
    const unsigned wb1 = wb0 + 5 * 64 * C64_LDC * 2;
    const unsigned pb = (unsigned)(uintptr_t)(c64_lptr_t)patch + ((lr & 15) * C64_LDC + lh * 8) * 2;
    const unsigned ab0 = pb + (wm * 4 + (lr >> 4)) * C64_ROWP * 2, ab1 = ab0 + 2 * C64_ROWP * 2;
    for (int tile = tile0; tile < tile1; ++tile) {
        const int b = tile / per_b, rr = tile - b * per_b, tf = rr / p.tiles_t, tt = rr - tf * p.tiles_t;
        if (b != run_b) { flush(); run_b = b; }
        f32x16_t acc[2][1];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
        C64Frag fr[3];
        asm volatile("" ::: "memory");                      // (the asm reads below are LDS accesses the compiler cannot see)
        c64_read<0>(fr[0], wb0, wb1, ab0, ab1);
        c64_read<1>(fr[1], wb0, wb1, ab0, ab1);
        c64_items<0>(fr, wreg, wb0, wb1, ab0, ab1, acc[0][0], acc[1][0]);
        asm volatile("" ::: "memory");
        __syncthreads();                                    // every wave is done reading patch(i): the slabs may overlay it
        conv3_epilogue<bf16, 1, RES>(p, acc, stage_all + wave * 64 * C64_SW, bsv, b, tf, tt, wn * 32, p.partial ? &run : nullptr);
        __syncthreads();                                    // every wave is done with its slab
        if (tile + 1 < tile1) park();
        if (tile + 2 < tile1) prefetch(tile + 2);
        __syncthreads();                                    // patch(i+1) is visible
    }
    flush();
}

// stats[b][c] = {mean, rstd} from the per-(workgroup, wave) partials of conv3x3_c32_kernel: a few dozen
// {mean, M2, n} triples per channel, merged in two plain passes by one thread per channel.
__global__ __launch_bounds__(1024) void conv_c32_stats_finalize(const float* __restrict__ partial, float* __restrict__ stats,
                                                                int per_b, int tiles_per_wg, int total, float ntot, float eps,
                                                                int C) {
    __shared__ float sh[32][33];
    __shared__ float gmean[32];
    // C channels (32: conv3x3_c32_kernel, 64: conv3x3_c64_kernel), 32 per workgroup (blockIdx.y)
    const int cl = threadIdx.x & 31, c = blockIdx.y * 32 + cl, g = threadIdx.x >> 5, b = blockIdx.x;
    const int nslot = (per_b + tiles_per_wg - 1) / tiles_per_wg + 1;
    const int wg0 = (b * per_b) / tiles_per_wg, wg1 = min(((b + 1) * per_b - 1) / tiles_per_wg, (total - 1) / tiles_per_wg);
    const int nitem = (wg1 - wg0 + 1) * 4;                 // (slot, wave) pairs, contiguous in the buffer
    const float* base = partial + (int64_t)b * nslot * 4 * C * 3 + c * 3;
    float acc = 0.f;
    for (int i = g; i < nitem; i += 32) { const float* d = base + (int64_t)i * C * 3; acc += d[2] * d[0]; }
    sh[g][cl] = acc;
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += sh[q][cl];
        gmean[cl] = t / ntot;
    }
    __syncthreads();
    const float mean = gmean[cl];
    acc = 0.f;
    for (int i = g; i < nitem; i += 32) {
        const float* d = base + (int64_t)i * C * 3;
        const float dd = d[0] - mean;
        acc += d[2] > 0.f ? d[1] + d[2] * dd * dd : 0.f;
    }
    sh[g][cl] = acc;
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += sh[q][cl];
        stats[((int64_t)b * C + c) * 2] = mean;
        stats[((int64_t)b * C + c) * 2 + 1] = rsqrtf(t / ntot + eps);
    }
}

// merge the per-slab partials -> stats[b][c] = {mean, rstd}.  One 1024-thread workgroup per (32
// channels, b); two passes of plain sums (global mean, then M2 about it) -- no serial Chan chain.
__global__ __launch_bounds__(1024) void conv_stats_finalize(const float* __restrict__ partial, float* __restrict__ stats,
                                                            int F, int T, int C, int tiles_f, int tiles_t, float eps) {
    __shared__ float sh[32][33];
    __shared__ float gmean[32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
    const int nslab = tiles_f * 4 * tiles_t;
    const float* base = partial + (int64_t)b * nslab * C * 2 + (c < C ? c : 0) * 2;
    const float ntot = (float)F * (float)T;
    float acc = 0.f;
    for (int s0 = g; s0 < nslab; s0 += 128) {
        float2 v[4]; float nb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = min(s0 + u * 32, nslab - 1);
            v[u] = *reinterpret_cast<const float2*>(base + (int64_t)s * C * 2);
            const int sf = s / tiles_t, tt = s - sf * tiles_t;
            nb[u] = (s0 + u * 32 < nslab) ? (float)(min(4, max(0, F - sf * 4)) * min(TW, T - tt * TW)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += nb[u] * v[u].x;
    }
    sh[g][cl] = acc;
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += sh[q][cl];
        gmean[cl] = t / ntot;
    }
    __syncthreads();
    const float mean = gmean[cl];
    acc = 0.f;
    for (int s0 = g; s0 < nslab; s0 += 128) {
        float2 v[4]; float nb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = min(s0 + u * 32, nslab - 1);
            v[u] = *reinterpret_cast<const float2*>(base + (int64_t)s * C * 2);
            const int sf = s / tiles_t, tt = s - sf * tiles_t;
            nb[u] = (s0 + u * 32 < nslab) ? (float)(min(4, max(0, F - sf * 4)) * min(TW, T - tt * TW)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float d = v[u].x - mean; acc += nb[u] > 0.f ? v[u].y + nb[u] * d * d : 0.f; }
    }
    sh[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += sh[q][cl];
        stats[((int64_t)b * C + c) * 2] = mean;
        stats[((int64_t)b * C + c) * 2 + 1] = rsqrtf(t / ntot + eps);
    }
}

// ------------------------------------------------------------------------------------
// weight gradient, bf16.  WG = 4 waves; CT = channels per side held in LDS (32 or 64).
//   CT == 32: the 4 waves split the tile's pixels (4 rows each), every wave holds 9 taps.
//   CT == 64: waves 2x2 over (co half, ci half), every wave sees all 256 pixels.
// Each WG walks a range of pixel tiles accumulating in registers, then adds atomically.
// ------------------------------------------------------------------------------------
struct Wg3P {
    const void* dy; const void* x; float* dw;
    int B, F, T, Ci, Co, tiles_f, tiles_t, tiles_per_wg;
    float* ws;      // NULL: atomicAdd into dw; else partial sums as plain stores, slab [Co][9][Ci] per split (wg3_sum_kernel)
};

template <int CT>
__global__ __launch_bounds__(256, CT == 32 ? 2 : 1) void conv3x3_wgrad_kernel(Wg3P p) {
    constexpr int LD = CT + 8;                               // bf16 elements; rows stay 16-byte aligned
    constexpr int CPR = CT / 8, NDY = TH * TW * CPR / 256, NX = (PH * PW * CPR + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* dyl = reinterpret_cast<bf16*>(smem);               // [TH*TW][LD]
    bf16* xl = dyl + TH * TW * LD;                           // [PH*PW][LD]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int co0 = blockIdx.y * CT, ci0 = blockIdx.z * CT;
    const int wm = (CT == 64) ? (wave >> 1) : 0, wn = (CT == 64) ? (wave & 1) : 0;
    const bf16* dy = reinterpret_cast<const bf16*>(p.dy);
    const bf16* x = reinterpret_cast<const bf16*>(p.x);
    // lane roles for the transposing read: this lane supplies k = 8h + 4j + q of a 16-pixel k-step
    const int g = lane >> 4, h = lane >> 5, q = (lane & 15) >> 2, pq = lane & 3;
    const int colofs = 16 * (g & 1) + 4 * pq;
    const int kc0 = 8 * h + q, kc1 = kc0 + 4;

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ntiles = p.B * p.tiles_f * p.tiles_t;
    const int tile0 = blockIdx.x * p.tiles_per_wg, tile1 = min(ntiles, tile0 + p.tiles_per_wg);
    if (tile0 >= tile1) return;
    // next tile's dy tile and x patch wait in registers while this one is in the MFMAs
    Vec16<bf16> dyr[NDY], xr[NX];
    auto prefetch = [&](int tile) {
        int r = tile;
        const int tt = r % p.tiles_t; r /= p.tiles_t;
        const int tf = r % p.tiles_f, b = r / p.tiles_f;
        const int f0 = tf * TH, t0 = tt * TW;
#pragma unroll
        for (int it = 0; it < NDY; ++it) {
            const int idx = tid + it * 256, pp = idx / CPR, ch = idx % CPR;
            const int f = f0 + (pp >> 4), t = t0 + (pp & 15);
            dyr[it] = ld16_pred<bf16>(dy + (((int64_t)b * p.F + f) * p.T + t) * p.Co + co0 + ch * 8, dy, f < p.F && t < p.T);
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = tid + it * 256, pp = idx / CPR, ch = idx % CPR;
            const int pf = pp / PW, pt = pp - pf * PW;
            const int f = f0 - 1 + pf, t = t0 - 1 + pt;
            xr[it] = ld16_pred<bf16>(x + (((int64_t)b * p.F + f) * p.T + t) * p.Ci + ci0 + ch * 8, x,
                                     idx < PH * PW * CPR && f >= 0 && f < p.F && t >= 0 && t < p.T);
        }
    };
    prefetch(tile0);
    // rows of the tile (= k-steps) this wave contracts over
    const int kr_lo = (CT == 64) ? 0 : wave * 4, kr_hi = (CT == 64) ? TH : wave * 4 + 4;
    for (int tile = tile0; tile < tile1; ++tile) {
#pragma unroll
        for (int it = 0; it < NDY; ++it) {
            const int idx = tid + it * 256;
            st16<bf16>(dyl + (idx / CPR) * LD + (idx % CPR) * 8, dyr[it]);
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = tid + it * 256;
            if (idx < PH * PW * CPR) st16<bf16>(xl + (idx / CPR) * LD + (idx % CPR) * 8, xr[it]);
        }
        __syncthreads();
        if (tile + 1 < tile1) prefetch(tile + 1);
        // Walk the PATCH rows R: the x fragment of (row R, column shift kw) feeds the taps kh = 0..2 of
        // the output rows kr = R - kh, so each is read from LDS once instead of three times.
        s16x8_t a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0, a2 = a0;     // dy fragments of rows R, R-1, R-2
        // CT == 64: bounds are compile-time (0 .. TH + 1), the walk is fully unrolled and the row guards fold away
#pragma unroll
        for (int Ri = 0; Ri < (CT == 64 ? TH + 2 : 6); ++Ri) {
            constexpr int NR = CT == 64 ? TH : 4;          // output rows this wave contracts over
            const int R = kr_lo + Ri;
            a2 = a1; a1 = a0;
            if (Ri < NR) {
                const bf16* ap = dyl + (R * 16 + kc0) * LD + wm * 32 + colofs;
                a0 = tr_frag(ap, ap + 4 * LD);
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bf16* bp = xl + (R * PW + kc0 + kw) * LD + wn * 32 + colofs;
                const s16x8_t bfv = tr_frag(bp, bp + 4 * LD);
                const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, bfv);
                if (Ri < NR)                                     // kh = 0, kr = R
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0), bb, acc[kw], 0, 0, 0);
                if (Ri >= 1 && Ri - 1 < NR)                      // kh = 1
                    acc[3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), bb, acc[3 + kw], 0, 0, 0);
                if (Ri >= 2)                                     // kh = 2
                    acc[6 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a2), bb, acc[6 + kw], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- partial sums out: rows = co, cols = ci.  Atomics into dw, or (CT == 64: one writer per element and
    //      split) plain stores into this split's slab of the workspace
    const int lr = lane & 31;
    float* out = p.dw;
    const bool plain = CT == 64 && p.ws != nullptr;
    if (plain) out = p.ws + (int64_t)blockIdx.x * p.Co * 9 * p.Ci;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int co = co0 + wm * 32 + i, ci = ci0 + wn * 32 + lr;
            if (co < p.Co && ci < p.Ci) {
                float* dst = out + ((int64_t)co * 9 + tap) * p.Ci + ci;
                if (plain) *dst = acc[tap][r]; else atomicAdd(dst, acc[tap][r]);
            }
        }
}

// dw[e] = sum over the splits' slabs (overwrites dw).  256 threads = 16 float4 columns x 16 slab groups.
// Ci > 0: dw is written in the PARAMETER's layout [Co][Ci][3][3] (nn.Conv2d.weight) instead of the kernels'
// [Co][9][Ci]: the permute copy per convolution that the caller would otherwise make disappears.
__global__ __launch_bounds__(256) void wg3_sum_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t quads,
                                                      int nslab, int Ci) {
    __shared__ float4 sh[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int64_t qd = (int64_t)blockIdx.x * 16 + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qd < quads)
        for (int s = g; s < nslab; s += 16) {
            const float4 v = *reinterpret_cast<const float4*>(ws + ((int64_t)s * quads + qd) * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    sh[g][c] = a;
    __syncthreads();
    if (g == 0 && qd < quads) {
        float4 t = sh[0][c];
#pragma unroll
        for (int i = 1; i < 16; ++i) { const float4 v = sh[i][c]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        if (Ci <= 0) {
            *reinterpret_cast<float4*>(dw + qd * 4) = t;
        } else {
            const int64_t flat = qd * 4, row = flat / Ci;             // row = co * 9 + tap, 4 | Ci
            const int ci = (int)(flat - row * Ci), tap = (int)(row % 9);
            float* d = dw + ((row / 9) * Ci + ci) * 9 + tap;
            d[0] = t.x; d[9] = t.y; d[18] = t.z; d[27] = t.w;
        }
    }
}

// Ci = Co = 32 variant: wave w < 3 owns the column shift kw = w (taps kh = 0..2: 48 accumulator
// registers instead of 144), walks all 18 patch rows with the dy fragments of three rows rolling in
// registers.  ~130 VGPRs -> three workgroups per CU keep three tiles of loads in flight (the kernel
// is HBM-bound: 530 MB per launch).
__global__ __launch_bounds__(256, 3) void conv3x3_wgrad32_kernel(Wg3P p) {
    constexpr int CT = 32, LD = CT + 8, CPR = CT / 8, NDY = TH * TW * CPR / 256, NX = (PH * PW * CPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16 dyl[TH * TW * LD];
    __shared__ __attribute__((aligned(16))) bf16 xl[PH * PW * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bf16* dy = reinterpret_cast<const bf16*>(p.dy);
    const bf16* x = reinterpret_cast<const bf16*>(p.x);
    const int g = lane >> 4, h = lane >> 5, q = (lane & 15) >> 2, pq = lane & 3;
    const int colofs = 16 * (g & 1) + 4 * pq;
    const int kc0 = 8 * h + q;
    f32x16_t acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int ntiles = p.B * p.tiles_f * p.tiles_t;
    const int tile0 = blockIdx.x * p.tiles_per_wg, tile1 = min(ntiles, tile0 + p.tiles_per_wg);
    if (tile0 >= tile1) return;
    Vec16<bf16> dyr[NDY], xr[NX];
    auto prefetch = [&](int tile) {
        int r = tile;
        const int tt = r % p.tiles_t; r /= p.tiles_t;
        const int tf = r % p.tiles_f, b = r / p.tiles_f;
        const int f0 = tf * TH, t0 = tt * TW;
#pragma unroll
        for (int it = 0; it < NDY; ++it) {
            const int idx = tid + it * 256, pp = idx / CPR, ch = idx % CPR;
            const int f = f0 + (pp >> 4), t = t0 + (pp & 15);
            dyr[it] = ld16_pred<bf16>(dy + (((int64_t)b * p.F + f) * p.T + t) * CT + ch * 8, dy, f < p.F && t < p.T);
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = tid + it * 256, pp = idx / CPR, ch = idx % CPR;
            const int pf = pp / PW, pt = pp - pf * PW;
            const int f = f0 - 1 + pf, t = t0 - 1 + pt;
            xr[it] = ld16_pred<bf16>(x + (((int64_t)b * p.F + f) * p.T + t) * CT + ch * 8, x,
                                     idx < PH * PW * CPR && f >= 0 && f < p.F && t >= 0 && t < p.T);
        }
    };
    prefetch(tile0);
    const int kw = wave;
    for (int tile = tile0; tile < tile1; ++tile) {
#pragma unroll
        for (int it = 0; it < NDY; ++it) {
            const int idx = tid + it * 256;
            st16<bf16>(dyl + (idx / CPR) * LD + (idx % CPR) * 8, dyr[it]);
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = tid + it * 256;
            if (idx < PH * PW * CPR) st16<bf16>(xl + (idx / CPR) * LD + (idx % CPR) * 8, xr[it]);
        }
        __syncthreads();
        if (tile + 1 < tile1) prefetch(tile + 1);
        if (wave < 3) {
            s16x8_t a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0, a2 = a0;     // dy fragments of tile rows R, R-1, R-2
#pragma unroll
            for (int R = 0; R < TH + 2; ++R) {
                a2 = a1; a1 = a0;
                if (R < TH) {
                    const bf16* ap = dyl + (R * 16 + kc0) * LD + colofs;
                    a0 = tr_frag(ap, ap + 4 * LD);
                }
                const bf16* bp = xl + (R * PW + kc0 + kw) * LD + colofs;
                const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, tr_frag(bp, bp + 4 * LD));
                if (R < TH) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0), bb, acc[0], 0, 0, 0);
                if (R >= 1 && R - 1 < TH) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), bb, acc[1], 0, 0, 0);
                if (R >= 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a2), bb, acc[2], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (wave < 3) {
        const int lr = lane & 31;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p.ws) p.ws[(int64_t)blockIdx.x * (CT * 9 * CT) + ((int64_t)co * 9 + kh * 3 + kw) * CT + lr] = acc[kh][r];
                else atomicAdd(p.dw + ((int64_t)co * 9 + kh * 3 + kw) * CT + lr, acc[kh][r]);
            }
    }
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

// y = conv3x3_same(x, wp [Co][9][Ci]) + bias (+ res); if partial != NULL also the InstanceNorm
// statistics: partial scratch of B*(4*tiles_f*tiles_t+4)*Co*3 floats, stats [B][Co][2] = {mean, rstd}
// (stats == NULL with partial != NULL: the partials are produced but not merged -- used to time the kernel alone).
extern "C" int dtt_conv3x3_s1(const void* x, const void* wp, const float* bias, void* y, const void* res,
                              float* partial, float* stats, int B, int F, int T, int Ci, int Co, float eps, int dtype,
                              void* stream) {
    const int ept = dtype ? 8 : 4;
    DTT_CHECK(B > 0 && F > 0 && T > 0 && Ci % ept == 0 && Co % 32 == 0, "dtt_conv3x3_s1: unsupported Ci=%d Co=%d", Ci, Co);
    Conv3P p{x, wp, bias, y, res, partial, B, F, T, Ci, Co, (F + TH - 1) / TH, (T + TW - 1) / TW};
    if (dtype == 1 && Ci == 32 && Co == 32) {
        const int total = B * p.tiles_f * p.tiles_t;
        const int tpw = max(1, (total + 511) / 512);         // ONE resident wave of workgroups: 2 per CU x 256 CUs
        if (res) hipLaunchKernelGGL(conv3x3_c32_kernel<true>, dim3((total + tpw - 1) / tpw), dim3(256), 0, ST, p, tpw);
        else hipLaunchKernelGGL(conv3x3_c32_kernel<false>, dim3((total + tpw - 1) / tpw), dim3(256), 0, ST, p, tpw);
        if (partial && stats)
            hipLaunchKernelGGL(conv_c32_stats_finalize, dim3(B), dim3(1024), 0, ST, partial, stats, p.tiles_f * p.tiles_t,
                               tpw, total, (float)F * (float)T, eps, 32);
        DTT_LAUNCH_CHECK();
        return 0;
    }
    static const bool c64_on = [] { const char* e = getenv("DTT_CONV_C64"); return !(e && e[0] == '0'); }();
    if (dtype == 1 && Ci == 64 && Co == 64 && c64_on) {
        const int total = B * p.tiles_f * p.tiles_t;
        const int tpw = max(1, (total + 255) / 256);         // one persistent workgroup per CU
        auto kr = conv3x3_c64_kernel<true>;
        auto kn = conv3x3_c64_kernel<false>;
        static bool attr_set = false;
        if (!attr_set) {
            dtt_want_lds((const void*)kr, (int)C64_SMEM);
            dtt_want_lds((const void*)kn, (int)C64_SMEM);
            attr_set = true;
        }
        if (res) hipLaunchKernelGGL(kr, dim3((total + tpw - 1) / tpw), dim3(512), C64_SMEM, ST, p, tpw);
        else hipLaunchKernelGGL(kn, dim3((total + tpw - 1) / tpw), dim3(512), C64_SMEM, ST, p, tpw);
        if (partial && stats)
            hipLaunchKernelGGL(conv_c32_stats_finalize, dim3(B, 2), dim3(1024), 0, ST, partial, stats, p.tiles_f * p.tiles_t,
                               tpw, total, (float)F * (float)T, eps, 64);
        DTT_LAUNCH_CHECK();
        return 0;
    }
    // 64 output channels per workgroup (one LDS fragment read per MFMA, two workgroups per CU) unless that leaves
    // the launch with fewer than ~4.5 workgroups per CU: then 32 channels (three per CU, twice the workgroups) fills
    // the last round better -- B = 16 at 64 x 257 x 63 / 128 x 129 x 32 / 256 x 65 x 16: 42.9 -> 41.6, 38.9 -> 37.0,
    // 41.0 -> 38.1 us; at B = 64 (4352 workgroups) the 64-channel form stays 1-6 % ahead.
    int tn = (Co % 64 == 0) ? 2 : 1;
    if (tn == 2 && dtype == 1 && (int64_t)p.tiles_f * p.tiles_t * (Co / 64) * B <= 1152) tn = 1;
    const int bn = 32 * tn;
    const int ldc = KC + (dtype ? 8 : 4), es = dtype ? 2 : 4;
    size_t sm = (size_t)PH * (((size_t)PW * ldc * es + 255) / 256 * 256) + (size_t)9 * bn * ldc * es;
    sm = max(sm, (size_t)4 * 64 * (bn + 4) * 4);             // output staging overlays the operands
    dim3 grid(p.tiles_f * p.tiles_t * (Co / bn), B);
#define LAUNCH(TT, TN_) do { auto k = conv3x3_s1_kernel<TT, TN_>; \
        if (sm > 64 * 1024) dtt_want_lds((const void*)k, (int)sm); \
        hipLaunchKernelGGL(k, grid, dim3(256), sm, ST, p); } while (0)
    if (dtype) { if (tn == 2) LAUNCH(bf16, 2); else LAUNCH(bf16, 1); }
    else { if (tn == 2) LAUNCH(float, 2); else LAUNCH(float, 1); }
#undef LAUNCH
    if (partial && stats)
        hipLaunchKernelGGL(conv_stats_finalize, dim3((Co + 31) / 32, B), dim3(1024), 0, ST, partial, stats, F, T, Co,
                           p.tiles_f, p.tiles_t, eps);
    DTT_LAUNCH_CHECK();
    return 0;
}

namespace {
// split geometry of the weight-gradient kernels: ct = channel tile, nsplit = workgroups along the pixel tiles (= slabs
// of partial sums), tpw = tiles per workgroup
struct WgGeom { int ct, nsplit, tpw; bool c32; };
WgGeom wg_geom(int B, int F, int T, int Ci, int Co) {
    WgGeom g;
    const int ntiles = B * ((F + TH - 1) / TH) * ((T + TW - 1) / TW);
    g.c32 = Ci == 32 && Co == 32;
    if (g.c32) {
        g.ct = 32;
        g.tpw = (ntiles + 767) / 768;                       // one resident wave: 3 workgroups per CU
        g.nsplit = (ntiles + g.tpw - 1) / g.tpw;
        return g;
    }
    g.ct = (Ci % 64 == 0 && Co % 64 == 0) ? 64 : 32;
    const int combos = (Co / g.ct) * (Ci / g.ct);
    // one resident wave of workgroups (1 per CU at CT = 64, 2 at CT = 32): each workgroup ends with 9*CT*CT/256
    // partial sums per lane, so fewer, longer workgroups beat a finer split
    const int target = g.ct == 64 ? 256 : 512;
    int nsplit = max(1, min(ntiles, (target + combos - 1) / combos));
    g.tpw = (ntiles + nsplit - 1) / nsplit;
    g.nsplit = (ntiles + g.tpw - 1) / g.tpw;                // no empty workgroup: every slab is written in full
    return g;
}

int wgrad_launch(const void* dy, const void* x, float* dw, float* ws, int B, int F, int T, int Ci, int Co, void* stream,
                 bool param_layout = false) {
    const WgGeom g = wg_geom(B, F, T, Ci, Co);
    Wg3P p{dy, x, dw, B, F, T, Ci, Co, (F + TH - 1) / TH, (T + TW - 1) / TW, g.tpw, ws};
    if (g.c32) {
        hipLaunchKernelGGL(conv3x3_wgrad32_kernel, dim3(g.nsplit), dim3(256), 0, ST, p);
    } else {
        size_t sm = (size_t)(TH * TW + PH * PW) * (g.ct + 8) * 2;
        dim3 grid(g.nsplit, Co / g.ct, Ci / g.ct);
        if (g.ct == 64) {
            auto k = conv3x3_wgrad_kernel<64>;
            dtt_want_lds((const void*)k, (int)sm);
            hipLaunchKernelGGL(k, grid, dim3(256), sm, ST, p);
        } else {
            p.ws = nullptr;                                 // four waves per element: atomics
            hipLaunchKernelGGL(conv3x3_wgrad_kernel<32>, grid, dim3(256), sm, ST, p);
        }
    }
    if (p.ws) {
        const int64_t quads = (int64_t)Co * 9 * Ci / 4;
        hipLaunchKernelGGL(wg3_sum_kernel, dim3((unsigned)((quads + 15) / 16)), dim3(256), 0, ST, ws, dw, quads, g.nsplit,
                           param_layout ? Ci : 0);
    }
    DTT_LAUNCH_CHECK();
    return 0;
}
}  // namespace

// dw [Co][9][Ci] fp32 (zeroed by the caller) += sum over pixels; bf16 only.
extern "C" int dtt_conv3x3_wgrad(const void* dy, const void* x, float* dw, int B, int F, int T, int Ci, int Co,
                                 int dtype, void* stream) {
    DTT_CHECK(dtype == 1, "dtt_conv3x3_wgrad: bf16 only (fp32 goes through dtt_gemm)");
    DTT_CHECK(B > 0 && F > 0 && T > 0 && Ci % 32 == 0 && Co % 32 == 0, "dtt_conv3x3_wgrad: unsupported Ci=%d Co=%d", Ci, Co);
    return wgrad_launch(dy, x, dw, nullptr, B, F, T, Ci, Co, stream);
}

// The same product with the workgroups' partial sums stored to a caller workspace and summed by a second launch
// instead of fp32 atomics (a third of the kernel's time at 64 -> 64 channels).  dw is OVERWRITTEN (no zeroing
// needed).  dtt_conv3x3_wgrad_ws_bytes: size of the workspace for a shape, 0 where the atomic form is the only one
// (channel counts that are multiples of 32 but not of 64, other than 32 -> 32).
extern "C" int64_t dtt_conv3x3_wgrad_ws_bytes(int B, int F, int T, int Ci, int Co) {
    if (B <= 0 || F <= 0 || T <= 0 || Ci % 32 || Co % 32) return 0;
    const WgGeom g = wg_geom(B, F, T, Ci, Co);
    if (!g.c32 && g.ct != 64) return 0;
    return (int64_t)g.nsplit * Co * 9 * Ci * 4;
}
extern "C" int dtt_conv3x3_wgrad_ws(const void* dy, const void* x, float* dw, void* ws, int64_t ws_bytes, int B, int F,
                                    int T, int Ci, int Co, int dtype, void* stream) {
    DTT_CHECK(dtype == 1, "dtt_conv3x3_wgrad_ws: bf16 only (fp32 goes through dtt_gemm)");
    DTT_CHECK(B > 0 && F > 0 && T > 0 && Ci % 32 == 0 && Co % 32 == 0, "dtt_conv3x3_wgrad_ws: unsupported Ci=%d Co=%d", Ci, Co);
    const int64_t need = dtt_conv3x3_wgrad_ws_bytes(B, F, T, Ci, Co);
    DTT_CHECK(need > 0, "dtt_conv3x3_wgrad_ws: no workspace form for Ci=%d Co=%d (use dtt_conv3x3_wgrad)", Ci, Co);
    DTT_CHECK(ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
              "dtt_conv3x3_wgrad_ws: workspace of %lld bytes needed", (long long)need);
    return wgrad_launch(dy, x, dw, reinterpret_cast<float*>(ws), B, F, T, Ci, Co, stream);
}
// As dtt_conv3x3_wgrad_ws with dw written in nn.Conv2d.weight's own layout [Co][Ci][3][3].
extern "C" int dtt_conv3x3_wgrad_ws_param(const void* dy, const void* x, float* dw, void* ws, int64_t ws_bytes, int B, int F,
                                          int T, int Ci, int Co, int dtype, void* stream) {
    DTT_CHECK(dtype == 1, "dtt_conv3x3_wgrad_ws_param: bf16 only (fp32 goes through dtt_gemm)");
    DTT_CHECK(B > 0 && F > 0 && T > 0 && Ci % 32 == 0 && Co % 32 == 0, "dtt_conv3x3_wgrad_ws_param: unsupported Ci=%d Co=%d", Ci, Co);
    const int64_t need = dtt_conv3x3_wgrad_ws_bytes(B, F, T, Ci, Co);
    DTT_CHECK(need > 0, "dtt_conv3x3_wgrad_ws_param: no workspace form for Ci=%d Co=%d (use dtt_conv3x3_wgrad)", Ci, Co);
    DTT_CHECK(ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
              "dtt_conv3x3_wgrad_ws_param: workspace of %lld bytes needed", (long long)need);
    return wgrad_launch(dy, x, dw, reinterpret_cast<float*>(ws), B, F, T, Ci, Co, stream, true);
}
