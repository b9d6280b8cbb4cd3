// 256 x 256 x 64 bf16 GEMM for the large plain products of the step (A [M][K] and B [N][K] both contiguous along k,
// C [M][N] contiguous along n): the LSTM input projections x W_ih^T (latent.py:87-92; M = 33 280, N = 4 096,
// K = 256 at cfg4), the TC_FC_RNN output Linear (latent.py:93-96) and their likes.
//
// Why a second GEMM kernel.  gemm_fast_kernel (128 x 128 workgroup tiles, operands staged through registers into
// LDS) sits at 0.20-0.25 of the MFMA peak on every one of these shapes, whatever the split: a 128 x 128 x 64 step
// needs 32 KB of operands for 2 x 128 x 128 x 64 flop = 64 flop per byte, exactly the ratio of a CU's MFMA rate
// (4 096 flop/clk) to its vector-memory path (64 B/clk) -- the tile is bandwidth-bound inside the CU before LDS
// traffic is counted, and its ds_write_b128 stores (79 B/clk) take longer than its MFMAs.  Here:
//   * 256 x 256 tile per workgroup (128 flop per operand byte), 8 waves = 2 per SIMD, wave tile 128 x 64
//     (4 x 2 MFMA tiles of 32x32x16: 6 fragment reads per 8 MFMAs instead of 4 per 4);
//   * operands go global -> LDS by LDS-DMA (no registers, no ds_write), two 64 KB stages, ONE barrier per k-step of 64:
//     the DMA of step kt + 1 is issued before the MFMAs of step kt and waited for behind them;
//   * LDS image [row][64 k] without padding (DMA writes 1 KB per wave-instruction contiguously): the 16-byte piece c
//     of row r is stored at piece c ^ ((r >> 1) & 7) -- the lane's SOURCE address carries the swizzle -- which makes
//     the fragment reads (16 rows x one piece per lane group) conflict-free.
// Epilogue: bias[n], residual, bf16 or fp32 store, through an LDS slab per wave (16-byte stores along n).
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 256, BK = 32, NST = 4;        // four 32 KB stages: the DMA runs three k-steps ahead
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;            // 32 KB
constexpr int NDMA = 4;                                    // DMA wave-instructions per wave and k-step (2 A + 2 B)
constexpr int SLAB_W = 64 + 4;                             // epilogue slab row (floats)

// ds_read_b64_tr_b16 through inline asm: hipcc's wait-count pass makes every LDS read issued through the builtin
// wait for ALL outstanding LDS-DMA (`s_waitcnt vmcnt(0)` in front of the first fragment read of a k-step: the DMA of
// the next stage, issued just before, never overlapped the MFMAs).  The asm form is opaque to that pass.  One
// statement reads the 12 half-fragments of a k-substep (4 A + 2 B fragments, k-rows q and q + 4 each); rc_fence<N>
// waits until at most N younger LDS reads are outstanding (LDS returns in order) and only then hands the registers
// to the compiler (the "+v" operands: nothing may read or copy them before the wait).
struct RcFrag { s16x4_t lo[6], hi[6]; };
template <int OFF>
__device__ __forceinline__ void rc_read(RcFrag& f, const unsigned (&a)[6]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %12 offset:%18\n\tds_read_b64_tr_b16 %1, %12 offset:%19\n\t"
        "ds_read_b64_tr_b16 %2, %13 offset:%18\n\tds_read_b64_tr_b16 %3, %13 offset:%19\n\t"
        "ds_read_b64_tr_b16 %4, %14 offset:%18\n\tds_read_b64_tr_b16 %5, %14 offset:%19\n\t"
        "ds_read_b64_tr_b16 %6, %15 offset:%18\n\tds_read_b64_tr_b16 %7, %15 offset:%19\n\t"
        "ds_read_b64_tr_b16 %8, %16 offset:%18\n\tds_read_b64_tr_b16 %9, %16 offset:%19\n\t"
        "ds_read_b64_tr_b16 %10, %17 offset:%18\n\tds_read_b64_tr_b16 %11, %17 offset:%19"
        : "=&v"(f.lo[0]), "=&v"(f.hi[0]), "=&v"(f.lo[1]), "=&v"(f.hi[1]), "=&v"(f.lo[2]), "=&v"(f.hi[2]),
          "=&v"(f.lo[3]), "=&v"(f.hi[3]), "=&v"(f.lo[4]), "=&v"(f.hi[4]), "=&v"(f.lo[5]), "=&v"(f.hi[5])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "n"(OFF), "n"(OFF + 2048));
}
template <int N>
__device__ __forceinline__ void rc_fence(RcFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]),
                   "+v"(f.lo[3]), "+v"(f.hi[3]), "+v"(f.lo[4]), "+v"(f.hi[4]), "+v"(f.lo[5]), "+v"(f.hi[5])
                 : "n"(N));
}
__device__ __forceinline__ bf16x8_t rc_join(const s16x4_t& lo, const s16x4_t& hi) {
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}

// k-contiguous operands: the six ds_read_b128 of a k-substep in one statement for the same reason, and for a second
// one: left to itself hipcc sinks every fragment read to just in front of its first MFMA and waits lgkmcnt(0) there
// (it reuses 16 fragment registers) -- no LDS read ever ran under an MFMA of the same wave.
struct KcFrag { s16x8_t v[6]; };
__device__ __forceinline__ void kc_read(KcFrag& f, const unsigned (&a)[6]) {
    asm volatile(
        "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\t"
        "ds_read_b128 %3, %9\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %11"
        : "=&v"(f.v[0]), "=&v"(f.v[1]), "=&v"(f.v[2]), "=&v"(f.v[3]), "=&v"(f.v[4]), "=&v"(f.v[5])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]));
}
template <int N>
__device__ __forceinline__ void kc_fence(KcFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]), "+v"(f.v[3]), "+v"(f.v[4]), "+v"(f.v[5]) : "n"(N));
}

// RC: both operands contiguous along their rows, k-major (A(m,k) = A[k * a_ks + m], B(k,n) = B[k * b_ks + n]): the
// weight-gradient products dW = dG^T [h_{t-1} | x] of the LSTMs (K = rows of the step, split over grid.y, fp32
// atomic accumulation).  LDS image [64 k][256 rows], 16-byte piece c of k-row r stored at c ^ ((r & 3) << 2): the
// four k-rows one transposing read (ds_read_b64_tr_b16) touches fall into four different 16-bank slots.
// SHIFT (B only): the LSTM row shift, k -> k + sh_shift, zero where (k / sh_heads) % sh_L == sh_edge.
template <bool RC>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const DttGemm p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = (p.N + BN - 1) / BN, ntile = gridDim.x;
    int tile = blockIdx.x, ks = blockIdx.y;
    if (RC && gridDim.y % 8 == 0) {
        // split-K: an XCD (linear workgroup id % 8) takes whole k-splits, every output tile of them -- the 16 tiles of
        // a k-split read the same K / splitk rows of A and B, so each operand byte crosses the fabric once and is
        // shared through that XCD's L2 (tiles of one A panel per XCD: every XCD streams all of K for its panel,
        // 2.4x the traffic at 2048 x 512 x 33 280)
        const int lin = blockIdx.y * ntile + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
        ks = xcd * ((int)gridDim.y >> 3) + slot / ntile;
        tile = slot % ntile;
    } else {   // XCD-aware order: consecutive logical tiles (which share an A panel) run on one XCD
        const int per = ntile / 8;
        if (tile < per * 8) tile = (tile % 8) * per + tile / 8;
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    const int KT = p.K / BK, per_split = (KT + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt0 = ks * per_split, kt1 = min(KT, kt0 + per_split);
    if (kt0 >= kt1) return;

    // ---- DMA sources.  KC: wave-instruction j fills rows (wave * 2 + j) * 16 .. + 16 of a 256-row image (64 B per
    //      row, 16-byte piece c of row r stored at c ^ ((r >> 2) & 3)); RC: k-rows (wave * 2 + j) * 2, + 1 of a
    //      32-k-row image (512 B per k-row, piece c of k-row r at c ^ ((r & 3) << 2))
    const bf16* pa[2];
    const bf16* pb[2];
    int kb[2];                                             // RC + shift: k-row of the lane's piece inside the tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int L = (wave * 2 + j) * 64 + lane;
        if constexpr (!RC) {
            const int row = L >> 2, pos = L & 3, piece = pos ^ ((row >> 2) & 3);
            pa[j] = A + (int64_t)min(m0 + row, p.M - 1) * p.a_ms + (int64_t)kt0 * BK + piece * 8;
            pb[j] = B + (int64_t)min(n0 + row, p.N - 1) * p.b_ns + (int64_t)kt0 * BK + piece * 8;
        } else {
            const int krow = L >> 5, pos = L & 31, piece = pos ^ ((krow & 3) << 2);
            kb[j] = krow;
            pa[j] = A + (int64_t)(kt0 * BK + krow) * p.a_ks + min(m0 + piece * 8, p.M - 8);
            pb[j] = B + (int64_t)(kt0 * BK + krow + (p.b_mode == 2 ? p.sh_shift : 0)) * p.b_ks + min(n0 + piece * 8, p.N - 8);
        }
    }
    const int64_t stepA = RC ? (int64_t)BK * p.a_ks : BK, stepB = RC ? (int64_t)BK * p.b_ks : BK;
    // Every call issues exactly NDMA wave-instructions (the waits below count them); past the last k-step it
    // re-reads that step into a stage nobody will read.
    int nk = kt0;                                          // k-step the sources point at = min(nk, kt1 - 1)
    auto issue = [&](int stage) {
        unsigned char* sa = smem + stage * STAGE_BYTES + (wave * 2) * 1024;
        const int kt = min(nk, kt1 - 1);
        const bool adv = nk < kt1 - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)pa[j], (lptr_t)(sa + j * 1024), 16, 0, 0);
            pa[j] += adv ? stepA : 0;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16* src = pb[j];
            if constexpr (RC) {
                if (p.b_mode == 2) {
                    const int gk = kt * BK + kb[j];
                    if ((gk / p.sh_heads) % p.sh_L == p.sh_edge) src = reinterpret_cast<const bf16*>(&dtt_zero16);
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + BM * BK * 2 + j * 1024), 16, 0, 0);
            pb[j] += adv ? stepB : 0;
        }
        ++nk;
    };

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses.  KC: row * 64 bytes + ((2 s + lh) ^ swz(row)) * 16.
    // RC (transposing read): k-row 16 s + 8 lh + q, columns base + 16 g + 4 p .. + 3
    int offA[4], offB[2], swA[4], swB[2];
    if constexpr (!RC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = wm * 128 + i * 32 + lr; offA[i] = row * 64; swA[i] = (row >> 2) & 3; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int row = wn * 64 + j * 32 + lr; offB[j] = BM * BK * 2 + row * 64; swB[j] = (row >> 2) & 3; }
    } else {
        const int g = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = wm * 128 + i * 32 + 16 * g + 4 * pp;
            offA[i] = (8 * lh + q) * 512 + (((col >> 3) ^ (q << 2)) << 4) + (col & 7) * 2; swA[i] = 0;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + 16 * g + 4 * pp;
            offB[j] = BM * BK * 2 + (8 * lh + q) * 512 + (((col >> 3) ^ (q << 2)) << 4) + (col & 7) * 2; swB[j] = 0;
        }
    }

    // Pipeline over k-steps of 32 with fragment registers double-buffered across the barrier, so that LDS reads
    // run under MFMAs (eight waves in lockstep behind one barrier per k-step otherwise alternate between an
    // LDS phase and an MFMA phase).  D = NST - 1 steps of DMA lead.  Middle of step i: F0 = (step i, k 0..15) is being
    // multiplied, F1 = (step i, k 16..31) is on its way from LDS; `vmcnt((D - 2) NDMA)`: this wave's pieces of step
    // i + 1 have landed; barrier: everyone's have, and everyone is past its last read of step i - 1, whose stage
    // the DMA of step i + D refills; then F0 <- (step i + 1, k 0..15) under the MFMAs of F1.
    // (Raw s_barrier: __syncthreads' fence would drain vmcnt.)
    constexpr int D = NST - 1;
#define G256_SYNC(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N) : "memory")
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d);
    G256_SYNC((D - 1) * NDMA);
    int cs = 0, fs = D;                                    // stage of the current step, stage to refill
    const int nsteps = kt1 - kt0;
    if constexpr (RC) {
        const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
        unsigned ad[6];
#pragma unroll
        for (int a = 0; a < 4; ++a) ad[a] = lds0 + offA[a];
#pragma unroll
        for (int b = 0; b < 2; ++b) ad[4 + b] = lds0 + offB[b];
        RcFrag f0, f1;
#define RC_MMA(F) do { _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b) \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rc_join(F.lo[a], F.hi[a]), rc_join(F.lo[4 + b], F.hi[4 + b]), acc[a][b], 0, 0, 0); } while (0)
        rc_read<0>(f0, ad);
        for (int i = 0; i < nsteps; ++i) {
            rc_read<8192>(f1, ad);
            rc_fence<12>(f0);
            RC_MMA(f0);
            G256_SYNC((D - 2) * NDMA);
            issue(fs);
            fs = fs + 1 == NST ? 0 : fs + 1;
            cs = cs + 1 == NST ? 0 : cs + 1;
#pragma unroll
            for (int a = 0; a < 4; ++a) ad[a] = lds0 + cs * STAGE_BYTES + offA[a];
#pragma unroll
            for (int b = 0; b < 2; ++b) ad[4 + b] = lds0 + cs * STAGE_BYTES + offB[b];
            rc_read<0>(f0, ad);                            // (past the last step: a stage of re-read data, unused)
            rc_fence<12>(f1);
            RC_MMA(f1);
        }
        rc_fence<0>(f0);
#undef RC_MMA
    } else {
        const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
        unsigned o0[6], o1[6], ad0[6], ad1[6];             // byte offsets of the lane's pieces for k 0..15 / 16..31
#pragma unroll
        for (int a = 0; a < 4; ++a) { o0[a] = offA[a] + ((lh ^ swA[a]) << 4); o1[a] = offA[a] + (((2 + lh) ^ swA[a]) << 4); }
#pragma unroll
        for (int b = 0; b < 2; ++b) { o0[4 + b] = offB[b] + ((lh ^ swB[b]) << 4); o1[4 + b] = offB[b] + (((2 + lh) ^ swB[b]) << 4); }
#pragma unroll
        for (int a = 0; a < 6; ++a) { ad0[a] = lds0 + o0[a]; ad1[a] = lds0 + o1[a]; }
        KcFrag f0, f1;
#define KC_MMA(F) do { _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b) \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, F.v[a]), __builtin_bit_cast(bf16x8_t, F.v[4 + b]), acc[a][b], 0, 0, 0); } while (0)
        kc_read(f0, ad0);
        for (int i = 0; i < nsteps; ++i) {
            kc_read(f1, ad1);
            kc_fence<6>(f0);
            KC_MMA(f0);
            G256_SYNC((D - 2) * NDMA);
            issue(fs);
            fs = fs + 1 == NST ? 0 : fs + 1;
            cs = cs + 1 == NST ? 0 : cs + 1;
#pragma unroll
            for (int a = 0; a < 6; ++a) { ad0[a] = lds0 + cs * STAGE_BYTES + o0[a]; ad1[a] = lds0 + cs * STAGE_BYTES + o1[a]; }
            kc_read(f0, ad0);                              // (past the last step: a stage of re-read data, unused)
            kc_fence<6>(f1);
            KC_MMA(f1);
        }
        kc_fence<0>(f0);
#undef KC_MMA
    }
#undef G256_SYNC
    // the trailing (dummy) DMAs still write LDS: drain them before the slab overlays the stages / the kernel ends
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    if (p.atomic && !p.ws) {
        // split-K partial sums: one register of a 32x32 accumulator is 32 consecutive n in two rows
        float* Cf = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gn = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (gm < p.M && gn < p.N) atomicAdd(Cf + (int64_t)gm * p.c_ms + gn, acc[i][j][r] * p.alpha);
                }
            }
        return;
    }
    // ---- epilogue: one 32 x 64 fp32 slab per wave at a time through LDS, 16-byte stores along n
    // (split-K with a workspace: the partial tile of k-split ks goes to ws[ks][M][N], fp32, summed by splitk_sum_kernel)
    float* slab = reinterpret_cast<float*>(smem) + wave * 32 * SLAB_W;
    const bf16* R = p.ws ? nullptr : reinterpret_cast<const bf16*>(p.R);
    float* Cf32 = p.ws ? reinterpret_cast<float*>(p.ws) + (int64_t)ks * p.M * p.N : reinterpret_cast<float*>(p.C);
    const int64_t ldc = p.ws ? p.N : p.c_ms;
    const bool f32out = p.ws || p.out_f32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + lr;
            const float bv = (p.bias_mode == 1 && gn < p.N && !p.ws) ? p.bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLAB_W + j * 32 + lr] = acc[i][j][r] * p.alpha + bv;
        }
        // (wave-private slab: LDS operations of one wave stay in order, no barrier)
        const int gm0 = m0 + wm * 128 + i * 32, gn0 = n0 + wn * 64;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = lane + it * 64, row = idx >> 3, ch = idx & 7;          // 8 chunks of 8 outputs per row
            const int gm = gm0 + row, gn = gn0 + ch * 8;
            if (gm >= p.M || gn >= p.N) continue;
            float v[8];
            const float4 s0 = *reinterpret_cast<const float4*>(slab + row * SLAB_W + ch * 8);
            const float4 s1 = *reinterpret_cast<const float4*>(slab + row * SLAB_W + ch * 8 + 4);
            v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w; v[4] = s1.x; v[5] = s1.y; v[6] = s1.z; v[7] = s1.w;
            if (R) {
                const Vec16<bf16> rv = ld16<bf16>(R + (int64_t)gm * p.r_ms + gn);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv.get(e);
            }
            if (f32out) {
                float* dst = Cf32 + (int64_t)gm * ldc + gn;
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                Vec16<bf16> ov;
#pragma unroll
                for (int e = 0; e < 8; e += 2) ov.set2(e, v[e], v[e + 1]);
                st16<bf16>(reinterpret_cast<bf16*>(p.C) + (int64_t)gm * ldc + gn, ov);
            }
        }
    }
}

// C[m][n] += sum over the k-splits' partial tiles (one writer per element: no atomics)
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ ws, float* __restrict__ Cm, int M, int N,
                                                         int64_t c_ms, int sk) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, per_row = N >> 2;
    if (q >= (int64_t)M * per_row) return;
    const int m = (int)(q / per_row), n = (int)(q % per_row) << 2;
    const float* src = ws + (int64_t)m * N + n;
    const int64_t slab = (int64_t)M * N;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < sk; s += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)min(s + u, sk - 1) * slab);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s + u < sk) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    float4* dst = reinterpret_cast<float4*>(Cm + (int64_t)m * c_ms + n);
    float4 c = *dst;
    c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
    *dst = c;
}

// split-K factor of the row-contiguous form: one workgroup per CU, never fewer than 8 k-steps of 64 per split, and no
// empty split (the last ones of ceil(KT / sk) steps each)
int rc_splits(const DttGemm& p, int64_t tiles) {
    int sk = (int)std::max<int64_t>(1, std::min<int64_t>(p.K / 512, (256 + tiles - 1) / tiles));
    const int KT = p.K / BK, per = (KT + sk - 1) / sk;
    return (KT + per - 1) / per;
}
bool rc_covered(const DttGemm& p) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return p.dtype == 1 && !p.a_mode && p.batch == 1 && !p.act && p.c_ns == 1 && p.K % BK == 0 && al(p.A) && al(p.B) &&
           al(p.C) && p.M >= 192 && p.N >= 192 && p.a_ms == 1 && p.b_ns == 1 && (p.b_mode == 0 || p.b_mode == 2) &&
           p.atomic && p.out_f32 && !p.bias_mode && !p.R && p.a_ks % 8 == 0 && p.b_ks % 8 == 0 && p.M % 8 == 0 &&
           p.N % 8 == 0 && p.K >= 4096 && p.N >= 256;
}

bool gemm256_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("DTT_GEMM256"); on = e ? atoi(e) : 1; }
    return on != 0;
}

}  // namespace

int64_t dtt_gemm256_ws_bytes(const DttGemm& p) {
    if (!gemm256_enabled() || !rc_covered(p) || p.c_ms % 4) return 0;
    const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int sk = rc_splits(p, tiles);
    return sk < 2 ? 0 : (int64_t)sk * p.M * p.N * 4;
}

// 1: not covered (the caller goes on to the general kernels); 0 / negative as dtt_gemm
int dtt_gemm256_try(const DttGemm& p, hipStream_t st) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!gemm256_enabled() || p.dtype != 1 || p.a_mode || p.batch != 1 || p.act || p.c_ns != 1 || p.K % BK ||
        !al(p.A) || !al(p.B) || !al(p.C) || p.M < 192 || p.N < 192)
        return 1;
    const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int sm = NST * STAGE_BYTES;
    static bool attr = false;
    if (!attr) {
        dtt_want_lds((const void*)gemm256_kernel<false>, sm);
        dtt_want_lds((const void*)gemm256_kernel<true>, sm);
        attr = true;
    }
    const bool kc = p.a_ks == 1 && p.b_ks == 1 && p.b_mode == 0;
    const bool rc = p.a_ms == 1 && p.b_ns == 1 && (p.b_mode == 0 || p.b_mode == 2);
    if (kc) {
        // plain store, k-contiguous operands (DTT_GEMM256_KC=0: leave them to gemmx.hip, which has this tile too)
        static int kc_on = -1;
        if (kc_on < 0) { const char* e = getenv("DTT_GEMM256_KC"); kc_on = e ? atoi(e) : 1; }
        if (!kc_on) return 1;
        if (p.splitk != 1 || p.atomic || p.a_ms % 8 || p.b_ns % 8 || p.N % 8 || p.c_ms % 8 || p.bias_mode > 1 ||
            (p.R && (p.r_ns != 1 || p.r_ms % 8 || !al(p.R))) || tiles < 128)
            return 1;                                      // under half a tile per CU: the 128 x 128 form fills the chip better
        DttGemm q = p;
        q.ws = nullptr;
        hipLaunchKernelGGL(gemm256_kernel<false>, dim3((unsigned)tiles), dim3(512), sm, st, q);
    } else if (rc) {
        // split-K weight gradients, row-contiguous operands
        if (!rc_covered(p)) return 1;
        const int sk = rc_splits(p, tiles);
        DttGemm q = p;
        const int64_t need = (int64_t)sk * p.M * p.N * 4;
        if (sk < 2 || !q.ws || q.ws_bytes < need || !al(q.ws) || p.c_ms % 4) q.ws = nullptr;
        hipLaunchKernelGGL(gemm256_kernel<true>, dim3((unsigned)tiles, sk), dim3(512), sm, st, q);
        if (q.ws) {
            const int64_t quads = (int64_t)p.M * (p.N / 4);
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<const float*>(q.ws), reinterpret_cast<float*>(p.C), p.M, p.N, p.c_ms, sk);
        }
    } else {
        return 1;
    }
    DTT_LAUNCH_CHECK();
    return 0;
}
