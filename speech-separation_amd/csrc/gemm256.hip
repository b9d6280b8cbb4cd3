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

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;            // 64 KB
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
    if (RC && gridDim.y % 8 == 0 && p.cphase == 0) {
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

    // ---- DMA sources.  KC: wave-instruction j fills rows (wave * 4 + j) * 8 .. + 8 of a 256-row image;
    //      RC: k-rows (wave * 4 + j) * 2, + 1 of a 64-k-row image
    const bf16* pa[4];
    const bf16* pb[4];
    int kb[4];                                             // RC + shift: k-row of the lane's piece inside the tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int L = (wave * 4 + j) * 64 + lane;
        if constexpr (!RC) {
            const int row = L >> 3, pos = L & 7, piece = pos ^ ((row >> 1) & 7);
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
    auto issue = [&](int stage, int kt) {
        unsigned char* sa = smem + stage * STAGE_BYTES + (wave * 4) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)pa[j], (lptr_t)(sa + j * 1024), 16, 0, 0);
            pa[j] += stepA;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16* src = pb[j];
            if constexpr (RC) {
                if (p.b_mode == 2) {
                    const int gk = kt * BK + kb[j];
                    if ((gk / p.sh_heads) % p.sh_L == p.sh_edge) src = reinterpret_cast<const bf16*>(&dtt_zero16);
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + BM * BK * 2 + j * 1024), 16, 0, 0);
            pb[j] += stepB;
        }
    };

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses.  KC: row * 128 bytes + ((2 s + lh) ^ swz(row)) * 16.
    // RC (transposing read, see tr_frag): k-row 16 s + 8 lh + q, columns base + 16 g + 4 p .. + 3
    int offA[4], offB[2], swA[4], swB[2];
    if constexpr (!RC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = wm * 128 + i * 32 + lr; offA[i] = row * 128; swA[i] = (row >> 1) & 7; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int row = wn * 64 + j * 32 + lr; offB[j] = BM * BK * 2 + row * 128; swB[j] = (row >> 1) & 7; }
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

    issue(0, kt0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    __syncthreads();
    if constexpr (RC) {
        const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
        for (int kt = kt0; kt < kt1; ++kt) {
            const unsigned sb = lds0 + ((kt - kt0) & 1) * STAGE_BYTES;
            if (kt + 1 < kt1) issue((kt - kt0 + 1) & 1, kt + 1);
            unsigned ad[6];
#pragma unroll
            for (int i = 0; i < 4; ++i) ad[i] = sb + offA[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) ad[4 + j] = sb + offB[j];
            RcFrag f0, f1;
#define RC_MMA(F) do { _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rc_join(F.lo[i], F.hi[i]), rc_join(F.lo[4 + j], F.hi[4 + j]), acc[i][j], 0, 0, 0); } while (0)
            rc_read<0>(f0, ad);
            rc_read<8192>(f1, ad);
            rc_fence<12>(f0);
            RC_MMA(f0);
            rc_read<16384>(f0, ad);
            rc_fence<12>(f1);
            RC_MMA(f1);
            rc_read<24576>(f1, ad);
            rc_fence<12>(f0);
            RC_MMA(f0);
            rc_fence<0>(f1);
            RC_MMA(f1);
#undef RC_MMA
            __builtin_amdgcn_s_waitcnt(0x0F70);            // this wave's DMA of the next stage has landed
            __syncthreads();                               // everyone's has, and everyone is done reading this stage
        }
    } else {
    for (int kt = kt0; kt < kt1; ++kt) {
        const unsigned char* st = smem + ((kt - kt0) & 1) * STAGE_BYTES;
        if (kt + 1 < kt1) issue((kt - kt0 + 1) & 1, kt + 1);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            s16x8_t af[4], bf_[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const s16x8_t*>(st + offA[i] + (((2 * s + lh) ^ swA[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) bf_[j] = *reinterpret_cast<const s16x8_t*>(st + offB[j] + (((2 * s + lh) ^ swB[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bf_[j]), acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                // this wave's DMA of the next stage has landed
        __syncthreads();                                   // everyone's has, and everyone is done reading this stage
    }
    }

    if (p.atomic) {
        if (p.cphase & 2) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.C)[0] = 1.f; return; }
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
    float* slab = reinterpret_cast<float*>(smem) + wave * 32 * SLAB_W;
    const bf16* R = reinterpret_cast<const bf16*>(p.R);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + lr;
            const float bv = (p.bias_mode == 1 && gn < p.N) ? p.bias[gn] : 0.f;
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
            if (p.out_f32) {
                float* dst = reinterpret_cast<float*>(p.C) + (int64_t)gm * p.c_ms + gn;
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                Vec16<bf16> ov;
#pragma unroll
                for (int e = 0; e < 8; e += 2) ov.set2(e, v[e], v[e + 1]);
                st16<bf16>(reinterpret_cast<bf16*>(p.C) + (int64_t)gm * p.c_ms + gn, ov);
            }
        }
    }
}

bool gemm256_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("DTT_GEMM256"); on = e ? atoi(e) : 1; }
    return on != 0;
}

}  // namespace

// 1: not covered (the caller goes on to the general kernels); 0 / negative as dtt_gemm
int dtt_gemm256_try(const DttGemm& p, hipStream_t st) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!gemm256_enabled() || p.dtype != 1 || p.a_mode || p.batch != 1 || p.act || p.c_ns != 1 || p.K % BK ||
        !al(p.A) || !al(p.B) || !al(p.C) || p.M < 192 || p.N < 192)
        return 1;
    const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int sm = 2 * STAGE_BYTES;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)gemm256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        hipFuncSetAttribute((const void*)gemm256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        attr = true;
    }
    const bool kc = p.a_ks == 1 && p.b_ks == 1 && p.b_mode == 0;
    const bool rc = p.a_ms == 1 && p.b_ns == 1 && (p.b_mode == 0 || p.b_mode == 2);
    if (kc) {
        // plain store, k-contiguous operands
        if (p.splitk != 1 || p.atomic || p.a_ms % 8 || p.b_ns % 8 || p.N % 8 || p.c_ms % 8 || p.bias_mode > 1 ||
            (p.R && (p.r_ns != 1 || p.r_ms % 8 || !al(p.R))) || tiles < 192)
            return 1;                                      // fewer tiles than CUs: the 128 x 128 form fills the chip better
        hipLaunchKernelGGL(gemm256_kernel<false>, dim3((unsigned)tiles), dim3(512), sm, st, p);
    } else if (rc) {
        // split-K weight gradients, row-contiguous operands
        if (!p.atomic || !p.out_f32 || p.bias_mode || p.R || p.a_ks % 8 || p.b_ks % 8 || p.M % 8 || p.N % 8 ||
            p.K / BK < 64 || p.N < (getenv("DTT_NMIN") ? atoi(getenv("DTT_NMIN")) : 512))             // (one column of tiles: the split-K atomics outweigh the mainloop)
            return 1;
        int sk = (int)std::max<int64_t>(1, std::min<int64_t>(p.K / BK / 8, (256 + tiles - 1) / tiles));
        { DttGemm q = p; if (getenv("DTT_MAP0")) q.cphase = 1; if (getenv("DTT_NOEPI")) q.cphase |= 2; if (getenv("DTT_SK")) sk = atoi(getenv("DTT_SK"));
        hipLaunchKernelGGL(gemm256_kernel<true>, dim3((unsigned)tiles, sk), dim3(512), sm, st, q); }
    } else {
        return 1;
    }
    DTT_LAUNCH_CHECK();
    return 0;
}
