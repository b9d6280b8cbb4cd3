// Persistent recurrent LSTM kernels (forward and backward-through-time) for the
// Improved Dual-Path modules (reference: src/model/DTTNet/modules/latent.py:87-92,
// nn.LSTM(C', 2C', bidirectional)).  One workgroup owns a block of NB sequences of
// one direction and walks all L steps itself: h lives in LDS, c in registers, the
// recurrent weights stream from L2 in MFMA-fragment order (packed by the host), the
// input projection x*W_ih^T (+biases) is precomputed by dtt_gemm.
//
// Row convention of every [rows][...] buffer: row = (outer*L + step)*heads + head;
// sequence n = outer*heads + head.
#include <algorithm>
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f / (1.0f + __expf(2.0f * x)); }

struct LstmP {
    const void* xg;      // [rows][2][4H]  pre-activations from the input projection (T)
    const void* wp;      // [2][4H/32][H/16][64][8] packed W_hh (T)           (fwd)
    void* y;             // [rows][2H] hidden states (T)
    void* gates;         // [rows][2][4H] post-activation i,f,g,o (T)
    float* cst;          // [rows][2][H] cell states (f32)
    int N, L, heads, H;
    int nb;              // sequences per workgroup: 32, 16 or 8 (the MFMA row tile stays 32; see lstm_block_rows)
};

// NW = min(8, H/32) waves; the hidden columns are processed in NP = H/(32*NW) sequential passes,
// wave w owning columns j = (q*NW + w)*32 .. +31 in pass q, all four gates of a column in one lane.
template <typename T, int PF, int NBT>
__global__ __launch_bounds__(512) void lstm_fwd_kernel(LstmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, LDH = H + MM<T>::PAD;
    constexpr int NB = 32;
    T* hbuf = reinterpret_cast<T*>(smem);                 // [2][NB][LDH]
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int dir = blockIdx.y, n0 = blockIdx.x * NBT;
    const int KS = H / 16, G4 = 4 * H;
    constexpr int RMAX = NBT / 2;                       // accumulator rows r < RMAX hold the block's NBT sequences
    const T* xg = reinterpret_cast<const T*>(p.xg);
    const T* wp = reinterpret_cast<const T*>(p.wp) + (int64_t)dir * G4 * H;
    T* y = reinterpret_cast<T*>(p.y);
    T* gates = reinterpret_cast<T*>(p.gates);

    // per-lane sequence rows of the C layout: r -> local row (r&3)+8(r>>2)+4*lh.
    // rb = row of step 0 of that sequence (-1: beyond N); all offsets fit 32 bits (checked on the host)
    int rb[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + nl;
        rb[r] = (n < p.N) ? (n / p.heads) * p.L * p.heads + (n % p.heads) : -1;
    }
    const int NP = H / (32 * nw);
    for (int s = 0; s < p.L; ++s) {
        const int t = dir ? p.L - 1 - s : s;
        const int tprev = dir ? t + 1 : t - 1;
        const int cur = s & 1;
        const T* hb = hbuf + cur * NB * LDH;
        T* hn = hbuf + (cur ^ 1) * NB * LDH;
        for (int q = 0; q < NP; ++q) {
            const int jt = q * nw + wave;                 // 32-column tile of this pass
            f32x16_t acc[4];
            // ---- init accumulators with the input projection; previous cell state comes back from
            // the global cst buffer (written by this same lane one step ago, L2-hot)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g][r] = 0.f;
                if (r < RMAX) {                           // rows of the MFMA tile beyond the block cost nothing
                    const bool ok = rb[r] >= 0;
                    const uint32_t row = (uint32_t)(ok ? rb[r] + t * p.heads : 0);
                    const T* src = xg + (row * 2u + dir) * (uint32_t)G4 + jt * 32 + lr;
#pragma unroll
                    for (int g = 0; g < 4; ++g) { const float v = ldf(src + g * H); acc[g][r] = ok ? v : 0.f; }
                }
            }
            // ---- recurrent term h_{t-1} * W_hh^T; W fragments are prefetched one k-step ahead
            if (s > 0) {
                typename MM<T>::Frag bq[PF + 1][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bq[0][g] = MM<T>::ldfrag(wp + (((int64_t)(g * (H / 32) + jt) * KS) * 64 + lane) * 8);
                for (int ks = 0; ks < KS; ks += 2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k = ks + u;
                        if (PF) {
                            if (k + 1 < KS) {
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    bq[(u ^ 1) & PF][g] =
                                        MM<T>::ldfrag(wp + (((int64_t)(g * (H / 32) + jt) * KS + k + 1) * 64 + lane) * 8);
                            }
                        } else if (k > 0) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                bq[0][g] = MM<T>::ldfrag(wp + (((int64_t)(g * (H / 32) + jt) * KS + k) * 64 + lane) * 8);
                        }
                        typename MM<T>::Frag af = MM<T>::ldfrag(hb + lr * LDH + k * 16 + lh * 8);
#pragma unroll
                        for (int g = 0; g < 4; ++g) MM<T>::mma(af, bq[u & PF][g], acc[g]);
                    }
                }
            }
            // ---- gates, state update, stores
            const int j = jt * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (r >= RMAX) { stf(hn + nl * LDH + j, 0.f); continue; }
                const bool ok = rb[r] >= 0;
                const uint32_t row = (uint32_t)(ok ? rb[r] + t * p.heads : 0);
                float ig = sigmoid_f(acc[0][r]), fg = sigmoid_f(acc[1][r]);
                float gg = tanh_f(acc[2][r]), og = sigmoid_f(acc[3][r]);
                const uint32_t rowp = (uint32_t)(ok && s > 0 ? rb[r] + tprev * p.heads : 0);
                const float cpl = p.cst[(rowp * 2u + dir) * (uint32_t)H + j];     // unconditional load, then select
                const float cprev = (ok && s > 0) ? cpl : 0.f;
                float c = fg * cprev + ig * gg;
                float h = og * tanh_f(c);
                stf(hn + nl * LDH + j, h);
                if (ok) {
                    stf(y + (row * 2u + dir) * (uint32_t)H + j, h);
                    p.cst[(row * 2u + dir) * (uint32_t)H + j] = c;
                    if (gates) {
                        T* gd = gates + (row * 2u + dir) * (uint32_t)G4 + j;
                        stf(gd, ig); stf(gd + H, fg); stf(gd + 2 * H, gg); stf(gd + 3 * H, og);
                    }
                }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------
// bf16, H = 256 forward with the input projection fused in: per step
//     G = [x_t | h_{t-1}] * [W_ih | W_hh]^T + b            (K = C' + H)
// so the [rows][2][4H] pre-activation tensor (545 MB at the bench size) is never written or read and
// its GEMM disappears.  Both weight matrices are streamed from L2 every step (768 KB per workgroup),
// register-prefetched several k-steps ahead; x_t is fetched one step ahead by LDS-DMA in MFMA
// A-fragment order (one 1 KB instruction per k-step, per-lane source rows), shared by the 8 waves.
// Saved activations use a private blocked layout that both recurrent kernels address as 16-byte
// vectors in the MFMA accumulator order (lane = hidden column j, 16 sequences of the block):
//     gates_blk [2 dir][blocks][L][4][H][2][16] bf16      cst_blk [2 dir][blocks][L][H][2][16] fp32
// y stays row-major ([rows][2H], consumed by GEMMs) and is copied out of the h buffer in LDS.
struct LstmFP {
    const void* xn; const void* wihp; const void* whhp; const float* bias;
    void* y; void* gates; float* cst;
    int N, L, heads, Cp, nblk;
};

// TM 32-sequence blocks per workgroup: every streamed weight fragment feeds TM MFMAs (the step is
// bound by the weight stream, 768 KB per step per CU, so TM = 2 nearly halves the time per sequence).
// Weight stream: measured, an LDS-DMA ring made the k-loop LDS-bandwidth bound (every weight byte is
// written to and read back from LDS: 1.7 MB of LDS traffic per step per CU = 5 us).  The fragments
// are therefore prefetched straight into registers, D k-steps (4 fragments each) ahead; LDS carries
// only the A operand.  KT % D == 0 keeps the register slots compile-time across time steps.
template <int KI, int D>
__global__ __launch_bounds__(512) void lstm_fwd_fused_kernel(LstmFP p) {
    constexpr int H = 256, NB = 32, LDH = H + 8, KH = 16, KT = KI + KH, G4 = 1024;
    static_assert(KT % D == 0, "prefetch slots must be static across time steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* hbuf = reinterpret_cast<bf16*>(smem);                       // [NB][LDH]
    bf16* xabuf = hbuf + NB * LDH;                                    // [2][KI] fragments of 512 elements
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int jts = __builtin_amdgcn_readfirstlane(wave);
    const int dir = blockIdx.y, blk = blockIdx.x, n0 = blk * NB;
    const int Cp = KI * 16;
    const char* wih = reinterpret_cast<const char*>(p.wihp) + ((size_t)dir * G4 * Cp + (size_t)jts * KI * 512) * 2;
    const char* whh = reinterpret_cast<const char*>(p.whhp) + ((size_t)dir * G4 * H + (size_t)jts * KH * 512) * 2;
    const bf16* xn = reinterpret_cast<const bf16*>(p.xn);
    const int j = wave * 32 + lr;
    uint32_t lane16 = lane * 16;
    // fragment (kk, g) of the concatenated K (W_ih k-steps first, then W_hh): uniform base + lane*16
    auto wfrag = [&](int kk, int g) -> s16x8_t {
        asm volatile("" : "+v"(lane16));                   // keep the 96 addresses out of VGPRs
        const char* sb = kk < KI ? wih + (size_t)(g * (H / 32) * KI + kk) * 1024
                                 : whh + (size_t)(g * (H / 32) * KH + (kk - KI)) * 1024;
        return *reinterpret_cast<const s16x8_t*>(sb + lane16);
    };
    // x_t in A-fragment order by LDS-DMA: wave w fetches k-step w (lane = row lr, k half lh), one step ahead
    const int nA = n0 + lr;
    const bool okA = nA < p.N;
    const int rbA = okA ? (nA / p.heads) * p.L * p.heads + (nA % p.heads) : 0;
    auto issue_x = [&](int t, int buf) {
        if (jts < KI) {
            const bf16* src = okA ? xn + (size_t)(rbA + t * p.heads) * Cp + jts * 16 + lh * 8
                                  : reinterpret_cast<const bf16*>(&dtt_zero16);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xabuf + (buf * KI + jts) * 512), 16, 0, 0);
        }
    };
    int rbY[2];                                            // rows of the y copy-out: local row (tid>>5) + 16u
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int n = n0 + (tid >> 5) + 16 * u;
        rbY[u] = n < p.N ? (n / p.heads) * p.L * p.heads + (n % p.heads) : -1;
    }
    for (int i = tid; i < NB * LDH / 8; i += 512) reinterpret_cast<uint4*>(hbuf)[i] = make_uint4(0, 0, 0, 0);   // h_{-1} = 0
    issue_x(dir ? p.L - 1 : 0, 0);
    s16x8_t bq[D][4];
#pragma unroll
    for (int kk = 0; kk < D; ++kk)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[kk][g] = wfrag(kk, g);
    float bsv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bsv[g] = p.bias[dir * G4 + g * H + j];
    float cst[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cst[r] = 0.f;
    __syncthreads();                                       // (drains the x fetch: LDS-DMA counts on vmcnt)
    const bool save = p.gates != nullptr;                  // gates and cst are saved together

    for (int s = 0; s < p.L; ++s) {
        const int t = dir ? p.L - 1 - s : s;
        const bool more = s + 1 < p.L;
        if (more) issue_x(dir ? t - 1 : t + 1, (s + 1) & 1);          // x of the next step
        const bf16* xa = xabuf + (s & 1) * KI * 512;
        f32x16_t acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = bsv[g];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const s16x8_t af = kk < KI ? *reinterpret_cast<const s16x8_t*>(xa + kk * 512 + lane * 8)
                                       : MM<bf16>::ldfrag(hbuf + lr * LDH + (kk - KI) * 16 + lh * 8);
#pragma unroll
            for (int g = 0; g < 4; ++g) MM<bf16>::mma(af, bq[kk % D][g], acc[g]);
            if (kk + D < KT || more) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[kk % D][g] = wfrag((kk + D) % KT, g);
            }
        }
        __syncthreads();                                  // every wave has read h_{t-1}
        // ---- gates, state update; 16-byte stores in accumulator order
        const size_t slab = ((size_t)dir * p.nblk + blk) * p.L + t;
        bf16* gd = reinterpret_cast<bf16*>(p.gates) + slab * (4 * H * 32);
        float* cd = p.cst + slab * (H * 32) + (j * 2 + lh) * 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            Vec16<bf16> gv[4];
            float4 cv[2];
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                float go[4][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int rr = half * 8 + q + e;
                    const float ig = sigm_fast(acc[0][rr]), fg = sigm_fast(acc[1][rr]);
                    const float gg = tanh_fast(acc[2][rr]), og = sigm_fast(acc[3][rr]);
                    const float c = fg * cst[rr] + ig * gg;
                    cst[rr] = c;
                    (&cv[(q + e) >> 2].x)[(q + e) & 3] = c;
                    const int nl = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                    stf(hbuf + nl * LDH + j, og * tanh_fast(c));
                    go[0][e] = ig; go[1][e] = fg; go[2][e] = gg; go[3][e] = og;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) gv[g].set2(q, go[g][0], go[g][1]);
            }
            if (save) {
#pragma unroll
                for (int g = 0; g < 4; ++g) st16<bf16>(gd + ((g * H + j) * 2 + lh) * 16 + half * 8, gv[g]);
                *reinterpret_cast<float4*>(cd + half * 8) = cv[0];
                *reinterpret_cast<float4*>(cd + half * 8 + 4) = cv[1];
            }
        }
        __syncthreads();                                  // h_t complete (and x_{t+1} landed: the barrier drains vmcnt)
        // ---- y_t rows straight from the h buffer: 16-byte chunks, 32 per row
        bf16* y = reinterpret_cast<bf16*>(p.y);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nl = (tid >> 5) + 16 * u, ch = tid & 31;
            const uint4 v = *reinterpret_cast<const uint4*>(hbuf + nl * LDH + ch * 8);
            if (rbY[u] >= 0) *reinterpret_cast<uint4*>(y + ((size_t)(rbY[u] + t * p.heads) * 2 + dir) * H + ch * 8) = v;
        }
    }
}

struct LstmBwdP {
    const void* dy;      // [rows][2H] gradient wrt hidden outputs (T)
    const void* gates;   // [rows][2][4H] saved activations (T)
    const float* cst;    // [rows][2][H]
    const void* wpt;     // [2][H/32][4H/16][64][8] packed W_hh for dh = dG * W_hh (T)
    void* dG;            // [rows][2][4H] gate pre-activation gradients (T)
    int N, L, heads, H, kch;     // kch: gates staged in LDS per pass (4, 2 or 1)
    int nb;                      // sequences per workgroup (32, 16 or 8)
};

// STAGE_ALL: all four gate gradients of a step fit LDS at once and are written there straight from
// registers; otherwise they are staged chunk by chunk from the (L2-hot) global dG this workgroup
// has just written (workgroup-scope release/acquire around the barrier).
template <typename T, int NP, bool STAGE_ALL>
__global__ __launch_bounds__(512) void lstm_bwd_kernel(LstmBwdP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, kch = p.kch, LDG = kch * H + MM<T>::PAD, LDD = H + 4;
    constexpr int NB = 32, EPT = MM<T>::EPT, NCH = NP * 16 / EPT;
    T* dgbuf = reinterpret_cast<T*>(smem);                // [NB][LDG]
    // dh_{t-1} in fp32, [NB][LDD]: overlays dgbuf between the end of a step's GEMM and the next gate math
    float* dhbuf = reinterpret_cast<float*>(smem);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int dir = blockIdx.y, n0 = blockIdx.x * p.nb;
    const int G4 = 4 * H, KS = G4 / 16;
    const T* dy = reinterpret_cast<const T*>(p.dy);
    const T* gates = reinterpret_cast<const T*>(p.gates);
    const T* wpt = reinterpret_cast<const T*>(p.wpt) + (int64_t)dir * G4 * H;
    T* dG = reinterpret_cast<T*>(p.dG);

    // The gate math runs on row-major 16-byte chunks: the 32 x H/EPT chunks of the block's rows are dealt
    // to the threads NCH apiece (chunk ci = tid + i * blockDim: row ci / cpr, columns EPT * (ci % cpr)), so
    // gates, cell states and dy arrive as 16-byte vectors and dG leaves as 16-byte vectors (the first form
    // of this kernel did the math in the MFMA accumulator layout with 2-byte accesses: 120 us per step at
    // H = 512).  dc stays in registers in this layout; dh comes back from the GEMM through LDS.
    const int cpr = H / EPT;
    int rbc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int nl = (threadIdx.x + i * blockDim.x) / cpr, n = n0 + nl;
        rbc[i] = (nl < p.nb && n < p.N) ? (n / p.heads) * p.L * p.heads + (n % p.heads) : -1;
    }
    float dcs[NCH][EPT];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < EPT; ++e) dcs[i][e] = 0.f;

    for (int s = p.L - 1; s >= 0; --s) {
        const int t = dir ? p.L - 1 - s : s;
        const int tprev = dir ? t + 1 : t - 1;          // forward-pass predecessor step
        float dhv[NCH][EPT];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = threadIdx.x + i * blockDim.x, nl = ci / cpr, cc = ci - nl * cpr;
#pragma unroll
            for (int e4 = 0; e4 < EPT; e4 += 4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s < p.L - 1) v = *reinterpret_cast<const float4*>(dhbuf + nl * LDD + cc * EPT + e4);
                dhv[i][e4] = v.x; dhv[i][e4 + 1] = v.y; dhv[i][e4 + 2] = v.z; dhv[i][e4 + 3] = v.w;
            }
        }
        __syncthreads();                                  // every thread holds its dh: the region becomes dgbuf
#pragma unroll (NP >= 4 ? 2 : NCH)
        for (int i = 0; i < NCH; ++i) {
            const int ci = threadIdx.x + i * blockDim.x, nl = ci / cpr, cc = ci - nl * cpr;
            const bool ok = rbc[i] >= 0;
            if (nl >= p.nb) {                                 // rows of the MFMA tile beyond the block: zeros, no traffic
                if (STAGE_ALL) {
                    Vec16<T> z;
#pragma unroll
                    for (int e = 0; e < EPT; ++e) z.set(e, 0.f);
                    T* ld = dgbuf + nl * LDG + cc * EPT;
                    st16<T>(ld, z); st16<T>(ld + H, z); st16<T>(ld + 2 * H, z); st16<T>(ld + 3 * H, z);
                }
                continue;
            }
            // loads are unconditional (row 0 stands in for out-of-range sequences)
            const uint32_t row = (uint32_t)(ok ? rbc[i] + t * p.heads : 0);
            const uint32_t rowp = (uint32_t)(ok && s > 0 ? rbc[i] + tprev * p.heads : 0);
            const T* gs = gates + (row * 2u + dir) * (uint32_t)G4 + cc * EPT;
            const Vec16<T> vi = ld16<T>(gs), vf = ld16<T>(gs + H), vg = ld16<T>(gs + 2 * H), vo = ld16<T>(gs + 3 * H);
            const Vec16<T> vdy = ld16<T>(dy + (row * 2u + dir) * (uint32_t)H + cc * EPT);
            const float* cs = p.cst + (row * 2u + dir) * (uint32_t)H + cc * EPT;
            const float* cps = p.cst + (rowp * 2u + dir) * (uint32_t)H + cc * EPT;
            float cv[EPT], cpv[EPT];
#pragma unroll
            for (int e4 = 0; e4 < EPT; e4 += 4) {
                const float4 a = *reinterpret_cast<const float4*>(cs + e4), b = *reinterpret_cast<const float4*>(cps + e4);
                cv[e4] = a.x; cv[e4 + 1] = a.y; cv[e4 + 2] = a.z; cv[e4 + 3] = a.w;
                cpv[e4] = b.x; cpv[e4 + 1] = b.y; cpv[e4 + 2] = b.z; cpv[e4 + 3] = b.w;
            }
            Vec16<T> oi, of, og_, oo;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float ig = vi.get(e), fg = vf.get(e), gg = vg.get(e), og = vo.get(e);
                const float cp = (s > 0) ? cpv[e] : 0.f;
                const float dh = vdy.get(e) + dhv[i][e];
                const float tc = tanh_f(cv[e]);
                float dog_ = dh * tc * og * (1.f - og);
                const float dc = dcs[i][e] + dh * og * (1.f - tc * tc);
                float di = dc * gg * ig * (1.f - ig);
                float df = dc * cp * fg * (1.f - fg);
                float dg = dc * ig * (1.f - gg * gg);
                dcs[i][e] = dc * fg;
                if (!ok) { di = 0.f; df = 0.f; dg = 0.f; dog_ = 0.f; }
                oi.set(e, di); of.set(e, df); og_.set(e, dg); oo.set(e, dog_);
            }
            if (ok) {
                T* gd = dG + (row * 2u + dir) * (uint32_t)G4 + cc * EPT;
                st16<T>(gd, oi); st16<T>(gd + H, of); st16<T>(gd + 2 * H, og_); st16<T>(gd + 3 * H, oo);
            }
            if (STAGE_ALL) {
                T* ld = dgbuf + nl * LDG + cc * EPT;
                st16<T>(ld, oi); st16<T>(ld + H, of); st16<T>(ld + 2 * H, og_); st16<T>(ld + 3 * H, oo);
            }
        }
        if (s == 0) break;
        // ---- dh_{t-1} = dG_t * W_hh, K = 4H
        f32x16_t dhr[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) dhr[q][r] = 0.f;
        if (!STAGE_ALL) __threadfence_block();
        for (int g0 = 0; g0 < 4; g0 += kch) {
            if (!STAGE_ALL) {
                if (g0) __syncthreads();
                else __syncthreads();
                // stage dG[rows of this block][g0*H .. (g0+kch)*H) from global
                const int cpr = kch * H / EPT;
                for (int idx = threadIdx.x; idx < NB * cpr; idx += blockDim.x) {
                    const int nl = idx / cpr, ch = idx - nl * cpr;
                    const int n = n0 + nl;
                    const bool okr = nl < p.nb && n < p.N;
                    const int nn = okr ? n : 0;
                    const uint32_t row = (uint32_t)((nn / p.heads) * p.L * p.heads + (nn % p.heads) + t * p.heads);
                    Vec16<T> v = ld16_pred<T>(dG + (row * 2u + dir) * (uint32_t)G4 + g0 * H + ch * EPT, dG, okr);
                    st16<T>(dgbuf + nl * LDG + ch * EPT, v);
                }
            }
            __syncthreads();
            const int ks0 = g0 * H / 16, ks1 = (g0 + kch) * H / 16;       // even count: H % 32 == 0
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int jt = q * nw + wave;
                // W_hh fragments stream from L2: a ring of BD k-steps in flight (one k-step ahead, as this
                // loop first had it, left every MFMA waiting for an L2 round trip: 128 of them per pass at H = 512)
                constexpr int BD = 8;
                typename MM<T>::Frag bq[BD];
                const T* wq = wpt + ((int64_t)jt * KS * 64 + lane) * 8;
#pragma unroll
                for (int u = 0; u < BD; ++u)
                    if (ks0 + u < ks1) bq[u] = MM<T>::ldfrag(wq + (int64_t)(ks0 + u) * 512);
                for (int ks = ks0; ks < ks1; ks += BD) {
#pragma unroll
                    for (int u = 0; u < BD; ++u) {
                        const int k = ks + u;
                        if (k < ks1) {
                            typename MM<T>::Frag af = MM<T>::ldfrag(dgbuf + lr * LDG + (k - ks0) * 16 + lh * 8);
                            MM<T>::mma(af, bq[u], dhr[q]);
                            if (k + BD < ks1) bq[u] = MM<T>::ldfrag(wq + (int64_t)(k + BD) * 512);
                        }
                    }
                }
            }
            __syncthreads();
        }
        // every wave is done with dgbuf: dh_{t-1} goes to LDS in the accumulator layout, to be picked up
        // as row-major chunks at the top of the next step
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int j = (q * nw + wave) * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) dhbuf[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDD + j] = dhr[q][r];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// bf16, H = 256 backward through time over the blocked activation layout of lstm_fwd_fused_kernel.
// Per step: the saved gates / cell states arrive as 16-byte vectors (prefetched one step ahead), dy
// rows are staged through LDS by cooperative 16-byte loads, the gate gradients go to LDS once (the
// A operand of dh_{t-1} = dG_t * W_hh and the source of the row-major dG rows, written back as
// 16-byte stores), and the W_hh fragments (512 KB per step) are register-prefetched D k-steps ahead.
struct LstmBP {
    const void* dy; const void* gates; const float* cst; const void* wpt; void* dG; float* dbias;
    int N, L, heads, nblk;
};

template <int D>
__global__ __launch_bounds__(512) void lstm_bwd_blk_kernel(LstmBP p) {
    constexpr int H = 256, NB = 32, G4 = 1024, LDG = G4 + 8, LDY = H + 8, KS = 64;
    static_assert(KS % D == 0, "prefetch slots must be static across time steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* dgbuf = reinterpret_cast<bf16*>(smem);                      // [NB][LDG]
    bf16* dybuf = dgbuf + NB * LDG;                                   // [NB][LDY]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int jts = __builtin_amdgcn_readfirstlane(wave);
    const int dir = blockIdx.y, blk = blockIdx.x, n0 = blk * NB;
    const int j = wave * 32 + lr;
    const bf16* dy = reinterpret_cast<const bf16*>(p.dy);
    const bf16* gates = reinterpret_cast<const bf16*>(p.gates);
    bf16* dG = reinterpret_cast<bf16*>(p.dG);
    const char* wpt = reinterpret_cast<const char*>(p.wpt) + ((size_t)dir * G4 * H + (size_t)jts * KS * 512) * 2;
    uint32_t lane16 = lane * 16;
    auto wfrag = [&](int k) -> s16x8_t {
        asm volatile("" : "+v"(lane16));
        return *reinterpret_cast<const s16x8_t*>(wpt + (size_t)k * 1024 + lane16);
    };
    // cooperative row roles: dy rows (32 chunks of 16 bytes per row) and dG rows (128 chunks per row)
    int rbY[2], rbG[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int n = n0 + (tid >> 5) + 16 * u;
        rbY[u] = n < p.N ? (n / p.heads) * p.L * p.heads + (n % p.heads) : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int n = n0 + (tid >> 7) + 4 * u;
        rbG[u] = n < p.N ? (n / p.heads) * p.L * p.heads + (n % p.heads) : -1;
    }
    auto slab_of = [&](int t) { return ((size_t)dir * p.nblk + blk) * p.L + t; };
    auto load_dy = [&](int t, uint4 (&v)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16* src = rbY[u] >= 0 ? dy + ((size_t)(rbY[u] + t * p.heads) * 2 + dir) * H + (tid & 31) * 8
                                          : reinterpret_cast<const bf16*>(&dtt_zero16);
            v[u] = *reinterpret_cast<const uint4*>(src);
        }
    };
    auto load_gates = [&](int t, Vec16<bf16> (&gv)[4][2]) {
        const bf16* gd = gates + slab_of(t) * (4 * H * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            gv[g][0] = ld16<bf16>(gd + ((g * H + j) * 2 + lh) * 16);
            gv[g][1] = ld16<bf16>(gd + ((g * H + j) * 2 + lh) * 16 + 8);
        }
    };
    auto load_c = [&](int t, float4 (&cv)[4]) {
        const float* cd = p.cst + slab_of(t) * (H * 32) + (j * 2 + lh) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) cv[q] = *reinterpret_cast<const float4*>(cd + q * 4);
    };

    float dcs[16];
    f32x16_t dhr;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dcs[r] = 0.f; dhr[r] = 0.f; }
    float dbs0 = 0.f, dbs1 = 0.f;                          // bias gradient of gate columns 2*tid, 2*tid+1
    const int tl = dir ? 0 : p.L - 1;                      // step processed first
    uint4 dyv[2];
    Vec16<bf16> gv[4][2];
    float4 cv[4], cpv[4], cnx[4];                          // c_t, c_{t-1} (forward order), prefetch
    load_dy(tl, dyv);
    load_gates(tl, gv);
    load_c(tl, cv);
    if (p.L > 1) load_c(dir ? 1 : p.L - 2, cpv);
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) cpv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    s16x8_t bq[D];
#pragma unroll
    for (int k = 0; k < D; ++k) bq[k] = wfrag(k);

    for (int s = p.L - 1; s >= 0; --s) {
        const int t = dir ? p.L - 1 - s : s;
        const int tprev = dir ? t + 1 : t - 1;          // forward-pass predecessor step
        // dy rows of this step -> LDS
#pragma unroll
        for (int u = 0; u < 2; ++u)
            *reinterpret_cast<uint4*>(dybuf + ((tid >> 5) + 16 * u) * LDY + (tid & 31) * 8) = dyv[u];
        __syncthreads();                                  // dybuf ready; dgbuf free (GEMM of the previous step done)
        if (s > 0) load_dy(tprev, dyv);                   // next step's rows, landed by the time they are staged
        if (s > 1) load_c(dir ? t + 2 : t - 2, cnx);      // cell state two steps back: next step's c_{t-1}
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float ig = gv[0][r >> 3].get(r & 7), fg = gv[1][r >> 3].get(r & 7);
            const float gg = gv[2][r >> 3].get(r & 7), og = gv[3][r >> 3].get(r & 7);
            const float c = (&cv[r >> 2].x)[r & 3], cp = (&cpv[r >> 2].x)[r & 3];
            const float dh = ldf(dybuf + nl * LDY + j) + dhr[r];
            const float tc = tanh_fast(c);
            const float dog_ = dh * tc * og * (1.f - og);
            const float dc = dcs[r] + dh * og * (1.f - tc * tc);
            const float di = dc * gg * ig * (1.f - ig);
            const float df = dc * cp * fg * (1.f - fg);
            const float dg = dc * ig * (1.f - gg * gg);
            dcs[r] = dc * fg;
            bf16* ld = dgbuf + nl * LDG + j;
            stf(ld, di); stf(ld + H, df); stf(ld + 2 * H, dg); stf(ld + 3 * H, dog_);
        }
        if (s > 0) {
            load_gates(tprev, gv);                        // prefetch the next step's activations
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cv[q] = cpv[q];
                cpv[q] = s > 1 ? cnx[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();                                  // dG_t complete in LDS
        if (p.dbias) {                                    // column sums of dG_t (padding sequences hold exact zeros)
#pragma unroll 8
            for (int nl = 0; nl < NB; ++nl) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(dgbuf + nl * LDG + 2 * tid);
                dbs0 += __uint_as_float(w << 16);
                dbs1 += __uint_as_float(w & 0xffff0000u);
            }
        }
        // ---- dh_{t-1} = dG_t * W_hh, K = 4H
        if (s > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dhr[r] = 0.f;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const s16x8_t af = MM<bf16>::ldfrag(dgbuf + lr * LDG + k * 16 + lh * 8);
                MM<bf16>::mma(af, bq[k % D], dhr);
                if (k + D < KS || s > 1) bq[k % D] = wfrag((k + D) % KS);
            }
        }
        // ---- dG rows out, row-major [row][dir][4H]: 8 chunks of 16 bytes per thread.  After the GEMM:
        // vmcnt retires in order, so stores ahead of the weight loads would stall their waits.
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int nl = (tid >> 7) + 4 * u, ch = tid & 127;
            const uint4 v = *reinterpret_cast<const uint4*>(dgbuf + nl * LDG + ch * 8);
            if (rbG[u] >= 0) *reinterpret_cast<uint4*>(dG + ((size_t)(rbG[u] + t * p.heads) * 2 + dir) * G4 + ch * 8) = v;
        }
    }
    if (p.dbias) {
        atomicAdd(p.dbias + dir * G4 + 2 * tid, dbs0);
        atomicAdd(p.dbias + dir * G4 + 2 * tid + 1, dbs1);
    }
}

template <typename T, int PF>
int launch_fwd(const LstmP& p, int nw, hipStream_t st) {
    size_t sm = (size_t)2 * 32 * (p.H + MM<T>::PAD) * sizeof(T);
    DTT_CHECK(sm <= 160 * 1024, "dtt_lstm_fwd: H=%d does not fit LDS in this dtype", p.H);
    const dim3 grid((p.N + p.nb - 1) / p.nb, 2), block(nw * 64);
#define LAUNCH_NB(NBT_) do { auto k = lstm_fwd_kernel<T, PF, NBT_>; \
        if (sm > 64 * 1024) dtt_want_lds((const void*)k, (int)sm); \
        hipLaunchKernelGGL(k, grid, block, sm, st, p); } while (0)
    if (p.nb == 8) LAUNCH_NB(8); else if (p.nb == 16) LAUNCH_NB(16); else LAUNCH_NB(32);
#undef LAUNCH_NB
    DTT_LAUNCH_CHECK();
    return 0;
}

template <typename T, int NP>
int launch_bwd(LstmBwdP p, int nw, hipStream_t st) {
    int kch = 4;
    while (kch > 1 && (size_t)32 * (kch * p.H + MM<T>::PAD) * sizeof(T) > 150 * 1024) kch >>= 1;
    size_t sm = (size_t)32 * (kch * p.H + MM<T>::PAD) * sizeof(T);
    sm = std::max(sm, (size_t)32 * (p.H + 4) * sizeof(float));          // dh_{t-1} (fp32) overlays the staging buffer
    DTT_CHECK(sm <= 160 * 1024, "dtt_lstm_bwd: H=%d does not fit LDS", p.H);
    p.kch = kch;
    if (kch == 4) {
        auto k = lstm_bwd_kernel<T, NP, true>;
        if (sm > 64 * 1024) dtt_want_lds((const void*)k, (int)sm);
        hipLaunchKernelGGL(k, dim3((p.N + p.nb - 1) / p.nb, 2), dim3(nw * 64), sm, st, p);
    } else {
        auto k = lstm_bwd_kernel<T, NP, false>;
        if (sm > 64 * 1024) dtt_want_lds((const void*)k, (int)sm);
        hipLaunchKernelGGL(k, dim3((p.N + p.nb - 1) / p.nb, 2), dim3(nw * 64), sm, st, p);
    }
    DTT_LAUNCH_CHECK();
    return 0;
}

// Sequences per workgroup of the general kernels.  A workgroup streams all of W_hh every step whatever its
// row count, but its gate math and activation traffic scale with the rows: when the 32-row blocks leave CUs
// idle (cfg4: 512 sequences = 32 workgroups) narrower blocks spread that part over more CUs.
int lstm_block_rows(int N) {
    int nb = 32;
    while (nb > 8 && 2 * ((N + nb / 2 - 1) / (nb / 2)) <= 256) nb >>= 1;
    return nb;
}

int lstm_shape(int H, int& nw, int& np) {
    nw = H / 32 < 8 ? H / 32 : 8;
    np = H / (32 * nw);
    return (H % 32 == 0 && H >= 32 && H % (32 * nw) == 0 && (np == 1 || np == 2 || np == 4)) ? 0 : -1;
}

}  // namespace

extern "C" int dtt_lstm_fwd(const void* xg, const void* wp, void* y, void* gates, float* cst, int N, int L, int heads,
                            int H, int dtype, void* stream) {
    int nw, np;
    DTT_CHECK(lstm_shape(H, nw, np) == 0 && N > 0 && L > 0 && heads > 0 && N % heads == 0,
              "dtt_lstm_fwd: unsupported shape H=%d N=%d L=%d", H, N, L);
    DTT_CHECK((int64_t)N * L * 8 * H < (1LL << 31), "dtt_lstm_fwd: buffers exceed 32-bit element offsets");
    LstmP p{xg, wp, y, gates, cst, N, L, heads, H, 32};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DTT_CHECK(cst != nullptr, "dtt_lstm_fwd: cst scratch is required (cell state round-trips through it)");
    p.nb = lstm_block_rows(N);
    return dtype ? launch_fwd<bf16, 1>(p, nw, st) : launch_fwd<float, 1>(p, nw, st);
}

extern "C" int dtt_lstm_bwd(const void* dy, const void* gates, const float* cst, const void* wpt, void* dG, int N,
                            int L, int heads, int H, int dtype, void* stream) {
    int nw, np;
    DTT_CHECK(lstm_shape(H, nw, np) == 0 && N > 0 && L > 0 && heads > 0 && N % heads == 0,
              "dtt_lstm_bwd: unsupported shape H=%d N=%d L=%d", H, N, L);
    DTT_CHECK((int64_t)N * L * 8 * H < (1LL << 31), "dtt_lstm_bwd: buffers exceed 32-bit element offsets");
    LstmBwdP p{dy, gates, cst, wpt, dG, N, L, heads, H, 4, lstm_block_rows(N)};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype) {
        if (np == 1) return launch_bwd<bf16, 1>(p, nw, st);
        if (np == 2) return launch_bwd<bf16, 2>(p, nw, st);
        return launch_bwd<bf16, 4>(p, nw, st);
    }
    if (np == 1) return launch_bwd<float, 1>(p, nw, st);
    if (np == 2) return launch_bwd<float, 2>(p, nw, st);
    return launch_bwd<float, 4>(p, nw, st);
}

// Fused forward for bf16, H = 256, C' = 128 (see lstm_fwd_fused_kernel).  gates_blk / cst_blk may be
// NULL when no backward pass follows.  Returns 1 (no error set) when the shape is not covered: the
// caller then uses dtt_gemm + dtt_lstm_fwd.
extern "C" int dtt_lstm_fwd_fused(const void* xn, const void* wihp, const void* whhp, const float* bias, void* y,
                                  void* gates_blk, float* cst_blk, int N, int L, int heads, int H, int Cp, int dtype,
                                  void* stream) {
    if (!(dtype == 1 && H == 256 && Cp == 128)) return 1;
    DTT_CHECK(N > 0 && L > 0 && heads > 0 && N % heads == 0, "dtt_lstm_fwd_fused: bad shape N=%d L=%d", N, L);
    DTT_CHECK((int64_t)((N + 31) / 32) * 32 * L * 8 * H < (1LL << 31), "dtt_lstm_fwd_fused: buffers exceed 32-bit offsets");
    DTT_CHECK((gates_blk == nullptr) == (cst_blk == nullptr), "dtt_lstm_fwd_fused: gates_blk and cst_blk go together");
    const int nblk = (N + 31) / 32;
    LstmFP p{xn, wihp, whhp, bias, y, gates_blk, cst_blk, N, L, heads, Cp, nblk};
    constexpr int KI = 8;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t sm = (size_t)32 * (256 + 8) * 2 + 2 * KI * 1024;
    auto k = lstm_fwd_fused_kernel<KI, 4>;
    hipLaunchKernelGGL(k, dim3(nblk, 2), dim3(512), sm, st, p);
    DTT_LAUNCH_CHECK();
    return 0;
}

// Backward through time over the blocked activations written by dtt_lstm_fwd_fused (bf16, H = 256).
// dy [rows][2H], wpt [2][H/32][4H/16][64][8] (as dtt_lstm_bwd), dG [rows][2][4H] row-major;
// dbias [2][4H] fp32 (zeroed by the caller, may be NULL) receives the column sums of dG = d(b_ih) = d(b_hh).
extern "C" int dtt_lstm_bwd_blk(const void* dy, const void* gates_blk, const float* cst_blk, const void* wpt, void* dG,
                                float* dbias, int N, int L, int heads, int H, int dtype, void* stream) {
    DTT_CHECK(dtype == 1 && H == 256, "dtt_lstm_bwd_blk: bf16, H = 256 only (got dtype %d, H %d)", dtype, H);
    DTT_CHECK(N > 0 && L > 0 && heads > 0 && N % heads == 0, "dtt_lstm_bwd_blk: bad shape N=%d L=%d", N, L);
    DTT_CHECK(dy && gates_blk && cst_blk && wpt && dG, "dtt_lstm_bwd_blk: null buffer");
    const int nblk = (N + 31) / 32;
    LstmBP p{dy, gates_blk, cst_blk, wpt, dG, dbias, N, L, heads, nblk};
    const size_t sm = (size_t)32 * (1024 + 8) * 2 + 32 * (256 + 8) * 2;
    auto k = lstm_bwd_blk_kernel<8>;
    dtt_want_lds((const void*)k, (int)sm);
    hipLaunchKernelGGL(k, dim3(nblk, 2), dim3(512), sm, reinterpret_cast<hipStream_t>(stream), p);
    DTT_LAUNCH_CHECK();
    return 0;
}

