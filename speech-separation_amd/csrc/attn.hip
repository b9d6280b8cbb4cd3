// Cross-attention core of CrossAttentionPositional (reference: src/model/VideoAudioFusions/
// cross_attention_positional.py:66-69, nn.MultiheadAttention(batch_first) after its input projections):
//     O[b, i, h*dh + d] = sum_j softmax_j(scale * <Q[b,i,h,:], K[b,j,h,:]>) V[b, j, h*dh + d]
// for Tq = 16 audio tokens, Tk = 50 (150) video tokens, 2 heads.  fp32 in and out (the token network runs outside
// autocast), exact-fp32 MFMA tiles (v_mfma_f32_32x32x2_f32), one workgroup per (batch item, head).  The operands are
// a few tens of KB and L2-resident: they are read straight from global memory in MFMA operand order; only the
// score / probability tile lives in LDS.  Forward keeps the probabilities for backward.
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {
constexpr int NTH = 256, TQ_MAX = 32, TK_MAX = 160;

struct AttnP {
    const float* q; const float* k; const float* v; const float* dout;
    float* out; float* prob; float* dq; float* dk; float* dv;
    int B, Tq, Tk, C, heads;
    float scale;
};

// T[i][j] = sum_d X[i][d] * Y[j][d] over the head's dh columns: X rows i < nx (one 32-row tile), Y rows j < ny.
// Result (times mul) into the LDS tile t[32][ldt]; rows / columns beyond nx / ny are written as `fill`.
__device__ __forceinline__ void rows_dot_rows(const float* __restrict__ X, int nx, const float* __restrict__ Y, int ny,
                                              int C, int dh, float mul, float fill, float* t, int ldt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int nct = (ny + 31) / 32;
    for (int ct = wave; ct < nct; ct += NTH / 64) {
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int j = ct * 32 + lr;
        const float* xr = X + (size_t)min(lr, nx - 1) * C + 4 * lh;
        const float* yr = Y + (size_t)min(j, ny - 1) * C + 4 * lh;
        for (int d = 0; d < dh; d += 8) {              // lane half lh contracts k = d + 4 lh .. + 3 (same map on both sides)
            const float4 a = *reinterpret_cast<const float4*>(xr + d);
            const float4 b = *reinterpret_cast<const float4*>(yr + d);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            t[i * ldt + j] = (i < nx && j < ny) ? acc[r] * mul : fill;
        }
    }
}

// Z[i][d] (+)= sum_j A[i][j] * Y[j][d]: A is the LDS tile (rows i, K index j; TRANS: A = t[j][i] instead of t[i][j]),
// Y rows j < nk from global, output rows i < nm to global Z[i][C-strided], all dh columns of the head.
template <bool TRANS>
__device__ __forceinline__ void tile_times_rows(const float* t, int ldt, int nm, int nk, const float* __restrict__ Y,
                                                int C, int dh, float mul, float* __restrict__ Z) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int nrt = (nm + 31) / 32, nct = dh / 32;
    for (int tl = wave; tl < nrt * nct; tl += NTH / 64) {
        const int rt = tl / nct, ct = tl - rt * nct;
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int ia = rt * 32 + lr;
        for (int k0 = 0; k0 < nk; k0 += 2) {
            const int kk = k0 + lh;
            const bool ok = kk < nk;
            const float a = (ok && ia < nm) ? (TRANS ? t[kk * ldt + ia] : t[ia * ldt + kk]) : 0.f;
            const float b = Y[(size_t)(ok ? kk : 0) * C + ct * 32 + lr];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ok ? b : 0.f, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (i < nm) Z[(size_t)i * C + ct * 32 + lr] = acc[r] * mul;
        }
    }
}

__global__ __launch_bounds__(NTH) void xattn_fwd_kernel(AttnP p) {
    __shared__ float s[TQ_MAX * (TK_MAX + 1)];
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads, dh = p.C / p.heads, ldt = TK_MAX + 1;
    const float* q = p.q + (size_t)b * p.Tq * p.C + h * dh;
    const float* k = p.k + (size_t)b * p.Tk * p.C + h * dh;
    const float* v = p.v + (size_t)b * p.Tk * p.C + h * dh;
    rows_dot_rows(q, p.Tq, k, p.Tk, p.C, dh, p.scale, -INFINITY, s, ldt);
    __syncthreads();
    // softmax over j: 8 lanes per row (32 rows x 8 = 256 threads)
    {
        const int i = threadIdx.x >> 3, sub = threadIdx.x & 7;
        float m = -INFINITY;
        for (int j = sub; j < p.Tk; j += 8) m = fmaxf(m, s[i * ldt + j]);
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sum = 0.f;
        if (i < p.Tq)
            for (int j = sub; j < p.Tk; j += 8) { const float e = __expf(s[i * ldt + j] - m); s[i * ldt + j] = e; sum += e; }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = i < p.Tq ? 1.f / sum : 0.f;
        float* pr = p.prob + (((size_t)b * p.heads + h) * p.Tq + min(i, p.Tq - 1)) * p.Tk;
        for (int j = sub; j < p.Tk; j += 8) {
            const float pv = i < p.Tq ? s[i * ldt + j] * inv : 0.f;
            s[i * ldt + j] = pv;
            if (i < p.Tq) pr[j] = pv;
        }
    }
    __syncthreads();
    tile_times_rows<false>(s, ldt, p.Tq, p.Tk, v, p.C, dh, 1.f, p.out + (size_t)b * p.Tq * p.C + h * dh);
}

__global__ __launch_bounds__(NTH) void xattn_bwd_kernel(AttnP p) {
    __shared__ float s[TQ_MAX * (TK_MAX + 1)];         // dP, then dS
    __shared__ float pr[TQ_MAX * (TK_MAX + 1)];        // P
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads, dh = p.C / p.heads, ldt = TK_MAX + 1;
    const size_t qo = (size_t)b * p.Tq * p.C + h * dh, ko = (size_t)b * p.Tk * p.C + h * dh;
    const float* P = p.prob + ((size_t)b * p.heads + h) * p.Tq * p.Tk;
    for (int i = threadIdx.x; i < TQ_MAX * p.Tk; i += NTH) {
        const int r = i / p.Tk, j = i - r * p.Tk;
        pr[r * ldt + j] = r < p.Tq ? P[(size_t)r * p.Tk + j] : 0.f;
    }
    rows_dot_rows(p.dout + qo, p.Tq, p.v + ko, p.Tk, p.C, dh, 1.f, 0.f, s, ldt);       // dP = dO V^T
    __syncthreads();
    {
        const int i = threadIdx.x >> 3, sub = threadIdx.x & 7;
        float dot = 0.f;
        for (int j = sub; j < p.Tk; j += 8) dot += s[i * ldt + j] * pr[i * ldt + j];
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        for (int j = sub; j < p.Tk; j += 8) s[i * ldt + j] = pr[i * ldt + j] * (s[i * ldt + j] - dot);   // dS (unscaled)
    }
    __syncthreads();
    tile_times_rows<false>(s, ldt, p.Tq, p.Tk, p.k + ko, p.C, dh, p.scale, p.dq + qo);     // dQ = scale dS K
    tile_times_rows<true>(s, ldt, p.Tk, p.Tq, p.q + qo, p.C, dh, p.scale, p.dk + ko);      // dK = scale dS^T Q
    tile_times_rows<true>(pr, ldt, p.Tk, p.Tq, p.dout + qo, p.C, dh, 1.f, p.dv + ko);      // dV = P^T dO
}

int attn_check(int B, int Tq, int Tk, int C, int heads, const char* what) {
    DTT_CHECK(B > 0 && heads > 0 && C % heads == 0 && (C / heads) % 32 == 0 && Tq >= 1 && Tq <= TQ_MAX && Tk >= 1 && Tk <= TK_MAX,
              "%s: unsupported shape Tq=%d (<= %d) Tk=%d (<= %d) C=%d heads=%d (head width a multiple of 32)", what, Tq,
              TQ_MAX, Tk, TK_MAX, C, heads);
    return 0;
}
}  // namespace

// q [B][Tq][C], k, v [B][Tk][C] fp32 (already projected; head h owns columns h*C/heads ..), out [B][Tq][C],
// prob [B][heads][Tq][Tk] (softmax probabilities, kept for backward).
extern "C" int dtt_xattn_fwd(const float* q, const float* k, const float* v, float* out, float* prob, int B, int Tq, int Tk,
                             int C, int heads, float scale, void* stream) {
    if (int rc = attn_check(B, Tq, Tk, C, heads, "dtt_xattn_fwd")) return rc;
    DTT_CHECK(q && k && v && out && prob, "dtt_xattn_fwd: null buffer");
    AttnP p{q, k, v, nullptr, out, prob, nullptr, nullptr, nullptr, B, Tq, Tk, C, heads, scale};
    hipLaunchKernelGGL(xattn_fwd_kernel, dim3(B * heads), dim3(NTH), 0, reinterpret_cast<hipStream_t>(stream), p);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_xattn_bwd(const float* q, const float* k, const float* v, const float* prob, const float* dout, float* dq,
                             float* dk, float* dv, int B, int Tq, int Tk, int C, int heads, float scale, void* stream) {
    if (int rc = attn_check(B, Tq, Tk, C, heads, "dtt_xattn_bwd")) return rc;
    DTT_CHECK(q && k && v && prob && dout && dq && dk && dv, "dtt_xattn_bwd: null buffer");
    AttnP p{q, k, v, dout, nullptr, const_cast<float*>(prob), dq, dk, dv, B, Tq, Tk, C, heads, scale};
    hipLaunchKernelGGL(xattn_bwd_kernel, dim3(B * heads), dim3(NTH), 0, reinterpret_cast<hipStream_t>(stream), p);
    DTT_LAUNCH_CHECK();
    return 0;
}
