// BatchNorm (+ residual) + activation on channels-last tensors [P pixels][C], forward and backward, for the
// lip encoders on the audio-visual path (reference: src/model/VideoEncoders/resnet.py:36-81 BasicBlock with
// relu_type="swish": bn -> swish, bn -> (+ residual) -> swish, and the downsample bn without activation;
// src/model/VideoEncoders/resnet_video_encoder.py:40-44).  HBM-bound passes: one thread owns a 16-byte channel
// chunk with its per-channel constants in registers and walks pixels with four loads in flight; reductions are
// two-stage and deterministic.  Statistics {mean, rstd} come from dtt_in_stats (B = 1: the whole batch is one
// plane).  The activation input is recomputed in backward from x (and the residual), so nothing but x is saved.
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {
constexpr int NT = 256;

// act: 0 identity, 1 swish x * sigmoid(x), 2 relu
template <int ACT> __device__ __forceinline__ float act_f(float a) {
    if (ACT == 1) return a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a));
    if (ACT == 2) return fmaxf(a, 0.f);
    return a;
}
template <int ACT> __device__ __forceinline__ float act_grad(float a) {
    if (ACT == 1) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a));
        return sg * (1.f + a * (1.f - sg));
    }
    if (ACT == 2) return a > 0.f ? 1.f : 0.f;
    return 1.f;
}

template <typename T, int ACT>
__global__ __launch_bounds__(NT) void bn_act_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ z, int P, int C,
                                                        int chunk) {
    constexpr int EPT = Vec16<T>::N, U = 4;
    const int cpp = C / EPT, ppi = NT / cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    if (pl >= ppi) return;
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    float sc[EPT], sh[EPT];                                // z = act(x * sc + sh (+ res))
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        const float mean = stats[c * 2], rstd = stats[c * 2 + 1];
        sc[e] = gamma[c] * rstd;
        sh[e] = beta[c] - mean * sc[e];
    }
    const T* xb = x + cq * EPT;
    const T* rb = res ? res + cq * EPT : nullptr;
    T* zb = z + cq * EPT;
    for (int p = p0 + pl; p < p1; p += U * ppi) {
        Vec16<T> v[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            v[u] = ld16_pred<T>(xb + (int64_t)q * C, xb, q < p1);
            if (rb) r[u] = ld16_pred<T>(rb + (int64_t)q * C, rb, q < p1);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            float o[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) o[e] = act_f<ACT>(v[u].get(e) * sc[e] + sh[e] + (rb ? r[u].get(e) : 0.f));
            Vec16<T> ov;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < EPT; e += 2) ov.set2(e, o[e], o[e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < EPT; ++e) ov.set(e, o[e]);
            }
            if (q < p1) st16<T>(zb + (int64_t)q * C, ov);
        }
    }
}

// per-thread vectors -> per-channel sums over the threads that own the same channel chunk (tid % cpp)
template <int EPT>
__device__ __forceinline__ void reduce_ch(const float (&v)[EPT], float* red, float* out, int cpp, int nact, int C) {
    const int tid = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) red[tid * EPT + e] = (tid < nact) ? v[e] : 0.f;
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        const int cqq = c / EPT, e = c % EPT;
        float s = 0.f;
        for (int t = cqq; t < nact; t += cpp) s += red[t * EPT + e];
        out[c] = s;
    }
    __syncthreads();
}

// pass 1: partial[chunk][C][2] = {sum da, sum da * xhat},  da = dz * act'(gamma * xhat + beta (+ res))
template <typename T, int ACT>
__global__ __launch_bounds__(NT) void bn_act_bwd_partial(const T* __restrict__ x, const T* __restrict__ res,
                                                         const T* __restrict__ dz, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ partial, int P, int C, int chunk) {
    constexpr int EPT = Vec16<T>::N, U = 4;
    extern __shared__ float sm[];
    float* red = sm;
    float* chv = sm + NT * EPT;
    const int cpp = C / EPT, ppi = NT / cpp, nact = ppi * cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    float s1[EPT], s2[EPT], mean[EPT], rstd[EPT], ga[EPT], be[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = (cq * EPT + e) % C;
        s1[e] = 0.f; s2[e] = 0.f;
        mean[e] = stats[c * 2]; rstd[e] = stats[c * 2 + 1]; ga[e] = gamma[c]; be[e] = beta[c];
    }
    if (tid < nact) {
        const T* xb = x + cq * EPT;
        const T* rb = res ? res + cq * EPT : nullptr;
        const T* gb = dz + cq * EPT;
        for (int p = p0 + pl; p < p1; p += U * ppi) {
            Vec16<T> v[U], r[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = p + u * ppi;
                v[u] = ld16_pred<T>(xb + (int64_t)q * C, xb, q < p1);
                g[u] = ld16_pred<T>(gb + (int64_t)q * C, gb, q < p1);      // zero beyond the chunk: contributes nothing
                if (rb) r[u] = ld16_pred<T>(rb + (int64_t)q * C, rb, q < p1);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const float xh = (v[u].get(e) - mean[e]) * rstd[e];
                    const float d = g[u].get(e) * act_grad<ACT>(ga[e] * xh + be[e] + (rb ? r[u].get(e) : 0.f));
                    s1[e] += d; s2[e] += d * xh;
                }
        }
    }
    float* dst = partial + ((int64_t)blockIdx.x * C) * 2;
    reduce_ch<EPT>(s1, red, chv, cpp, nact, C);
    for (int c = tid; c < C; c += NT) dst[c * 2] = chv[c];
    reduce_ch<EPT>(s2, red, chv, cpp, nact, C);
    for (int c = tid; c < C; c += NT) dst[c * 2 + 1] = chv[c];
}

// sums[c] = {sum da, sum da * xhat} over all chunks (fixed order); dgb[c] = {dgamma, dbeta}
__global__ __launch_bounds__(256) void bn_bwd_finalize(const float* __restrict__ partial, float* __restrict__ sums,
                                                       float* __restrict__ dgb, int C, int nchunks) {
    __shared__ float sh[2][8][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float a = 0.f, q = 0.f;
    if (c < C)
        for (int k = g; k < nchunks; k += 8) {
            const float2 s = *reinterpret_cast<const float2*>(partial + ((int64_t)k * C + c) * 2);
            a += s.x; q += s.y;
        }
    sh[0][g][cl] = a; sh[1][g][cl] = q;
    __syncthreads();
    if (g == 0 && c < C) {
        float ta = 0.f, tq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ta += sh[0][i][cl]; tq += sh[1][i][cl]; }
        sums[c * 2] = ta; sums[c * 2 + 1] = tq;
        dgb[c * 2] = tq; dgb[c * 2 + 1] = ta;
    }
}

// pass 2: dx = gamma * rstd * (da - S1/n - xhat * S2/n)  (training) or gamma * rstd * da (frozen statistics);
// dres = da
template <typename T, int ACT>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply(const T* __restrict__ x, const T* __restrict__ res,
                                                       const T* __restrict__ dz, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ sums, T* __restrict__ dx,
                                                       T* __restrict__ dres, int P, int C, int chunk, int training) {
    constexpr int EPT = Vec16<T>::N, U = 4;
    const int cpp = C / EPT, ppi = NT / cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    if (pl >= ppi) return;
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    float mean[EPT], rstd[EPT], ga[EPT], be[EPT], m1[EPT], m2[EPT];
    const float inv_n = training ? 1.f / (float)P : 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        mean[e] = stats[c * 2]; rstd[e] = stats[c * 2 + 1]; ga[e] = gamma[c]; be[e] = beta[c];
        m1[e] = sums[c * 2] * inv_n; m2[e] = sums[c * 2 + 1] * inv_n;
    }
    const T* xb = x + cq * EPT;
    const T* rb = res ? res + cq * EPT : nullptr;
    const T* gb = dz + cq * EPT;
    T* ob = dx + cq * EPT;
    T* orb = dres ? dres + cq * EPT : nullptr;
    for (int p = p0 + pl; p < p1; p += U * ppi) {
        Vec16<T> v[U], r[U], g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            v[u] = ld16_pred<T>(xb + (int64_t)q * C, xb, q < p1);
            g[u] = ld16_pred<T>(gb + (int64_t)q * C, gb, q < p1);
            if (rb) r[u] = ld16_pred<T>(rb + (int64_t)q * C, rb, q < p1);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            float o[EPT], da[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float xh = (v[u].get(e) - mean[e]) * rstd[e];
                da[e] = g[u].get(e) * act_grad<ACT>(ga[e] * xh + be[e] + (rb ? r[u].get(e) : 0.f));
                o[e] = ga[e] * rstd[e] * (da[e] - m1[e] - xh * m2[e]);
            }
            Vec16<T> ov, rv;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < EPT; e += 2) { ov.set2(e, o[e], o[e + 1]); rv.set2(e, da[e], da[e + 1]); }
            } else {
#pragma unroll
                for (int e = 0; e < EPT; ++e) { ov.set(e, o[e]); rv.set(e, da[e]); }
            }
            if (q < p1) {
                st16<T>(ob + (int64_t)q * C, ov);
                if (orb) st16<T>(orb + (int64_t)q * C, rv);
            }
        }
    }
}

// running_mean / running_var update of nn.BatchNorm (momentum form, unbiased variance), and the frozen-statistics
// {mean, rstd} table for eval mode.
__global__ void bn_running_kernel(const float* __restrict__ stats, float* __restrict__ rmean, float* __restrict__ rvar,
                                  int C, float n, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c * 2], rstd = stats[c * 2 + 1];
    const float var = fmaxf(1.f / (rstd * rstd) - eps, 0.f);
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
}
__global__ void bn_frozen_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                       float* __restrict__ stats, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats[c * 2] = rmean[c];
    stats[c * 2 + 1] = rsqrtf(rvar[c] + eps);
}

int bn_chunk(int P) {
    int nch = P / 64;
    if (nch > 2048) nch = 2048;
    if (nch < 1) nch = 1;
    return (P + nch - 1) / nch;
}
}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int dtt_bn_chunks(int P) { const int ch = bn_chunk(P); return (P + ch - 1) / ch; }

static int bn_check(int P, int C, int dtype, const char* what) {
    const int ept = dtype ? 8 : 4;
    DTT_CHECK(P > 0 && C > 0 && C % ept == 0 && C / ept <= NT, "%s: C=%d must be a multiple of %d and at most %d", what, C, ept, NT * ept);
    return 0;
}

extern "C" int dtt_bn_act_fwd(const void* x, const void* res, const float* stats, const float* gamma, const float* beta,
                              void* z, int P, int C, int act, int dtype, void* stream) {
    if (int rc = bn_check(P, C, dtype, "dtt_bn_act_fwd")) return rc;
    DTT_CHECK(x && stats && gamma && beta && z && act >= 0 && act <= 2, "dtt_bn_act_fwd: bad arguments");
    const int chunk = bn_chunk(P), nch = (P + chunk - 1) / chunk;
#define BF(TT, A) hipLaunchKernelGGL((bn_act_fwd_kernel<TT, A>), dim3(nch), dim3(NT), 0, ST, (const TT*)x, (const TT*)res, stats, gamma, beta, (TT*)z, P, C, chunk)
    if (dtype) { if (act == 1) BF(bf16, 1); else if (act == 2) BF(bf16, 2); else BF(bf16, 0); }
    else { if (act == 1) BF(float, 1); else if (act == 2) BF(float, 2); else BF(float, 0); }
#undef BF
    DTT_LAUNCH_CHECK();
    return 0;
}

// partial: dtt_bn_chunks(P) * C * 2 floats of scratch; sums [C][2] scratch; dgb [C][2] = {dgamma, dbeta};
// dres may be NULL (then no residual gradient is written; res may still be given).
extern "C" int dtt_bn_act_bwd(const void* x, const void* res, const void* dz, const float* stats, const float* gamma,
                              const float* beta, float* partial, float* sums, float* dgb, void* dx, void* dres, int P,
                              int C, int act, int training, int dtype, void* stream) {
    if (int rc = bn_check(P, C, dtype, "dtt_bn_act_bwd")) return rc;
    DTT_CHECK(x && dz && stats && gamma && beta && partial && sums && dgb && dx && act >= 0 && act <= 2, "dtt_bn_act_bwd: bad arguments");
    const int chunk = bn_chunk(P), nch = (P + chunk - 1) / chunk;
    const int ept = dtype ? 8 : 4;
    const size_t sm = (size_t)(NT * ept + C) * sizeof(float);
#define BP(TT, A) hipLaunchKernelGGL((bn_act_bwd_partial<TT, A>), dim3(nch), dim3(NT), sm, ST, (const TT*)x, (const TT*)res, (const TT*)dz, stats, gamma, beta, partial, P, C, chunk)
    if (dtype) { if (act == 1) BP(bf16, 1); else if (act == 2) BP(bf16, 2); else BP(bf16, 0); }
    else { if (act == 1) BP(float, 1); else if (act == 2) BP(float, 2); else BP(float, 0); }
#undef BP
    hipLaunchKernelGGL(bn_bwd_finalize, dim3((C + 31) / 32), dim3(256), 0, ST, partial, sums, dgb, C, nch);
#define BA(TT, A) hipLaunchKernelGGL((bn_act_bwd_apply<TT, A>), dim3(nch), dim3(NT), 0, ST, (const TT*)x, (const TT*)res, (const TT*)dz, stats, gamma, beta, sums, (TT*)dx, (TT*)dres, P, C, chunk, training)
    if (dtype) { if (act == 1) BA(bf16, 1); else if (act == 2) BA(bf16, 2); else BA(bf16, 0); }
    else { if (act == 1) BA(float, 1); else if (act == 2) BA(float, 2); else BA(float, 0); }
#undef BA
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_bn_running(const float* stats, float* running_mean, float* running_var, int C, int64_t n, float momentum,
                              float eps, void* stream) {
    DTT_CHECK(stats && running_mean && running_var && C > 0 && n > 0, "dtt_bn_running: bad arguments");
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 127) / 128), dim3(128), 0, ST, stats, running_mean, running_var, C, (float)n, momentum, eps);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_bn_frozen_stats(const float* running_mean, const float* running_var, float* stats, int C, float eps,
                                   void* stream) {
    DTT_CHECK(stats && running_mean && running_var && C > 0, "dtt_bn_frozen_stats: bad arguments");
    hipLaunchKernelGGL(bn_frozen_stats_kernel, dim3((C + 127) / 128), dim3(128), 0, ST, running_mean, running_var, stats, C, eps);
    DTT_LAUNCH_CHECK();
    return 0;
}
