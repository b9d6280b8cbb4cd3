// BatchNorm (+ residual) + activation on channels-last tensors [P pixels][C], forward and backward, for the
// lip encoders on the audio-visual path (reference: src/model/VideoEncoders/resnet.py:36-81 BasicBlock with
// relu_type="swish": bn -> swish, bn -> (+ residual) -> swish, and the downsample bn without activation;
// src/model/VideoEncoders/resnet_video_encoder.py:40-44).  HBM-bound passes: one thread owns a 16-byte channel
// chunk with its per-channel constants in registers and walks pixels with four loads in flight; reductions are
// two-stage and deterministic.  Statistics {mean, rstd} come from dtt_in_stats (B = 1: the whole batch is one
// plane).  The activation input is recomputed in backward from x (and the residual), so nothing but x is saved.
#include <algorithm>
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

// batch statistics, pass 1: partial[chunk][C][2] = {mean, M2} of the chunk's pixels in ONE sweep: sums of (x - pivot)
// and (x - pivot)^2 with the chunk's first pixel as the pivot (M2 = S2 - S1^2 / n; the pivot keeps S1^2 / n small
// against S2).  The two-sweep form read the tensor twice from HBM: 512 chunks of a 400 MB tensor are not L2-resident.
template <typename T>
__global__ __launch_bounds__(NT) void bn_stats_partial(const T* __restrict__ x, float* __restrict__ partial, int P, int C,
                                                       int chunk) {
    constexpr int EPT = Vec16<T>::N, U = 4;
    extern __shared__ float sm[];
    float* red = sm;
    float* chv = sm + NT * EPT;
    const int cpp = C / EPT, ppi = NT / cpp, nact = ppi * cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    const T* base = x + (cq % cpp) * EPT;
    float s1[EPT], s2[EPT], piv[EPT];
    {
        const Vec16<T> pv = ld16<T>(base + (int64_t)p0 * C);
#pragma unroll
        for (int e = 0; e < EPT; ++e) { piv[e] = pv.get(e); s1[e] = 0.f; s2[e] = 0.f; }
    }
    if (tid < nact)
        for (int p = p0 + pl; p < p1; p += U * ppi) {
            Vec16<T> v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld16_pred<T>(base + (int64_t)(p + u * ppi) * C, base + (int64_t)p0 * C, p + u * ppi < p1);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const float d = (p + u * ppi < p1) ? v[u].get(e) - piv[e] : 0.f;
                    s1[e] += d; s2[e] += d * d;
                }
        }
    const float inv = 1.f / (float)(p1 - p0);
    float* dst = partial + ((int64_t)blockIdx.x * C) * 2;
    reduce_ch<EPT>(s1, red, chv, cpp, nact, C);
    float t1[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) t1[e] = chv[(cq % cpp) * EPT + e];
    reduce_ch<EPT>(s2, red, chv, cpp, nact, C);
    if (tid < cpp)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            dst[(cq * EPT + e) * 2] = piv[e] + t1[e] * inv;
            dst[(cq * EPT + e) * 2 + 1] = fmaxf(chv[cq * EPT + e] - t1[e] * t1[e] * inv, 0.f);
        }
}

// pass 2: Chan merge of the chunk partials, FG lane groups per channel each walking every FG-th chunk, then a
// fixed-order merge of the FG (a single serial walk over ~500 chunks per channel cost 270 us per BatchNorm, 8
// groups 21 us, 32 groups of a 1024-thread workgroup ~8); stats[c] = {mean, rstd}
constexpr int FG = 32;
// run: the running-buffer update of nn.BatchNorm in the same launch (NULL buffers: none).  Column c of a tensor held as
// halves of h channels padded to hp (lipunit.hip) is logical channel (c / hp) * h + c % hp; dense tensors pass h = hp = C.
struct BnRun { float* rmean; float* rvar; float momentum; int h, hp; };
__global__ __launch_bounds__(32 * FG) void bn_stats_finalize(const float* __restrict__ partial, float* __restrict__ stats, int P,
                                                         int C, int chunk, int nchunks, float eps, BnRun run) {
    __shared__ float sh[3][FG][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C)
        for (int k = g; k < nchunks; k += FG) {
            const float2 s = *reinterpret_cast<const float2*>(partial + ((int64_t)k * C + c) * 2);
            const float nb = (float)(min(P, (k + 1) * chunk) - k * chunk);
            const float d = s.x - mean, nn = n + nb;
            mean += d * nb / nn;
            m2 += s.y + d * d * n * nb / nn;
            n = nn;
        }
    sh[0][g][cl] = n; sh[1][g][cl] = mean; sh[2][g][cl] = m2;
    __syncthreads();
    if (g == 0 && c < C) {
        float tn = 0.f, tm = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < FG; ++i) {
            const float nb = sh[0][i][cl];
            if (nb > 0.f) {
                const float d = sh[1][i][cl] - tm, nn = tn + nb;
                tm += d * nb / nn;
                t2 += sh[2][i][cl] + d * d * tn * nb / nn;
                tn = nn;
            }
        }
        stats[c * 2] = tm;
        stats[c * 2 + 1] = rsqrtf(t2 / tn + eps);
        if (run.rmean) {
            const int half = c / run.hp, pos = c - half * run.hp;
            if (pos < run.h) {
                const int lc = half * run.h + pos;
                const float unb = t2 / tn * (tn > 1.f ? tn / (tn - 1.f) : 1.f);
                run.rmean[lc] = (1.f - run.momentum) * run.rmean[lc] + run.momentum * tm;
                run.rvar[lc] = (1.f - run.momentum) * run.rvar[lc] + run.momentum * unb;
            }
        }
    }
}

// pass 1: partial[chunk][C][2] = {sum da, sum da * xhat},  da = dz * act'(gamma * xhat + beta (+ res))
// POOL: dz is not given per pixel but as the gradient of a 3x3 / stride-2 / pad-1 max pool of the activation output
// (dz = pooled gradient [N][Ho][Wo][C], idx = the pool's argmax bytes): the pixel's gradient is gathered from the <= 4
// windows that cover it (pool_grad), so the pre-pool gradient tensor is never materialised.
struct PoolGeom { const unsigned char* idx; int H, W, Ho, Wo; };
template <typename T>
__device__ __forceinline__ Vec16<T> pool_grad(const T* __restrict__ dyp, const PoolGeom& g, int q, int C, int cq, bool ok) {
    constexpr int EPT = Vec16<T>::N;
    const int w = q % g.W, r = q / g.W, h = r % g.H, n = r / g.H;
    // The windows that cover row h: ho0 = (h + 1) / 2 at kh0 = h + 1 - 2 ho0 (0 for odd h, 1 for even h) and, for odd
    // h only, ho0 - 1 at kh0 + 2.  Columns alike.  All four candidates are loaded unconditionally (a dead candidate
    // reads the tensor's first element and is masked): four independent loads in flight, no branch.
    const int ho0 = (h + 1) >> 1, kh0 = h + 1 - 2 * ho0, wo0 = (w + 1) >> 1, kw0 = w + 1 - 2 * wo0;
    Vec16<T> gv[4];
    unsigned long long ib[4];
    bool live[4];
    unsigned char want[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ho = ho0 - a, wo = wo0 - b, kh = kh0 + 2 * a, kw = kw0 + 2 * b, k = a * 2 + b;
            live[k] = ok && ho >= 0 && ho < g.Ho && wo >= 0 && wo < g.Wo && kh <= 2 && kw <= 2;
            want[k] = (unsigned char)(kh * 3 + kw);
            const int64_t ob = live[k] ? (((int64_t)n * g.Ho + ho) * g.Wo + wo) * C + cq * EPT : 0;
            gv[k] = ld16<T>(dyp + ob);
            if (EPT == 8) ib[k] = *reinterpret_cast<const unsigned long long*>(g.idx + ob);
            else ib[k] = *reinterpret_cast<const unsigned int*>(g.idx + ob);
        }
    float acc[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            acc[e] += (live[k] && (unsigned char)(ib[k] >> (8 * e)) == want[k]) ? gv[k].get(e) : 0.f;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < EPT; ++e) o.set(e, acc[e]);
    return o;
}

template <typename T, int ACT, bool POOL = false>
__global__ __launch_bounds__(NT) void bn_act_bwd_partial(const T* __restrict__ x, const T* __restrict__ res,
                                                         const T* __restrict__ dz, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ partial, int P, int C, int chunk,
                                                         PoolGeom pg = PoolGeom{}) {
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
                if constexpr (POOL) g[u] = pool_grad<T>(dz, pg, q, C, cq, q < p1);
                else g[u] = ld16_pred<T>(gb + (int64_t)q * C, gb, q < p1);      // zero beyond the chunk: contributes nothing
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
__global__ __launch_bounds__(32 * FG) void bn_bwd_finalize(const float* __restrict__ partial, float* __restrict__ sums,
                                                          float* __restrict__ dgb, int C, int nchunks) {
    __shared__ float sh[2][FG][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float a = 0.f, q = 0.f;
    if (c < C)
        for (int k = g; k < nchunks; k += FG) {
            const float2 s = *reinterpret_cast<const float2*>(partial + ((int64_t)k * C + c) * 2);
            a += s.x; q += s.y;
        }
    sh[0][g][cl] = a; sh[1][g][cl] = q;
    __syncthreads();
    if (g == 0 && c < C) {
        float ta = 0.f, tq = 0.f;
#pragma unroll
        for (int i = 0; i < FG; ++i) { ta += sh[0][i][cl]; tq += sh[1][i][cl]; }
        sums[c * 2] = ta; sums[c * 2 + 1] = tq;
        dgb[c * 2] = tq; dgb[c * 2 + 1] = ta;
    }
}

// pass 2: dx = gamma * rstd * (da - S1/n - xhat * S2/n)  (training) or gamma * rstd * da (frozen statistics);
// dres = da
template <typename T, int ACT, bool POOL = false>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply(const T* __restrict__ x, const T* __restrict__ res,
                                                       const T* __restrict__ dz, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ sums, T* __restrict__ dx,
                                                       T* __restrict__ dres, int P, int C, int chunk, int training,
                                                       PoolGeom pg = PoolGeom{}) {
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
            if constexpr (POOL) g[u] = pool_grad<T>(dz, pg, q, C, cq, q < p1);
            else g[u] = ld16_pred<T>(gb + (int64_t)q * C, gb, q < p1);
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

// act(batch_norm(x)) followed by the 3x3 / stride-2 / pad-1 max pool in one pass: one lane = one pooled pixel x one
// 16-byte channel chunk; the nine activation values are recomputed from x (2.25 reads per input element, served by
// L1 / L2), rounded to T as the unfused pair would store them, first maximum in (kh, kw) scan order.  The 400 MB
// pre-pool tensor of the lip front end is neither written nor read.
template <typename T, int ACT>
__global__ __launch_bounds__(NT) void bn_act_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             T* __restrict__ y, unsigned char* __restrict__ idx, int N, int H,
                                                             int W, int C, int Ho, int Wo) {
    constexpr int EPT = Vec16<T>::N;
    const int cpp = C / EPT;
    const int64_t total = (int64_t)N * Ho * Wo * cpp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int cq = (int)(i % cpp);
        int64_t q = i / cpp;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho);
        const int n = (int)(q / Ho);
        float sc[EPT], sh[EPT], best[EPT];
        unsigned char bi[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int c = cq * EPT + e;
            const float mean = stats[c * 2], rstd = stats[c * 2 + 1];
            sc[e] = gamma[c] * rstd;
            sh[e] = beta[c] - mean * sc[e];
            best[e] = -INFINITY; bi[e] = 4;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = 2 * ho - 1 + kh, w = 2 * wo - 1 + kw;
                if (h < 0 || h >= H || w < 0 || w >= W) continue;
                const Vec16<T> v = ld16<T>(x + (((int64_t)n * H + h) * W + w) * C + cq * EPT);
                Vec16<T> zr;
#pragma unroll
                for (int e = 0; e < EPT; ++e) zr.set(e, act_f<ACT>(v.get(e) * sc[e] + sh[e]));
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const float f = zr.get(e);
                    if (f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(kh * 3 + kw); }
                }
            }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPT; ++e) o.set(e, best[e]);
        const int64_t ob = (((int64_t)n * Ho + ho) * Wo + wo) * C + cq * EPT;
        st16<T>(y + ob, o);
#pragma unroll
        for (int e = 0; e < EPT; ++e) idx[ob + e] = bi[e];
    }
}

// ---------------------------------------------------------------------------------------------
// MaxPool 3x3, stride 2, padding 1 on channels-last [N][H][W][C] (the spatial part of the lip encoders'
// MaxPool3d((1,3,3),(1,2,2),(0,1,1)), resnet_video_encoder.py:44).  The window position of the maximum (first
// maximum in (kh, kw) scan order, as ATen picks it) is kept as one byte per output element; backward is a gather
// over the <= 4 windows that cover an input pixel: no atomics, deterministic.
template <typename T>
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         unsigned char* __restrict__ idx, int N, int H, int W, int C,
                                                         int Ho, int Wo) {
    constexpr int EPT = Vec16<T>::N;
    const int cpp = C / EPT;
    const int64_t total = (int64_t)N * Ho * Wo * cpp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int cq = (int)(i % cpp);
        int64_t q = i / cpp;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho);
        const int n = (int)(q / Ho);
        float best[EPT];
        unsigned char bi[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) { best[e] = -INFINITY; bi[e] = 4; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = 2 * ho - 1 + kh, w = 2 * wo - 1 + kw;
                if (h < 0 || h >= H || w < 0 || w >= W) continue;
                const Vec16<T> v = ld16<T>(x + (((int64_t)n * H + h) * W + w) * C + cq * EPT);
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const float f = v.get(e);
                    if (f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(kh * 3 + kw); }
                }
            }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPT; ++e) o.set(e, best[e]);
        const int64_t ob = (((int64_t)n * Ho + ho) * Wo + wo) * C + cq * EPT;
        st16<T>(y + ob, o);
#pragma unroll
        for (int e = 0; e < EPT; ++e) idx[ob + e] = bi[e];
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                         T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    constexpr int EPT = Vec16<T>::N;
    const int cpp = C / EPT;
    const int64_t total = (int64_t)N * H * W * cpp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int cq = (int)(i % cpp);
        int64_t q = i / cpp;
        const int w = (int)(q % W); q /= W;
        const int h = (int)(q % H);
        const int n = (int)(q / H);
        float acc[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
        // windows (ho, wo) with 2*ho - 1 + kh == h: ho in {(h+1)/2 (kh = h+1-2ho), ...}
        for (int ho = (h + 1) / 2; ho >= 0 && 2 * ho - 1 <= h; --ho) {
            const int kh = h + 1 - 2 * ho;
            if (kh > 2) break;
            if (ho >= Ho) continue;
            for (int wo = (w + 1) / 2; wo >= 0 && 2 * wo - 1 <= w; --wo) {
                const int kw = w + 1 - 2 * wo;
                if (kw > 2) break;
                if (wo >= Wo) continue;
                const int64_t ob = (((int64_t)n * Ho + ho) * Wo + wo) * C + cq * EPT;
                const Vec16<T> g = ld16<T>(dy + ob);
                const unsigned char want = (unsigned char)(kh * 3 + kw);
                unsigned char ib[EPT];
                if (EPT == 8) *reinterpret_cast<uint2*>(ib) = *reinterpret_cast<const uint2*>(idx + ob);
                else *reinterpret_cast<uint32_t*>(ib) = *reinterpret_cast<const uint32_t*>(idx + ob);
#pragma unroll
                for (int e = 0; e < EPT; ++e) acc[e] += ib[e] == want ? g.get(e) : 0.f;
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPT; ++e) o.set(e, acc[e]);
        st16<T>(dx + (((int64_t)n * H + h) * W + w) * C + cq * EPT, o);
    }
}

int bn_chunk(int P) {
    int nch = P / 64;
    if (nch > 512) nch = 512;                               // ~2 workgroups per CU; the merges walk nch / 8 partials
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

// Batch statistics over all P pixels: stats [C][2] = {mean, rstd} (biased variance + eps, as nn.BatchNorm normalises);
// partial: dtt_bn_chunks(P) * C * 2 floats of scratch.
static int bn_stats_launch(const void* x, float* partial, float* stats, int P, int C, float eps, int dtype, BnRun run, void* stream);
extern "C" int dtt_bn_stats(const void* x, float* partial, float* stats, int P, int C, float eps, int dtype, void* stream) {
    return bn_stats_launch(x, partial, stats, P, C, eps, dtype, BnRun{nullptr, nullptr, 0.f, C, C}, stream);
}
// ... and the running-buffer update (momentum form, unbiased variance) in the same launches; h / hp: the halves layout
// of lipunit.hip (h = hp = C for a dense tensor)
extern "C" int dtt_bn_stats_running(const void* x, float* partial, float* stats, int P, int C, float eps, int dtype,
                                    float* running_mean, float* running_var, float momentum, int h, int hp, void* stream) {
    DTT_CHECK(running_mean && running_var && h > 0 && hp >= h && C % hp == 0, "dtt_bn_stats_running: bad arguments");
    return bn_stats_launch(x, partial, stats, P, C, eps, dtype, BnRun{running_mean, running_var, momentum, h, hp}, stream);
}
static int bn_stats_launch(const void* x, float* partial, float* stats, int P, int C, float eps, int dtype, BnRun run, void* stream) {
    if (int rc = bn_check(P, C, dtype, "dtt_bn_stats")) return rc;
    DTT_CHECK(x && partial && stats, "dtt_bn_stats: bad arguments");
    const int chunk = bn_chunk(P), nch = (P + chunk - 1) / chunk;
    const int ept = dtype ? 8 : 4;
    const size_t sm = (size_t)(NT * ept + C) * sizeof(float);
    if (dtype) hipLaunchKernelGGL(bn_stats_partial<bf16>, dim3(nch), dim3(NT), sm, ST, (const bf16*)x, partial, P, C, chunk);
    else hipLaunchKernelGGL(bn_stats_partial<float>, dim3(nch), dim3(NT), sm, ST, (const float*)x, partial, P, C, chunk);
    hipLaunchKernelGGL(bn_stats_finalize, dim3((C + 31) / 32), dim3(32 * FG), 0, ST, partial, stats, P, C, chunk, nch, eps, run);
    DTT_LAUNCH_CHECK();
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
    hipLaunchKernelGGL(bn_bwd_finalize, dim3((C + 31) / 32), dim3(32 * FG), 0, ST, partial, sums, dgb, C, nch);
#define BA(TT, A) hipLaunchKernelGGL((bn_act_bwd_apply<TT, A>), dim3(nch), dim3(NT), 0, ST, (const TT*)x, (const TT*)res, (const TT*)dz, stats, gamma, beta, sums, (TT*)dx, (TT*)dres, P, C, chunk, training)
    if (dtype) { if (act == 1) BA(bf16, 1); else if (act == 2) BA(bf16, 2); else BA(bf16, 0); }
    else { if (act == 1) BA(float, 1); else if (act == 2) BA(float, 2); else BA(float, 0); }
#undef BA
    DTT_LAUNCH_CHECK();
    return 0;
}

// BatchNorm + activation + 3x3 / stride-2 / pad-1 max pool fused (the lip front end, resnet_video_encoder.py:38-44):
// forward from given statistics (dtt_bn_stats / dtt_bn_frozen_stats) -> pooled y [N][Ho][Wo][C] + argmax bytes; backward
// from the pooled gradient: dx (gradient of the BatchNorm input) and dgb = {dgamma, dbeta}, same scratch as dtt_bn_act_bwd.
extern "C" int dtt_bn_act_pool_fwd(const void* x, const float* stats, const float* gamma, const float* beta, void* y,
                                   unsigned char* idx, int N, int H, int W, int C, int act, int dtype, void* stream) {
    const int ept = dtype ? 8 : 4;
    DTT_CHECK(x && stats && gamma && beta && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % ept == 0 && act >= 0 && act <= 2,
              "dtt_bn_act_pool_fwd: bad arguments (C=%d)", C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / ept);
    const int grid = (int)std::min<int64_t>((total + NT - 1) / NT, 65536);
#define PF(TT, A) hipLaunchKernelGGL((bn_act_pool_fwd_kernel<TT, A>), dim3(grid), dim3(NT), 0, ST, (const TT*)x, stats, gamma, beta, (TT*)y, idx, N, H, W, C, Ho, Wo)
    if (dtype) { if (act == 1) PF(bf16, 1); else if (act == 2) PF(bf16, 2); else PF(bf16, 0); }
    else { if (act == 1) PF(float, 1); else if (act == 2) PF(float, 2); else PF(float, 0); }
#undef PF
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_bn_act_pool_bwd(const void* x, const void* dyp, const unsigned char* idx, const float* stats,
                                   const float* gamma, const float* beta, float* partial, float* sums, float* dgb, void* dx,
                                   int N, int H, int W, int C, int act, int training, int dtype, void* stream) {
    DTT_CHECK(N > 0 && H > 0 && W > 0 && (int64_t)N * H * W < (1LL << 31), "dtt_bn_act_pool_bwd: bad shape");
    const int P = N * H * W;
    if (int rc = bn_check(P, C, dtype, "dtt_bn_act_pool_bwd")) return rc;
    DTT_CHECK(x && dyp && idx && stats && gamma && beta && partial && sums && dgb && dx && act >= 0 && act <= 2,
              "dtt_bn_act_pool_bwd: bad arguments");
    const int chunk = bn_chunk(P), nch = (P + chunk - 1) / chunk;
    const int ept = dtype ? 8 : 4;
    const size_t sm = (size_t)(NT * ept + C) * sizeof(float);
    const PoolGeom pg{idx, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
#define BP(TT, A) hipLaunchKernelGGL((bn_act_bwd_partial<TT, A, true>), dim3(nch), dim3(NT), sm, ST, (const TT*)x, (const TT*)nullptr, (const TT*)dyp, stats, gamma, beta, partial, P, C, chunk, pg)
    if (dtype) { if (act == 1) BP(bf16, 1); else if (act == 2) BP(bf16, 2); else BP(bf16, 0); }
    else { if (act == 1) BP(float, 1); else if (act == 2) BP(float, 2); else BP(float, 0); }
#undef BP
    hipLaunchKernelGGL(bn_bwd_finalize, dim3((C + 31) / 32), dim3(32 * FG), 0, ST, partial, sums, dgb, C, nch);
#define BA(TT, A) hipLaunchKernelGGL((bn_act_bwd_apply<TT, A, true>), dim3(nch), dim3(NT), 0, ST, (const TT*)x, (const TT*)nullptr, (const TT*)dyp, stats, gamma, beta, sums, (TT*)dx, (TT*)nullptr, P, C, chunk, training, pg)
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

// MaxPool 3x3 / stride 2 / padding 1 on channels-last [N][H][W][C]; idx: one byte per output element (window
// position of the maximum), Ho = (H - 1) / 2 + 1, Wo likewise.
extern "C" int dtt_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int dtype,
                                    void* stream) {
    const int ept = dtype ? 8 : 4;
    DTT_CHECK(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % ept == 0, "dtt_maxpool3x3s2_fwd: bad arguments (C=%d)", C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / ept);
    const int grid = (int)std::min<int64_t>((total + NT - 1) / NT, 65536);
    if (dtype) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, (const bf16*)x, (bf16*)y, idx, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(NT), 0, ST, (const float*)x, (float*)y, idx, N, H, W, C, Ho, Wo);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_maxpool3x3s2_bwd(const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C, int dtype,
                                    void* stream) {
    const int ept = dtype ? 8 : 4;
    DTT_CHECK(dy && dx && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % ept == 0, "dtt_maxpool3x3s2_bwd: bad arguments (C=%d)", C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * H * W * (C / ept);
    const int grid = (int)std::min<int64_t>((total + NT - 1) / NT, 65536);
    if (dtype) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, (const bf16*)dy, idx, (bf16*)dx, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(NT), 0, ST, (const float*)dy, idx, (float*)dx, N, H, W, C, Ho, Wo);
    DTT_LAUNCH_CHECK();
    return 0;
}
