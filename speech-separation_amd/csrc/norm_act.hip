// InstanceNorm2d(affine=False)+GELU and GroupNorm kernels on channels-last
// tensors, forward and backward.  HBM-bound: 16-byte vector accesses, two-stage
// deterministic reductions (no float atomics).
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

constexpr int NT = 256;

// ---- block-level helper: reduce per-thread EPT-vectors over the threads that own
// the same channel chunk (tid % cpp), result per channel in `out[C]` (LDS).
template <int EPT>
__device__ __forceinline__ void reduce_channels(const float (&v)[EPT], float* red, float* out, int cpp, int nact,
                                                int C) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (cpp <= 32 && (cpp & (cpp - 1)) == 0) {
        // cpp divides the wave: the lanes of a channel chunk are lane % cpp apart, so the wave folds itself by
        // xor-shuffles and only NT/64 rows of C floats go through LDS (the general walk below costs ~5 us per
        // workgroup at C = 32: 64 dependent LDS reads by 32 threads, twice per kernel)
        float w[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) w[e] = (tid < nact) ? v[e] : 0.f;
        for (int off = cpp; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < EPT; ++e) w[e] += __shfl_xor(w[e], off, 64);
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < cpp)
#pragma unroll
            for (int e = 0; e < EPT; ++e) red[wave * C + lane * EPT + e] = w[e];
        __syncthreads();
        for (int c = tid; c < C; c += NT) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NT / 64; ++k) s += red[k * C + c];
            out[c] = s;
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) red[tid * EPT + e] = (tid < nact) ? v[e] : 0.f;
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        int cq = c / EPT, e = c % EPT;
        float s = 0.f;
        for (int t = cq; t < nact; t += cpp) s += red[t * EPT + e];
        out[c] = s;
    }
    __syncthreads();
}

// =================== InstanceNorm statistics ===================
// y: [B][P][C]; chunk of pixels per block; partial[b][chunk][C][2] = {mean, M2}
template <typename T>
__global__ __launch_bounds__(NT) void in_stats_partial(const T* __restrict__ y, float* __restrict__ partial, int P,
                                                       int C, int chunk) {
    constexpr int EPT = Vec16<T>::N;
    extern __shared__ float sm[];
    float* red = sm;                 // NT*EPT
    float* chv = sm + NT * EPT;      // C
    const int cpp = C / EPT, ppi = NT / cpp, nact = ppi * cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    const int b = blockIdx.y, ck = blockIdx.x;
    const int p0 = ck * chunk, p1 = min(P, p0 + chunk);
    const T* base = y + ((int64_t)b * P) * C + cq * EPT;
    float acc[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
    if (tid < nact)
        for (int p = p0 + pl; p < p1; p += ppi) {
            Vec16<T> v = ld16<T>(base + (int64_t)p * C);
#pragma unroll
            for (int e = 0; e < EPT; ++e) acc[e] += v.get(e);
        }
    reduce_channels<EPT>(acc, red, chv, cpp, nact, C);
    const float inv = 1.f / (float)(p1 - p0);
    float mean[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) { mean[e] = chv[cq * EPT + e] * inv; acc[e] = 0.f; }
    if (tid < nact)
        for (int p = p0 + pl; p < p1; p += ppi) {
            Vec16<T> v = ld16<T>(base + (int64_t)p * C);
#pragma unroll
            for (int e = 0; e < EPT; ++e) { float d = v.get(e) - mean[e]; acc[e] += d * d; }
        }
    float* dst = partial + (((int64_t)b * gridDim.x + ck) * C) * 2;
    if (tid < cpp)
#pragma unroll
        for (int e = 0; e < EPT; ++e) dst[(cq * EPT + e) * 2] = mean[e];
    reduce_channels<EPT>(acc, red, chv, cpp, nact, C);
    for (int c = tid; c < C; c += NT) dst[c * 2 + 1] = chv[c];
}

// merge chunk partials (Chan et al.) -> stats[b][c] = {mean, rstd}: 8 lane groups per channel each walk every
// eighth chunk, then a fixed-order merge of the 8 (one serial walk per channel: 25 us per call)
__global__ __launch_bounds__(256) void in_stats_finalize(const float* __restrict__ partial, float* __restrict__ stats, int P,
                                                         int C, int chunk, int nchunks, float eps) {
    __shared__ float sh[3][8][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C)
        for (int k = g; k < nchunks; k += 8) {
            const float* s = partial + (((int64_t)b * nchunks + k) * C + c) * 2;
            const float nb = (float)(min(P, (k + 1) * chunk) - k * chunk);
            const float d = s[0] - mean, nn = n + nb;
            mean += d * nb / nn;
            m2 += s[1] + d * d * n * nb / nn;
            n = nn;
        }
    sh[0][g][cl] = n; sh[1][g][cl] = mean; sh[2][g][cl] = m2;
    __syncthreads();
    if (g == 0 && c < C) {
        float tn = 0.f, tm = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float nb = sh[0][i][cl];
            if (nb > 0.f) {
                const float d = sh[1][i][cl] - tm, nn = tn + nb;
                tm += d * nb / nn;
                t2 += sh[2][i][cl] + d * d * tn * nb / nn;
                tn = nn;
            }
        }
        stats[((int64_t)b * C + c) * 2] = tm;
        stats[((int64_t)b * C + c) * 2 + 1] = rsqrtf(t2 / tn + eps);
    }
}

// z[b][fz][tz][c] = gelu((y[b][f][t][c]-mean)*rstd), cropping (Fy,Ty) -> (Fz,Tz).
// grid (pixel chunks, B); a thread keeps one 16-byte channel chunk (its statistics in registers)
// and walks pixels with four loads in flight.
template <typename T>
__global__ __launch_bounds__(NT) void in_gelu_fwd(const T* __restrict__ y, const float* __restrict__ stats,
                                                  T* __restrict__ z, int Fy, int Ty, int Fz, int Tz, int C, int chunk) {
    constexpr int EPT = Vec16<T>::N, U = 4;
    const int cpp = C / EPT, ppi = NT / cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    if (pl >= ppi) return;
    const int b = blockIdx.y, Pz = Fz * Tz;
    const int p0 = blockIdx.x * chunk, p1 = min(Pz, p0 + chunk);
    float mean[EPT], rstd[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        mean[e] = stats[((int64_t)b * C + cq * EPT + e) * 2];
        rstd[e] = stats[((int64_t)b * C + cq * EPT + e) * 2 + 1];
    }
    const T* yb = y + (int64_t)b * Fy * Ty * C + cq * EPT;
    T* zb = z + (int64_t)b * Pz * C + cq * EPT;
    const bool same = Ty == Tz;
    for (int p = p0 + pl; p < p1; p += U * ppi) {
        Vec16<T> v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            int src = q;
            if (!same) { const int f = q / Tz; src = f * Ty + (q - f * Tz); }
            v[u] = ld16_pred<T>(yb + (int64_t)src * C, yb, q < p1);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            Vec16<T> o;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < EPT; e += 2)
                    o.set2(e, gelu_t<T>((v[u].get(e) - mean[e]) * rstd[e]), gelu_t<T>((v[u].get(e + 1) - mean[e + 1]) * rstd[e + 1]));
            } else {
#pragma unroll
                for (int e = 0; e < EPT; ++e) o.set(e, gelu_t<T>((v[u].get(e) - mean[e]) * rstd[e]));
            }
            if (q < p1) st16<T>(zb + (int64_t)q * C, o);
        }
    }
}

// backward pass 1: partial[b][chunk][C][2] = {sum dyh, sum dyh*yh} over the FULL (Fy,Ty) plane,
// dz is zero outside the (Fz,Tz) crop.
template <typename T>
__global__ __launch_bounds__(NT) void in_gelu_bwd_partial(const T* __restrict__ y, const T* __restrict__ dz,
                                                          const float* __restrict__ stats,
                                                          float* __restrict__ partial, int Fy, int Ty, int Fz,
                                                          int Tz, int C, int chunk) {
    constexpr int EPT = Vec16<T>::N;
    extern __shared__ float sm[];
    float* red = sm;
    float* chv = sm + NT * EPT;
    const int cpp = C / EPT, ppi = NT / cpp, nact = ppi * cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    const int b = blockIdx.y, ck = blockIdx.x, P = Fy * Ty;
    const int p0 = ck * chunk, p1 = min(P, p0 + chunk);
    float s1[EPT], s2[EPT], mean[EPT], rstd[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        s1[e] = 0.f; s2[e] = 0.f;
        mean[e] = stats[((int64_t)b * C + cq * EPT + e) * 2];
        rstd[e] = stats[((int64_t)b * C + cq * EPT + e) * 2 + 1];
    }
    if (tid < nact) {
        constexpr int U = 4;
        const T* yb = y + (int64_t)b * P * C + cq * EPT;
        const T* gb = dz + (int64_t)b * Fz * Tz * C + cq * EPT;
        for (int p = p0 + pl; p < p1; p += U * ppi) {
            Vec16<T> v[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = p + u * ppi;
                const int f = q / Ty, t = q - f * Ty;
                const bool in = q < p1 && f < Fz && t < Tz;        // dz is zero outside the crop
                v[u] = ld16_pred<T>(yb + (int64_t)q * C, yb, in);
                g[u] = ld16_pred<T>(gb + ((int64_t)f * Tz + t) * C, gb, in);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const float yh = (v[u].get(e) - mean[e]) * rstd[e];
                    const float d = g[u].get(e) * gelu_grad_t<T>(yh);
                    s1[e] += d; s2[e] += d * yh;
                }
        }
    }
    float* dst = partial + (((int64_t)b * gridDim.x + ck) * C) * 2;
    reduce_channels<EPT>(s1, red, chv, cpp, nact, C);
    for (int c = tid; c < C; c += NT) dst[c * 2] = chv[c];
    reduce_channels<EPT>(s2, red, chv, cpp, nact, C);
    for (int c = tid; c < C; c += NT) dst[c * 2 + 1] = chv[c];
}

// sums[b][c][2] = sum over chunks / n.  One workgroup per (32 channels, b): 8 groups of lanes take every
// eighth chunk (a serial walk over all chunks by one lane per channel cost 10-24 us per launch, 48 launches
// per step), then a fixed-order merge through LDS (deterministic).
__global__ __launch_bounds__(256) void in_bwd_finalize(const float* __restrict__ partial, float* __restrict__ sums, int C,
                                                       int nchunks, float inv_n) {
    __shared__ float sh[2][8][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
    float a = 0.f, q = 0.f;
    if (c < C) {
        for (int k = g; k < nchunks; k += 8) {
            const float2 s = *reinterpret_cast<const float2*>(partial + (((int64_t)b * nchunks + k) * C + c) * 2);
            a += s.x; q += s.y;
        }
    }
    sh[0][g][cl] = a; sh[1][g][cl] = q;
    __syncthreads();
    if (g == 0 && c < C) {
        float ta = 0.f, tq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ta += sh[0][i][cl]; tq += sh[1][i][cl]; }
        sums[((int64_t)b * C + c) * 2] = ta * inv_n;
        sums[((int64_t)b * C + c) * 2 + 1] = tq * inv_n;
    }
}

// dy = rstd * (dyh - m1 - yh*m2) over the full plane; same thread layout as in_gelu_fwd
template <typename T>
__global__ __launch_bounds__(NT) void in_gelu_bwd_apply(const T* __restrict__ y, const T* __restrict__ dz,
                                                        const float* __restrict__ stats,
                                                        const float* __restrict__ sums, T* __restrict__ dy,
                                                        int Fy, int Ty, int Fz, int Tz, int C, int chunk) {
    constexpr int EPT = Vec16<T>::N, U = 4;
    const int cpp = C / EPT, ppi = NT / cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    if (pl >= ppi) return;
    const int b = blockIdx.y, P = Fy * Ty;
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    float mean[EPT], rstd[EPT], m1[EPT], m2[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        mean[e] = stats[((int64_t)b * C + cq * EPT + e) * 2];
        rstd[e] = stats[((int64_t)b * C + cq * EPT + e) * 2 + 1];
        m1[e] = sums[((int64_t)b * C + cq * EPT + e) * 2];
        m2[e] = sums[((int64_t)b * C + cq * EPT + e) * 2 + 1];
    }
    const T* yb = y + (int64_t)b * P * C + cq * EPT;
    const T* gb = dz + (int64_t)b * Fz * Tz * C + cq * EPT;
    T* ob = dy + (int64_t)b * P * C + cq * EPT;
    const bool same = Ty == Tz && Fy == Fz;
    for (int p = p0 + pl; p < p1; p += U * ppi) {
        Vec16<T> v[U], g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            bool in = q < p1;
            int64_t gofs = q;
            if (!same) { const int f = q / Ty, t = q - f * Ty; in = in && f < Fz && t < Tz; gofs = (int64_t)f * Tz + t; }
            v[u] = ld16_pred<T>(yb + (int64_t)q * C, yb, q < p1);
            g[u] = ld16_pred<T>(gb + gofs * C, gb, in);           // zero outside the crop
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * ppi;
            float r[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float yh = (v[u].get(e) - mean[e]) * rstd[e];
                const float d = g[u].get(e) * gelu_grad_t<T>(yh);
                r[e] = rstd[e] * (d - m1[e] - yh * m2[e]);
            }
            Vec16<T> o;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < EPT; e += 2) o.set2(e, r[e], r[e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < EPT; ++e) o.set(e, r[e]);
            }
            if (q < p1) st16<T>(ob + (int64_t)q * C, o);
        }
    }
}

// =================== GroupNorm over (gs channels x L rows) ===================
struct GnGeom {
    int nslab, NI, L, C, Cp, gs;
    int64_t OS, IS, RS;            // slab o -> x + (o/NI)*OS + (o%NI)*IS ; row stride RS
};

template <typename T>
__global__ __launch_bounds__(NT) void gn_fwd(const T* __restrict__ x, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, T* __restrict__ xn,
                                             T* __restrict__ xperm, float* __restrict__ stats, GnGeom g,
                                             float eps) {
    constexpr int EPT = Vec16<T>::N;
    __shared__ float red[NT];
    __shared__ float gv[128];
    const int cpp = g.C / EPT, rpi = NT / cpp, nact = rpi * cpp, ng = g.C / g.gs;
    const int tid = threadIdx.x, rl = tid / cpp, cq = tid % cpp, o = blockIdx.x;
    const int grp = (cq * EPT) / g.gs;
    const T* xs = x + (int64_t)(o / g.NI) * g.OS + (int64_t)(o % g.NI) * g.IS + cq * EPT;
    const float inv = 1.f / (float)(g.gs * g.L);
    float mean = 0.f, rstd = 0.f;
    // the thread's rows stay in registers across the three sweeps when they fit (L <= 16 * rpi): one
    // global read instead of three
    constexpr int MAXR = 16;
    const bool keep = g.L <= MAXR * rpi;
    Vec16<T> rows[MAXR];
    if (keep && tid < nact) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int l = rl + i * rpi;
            rows[i] = ld16_pred<T>(xs + (int64_t)l * g.RS, xs, l < g.L);
        }
    }
    const int cpg = g.gs / EPT;                             // 16-byte chunks per group
    for (int pass = 0; pass < 2; ++pass) {
        float a = 0.f;
        if (tid < nact) {
            if (keep) {
#pragma unroll
                for (int i = 0; i < MAXR; ++i)
                    if (rl + i * rpi < g.L) {
#pragma unroll
                        for (int e = 0; e < EPT; ++e) { float d = rows[i].get(e) - mean; a += pass ? d * d : d; }
                    }
            } else {
                for (int l = rl; l < g.L; l += rpi) {
                    Vec16<T> v = ld16<T>(xs + (int64_t)l * g.RS);
#pragma unroll
                    for (int e = 0; e < EPT; ++e) { float d = v.get(e) - mean; a += pass ? d * d : d; }
                }
            }
        }
        __syncthreads();
        red[tid] = (tid < nact) ? a : 0.f;
        __syncthreads();
        if (tid < ng) {                                     // group tid: chunks tid*cpg .. +cpg-1 of every row lane
            float s = 0.f;
            for (int r2 = 0; r2 < rpi; ++r2)
                for (int c2 = 0; c2 < cpg; ++c2) s += red[r2 * cpp + tid * cpg + c2];
            gv[tid] = s * inv;
        }
        __syncthreads();
        if (pass == 0) mean = gv[grp];
        else rstd = rsqrtf(gv[grp] + eps);
    }
    if (tid < nact && rl == 0 && (cq * EPT) % g.gs == 0) {
        stats[((int64_t)o * ng + grp) * 2] = mean;
        stats[((int64_t)o * ng + grp) * 2 + 1] = rstd;
    }
    if (tid < nact) {
        float ga[EPT], be[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            int cp = (cq * EPT + e) % g.Cp;
            ga[e] = gamma[cp]; be[e] = beta[cp];
        }
        auto emit = [&](int l, const Vec16<T>& v) {
            Vec16<T> w;
#pragma unroll
            for (int e = 0; e < EPT; ++e) w.set(e, (v.get(e) - mean) * rstd * ga[e] + be[e]);
            int64_t oo = ((int64_t)o * g.L + l) * g.C + cq * EPT;
            st16<T>(xn + oo, w);
            if (xperm) st16<T>(xperm + oo, v);
        };
        if (keep) {
#pragma unroll
            for (int i = 0; i < MAXR; ++i)
                if (rl + i * rpi < g.L) emit(rl + i * rpi, rows[i]);
        } else {
            for (int l = rl; l < g.L; l += rpi) emit(l, ld16<T>(xs + (int64_t)l * g.RS));
        }
    }
}

// backward: dx (strided like x) = gn_bwd(dxn) + dres ; pgb[o][C][2] = {dgamma, dbeta} partials
template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd(const T* __restrict__ x, const T* __restrict__ dxn,
                                             const T* __restrict__ dres, const float* __restrict__ gamma,
                                             const float* __restrict__ stats, T* __restrict__ dx,
                                             float* __restrict__ pgb, GnGeom g) {
    constexpr int EPT = Vec16<T>::N;
    extern __shared__ float sm[];
    float* red = sm;                   // NT*EPT
    float* chv = sm + NT * EPT;        // C
    __shared__ float r1[NT], r2[NT], g1[128], g2[128];
    const int cpp = g.C / EPT, rpi = NT / cpp, nact = rpi * cpp, ng = g.C / g.gs;
    const int tid = threadIdx.x, rl = tid / cpp, cq = tid % cpp, o = blockIdx.x;
    const int grp = (cq * EPT) / g.gs;
    const int64_t xoff = (int64_t)(o / g.NI) * g.OS + (int64_t)(o % g.NI) * g.IS + cq * EPT;
    float mean = 0.f, rstd = 0.f;
    if (tid < nact) { mean = stats[((int64_t)o * ng + grp) * 2]; rstd = stats[((int64_t)o * ng + grp) * 2 + 1]; }
    float ga[EPT], dg[EPT], db[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) { ga[e] = (tid < nact) ? gamma[(cq * EPT + e) % g.Cp] : 0.f; dg[e] = 0.f; db[e] = 0.f; }
    float a1 = 0.f, a2 = 0.f;
    if (tid < nact)
        for (int l = rl; l < g.L; l += rpi) {
            Vec16<T> v = ld16<T>(x + xoff + (int64_t)l * g.RS);
            Vec16<T> d = ld16<T>(dxn + ((int64_t)o * g.L + l) * g.C + cq * EPT);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                float xh = (v.get(e) - mean) * rstd, dd = d.get(e), gg = dd * ga[e];
                a1 += gg; a2 += gg * xh;
                dg[e] += dd * xh; db[e] += dd;
            }
        }
    r1[tid] = (tid < nact) ? a1 : 0.f;
    r2[tid] = (tid < nact) ? a2 : 0.f;
    __syncthreads();
    if (tid < ng) {
        float s1 = 0.f, s2 = 0.f;
        const int cpg = g.gs / EPT;
        for (int r2i = 0; r2i < rpi; ++r2i)
            for (int c2 = 0; c2 < cpg; ++c2) { const int t = r2i * cpp + tid * cpg + c2; s1 += r1[t]; s2 += r2[t]; }
        const float inv = 1.f / (float)(g.gs * g.L);
        g1[tid] = s1 * inv; g2[tid] = s2 * inv;
    }
    __syncthreads();
    const float m1 = g1[grp], m2 = g2[grp];
    if (tid < nact)
        for (int l = rl; l < g.L; l += rpi) {
            Vec16<T> v = ld16<T>(x + xoff + (int64_t)l * g.RS);
            int64_t oo = ((int64_t)o * g.L + l) * g.C + cq * EPT;
            Vec16<T> d = ld16<T>(dxn + oo);
            Vec16<T> w;
            if (dres) {
                Vec16<T> r = ld16<T>(dres + oo);
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    float xh = (v.get(e) - mean) * rstd;
                    w.set(e, rstd * (d.get(e) * ga[e] - m1 - xh * m2) + r.get(e));
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    float xh = (v.get(e) - mean) * rstd;
                    w.set(e, rstd * (d.get(e) * ga[e] - m1 - xh * m2));
                }
            }
            st16<T>(dx + xoff + (int64_t)l * g.RS, w);
        }
    reduce_channels<EPT>(dg, red, chv, cpp, nact, g.C);
    for (int c = tid; c < g.C; c += NT) pgb[((int64_t)o * g.C + c) * 2] = chv[c];
    reduce_channels<EPT>(db, red, chv, cpp, nact, g.C);
    for (int c = tid; c < g.C; c += NT) pgb[((int64_t)o * g.C + c) * 2 + 1] = chv[c];
}

// =================== column sums: out[chunk][w] = sum_{r in chunk} in[r][w] ===================
template <typename T>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t R,
                                                    int W, int64_t ld, int64_t rows_per) {
    int w = blockIdx.x * NT + threadIdx.x;
    if (w >= W) return;
    int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += ldf(in + r * ld + w);
    out[(int64_t)blockIdx.y * W + w] = s;
}

// Vectorised variant: a thread owns one 16-byte column chunk and every RL-th row of the block's row
// range (4 loads in flight), row lanes are folded through LDS.  Needs W, ld multiples of the chunk.
template <typename T>
__global__ __launch_bounds__(NT) void colsum_vec_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t R,
                                                        int W, int64_t ld, int64_t rows_per, int CWb) {
    constexpr int E = Vec16<T>::N;
    __shared__ float red[NT * E];
    const int RL = NT / CWb;
    const int cl = threadIdx.x % CWb, rl = threadIdx.x / CWb;
    const int chunk = blockIdx.x * CWb + cl;
    const bool okc = chunk * E < W && rl < RL;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float s[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    const T* base = in + (okc ? (int64_t)chunk * E : 0);
    for (int64_t r = r0 + rl; r < r1; r += 4 * RL) {
        Vec16<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t rr = r + (int64_t)u * RL;
            v[u] = ld16_pred<T>(base + rr * ld, base + r0 * ld, okc && rr < r1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) s[e] += v[u].get(e);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[threadIdx.x * E + e] = s[e];
    __syncthreads();
    // thread t < CWb*E sums column (t) of the block over the row lanes
    for (int c = threadIdx.x; c < CWb * E; c += NT) {
        const int gw = blockIdx.x * CWb * E + c;
        if (gw >= W) continue;
        float a = 0.f;
        for (int q = 0; q < RL; ++q) a += red[(q * CWb + c / E) * E + (c % E)];
        out[(int64_t)blockIdx.y * W + gw] = a;
    }
}

// Weighted column sums: out[chunk][m][w] = sum_r a[r][m] * in[r][w], m < MW <= 8 (the weight gradient
// of a 1x1 conv with a handful of outputs, e.g. the mask head: dW[2S][C] = dmask^T x).  Same thread
// layout as colsum_vec_kernel: HBM-bound, one pass over `in`.
template <typename T>
__global__ __launch_bounds__(NT) void wcolsum_kernel(const T* __restrict__ a, const T* __restrict__ in,
                                                     float* __restrict__ out, int64_t R, int W, int MW,
                                                     int64_t rows_per, int CWb) {
    constexpr int E = Vec16<T>::N;
    __shared__ float red[NT * E];
    const int RL = NT / CWb;
    const int cl = threadIdx.x % CWb, rl = threadIdx.x / CWb;
    const int chunk = blockIdx.x * CWb + cl;
    const bool okc = chunk * E < W && rl < RL;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float s[8][E];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int e = 0; e < E; ++e) s[m][e] = 0.f;
    const T* base = in + (okc ? (int64_t)chunk * E : 0);
    for (int64_t r = r0 + rl; r < r1; r += 2 * RL) {
        Vec16<T> v[2];
        float av[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t rr = r + (int64_t)u * RL;
            const bool ok = okc && rr < r1;
            v[u] = ld16_pred<T>(base + rr * W, base, ok);
#pragma unroll
            for (int m = 0; m < 8; ++m) av[u][m] = (ok && m < MW) ? ldf(a + (ok ? rr : r0) * MW + (m < MW ? m : 0)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int e = 0; e < E; ++e) s[m][e] += av[u][m] * v[u].get(e);
    }
    for (int m = 0; m < MW; ++m) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) red[threadIdx.x * E + e] = s[m][e];
        __syncthreads();
        for (int c = threadIdx.x; c < CWb * E; c += NT) {
            const int gw = blockIdx.x * CWb * E + c;
            if (gw >= W) continue;
            float acc = 0.f;
            for (int q = 0; q < RL; ++q) acc += red[(q * CWb + c / E) * E + (c % E)];
            out[((int64_t)blockIdx.y * MW + m) * W + gw] = acc;
        }
    }
}

int ept_of(int dtype) { return dtype ? 8 : 4; }

// pixels per workgroup for the elementwise IN kernels: ~16 workgroups per CU of up to 8 unrolled sweeps of the block
// each, but never fewer than ~4 workgroups per CU: at 512 channels a sweep is 16 pixels, and 8 of them per
// workgroup left the 65 x 16 level with 144 workgroups for 256 CUs (2.1 TB/s; level 0 with 512: 4.2 TB/s)
int ew_chunk(int B, int P, int cpp) {
    const int ppi = NT / cpp, want = max(1, 4096 / max(1, B)), sweep = ppi * 4;
    int ch = (P + want - 1) / want;
    int s = 8;
    while (s > 1 && (int64_t)B * ((P + sweep * s - 1) / (sweep * s)) < 1024) s >>= 1;
    ch = max(ch, sweep * s);
    return (ch + sweep - 1) / sweep * sweep;
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int dtt_in_stats(const void* y, float* partial, float* stats, int B, int P, int C, int chunk, float eps,
                            int dtype, void* stream) {
    const int ept = ept_of(dtype);
    DTT_CHECK(C % ept == 0 && C / ept <= NT && C <= 2048, "dtt_in_stats: unsupported C=%d", C);
    DTT_CHECK(chunk > 0 && B > 0 && P > 0, "dtt_in_stats: bad shape");
    const int nch = (P + chunk - 1) / chunk;
    size_t sm = (NT * ept + C) * sizeof(float);
    if (dtype) hipLaunchKernelGGL(in_stats_partial<bf16>, dim3(nch, B), dim3(NT), sm, ST, (const bf16*)y, partial, P, C, chunk);
    else hipLaunchKernelGGL(in_stats_partial<float>, dim3(nch, B), dim3(NT), sm, ST, (const float*)y, partial, P, C, chunk);
    hipLaunchKernelGGL(in_stats_finalize, dim3((C + 31) / 32, B), dim3(256), 0, ST, partial, stats, P, C, chunk, nch, eps);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_in_gelu_fwd(const void* y, const float* stats, void* z, int B, int Fy, int Ty, int Fz, int Tz,
                               int C, int dtype, void* stream) {
    const int ept = ept_of(dtype);
    DTT_CHECK(C % ept == 0 && Fz <= Fy && Tz <= Ty, "dtt_in_gelu_fwd: bad shape");
    DTT_CHECK(C / ept <= NT, "dtt_in_gelu_fwd: C=%d too wide", C);
    const int Pz = Fz * Tz, ech = ew_chunk(B, Pz, C / ept);
    dim3 grid((Pz + ech - 1) / ech, B);
    if (dtype) hipLaunchKernelGGL(in_gelu_fwd<bf16>, grid, dim3(NT), 0, ST, (const bf16*)y, stats, (bf16*)z, Fy, Ty, Fz, Tz, C, ech);
    else hipLaunchKernelGGL(in_gelu_fwd<float>, grid, dim3(NT), 0, ST, (const float*)y, stats, (float*)z, Fy, Ty, Fz, Tz, C, ech);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_in_gelu_bwd(const void* y, const void* dz, const float* stats, float* partial, float* sums,
                               void* dy, int B, int Fy, int Ty, int Fz, int Tz, int C, int chunk, int dtype,
                               void* stream) {
    const int ept = ept_of(dtype);
    DTT_CHECK(C % ept == 0 && C / ept <= NT && C <= 2048 && Fz <= Fy && Tz <= Ty && chunk > 0,
              "dtt_in_gelu_bwd: bad shape");
    const int P = Fy * Ty, nch = (P + chunk - 1) / chunk;
    size_t sm = (NT * ept + C) * sizeof(float);
    if (dtype) hipLaunchKernelGGL(in_gelu_bwd_partial<bf16>, dim3(nch, B), dim3(NT), sm, ST, (const bf16*)y, (const bf16*)dz, stats, partial, Fy, Ty, Fz, Tz, C, chunk);
    else hipLaunchKernelGGL(in_gelu_bwd_partial<float>, dim3(nch, B), dim3(NT), sm, ST, (const float*)y, (const float*)dz, stats, partial, Fy, Ty, Fz, Tz, C, chunk);
    hipLaunchKernelGGL(in_bwd_finalize, dim3((C + 31) / 32, B), dim3(256), 0, ST, partial, sums, C, nch, 1.f / (float)P);
    const int ech = ew_chunk(B, P, C / ept);
    dim3 grid((P + ech - 1) / ech, B);
    if (dtype) hipLaunchKernelGGL(in_gelu_bwd_apply<bf16>, grid, dim3(NT), 0, ST, (const bf16*)y, (const bf16*)dz, stats, sums, (bf16*)dy, Fy, Ty, Fz, Tz, C, ech);
    else hipLaunchKernelGGL(in_gelu_bwd_apply<float>, grid, dim3(NT), 0, ST, (const float*)y, (const float*)dz, stats, sums, (float*)dy, Fy, Ty, Fz, Tz, C, ech);
    DTT_LAUNCH_CHECK();
    return 0;
}

static int gn_check(const GnGeom& g, int ept) {
    DTT_CHECK(g.C % ept == 0 && g.C / ept <= NT && g.gs % ept == 0 && g.C % g.gs == 0 && g.C / g.gs <= 128 &&
                  g.Cp % g.gs == 0 && g.C % g.Cp == 0 && g.nslab > 0 && g.L > 0 && g.NI > 0 && g.C <= 2048,
              "groupnorm: unsupported geometry C=%d Cp=%d gs=%d", g.C, g.Cp, g.gs);
    return 0;
}

extern "C" int dtt_gn_fwd(const void* x, const float* gamma, const float* beta, void* xn, void* xperm, float* stats,
                          int nslab, int NI, int L, int C, int Cp, int gs, int64_t OS, int64_t IS, int64_t RS,
                          float eps, int dtype, void* stream) {
    GnGeom g{nslab, NI, L, C, Cp, gs, OS, IS, RS};
    if (gn_check(g, ept_of(dtype))) return -1;
    if (dtype) hipLaunchKernelGGL(gn_fwd<bf16>, dim3(nslab), dim3(NT), 0, ST, (const bf16*)x, gamma, beta, (bf16*)xn, (bf16*)xperm, stats, g, eps);
    else hipLaunchKernelGGL(gn_fwd<float>, dim3(nslab), dim3(NT), 0, ST, (const float*)x, gamma, beta, (float*)xn, (float*)xperm, stats, g, eps);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_gn_bwd(const void* x, const void* dxn, const void* dres, const float* gamma, const float* stats,
                          void* dx, float* pgb, int nslab, int NI, int L, int C, int Cp, int gs, int64_t OS,
                          int64_t IS, int64_t RS, int dtype, void* stream) {
    GnGeom g{nslab, NI, L, C, Cp, gs, OS, IS, RS};
    const int ept = ept_of(dtype);
    if (gn_check(g, ept)) return -1;
    size_t sm = (NT * ept + C) * sizeof(float);
    if (dtype) hipLaunchKernelGGL(gn_bwd<bf16>, dim3(nslab), dim3(NT), sm, ST, (const bf16*)x, (const bf16*)dxn, (const bf16*)dres, gamma, stats, (bf16*)dx, pgb, g);
    else hipLaunchKernelGGL(gn_bwd<float>, dim3(nslab), dim3(NT), sm, ST, (const float*)x, (const float*)dxn, (const float*)dres, gamma, stats, (float*)dx, pgb, g);
    DTT_LAUNCH_CHECK();
    return 0;
}

// out[nchunks][W]; call again on `out` with nchunks=1 to finish (deterministic two-stage sum)
extern "C" int dtt_colsum(const void* in, float* out, int64_t R, int W, int64_t ld, int nchunks, int dtype,
                          void* stream) {
    DTT_CHECK(R > 0 && W > 0 && nchunks > 0 && nchunks <= 65535, "dtt_colsum: bad shape");
    int64_t per = (R + nchunks - 1) / nchunks;
    const int E = ept_of(dtype);
    if (W % E == 0 && ld % E == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && per >= 8) {
        int cw = W / E, CWb = 1;
        while (CWb < cw && CWb < NT) CWb <<= 1;              // power of two <= NT
        dim3 vgrid((cw + CWb - 1) / CWb, nchunks);
        if (dtype) hipLaunchKernelGGL(colsum_vec_kernel<bf16>, vgrid, dim3(NT), 0, ST, (const bf16*)in, out, R, W, ld, per, CWb);
        else hipLaunchKernelGGL(colsum_vec_kernel<float>, vgrid, dim3(NT), 0, ST, (const float*)in, out, R, W, ld, per, CWb);
        DTT_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((W + NT - 1) / NT, nchunks);
    if (dtype) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(NT), 0, ST, (const bf16*)in, out, R, W, ld, per);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(NT), 0, ST, (const float*)in, out, R, W, ld, per);
    DTT_LAUNCH_CHECK();
    return 0;
}

// out[nchunks][MW][W] = per-chunk partial sums of a[r][m] * in[r][w]; finish with dtt_colsum over the
// chunks (W' = MW*W).  a [R][MW], in [R][W] (ld = W), same dtype; W a multiple of the 16-byte chunk.
extern "C" int dtt_wcolsum(const void* a, const void* in, float* out, int64_t R, int W, int MW, int nchunks, int dtype,
                           void* stream) {
    const int E = ept_of(dtype);
    DTT_CHECK(R > 0 && W > 0 && MW > 0 && MW <= 8 && nchunks > 0 && nchunks <= 65535 && W % E == 0,
              "dtt_wcolsum: bad shape W=%d MW=%d", W, MW);
    DTT_CHECK((reinterpret_cast<uintptr_t>(in) & 15) == 0, "dtt_wcolsum: unaligned input");
    const int64_t per = (R + nchunks - 1) / nchunks;
    int cw = W / E, CWb = 1;
    while (CWb < cw && CWb < NT) CWb <<= 1;
    dim3 grid((cw + CWb - 1) / CWb, nchunks);
    if (dtype) hipLaunchKernelGGL(wcolsum_kernel<bf16>, grid, dim3(NT), 0, ST, (const bf16*)a, (const bf16*)in, out, R, W, MW, per, CWb);
    else hipLaunchKernelGGL(wcolsum_kernel<float>, grid, dim3(NT), 0, ST, (const float*)a, (const float*)in, out, R, W, MW, per, CWb);
    DTT_LAUNCH_CHECK();
    return 0;
}

