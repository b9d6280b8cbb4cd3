// Small HBM-bound elementwise kernels on channels-last tensors: residual add,
// multiplicative U-Net skip (decoder.py:43), FiLM apply (block_fusion.py:30-37),
// frequency-mean tokens and broadcast-concat of the cross-attention fusion
// (cross_attention_positional.py:55,77-78).
#include <algorithm>
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {
constexpr int NT = 256;

template <typename T, int OP>
__global__ __launch_bounds__(NT) void binary_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                    T* __restrict__ out, int64_t nvec) {
    constexpr int EPT = Vec16<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * NT) {
        Vec16<T> x = ld16<T>(a + i * EPT), y = ld16<T>(b + i * EPT), o;
#pragma unroll
        for (int e = 0; e < EPT; ++e) o.set(e, OP == 0 ? x.get(e) + y.get(e) : x.get(e) * y.get(e));
        st16<T>(out + i * EPT, o);
    }
}

// FiLM: y[b,f,t,c] = (1 + gc[b,t,c] + gf[b,t,f]) * x + bc[b,t,c] + bf[b,t,f]
// gb_c: [B][T][2C] = (gamma_c | beta_c), gb_f: [B][T][2F] = (gamma_f | beta_f), both f32
template <typename T>
__global__ __launch_bounds__(NT) void film_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gbc,
                                                      const float* __restrict__ gbf, T* __restrict__ y, int B, int F,
                                                      int Tn, int C) {
    constexpr int EPT = Vec16<T>::N;
    const int cpp = C / EPT;
    const int64_t total = (int64_t)B * F * Tn * cpp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int cq = (int)(i % cpp);
        int64_t pix = i / cpp;
        int t = (int)(pix % Tn);
        int64_t r = pix / Tn;
        int f = (int)(r % F), b = (int)(r / F);
        const float* pc = gbc + ((int64_t)b * Tn + t) * 2 * C + cq * EPT;
        const float* pf = gbf + ((int64_t)b * Tn + t) * 2 * F;
        float gf = pf[f], bf = pf[F + f];
        Vec16<T> v = ld16<T>(x + pix * C + cq * EPT), o;
#pragma unroll
        for (int e = 0; e < EPT; ++e) o.set(e, (1.f + pc[e] + gf) * v.get(e) + pc[C + e] + bf);
        st16<T>(y + pix * C + cq * EPT, o);
    }
}

// FiLM backward: dx = (1+gamma) * dy ; per (b,t): dgc[c] = sum_f dy*x, dbc[c] = sum_f dy,
// dgf[f] = sum_c dy*x, dbf[f] = sum_c dy.  One block per (b,t).
template <typename T>
__global__ __launch_bounds__(NT) void film_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                      const float* __restrict__ gbc, const float* __restrict__ gbf,
                                                      T* __restrict__ dx, float* __restrict__ dgbc,
                                                      float* __restrict__ dgbf, int B, int F, int Tn, int C) {
    constexpr int EPT = Vec16<T>::N;
    extern __shared__ float sm[];          // 2*NT*EPT
    const int cpp = C / EPT, fpi = NT / cpp, nact = fpi * cpp;
    const int tid = threadIdx.x, fl = tid / cpp, cq = tid % cpp;
    const int b = blockIdx.x / Tn, t = blockIdx.x % Tn;
    const float* pc = gbc + ((int64_t)b * Tn + t) * 2 * C + cq * EPT;
    const float* pf = gbf + ((int64_t)b * Tn + t) * 2 * F;
    float* dpf = dgbf + ((int64_t)b * Tn + t) * 2 * F;
    float ag[EPT], ab[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) { ag[e] = 0.f; ab[e] = 0.f; }
    for (int f0 = 0; f0 < F; f0 += fpi) {
        const int f = f0 + fl;
        float sg = 0.f, sb = 0.f;
        if (tid < nact && f < F) {
            int64_t o = (((int64_t)b * F + f) * Tn + t) * C + cq * EPT;
            Vec16<T> v = ld16<T>(x + o), g = ld16<T>(dy + o), w;
            float gf = pf[f];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                float gg = g.get(e), xv = v.get(e);
                w.set(e, (1.f + pc[e] + gf) * gg);
                ag[e] += gg * xv; ab[e] += gg;
                sg += gg * xv; sb += gg;
            }
            st16<T>(dx + o, w);
        }
        // reduce sg/sb over the cpp threads of one f (consecutive tids)
        __syncthreads();
        sm[tid] = sg; sm[NT + tid] = sb;
        __syncthreads();
        if (tid < fpi && f0 + tid < F) {
            float s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < cpp; ++k) { s1 += sm[tid * cpp + k]; s2 += sm[NT + tid * cpp + k]; }
            dpf[f0 + tid] = s1; dpf[F + f0 + tid] = s2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) { sm[tid * EPT + e] = (tid < nact) ? ag[e] : 0.f; sm[NT * EPT + tid * EPT + e] = (tid < nact) ? ab[e] : 0.f; }
    __syncthreads();
    float* dpc = dgbc + ((int64_t)b * Tn + t) * 2 * C;
    for (int c = tid; c < C; c += NT) {
        int q = c / EPT, e = c % EPT;
        float s1 = 0.f, s2 = 0.f;
        for (int k = q; k < nact; k += cpp) { s1 += sm[k * EPT + e]; s2 += sm[NT * EPT + k * EPT + e]; }
        dpc[c] = s1; dpc[C + c] = s2;
    }
}

// tokens[b][t][c] = mean_f x[b,f,t,c]   (f32 out)
template <typename T>
__global__ __launch_bounds__(NT) void fmean_kernel(const T* __restrict__ x, float* __restrict__ tok, int B, int F,
                                                   int Tn, int C) {
    const int64_t total = (int64_t)B * Tn * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int c = (int)(i % C);
        int64_t r = i / C;
        int t = (int)(r % Tn), b = (int)(r / Tn);
        float s = 0.f;
        for (int f = 0; f < F; ++f) s += ldf(x + (((int64_t)b * F + f) * Tn + t) * C + c);
        tok[i] = s / (float)F;
    }
}

// out[b,f,t,0:C] = x ; out[b,f,t,C:2C] = enh[b,t,c] (f32 tokens)
template <typename T>
__global__ __launch_bounds__(NT) void bcat_fwd_kernel(const T* __restrict__ x, const float* __restrict__ enh,
                                                      T* __restrict__ out, int B, int F, int Tn, int C) {
    constexpr int EPT = Vec16<T>::N;
    const int cpp = 2 * C / EPT, half = C / EPT;
    const int64_t total = (int64_t)B * F * Tn * cpp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int cq = (int)(i % cpp);
        int64_t pix = i / cpp;
        Vec16<T> v;
        if (cq < half) v = ld16<T>(x + pix * C + cq * EPT);
        else {
            int t = (int)(pix % Tn);
            int b = (int)(pix / ((int64_t)F * Tn));
            const float* e0 = enh + ((int64_t)b * Tn + t) * C + (cq - half) * EPT;
#pragma unroll
            for (int e = 0; e < EPT; ++e) v.set(e, e0[e]);
        }
        st16<T>(out + pix * 2 * C + cq * EPT, v);
    }
}

// dx[b,f,t,c] = dout[...,c] + dtok[b,t,c]/F ; denh[b,t,c] = sum_f dout[...,C+c]   (dx or denh may be NULL: that half
// of the work is skipped -- the two halves run as two calls when dtok only exists after denh has been consumed)
template <typename T>
__global__ __launch_bounds__(NT) void bcat_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ dtok,
                                                      T* __restrict__ dx, float* __restrict__ denh, int B, int F,
                                                      int Tn, int C) {
    const int64_t total = (int64_t)B * Tn * C;
    const float invF = 1.f / (float)F;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int c = (int)(i % C);
        int64_t r = i / C;
        int t = (int)(r % Tn), b = (int)(r / Tn);
        float s = 0.f, dt = dtok ? dtok[i] * invF : 0.f;
        for (int f = 0; f < F; ++f) {
            int64_t pix = ((int64_t)b * F + f) * Tn + t;
            if (denh) s += ldf(dout + pix * 2 * C + C + c);
            if (dx) stf(dx + pix * C + c, ldf(dout + pix * 2 * C + c) + dt);
        }
        if (denh) denh[i] = s;
    }
}

// Weight of a stride-1 conv3x3 for its data-gradient conv: out[ci][tap][co] = w[co][ci][8 - tap]
// (taps rotated by 180 degrees, channels swapped, cast), one launch instead of flip + permuted copy.
template <typename T>
__global__ __launch_bounds__(NT) void pack_conv3_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int Co,
                                                              int Ci) {
    const int total = Ci * 9 * Co;
    for (int i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
        const int co = i % Co, r = i / Co, tap = r % 9, ci = r / 9;
        stf(out + i, w[((int64_t)co * Ci + ci) * 9 + (8 - tap)]);
    }
}

int ept_of(int dtype) { return dtype ? 8 : 4; }
int grid_for(int64_t total) { int64_t g = (total + NT - 1) / NT; return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192); }
// Lip front end: frames t - kt/2 .. t + kt/2 of a clip stacked as the channels of frame t (zero frames beyond the
// clip's ends, zero channels up to 8), channels-last, with padl / padr zero columns on either side of every row:
// out[(n T + t)][h][padl + W + padr][8] from x[n][T][H][W] fp32.  One lane = one output pixel (16 bytes of bf16 / 32
// of fp32); lanes run along w, so every one of the kt reads is coalesced.
template <typename T>
__global__ __launch_bounds__(256) void stack_frames_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int Tn,
                                                           int H, int W, int padl, int padr, int kt) {
    const int Wp = padl + W + padr;
    const int64_t total = (int64_t)N * Tn * H * Wp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int wp = (int)(i % Wp), w = wp - padl;
        int64_t r = i / Wp;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % Tn);
        const int64_t n = r / Tn;
        const bool inside = w >= 0 && w < W;
        const int hw = h * W + (inside ? w : 0), HW = H * W;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ts = t + c - kt / 2;
            v[c] = (inside && c < kt && ts >= 0 && ts < Tn) ? x[(n * Tn + ts) * HW + hw] : 0.f;
        }
        if constexpr (sizeof(T) == 2) {
            Vec16<bf16> o;
#pragma unroll
            for (int c = 0; c < 8; c += 2) o.set2(c, v[c], v[c + 1]);
            st16<bf16>(reinterpret_cast<bf16*>(out) + i * 8, o);
        } else {
            float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + i * 8);
            d[0] = make_float4(v[0], v[1], v[2], v[3]);
            d[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// Linear interpolation over time, F.interpolate(mode="linear", align_corners=False) of ATen (block_fusion.py:18-19,
// temporal_alignment.py:19-22): source position (t + 0.5) * Tin / Tout - 0.5 clamped at 0, taps i0 = floor, i1 =
// min(i0 + 1, Tin - 1).  x [B][Tin][D] -> out [B][Tout][D], fp32.  Backward: every input frame gathers from the output
// frames that tap it (the taps are monotone in t; the loop over t is uniform across the workgroup): no atomics.
__device__ __forceinline__ void lerp_taps(int t, float scale, int Tin, int& i0, int& i1, float& l1) {
    float src = ((float)t + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > Tin - 1) i0 = Tin - 1;
    i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    l1 = src - (float)i0;
}
__global__ __launch_bounds__(NT) void lerp_time_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int Tin,
                                                           int Tout, int D, float scale, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    const int64_t r = i / D;
    const int t = (int)(r % Tout);
    const int64_t b = r / Tout;
    int i0, i1;
    float l1;
    lerp_taps(t, scale, Tin, i0, i1, l1);
    const float* xb = x + b * Tin * D + d;
    out[i] = (1.f - l1) * xb[(int64_t)i0 * D] + l1 * xb[(int64_t)i1 * D];
}
__global__ __launch_bounds__(NT) void lerp_time_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int Tin,
                                                           int Tout, int D, float scale, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    const int64_t r = i / D;
    const int ti = (int)(r % Tin);
    const int64_t b = r / Tin;
    const float* gb = dout + b * Tout * D + d;
    float acc = 0.f;
    for (int t = 0; t < Tout; ++t) {
        int i0, i1;
        float l1;
        lerp_taps(t, scale, Tin, i0, i1, l1);
        if (i0 > ti) break;                                  // taps are non-decreasing in t
        if (i1 < ti) continue;
        const float g = gb[(int64_t)t * D];
        if (i0 == ti) acc += (1.f - l1) * g;
        if (i1 == ti) acc += l1 * g;
    }
    dx[i] = acc;
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int dtt_binary(int op, const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    const int ept = ept_of(dtype);
    DTT_CHECK(n > 0 && n % ept == 0 && (op == 0 || op == 1), "dtt_binary: n=%lld must be a multiple of %d", (long long)n, ept);
    int64_t nv = n / ept;
    int grid = grid_for(nv);
#define BK(TT, OP) hipLaunchKernelGGL((binary_kernel<TT, OP>), dim3(grid), dim3(NT), 0, ST, (const TT*)a, (const TT*)b, (TT*)out, nv)
    if (dtype) { if (op) BK(bf16, 1); else BK(bf16, 0); }
    else { if (op) BK(float, 1); else BK(float, 0); }
#undef BK
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_stack_frames(const float* x, void* out, int N, int T, int H, int W, int kt, int pad_left, int pad_right,
                                int dtype, void* stream) {
    DTT_CHECK(x && out && N > 0 && T > 0 && H > 0 && W > 0 && kt >= 1 && kt <= 8 && (kt & 1) && pad_left >= 0 && pad_right >= 0,
              "dtt_stack_frames: bad shape (kt=%d)", kt);
    const int64_t total = (int64_t)N * T * H * (pad_left + W + pad_right);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
    if (dtype) hipLaunchKernelGGL(stack_frames_kernel<bf16>, dim3(grid), dim3(256), 0, ST, x, (bf16*)out, N, T, H, W, pad_left, pad_right, kt);
    else hipLaunchKernelGGL(stack_frames_kernel<float>, dim3(grid), dim3(256), 0, ST, x, (float*)out, N, T, H, W, pad_left, pad_right, kt);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_pack_conv3_dgrad(const float* w, void* out, int Co, int Ci, int dtype, void* stream) {
    DTT_CHECK(w && out && Co > 0 && Ci > 0 && (int64_t)Co * Ci * 9 < (1LL << 31), "dtt_pack_conv3_dgrad: bad shape Co=%d Ci=%d", Co, Ci);
    const int grid = grid_for((int64_t)Co * Ci * 9);
    if (dtype) hipLaunchKernelGGL(pack_conv3_dgrad_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, w, (bf16*)out, Co, Ci);
    else hipLaunchKernelGGL(pack_conv3_dgrad_kernel<float>, dim3(grid), dim3(NT), 0, ST, w, (float*)out, Co, Ci);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_film_fwd(const void* x, const float* gbc, const float* gbf, void* y, int B, int F, int T, int C,
                            int dtype, void* stream) {
    DTT_CHECK(C % ept_of(dtype) == 0, "dtt_film_fwd: C=%d", C);
    int grid = grid_for((int64_t)B * F * T * (C / ept_of(dtype)));
    if (dtype) hipLaunchKernelGGL(film_fwd_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, (const bf16*)x, gbc, gbf, (bf16*)y, B, F, T, C);
    else hipLaunchKernelGGL(film_fwd_kernel<float>, dim3(grid), dim3(NT), 0, ST, (const float*)x, gbc, gbf, (float*)y, B, F, T, C);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_film_bwd(const void* x, const void* dy, const float* gbc, const float* gbf, void* dx, float* dgbc,
                            float* dgbf, int B, int F, int T, int C, int dtype, void* stream) {
    const int ept = ept_of(dtype);
    DTT_CHECK(C % ept == 0 && C / ept <= NT, "dtt_film_bwd: C=%d", C);
    size_t sm = (size_t)2 * NT * ept * sizeof(float);
    if (dtype) hipLaunchKernelGGL(film_bwd_kernel<bf16>, dim3(B * T), dim3(NT), sm, ST, (const bf16*)x, (const bf16*)dy, gbc, gbf, (bf16*)dx, dgbc, dgbf, B, F, T, C);
    else hipLaunchKernelGGL(film_bwd_kernel<float>, dim3(B * T), dim3(NT), sm, ST, (const float*)x, (const float*)dy, gbc, gbf, (float*)dx, dgbc, dgbf, B, F, T, C);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_fmean(const void* x, float* tok, int B, int F, int T, int C, int dtype, void* stream) {
    int grid = grid_for((int64_t)B * T * C);
    if (dtype) hipLaunchKernelGGL(fmean_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, (const bf16*)x, tok, B, F, T, C);
    else hipLaunchKernelGGL(fmean_kernel<float>, dim3(grid), dim3(NT), 0, ST, (const float*)x, tok, B, F, T, C);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_bcat_fwd(const void* x, const float* enh, void* out, int B, int F, int T, int C, int dtype,
                            void* stream) {
    DTT_CHECK(C % ept_of(dtype) == 0, "dtt_bcat_fwd: C=%d", C);
    int grid = grid_for((int64_t)B * F * T * (2 * C / ept_of(dtype)));
    if (dtype) hipLaunchKernelGGL(bcat_fwd_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, (const bf16*)x, enh, (bf16*)out, B, F, T, C);
    else hipLaunchKernelGGL(bcat_fwd_kernel<float>, dim3(grid), dim3(NT), 0, ST, (const float*)x, enh, (float*)out, B, F, T, C);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_bcat_bwd(const void* dout, const float* dtok, void* dx, float* denh, int B, int F, int T, int C,
                            int dtype, void* stream) {
    int grid = grid_for((int64_t)B * T * C);
    if (dtype) hipLaunchKernelGGL(bcat_bwd_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, (const bf16*)dout, dtok, (bf16*)dx, denh, B, F, T, C);
    else hipLaunchKernelGGL(bcat_bwd_kernel<float>, dim3(grid), dim3(NT), 0, ST, (const float*)dout, dtok, (float*)dx, denh, B, F, T, C);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_lerp_time_fwd(const float* x, float* out, int B, int Tin, int Tout, int D, void* stream) {
    DTT_CHECK(x && out && B > 0 && Tin > 0 && Tout > 0 && D > 0, "dtt_lerp_time_fwd: bad arguments");
    const int64_t total = (int64_t)B * Tout * D;
    DTT_CHECK((total + NT - 1) / NT < (1LL << 31), "dtt_lerp_time_fwd: too large");
    hipLaunchKernelGGL(lerp_time_fwd_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0,
                       static_cast<hipStream_t>(stream), x, out, Tin, Tout, D, (float)Tin / (float)Tout, total);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_lerp_time_bwd(const float* dout, float* dx, int B, int Tin, int Tout, int D, void* stream) {
    DTT_CHECK(dout && dx && B > 0 && Tin > 0 && Tout > 0 && D > 0, "dtt_lerp_time_bwd: bad arguments");
    const int64_t total = (int64_t)B * Tin * D;
    DTT_CHECK((total + NT - 1) / NT < (1LL << 31), "dtt_lerp_time_bwd: too large");
    hipLaunchKernelGGL(lerp_time_bwd_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0,
                       static_cast<hipStream_t>(stream), dout, dx, Tin, Tout, D, (float)Tin / (float)Tout, total);
    DTT_LAUNCH_CHECK();
    return 0;
}
