// ShuffleNetV2 units of the FiLM model's frozen lip-reading front end, forward only (reference:
// src/model/lipreading/models/shufflenetv2.py:42-104 InvertedResidual, :26-40 channel_shuffle, :151-152 conv_last +
// globalpool; run per frame by src/model/lipreading/model.py:164-174 behind DTTNetPlus/modules/video_encoder.py:41-47).
// Activations are channels-last [pixels][channels]; a unit's 1x1 convolutions are dtt_gemm calls on those rows, and
// the kernels here are the HBM-bound passes between them, each reading its input once and writing its output once:
//   * depthwise 3x3 (stride 1 | 2, pad 1) with the BatchNorm + ReLU of the 1x1 conv in front of it applied to the
//     taps as they are loaded (the normalised tensor is never written);
//   * the BatchNorm BEHIND the depthwise conv (no activation) folded into the next 1x1 conv's weights and a bias
//     (W diag(sc), W sh): one tiny launch instead of a pass over the tensor;
//   * BatchNorm + ReLU of the branch outputs fused with concat + channel_shuffle(groups = 2): out[2i] = a[i],
//     out[2i + 1] = b[i], a = the untouched first half of the unit's input (stride 1) or branch 1 (stride 2);
//   * BatchNorm + ReLU + AvgPool2d(3) behind conv_last.
// Channel layout: a tensor of C = 2h logical channels is stored as two halves of h channels, each padded to hp = a
// multiple of 8 columns of zeros, so that the second half (the stride-1 unit's branch input) starts 16-byte aligned
// for any h (58, 116, 232 at width 1.0): logical channel j sits in column (j / h) * hp + j % h.
#include <algorithm>
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {
constexpr int NT = 256;

// One thread owns a 16-byte channel chunk of one output COLUMN of one frame and walks down the output rows with a
// 3-row window of the input in registers: the nine tap weights and the BatchNorm constants are loaded once per thread
// (per-output threads re-read them for every pixel: 350 bytes of L1 traffic per 16 bytes written), each input vector
// once per column triple, and the next row's loads are issued before the current row is summed.  Lanes run over
// (column, chunk): a wave reads contiguous spans of an image row.
template <typename T, int STRIDE>
__global__ __launch_bounds__(NT) void dw3x3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ stats, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, T* __restrict__ y, int H, int W, int C,
                                                   int Ho, int Wo, int total) {
    constexpr int EPT = Vec16<T>::N, NEW = STRIDE;         // input rows that enter the window per output row
    const int cpp = C / EPT;
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= total) return;
    const int cq = idx % cpp, r = idx / cpp, wo = r % Wo, n = r / Wo;
    const bool pre = stats != nullptr;
    float sc[EPT], sh[EPT], wr[9][EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        sc[e] = pre ? gamma[c] * stats[c * 2 + 1] : 1.f;
        sh[e] = pre ? beta[c] - stats[c * 2] * sc[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < EPT; ++e) wr[k][e] = w[k * C + cq * EPT + e];
    const int w0 = wo * STRIDE - 1;
    bool okw[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) okw[kw] = w0 + kw >= 0 && w0 + kw < W;
    const T* xb = x + (int64_t)n * H * W * C + cq * EPT;
    T* yb = y + ((int64_t)n * Ho * Wo + wo) * C + cq * EPT;
    Vec16<T> win[3][3], nx[NEW][3];
    bool rok[3], nok[NEW];
    auto load_row = [&](int hh, Vec16<T> (&dst)[3], bool& ok) {
        ok = hh >= 0 && hh < H;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) dst[kw] = ld16_pred<T>(xb + ((int64_t)hh * W + w0 + kw) * C, xb, ok && okw[kw]);
    };
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) load_row(kh - 1, win[kh], rok[kh]);
    for (int ho = 0; ho < Ho; ++ho) {
#pragma unroll
        for (int i = 0; i < NEW; ++i) load_row((ho + 1) * STRIDE - 1 + (3 - NEW) + i, nx[i], nok[i]);   // next window's new rows
        float acc[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool ok = rok[kh] && okw[kw];          // zero padding is applied to the ACTIVATED input
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    float a = win[kh][kw].get(e);
                    if (pre) a = fmaxf(a * sc[e] + sh[e], 0.f);
                    acc[e] += ok ? a * wr[kh * 3 + kw][e] : 0.f;
                }
            }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPT; ++e) o.set(e, acc[e]);
        st16<T>(yb + (int64_t)ho * Wo * C, o);
#pragma unroll
        for (int kh = 0; kh < 3 - NEW; ++kh) {
            rok[kh] = rok[kh + NEW];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) win[kh][kw] = win[kh + NEW][kw];
        }
#pragma unroll
        for (int i = 0; i < NEW; ++i) {
            rok[3 - NEW + i] = nok[i];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) win[3 - NEW + i][kw] = nx[i][kw];
        }
    }
}

struct ShufP {
    const void *a, *b;
    void* out;
    const float *sa, *ga, *ba, *sb, *gb, *bb;
    int64_t P;
    int lda, ldb, h, hp, ppb;
};

constexpr int SHUF_MAXH = 512;

template <typename T>
__global__ __launch_bounds__(NT) void bn_shuffle_kernel(ShufP p) {
    constexpr int EPT = Vec16<T>::N, HE = EPT / 2;
    __shared__ float cst[4][SHUF_MAXH];                    // sc_a, sh_a, sc_b, sh_b per source channel
    for (int c = threadIdx.x; c < p.h; c += NT) {
        float s = 1.f, t = 0.f;
        if (p.sa) { s = p.ga[c] * p.sa[c * 2 + 1]; t = p.ba[c] - p.sa[c * 2] * s; }
        cst[0][c] = s; cst[1][c] = t;
        s = 1.f; t = 0.f;
        if (p.sb) { s = p.gb[c] * p.sb[c * 2 + 1]; t = p.bb[c] - p.sb[c * 2] * s; }
        cst[2][c] = s; cst[3][c] = t;
    }
    __syncthreads();
    const T* a = static_cast<const T*>(p.a);
    const T* b = static_cast<const T*>(p.b);
    T* out = static_cast<T*>(p.out);
    const int cpp = 2 * p.hp / EPT;
    const int64_t p0 = (int64_t)blockIdx.x * p.ppb;
    const int64_t p1 = p0 + p.ppb < p.P ? p0 + p.ppb : p.P;
    const int64_t items = (p1 - p0) * cpp;
    const bool acta = p.sa != nullptr, actb = p.sb != nullptr;
    for (int64_t it = threadIdx.x; it < items; it += NT) {
        const int cq = (int)(it % cpp);
        const int64_t pix = p0 + it / cpp;
        const int pc0 = cq * EPT, half = pc0 / p.hp, pos0 = pc0 - half * p.hp;
        const int i0 = (half * p.h + pos0) >> 1;           // even logical channel 2 i0 -> a[i0], 2 i0 + 1 -> b[i0]
        const T* ar = a + pix * p.lda + i0;
        const T* br = b + pix * p.ldb + i0;
        float o[EPT];
#pragma unroll
        for (int e = 0; e < HE; ++e) {
            const bool live = pos0 + 2 * e < p.h;          // (h even: the pair 2e, 2e + 1 is live or dead together)
            float va = 0.f, vb = 0.f;
            if (live) {
                va = ldf(ar + e);
                vb = ldf(br + e);
                va = va * cst[0][i0 + e] + cst[1][i0 + e];
                vb = vb * cst[2][i0 + e] + cst[3][i0 + e];
                if (acta) va = fmaxf(va, 0.f);
                if (actb) vb = fmaxf(vb, 0.f);
            }
            o[2 * e] = va;
            o[2 * e + 1] = vb;
        }
        Vec16<T> ov;
#pragma unroll
        for (int e = 0; e < EPT; ++e) ov.set(e, o[e]);
        st16<T>(out + pix * (2 * p.hp) + pc0, ov);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_relu_avgpool_kernel(const T* __restrict__ y, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ out, int64_t total, int S, int C) {
    constexpr int EPT = Vec16<T>::N;
    const int cpp = C / EPT;
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (idx >= total) return;
    const int cq = (int)(idx % cpp);
    const int64_t f = idx / cpp;
    float sc[EPT], sh[EPT], acc[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        sc[e] = gamma[c] * stats[c * 2 + 1];
        sh[e] = beta[c] - stats[c * 2] * sc[e];
        acc[e] = 0.f;
    }
    const T* yb = y + f * S * C + cq * EPT;
    for (int s = 0; s < S; ++s) {
        const Vec16<T> v = ld16<T>(yb + (int64_t)s * C);
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[e] += fmaxf(v.get(e) * sc[e] + sh[e], 0.f);
    }
    const float inv = 1.f / (float)S;
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[f * C + cq * EPT + e] = acc[e] * inv;
}

template <typename T>
__global__ __launch_bounds__(NT) void pw_fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ wout, float* __restrict__ bias, int K) {
    __shared__ float red[NT / 64];
    const int n = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += NT) {
        const float sc = gamma[k] * stats[k * 2 + 1];
        const float sh = beta[k] - stats[k * 2] * sc;
        const float wv = w[(int64_t)n * K + k];
        stf(wout + (int64_t)n * K + k, wv * sc);
        s += wv * sh;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        bias[n] = t;
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

extern "C" int dtt_dwconv3x3_bnrelu(const void* x, const float* w9c, const float* stats, const float* gamma,
                                    const float* beta, void* y, int N, int H, int W, int C, int stride, int dtype,
                                    void* stream) {
    const int ept = dtype == 1 ? 8 : 4;
    DTT_CHECK(dtype == 0 || dtype == 1, "dtt_dwconv3x3_bnrelu: bad dtype %d", dtype);
    DTT_CHECK(x && w9c && y && N > 0 && H > 0 && W > 0 && C > 0 && C % ept == 0 && (stride == 1 || stride == 2),
              "dtt_dwconv3x3_bnrelu: bad arguments (N=%d H=%d W=%d C=%d stride=%d)", N, H, W, C, stride);
    DTT_CHECK((stats == nullptr) == (gamma == nullptr) && (stats == nullptr) == (beta == nullptr),
              "dtt_dwconv3x3_bnrelu: stats, gamma and beta go together");
    DTT_CHECK(al16(x) && al16(y) && al16(w9c), "dtt_dwconv3x3_bnrelu: operands must be 16-byte aligned");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t total = (int64_t)N * Wo * (C / ept);
    DTT_CHECK(total < (1LL << 31) && (int64_t)H * W * C < (1LL << 31), "dtt_dwconv3x3_bnrelu: too many columns");
    const dim3 grid((unsigned)((total + NT - 1) / NT));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DTT_DW(T, S) hipLaunchKernelGGL((dw3x3_kernel<T, S>), grid, dim3(NT), 0, st, static_cast<const T*>(x), w9c, stats, gamma, \
                                         beta, static_cast<T*>(y), H, W, C, Ho, Wo, (int)total)
    if (dtype == 1) { if (stride == 1) DTT_DW(bf16, 1); else DTT_DW(bf16, 2); }
    else { if (stride == 1) DTT_DW(float, 1); else DTT_DW(float, 2); }
#undef DTT_DW
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_bn_relu_shuffle(const void* a, int lda, const float* stats_a, const float* gamma_a, const float* beta_a,
                                   const void* b, int ldb, const float* stats_b, const float* gamma_b, const float* beta_b,
                                   void* out, int64_t P, int h, int hp, int dtype, void* stream) {
    const int ept = dtype == 1 ? 8 : 4;
    DTT_CHECK(dtype == 0 || dtype == 1, "dtt_bn_relu_shuffle: bad dtype %d", dtype);
    DTT_CHECK(a && b && out && P > 0 && h > 0 && h % 2 == 0 && hp >= h && hp % 8 == 0 && hp <= SHUF_MAXH && lda >= h && ldb >= h,
              "dtt_bn_relu_shuffle: bad arguments (P=%lld h=%d hp=%d lda=%d ldb=%d)", (long long)P, h, hp, lda, ldb);
    DTT_CHECK((stats_a == nullptr) == (gamma_a == nullptr) && (stats_a == nullptr) == (beta_a == nullptr) &&
              (stats_b == nullptr) == (gamma_b == nullptr) && (stats_b == nullptr) == (beta_b == nullptr),
              "dtt_bn_relu_shuffle: stats, gamma and beta go together");
    DTT_CHECK(al16(out), "dtt_bn_relu_shuffle: out must be 16-byte aligned");
    ShufP p;
    p.a = a; p.b = b; p.out = out;
    p.sa = stats_a; p.ga = gamma_a; p.ba = beta_a; p.sb = stats_b; p.gb = gamma_b; p.bb = beta_b;
    p.P = P; p.lda = lda; p.ldb = ldb; p.h = h; p.hp = hp;
    const int cpp = 2 * hp / ept;
    p.ppb = std::max(1, 16 * NT / cpp);                     // ~16 items per thread: the constants' set-up is amortised
    const int64_t nb = (P + p.ppb - 1) / p.ppb;
    DTT_CHECK(nb < (1LL << 31), "dtt_bn_relu_shuffle: too many pixels");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 1) hipLaunchKernelGGL(bn_shuffle_kernel<bf16>, dim3((unsigned)nb), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL(bn_shuffle_kernel<float>, dim3((unsigned)nb), dim3(NT), 0, st, p);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_bn_relu_avgpool(const void* y, const float* stats, const float* gamma, const float* beta, float* out,
                                   int NF, int S, int C, int dtype, void* stream) {
    const int ept = dtype == 1 ? 8 : 4;
    DTT_CHECK(dtype == 0 || dtype == 1, "dtt_bn_relu_avgpool: bad dtype %d", dtype);
    DTT_CHECK(y && stats && gamma && beta && out && NF > 0 && S > 0 && C > 0 && C % ept == 0 && al16(y),
              "dtt_bn_relu_avgpool: bad arguments (NF=%d S=%d C=%d)", NF, S, C);
    const int64_t total = (int64_t)NF * (C / ept);
    const dim3 grid((unsigned)((total + NT - 1) / NT));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 1)
        hipLaunchKernelGGL(bn_relu_avgpool_kernel<bf16>, grid, dim3(NT), 0, st, static_cast<const bf16*>(y), stats, gamma,
                           beta, out, total, S, C);
    else
        hipLaunchKernelGGL(bn_relu_avgpool_kernel<float>, grid, dim3(NT), 0, st, static_cast<const float*>(y), stats, gamma,
                           beta, out, total, S, C);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_pw_fold_bn(const float* w, const float* stats, const float* gamma, const float* beta, void* wout,
                              float* bias, int N, int K, int dtype, void* stream) {
    DTT_CHECK(dtype == 0 || dtype == 1, "dtt_pw_fold_bn: bad dtype %d", dtype);
    DTT_CHECK(w && stats && gamma && beta && wout && bias && N > 0 && K > 0, "dtt_pw_fold_bn: bad arguments (N=%d K=%d)", N, K);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 1)
        hipLaunchKernelGGL(pw_fold_bn_kernel<bf16>, dim3(N), dim3(NT), 0, st, w, stats, gamma, beta, static_cast<bf16*>(wout),
                           bias, K);
    else
        hipLaunchKernelGGL(pw_fold_bn_kernel<float>, dim3(N), dim3(NT), 0, st, w, stats, gamma, beta, static_cast<float*>(wout),
                           bias, K);
    DTT_LAUNCH_CHECK();
    return 0;
}
