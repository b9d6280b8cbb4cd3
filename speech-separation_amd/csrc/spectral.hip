// STFT / iSTFT glue around the DFT GEMMs, the 1x1 input conv and the fused
// out-conv + complex-style mask + polar kernel (forward and backward).
// Reference: src/datasets/base_dataset.py:154-167 (STFT -> abs/angle),
// src/model/DTTNet/modules/encoder.py:26-27 (stack + 1x1 conv),
// src/model/DTTNet/modules/decoder.py:29 + src/model/DTTNet/model_dttnet.py:47-85
// (out conv, mask, torch.polar, torch.istft with a rectangular window).
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {
constexpr int NT = 256;

// ---------------- reflect padding (center=True) ----------------
__global__ void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ xp, int L, int pad, int Lp) {
    int b = blockIdx.y;
    for (int i = blockIdx.x * NT + threadIdx.x; i < Lp; i += gridDim.x * NT) {
        int j = i - pad;
        if (j < 0) j = -j;
        if (j >= L) j = 2 * (L - 1) - j;
        xp[(int64_t)b * Lp + i] = x[(int64_t)b * L + j];
    }
}

// ---------------- (re, im) [B][T][ld] -> mag, phase [B][F][T] ----------------
__global__ __launch_bounds__(NT) void stft_post_kernel(const float* __restrict__ reim, float* __restrict__ mag,
                                                       float* __restrict__ phase, int F, int T, int ld, int imoff) {
    __shared__ float sre[32][33], sim[32][33];
    const int b = blockIdx.z, f0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int t = t0 + i, f = f0 + tx;
        float re = 0.f, im = 0.f;
        if (t < T && f < F) {
            const float* r = reim + ((int64_t)b * T + t) * ld;
            re = r[f];
            im = (f == 0 || f == F - 1) ? 0.f : r[imoff + f];     // rfft: DC / Nyquist imaginary is +0
        }
        sre[i][tx] = re; sim[i][tx] = im;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int f = f0 + i, t = t0 + tx;
        if (f < F && t < T) {
            float re = sre[tx][i], im = sim[tx][i];
            int64_t o = ((int64_t)b * F + f) * T + t;
            mag[o] = hypotf(re, im);
            phase[o] = atan2f(im, re);
        }
    }
}

// ---------------- 1x1 input conv: planar (spec, phase) -> NHWC [B*P][C] ----------------
template <typename T>
__global__ __launch_bounds__(NT) void init_conv_fwd_kernel(const float* __restrict__ spec,
                                                           const float* __restrict__ phase,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           T* __restrict__ out, int64_t P, int C) {
    constexpr int EPT = Vec16<T>::N;
    const int cpp = C / EPT;
    const int64_t total = P * cpp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int cq = (int)(i % cpp);
        int64_t p = i / cpp;
        float s = spec[p], ph = phase[p];
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            int c = cq * EPT + e;
            o.set(e, w[c * 2] * s + w[c * 2 + 1] * ph + bias[c]);
        }
        st16<T>(out + p * C + cq * EPT, o);
    }
}

// partial[chunk][C][3] = sum_p dout[p][c] * {spec[p], phase[p], 1}
template <typename T>
__global__ __launch_bounds__(NT) void init_conv_bwd_kernel(const float* __restrict__ spec,
                                                           const float* __restrict__ phase,
                                                           const T* __restrict__ dout, float* __restrict__ partial,
                                                           int64_t P, int C, int64_t chunk) {
    constexpr int EPT = Vec16<T>::N;
    extern __shared__ float sm[];            // NT*EPT*3
    const int cpp = C / EPT, ppi = NT / cpp, nact = ppi * cpp;
    const int tid = threadIdx.x, pl = tid / cpp, cq = tid % cpp;
    const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    float a[EPT], b[EPT], c1[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) { a[e] = 0.f; b[e] = 0.f; c1[e] = 0.f; }
    if (tid < nact)
        for (int64_t p = p0 + pl; p < p1; p += ppi) {
            Vec16<T> d = ld16<T>(dout + p * C + cq * EPT);
            float s = spec[p], ph = phase[p];
#pragma unroll
            for (int e = 0; e < EPT; ++e) { float g = d.get(e); a[e] += g * s; b[e] += g * ph; c1[e] += g; }
        }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        sm[(tid * EPT + e) * 3] = (tid < nact) ? a[e] : 0.f;
        sm[(tid * EPT + e) * 3 + 1] = (tid < nact) ? b[e] : 0.f;
        sm[(tid * EPT + e) * 3 + 2] = (tid < nact) ? c1[e] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < C * 3; i += NT) {
        int c = i / 3, k = i % 3, q = c / EPT, e = c % EPT;
        float s = 0.f;
        for (int t = q; t < nact; t += cpp) s += sm[(t * EPT + e) * 3 + k];
        partial[(int64_t)blockIdx.x * C * 3 + i] = s;
    }
}

// ---------------- fused out conv (C -> 2S) + mask + polar ----------------
// x [B][F][T][C]; w [2S][C], bias [2S] f32; mix spec/phase [B][F][T] f32
// outputs: pspec, pphase [S][B][F][T] f32; Z [S*B][T][ldz] (re at f, im at imoff+f) f32: the reference's
// polar / istft stay fp32 (complex64) under bf16 autocast, so the spectrum handed to the inverse DFT does too
template <typename T, int S>
__global__ __launch_bounds__(NT) void mask_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ mspec,
                                                      const float* __restrict__ mphase, float* __restrict__ pspec,
                                                      float* __restrict__ pphase, float* __restrict__ Z, int B, int F,
                                                      int Tn, int C, int ldz, int imoff) {
    constexpr int EPT = Vec16<T>::N;
    extern __shared__ float sm[];
    float* sw = sm;                                   // [2S][C]
    float* tre = sm + 2 * S * C;                      // [S][16][17]
    float* tim = tre + S * 16 * 17;
    for (int i = threadIdx.x; i < 2 * S * C; i += NT) sw[i] = w[i];
    __syncthreads();
    const int b = blockIdx.z, f0 = blockIdx.x * 16, t0 = blockIdx.y * 16;
    const int tf = threadIdx.x >> 4, tt = threadIdx.x & 15;
    const int f = f0 + tf, t = t0 + tt;
    const bool ok = f < F && t < Tn;
    float m[2 * S];
#pragma unroll
    for (int k = 0; k < 2 * S; ++k) m[k] = bias[k];
    if (ok) {
        const T* xp = x + (((int64_t)b * F + f) * Tn + t) * C;
        for (int c0 = 0; c0 < C; c0 += EPT) {
            Vec16<T> v = ld16<T>(xp + c0);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                float xv = v.get(e);
#pragma unroll
                for (int k = 0; k < 2 * S; ++k) m[k] += xv * sw[k * C + c0 + e];
            }
        }
        const int64_t o = ((int64_t)b * F + f) * Tn + t;
        const float ms_ = mspec[o], mp_ = mphase[o];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float sp = ms_ * m[2 * s] - mp_ * m[2 * s + 1];
            float ph = ms_ * m[2 * s + 1] + mp_ * m[2 * s];
            pspec[(int64_t)s * B * F * Tn + o] = sp;
            pphase[(int64_t)s * B * F * Tn + o] = ph;
            float sn, cs;
            sincosf(ph, &sn, &cs);
            tre[(s * 16 + tt) * 17 + tf] = sp * cs;
            tim[(s * 16 + tt) * 17 + tf] = sp * sn;
        }
    }
    __syncthreads();
    // transposed write: thread (tt2 = tid>>4, tf2 = tid&15) -> Z[(s,b)][t0+tt2][f0+tf2]
    const int tt2 = threadIdx.x >> 4, tf2 = threadIdx.x & 15;
    if (t0 + tt2 < Tn && f0 + tf2 < F) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float* zr = Z + (((int64_t)s * B + b) * Tn + t0 + tt2) * ldz;
            zr[f0 + tf2] = tre[(s * 16 + tt2) * 17 + tf2];
            zr[imoff + f0 + tf2] = tim[(s * 16 + tt2) * 17 + tf2];
        }
    }
}

// backward: dspec/dphase [S][B][F][T] f32 (from the loss), dZ [S*B][T][ldz] f32 (iSTFT bwd GEMM)
// -> dx [B][F][T][C] (T), dmask [B*F*T][2S] (T)  (dw / db come from dtt_gemm / dtt_colsum on dmask)
template <typename T, int S>
__global__ __launch_bounds__(NT) void mask_bwd_kernel(const float* __restrict__ w, const float* __restrict__ mspec,
                                                      const float* __restrict__ mphase,
                                                      const float* __restrict__ pspec,
                                                      const float* __restrict__ pphase,
                                                      const float* __restrict__ dspec,
                                                      const float* __restrict__ dphase,
                                                      const float* __restrict__ dZ, T* __restrict__ dx,
                                                      T* __restrict__ dmask, int B, int F, int Tn, int C, int ldz,
                                                      int imoff) {
    constexpr int EPT = Vec16<T>::N;
    extern __shared__ float sm[];
    float* sw = sm;
    float* tre = sm + 2 * S * C;
    float* tim = tre + S * 16 * 17;
    for (int i = threadIdx.x; i < 2 * S * C; i += NT) sw[i] = w[i];
    const int b = blockIdx.z, f0 = blockIdx.x * 16, t0 = blockIdx.y * 16;
    const int tt2 = threadIdx.x >> 4, tf2 = threadIdx.x & 15;
    {
        const bool ok2 = t0 + tt2 < Tn && f0 + tf2 < F;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float re = 0.f, im = 0.f;
            if (ok2) {
                const float* zr = dZ + (((int64_t)s * B + b) * Tn + t0 + tt2) * ldz;
                re = zr[f0 + tf2];
                im = zr[imoff + f0 + tf2];
            }
            tre[(s * 16 + tt2) * 17 + tf2] = re;
            tim[(s * 16 + tt2) * 17 + tf2] = im;
        }
    }
    __syncthreads();
    const int tf = threadIdx.x >> 4, tt = threadIdx.x & 15;
    const int f = f0 + tf, t = t0 + tt;
    const bool valid = f < F && t < Tn;
    const int64_t o = valid ? ((int64_t)b * F + f) * Tn + t : 0;
    const float ms_ = mspec[o], mp_ = mphase[o];
    float dm[2 * S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int64_t os = (int64_t)s * B * F * Tn + o;
        float sp = pspec[os], ph = pphase[os];
        float dre = tre[(s * 16 + tt) * 17 + tf], dim = tim[(s * 16 + tt) * 17 + tf];
        float sn, cs;
        sincosf(ph, &sn, &cs);
        // torch.polar autograd: grad_abs = Re(conj(g) * sgn(result)) = sign(abs) * (g_re cos + g_im sin)
        float sg = sp > 0.f ? 1.f : (sp < 0.f ? -1.f : 0.f);
        float d_s = (dspec ? dspec[os] : 0.f) + sg * (dre * cs + dim * sn);
        float d_p = (dphase ? dphase[os] : 0.f) + sp * (dim * cs - dre * sn);
        dm[2 * s] = ms_ * d_s + mp_ * d_p;
        dm[2 * s + 1] = -mp_ * d_s + ms_ * d_p;
    }
    if (valid) {
        if constexpr (sizeof(T) == 2 && S == 2) {          // four bf16 = one 8-byte store instead of four 2-byte ones
            uint2 pk;
            pk.x = pack_bf16x2(dm[0], dm[1]);
            pk.y = pack_bf16x2(dm[2], dm[3]);
            *reinterpret_cast<uint2*>(dmask + o * 4) = pk;
        } else {
#pragma unroll
            for (int k = 0; k < 2 * S; ++k) stf(dmask + o * 2 * S + k, dm[k]);
        }
    }
    // dx = dmask * W: the mask gradients of the tile go through LDS so that consecutive lanes write
    // consecutive 16-byte pieces (a tile row of 16 pixels is one contiguous run of 16*C elements);
    // one pixel per lane left every store instruction with 16 bytes per 2C-byte stride
    float* sdm = tim + S * 16 * 17;
#pragma unroll
    for (int k = 0; k < 2 * S; ++k) sdm[threadIdx.x * 2 * S + k] = valid ? dm[k] : 0.f;
    __syncthreads();
    const int cpr = C / EPT;
    for (int id = threadIdx.x; id < 256 * cpr; id += NT) {
        const int px = id / cpr, ch = id - px * cpr;
        const int pf = f0 + (px >> 4), pt = t0 + (px & 15);
        if (pf >= F || pt >= Tn) continue;
        float dk[2 * S];
#pragma unroll
        for (int k = 0; k < 2 * S; ++k) dk[k] = sdm[px * 2 * S + k];
        Vec16<T> v;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 2 * S; ++k) a += dk[k] * sw[k * C + ch * EPT + e];
            v.set(e, a);
        }
        st16<T>(dx + (((int64_t)b * F + pf) * Tn + pt) * C + ch * EPT, v);
    }
}

// ---------------- overlap-add (rectangular window) ----------------
__device__ __forceinline__ float ola_env(int sp, int n_fft, int hop, int Tn) {
    // number of frames t in [0,Tn) with 0 <= sp - t*hop < n_fft
    int tmax = min(Tn - 1, sp / hop);
    int tmin = max(0, (sp - n_fft + hop) / hop);      // ceil((sp - n_fft + 1)/hop)
    if (sp - n_fft + 1 <= 0) tmin = 0;
    return (float)(tmax - tmin + 1);
}

// frames [SB][T][n_fft] f32 -> out [SB][length] f32
__global__ __launch_bounds__(NT) void ola_fwd_kernel(const float* __restrict__ frames, float* __restrict__ out,
                                                     int n_fft, int hop, int Tn, int length) {
    const int sb = blockIdx.y;
    for (int s = blockIdx.x * NT + threadIdx.x; s < length; s += gridDim.x * NT) {
        int sp = s + n_fft / 2;
        float v = 0.f;
        if (sp < n_fft + hop * (Tn - 1)) {
            int tmax = min(Tn - 1, sp / hop);
            int tmin = (sp - n_fft + 1 <= 0) ? 0 : (sp - n_fft + hop) / hop;
            float a = 0.f;
            for (int t = tmin; t <= tmax; ++t) a += frames[((int64_t)sb * Tn + t) * n_fft + sp - t * hop];
            v = a / (float)(tmax - tmin + 1);
        }
        out[(int64_t)sb * length + s] = v;
    }
}

// dout [SB][length] f32 -> gpad [SB][Lp] (T): gradient of the padded OLA signal (already / env)
template <typename T>
__global__ __launch_bounds__(NT) void ola_bwd_kernel(const float* __restrict__ dout, T* __restrict__ gpad, int n_fft,
                                                     int hop, int Tn, int length, int Lp) {
    const int sb = blockIdx.y;
    for (int sp = blockIdx.x * NT + threadIdx.x; sp < Lp; sp += gridDim.x * NT) {
        int s = sp - n_fft / 2;
        float v = 0.f;
        if (s >= 0 && s < length) v = dout[(int64_t)sb * length + s] / ola_env(sp, n_fft, hop, Tn);
        stf(gpad + (int64_t)sb * Lp + sp, v);
    }
}

int ept_of(int dtype) { return dtype ? 8 : 4; }
int grid_for(int64_t total) { int64_t g = (total + NT - 1) / NT; return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192); }
}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int dtt_reflect_pad(const float* x, float* xp, int B, int L, int pad, void* stream) {
    DTT_CHECK(B > 0 && L > pad && pad >= 0, "dtt_reflect_pad: bad shape (L=%d pad=%d)", L, pad);
    int Lp = L + 2 * pad;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(min(64, (Lp + NT - 1) / NT), B), dim3(NT), 0, ST, x, xp, L, pad, Lp);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_stft_post(const float* reim, float* mag, float* phase, int B, int F, int T, int ld, int imoff,
                             void* stream) {
    DTT_CHECK(B > 0 && F > 1 && T > 0 && imoff >= F && ld >= imoff + F, "dtt_stft_post: bad shape");
    hipLaunchKernelGGL(stft_post_kernel, dim3((F + 31) / 32, (T + 31) / 32, B), dim3(NT), 0, ST, reim, mag, phase, F, T, ld, imoff);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_init_conv_fwd(const float* spec, const float* phase, const float* w, const float* bias, void* out,
                                 int64_t P, int C, int dtype, void* stream) {
    DTT_CHECK(C % ept_of(dtype) == 0 && P > 0, "dtt_init_conv_fwd: bad shape C=%d", C);
    int grid = grid_for(P * (C / ept_of(dtype)));
    if (dtype) hipLaunchKernelGGL(init_conv_fwd_kernel<bf16>, dim3(grid), dim3(NT), 0, ST, spec, phase, w, bias, (bf16*)out, P, C);
    else hipLaunchKernelGGL(init_conv_fwd_kernel<float>, dim3(grid), dim3(NT), 0, ST, spec, phase, w, bias, (float*)out, P, C);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_init_conv_bwd(const float* spec, const float* phase, const void* dout, float* partial, int64_t P,
                                 int C, int nchunks, int dtype, void* stream) {
    const int ept = ept_of(dtype);
    DTT_CHECK(C % ept == 0 && C / ept <= NT && P > 0 && nchunks > 0, "dtt_init_conv_bwd: bad shape C=%d", C);
    int64_t chunk = (P + nchunks - 1) / nchunks;
    size_t sm = (size_t)NT * ept * 3 * sizeof(float);
    if (dtype) hipLaunchKernelGGL(init_conv_bwd_kernel<bf16>, dim3(nchunks), dim3(NT), sm, ST, spec, phase, (const bf16*)dout, partial, P, C, chunk);
    else hipLaunchKernelGGL(init_conv_bwd_kernel<float>, dim3(nchunks), dim3(NT), sm, ST, spec, phase, (const float*)dout, partial, P, C, chunk);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_mask_fwd(const void* x, const float* w, const float* bias, const float* mspec, const float* mphase,
                            float* pspec, float* pphase, void* Z, int B, int F, int T, int C, int S, int ldz,
                            int imoff, int dtype, void* stream) {
    DTT_CHECK(C % ept_of(dtype) == 0 && (S == 2 || S == 1 || S == 3) && imoff >= F && ldz >= imoff + F,
              "dtt_mask_fwd: bad shape C=%d S=%d", C, S);
    dim3 grid((F + 15) / 16, (T + 15) / 16, B);
    size_t sm = (size_t)(2 * S * C + 2 * S * 16 * 17) * sizeof(float);
#define MF(TT, SS) hipLaunchKernelGGL((mask_fwd_kernel<TT, SS>), grid, dim3(NT), sm, ST, (const TT*)x, w, bias, mspec, mphase, pspec, pphase, (float*)Z, B, F, T, C, ldz, imoff)
    if (dtype) { if (S == 1) MF(bf16, 1); else if (S == 2) MF(bf16, 2); else MF(bf16, 3); }
    else { if (S == 1) MF(float, 1); else if (S == 2) MF(float, 2); else MF(float, 3); }
#undef MF
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_mask_bwd(const float* w, const float* mspec, const float* mphase, const float* pspec,
                            const float* pphase, const float* dspec, const float* dphase, const float* dZ, void* dx,
                            void* dmask, int B, int F, int T, int C, int S, int ldz, int imoff, int dtype,
                            void* stream) {
    DTT_CHECK(C % ept_of(dtype) == 0 && (S == 2 || S == 1 || S == 3) && imoff >= F && ldz >= imoff + F,
              "dtt_mask_bwd: bad shape C=%d S=%d", C, S);
    dim3 grid((F + 15) / 16, (T + 15) / 16, B);
    size_t sm = (size_t)(2 * S * C + 2 * S * 16 * 17 + 256 * 2 * S) * sizeof(float);
#define MB(TT, SS) hipLaunchKernelGGL((mask_bwd_kernel<TT, SS>), grid, dim3(NT), sm, ST, w, mspec, mphase, pspec, pphase, dspec, dphase, dZ, (TT*)dx, (TT*)dmask, B, F, T, C, ldz, imoff)
    if (dtype) { if (S == 1) MB(bf16, 1); else if (S == 2) MB(bf16, 2); else MB(bf16, 3); }
    else { if (S == 1) MB(float, 1); else if (S == 2) MB(float, 2); else MB(float, 3); }
#undef MB
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_ola_fwd(const float* frames, float* out, int SB, int T, int n_fft, int hop, int length,
                           void* stream) {
    DTT_CHECK(SB > 0 && T > 0 && n_fft > 0 && hop > 0 && length > 0, "dtt_ola_fwd: bad shape");
    hipLaunchKernelGGL(ola_fwd_kernel, dim3(min(64, (length + NT - 1) / NT), SB), dim3(NT), 0, ST, frames, out, n_fft, hop, T, length);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_ola_bwd(const float* dout, void* gpad, int SB, int T, int n_fft, int hop, int length, int dtype,
                           void* stream) {
    DTT_CHECK(SB > 0 && T > 0 && n_fft > 0 && hop > 0 && length > 0, "dtt_ola_bwd: bad shape");
    int Lp = n_fft + hop * (T - 1);
    dim3 grid(min(64, (Lp + NT - 1) / NT), SB);
    if (dtype) hipLaunchKernelGGL(ola_bwd_kernel<bf16>, grid, dim3(NT), 0, ST, dout, (bf16*)gpad, n_fft, hop, T, length, Lp);
    else hipLaunchKernelGGL(ola_bwd_kernel<float>, grid, dim3(NT), 0, ST, dout, (float*)gpad, n_fft, hop, T, length, Lp);
    DTT_LAUNCH_CHECK();
    return 0;
}
