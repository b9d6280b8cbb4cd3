// Common device helpers for the gfx950 (MI355X / CDNA4) DTTNet kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <mutex>
#include <map>
#include <utility>

struct bf16 { unsigned short x; };          // raw bfloat16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 has it in hardware (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return bf2f(p->x); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { p->x = f2bf(v); }

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float4 v;
    __device__ __forceinline__ float get(int i) const { return (&v.x)[i]; }
    __device__ __forceinline__ void set(int i, float f) { (&v.x)[i] = f; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    uint4 v;
    __device__ __forceinline__ float get(int i) const {
        uint32_t w = (&v.x)[i >> 1];
        return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
    }
    __device__ __forceinline__ void set(int i, float f) {
        uint32_t b = f2bf(f);
        uint32_t& w = (&v.x)[i >> 1];
        w = (i & 1) ? ((w & 0x0000ffffu) | (b << 16)) : ((w & 0xffff0000u) | b);
    }
    __device__ __forceinline__ void set2(int i, float lo, float hi) { (&v.x)[i >> 1] = pack_bf16x2(lo, hi); }   // i even
};
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}
// Predicated 16-byte load WITHOUT a branch and WITHOUT a select on the loaded data: hipcc turns
// `if (ok) v = load` into a branch plus an s_waitcnt vmcnt(0) per load, and a select on the result
// makes it wait for the data right after issuing the load (no overlap with the MFMAs that follow).
// So the address is selected instead: lanes that are off read 16 bytes of zeros from a device
// global.  (`safe` is unused, kept for call-site readability.)
static __device__ uint4 dtt_zero16;          // zero-initialised, never written
template <typename T> __device__ __forceinline__ Vec16<T> ld16_pred(const T* p, const T* safe, bool ok) {
    (void)safe;
    return ld16<T>(ok ? p : reinterpret_cast<const T*>(&dtt_zero16));
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
    Vec16<T> r;
    r.v.x = 0; r.v.y = 0; r.v.z = 0; r.v.w = 0;
    return r;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// bf16 path: GELU and its derivative from ONE exponential.  erf by Abramowitz-Stegun 7.1.26
// (|error| < 1.5e-7, far below bf16 resolution): ~16 VALU ops instead of ~50 for erff.  The fp32
// (parity) path keeps erff.  e = exp(-x^2/2) serves both erf(x/sqrt2) and the Gaussian density.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float h = 0.5f * poly * e;                  // 0.5 * erfc(|x|/sqrt2)
    cdf = x >= 0.f ? 1.0f - h : h;
    pdf = 0.39894228040143268f * e;
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) {
    if constexpr (sizeof(T) == 4) return gelu_f(x);
    float c, p; gelu_parts(x, c, p); return x * c;
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) {
    if constexpr (sizeof(T) == 4) return gelu_grad_f(x);
    float c, p; gelu_parts(x, c, p); return c + x * p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}


// ---- MFMA 32x32 tile helpers: one 16-wide K chunk per call.  Lane l holds row/col
// (l & 31) and the 8 k values 8*(l>>5)..+7 of the chunk (bf16: one x16 MFMA; f32:
// eight 32x32x2 MFMAs, instruction j contracting k in {j, 8+j}).
template <typename T> struct MM;
template <> struct MM<bf16> {
    static constexpr int EPT = 8;
    static constexpr int PAD = 8;            // LDS row pad (elements): 80-byte rows at BK=32
    typedef s16x8_t Frag;
    static __device__ __forceinline__ Frag ldfrag(const bf16* p) { return *reinterpret_cast<const s16x8_t*>(p); }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                      acc, 0, 0, 0);
    }
};
template <> struct MM<float> {
    static constexpr int EPT = 4;
    static constexpr int PAD = 4;            // 144-byte rows at BK=32
    struct Frag { float4 lo, hi; };
    static __device__ __forceinline__ Frag ldfrag(const float* p) {
        Frag f;
        f.lo = *reinterpret_cast<const float4*>(p);
        f.hi = *reinterpret_cast<const float4*>(p + 4);
        return f;
    }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
    }
};

// Transposing LDS read (gfx950 ds_read_b64_tr_b16): from a natural [k rows][columns] bf16 image,
// lane l (group g = l>>4, q = (l&15)>>2, p = l&3) passes the address of row (kbase + 8*(l>>5) + q),
// columns c0 + 16*(g&1) + 4p .. +3, and receives for column c0 + (l&31) the 8 consecutive k
// 8*(l>>5) .. +7 -- exactly the 32x32x16 MFMA operand fragment.  r0/r1: row addresses for the
// first / second group of 4 k.  Every lane of the wave must execute it.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ s16x8_t tr_frag(const bf16* r0, const bf16* r1) {
    s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)r0);
    s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)r1);
    s16x8_t v;
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    return v;
}

// Host side error plumbing (api.cpp)
extern "C" const char* dtt_last_error();
void dtt_set_error(const char* fmt, ...);
#define DTT_CHECK(cond, ...) do { if (!(cond)) { dtt_set_error(__VA_ARGS__); return -1; } } while (0)
// Dynamic LDS beyond 64 KB needs hipFuncSetAttribute(MaxDynamicSharedMemorySize): once per (kernel, device) and size
// reached, not per launch -- it is host time on every launch otherwise, and one more runtime call beside whatever
// other host threads are doing (c10d's watchdog queries events from a thread of its own).
inline void dtt_want_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> have;
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    int& h = have[std::make_pair(fn, dev)];
    if (bytes > h) {
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        h = bytes;
    }
}

#define DTT_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) { \
    dtt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); return -2; } } while (0)
