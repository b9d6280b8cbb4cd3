// Gradient clip + AdamW over all parameters in two launches (the step right
// after the hot path).  Reference: torch.optim.AdamW (train.yaml:20-23) after clip_grad_norm_
// (base_trainer.py:478-490, max_grad_norm train.yaml:59) -- ~30 launches and ~1.5 ms of host time per step
// through torch's foreach path; here one sum-of-squares pass and one update pass over a table of tensors.
//
//   total = sqrt(sum_i |g_i|^2);  coef = min(1, max_norm / (total + 1e-6))           (clip_grad_norm_)
//   g = coef * g;  p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)                        (AdamW, decoupled decay)
//
// The work is cut into chunks of CH elements; chunk c = (tensor, offset).  The per-chunk sums of
// squares are merged in a fixed order by every workgroup of the update pass (a few hundred floats), so
// the norm is deterministic and no third launch is needed.
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {
constexpr int NT = 256, CH = 32768;

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(NT) void adamw_sumsq_kernel(const DttOptTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                         float* __restrict__ partial) {
    __shared__ float sh[4];
    const int2 ck = chunks[blockIdx.x];
    const DttOptTensor t = tab[ck.x];
    const float* g = reinterpret_cast<const float*>(t.g);
    float acc = 0.f;
    if (g) {
        const int64_t beg = (int64_t)ck.y * CH, end = min(t.n, beg + CH);
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
            const int64_t nv = (end - beg) >> 2;
            const float4* g4 = reinterpret_cast<const float4*>(g + beg);
            for (int64_t i = threadIdx.x; i < nv; i += NT) {
                const float4 x = g4[i];
                acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            }
            for (int64_t i = beg + (nv << 2) + threadIdx.x; i < end; i += NT) acc += g[i] * g[i];
        } else {
            for (int64_t i = beg + threadIdx.x; i < end; i += NT) acc += g[i] * g[i];
        }
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Deterministic merge of the per-chunk sums of squares into one float (the sharded step all-reduces it).
__global__ __launch_bounds__(NT) void adamw_merge_kernel(const float* __restrict__ partial, int nchunks, float* __restrict__ out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nchunks; i += NT) acc += partial[i];
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = s;
}

// step_dev (device int32, may be NULL): AdamW step count after this update; when given, the bias corrections are
// computed here instead of on the host, so that a captured HIP graph replays with the count its caller keeps on
// the device.  sumsq_dev (may be NULL): total squared gradient norm, already merged (and, for a sharded step,
// all-reduced) -- replaces the merge of `partial`.  gscale: factor applied to every gradient before anything else
// (1 / world size when the collective summed instead of averaging); the reported norm is that of the scaled gradients.
__global__ __launch_bounds__(NT) void adamw_update_kernel(const DttOptTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                          const float* __restrict__ partial, int nchunks, float max_norm,
                                                          float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                          const int32_t* __restrict__ step_dev,
                                                          const float* __restrict__ sumsq_dev, float gscale,
                                                          float* __restrict__ norm_out) {
    __shared__ float sh[4];
    float coef = gscale;
    if (step_dev) {
        const double st = (double)step_dev[0];
        bc1 = (float)(1.0 - pow((double)b1, st));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, st));
    }
    if (max_norm > 0.f) {
        float ss;
        if (sumsq_dev) ss = sumsq_dev[0];
        else {
            float acc = 0.f;
            for (int i = threadIdx.x; i < nchunks; i += NT) acc += partial[i];
            ss = block_sum(acc, sh);
        }
        const float total = sqrtf(ss) * gscale;
        coef = gscale * fminf(1.f, max_norm / (total + 1e-6f));
        if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
    }
    const int2 ck = chunks[blockIdx.x];
    const DttOptTensor t = tab[ck.x];
    const float* g = reinterpret_cast<const float*>(t.g);
    if (!g) return;                                        // no gradient this step: parameter untouched
    float* p = reinterpret_cast<float*>(t.p);
    const float decay = 1.f - t.lr * t.wd, step = t.lr / bc1;
    const int64_t beg = (int64_t)ck.y * CH, end = min(t.n, beg + CH);
    auto upd = [&](float& pv, float gv, float& mv, float& vv) {
        gv *= coef;
        pv *= decay;
        mv = b1 * mv + (1.f - b1) * gv;
        vv = b2 * vv + (1.f - b2) * gv * gv;
        pv -= step * mv / (sqrtf(vv) / bc2_sqrt + eps);
    };
    const bool al = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(t.m) |
                      reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    int64_t done = beg;
    if (al) {
        const int64_t nv = (end - beg) >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g + beg);
        float4* p4 = reinterpret_cast<float4*>(p + beg);
        float4* m4 = reinterpret_cast<float4*>(t.m + beg);
        float4* v4 = reinterpret_cast<float4*>(t.v + beg);
        for (int64_t i = threadIdx.x; i < nv; i += NT) {
            const float4 gq = g4[i];
            float4 pq = p4[i], mq = m4[i], vq = v4[i];
            upd(pq.x, gq.x, mq.x, vq.x); upd(pq.y, gq.y, mq.y, vq.y);
            upd(pq.z, gq.z, mq.z, vq.z); upd(pq.w, gq.w, mq.w, vq.w);
            p4[i] = pq; m4[i] = mq; v4[i] = vq;
        }
        done = beg + (nv << 2);
    }
    for (int64_t i = done + threadIdx.x; i < end; i += NT) upd(p[i], g[i], t.m[i], t.v[i]);
}
}  // namespace

// tab [ntensors] and chunks [nchunks] = {tensor index, chunk index within the tensor} live in device
// memory (the caller refreshes tab[].g and tab[].lr each step); partial: nchunks floats of scratch.
// max_norm <= 0: no clipping (the sum-of-squares pass is skipped).  norm_out (may be NULL) receives the
// total gradient norm before clipping.  step >= 1 is the AdamW step count after this update.
extern "C" int dtt_clip_adamw(const DttOptTensor* tab, int ntensors, const int32_t* chunks, int nchunks, float* partial,
                              float max_norm, float beta1, float beta2, float eps, int step, float* norm_out, void* stream) {
    DTT_CHECK(tab && chunks && partial && ntensors > 0 && nchunks > 0 && step >= 1, "dtt_clip_adamw: bad arguments");
    DTT_CHECK(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "dtt_clip_adamw: bad hyper-parameters");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int2* ck = reinterpret_cast<const int2*>(chunks);
    if (max_norm > 0.f) hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(nchunks), dim3(NT), 0, st, tab, ck, partial);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_update_kernel, dim3(nchunks), dim3(NT), 0, st, tab, ck, partial, nchunks, max_norm, beta1, beta2,
                       eps, (float)bc1, (float)sqrt(bc2), (const int32_t*)nullptr, (const float*)nullptr, 1.0f, norm_out);
    DTT_LAUNCH_CHECK();
    return 0;
}

// Per-chunk sums of squares of the gradients in `tab`, merged in a fixed order into sumsq_out[0] (2 launches).
// First half of the sharded step: every rank runs it on its shard of the reduce-scattered
// gradients, all-reduces the one float, and hands it to dtt_clip_adamw_ex as sumsq_dev.
extern "C" int dtt_grad_sumsq(const DttOptTensor* tab, int ntensors, const int32_t* chunks, int nchunks, float* partial,
                              float* sumsq_out, void* stream) {
    DTT_CHECK(tab && chunks && partial && sumsq_out && ntensors > 0 && nchunks > 0, "dtt_grad_sumsq: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(nchunks), dim3(NT), 0, st, tab, reinterpret_cast<const int2*>(chunks), partial);
    hipLaunchKernelGGL(adamw_merge_kernel, dim3(1), dim3(NT), 0, st, partial, nchunks, sumsq_out);
    DTT_LAUNCH_CHECK();
    return 0;
}

// dtt_clip_adamw with (a) the step count read from device memory (HIP-graph replay), (b) an externally merged
// squared norm (sharded step) and (c) a gradient scale folded into the clip coefficient.  step_dev and sumsq_dev
// may each be NULL: then `step` (>= 1) is used / the norm is computed here from the gradients in `tab`.
extern "C" int dtt_clip_adamw_ex(const DttOptTensor* tab, int ntensors, const int32_t* chunks, int nchunks, float* partial,
                                 float max_norm, float beta1, float beta2, float eps, int step, const int32_t* step_dev,
                                 const float* sumsq_dev, float grad_scale, float* norm_out, void* stream) {
    DTT_CHECK(tab && chunks && partial && ntensors > 0 && nchunks > 0 && (step_dev || step >= 1) && grad_scale > 0.f,
              "dtt_clip_adamw_ex: bad arguments");
    DTT_CHECK(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "dtt_clip_adamw_ex: bad hyper-parameters");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int2* ck = reinterpret_cast<const int2*>(chunks);
    if (max_norm > 0.f && !sumsq_dev) hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(nchunks), dim3(NT), 0, st, tab, ck, partial);
    const int sh = step >= 1 ? step : 1;
    const double bc1 = 1.0 - pow((double)beta1, sh), bc2 = 1.0 - pow((double)beta2, sh);
    hipLaunchKernelGGL(adamw_update_kernel, dim3(nchunks), dim3(NT), 0, st, tab, ck, partial, nchunks, max_norm, beta1, beta2,
                       eps, (float)bc1, (float)sqrt(bc2), step_dev, sumsq_dev, grad_scale, norm_out);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_clip_adamw_chunk(void) { return CH; }
