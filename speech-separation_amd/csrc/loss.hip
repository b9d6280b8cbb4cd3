// Fused PIT-L1 loss (forward + backward) and SI-SNR reductions.  HBM-bound:
// every prediction / source plane is read exactly once per pass; reductions are
// two-stage and deterministic.
// Reference: src/loss/dttnet_loss.py:28-34, src/loss/pit_loss_wrapper.py:41-71,
// src/loss/dttnet_loss_wrapper.py:41-62, src/metrics/si_snr.py:26-34.
#include <mutex>
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {
constexpr int NT = 256;
constexpr int MAXS = 4;

struct SrcPtrs { const float* p[MAXS]; };

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}

// pairwise |pred_i - src_j| sums over one plane kind.
// pred [S][B][n]; src_j [B][n]; partial[b][chunk][S*S]
template <int S>
__global__ __launch_bounds__(NT) void l1_pair_kernel(const float* __restrict__ pred, SrcPtrs src,
                                                     float* __restrict__ partial, int B, int64_t n, int64_t chunk) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    float acc[S * S];
#pragma unroll
    for (int k = 0; k < S * S; ++k) acc[k] = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += NT) {
        float pv[S], sv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            pv[s] = pred[((int64_t)s * B + b) * n + i];
            sv[s] = src.p[s][(int64_t)b * n + i];
        }
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int c = 0; c < S; ++c) acc[a * S + c] += fabsf(sv[c] - pv[a]);
    }
#pragma unroll
    for (int k = 0; k < S * S; ++k) {
        float s = block_sum(acc[k], red);
        if (threadIdx.x == 0) partial[((int64_t)b * gridDim.x + blockIdx.x) * S * S + k] = s;
    }
}

__device__ __forceinline__ bool next_perm(int* a, int n) {
    int i = n - 2;
    while (i >= 0 && a[i] >= a[i + 1]) --i;
    if (i < 0) return false;
    int j = n - 1;
    while (a[j] <= a[i]) --j;
    int t = a[i]; a[i] = a[j]; a[j] = t;
    for (int l = i + 1, r = n - 1; l < r; ++l, --r) { t = a[l]; a[l] = a[r]; a[r] = t; }
    return true;
}

// cost[b][i][j] = (spec+phase sums)/nft + audio sum / L ; permutations in itertools
// (= lexicographic) order, first minimum wins; perm[pred] = source.
__global__ void pit_select_kernel(const float* __restrict__ p_spec, const float* __restrict__ p_phase,
                                  const float* __restrict__ p_audio, int nc_sp, int nc_au, float inv_ft, float inv_l,
                                  int S, int B, int fixed, float* __restrict__ item_loss,
                                  int64_t* __restrict__ permuts, float* __restrict__ loss) {
    __shared__ float sl[1024];
    const int b = threadIdx.x;
    float best = 0.f;
    if (b < B) {
        float cost[MAXS * MAXS];
        for (int k = 0; k < S * S; ++k) {
            float a = 0.f, c = 0.f, d = 0.f;
            for (int ch = 0; ch < nc_sp; ++ch) {
                a += p_spec[((int64_t)b * nc_sp + ch) * S * S + k];
                c += p_phase[((int64_t)b * nc_sp + ch) * S * S + k];
            }
            for (int ch = 0; ch < nc_au; ++ch) d += p_audio[((int64_t)b * nc_au + ch) * S * S + k];
            // reference: (|ds| + |dp|).mean(F,T) + |da|.mean(L)
            cost[k] = (a + c) * inv_ft + d * inv_l;
        }
        int perm[MAXS], bestp[MAXS];
        for (int i = 0; i < S; ++i) { perm[i] = i; bestp[i] = i; }
        bool first = true;
        do {
            float tot = 0.f;
            for (int i = 0; i < S; ++i) tot += cost[i * S + perm[i]];
            if (first || tot < best) { best = tot; for (int i = 0; i < S; ++i) bestp[i] = perm[i]; }
            first = false;
        } while (!fixed && next_perm(perm, S));
        item_loss[b] = best;
        for (int i = 0; i < S; ++i) permuts[(int64_t)b * S + i] = bestp[i];
    }
    sl[threadIdx.x] = (b < B) ? best : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < B; ++i) s += sl[i];
        *loss = s / (float)B;
    }
}

// d pred[s][b][i] = scale * sign(pred - src[perm[b][s]])
template <int S>
__global__ __launch_bounds__(NT) void l1_bwd_kernel(const float* __restrict__ pred, SrcPtrs src,
                                                    const int64_t* __restrict__ permuts,
                                                    const float* __restrict__ gloss, float scale,
                                                    float* __restrict__ dpred, int B, int64_t n) {
    const int b = blockIdx.y;
    const float g = gloss[b] * scale;
    int pm[S];
#pragma unroll
    for (int s = 0; s < S; ++s) pm[s] = (int)permuts[(int64_t)b * S + s];
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        float sv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) sv[s] = src.p[s][(int64_t)b * n + i];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float t = sv[0];
#pragma unroll
            for (int c = 1; c < S; ++c) t = (pm[s] == c) ? sv[c] : t;
            float d = pred[((int64_t)s * B + b) * n + i] - t;
            dpred[((int64_t)s * B + b) * n + i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
        }
    }
}

// ---------------- SI-SNR: est[E][B][L] vs src_j[B][L] ----------------
// pass 1: dot[e][j][b] = <est_e, src_j>, tt[j][b] = |src_j|^2   (partial per chunk)
// pass 2: with alpha = dot/(tt+eps): num = sum (alpha t)^2, den = sum (p - alpha t)^2
template <int PASS>
__global__ __launch_bounds__(NT) void sisnr_kernel(const float* __restrict__ est, SrcPtrs src, int E, int S, int B,
                                                   int64_t L, int64_t chunk, const float* __restrict__ alpha,
                                                   float* __restrict__ partial) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.y, e = blockIdx.z / S, j = blockIdx.z % S;
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = min(L, i0 + chunk);
    const float* p = est + ((int64_t)e * B + b) * L;
    const float* t = src.p[j] + (int64_t)b * L;
    float a0 = 0.f, a1 = 0.f;
    const float al = PASS ? alpha[((int64_t)e * S + j) * B + b] : 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += NT) {
        float pv = p[i], tv = t[i];
        if (PASS == 0) { a0 += pv * tv; a1 += tv * tv; }
        else { float st = al * tv, en = pv - st; a0 += st * st; a1 += en * en; }
    }
    float s0 = block_sum(a0, red), s1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
        float* d = partial + ((((int64_t)e * S + j) * B + b) * gridDim.x + blockIdx.x) * 2;
        d[0] = s0; d[1] = s1;
    }
}

// MODE 0: alpha = dot/(tt+eps) ; MODE 1: out = 10 log10(num/(den+eps))
template <int MODE>
__global__ void sisnr_finalize(const float* __restrict__ partial, int n, int nchunks, float eps,
                               float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f, c = 0.f;
    for (int k = 0; k < nchunks; ++k) { a += partial[((int64_t)i * nchunks + k) * 2]; c += partial[((int64_t)i * nchunks + k) * 2 + 1]; }
    out[i] = MODE ? 10.f * log10f(a / (c + eps)) : a / (c + eps);
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

// partial buffers: p_* [B][nchunks][S*S].  kind: pred [S][B][n] vs S source planes [B][n]
extern "C" int dtt_l1_pairs(const float* pred, const float* const* src, float* partial, int S, int B, int64_t n,
                            int nchunks, void* stream) {
    DTT_CHECK(S >= 1 && S <= MAXS && B > 0 && n > 0 && nchunks > 0, "dtt_l1_pairs: bad shape S=%d", S);
    SrcPtrs sp{};
    for (int s = 0; s < S; ++s) sp.p[s] = src[s];
    int64_t chunk = (n + nchunks - 1) / nchunks;
    dim3 grid(nchunks, B);
    switch (S) {
        case 1: hipLaunchKernelGGL(l1_pair_kernel<1>, grid, dim3(NT), 0, ST, pred, sp, partial, B, n, chunk); break;
        case 2: hipLaunchKernelGGL(l1_pair_kernel<2>, grid, dim3(NT), 0, ST, pred, sp, partial, B, n, chunk); break;
        case 3: hipLaunchKernelGGL(l1_pair_kernel<3>, grid, dim3(NT), 0, ST, pred, sp, partial, B, n, chunk); break;
        default: hipLaunchKernelGGL(l1_pair_kernel<4>, grid, dim3(NT), 0, ST, pred, sp, partial, B, n, chunk); break;
    }
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_pit_select(const float* p_spec, const float* p_phase, const float* p_audio, int nc_sp, int nc_au,
                              int64_t FT, int64_t L, int S, int B, int fixed_order, float* item_loss,
                              int64_t* permuts, float* loss, void* stream) {
    DTT_CHECK(S >= 1 && S <= MAXS && B > 0 && B <= 1024, "dtt_pit_select: bad shape S=%d B=%d", S, B);
    hipLaunchKernelGGL(pit_select_kernel, dim3(1), dim3(1024), 0, ST, p_spec, p_phase, p_audio, nc_sp, nc_au,
                       1.f / (float)FT, 1.f / (float)L, S, B, fixed_order, item_loss, permuts, loss);
    DTT_LAUNCH_CHECK();
    return 0;
}

extern "C" int dtt_l1_bwd(const float* pred, const float* const* src, const int64_t* permuts, const float* gloss,
                          float scale, float* dpred, int S, int B, int64_t n, void* stream) {
    DTT_CHECK(S >= 1 && S <= MAXS && B > 0 && n > 0, "dtt_l1_bwd: bad shape");
    SrcPtrs sp{};
    for (int s = 0; s < S; ++s) sp.p[s] = src[s];
    int gx = (int)((n + NT - 1) / NT < 256 ? (n + NT - 1) / NT : 256);
    dim3 grid(gx, B);
    switch (S) {
        case 1: hipLaunchKernelGGL(l1_bwd_kernel<1>, grid, dim3(NT), 0, ST, pred, sp, permuts, gloss, scale, dpred, B, n); break;
        case 2: hipLaunchKernelGGL(l1_bwd_kernel<2>, grid, dim3(NT), 0, ST, pred, sp, permuts, gloss, scale, dpred, B, n); break;
        case 3: hipLaunchKernelGGL(l1_bwd_kernel<3>, grid, dim3(NT), 0, ST, pred, sp, permuts, gloss, scale, dpred, B, n); break;
        default: hipLaunchKernelGGL(l1_bwd_kernel<4>, grid, dim3(NT), 0, ST, pred, sp, permuts, gloss, scale, dpred, B, n); break;
    }
    DTT_LAUNCH_CHECK();
    return 0;
}

// est [E][B][L]; out[e][j][b] = SI-SNR(est_e, src_j) ; scratch: partial [E*S*B*nchunks*2], alpha [E*S*B]
extern "C" int dtt_sisnr(const float* est, const float* const* src, float* partial, float* alpha, float* out, int E,
                         int S, int B, int64_t L, int nchunks, float eps, void* stream) {
    DTT_CHECK(S >= 1 && S <= MAXS && E >= 1 && B > 0 && L > 0 && nchunks > 0 && E * S <= 65535,
              "dtt_sisnr: bad shape");
    SrcPtrs sp{};
    for (int s = 0; s < S; ++s) sp.p[s] = src[s];
    int64_t chunk = (L + nchunks - 1) / nchunks;
    dim3 grid(nchunks, B, E * S);
    const int n = E * S * B;
    hipLaunchKernelGGL(sisnr_kernel<0>, grid, dim3(NT), 0, ST, est, sp, E, S, B, L, chunk, (const float*)nullptr, partial);
    hipLaunchKernelGGL(sisnr_finalize<0>, dim3((n + 127) / 128), dim3(128), 0, ST, partial, n, nchunks, eps, alpha);
    hipLaunchKernelGGL(sisnr_kernel<1>, grid, dim3(NT), 0, ST, est, sp, E, S, B, L, chunk, alpha, partial);
    hipLaunchKernelGGL(sisnr_finalize<1>, dim3((n + 127) / 128), dim3(128), 0, ST, partial, n, nchunks, eps, out);
    DTT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// dtt_sisnr for callers that cannot allocate (HIP-graph capture regions, tight serving loops, harnesses
// that forbid hipMalloc): the chunk partials and the projection coefficients live in a file-scope device
// array instead of caller-provided scratch.  No counterpart in the reference.
//
// Each GPU holds its own copy of the array at its own address for the life of the process (1 MiB).
// The array's NAME is only a host-side stand-in: the device address is looked up once per device with
// hipGetSymbolAddress (that device current) and kept; kernels and every other use take that pointer.
// The kernels overwrite every word they later read, so nothing has to be cleared between calls; calls
// that overlap on one GPU (two streams) race on the array.
namespace {
constexpr int SISNR_SCRATCH_FLOATS = 256 * 1024;
static __device__ __align__(256) float g_sisnr_scratch[SISNR_SCRATCH_FLOATS];

constexpr int MAX_DEVICES = 64;
float* g_scratch_ptr[MAX_DEVICES];                       // per device: 0 until looked up
std::mutex g_scratch_mu;

// Device pointer of this device's copy of the scratch array (looked up once per device, that device current).
int sisnr_scratch_for_stream(hipStream_t st, float** out) {
    int dev = -1;
    if (st == nullptr || hipStreamGetDevice(st, &dev) != hipSuccess) {
        DTT_CHECK(hipGetDevice(&dev) == hipSuccess, "dtt_sisnr_noalloc: no current device");
    }
    DTT_CHECK(dev >= 0 && dev < MAX_DEVICES, "dtt_sisnr_noalloc: device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        int cur = -1;
        DTT_CHECK(hipGetDevice(&cur) == hipSuccess, "dtt_sisnr_noalloc: no current device");
        if (cur != dev) DTT_CHECK(hipSetDevice(dev) == hipSuccess, "dtt_sisnr_noalloc: cannot select device %d", dev);
        float* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_sisnr_scratch));
        if (cur != dev) (void)hipSetDevice(cur);
        DTT_CHECK(e == hipSuccess && p != nullptr, "dtt_sisnr_noalloc: hipGetSymbolAddress failed on device %d: %s", dev,
                  hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" int dtt_sisnr_noalloc_max_floats(void) { return SISNR_SCRATCH_FLOATS; }

// Same result as dtt_sisnr called with nchunks = max(1, min(32, L / 2048)) (what dtt_amd's metrics use),
// bit for bit: the same four launches, scratch = this GPU's static array.  E * S * B * (2 * nchunks + 1)
// floats must fit dtt_sisnr_noalloc_max_floats().
extern "C" int dtt_sisnr_noalloc(const float* est, const float* const* src, float* out, int E, int S, int B, int64_t L,
                                 float eps, void* stream) {
    DTT_CHECK(est && src && out, "dtt_sisnr_noalloc: null buffer");
    DTT_CHECK(S >= 1 && S <= MAXS && E >= 1 && B > 0 && L > 0 && E * S <= 65535, "dtt_sisnr_noalloc: bad shape");
    int nchunks = (int)(L / 2048);
    nchunks = nchunks < 1 ? 1 : (nchunks > 32 ? 32 : nchunks);
    const int64_t n = (int64_t)E * S * B;
    const int64_t need = n * (2 * nchunks + 1);
    DTT_CHECK(need <= SISNR_SCRATCH_FLOATS,
              "dtt_sisnr_noalloc: E*S*B*(2*nchunks+1) = %lld floats of scratch exceed the static %d (use dtt_sisnr with "
              "caller-provided scratch for this size)", (long long)need, SISNR_SCRATCH_FLOATS);
    float* scratch = nullptr;
    if (int rc = sisnr_scratch_for_stream(ST, &scratch)) return rc;
    float* partial = scratch;                              // [n][nchunks][2]
    float* alpha = scratch + n * 2 * nchunks;              // [n]
    SrcPtrs sp{};
    for (int s = 0; s < S; ++s) sp.p[s] = src[s];
    const int64_t chunk = (L + nchunks - 1) / nchunks;
    dim3 grid(nchunks, B, E * S);
    const int ni = (int)n;
    hipLaunchKernelGGL(sisnr_kernel<0>, grid, dim3(NT), 0, ST, est, sp, E, S, B, L, chunk, (const float*)nullptr, partial);
    hipLaunchKernelGGL(sisnr_finalize<0>, dim3((ni + 127) / 128), dim3(128), 0, ST, partial, ni, nchunks, eps, alpha);
    hipLaunchKernelGGL(sisnr_kernel<1>, grid, dim3(NT), 0, ST, est, sp, E, S, B, L, chunk, alpha, partial);
    hipLaunchKernelGGL(sisnr_finalize<1>, dim3((ni + 127) / 128), dim3(128), 0, ST, partial, ni, nchunks, eps, out);
    DTT_LAUNCH_CHECK();
    return 0;
}
