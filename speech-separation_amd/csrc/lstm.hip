// Persistent recurrent LSTM kernels (forward and backward-through-time) for the
// Improved Dual-Path modules (reference: src/model/DTTNet/modules/latent.py:87-92,
// nn.LSTM(C', 2C', bidirectional)).  One workgroup owns a block of NB sequences of
// one direction and walks all L steps itself: h lives in LDS, c in registers, the
// recurrent weights stream from L2 in MFMA-fragment order (packed by the host), the
// input projection x*W_ih^T (+biases) is precomputed by dtt_gemm.
//
// Row convention of every [rows][...] buffer: row = (outer*L + step)*heads + head;
// sequence n = outer*heads + head.
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f / (1.0f + __expf(2.0f * x)); }

struct LstmP {
    const void* xg;      // [rows][2][4H]  pre-activations from the input projection (T)
    const void* wp;      // [2][4H/32][H/16][64][8] packed W_hh (T)           (fwd)
    void* y;             // [rows][2H] hidden states (T)
    void* gates;         // [rows][2][4H] post-activation i,f,g,o (T)
    float* cst;          // [rows][2][H] cell states (f32)
    int N, L, heads, H;
};

// wave w owns hidden columns j in [(w*TJ)*32, (w*TJ+TJ)*32) for all four gates.
template <typename T, int TM, int TJ>
__global__ __launch_bounds__(512) void lstm_fwd_kernel(LstmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, LDH = H + MM<T>::PAD;
    constexpr int NB = 32 * TM;
    T* hbuf = reinterpret_cast<T*>(smem);                 // [2][NB][LDH]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int dir = blockIdx.y, n0 = blockIdx.x * NB;
    const int KS = H / 16, G4 = 4 * H;
    const T* xg = reinterpret_cast<const T*>(p.xg);
    const T* wp = reinterpret_cast<const T*>(p.wp) + (int64_t)dir * G4 * H;
    T* y = reinterpret_cast<T*>(p.y);
    T* gates = reinterpret_cast<T*>(p.gates);

    // per-lane sequence rows of the C layout: r -> local row (r&3)+8(r>>2)+4*lh
    float cst[TM][TJ][16];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) cst[a][b][r] = 0.f;

    for (int s = 0; s < p.L; ++s) {
        const int t = dir ? p.L - 1 - s : s;
        const int cur = s & 1;
        f32x16_t acc[TM][4][TJ];
        // ---- init accumulators with the input projection
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bool ok = n < p.N;
                const int64_t row = ((int64_t)(n / p.heads) * p.L + t) * p.heads + (n % p.heads);
                const T* src = xg + (row * 2 + dir) * G4;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int b = 0; b < TJ; ++b)
                        acc[a][g][b][r] = ok ? ldf(src + g * H + (wave * TJ + b) * 32 + lr) : 0.f;
            }
        // ---- recurrent term h_{t-1} * W_hh^T
        if (s > 0) {
            const T* hb = hbuf + cur * NB * LDH;
            for (int ks = 0; ks < KS; ++ks) {
                typename MM<T>::Frag af[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) af[a] = MM<T>::ldfrag(hb + (a * 32 + lr) * LDH + ks * 16 + lh * 8);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int b = 0; b < TJ; ++b) {
                        const int tile = g * (H / 32) + wave * TJ + b;
                        typename MM<T>::Frag bf = MM<T>::ldfrag(wp + (((int64_t)tile * KS + ks) * 64 + lane) * 8);
#pragma unroll
                        for (int a = 0; a < TM; ++a) MM<T>::mma(af[a], bf, acc[a][g][b]);
                    }
            }
        }
        // ---- gates, state update, stores
        T* hn = hbuf + (cur ^ 1) * NB * LDH;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int n = n0 + nl;
                const bool ok = n < p.N;
                const int64_t row = ((int64_t)(n / p.heads) * p.L + t) * p.heads + (n % p.heads);
#pragma unroll
                for (int b = 0; b < TJ; ++b) {
                    const int j = (wave * TJ + b) * 32 + lr;
                    float ig = sigmoid_f(acc[a][0][b][r]), fg = sigmoid_f(acc[a][1][b][r]);
                    float gg = tanh_f(acc[a][2][b][r]), og = sigmoid_f(acc[a][3][b][r]);
                    float c = fg * cst[a][b][r] + ig * gg;
                    cst[a][b][r] = c;
                    float h = og * tanh_f(c);
                    stf(hn + nl * LDH + j, h);
                    if (ok) {
                        stf(y + row * 2 * H + dir * H + j, h);
                        if (gates) {
                            T* gd = gates + (row * 2 + dir) * G4 + j;
                            stf(gd, ig); stf(gd + H, fg); stf(gd + 2 * H, gg); stf(gd + 3 * H, og);
                            p.cst[(row * 2 + dir) * H + j] = c;
                        }
                    }
                }
            }
        __syncthreads();
    }
}

struct LstmBwdP {
    const void* dy;      // [rows][2H] gradient wrt hidden outputs (T)
    const void* gates;   // [rows][2][4H] saved activations (T)
    const float* cst;    // [rows][2][H]
    const void* wpt;     // [2][H/32][4H/16][64][8] packed W_hh for dh = dG * W_hh (T)
    void* dG;            // [rows][2][4H] gate pre-activation gradients (T)
    int N, L, heads, H, kch;     // kch: gates staged in LDS per pass (4, 2 or 1)
};

template <typename T, int TM, int TJ>
__global__ __launch_bounds__(512) void lstm_bwd_kernel(LstmBwdP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, kch = p.kch, LDG = kch * H + MM<T>::PAD;
    constexpr int NB = 32 * TM;
    T* dgbuf = reinterpret_cast<T*>(smem);                // [NB][LDG]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int dir = blockIdx.y, n0 = blockIdx.x * NB;
    const int G4 = 4 * H, KS = G4 / 16;
    const T* dy = reinterpret_cast<const T*>(p.dy);
    const T* gates = reinterpret_cast<const T*>(p.gates);
    const T* wpt = reinterpret_cast<const T*>(p.wpt) + (int64_t)dir * G4 * H;
    T* dG = reinterpret_cast<T*>(p.dG);

    float dcs[TM][TJ][16];
    f32x16_t dhr[TM][TJ];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dcs[a][b][r] = 0.f; dhr[a][b][r] = 0.f; }

    for (int s = p.L - 1; s >= 0; --s) {
        const int t = dir ? p.L - 1 - s : s;
        const int tprev = dir ? t + 1 : t - 1;          // forward-pass predecessor step
        float dgv[TM][4][TJ][16];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int n = n0 + nl;
                const bool ok = n < p.N;
                const int64_t rb = (int64_t)(n / p.heads) * p.L, hd = n % p.heads;
                const int64_t row = (rb + t) * p.heads + hd, rowp = (rb + tprev) * p.heads + hd;
#pragma unroll
                for (int b = 0; b < TJ; ++b) {
                    const int j = (wave * TJ + b) * 32 + lr;
                    float di = 0.f, df = 0.f, dg = 0.f, dog_ = 0.f;
                    if (ok) {
                        const T* gs = gates + (row * 2 + dir) * G4 + j;
                        float ig = ldf(gs), fg = ldf(gs + H), gg = ldf(gs + 2 * H), og = ldf(gs + 3 * H);
                        float c = p.cst[(row * 2 + dir) * H + j];
                        float cp = (s > 0) ? p.cst[(rowp * 2 + dir) * H + j] : 0.f;
                        float dh = ldf(dy + row * 2 * H + dir * H + j) + dhr[a][b][r];
                        float tc = tanh_f(c);
                        dog_ = dh * tc * og * (1.f - og);
                        float dc = dcs[a][b][r] + dh * og * (1.f - tc * tc);
                        di = dc * gg * ig * (1.f - ig);
                        df = dc * cp * fg * (1.f - fg);
                        dg = dc * ig * (1.f - gg * gg);
                        dcs[a][b][r] = dc * fg;
                        T* gd = dG + (row * 2 + dir) * G4 + j;
                        stf(gd, di); stf(gd + H, df); stf(gd + 2 * H, dg); stf(gd + 3 * H, dog_);
                    }
                    dgv[a][0][b][r] = di; dgv[a][1][b][r] = df; dgv[a][2][b][r] = dg; dgv[a][3][b][r] = dog_;
                }
            }
        if (s == 0) break;
        // ---- dh_{t-1} = dG_t * W_hh, K = 4H staged through LDS kch gates at a time
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TJ; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dhr[a][b][r] = 0.f;
        for (int g0 = 0; g0 < 4; g0 += kch) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < g0 || g >= g0 + kch) continue;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TJ; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int nl = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            stf(dgbuf + nl * LDG + (g - g0) * H + (wave * TJ + b) * 32 + lr, dgv[a][g][b][r]);
                        }
            }
            __syncthreads();
            const int ks0 = g0 * H / 16, ks1 = (g0 + kch) * H / 16;
            for (int ks = ks0; ks < ks1; ++ks) {
                typename MM<T>::Frag af[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    af[a] = MM<T>::ldfrag(dgbuf + (a * 32 + lr) * LDG + (ks - ks0) * 16 + lh * 8);
#pragma unroll
                for (int b = 0; b < TJ; ++b) {
                    const int tile = wave * TJ + b;
                    typename MM<T>::Frag bf = MM<T>::ldfrag(wpt + (((int64_t)tile * KS + ks) * 64 + lane) * 8);
#pragma unroll
                    for (int a = 0; a < TM; ++a) MM<T>::mma(af[a], bf, dhr[a][b]);
                }
            }
            __syncthreads();
        }
    }
}

template <typename T, int TM, int TJ>
int launch_fwd(const LstmP& p, hipStream_t st) {
    const int nw = p.H / (32 * TJ), NB = 32 * TM;
    size_t sm = (size_t)2 * NB * (p.H + MM<T>::PAD) * sizeof(T);
    auto k = lstm_fwd_kernel<T, TM, TJ>;
    if (sm > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(k, dim3((p.N + NB - 1) / NB, 2), dim3(nw * 64), sm, st, p);
    DTT_LAUNCH_CHECK();
    return 0;
}

template <typename T, int TM, int TJ>
int launch_bwd(LstmBwdP p, hipStream_t st) {
    const int nw = p.H / (32 * TJ), NB = 32 * TM;
    int kch = 4;
    while (kch > 1 && (size_t)NB * (kch * p.H + MM<T>::PAD) * sizeof(T) > 150 * 1024) kch >>= 1;
    size_t sm = (size_t)NB * (kch * p.H + MM<T>::PAD) * sizeof(T);
    DTT_CHECK(sm <= 160 * 1024, "dtt_lstm_bwd: H=%d does not fit LDS", p.H);
    p.kch = kch;
    auto k = lstm_bwd_kernel<T, TM, TJ>;
    if (sm > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(k, dim3((p.N + NB - 1) / NB, 2), dim3(nw * 64), sm, st, p);
    DTT_LAUNCH_CHECK();
    return 0;
}

int pick_tj(int H) { return H > 256 ? 2 : 1; }

}  // namespace

extern "C" int dtt_lstm_fwd(const void* xg, const void* wp, void* y, void* gates, float* cst, int N, int L, int heads,
                            int H, int dtype, void* stream) {
    DTT_CHECK(H % 32 == 0 && H >= 32 && H <= 512 && N > 0 && L > 0 && heads > 0 && N % heads == 0,
              "dtt_lstm_fwd: unsupported shape H=%d N=%d L=%d", H, N, L);
    LstmP p{xg, wp, y, gates, cst, N, L, heads, H};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int tj = pick_tj(H);
    DTT_CHECK(H % (32 * tj) == 0 && H / (32 * tj) <= 8, "dtt_lstm_fwd: H=%d", H);
    const bool big = N >= 64 * 200 && (size_t)2 * 64 * (H + 8) * (dtype ? 2 : 4) <= 150 * 1024;
    if (dtype) {
        if (tj == 2) return launch_fwd<bf16, 1, 2>(p, st);
        return big ? launch_fwd<bf16, 2, 1>(p, st) : launch_fwd<bf16, 1, 1>(p, st);
    }
    if (tj == 2) return launch_fwd<float, 1, 2>(p, st);
    return big ? launch_fwd<float, 2, 1>(p, st) : launch_fwd<float, 1, 1>(p, st);
}

extern "C" int dtt_lstm_bwd(const void* dy, const void* gates, const float* cst, const void* wpt, void* dG, int N,
                            int L, int heads, int H, int dtype, void* stream) {
    DTT_CHECK(H % 32 == 0 && H >= 32 && H <= 512 && N > 0 && L > 0 && heads > 0 && N % heads == 0,
              "dtt_lstm_bwd: unsupported shape H=%d N=%d L=%d", H, N, L);
    LstmBwdP p{dy, gates, cst, wpt, dG, N, L, heads, H, 4};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int tj = pick_tj(H);
    DTT_CHECK(H % (32 * tj) == 0 && H / (32 * tj) <= 8, "dtt_lstm_bwd: H=%d", H);
    if (dtype) return tj == 2 ? launch_bwd<bf16, 1, 2>(p, st) : launch_bwd<bf16, 1, 1>(p, st);
    return tj == 2 ? launch_bwd<float, 1, 2>(p, st) : launch_bwd<float, 1, 1>(p, st);
}
