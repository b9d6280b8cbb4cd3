// MFMA GEMM for gfx950 with gather loaders (implicit-GEMM conv, conv wgrad,
// LSTM row shift) and a fused epilogue.  One kernel template, instantiated per
// (dtype, tile config).  64-wide wavefronts, 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32), LDS-staged operands with
// register prefetch of the next K-tile.
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

// conv tap geometry: source pixel of output pixel (fo,to) for tap (kh,kw)
__device__ __forceinline__ bool conv_src(const DttGemm& p, int fo, int to, int tap, int& fi, int& ti) {
    int kh = tap / 3, kw = tap - kh * 3;
    if (!p.ctrans) {
        fi = fo * p.cstride + kh - 1;
        ti = to * p.cstride + kw - 1;
        return fi >= 0 && fi < p.cFi && ti >= 0 && ti < p.cTi;
    }
    int a = fo + 1 - kh, b = to + 1 - kw;     // transposed conv, stride 2, pad 1
    fi = a >> 1; ti = b >> 1;
    return a >= 0 && b >= 0 && !(a & 1) && !(b & 1) && fi < p.cFi && ti < p.cTi;
}

// Operand tile loader.  The LDS image is always [ROWS][BK + PAD], k contiguous.
// A tile: rows = m.  B tile: rows = n.  `contig_k` = the operand is contiguous
// along k (vector loads along k), else contiguous along rows (vector loads along
// rows, transposing LDS writes); neither -> scalar loads.
template <typename T, int ROWS, int BK>
struct TileLoader {
    static constexpr int EPT = MM<T>::EPT;
    static constexpr int LDS_LD = BK + MM<T>::PAD;
    static constexpr int NCH = ROWS * BK / EPT;          // 16-byte chunks per tile
    static constexpr int NIT = (NCH + 255) / 256;
    Vec16<T> regs[NIT];

    // isA: operand A (row stride rs = a_ms, k stride ks = a_ks) else B (rs = b_ns, ks = b_ks).
    // Loads are issued UNCONDITIONALLY from a clamped address and zeroed afterwards (ld16_pred): a
    // branch per load makes hipcc wait vmcnt(0) after each one.  Only partial / misaligned edge
    // chunks take the scalar path.
    __device__ __forceinline__ void load(const DttGemm& p, bool isA, const T* base, int64_t rs, int64_t ks,
                                         int r0, int rmax, int k0, int kmax, int mode, int z_tap) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int idx = tid + it * 256;
            const bool live = idx < NCH;
            if (!live) idx = 0;
            const T* src = base;
            bool ok = false;            // element (gr, gk) of the chunk exists
            int nvalid = 0;             // how many of the EPT elements of the chunk exist
            int64_t estride = 1;        // stride between the chunk's elements (1: vectorisable)
            if (ks == 1 || (isA && mode == 1)) {
                // ---- k-contiguous: chunk = EPT consecutive k of one row
                int row = idx / (BK / EPT), ch = idx % (BK / EPT);
                int gr = r0 + row, gk = k0 + ch * EPT;
                ok = live && gr < rmax && gk < kmax;
                nvalid = min(EPT, kmax - gk);
                if (isA && mode == 1) {                // implicit-GEMM conv gather
                    int gkc = ok ? gk : 0, grc = ok ? gr : 0;
                    int tap = gkc / p.cCi, ci = gkc - tap * p.cCi;
                    int pix = grc;
                    int b = pix / (p.cFo * p.cTo); pix -= b * p.cFo * p.cTo;
                    int fo = pix / p.cTo, to = pix - fo * p.cTo;
                    int fi, ti;
                    ok = conv_src(p, fo, to, tap, fi, ti) && ok;
                    src = base + ((int64_t)(b * p.cFi + fi) * p.cTi + ti) * p.cCi + ci;
                } else {
                    src = base + (int64_t)gr * rs + gk;
                }
            } else if (rs == 1) {
                // ---- row-contiguous: chunk = EPT consecutive rows at one k
                int kl = idx / (ROWS / EPT), rc = idx % (ROWS / EPT);
                int gr = r0 + rc * EPT, gk = k0 + kl;
                ok = live && gr < rmax && gk < kmax;
                nvalid = min(EPT, rmax - gr);
                if (!isA && mode == 1) {               // conv wgrad: k = output pixel
                    int pix = ok ? gk : 0;
                    int b = pix / (p.cFo * p.cTo); pix -= b * p.cFo * p.cTo;
                    int fo = pix / p.cTo, to = pix - fo * p.cTo;
                    int fi, ti;
                    ok = conv_src(p, fo, to, z_tap, fi, ti) && ok;
                    src = base + ((int64_t)(b * p.cFi + fi) * p.cTi + ti) * p.cCi + gr;
                } else if (!isA && mode == 2) {        // LSTM h_{t-1}: shifted row
                    ok = ok && ((gk / p.sh_heads) % p.sh_L) != p.sh_edge;
                    src = base + (int64_t)(gk + p.sh_shift) * ks + gr;
                } else {
                    src = base + (int64_t)gk * ks + gr;
                }
            } else {
                // ---- generic strides: scalar loads, k-major chunk
                int row = idx / (BK / EPT), ch = idx % (BK / EPT);
                int gr = r0 + row, gk = k0 + ch * EPT;
                ok = live && gr < rmax && gk < kmax;
                nvalid = min(EPT, kmax - gk);
                src = base + (int64_t)gr * rs + (int64_t)gk * ks;
                estride = ks;
            }
            const bool fast = ok && estride == 1 && nvalid >= EPT && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
            Vec16<T> v = ld16_pred<T>(src, base_aligned(base), fast);
            if (ok && !fast) {
#pragma unroll
                for (int e = 0; e < EPT; ++e)
                    if (e < nvalid) v.set(e, ldf(src + (int64_t)e * estride));
            }
            regs[it] = v;
        }
    }
    static __device__ __forceinline__ const T* base_aligned(const T* b) {
        return reinterpret_cast<const T*>(reinterpret_cast<uintptr_t>(b) & ~(uintptr_t)15);
    }

    __device__ __forceinline__ void store(T* lds, bool isA, int64_t rs, int64_t ks, int mode) {
        const int tid = threadIdx.x;
        const bool kc = (ks == 1 || (isA && mode == 1)) || rs != 1;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int idx = tid + it * 256;
            if (idx < NCH) {
                if (kc) {
                    int row = idx / (BK / EPT), ch = idx % (BK / EPT);
                    st16<T>(lds + row * LDS_LD + ch * EPT, regs[it]);
                } else {
                    int kl = idx / (ROWS / EPT), rc = idx % (ROWS / EPT);
                    if constexpr (sizeof(T) == 2) {
                        // bf16: keep the natural [k][rows] image; fragments come from transposing reads
                        st16<T>(lds + kl * (ROWS + 8) + rc * EPT, regs[it]);
                    } else {
#pragma unroll
                        for (int e = 0; e < EPT; ++e)
                            *reinterpret_cast<float*>(lds + (rc * EPT + e) * LDS_LD + kl) = regs[it].get(e);
                    }
                }
            }
        }
    }
};

// WGM x WGN x WGK waves (product 4), each wave TM x TN tiles of 32x32.
template <typename T, int WGM, int WGN, int WGK, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(const DttGemm p) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, BK = 32 * WGK;
    constexpr int LD = BK + MM<T>::PAD;
    constexpr int ASZ = BM * LD > BK * (BM + 8) ? BM * LD : BK * (BM + 8);
    constexpr int BSZ = BN * LD > BK * (BN + 8) ? BN * LD : BK * (BN + 8);
    __shared__ __attribute__((aligned(16))) T lds[ASZ + BSZ];
    T* As = lds;
    T* Bs = lds + ASZ;
    // bf16 operands that are contiguous along rows (not k) sit in LDS as [k][rows] and are read transposed
    const bool a_tr = sizeof(T) == 2 && !(p.a_ks == 1 || p.a_mode == 1) && p.a_ms == 1;
    const bool b_tr = sizeof(T) == 2 && p.b_ks != 1 && p.b_ns == 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tm_idx = blockIdx.x / tiles_n, tn_idx = blockIdx.x % tiles_n;
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;
    const int z = blockIdx.y;
    const int bz = z / p.splitk, sk = z % p.splitk;

    // K range of this split (multiples of BK)
    const int ktiles = (p.K + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = sk * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;

    const T* A = reinterpret_cast<const T*>(p.A) + (p.a_mode == 1 ? 0 : (int64_t)bz * p.a_bs);
    const T* B = reinterpret_cast<const T*>(p.B) + (p.b_mode == 1 ? 0 : (int64_t)bz * p.b_bs);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wk = wave % WGK, wn = (wave / WGK) % WGN, wm = wave / (WGK * WGN);
    const int lr = lane & 31, lh = lane >> 5;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<T, BM, BK> la;
    TileLoader<T, BN, BK> lb;
    la.load(p, true, A, p.a_ms, p.a_ks, m0, p.M, kt0 * BK, p.K, p.a_mode, bz);
    lb.load(p, false, B, p.b_ns, p.b_ks, n0, p.N, kt0 * BK, p.K, p.b_mode, bz);

    for (int kt = kt0; kt < kt1; ++kt) {
        la.store(As, true, p.a_ms, p.a_ks, p.a_mode);
        lb.store(Bs, false, p.b_ns, p.b_ks, p.b_mode);
        __syncthreads();
        if (kt + 1 < kt1) {
            la.load(p, true, A, p.a_ms, p.a_ks, m0, p.M, (kt + 1) * BK, p.K, p.a_mode, bz);
            lb.load(p, false, B, p.b_ns, p.b_ks, n0, p.N, (kt + 1) * BK, p.K, p.b_mode, bz);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kofs = wk * 32 + s * 16 + lh * 8;
            typename MM<T>::Frag af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (sizeof(T) == 2) {
                    if (a_tr) {
                        const bf16* r0 = reinterpret_cast<const bf16*>(As) + (kofs + ((lane & 15) >> 2)) * (BM + 8) +
                                         (wm * TM + i) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                        af[i] = tr_frag(r0, r0 + 4 * (BM + 8));
                        continue;
                    }
                }
                af[i] = MM<T>::ldfrag(As + ((wm * TM + i) * 32 + lr) * LD + kofs);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (sizeof(T) == 2) {
                    if (b_tr) {
                        const bf16* r0 = reinterpret_cast<const bf16*>(Bs) + (kofs + ((lane & 15) >> 2)) * (BN + 8) +
                                         (wn * TN + j) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                        bf[j] = tr_frag(r0, r0 + 4 * (BN + 8));
                        continue;
                    }
                }
                bf[j] = MM<T>::ldfrag(Bs + ((wn * TN + j) * 32 + lr) * LD + kofs);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) MM<T>::mma(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cf = reinterpret_cast<float*>(p.C) + (int64_t)bz * p.c_bs;
    T* Ct = reinterpret_cast<T*>(p.C) + (int64_t)bz * p.c_bs;
    const T* R = p.R ? reinterpret_cast<const T*>(p.R) + (int64_t)bz * p.r_bs : nullptr;
    const bool first_split = (sk == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + (wn * TN + j) * 32 + lr;
            if (gn >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (gm >= p.M) continue;
                float v = acc[i][j][r] * p.alpha;
                if (p.atomic) {
                    if (first_split && wk == 0 && p.bias_mode) v += p.bias_mode == 1 ? p.bias[gn] : p.bias[gm];
                    atomicAdd(Cf + (int64_t)gm * p.c_ms + (int64_t)gn * p.c_ns, v);
                } else {
                    if (p.bias_mode == 1) v += p.bias[gn];
                    else if (p.bias_mode == 2) v += p.bias[gm];
                    if (p.act == 1) v = gelu_f(v);
                    else if (p.act == 2) v = sigmoid_f(v);
                    if (R) v += ldf(R + (int64_t)gm * p.r_ms + (int64_t)gn * p.r_ns);
                    const int64_t o = (int64_t)gm * p.c_ms + (int64_t)gn * p.c_ns;
                    if (p.out_f32) Cf[o] = v;
                    else stf(Ct + o, v);
                }
            }
        }
}

template <typename T, int WGM, int WGN, int WGK, int TM, int TN>
int launch(const DttGemm& p, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), p.batch * p.splitk);
    hipLaunchKernelGGL((gemm_kernel<T, WGM, WGN, WGK, TM, TN>), grid, dim3(256), 0, st, p);
    DTT_LAUNCH_CHECK();
    return 0;
}

template <typename T> int dispatch(const DttGemm& p, hipStream_t st) {
    if (p.atomic && p.M <= 32 && p.N <= 32) return launch<T, 1, 1, 4, 1, 1>(p, st);
    if (p.atomic && p.M <= 64 && p.N <= 64) return launch<T, 2, 2, 1, 1, 1>(p, st);
    if (p.N <= 32) return launch<T, 4, 1, 1, 1, 1>(p, st);
    if (p.M <= 32) return launch<T, 1, 4, 1, 1, 1>(p, st);
    if (p.N <= 64) return launch<T, 2, 2, 1, 2, 1>(p, st);
    if (p.M <= 64) return launch<T, 2, 2, 1, 1, 2>(p, st);
    return launch<T, 2, 2, 1, 2, 2>(p, st);
}

}  // namespace

extern "C" int dtt_gemm(const DttGemm* pp, void* stream) {
    DttGemm p = *pp;
    DTT_CHECK(p.A && p.B && p.C, "dtt_gemm: null operand");
    DTT_CHECK(p.M > 0 && p.N > 0 && p.K > 0 && p.batch > 0, "dtt_gemm: bad shape M=%d N=%d K=%d batch=%d",
              p.M, p.N, p.K, p.batch);
    DTT_CHECK(p.dtype == 0 || p.dtype == 1, "dtt_gemm: bad dtype %d", p.dtype);
    if (p.splitk < 1) p.splitk = 1;
    DTT_CHECK(p.splitk == 1 || p.atomic, "dtt_gemm: split-K needs atomic accumulation");
    DTT_CHECK(!p.atomic || (p.act == 0 && p.R == nullptr), "dtt_gemm: atomic mode has no act/residual");
    DTT_CHECK((int64_t)p.batch * p.splitk <= 65535, "dtt_gemm: grid.y too large");
    if (p.a_mode == 1) {
        DTT_CHECK(p.K == 9 * p.cCi && p.M == p.cB * p.cFo * p.cTo, "dtt_gemm: conv geometry mismatch (K=%d Ci=%d M=%d)",
                  p.K, p.cCi, p.M);
        DTT_CHECK(p.cCi % (p.dtype ? 8 : 4) == 0, "dtt_gemm: conv Ci=%d must be a multiple of 16 bytes", p.cCi);
        DTT_CHECK(!p.ctrans || p.cstride == 2, "dtt_gemm: transposed conv needs stride 2");
    }
    if (p.b_mode == 1) {
        DTT_CHECK(p.K == p.cB * p.cFo * p.cTo && p.batch == 9 && p.N == p.cCi && p.b_ns == 1,
                  "dtt_gemm: wgrad geometry mismatch");
        DTT_CHECK(!p.ctrans, "dtt_gemm: wgrad gather is always the strided (non-transposed) form");
    }
    if (p.b_mode == 2) DTT_CHECK(p.sh_heads > 0 && p.sh_L > 0 && p.b_ns == 1, "dtt_gemm: bad shift params");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return p.dtype == 1 ? dispatch<bf16>(p, st) : dispatch<float>(p, st);
}
