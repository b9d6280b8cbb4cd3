// MFMA GEMM for gfx950 with gather loaders (implicit-GEMM conv, conv wgrad,
// LSTM row shift) and a fused epilogue.  One kernel template, instantiated per
// (dtype, tile config).  64-wide wavefronts, 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32), LDS-staged operands with
// register prefetch of the next K-tile.
#include <type_traits>
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

// Output-parity form of the transposed conv (DttGemm::cphase): GEMM row gm -> output pixel, and the
// row of C (= pixel index in the full [B][cFo][cTo] plane) it is stored to.
__device__ __forceinline__ void phase_dims(const DttGemm& p, int& pf, int& pt, int& Fp, int& Tp) {
    const int ph = p.cphase - 1;
    pf = ph >> 1; pt = ph & 1;
    Fp = (p.cFo - pf + 1) >> 1; Tp = (p.cTo - pt + 1) >> 1;
}
__device__ __forceinline__ int64_t c_row(const DttGemm& p, int gm) {
    if (!p.cphase) return gm;
    int pf, pt, Fp, Tp;
    phase_dims(p, pf, pt, Fp, Tp);
    int b = gm / (Fp * Tp), r = gm - b * Fp * Tp;
    int i = r / Tp, jx = r - i * Tp;
    return ((int64_t)b * p.cFo + 2 * i + pf) * p.cTo + 2 * jx + pt;
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
                    int b, fi, ti;
                    if (p.cphase) {
                        int pf, pt, Fp, Tp;
                        phase_dims(p, pf, pt, Fp, Tp);
                        b = pix / (Fp * Tp); pix -= b * Fp * Tp;
                        const int i = pix / Tp, jx = pix - i * Tp;
                        const int nkw = pt ? 2 : 1, ih = tap / nkw, iw = tap - ih * nkw;
                        const int a = 2 * i + pf + 1 - (pf ? 2 * ih : 1), bb = 2 * jx + pt + 1 - (pt ? 2 * iw : 1);
                        fi = a >> 1; ti = bb >> 1;          // a, bb are even by construction
                        ok = ok && a >= 0 && bb >= 0 && fi < p.cFi && ti < p.cTi;
                    } else {
                        b = pix / (p.cFo * p.cTo); pix -= b * p.cFo * p.cTo;
                        int fo = pix / p.cTo, to = pix - fo * p.cTo;
                        ok = conv_src(p, fo, to, tap, fi, ti) && ok;
                    }
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

// Second half of the staged epilogue: one wave writes its parked 32 x (TN*32) fp32 slab as 16-byte
// chunks of EO outputs (EO = 8: bf16 C, EO = 4: fp32 C), adding the residual (type T) on the way.
template <typename T, int TN, int EO>
__device__ __forceinline__ void stage_out(const DttGemm& p, const float* stage, unsigned char* Cb, const T* R,
                                          int gm0, int gn0, int lane) {
    constexpr int SW = TN * 32 + 4, CPR = TN * 32 / EO, NIT = 32 * CPR / 64, OSZ = EO == 8 ? 2 : 4;
    constexpr bool RVEC = Vec16<T>::N == EO;            // residual chunk is one 16-byte vector too
    Vec16<T> rv[NIT];
    if (R) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + it * 64, row = idx / CPR, ch = idx % CPR;
            const int gm = gm0 + row, gn = gn0 + ch * EO;
            const T* rp = R + (int64_t)gm * p.r_ms + gn;
            const bool live = gm < p.M && gn < p.N;
            if constexpr (RVEC) {
                const bool rfast = live && gn + EO <= p.N && (reinterpret_cast<uintptr_t>(rp) & 15) == 0;
                rv[it] = ld16_pred<T>(rp, R, rfast);
                if (live && !rfast) {
                    for (int e = 0; e < EO; ++e)
                        if (gn + e < p.N) rv[it].set(e, ldf(rp + e));
                }
            } else {
                rv[it] = zero16<T>();
#pragma unroll
                for (int e = 0; e < EO; ++e) {
                    const bool okr = live && gn + e < p.N;
                    const float x = ldf(okr ? rp + e : R);
                    rv[it].set(e, okr ? x : 0.f);
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = lane + it * 64, row = idx / CPR, ch = idx % CPR;
        const int gm = gm0 + row, gn = gn0 + ch * EO;
        const int nv = min(EO, p.N - gn);
        const bool live = gm < p.M && nv > 0;
        float v[EO];
#pragma unroll
        for (int q = 0; q < EO / 4; ++q) {
            const float4 s = *reinterpret_cast<const float4*>(stage + row * SW + ch * EO + q * 4);
            v[q * 4] = s.x; v[q * 4 + 1] = s.y; v[q * 4 + 2] = s.z; v[q * 4 + 3] = s.w;
        }
        if (R) {
#pragma unroll
            for (int e = 0; e < EO; ++e) v[e] += rv[it].get(e);
        }
        unsigned char* dst = Cb + (c_row(p, gm) * p.c_ms + gn) * OSZ;
        const bool cfast = live && nv == EO && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
        if (cfast) {
            if constexpr (EO == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                Vec16<bf16> ov;
#pragma unroll
                for (int e = 0; e < 8; e += 2) ov.set2(e, v[e], v[e + 1]);
                st16<bf16>(reinterpret_cast<bf16*>(dst), ov);
            }
        } else if (live) {
            for (int e = 0; e < nv; ++e) {
                if constexpr (EO == 4) reinterpret_cast<float*>(dst)[e] = v[e];
                else stf(reinterpret_cast<bf16*>(dst) + e, v[e]);
            }
        }
    }
}

// Fused epilogue shared by the GEMM kernels: alpha, bias, activation, residual, store / atomic add.
// Must be entered by all waves after a barrier (it reuses the operand LDS as staging).
template <int ACT> __device__ __forceinline__ float apply_act(float v) {
    if (ACT == 1) return gelu_f(v);
    if (ACT == 2) return sigmoid_f(v);
    return v;
}

// FASTONLY: the caller guarantees c_ns == 1 and r_ns == 1 (host check), so the element-wise
// fallback is not even compiled in.
template <typename T, int WGM, int WGN, int WGK, int TM, int TN, int BM, int BK, bool FASTONLY = false>
__device__ __forceinline__ void gemm_epilogue(const DttGemm& p, f32x16_t (&acc)[TM][TN], unsigned char* lds_raw,
                                              int m0, int n0, int bz, int sk) {
    constexpr int SW = TN * 32 + 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wk = wave % WGK, wn = (wave / WGK) % WGN, wm = wave / (WGK * WGN);
    const int lr = lane & 31, lh = lane >> 5;
    // ---- epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cf = reinterpret_cast<float*>(p.C) + (int64_t)bz * p.c_bs;
    T* Ct = reinterpret_cast<T*>(p.C) + (int64_t)bz * p.c_bs;
    const T* R = p.R ? reinterpret_cast<const T*>(p.R) + (int64_t)bz * p.r_bs : nullptr;
    const bool first_split = (sk == 0);
    if constexpr (WGK == 1) {
        // ---- staged epilogue (plain stores, n-contiguous C): each wave parks a 32 x (TN*32) slab of
        // its accumulators (bias and activation already applied: a lane owns one column) in LDS and
        // writes it back out as 16-byte chunks along n, with the residual read the same way.  (The
        // scalar path below costs one 2-byte store per element.)
        if (!p.atomic && (FASTONLY || (p.c_ns == 1 && (R == nullptr || p.r_ns == 1)))) {
            float* stage = reinterpret_cast<float*>(lds_raw) + wave * 32 * SW;
            const bool o4 = p.out_f32 || sizeof(T) == 4;
            unsigned char* Cb = reinterpret_cast<unsigned char*>(p.C) + (int64_t)bz * p.c_bs * (o4 ? 4 : 2);
            const float* bp = p.bias_mode ? p.bias : reinterpret_cast<const float*>(&dtt_zero16);
            float bn[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = n0 + (wn * TN + j) * 32 + lr;
                const float b = bp[(p.bias_mode == 1 && gn < p.N) ? gn : 0];
                bn[j] = p.bias_mode == 1 ? b : 0.f;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float bm[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float b = bp[(p.bias_mode == 2 && gm < p.M) ? gm : 0];
                    bm[r] = p.bias_mode == 2 ? b : 0.f;
                }
                auto park = [&](auto act_tag) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SW + j * 32 + lr] =
                                apply_act<decltype(act_tag)::value>(acc[i][j][r] * p.alpha + bn[j] + bm[r]);
                };
                if (p.act == 0) park(std::integral_constant<int, 0>{});
                else if (p.act == 1) park(std::integral_constant<int, 1>{});
                else park(std::integral_constant<int, 2>{});
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int gm0 = m0 + (wm * TM + i) * 32, gn0 = n0 + wn * TN * 32;
                if (o4) stage_out<T, TN, 4>(p, stage, Cb, R, gm0, gn0, lane);
                else stage_out<T, TN, 8>(p, stage, Cb, R, gm0, gn0, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            return;
        }
    }
    if constexpr (FASTONLY) {
        // atomic accumulation (split-K partial sums), n-contiguous fp32 C
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = n0 + (wn * TN + j) * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[i][j][r] * p.alpha;
                    if (gn < p.N && gm < p.M) atomicAdd(Cf + (int64_t)gm * p.c_ms + gn, v);
                }
            }
        return;
    }
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
                    const int64_t o = c_row(p, gm) * p.c_ms + (int64_t)gn * p.c_ns;
                    if (p.out_f32) Cf[o] = v;
                    else stf(Ct + o, v);
                }
            }
        }
}

// WGM x WGN x WGK waves (product 4), each wave TM x TN tiles of 32x32.
template <typename T, int WGM, int WGN, int WGK, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(const DttGemm p) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, BK = 32 * WGK;
    constexpr int LD = BK + MM<T>::PAD;
    constexpr int ASZ = BM * LD > BK * (BM + 8) ? BM * LD : BK * (BM + 8);
    constexpr int BSZ = BN * LD > BK * (BN + 8) ? BN * LD : BK * (BN + 8);
    constexpr int SW = TN * 32 + 4;                       // staged-epilogue row (floats), see below
    constexpr int OPB = (ASZ + BSZ) * (int)sizeof(T), STB = WGK == 1 ? 4 * 32 * SW * 4 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[OPB > STB ? OPB : STB];
    T* As = reinterpret_cast<T*>(lds_raw);
    T* Bs = As + ASZ;
    // bf16 operands that are contiguous along rows (not k) sit in LDS as [k][rows] and are read transposed
    const bool a_tr = sizeof(T) == 2 && !(p.a_ks == 1 || p.a_mode == 1) && p.a_ms == 1;
    const bool b_tr = sizeof(T) == 2 && p.b_ks != 1 && p.b_ns == 1;

    // XCD-aware workgroup order: the hardware deals consecutive workgroups round-robin to the 8 XCDs
    // (one L2 each).  Renumber so that consecutive LOGICAL ids -- the tiles of one batch / K-split,
    // which share operand panels -- run on the same XCD at the same time.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntile = gridDim.x;
    int tile = blockIdx.x, z = blockIdx.y;
    {
        const int64_t total = (int64_t)ntile * gridDim.y;
        const int64_t lin = (int64_t)blockIdx.y * ntile + blockIdx.x;
        const int64_t per = total / 8;
        if (lin < per * 8) {
            const int64_t lg = (lin % 8) * per + lin / 8;
            tile = (int)(lg % ntile);
            z = (int)(lg / ntile);
        }
    }
    const int tm_idx = tile / tiles_n, tn_idx = tile % tiles_n;
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;
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

    gemm_epilogue<T, WGM, WGN, WGK, TM, TN, BM, BK>(p, acc, lds_raw, m0, n0, bz, sk);
}

// ---------------------------------------------------------------------------------------------
// Fast path: operand layouts fixed at compile time (KC: contiguous along k, RC: contiguous along
// rows, RCS: RC with the LSTM row shift), every chunk 16-byte aligned (checked on the host), per-
// thread source pointers computed once and advanced by a constant per K-tile.  BK = 64 (bf16) so
// that k-contiguous rows are read 128 bytes at a time.
enum { OP_KC = 0, OP_RC = 1, OP_RCS = 2, OP_CONV = 3, OP_WG = 4 };   // OP_WG: B rows gathered at conv tap (wgrad)     // OP_CONV: A rows = output pixels of a 3x3 conv (gather)

template <typename T, int KIND, int ROWS, int BK>
struct FastLoader {
    static constexpr int EPT = MM<T>::EPT;
    static constexpr int LD = BK + MM<T>::PAD;              // KC image [ROWS][LD]
    static constexpr int LDR = ROWS + 8;                    // RC bf16 image [BK][LDR]
    static constexpr int NCH = ROWS * BK / EPT;
    static constexpr int NIT = NCH / 256;
    static_assert(NCH % 256 == 0, "tile must be a multiple of 256 chunks");
    const T* src[NIT];
    const T* safe;
    int kofs[NIT];          // k offset of the chunk inside the tile
    bool rok[NIT];
    int64_t step;           // pointer advance per K-tile
    Vec16<T> regs[NIT];
    int nv[NIT];            // KC with K not a multiple of the chunk: valid elements of the chunk just loaded
    bool ragged;
    int pfo[NIT], pto[NIT]; // OP_CONV: output pixel of the chunk's row
    static constexpr bool KCL = KIND == OP_KC || KIND == OP_CONV;      // k-contiguous LDS image

    __device__ __forceinline__ void init(const T* base, int64_t rs, int64_t ks, int r0, int rmax, int kbeg, int64_t shift) {
        safe = base;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (KCL) {
                const int row = idx / (BK / EPT), ch = idx % (BK / EPT);
                rok[it] = r0 + row < rmax;
                kofs[it] = ch * EPT;
                src[it] = base + (int64_t)(rok[it] ? r0 + row : 0) * rs + kbeg + ch * EPT;
            } else {
                const int kl = idx / (ROWS / EPT), rc = idx % (ROWS / EPT);
                rok[it] = r0 + rc * EPT < rmax;
                kofs[it] = kl;
                src[it] = base + ((int64_t)(kbeg + kl) + shift) * ks + (rok[it] ? r0 + rc * EPT : 0);
            }
        }
        step = KCL ? BK : (int64_t)BK * ks;
        ragged = false;
    }
    // OP_CONV: decode each chunk's output pixel once; src[] = base of its batch item
    __device__ __forceinline__ void init_conv(const DttGemm& p, const T* base, int r0, int rmax) {
        safe = base;
        ragged = false;
        step = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int row = idx / (BK / EPT), ch = idx % (BK / EPT);
            rok[it] = r0 + row < rmax;
            kofs[it] = ch * EPT;
            int pix = rok[it] ? r0 + row : 0, b;
            if (p.cphase) {
                int pf, pt, Fp, Tp;
                phase_dims(p, pf, pt, Fp, Tp);
                b = pix / (Fp * Tp); pix -= b * Fp * Tp;
                const int i = pix / Tp;
                pfo[it] = 2 * i + pf; pto[it] = 2 * (pix - i * Tp) + pt;
            } else {
                b = pix / (p.cFo * p.cTo); pix -= b * p.cFo * p.cTo;
                pfo[it] = pix / p.cTo; pto[it] = pix - pfo[it] * p.cTo;
            }
            src[it] = base + (int64_t)b * p.cFi * p.cTi * p.cCi;
            nv[it] = EPT;
        }
    }
    // OP_WG: k = output pixel of the conv, rows = input channels of the pixel that tap `wtap` reads
    int wtap, wrow[NIT], wkh[NIT];
    __device__ __forceinline__ void init_wg(const T* base, int r0, int rmax, int tap, int runlen) {
        safe = base; ragged = false; step = 0; wtap = tap;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int kl = idx / (ROWS / EPT), rc = idx % (ROWS / EPT);
            rok[it] = r0 + rc * EPT < rmax;
            kofs[it] = kl;
            wrow[it] = rok[it] ? r0 + rc * EPT : 0;
            wkh[it] = 0;
            if (runlen) { wkh[it] = wrow[it] / runlen; wrow[it] -= wkh[it] * runlen; }   // general geometry: n -> (kh, run offset)
            src[it] = base;
        }
    }
    __device__ __forceinline__ void load_wg(const DttGemm& p, int k0, int kmax) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int gk = k0 + kofs[it];
            bool ok = rok[it] && gk < kmax;
            int pix = ok ? gk : 0;
            const int b = pix / (p.cFo * p.cTo); pix -= b * p.cFo * p.cTo;
            const int fo = pix / p.cTo, to = pix - fo * p.cTo;
            int fi, ti;
            if (p.cKW) {
                fi = fo * p.cstride + wkh[it] - p.cPadF; ti = to * p.cstride - p.cPadT;
                ok = ok && fi >= 0 && fi < p.cFi && ti >= 0 && ti + p.cKW <= p.cTi;
            } else {
                ok = conv_src(p, fo, to, wtap, fi, ti) && ok;
            }
            regs[it] = ld16_pred<T>(src[it] + ((int64_t)(b * p.cFi + fi) * p.cTi + ti) * p.cCi + wrow[it], safe, ok);
        }
    }
    __device__ __forceinline__ void load_conv(const DttGemm& p, int k0, int kmax) {
        int pf = 0, pt = 0;
        if (p.cphase) { pf = (p.cphase - 1) >> 1; pt = (p.cphase - 1) & 1; }
        // channel counts are powers of two in every DTTNet configuration: no integer division per chunk
        const bool ci_pow2 = (p.cCi & (p.cCi - 1)) == 0;
        const int ci_sh = __ffs(p.cCi) - 1;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int gk = k0 + kofs[it];
            bool ok = rok[it] && gk < kmax;
            const int tap = ci_pow2 ? gk >> ci_sh : gk / p.cCi, ci = gk - tap * p.cCi;
            int kh, kw;
            if (p.cphase) { const int nkw = pt ? 2 : 1, ih = tap / nkw; kh = pf ? 2 * ih : 1; kw = pt ? 2 * (tap - ih * nkw) : 1; }
            else if (p.cKW) { kh = tap / p.cKW; kw = tap - kh * p.cKW; }
            else { kh = tap / 3; kw = tap - kh * 3; }
            int fi, ti;
            if (!p.ctrans) {
                const int padf = p.cKW ? p.cPadF : 1, padt = p.cKW ? p.cPadT : 1;
                fi = pfo[it] * p.cstride + kh - padf; ti = pto[it] * p.cstride + kw - padt;
                ok = ok && fi >= 0 && fi < p.cFi && ti >= 0 && ti < p.cTi;
            } else {
                const int a = pfo[it] + 1 - kh, bb = pto[it] + 1 - kw;
                fi = a >> 1; ti = bb >> 1;
                ok = ok && a >= 0 && bb >= 0 && !(a & 1) && !(bb & 1) && fi < p.cFi && ti < p.cTi;
            }
            regs[it] = ld16_pred<T>(src[it] + ((int64_t)fi * p.cTi + ti) * p.cCi + ci, safe, ok);
        }
    }
    // k0: global k of the tile start; kmax: exclusive k bound
    __device__ __forceinline__ void load(const DttGemm& p, int k0, int kmax) {
        if constexpr (KIND == OP_CONV) { load_conv(p, k0, kmax); return; }
        if constexpr (KIND == OP_WG) { load_wg(p, k0, kmax); return; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int gk = k0 + kofs[it];
            bool ok = rok[it] && gk < kmax;
            if (KIND == OP_RCS) ok = ok && ((gk / p.sh_heads) % p.sh_L) != p.sh_edge;
            if (KIND == OP_KC) nv[it] = kmax - gk;          // >= EPT except in a ragged last chunk (reads row padding)
            regs[it] = ld16_pred<T>(src[it], safe, ok);
            src[it] += step;
        }
    }
    __device__ __forceinline__ void store(T* lds) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (KCL) {
                const int row = idx / (BK / EPT), ch = idx % (BK / EPT);
                if (ragged && nv[it] < EPT) {               // zero what lies beyond K (padding may hold anything)
#pragma unroll
                    for (int e = 0; e < EPT; ++e)
                        if (e >= nv[it]) regs[it].set(e, 0.f);
                }
                st16<T>(lds + row * LD + ch * EPT, regs[it]);
            } else {
                const int kl = idx / (ROWS / EPT), rc = idx % (ROWS / EPT);
                if constexpr (sizeof(T) == 2) {
                    st16<T>(lds + kl * LDR + rc * EPT, regs[it]);
                } else {
#pragma unroll
                    for (int e = 0; e < EPT; ++e) *reinterpret_cast<float*>(lds + (rc * EPT + e) * LD + kl) = regs[it].get(e);
                }
            }
        }
    }
    static constexpr int SIZE = (KCL || sizeof(T) == 4) ? ROWS * LD : BK * LDR;
    // MFMA fragment of the 32-row block rb at k offset kk (a multiple of 16) of the tile
    static __device__ __forceinline__ typename MM<T>::Frag frag(const T* lds, int rb, int kk, int lane) {
        if constexpr (sizeof(T) == 2 && !KCL) {
            const bf16* r0 = reinterpret_cast<const bf16*>(lds) + (kk + 8 * (lane >> 5) + ((lane & 15) >> 2)) * LDR + rb * 32 +
                             16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            return tr_frag(r0, r0 + 4 * LDR);
        } else {
            return MM<T>::ldfrag(lds + (rb * 32 + (lane & 31)) * LD + kk + 8 * (lane >> 5));
        }
    }
};

// three workgroups per CU: the 128x128 tile needs 138-168 VGPRs, and the compiler's default budget (256)
// leaves it at two; 170 registers cost nothing (no spills) and hide more of the load -> LDS -> MFMA chain
#ifndef GEMM_MINB
#define GEMM_MINB 3
#endif
template <typename T, int AK, int BKD, int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256, GEMM_MINB) void gemm_fast_kernel(const DttGemm p) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, BK = sizeof(T) == 2 ? 64 : 32;
    typedef FastLoader<T, AK, BM, BK> LA;
    typedef FastLoader<T, BKD, BN, BK> LB;
    constexpr int SW = TN * 32 + 4;
    constexpr int OPB = (LA::SIZE + LB::SIZE) * (int)sizeof(T), STB = 4 * 32 * SW * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[OPB > STB ? OPB : STB];
    T* As = reinterpret_cast<T*>(lds_raw);
    T* Bs = As + LA::SIZE;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntile = gridDim.x;
    int tile = blockIdx.x, z = blockIdx.y;
    {   // XCD-aware order (see gemm_kernel)
        const int64_t total = (int64_t)ntile * gridDim.y;
        const int64_t lin = (int64_t)blockIdx.y * ntile + blockIdx.x;
        const int64_t per = total / 8;
        if (lin < per * 8) {
            const int64_t lg = (lin % 8) * per + lin / 8;
            tile = (int)(lg % ntile);
            z = (int)(lg / ntile);
        }
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int bz = z / p.splitk, sk = z % p.splitk;
    const int ktiles = (p.K + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = sk * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wn = wave % WGN, wm = wave / WGN;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LA la;
    LB lb;
    if constexpr (AK == OP_CONV) la.init_conv(p, reinterpret_cast<const T*>(p.A), m0, p.M);
    else la.init(reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.a_bs, p.a_ms, p.a_ks, m0, p.M, kt0 * BK, 0);
    if constexpr (BKD == OP_WG) lb.init_wg(reinterpret_cast<const T*>(p.B), n0, p.N, bz, p.cKW ? p.cKW * p.cCi : 0);
    else lb.init(reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.b_bs, p.b_ns, p.b_ks, n0, p.N, kt0 * BK,
                 BKD == OP_RCS ? p.sh_shift : 0);
    la.ragged = lb.ragged = (p.K % MM<T>::EPT) != 0;
    la.load(p, kt0 * BK, p.K);
    lb.load(p, kt0 * BK, p.K);
    for (int kt = kt0; kt < kt1; ++kt) {
        la.store(As);
        lb.store(Bs);
        __syncthreads();
        if (kt + 1 < kt1) {
            la.load(p, (kt + 1) * BK, p.K);
            lb.load(p, (kt + 1) * BK, p.K);
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            typename MM<T>::Frag af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = LA::frag(As, wm * TM + i, s * 16, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = LB::frag(Bs, wn * TN + j, s * 16, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) MM<T>::mma(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }
    gemm_epilogue<T, WGM, WGN, 1, TM, TN, BM, BK, true>(p, acc, lds_raw, m0, n0, bz, sk);
}

template <typename T, int AK, int BKD, int WGM, int WGN, int TM, int TN>
int launch_fast(const DttGemm& p, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), p.batch * p.splitk);
    hipLaunchKernelGGL((gemm_fast_kernel<T, AK, BKD, WGM, WGN, TM, TN>), grid, dim3(256), 0, st, p);
    DTT_LAUNCH_CHECK();
    return 0;
}

template <typename T, int AK, int BKD> int dispatch_fast_tile(const DttGemm& p, hipStream_t st) {
    // Small products (the fp32 token network of the cross-attention fusion: 800 x 256 x 256 and the like) leave most
    // CUs idle at 128 x 128 tiles (4-16 workgroups, 35-57 us each on the fp32 MFMA path): with fewer than half a
    // workgroup per CU take the 32 x 128 or 128 x 32 tile, whichever gives more workgroups (a quarter of the work
    // per workgroup)
    auto cd = [](int a, int b) { return (int64_t)((a + b - 1) / b); };
    const int64_t zs = (int64_t)p.batch * p.splitk;
    if (p.M > 64 && p.N > 64 && cd(p.M, 128) * cd(p.N, 128) * zs < 128) {
        if (cd(p.M, 32) * cd(p.N, 128) >= cd(p.M, 128) * cd(p.N, 32)) return launch_fast<T, AK, BKD, 1, 4, 1, 1>(p, st);
        return launch_fast<T, AK, BKD, 4, 1, 1, 1>(p, st);
    }
    if (p.M <= 32 && p.N > 32) return launch_fast<T, AK, BKD, 1, 4, 1, 1>(p, st);
    if (p.M <= 64 && p.N > 64) return launch_fast<T, AK, BKD, 1, 4, 2, 1>(p, st);
    if (p.N <= 32) return launch_fast<T, AK, BKD, 4, 1, 1, 1>(p, st);
    if (p.N <= 64) return launch_fast<T, AK, BKD, 2, 2, 2, 1>(p, st);
    return launch_fast<T, AK, BKD, 2, 2, 2, 2>(p, st);
}

// returns 1 if the problem does not qualify for the fast path
template <typename T> int dispatch_fast(const DttGemm& p, hipStream_t st) {
    constexpr int E = MM<T>::EPT;
    if (p.a_mode == 1) {
        // implicit-GEMM conv rows (stride-2 / transposed convs): weights k-contiguous
        if (p.b_mode != 0 || p.batch != 1 || p.b_ks != 1 || p.b_ns % E != 0 || p.cCi % E != 0 || p.c_ns != 1 ||
            (p.R && p.r_ns != 1) || (p.atomic && p.bias_mode))
            return 1;
        if ((reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.B) & 15)) return 1;
        return dispatch_fast_tile<T, OP_CONV, OP_KC>(p, st);
    }
    if (p.a_mode == 0 && p.b_mode == 1) {
        // conv weight gradient: A = dy (or x) [pixel][M] row-contiguous, B gathered at the tap = batch index
        if (!(p.a_ms == 1 && p.a_ks % E == 0 && (p.M % E == 0 || p.a_ks >= (p.M + E - 1) / E * E) && p.a_bs == 0 &&
              p.cCi % E == 0 && p.N % E == 0 && p.c_ns == 1 && p.atomic && !p.bias_mode))
            return 1;
        if ((reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.B) & 15)) return 1;
        return dispatch_fast_tile<T, OP_RC, OP_WG>(p, st);
    }
    if (p.a_mode != 0 || (p.b_mode != 0 && p.b_mode != 2)) return 1;
    if (p.c_ns != 1 || (p.R && p.r_ns != 1) || (p.atomic && p.bias_mode)) return 1;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    auto ru = [](int v) { return (int64_t)((v + E - 1) / E * E); };
    if (!al(p.A) || !al(p.B)) return 1;
    int ak, bk;
    const bool kfull = p.K % E == 0;                       // else the ragged last chunk must lie in row padding
    if (p.a_ks == 1 && p.a_ms % E == 0 && (kfull || p.a_ms >= ru(p.K))) ak = OP_KC;
    else if (p.a_ms == 1 && p.a_ks % E == 0 && (p.M % E == 0 || p.a_ks >= ru(p.M))) ak = OP_RC;   // ragged: reads row padding
    else return 1;
    if (p.batch > 1 && (p.a_bs % E != 0 || p.b_bs % E != 0)) return 1;
    if (p.b_mode == 2) {
        if (!(p.b_ns == 1 && p.b_ks % E == 0 && (p.N % E == 0 || p.b_ks >= ru(p.N)))) return 1;
        bk = OP_RCS;
    } else if (p.b_ks == 1 && p.b_ns % E == 0 && (kfull || p.b_ns >= ru(p.K))) bk = OP_KC;
    else if (p.b_ns == 1 && p.b_ks % E == 0 && (p.N % E == 0 || p.b_ks >= ru(p.N))) bk = OP_RC;
    else return 1;
    if (ak == OP_KC && bk == OP_KC) return dispatch_fast_tile<T, OP_KC, OP_KC>(p, st);
    if (ak == OP_KC && bk == OP_RC) return dispatch_fast_tile<T, OP_KC, OP_RC>(p, st);
    if (ak == OP_RC && bk == OP_RC) return dispatch_fast_tile<T, OP_RC, OP_RC>(p, st);
    if (ak == OP_RC && bk == OP_RCS) return dispatch_fast_tile<T, OP_RC, OP_RCS>(p, st);
    return 1;
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

int dtt_gemm256_try(const DttGemm& p, hipStream_t st);   // gemm256.hip
int dtt_gemmx_try(const DttGemm& p, hipStream_t st);     // gemmx.hip
int64_t dtt_gemm256_ws_bytes(const DttGemm& p);

extern "C" int64_t dtt_gemm_ws_bytes(const DttGemm* p) { return p ? dtt_gemm256_ws_bytes(*p) : 0; }

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
    if (p.a_mode == 1 && p.cphase) {
        const int pf = (p.cphase - 1) >> 1, pt = (p.cphase - 1) & 1;
        // (a residual R is indexed by the COMPACT GEMM row -- [M = B * Fp * Tp][N], the phase's own pixel order --
        // while C is scattered to the phase's pixels of the full plane: both epilogues do exactly that)
        DTT_CHECK(p.cphase >= 1 && p.cphase <= 4 && p.ctrans && p.cstride == 2 && !p.atomic,
                  "dtt_gemm: cphase needs a plain-store transposed stride-2 conv");
        DTT_CHECK(p.K == (pf ? 2 : 1) * (pt ? 2 : 1) * p.cCi &&
                      p.M == p.cB * ((p.cFo - pf + 1) / 2) * ((p.cTo - pt + 1) / 2),
                  "dtt_gemm: parity-phase geometry mismatch (K=%d M=%d)", p.K, p.M);
        DTT_CHECK(p.cCi % (p.dtype ? 8 : 4) == 0, "dtt_gemm: conv Ci=%d must be a multiple of 16 bytes", p.cCi);
    } else if (p.a_mode == 1 && p.cKW) {
        DTT_CHECK(p.cKW > 0 && p.cCi > 0 && p.K % (p.cKW * p.cCi) == 0 && p.M == p.cB * p.cFo * p.cTo && !p.ctrans &&
                      p.cPadF >= 0 && p.cPadT >= 0 && p.cstride >= 1,
                  "dtt_gemm: general conv geometry mismatch (K=%d KW=%d Ci=%d M=%d)", p.K, p.cKW, p.cCi, p.M);
        DTT_CHECK(p.cCi % (p.dtype ? 8 : 4) == 0, "dtt_gemm: conv Ci=%d must be a multiple of 16 bytes", p.cCi);
    } else if (p.a_mode == 1) {
        DTT_CHECK(p.K == 9 * p.cCi && p.M == p.cB * p.cFo * p.cTo, "dtt_gemm: conv geometry mismatch (K=%d Ci=%d M=%d)",
                  p.K, p.cCi, p.M);
        DTT_CHECK(p.cCi % (p.dtype ? 8 : 4) == 0, "dtt_gemm: conv Ci=%d must be a multiple of 16 bytes", p.cCi);
        DTT_CHECK(!p.ctrans || p.cstride == 2, "dtt_gemm: transposed conv needs stride 2");
    }
    if (p.b_mode == 1 && p.cKW) {
        DTT_CHECK(p.K == p.cB * p.cFo * p.cTo && p.batch == 1 && p.N % (p.cKW * p.cCi) == 0 && p.b_ns == 1 && p.cPadF >= 0 &&
                      p.cPadT >= 0 && p.cstride >= 1,
                  "dtt_gemm: general wgrad geometry mismatch");
        DTT_CHECK(!p.ctrans, "dtt_gemm: wgrad gather is always the strided (non-transposed) form");
    } else if (p.b_mode == 1) {
        DTT_CHECK(p.K == p.cB * p.cFo * p.cTo && p.batch == 9 && p.N == p.cCi && p.b_ns == 1,
                  "dtt_gemm: wgrad geometry mismatch");
        DTT_CHECK(!p.ctrans, "dtt_gemm: wgrad gather is always the strided (non-transposed) form");
    }
    if (p.b_mode == 2) DTT_CHECK(p.sh_heads > 0 && p.sh_L > 0 && p.b_ns == 1, "dtt_gemm: bad shift params");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = dtt_gemm256_try(p, st);
    if (rc != 1) return rc;
    rc = dtt_gemmx_try(p, st);
    if (rc != 1) return rc;
    rc = p.dtype == 1 ? dispatch_fast<bf16>(p, st) : dispatch_fast<float>(p, st);
    if (rc != 1) return rc;
    DTT_CHECK(!p.cKW, "dtt_gemm: the general conv geometry (cKW) needs 16-byte aligned, n-contiguous operands (fast kernels only)");
    return p.dtype == 1 ? dispatch<bf16>(p, st) : dispatch<float>(p, st);
}
