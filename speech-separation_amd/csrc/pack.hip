// One launch for every derived weight tensor of a training step.  A step needs ~400 small tensors derived from
// its parameters (bf16 casts, conv weights in tap-major order, stacked / transposed LSTM weights, padded TDF
// weights, ...): each was a 5 us launch of its own, strung between the big kernels (2.3 ms of a 47 ms step).
// Every one of them is a strided copy with an optional fp32 -> bf16 cast, so the host records them once as
// descriptors (source view, destination view, up to 4 dims) into a table and this kernel walks the table: a
// chunk table maps each workgroup to (descriptor, chunk of it).  The host orders a descriptor's dims by destination
// stride; dtt_multi_pack_nchunks(desc) says how many chunks the kernel cuts it into (pack_geom below).
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

constexpr int PACK_CHUNK = 8192;
constexpr int TILE_ELEMS = 4096;                 // elements of a transposing tile
constexpr int TILE_LDS = 4608;                   // floats: TI rows of TJ + 1 (TI <= 64, TJ <= 512, TI * TJ <= 4096)

// How a descriptor is walked (a pure function of the descriptor: the host counts chunks with it, the kernel
// decodes its chunk with it).  Of the 108 M elements a cfg4 step packs, 26 % are contiguous casts, 29 % conv
// weights into tap-major order and 44 % plain 2-D transposes (LSTM / Linear / TDF weights): the element-wise
// walk reads the last two kinds 4 bytes at a time at a stride (one cache line per element, 0.58 ms per step).
//   VEC:   1-D contiguous fp32 source, 16-byte aligned: float4 loads, 8- / 16-byte stores.
//   TILE:  the destination's last dim is contiguous, the source's unit-stride dim (+1 or -1: the data-gradient packs
//          walk the taps backwards) is another one (I, optionally with the dim that makes a dense PAIR with it in
//          memory, such as [Ci][tap] of a conv weight): TI x TJ tiles through LDS, read along I, written along J.
//   WALK:  everything else, element by element in destination order (the first form of this kernel).
enum { WALK = 0, VEC = 1, TILE = 2 };
struct PackGeom {
    int mode;
    int ib, ia;          // TILE: the source-contiguous dim, and the dim outside it when the pair is contiguous (-1)
    int nI, TI, TJ;      // extent of I (pair: product), tile extents
    int tiles_i, tiles_j;
    int b0, b1;          // the remaining (batch) dims, -1 when absent
    long long chunks;
};

// (a runtime subscript into the descriptor's arrays would move the whole descriptor into scratch memory)
template <typename T> __host__ __device__ __forceinline__ T sel3(const T (&a)[4], int k) { return k == 0 ? a[0] : k == 1 ? a[1] : a[2]; }

__host__ __device__ inline PackGeom pack_geom(const DttPackDesc& d) {
    PackGeom g;
    g.mode = WALK; g.ib = g.ia = g.b0 = g.b1 = -1; g.nI = g.TI = g.TJ = g.tiles_i = g.tiles_j = 0;
    g.chunks = (d.numel + PACK_CHUNK - 1) / PACK_CHUNK;
    if (d.numel <= 0 || d.numel >= (1ll << 31)) return g;
    const bool one_d = d.shape[0] == 1 && d.shape[1] == 1 && d.shape[2] == 1;
    if (d.dstride[3] != 1 && !one_d) return g;
    if (d.sstride[3] == 1 || d.shape[3] == 1) {
        if (one_d && d.sstride[3] == 1 && d.dstride[3] == 1 && d.src_dtype == 0 &&
            (reinterpret_cast<uintptr_t>(d.src) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(d.dst) & (d.dst_dtype ? 7 : 15)) == 0)
            g.mode = VEC;
        return g;
    }
    auto unit = [](int64_t st) { return st == 1 || st == -1; };
    if (d.shape[2] > 1 && unit(d.sstride[2])) g.ib = 2;
    else if (d.shape[1] > 1 && unit(d.sstride[1])) g.ib = 1;
    else if (d.shape[0] > 1 && unit(d.sstride[0])) g.ib = 0;
    if (g.ib < 0) return g;
    g.nI = sel3(d.shape, g.ib);
#pragma unroll
    for (int k = 2; k >= 0; --k) {                 // a dim whose source stride is I's extent: the pair is dense in memory
        const int64_t st = d.sstride[k];
        if (g.ia < 0 && k != g.ib && d.shape[k] > 1 && (st == g.nI || st == -(int64_t)g.nI) &&
            (long long)d.shape[k] * g.nI < (1 << 20))
            g.ia = k;
    }
    if (g.ia >= 0) g.nI *= sel3(d.shape, g.ia);
    g.TI = g.nI < 64 ? g.nI : 64;
    int tj = 1;
    while (tj * 2 * g.TI <= TILE_ELEMS && tj < 512 && tj < d.shape[3]) tj *= 2;
    g.TJ = tj;
    g.tiles_i = (g.nI + g.TI - 1) / g.TI;
    g.tiles_j = (d.shape[3] + g.TJ - 1) / g.TJ;
    long long nb = 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k == g.ib || k == g.ia) continue;
        if (g.b0 < 0) g.b0 = k; else g.b1 = k;
        nb *= d.shape[k];
    }
    g.mode = TILE;
    g.chunks = nb * g.tiles_i * g.tiles_j;
    return g;
}

__device__ __forceinline__ float pack_load(const DttPackDesc& d, int64_t o, bool sb) {
    return sb ? bf2f(reinterpret_cast<const unsigned short*>(d.src)[o]) : reinterpret_cast<const float*>(d.src)[o];
}
__device__ __forceinline__ void pack_store(const DttPackDesc& d, int64_t o, bool db, float v) {
    if (db) reinterpret_cast<unsigned short*>(d.dst)[o] = f2bf(v);
    else reinterpret_cast<float*>(d.dst)[o] = v;
}

// WALK: a thread decodes the multi-index of its first element once (three 32-bit divisions) and then steps by 256
// elements with carries: the straightforward form (64-bit div / mod per element) made this kernel ALU-bound at
// over a millisecond for the 120 M elements of a step.
__device__ void pack_walk(const DttPackDesc& d, unsigned chunk) {
    const unsigned s1 = d.shape[1], s2 = d.shape[2], s3 = d.shape[3];
    const unsigned numel = (unsigned)d.numel;
    unsigned e = chunk * PACK_CHUNK + threadIdx.x;
    unsigned r = e;
    unsigned i3 = r % s3; r /= s3;
    unsigned i2 = r % s2; r /= s2;
    unsigned i1 = r % s1;
    unsigned i0 = r / s1;
    // the step of 256 elements as a multi-index
    unsigned q = 256u;
    const unsigned a3 = q % s3; q /= s3;
    const unsigned a2 = q % s2; q /= s2;
    const unsigned a1 = q % s1;
    const unsigned a0 = q / s1;
    const bool sb = d.src_dtype != 0, db = d.dst_dtype != 0;
    for (int k = 0; k < PACK_CHUNK / 256 && e < numel; ++k, e += 256) {
        const int64_t so = i0 * d.sstride[0] + i1 * d.sstride[1] + i2 * d.sstride[2] + i3 * d.sstride[3];
        const int64_t dof = i0 * d.dstride[0] + i1 * d.dstride[1] + i2 * d.dstride[2] + i3 * d.dstride[3];
        pack_store(d, dof, db, pack_load(d, so, sb));
        i3 += a3; if (i3 >= s3) { i3 -= s3; ++i2; }
        i2 += a2; if (i2 >= s2) { i2 -= s2; ++i1; }
        i1 += a1; if (i1 >= s1) { i1 -= s1; ++i0; }
        i0 += a0;
    }
}

__device__ void pack_vec(const DttPackDesc& d, unsigned chunk) {
    const unsigned numel = (unsigned)d.numel;
    const float* __restrict__ src = reinterpret_cast<const float*>(d.src);
    const bool db = d.dst_dtype != 0;
#pragma unroll
    for (int k = 0; k < PACK_CHUNK / 1024; ++k) {
        const unsigned e = chunk * PACK_CHUNK + (k * 256 + threadIdx.x) * 4;
        if (e + 4 <= numel) {
            const float4 v = *reinterpret_cast<const float4*>(src + e);
            if (db) {
                uint2 o;
                o.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
                o.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(d.dst) + e) = o;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.dst) + e) = v;
            }
        } else {
            for (unsigned t = e; t < numel; ++t) pack_store(d, t, db, src[t]);
        }
    }
}

__device__ void pack_tile(const DttPackDesc& d, const PackGeom& g, unsigned chunk, float* lds) {
    unsigned t = chunk;
    const unsigned it = t % g.tiles_i; t /= g.tiles_i;
    const unsigned jt = t % g.tiles_j; t /= g.tiles_j;
    int64_t sbase = 0, dbase = 0;
    if (g.b1 >= 0) {
        const unsigned n1 = sel3(d.shape, g.b1);
        const unsigned x1 = t % n1; t /= n1;
        sbase += x1 * sel3(d.sstride, g.b1); dbase += x1 * sel3(d.dstride, g.b1);
    }
    if (g.b0 >= 0) { sbase += t * sel3(d.sstride, g.b0); dbase += t * sel3(d.dstride, g.b0); }
    const unsigned i0 = it * g.TI, j0 = jt * g.TJ;
    const unsigned ni = min((unsigned)g.TI, (unsigned)g.nI - i0), nj = min((unsigned)g.TJ, (unsigned)d.shape[3] - j0);
    const unsigned TI = g.TI, TJ = g.TJ, ld = TJ + 1;
    const bool sb = d.src_dtype != 0, db = d.dst_dtype != 0;
    const int64_t sJ = d.sstride[3];
    // read along I (the source's contiguous dim, or dense pair of dims); e / TI by a multiplier exact for e < 4096,
    // TI <= 64, ii / nb by one exact for ii, nb < 2^20
    const unsigned inv = (262144u + TI - 1) / TI;
    const unsigned total = TI * TJ;
    const unsigned nb = g.ia >= 0 ? (unsigned)sel3(d.shape, g.ib) : 1u;
    const unsigned long long minv = ((1ull << 40) + nb - 1) / nb;
    const int64_t sIa = g.ia >= 0 ? sel3(d.sstride, g.ia) : 0, sIb = sel3(d.sstride, g.ib);
    for (unsigned e = threadIdx.x; e < total; e += 256) {
        const unsigned j = (e * inv) >> 18, i = e - j * TI;
        if (i < ni && j < nj) {
            const unsigned ii = i0 + i;
            int64_t o;
            if (g.ia >= 0) { const unsigned a = (unsigned)((ii * minv) >> 40); o = a * sIa + (int64_t)(ii - a * nb) * sIb; }
            else o = (int64_t)ii * sIb;
            lds[i * ld + j] = pack_load(d, sbase + (int64_t)(j0 + j) * sJ + o, sb);
        }
    }
    __syncthreads();
    // written along J; I back to (ia, ib) for the destination offset when it is a pair
    const int64_t dIa = g.ia >= 0 ? sel3(d.dstride, g.ia) : 0, dIb = sel3(d.dstride, g.ib);
    const unsigned shj = __ffs(TJ) - 1;
    for (unsigned e = threadIdx.x; e < total; e += 256) {
        const unsigned i = e >> shj, j = e & (TJ - 1);
        if (i < ni && j < nj) {
            const unsigned ii = i0 + i;
            int64_t o;
            if (g.ia >= 0) { const unsigned a = (unsigned)((ii * minv) >> 40); o = a * dIa + (int64_t)(ii - a * nb) * dIb; }
            else o = (int64_t)ii * dIb;
            pack_store(d, dbase + o + (j0 + j), db, lds[i * ld + j]);
        }
    }
}

__global__ __launch_bounds__(256) void multi_pack_kernel(const DttPackDesc* __restrict__ descs, const int2* __restrict__ chunks) {
    __shared__ float lds[TILE_LDS];
    const int2 chv = chunks[blockIdx.x];
    // (uniform over the workgroup: through scalar registers, so the descriptor is fetched by scalar loads)
    const int2 ch = make_int2(__builtin_amdgcn_readfirstlane(chv.x), __builtin_amdgcn_readfirstlane(chv.y));
    const DttPackDesc d = descs[ch.x];
    const PackGeom g = pack_geom(d);
    if (g.mode == TILE) pack_tile(d, g, (unsigned)ch.y, lds);
    else if (g.mode == VEC) pack_vec(d, (unsigned)ch.y);
    else pack_walk(d, (unsigned)ch.y);
}

}  // namespace

extern "C" int dtt_multi_pack_chunk(void) { return PACK_CHUNK; }

extern "C" int64_t dtt_multi_pack_nchunks(const DttPackDesc* desc) {
    if (!desc) return -1;
    return pack_geom(*desc).chunks;
}

extern "C" int dtt_multi_pack(const DttPackDesc* descs, const int32_t* chunks, int nchunks, void* stream) {
    DTT_CHECK(descs && chunks && nchunks >= 0, "dtt_multi_pack: bad arguments");
    if (nchunks == 0) return 0;
    hipLaunchKernelGGL(multi_pack_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), descs,
                       reinterpret_cast<const int2*>(chunks));
    DTT_LAUNCH_CHECK();
    return 0;
}
