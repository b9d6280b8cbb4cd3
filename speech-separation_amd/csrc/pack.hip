// One launch for every derived weight tensor of a training step.  A step needs ~400 small tensors derived from
// its parameters (bf16 casts, conv weights in tap-major order, stacked / transposed LSTM weights, padded TDF
// weights, ...): each was a 5 us launch of its own, strung between the big kernels (2.3 ms of a 47 ms step).
// Every one of them is a strided copy with an optional fp32 -> bf16 cast, so the host records them once as
// descriptors (source view, destination view, up to 4 dims) into a table and this kernel walks the table: a
// chunk table maps each workgroup to (descriptor, first element).  Elements are walked in the row-major order of
// the descriptor's shape, which the host orders by destination stride (coalesced writes).
#include "common.h"
#include "../../include/dtt_hip.h"

namespace {

constexpr int PACK_CHUNK = 8192;

// A thread decodes the multi-index of its first element once (three 32-bit divisions) and then steps by 256
// elements with carries: the straightforward form (64-bit div / mod per element) made this kernel ALU-bound at
// over a millisecond for the 120 M elements of a step.
__global__ __launch_bounds__(256) void multi_pack_kernel(const DttPackDesc* __restrict__ descs, const int2* __restrict__ chunks) {
    const int2 ch = chunks[blockIdx.x];
    const DttPackDesc d = descs[ch.x];
    const unsigned s1 = d.shape[1], s2 = d.shape[2], s3 = d.shape[3];
    const unsigned numel = (unsigned)d.numel;
    unsigned e = (unsigned)ch.y * PACK_CHUNK + threadIdx.x;
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
        const float v = sb ? bf2f(reinterpret_cast<const unsigned short*>(d.src)[so]) : reinterpret_cast<const float*>(d.src)[so];
        if (db) reinterpret_cast<unsigned short*>(d.dst)[dof] = f2bf(v);
        else reinterpret_cast<float*>(d.dst)[dof] = v;
        i3 += a3; if (i3 >= s3) { i3 -= s3; ++i2; }
        i2 += a2; if (i2 >= s2) { i2 -= s2; ++i1; }
        i1 += a1; if (i1 >= s1) { i1 -= s1; ++i0; }
        i0 += a0;
    }
}

}  // namespace

extern "C" int dtt_multi_pack_chunk(void) { return PACK_CHUNK; }

extern "C" int dtt_multi_pack(const DttPackDesc* descs, const int32_t* chunks, int nchunks, void* stream) {
    DTT_CHECK(descs && chunks && nchunks >= 0, "dtt_multi_pack: bad arguments");
    if (nchunks == 0) return 0;
    hipLaunchKernelGGL(multi_pack_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), descs,
                       reinterpret_cast<const int2*>(chunks));
    DTT_LAUNCH_CHECK();
    return 0;
}
