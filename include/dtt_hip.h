/*
 * dtt_hip.h -- C ABI of libdtt_hip.so: hand-written gfx950 (MI355X / CDNA4)
 * kernels for the DTTNet separation forward + backward path.
 *
 * The reference (ArtemShamro/speech-separation) has no native/FFI layer: the
 * hot path sits behind PyTorch nn.Module / loss / metric classes
 * (src/model/__init__.py:1-9, src/loss/__init__.py:1-5,
 * src/metrics/__init__.py:1-4).  Each entry point below therefore cites the
 * PyTorch op sequence (reference file:line) it replaces.  The Python host
 * (speech-separation_amd/dtt_amd) binds these with ctypes and mirrors the
 * reference classes; see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (torch's caching
 *     allocator); the library never allocates, frees or keeps a pointer;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises;
 *   - dtype: 0 = float32, 1 = bfloat16 (activations / packed weights);
 *     statistics, biases, losses and weight gradients are always float32;
 *   - activations are channels-last: [B, F, T, C];
 *   - return 0 on success, negative on error (dtt_last_error() has the text).
 */
#ifndef DTT_HIP_H
#define DTT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* dtt_last_error(void);
int dtt_version(void);

/* ---------------------------------------------------------------------------
 * MFMA GEMM with fused gather / epilogue.  C[z][m][n] (+)= act(alpha * sum_k
 * A[z][m][k] * B[z][k][n] + bias) + R[z][m][n].
 * Replaces every nn.Linear / 1x1 conv / 3x3 conv / ConvTranspose2d and their
 * backward GEMMs on the path: block_tfc_tdf.py:12-32 (conv3x3, TDF linears),
 * block_sampling.py:8-28 (down / up), latent.py:87-96 (LSTM input projection,
 * output Linear), cross_attention_positional.py:21-49, model_dttnet.py:79-84
 * (iSTFT as inverse-DFT GEMM), base_dataset.py:164 (STFT as DFT GEMM).
 * ------------------------------------------------------------------------- */
typedef struct DttGemm {
    const void* A; const void* B; void* C; const float* bias; const void* R;
    int64_t a_ms, a_ks, a_bs;      /* element strides of A(m,k), batch stride   */
    int64_t b_ks, b_ns, b_bs;      /* element strides of B(k,n)                 */
    int64_t c_ms, c_ns, c_bs;      /* element strides of C(m,n)                 */
    int64_t r_ms, r_ns, r_bs;      /* element strides of R(m,n)                 */
    int32_t M, N, K, batch, splitk;
    int32_t dtype;                 /* A, B, R element type                      */
    int32_t out_f32;               /* 1: C is float32 regardless of dtype       */
    int32_t atomic;                /* 1: atomicAdd into float32 C (split-K)     */
    int32_t bias_mode;             /* 0 none, 1 bias[n], 2 bias[m]              */
    int32_t act;                   /* 0 none, 1 exact GELU, 2 sigmoid           */
    float alpha;
    /* gather modes: 0 plain strided; A: 1 = implicit-GEMM conv rows (m = output
     * pixel, k = tap*Ci + ci); B: 1 = conv wgrad gather (k = output pixel, tap =
     * batch index), 2 = row shift (LSTM h_{t-1}: k -> k + sh_shift, zero where
     * (k / sh_heads) % sh_L == sh_edge)                                         */
    int32_t a_mode, b_mode;
    int32_t cB, cFi, cTi, cCi, cFo, cTo, cstride, ctrans;   /* conv geometry     */
    int32_t sh_shift, sh_heads, sh_L, sh_edge;
} DttGemm;
int dtt_gemm(const DttGemm* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif
