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
    /* transposed conv (ctrans = 1) by output parity: 0 = all pixels, all nine taps (7 of 9 products
     * are structural zeros on average); 1 + 2*pf + pt = only the output pixels (fo, to) with
     * fo%2 == pf, to%2 == pt: M = B * ceil((cFo-pf)/2) * ceil((cTo-pt)/2), K = (pf?2:1)*(pt?2:1)*cCi
     * with the taps kh in (pf ? {0,2} : {1}), kw likewise; C rows are the pixels' rows of the full
     * [B][cFo][cTo] output (c_ms = row stride of that tensor); a residual R is indexed by the COMPACT
     * GEMM row m (the phase's own [B][Fp][Tp] pixel order), r_ms its row stride                 */
    int32_t cphase;
    int32_t reserved_;
    /* optional scratch for split-K (atomic = 1) products: with ws_bytes >= dtt_gemm_ws_bytes(p) the k-splits store
     * their partial tiles there (plain 16-byte stores) and a second launch adds their sum to C, instead of
     * splitk * M * N fp32 atomics (48 of the 123 us of a 2048 x 512 x 33 280 LSTM weight gradient).  NULL / too
     * small: atomics.  Contents are undefined afterwards; calls sharing one workspace must be stream-ordered.   */
    void* ws;
    int64_t ws_bytes;
    /* general conv geometry of the gather modes (0 = the 3x3 / pad 1 taps above): kernel width cKW, zero padding
     * cPadF rows above / cPadT columns left; source pixel of tap (kh, kw) = (fo * cstride + kh - cPadF,
     * to * cstride + kw - cPadT), non-transposed only.  a_mode 1: k = (kh * cKW + kw) * cCi + ci, K a multiple of
     * cKW * cCi.  b_mode 1: batch = 1 and n = kh * (cKW * cCi) + (kw * cCi + ci): the cKW pixels of a kernel row are
     * read as ONE contiguous run of cKW * cCi elements starting at the tap's first pixel, so the caller pads the
     * input's columns (cPadT = 0, cTi = padded width) -- only rows are predicated.  Used for the lip front end
     * (resnet_video_encoder.py:33-37: 7 x 7 / stride 2 over 5 stacked frames, dtt_stack_frames). */
    int32_t cKW, cPadF, cPadT, reserved2_;
} DttGemm;
int dtt_gemm(const DttGemm* p, void* stream);
/* bytes of workspace with which dtt_gemm(p) would take the store-and-sum split-K path; 0: it has no use for one */
int64_t dtt_gemm_ws_bytes(const DttGemm* p);
/* Diagnostic: dtt_gemm calls of this process that ran on the LDS-DMA tile kernels of gemmx.hip (bf16, k-contiguous
 * or stride-1 3x3 conv-row A operand, M >= 2048, N >= 64: block_tfc_tdf.py:12-17 convs and data gradients at
 * C >= 128, the ResNet-18 trunk's stride-1 convs, large Linears).  DTT_GEMMX=0 disables them. */
long long dtt_gemmx_launches(void);

/* Dedicated 3x3 stride-1 'same' conv on channels-last (block_tfc_tdf.py:12-32): LDS input
 * patches shared by the nine taps; wp = [Co][9][Ci]; optional fused residual and fused
 * InstanceNorm statistics (partial scratch of B*(4*ceil(F/16)*ceil(T/16)+4)*Co*3 floats (layout private to the kernels) -> stats
 * [B][Co][2] = {mean, rstd}).  The same kernel computes the data gradient (flipped weights).
 * dtt_conv3x3_wgrad: dw[Co][9][Ci] fp32 += sum_px dy[px][co] x[px+tap][ci] (bf16 operands,
 * transposing LDS reads; caller zeroes dw). */
int dtt_conv3x3_s1(const void* x, const void* wp, const float* bias, void* y, const void* res, float* partial,
                   float* stats, int B, int F, int T, int Ci, int Co, float eps, int dtype, void* stream);
int dtt_conv3x3_wgrad(const void* dy, const void* x, float* dw, int B, int F, int T, int Ci, int Co, int dtype,
                      void* stream);
/* The same weight gradient (reference: autograd of the Conv2d(c, c, 3, 1, 1) of block_tfc_tdf.py:12-17) with the
 * workgroups' partial sums stored to a caller workspace of dtt_conv3x3_wgrad_ws_bytes(...) bytes and added up by a
 * second launch instead of fp32 atomics; dw is OVERWRITTEN (no zeroing).  _ws_bytes returns 0 for channel counts
 * only the atomic form covers (multiples of 32 that are not multiples of 64, other than 32 -> 32). */
int64_t dtt_conv3x3_wgrad_ws_bytes(int B, int F, int T, int Ci, int Co);
int dtt_conv3x3_wgrad_ws(const void* dy, const void* x, float* dw, void* ws, int64_t ws_bytes, int B, int F, int T,
                         int Ci, int Co, int dtype, void* stream);
/* ... with dw in nn.Conv2d.weight's own layout [Co][Ci][3][3] (block_tfc_tdf.py:12-17: the parameter whose .grad it
 * becomes) instead of [Co][9][Ci]: no permute copy on the caller's side. */
int dtt_conv3x3_wgrad_ws_param(const void* dy, const void* x, float* dw, void* ws, int64_t ws_bytes, int B, int F, int T,
                               int Ci, int Co, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * InstanceNorm2d(affine=False, eps) + exact GELU on channels-last tensors
 * (block_tfc_tdf.py:15-16, block_sampling.py:10-11,26-27), incl. the decoder
 * crop (decoder.py:46-52): statistics over the full (Fy,Ty) plane, output /
 * incoming gradient on the cropped (Fz,Tz) plane.
 *   partial: [B][ceil(P/chunk)][C][2] scratch, stats/sums: [B][C][2] fp32.
 * ------------------------------------------------------------------------- */
int dtt_in_stats(const void* y, float* partial, float* stats, int B, int P, int C, int chunk, float eps,
                 int dtype, void* stream);
int dtt_in_gelu_fwd(const void* y, const float* stats, void* z, int B, int Fy, int Ty, int Fz, int Tz, int C,
                    int dtype, void* stream);
int dtt_in_gelu_bwd(const void* y, const void* dz, const float* stats, float* partial, float* sums, void* dy,
                    int B, int Fy, int Ty, int Fz, int Tz, int C, int chunk, int dtype, void* stream);

/* GroupNorm over (gs channels x L steps) per sequence, fused with the (F,T)
 * re-ordering of IDPModule (latent.py:47-54,82-85,100).  Slab o reads rows
 * x + (o/NI)*OS + (o%NI)*IS + l*RS and writes [o][l][C] contiguously; xperm
 * (optional) receives the un-normalised copy in the same order (residual).
 * gn_bwd adds dres (same order) and scatters dx back; pgb: [nslab][C][2]. */
int dtt_gn_fwd(const void* x, const float* gamma, const float* beta, void* xn, void* xperm, float* stats,
               int nslab, int NI, int L, int C, int Cp, int gs, int64_t OS, int64_t IS, int64_t RS, float eps,
               int dtype, void* stream);
int dtt_gn_bwd(const void* x, const void* dxn, const void* dres, const float* gamma, const float* stats,
               void* dx, float* pgb, int nslab, int NI, int L, int C, int Cp, int gs, int64_t OS, int64_t IS,
               int64_t RS, int dtype, void* stream);

/* Deterministic column sums (bias gradients): out[nchunks][W]. */
int dtt_colsum(const void* in, float* out, int64_t R, int W, int64_t ld, int nchunks, int dtype, void* stream);

/* Weighted column sums out[chunk][m][w] = sum_r a[r][m] * in[r][w] (m < MW <= 8): weight gradient of a
 * 1x1 conv with a handful of outputs -- the mask head decoder.out_conv (decoder.py:29), dW = dmask^T x.
 * Finish with dtt_colsum over the chunks (width MW*W). */
int dtt_wcolsum(const void* a, const void* in, float* out, int64_t R, int W, int MW, int nchunks, int dtype,
                void* stream);

/* Fused BiLSTM forward for bf16, H = 256, C' = 128 (the bench configuration): the input projection
 * x*W_ih^T + b_ih + b_hh of nn.LSTM (reference src/model/DTTNet/modules/latent.py:87-92) is computed
 * inside the recurrent kernel, so no [rows][2][4H] pre-activation tensor exists.
 *   xn [rows][Cp], wihp [2][4H/32][Cp/16][64][8], whhp [2][4H/32][H/16][64][8] (MFMA fragment order),
 *   bias [2][4H] fp32, y [rows][2H].  Saved for dtt_lstm_bwd_blk (NULL: inference):
 *   gates_blk [2][ceil(N/32)][L][4][H][2][16], cst_blk [2][ceil(N/32)][L][H][2][16] fp32.
 * Returns 1 without setting an error when the shape is not covered (use dtt_gemm + dtt_lstm_fwd). */
int dtt_lstm_fwd_fused(const void* xn, const void* wihp, const void* whhp, const float* bias, void* y,
                       void* gates_blk, float* cst_blk, int N, int L, int heads, int H, int Cp, int dtype,
                       void* stream);

/* Backward through time over the blocked activations of dtt_lstm_fwd_fused (bf16, H = 256): the
 * autograd of nn.LSTM in latent.py:87-92.  dy [rows][2H], wpt as for dtt_lstm_bwd, dG [rows][2][4H]
 * row-major (consumed by dtt_gemm for dW_ih, dW_hh, dx); dbias [2][4H] fp32, zeroed by the caller (or NULL),
 * accumulates the bias gradient (column sums of dG) atomically. */
int dtt_lstm_bwd_blk(const void* dy, const void* gates_blk, const float* cst_blk, const void* wpt, void* dG,
                     float* dbias, int N, int L, int heads, int H, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Persistent bidirectional LSTM recurrence, forward and backward through time
 * (nn.LSTM(C', 2C', bidirectional), latent.py:87-92; gate order i,f,g,o).
 * Rows: (outer*L + step)*heads + head.  xg [rows][2][4H] = x W_ih^T + b_ih + b_hh
 * (from dtt_gemm); wp / wpt = W_hh in MFMA fragment order (see dtt_amd/ops.py);
 * y [rows][2H]; gates [rows][2][4H]; cst [rows][2][H] fp32; dG [rows][2][4H].
 * ------------------------------------------------------------------------- */
int dtt_lstm_fwd(const void* xg, const void* wp, void* y, void* gates, float* cst, int N, int L, int heads, int H,
                 int dtype, void* stream);
int dtt_lstm_bwd(const void* dy, const void* gates, const float* cst, const void* wpt, void* dG, int N, int L,
                 int heads, int H, int dtype, void* stream);

/* Weight-stationary form of the same recurrence (nn.LSTM forward / backward, latent.py:87-92) for the wide
 * latents of the audio-visual models: the 4H gate columns of a direction are split over G workgroups that keep
 * their rows of W_hh resident in LDS for all L steps; h_t (forward) / dG_t (backward) is exchanged between the
 * CUs of a sequence group once per step through the output buffer (write-through stores + one counter per
 * group).  The grid (2 * S * G workgroups) is sized to the CU count: all workgroups must be co-resident.
 *   dtt_lstm_cluster_geometry: U hidden units per workgroup, G = H / U, S sequence groups; returns 1 when the
 *     shape is not covered (use dtt_lstm_fwd / dtt_lstm_bwd).
 *   xg [rows][2][4H] as for dtt_lstm_fwd; whh [2][4H][H] (PyTorch layout, dtype); whhT [2][H][4H] = its transpose;
 *   y [rows][2H]; gates_blk [2][NT][L][4][H][2][16] (dtype), cst_blk [2][NT][L][H][2][16] fp32, NT = ceil(N/32);
 *   dG [rows][2][4H]; dcbuf [2][NT][H][32] fp32 scratch; dbias [2][4H] fp32 zeroed by the caller (or NULL);
 *   sync: dtt_lstm_cluster_sync_words(S) uint32 of scratch (zeroed by the call). */
int dtt_lstm_cluster_geometry(int N, int H, int dtype, int* U, int* G, int* S);
int dtt_lstm_cluster_sync_words(int S);
int dtt_lstm_cluster_fwd(const void* xg, const void* whh, void* y, void* gates_blk, float* cst_blk, unsigned* sync,
                         int N, int L, int heads, int H, int dtype, void* stream);
int dtt_lstm_cluster_bwd(const void* dy, const void* gates_blk, const float* cst_blk, const void* whhT, void* dG,
                         float* dcbuf, float* dbias, unsigned* sync, int N, int L, int heads, int H, int dtype,
                         void* stream);

/* Register-stationary form of the same recurrence (nn.LSTM forward / backward, latent.py:87-92) for bf16 at
 * H = 512 and H = 256: W_hh lives in the VGPR/AGPR file of a cluster of 8 (H = 512) or 2 (H = 256) CUs for the
 * whole launch (one wave per SIMD, 256 registers of MFMA operand fragments per lane), the h / dG tile of the
 * previous step is fetched once per CU into LDS, and a cluster walks its 32-sequence tiles round-robin with one
 * hand-off counter per tile.  Arguments as for dtt_lstm_cluster_fwd / _bwd, except the layout of the saved
 * activations (private to the pair, same sizes): gates_blk [2][NT][L][H/4][32][4 gates][4 units] (bf16),
 * cst_blk [2][NT][L][H/4][32][4] fp32; dcbuf [2][NT][H/4][32][4] fp32 scratch.
 *   dtt_lstm_rs_supported: 0 when (N, H, dtype) is covered on this GPU, 1 otherwise.
 *   sync: dtt_lstm_rs_sync_words(N, H) uint32 of scratch (zeroed by the call). */
int dtt_lstm_rs_supported(int N, int H, int dtype);
int dtt_lstm_rs_sync_words(int N, int H);
int dtt_lstm_rs_fwd(const void* xg, const void* whh, void* y, void* gates_blk, float* cst_blk, unsigned* sync,
                    int N, int L, int heads, int H, int dtype, void* stream);
int dtt_lstm_rs_bwd(const void* dy, const void* gates_blk, const float* cst_blk, const void* whhT, void* dG,
                    float* dcbuf, float* dbias, unsigned* sync, int N, int L, int heads, int H, int dtype,
                    void* stream);
/* As dtt_lstm_rs_supported, with the step count: 1 also when N * L exceeds the 32-bit byte offsets of xg / dG or the
 * 31-bit offsets of the saved activations (the launch would be refused): the caller falls back to
 * dtt_lstm_cluster_* / dtt_lstm_*. */
int dtt_lstm_rs_covers(int N, int L, int H, int dtype);

/* Polls a hand-off wait of the register- / LDS-stationary LSTM kernels makes before it gives up and raises the status
 * words (process-wide, ~1 us per poll; 0 restores the default of 2^23 ~ 8 s).  The default rides out rank skew in a
 * data-parallel job, where a collective that waits for a peer can hold CUs the launch needs; tests set a small value.
 * No counterpart in the reference (nn.LSTM has no cross-workgroup hand-off). */
int dtt_lstm_handoff_spin_limit(unsigned polls);
/* Hand-off health (no counterpart in the reference: nn.LSTM, latent.py:87-92, cannot fail this way).  The
 * dtt_lstm_rs_* and dtt_lstm_cluster_* launches need all their workgroups co-resident (one per CU); a workgroup
 * that waits ~0.5 s for a peer gives up, raises the launch's status word (sync[0] / the tail word) AND a sticky
 * per-GPU word, and runs to completion on stale tiles: the results of that launch are garbage.
 *   dtt_lstm_{rs,cluster}_status_fold: one tiny launch on `stream`: *flag |= 1 (rs) / 2 (cluster) when a launch of
 *     that family has timed out on the current GPU since the last clear; clear != 0 re-arms the sticky word.
 *     flag: one uint32 in device memory (may be NULL with clear != 0).
 *   dtt_lstm_{rs,cluster}_diag_drop_workgroups(n): diagnostic -- the next launch of that family starts n workgroups
 *     short (what a CU-masked partition does to it), so tests can see the product notice. */
int dtt_lstm_rs_status_fold(unsigned* flag, int clear, void* stream);
int dtt_lstm_cluster_status_fold(unsigned* flag, int clear, void* stream);
void dtt_lstm_rs_diag_drop_workgroups(int n);
void dtt_lstm_cluster_diag_drop_workgroups(int n);

/* Every derived weight tensor of a step in one launch: the casts / permuted / stacked / transposed copies of
 * parameters that the reference's modules get from autocast and from PyTorch's own weight layouts
 * (block_tfc_tdf.py:12-32 conv weights, latent.py:87-96 LSTM / Linear weights, ...) are strided copies with an
 * optional fp32 -> bf16 cast.  descs: device table; chunks: device table of int32 pairs (descriptor index, chunk
 * index), one workgroup each, chunk indices 0 .. dtt_multi_pack_nchunks(desc) - 1 per descriptor (contiguous casts
 * go by vectors, transposing copies by tiles through LDS, the rest dtt_multi_pack_chunk() elements at a time,
 * row-major over `shape`: dims ordered by destination stride, numel < 2^31). */
typedef struct DttPackDesc {
    const void* src; void* dst;
    int64_t sstride[4], dstride[4];   /* element strides */
    int64_t numel;
    int32_t shape[4];
    int32_t src_dtype, dst_dtype;     /* 0 = float32, 1 = bfloat16 */
} DttPackDesc;
int dtt_multi_pack_chunk(void);
int64_t dtt_multi_pack_nchunks(const DttPackDesc* desc);
int dtt_multi_pack(const DttPackDesc* descs, const int32_t* chunks, int nchunks, void* stream);

/* ---------------------------------------------------------------------------
 * STFT / iSTFT glue (base_dataset.py:154-167; model_dttnet.py:77-85), the 1x1
 * input conv (encoder.py:26-27) and the fused out conv + mask + torch.polar
 * (decoder.py:29; model_dttnet.py:47-78) with its backward (incl. polar's
 * sgn() autograd rule).  Z (the complex spectrum handed to the inverse DFT,
 * re at column f, im at imoff + f) is float32 in every mode: torch.polar /
 * torch.istft stay complex64 under bf16 autocast.
 * ------------------------------------------------------------------------- */
int dtt_reflect_pad(const float* x, float* xp, int B, int L, int pad, void* stream);
int dtt_stft_post(const float* reim, float* mag, float* phase, int B, int F, int T, int ld, int imoff,
                  void* stream);
int dtt_init_conv_fwd(const float* spec, const float* phase, const float* w, const float* bias, void* out,
                      int64_t P, int C, int dtype, void* stream);
int dtt_init_conv_bwd(const float* spec, const float* phase, const void* dout, float* partial, int64_t P, int C,
                      int nchunks, int dtype, void* stream);
int dtt_mask_fwd(const void* x, const float* w, const float* bias, const float* mspec, const float* mphase,
                 float* pspec, float* pphase, void* Z, int B, int F, int T, int C, int S, int ldz, int imoff,
                 int dtype, void* stream);
int dtt_mask_bwd(const float* w, const float* mspec, const float* mphase, const float* pspec, const float* pphase,
                 const float* dspec, const float* dphase, const float* dZ, void* dx, void* dmask, int B, int F,
                 int T, int C, int S, int ldz, int imoff, int dtype, void* stream);
int dtt_ola_fwd(const float* frames, float* out, int SB, int T, int n_fft, int hop, int length, void* stream);
int dtt_ola_bwd(const float* dout, void* gpad, int SB, int T, int n_fft, int hop, int length, int dtype,
                void* stream);

/* ---------------------------------------------------------------------------
 * PIT L1 loss (dttnet_loss.py:28-34, pit_loss_wrapper.py:41-71,
 * dttnet_loss_wrapper.py:41-62) and SI-SNR (si_snr.py:26-34).
 * pred: [S][B][n]; src: host array of S device pointers to [B][n] planes.
 * ------------------------------------------------------------------------- */
int dtt_l1_pairs(const float* pred, const float* const* src, float* partial, int S, int B, int64_t n,
                 int nchunks, void* stream);
int dtt_pit_select(const float* p_spec, const float* p_phase, const float* p_audio, int nc_sp, int nc_au,
                   int64_t FT, int64_t L, int S, int B, int fixed_order, float* item_loss, int64_t* permuts,
                   float* loss, void* stream);
int dtt_l1_bwd(const float* pred, const float* const* src, const int64_t* permuts, const float* gloss, float scale,
               float* dpred, int S, int B, int64_t n, void* stream);
int dtt_sisnr(const float* est, const float* const* src, float* partial, float* alpha, float* out, int E, int S,
              int B, int64_t L, int nchunks, float eps, void* stream);

/* dtt_sisnr for callers that cannot allocate.  NO counterpart in the reference: it exists for callers of this
 * library that may not call hipMalloc (code under HIP-graph capture, tight serving loops, harnesses that forbid
 * allocation).  Same four launches and the same result, bit for bit, as dtt_sisnr with
 * nchunks = max(1, min(32, L / 2048)); the chunk partials and projection coefficients live in a static device
 * array of dtt_sisnr_noalloc_max_floats() floats (1 MiB, one copy per GPU for the life of the process) instead
 * of caller-provided scratch.  Sizes with E*S*B*(2*nchunks+1) floats beyond that are refused with an error.
 * The call allocates and frees nothing.  Calls that overlap on one GPU (different streams) race on the array. */
int dtt_sisnr_noalloc(const float* est, const float* const* src, float* out, int E, int S, int B, int64_t L,
                      float eps, void* stream);
int dtt_sisnr_noalloc_max_floats(void);

/* ---------------------------------------------------------------------------
 * Elementwise: op 0 add / 1 mul (decoder.py:43 skip multiply); FiLM apply
 * (block_fusion.py:30-37; gbc [B][T][2C] = gamma_c|beta_c, gbf [B][T][2F]);
 * frequency-mean tokens and broadcast concat of the cross-attention fusion
 * (cross_attention_positional.py:55,77-78).
 * ------------------------------------------------------------------------- */
int dtt_binary(int op, const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
/* Lip front end (resnet_video_encoder.py:33-37: Conv3d(1, 64, (5,7,7), stride (1,2,2), padding (2,3,3)) on a
 * one-channel clip): the kt neighbouring frames of every frame stacked as channels, zero frames beyond the clip's ends,
 * zero channels up to 8, channels-last, pad_left / pad_right zero columns around every row --
 * out[(n T + t)][H][pad_left + W + pad_right][8] (dtype) from x[n][T][H][W] (fp32) in one pass; the Conv3d is then a
 * per-frame Conv2d over this tensor (dtt_gemm with the general conv geometry cKW / cPadF / cPadT). */
int dtt_stack_frames(const float* x, void* out, int N, int T, int H, int W, int kt, int pad_left, int pad_right, int dtype,
                     void* stream);
/* Packed weight of the data-gradient conv of a stride-1 conv3x3 (autograd of F.conv2d, block_tfc_tdf.py:12-17):
 * w fp32 [Co][Ci][3][3] -> out [Ci][9][Co] with the taps rotated by 180 degrees, cast to dtype. */
int dtt_pack_conv3_dgrad(const float* w, void* out, int Co, int Ci, int dtype, void* stream);
int dtt_film_fwd(const void* x, const float* gbc, const float* gbf, void* y, int B, int F, int T, int C, int dtype,
                 void* stream);
int dtt_film_bwd(const void* x, const void* dy, const float* gbc, const float* gbf, void* dx, float* dgbc,
                 float* dgbf, int B, int F, int T, int C, int dtype, void* stream);
int dtt_fmean(const void* x, float* tok, int B, int F, int T, int C, int dtype, void* stream);
int dtt_bcat_fwd(const void* x, const float* enh, void* out, int B, int F, int T, int C, int dtype, void* stream);
int dtt_bcat_bwd(const void* dout, const float* dtok, void* dx, float* denh, int B, int F, int T, int C, int dtype,
                 void* stream);

/* Cross-attention core of CrossAttentionPositional (cross_attention_positional.py:66-69: nn.MultiheadAttention
 * behind its input projections): out = softmax(scale * q k^T) v per head, fp32, exact-fp32 MFMA tiles, one
 * workgroup per (item, head).  q [B][Tq][C], k, v [B][Tk][C], prob [B][heads][Tq][Tk] (kept for backward);
 * Tq <= 32, Tk <= 160, head width a multiple of 32. */
int dtt_xattn_fwd(const float* q, const float* k, const float* v, float* out, float* prob, int B, int Tq, int Tk, int C,
                  int heads, float scale, void* stream);
int dtt_xattn_bwd(const float* q, const float* k, const float* v, const float* prob, const float* dout, float* dq,
                  float* dk, float* dv, int B, int Tq, int Tk, int C, int heads, float scale, void* stream);

/* BatchNorm (+ residual) + activation on channels-last [P pixels][C] tensors, for the lip encoders on the
 * audio-visual path (VideoEncoders/resnet.py:36-81: bn -> swish; bn -> + residual -> swish; downsample bn;
 * resnet_video_encoder.py:40-44).  z = act(gamma * (x - mean) * rstd + beta (+ res)); act: 0 none, 1 swish, 2 relu.
 * stats [C][2] = {mean, rstd}: batch statistics from dtt_bn_stats in training, dtt_bn_frozen_stats(running
 * buffers) in eval.  Backward recomputes the activation input from x (and res): dx, dres (= gradient at the
 * activation input; may be NULL), dgb [C][2] = {dgamma, dbeta}; partial: dtt_bn_chunks(P) * C * 2 floats, sums
 * [C][2] floats of scratch.  dtt_bn_running: nn.BatchNorm's running-statistics update (unbiased variance). */
int dtt_bn_chunks(int P);
int dtt_bn_stats(const void* x, float* partial, float* stats, int P, int C, float eps, int dtype, void* stream);
/* dtt_bn_stats with the running-buffer update in the same launches; a tensor held as halves of h channels padded to hp
 * columns (the ShuffleNetV2 layout below) updates logical channel (c / hp) * h + c % hp; dense: h = hp = C. */
int dtt_bn_stats_running(const void* x, float* partial, float* stats, int P, int C, float eps, int dtype,
                         float* running_mean, float* running_var, float momentum, int h, int hp, void* stream);
int dtt_bn_act_fwd(const void* x, const void* res, const float* stats, const float* gamma, const float* beta, void* z,
                   int P, int C, int act, int dtype, void* stream);
int dtt_bn_act_bwd(const void* x, const void* res, const void* dz, const float* stats, const float* gamma,
                   const float* beta, float* partial, float* sums, float* dgb, void* dx, void* dres, int P, int C,
                   int act, int training, int dtype, void* stream);
int dtt_bn_running(const float* stats, float* running_mean, float* running_var, int C, int64_t n, float momentum,
                   float eps, void* stream);
int dtt_bn_frozen_stats(const float* running_mean, const float* running_var, float* stats, int C, float eps,
                        void* stream);

/* MaxPool 3x3 / stride 2 / padding 1 on channels-last [N][H][W][C]: the spatial part of the lip encoders'
 * MaxPool3d((1,3,3),(1,2,2),(0,1,1)) (resnet_video_encoder.py:44, lipreading/model.py:132).  idx: one byte per output
 * element (window position of the first maximum, ATen's choice); backward gathers (no atomics). */
int dtt_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int dtype, void* stream);
int dtt_maxpool3x3s2_bwd(const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C, int dtype,
                         void* stream);
/* BatchNorm + activation + the 3x3 / stride-2 / pad-1 max pool in one pass each way (lip front end,
 * resnet_video_encoder.py:38-44: BatchNorm3d -> Swish -> MaxPool3d((1,3,3),(1,2,2),(0,1,1))): forward from given
 * statistics to the pooled tensor y [N][Ho][Wo][C] and its argmax bytes; backward from the pooled gradient to the
 * gradient of the BatchNorm input and dgb = {dgamma, dbeta} (scratch as dtt_bn_act_bwd).  The pre-pool tensor and its
 * gradient are never materialised. */
int dtt_bn_act_pool_fwd(const void* x, const float* stats, const float* gamma, const float* beta, void* y,
                        unsigned char* idx, int N, int H, int W, int C, int act, int dtype, void* stream);
int dtt_bn_act_pool_bwd(const void* x, const void* dyp, const unsigned char* idx, const float* stats, const float* gamma,
                        const float* beta, float* partial, float* sums, float* dgb, void* dx, int N, int H, int W, int C,
                        int act, int training, int dtype, void* stream);

/* ShuffleNetV2 units of the FiLM model's frozen lip-reading front end, forward only (lipreading/models/
 * shufflenetv2.py:42-104 InvertedResidual, :26-40 channel_shuffle, :151-152 conv_last + globalpool; run per frame by
 * lipreading/model.py:164-174 behind DTTNetPlus/modules/video_encoder.py:41-47).  Channels-last rows; a tensor of
 * 2h channels is held as two halves of h channels, each padded with zero columns to hp (a multiple of 8): logical
 * channel j is column (j / h) * hp + j % h.  The 1x1 convolutions are dtt_gemm calls on these rows.
 *  dtt_dwconv3x3_bnrelu: depthwise 3x3, stride 1 | 2, padding 1 (shufflenetv2.py:59,69,83), x [N][H][W][C] ->
 *    y [N][Ho][Wo][C], w9c fp32 [9 taps][C]; with stats / gamma / beta (all or none) the taps are first passed through
 *    relu(batch_norm(.)) -- the BatchNorm + ReLU of the 1x1 conv in front (shufflenetv2.py:55-57), never written.
 *  dtt_pw_fold_bn: the BatchNorm behind the depthwise conv (no activation, shufflenetv2.py:60,70,84) folded into the
 *    next 1x1 conv: wout [N][K] (dtype) = w * sc[k], bias [N] = sum_k w sh[k]  (sc = gamma rstd, sh = beta - mean sc).
 *  dtt_bn_relu_shuffle: concat + channel_shuffle(groups 2) (shufflenetv2.py:92-104) fused with the BatchNorm + ReLU
 *    of the branch outputs: out [P][2 hp] logical channel 2i = A[i], 2i + 1 = B[i]; A / B rows of h channels with row
 *    strides lda / ldb (elements); a source whose stats are NULL is copied as it is (the untouched half of a stride-1
 *    unit, shufflenetv2.py:98), otherwise relu(batch_norm(.)) is applied.  Padding columns are written as zeros.
 *  dtt_bn_relu_avgpool: BatchNorm + ReLU + AvgPool2d over the S pixels of each frame (conv_last + globalpool,
 *    shufflenetv2.py:151-152,161-163): y [NF][S][C] -> out fp32 [NF][C]. */
int dtt_dwconv3x3_bnrelu(const void* x, const float* w9c, const float* stats, const float* gamma, const float* beta,
                         void* y, int N, int H, int W, int C, int stride, int dtype, void* stream);
int dtt_pw_fold_bn(const float* w, const float* stats, const float* gamma, const float* beta, void* wout, float* bias,
                   int N, int K, int dtype, void* stream);
int dtt_bn_relu_shuffle(const void* a, int lda, const float* stats_a, const float* gamma_a, const float* beta_a,
                        const void* b, int ldb, const float* stats_b, const float* gamma_b, const float* beta_b, void* out,
                        int64_t P, int h, int hp, int dtype, void* stream);
int dtt_bn_relu_avgpool(const void* y, const float* stats, const float* gamma, const float* beta, float* out, int NF,
                        int S, int C, int dtype, void* stream);
/* out[b][t][:] = linear interpolation over time of x[b][:][:] (F.interpolate(mode="linear", align_corners=False) on
 * the time axis, the FiLM block's resampling of the video embedding, block_fusion.py:18-19, and temporal_alignment.py:
 * 19-22): x fp32 [B][Tin][D] -> out fp32 [B][Tout][D]; the backward is the transposed gather (no atomics):
 * dx [B][Tin][D] from dout [B][Tout][D]. */
int dtt_lerp_time_fwd(const float* x, float* out, int B, int Tin, int Tout, int D, void* stream);
int dtt_lerp_time_bwd(const float* dout, float* dx, int B, int Tin, int Tout, int D, void* stream);

/* Gradient clip + AdamW over a table of fp32 tensors in two launches (the step after the hot path:
 * torch.optim.AdamW of train.yaml:20-23 after clip_grad_norm_ of base_trainer.py:478-490, train.yaml:59).
 * tab / chunks are device arrays; chunks[c] = {tensor index, chunk index} with chunks of
 * dtt_clip_adamw_chunk() elements; partial: nchunks floats of scratch.  A tensor whose g is NULL is
 * skipped.  max_norm <= 0 disables clipping; norm_out (may be NULL) receives the unclipped total norm. */
typedef struct {
    void* p; const void* g; float* m; float* v;
    int64_t n;
    float lr, wd;
} DttOptTensor;
int dtt_clip_adamw(const DttOptTensor* tab, int ntensors, const int32_t* chunks, int nchunks, float* partial,
                   float max_norm, float beta1, float beta2, float eps, int step, float* norm_out, void* stream);
int dtt_clip_adamw_chunk(void);
/* The same update with (a) the step count read from device memory (step_dev, int32: a captured HIP graph replays
 * with the count its caller increments on the device; NULL: use `step`), (b) an externally merged total squared
 * gradient norm (sumsq_dev; NULL: computed here) -- the sharded step of train.yaml:20-31,59 under data parallelism:
 * reduce-scatter, dtt_grad_sumsq on the shard, all-reduce of one float, this update on the shard, all-gather --
 * and (c) grad_scale folded into the clip coefficient (1 / world when the collective summed).
 * dtt_grad_sumsq: sum of squares of every gradient in tab, merged in a fixed order into sumsq_out[0]. */
int dtt_grad_sumsq(const DttOptTensor* tab, int ntensors, const int32_t* chunks, int nchunks, float* partial,
                   float* sumsq_out, void* stream);
int dtt_clip_adamw_ex(const DttOptTensor* tab, int ntensors, const int32_t* chunks, int nchunks, float* partial,
                      float max_norm, float beta1, float beta2, float eps, int step, const int32_t* step_dev,
                      const float* sumsq_dev, float grad_scale, float* norm_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
