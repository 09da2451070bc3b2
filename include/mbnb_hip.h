/*
 * mbnb_hip.h — C ABI of the MI355X (gfx950) quantized-linear backend.
 *
 * This is the drop-in boundary for the reference's native extension
 * `mps_bitsandbytes._C` (pybind11 module, mps_bitsandbytes/csrc/mps_bitsandbytes.mm:2726-2800)
 * on the quantized-linear hot path.  Each entry point names the reference
 * binding / function it replaces.  Differences from the reference boundary,
 * by design:
 *
 *   - plain C types only: device pointers, int64 sizes, int enums, an opaque
 *     hipStream_t passed as void*; no torch / ATen types;
 *   - the CALLER allocates everything (outputs and workspaces); the library
 *     never allocates, frees or retains device memory;
 *   - every call is asynchronous on `stream` (no device synchronisation);
 *     the library is re-entrant; it keeps no per-call state (workspaces travel as
 *     arguments) and reads no environment variables.  The only process-wide state is a
 *     mutex-protected record of which (device, kernel) pairs already had their dynamic-LDS
 *     limit raised -- an idempotent attribute set once per device, so several devices can be
 *     driven from one process or from one process each.  Every launch goes to the caller's `stream`;
 *     the library owns no stream, event or device buffer;
 *   - errors are returned as an int status (0 ok, <0 argument error,
 *     >0 hipError_t); no exception crosses the ABI.  mbnb_last_error()
 *     returns a thread-local description of the last failure;
 *   - there are NO `_cpu` variants (the reference dispatches tensors that are not on its device to
 *     pure-PyTorch code, functional.py:710-767; SURVEY.md 8b lists `_cpu` entry points as optional):
 *     every pointer is a device pointer and a call without a usable HIP device fails with the
 *     hipError_t of its first launch.  The CPU restatement of the reference lives in oracle/ and
 *     is test infrastructure only -- nothing in this library links, loads or calls it.
 *
 * ABI version 2 (round 4): ONE entry point per operation, as the reference has one binding per op
 * (mm:2726-2800).  Every operation that can use scratch takes (workspace, workspace_bytes) -- NULL / 0
 * is always valid and costs speed, never the result -- and has one `*_workspace_bytes` query; the
 * matmuls take a `flags` word.  Version 1's mbnb_matmul_4bit_ws / _ex / _sync, the `_kw` / `_dt` /
 * `_splitk_` queries, mbnb_linear_int8_ws / _ex, mbnb_linear_fp8_ex and mbnb_outlier_linear_ws are
 * gone; the overlapped-decode experiments behind `_sync` live in tools/exp/parked/.
 *
 * All tensors are dense, row-major, contiguous.  All pointers are DEVICE
 * pointers on the current HIP device (the caller selects the device).
 */
#ifndef MBNB_HIP_H
#define MBNB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBNB_ABI_VERSION 2

/* element dtypes */
enum { MBNB_F16 = 0, MBNB_BF16 = 1, MBNB_F32 = 2 };
/* 4-bit code tables: functional.py:21-32 (Metal copies mm:71-85) */
enum { MBNB_NF4 = 0, MBNB_FP4 = 1 };

/* status codes (<0: argument errors detected on the host before any launch) */
enum {
    MBNB_OK = 0,
    MBNB_ERR_ARG = -1,        /* null pointer / negative size / bad enum */
    MBNB_ERR_SHAPE = -2,      /* sizes inconsistent with each other */
    MBNB_ERR_UNSUPPORTED = -3 /* valid in the reference but not implemented here */
};

int mbnb_abi_version(void);
/* thread-local, never NULL; valid until the next failing call on this thread */
const char *mbnb_last_error(void);
/* name of the kernel family the last mbnb_matmul_4bit / mbnb_linear_int8 call on this thread
 * dispatched to ("gemv", "mfma128", "generic", ...) — for tests and the bench driver */
const char *mbnb_last_kernel(void);

/* ---------------------------------------------------------------------------
 * Absmax descriptor used by the 4-bit consumers.  Either plain f32 absmax
 * (absmax_i8 == NULL), or the reference's "double quantised" form
 * (QuantState.state2, functional.py:288-292): int8 codes + one f32 absmax2 per
 * `blocksize2` (256) codes, decoded in-kernel exactly as dequantize_blockwise
 * does (functional.py:592-594): absmax[i] = (float)q[i] * (absmax2[i / blocksize2] / 127.0f).
 * ------------------------------------------------------------------------- */
typedef struct mbnb_absmax {
    const float *absmax_f32;  /* [nblocks] or NULL */
    const int8_t *absmax_i8;  /* [nblocks] or NULL */
    const float *absmax2;     /* [ceil(nblocks / blocksize2)], required with absmax_i8 */
    int32_t blocksize2;       /* 256 in the reference */
} mbnb_absmax;

/* ---------------------------------------------------------------------------
 * quantize_4bit — replaces functional.quantize_4bit (functional.py:163-303; the
 * reference's native `_C.quantize_nf4/_fp4`, mm:1892-1925 / :2738, is a dead binding).
 *   A          [rows, cols] of `dtype`; a non-2-D tensor is passed as rows = 1
 *   cols_padded = ceil(cols / blocksize) * blocksize (+ blocksize if odd)  (functional.py:219-221)
 *   absmax_in  optional caller-supplied absmax [rows * cols_padded / blocksize] (the `absmax=` kwarg)
 *   packed     out, u8 [rows * cols_padded / 2]; byte j = idx[2j] | idx[2j+1] << 4
 *   absmax_out out, f32 [rows * cols_padded / blocksize]
 * blocksize: power of two in [1, 65536].  Bit-exact vs the reference CPU path.
 * ------------------------------------------------------------------------- */
int mbnb_quantize_4bit(const void *A, int dtype, int64_t rows, int64_t cols, int64_t cols_padded,
                       int blocksize, int quant_type, const float *absmax_in, uint8_t *packed,
                       float *absmax_out, void *stream);

/* quantize_4bit with compress_statistics=True in one launch — functional.py:163-303 including :288-292
 * (`absmax, state2 = quantize_blockwise(absmax, blocksize=256)`): packed nibbles as mbnb_quantize_4bit, and instead of
 * the f32 absmax its double-quantised form: int8 codes [rows * cols_padded / blocksize] and one f32 absmax2 per 256 codes
 * [ceil(nblocks / 256)], bit-identical to mbnb_quantize_4bit followed by mbnb_quantize_blockwise(absmax, 256).
 * blocksize: power of two in [8, 512] (otherwise MBNB_ERR_UNSUPPORTED: use the two calls). */
int mbnb_quantize_4bit_dq(const void *A, int dtype, int64_t rows, int64_t cols, int64_t cols_padded,
                          int blocksize, int quant_type, uint8_t *packed, int8_t *absmax_codes,
                          float *absmax2, void *stream);

/* dequantize_4bit — replaces functional.dequantize_4bit (functional.py:306-416; dead native
 * binding `_C.dequantize_nf4/_fp4`, mm:1927-1954 / :2739).  out [rows, cols] of out_dtype. */
int mbnb_dequantize_4bit(const uint8_t *packed, const mbnb_absmax *absmax, int64_t rows,
                         int64_t cols, int64_t cols_padded, int blocksize, int quant_type,
                         int out_dtype, void *out, void *stream);

/* quantize_blockwise — functional.py:469-539 (int8, flat blocks; used for the absmax
 * double-quant at :291 with blocksize 256).  blocksize in [1, 65536], any value. */
int mbnb_quantize_blockwise(const void *A, int dtype, int64_t numel, int blocksize,
                            const float *absmax_in, int8_t *out, float *absmax_out, void *stream);

/* dequantize_blockwise — functional.py:542-600. */
int mbnb_dequantize_blockwise(const int8_t *q, int64_t numel, const float *absmax, int blocksize,
                              int out_dtype, void *out, void *stream);

/* quantize_rowwise — functional.py:607-625; scales[r] = clamp(max|row|, 1e-8) (NOT /127). */
int mbnb_quantize_rowwise(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out,
                          float *scales, void *stream);

/* dequant_absmax, legacy (non-QuantState) form — functional.py:866-889: codes [rows, num_blocks] with one f32 scale per
 * `blocksize` codes of a row, scales [rows, dq_blocks]:  out[r, j] = (float)codes[r, j] * scales[r, j / blocksize] for
 * j < dq_blocks * blocksize, 0 beyond (the reference's zeros_like).  code_kind: 0 int8, 1 uint8, 2 f32.
 * (The QuantState form of the same function is mbnb_dequantize_blockwise.) */
int mbnb_dequant_absmax(const void *codes, int code_kind, int64_t rows, int64_t num_blocks, const float *scales,
                        int64_t dq_blocks, int blocksize, float *out, void *stream);

/* dequantize_rowwise — functional.py:628-636. */
int mbnb_dequantize_rowwise(const int8_t *q, const float *scales, int64_t rows, int64_t cols,
                            int out_dtype, void *out, void *stream);

/* double_quant — functional.py:814-863 (LLM.int8 row + column statistics; dead native binding
 * `_C.double_quant`, mm:2271-2303, is a different format and is not reproduced).
 * col_given / row_given: the stats buffers already hold caller-supplied statistics. */
int mbnb_double_quant(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out_col,
                      int8_t *out_row, float *col_stats, float *row_stats, int col_given,
                      int row_given, void *stream);

/* flags word of mbnb_matmul_4bit / mbnb_linear_int8 / mbnb_linear_fp8 and of their workspace queries.
 * MBNB_MATMUL_FUSED_ONLY: never dequantise the weight into the workspace -- keep the fused decode + MFMA kernels at
 * every M (for callers that cannot spare N x K_weight x 2 bytes of scratch); the workspace then serves split-K only. */
#define MBNB_MATMUL_FUSED_ONLY 1

/* ---------------------------------------------------------------------------
 * matmul_4bit — replaces `_C.matmul_nf4` / `_C.matmul_fp4`
 * (mm:1956-2032, :2099-2151, bindings :2741-2751; call sites functional.py:743,745)
 * with the numerics of the reference's CPU branch (functional.py:752-773):
 *   out[M,N] = cast_out( round_w( A[M,K] · dequant(W)[N,K]^T + bias[N] ) )
 * where dequant(W) is rounded to `w_dtype` (= QuantState.dtype) before the
 * contraction (functional.py:382), A and bias are given in `w_dtype`, products
 * accumulate in f32, the sum is rounded once to `w_dtype` and then cast to `out_dtype`.
 *   packed  u8 [N, K_weight/2]; absmax covers [N, K_weight/blocksize]
 *   K       activation width (= QuantState.shape[1]); K_weight >= K is the padded row length
 *   bias    optional [N] of w_dtype
 *   workspace  optional scratch of mbnb_matmul_4bit_workspace_bytes(...) bytes, 256-byte aligned; a shorter (or no)
 *           workspace costs the fast path, never the result.  Nothing in it survives the call.
 * Dispatch (16-bit weights, blocksize >= 32, 16-byte aligned rows): M = 1 -> wave-per-row GEMV (HBM-bound); 2 <= M <= 32
 * -> weight-streaming skinny MFMA kernel; 32 < M <= 512 -> k_gemm_small (weights decoded registers -> registers, K split
 * over f32 partials in the workspace); from 256 rows and 1.5 M outputs up, with a workspace that holds it: the weight
 * dequantised ONCE, [N, K_weight] in the weight dtype (the bits mbnb_dequantize_4bit writes), then the dense 256 x 256 MFMA
 * GEMM of csrc/gemm_dense.h -- the reference's own large-batch path (functional.py:753-767: dequantize_4bit, then
 * F.linear); without one (or with MBNB_MATMUL_FUSED_ONLY): the fused decode + MFMA kernels (csrc/gemm_fused4.h, gemm256.h).
 * f32 weights: with a workspace dequantise once + f32 MFMA GEMM (csrc/gemm_f32.hip), else the generic kernel; blocksize < 32,
 * K % 8 != 0 -> generic kernel.
 * ------------------------------------------------------------------------- */
int64_t mbnb_matmul_4bit_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight, int w_dtype, int flags);
int mbnb_matmul_4bit(const void *A, int64_t M, int64_t K, const uint8_t *packed,
                     const mbnb_absmax *absmax, int64_t N, int64_t K_weight, int blocksize,
                     int quant_type, int w_dtype, const void *bias, int out_dtype, void *out,
                     void *workspace, int64_t workspace_bytes, int flags, void *stream);

/* ---------------------------------------------------------------------------
 * matmul_int8 — replaces `_C.matmul_int8` (mm:1789-1834, kernel mm:155-196) /
 * functional.matmul_int8 (functional.py:788-793):
 *   out[M,N] = cast( int32(A[M,K] · B[K,N]) * (A_scales[m]/127) * (B_scales[n]/127) )
 * on the int8 MFMA.  B is read as the reference passes it, [K, N] row-major.  Large aligned problems (K % 128 == 0,
 * K >= 256, N % 16 == 0, 16-byte aligned A and B, >= 96 output tiles of 256 x 256) run in ONE launch WITHOUT a workspace:
 * the kernel transposes while it reads its LDS image (ds_read_b64_tr_b8; csrc/gemm_i8_inplace.h) and the query returns 0.
 * Smaller or unaligned problems need mbnb_matmul_int8_workspace_bytes(M, N, K) (= N*K) bytes to re-lay B out K-contiguous
 * for any MFMA kernel (workspace == NULL or too short: slow generic kernel).
 * ------------------------------------------------------------------------- */
int64_t mbnb_matmul_int8_workspace_bytes(int64_t M, int64_t N, int64_t K);
int mbnb_matmul_int8(const int8_t *A, const int8_t *B, const float *A_scales,
                     const float *B_scales, int64_t M, int64_t N, int64_t K, int out_dtype,
                     void *out, void *workspace, int64_t workspace_bytes, void *stream);

/* The dense half of the large-M path on its own: out[M, N] = A[M, K] * W[N, ldw]^T (+ bias) for an f16 / bf16 weight that is
 * already in the compute dtype (rows ldw >= K elements apart) -- the F.linear of functional.py:767 on the tensor
 * functional.py:756 produced.  mbnb_matmul_4bit calls it internally; exported for callers that keep a dequantised weight
 * (Linear8bit's cache, nn/linear8bit.py:70-85; Linear4bit.dequantize(), nn/linear4bit.py:204) and for tools/.  K % 64 == 0,
 * K >= 128, ldw % 8 == 0, A and W 16-byte aligned; f32 accumulation, one rounding to `dtype`, then the cast to out_dtype.
 * `slices`: bits 0-7 the number of K slices (> 1 splits K over slices * M * N * 4 bytes of workspace, partials added in slice
 * order; 1 needs no workspace); 0 = the library's own plan for this shape (what mbnb_matmul_4bit / mbnb_linear_int8 run after
 * their dequantise pass, so the caller gets the same bits); its partials need mbnb_gemm_dense_workspace_bytes(M, N, K) bytes
 * (0 when the plan does not split), a shorter workspace means one slice.  Bits 8-15: diagnostic tile selector (0 = the plan's).
 * The tile shape does not change the result's bits, the slice count does.
 * mbnb_gemm_dense_applies: 1 when the decode-once path serves this shape (from 256 rows and 1.5 M outputs up), else 0. */
int mbnb_gemm_dense_applies(int64_t M, int64_t N, int64_t K, int64_t ldw);
int64_t mbnb_gemm_dense_workspace_bytes(int64_t M, int64_t N, int64_t K);
int mbnb_gemm_dense(const void *A, const void *W, int dtype, const void *bias, int out_dtype, void *out,
                    int64_t M, int64_t N, int64_t K, int64_t ldw, void *workspace, int64_t workspace_bytes,
                    int slices, void *stream);

/* ---------------------------------------------------------------------------
 * linear_int8 — replaces `_C.linear_int8` (mm:1836-1886, kernel mm:203-305) with the numerics
 * of Linear8bit.forward (nn/linear8bit.py:70-102):
 *   out[M,N] = X[M,K] · round_dtype(W_i8[N,K] * (scales[n]/127))^T + bias[N]
 * X, bias, out in `dtype` (f16 / bf16 / f32); f32 accumulation; one rounding.
 * workspace: mbnb_linear_int8_workspace_bytes(M, N, K, flags) bytes (0 = not needed; 256-byte aligned): split-K partials for
 * mid-sized M; from 256 rows and 1.5 M outputs up the weight dequantised ONCE (dequantize_rowwise's bits, [N, K] in `dtype`)
 * followed by the dense MFMA GEMM of csrc/gemm_dense.h -- the reference's own two steps.  NULL / 0: the fused W8A16 kernels.
 * mbnb_linear_fp8 takes the same workspace and flags.
 * ------------------------------------------------------------------------- */
int64_t mbnb_linear_int8_workspace_bytes(int64_t M, int64_t N, int64_t K, int flags);
int mbnb_linear_int8(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W,
                     const float *W_scales, int64_t N, const void *bias, void *out, void *workspace,
                     int64_t workspace_bytes, int flags, void *stream);

/* ---------------------------------------------------------------------------
 * embedding_4bit — replaces `_C.embedding_4bit_nf4` / `_C.embedding_4bit_fp4` (host mm:2309-2388, kernels
 * mm:1213-1275, bindings :2765-2770) with the numerics of Embedding4bit.forward's Python path
 * (nn/embedding.py:83-138): out[t, :] = dequantize(row indices[t]) = code[nibble] * absmax[row, k / blocksize]
 * in f32, rounded to `out_dtype`; rows equal to padding_idx (when has_padding) are zeros (:133-136).
 *   weight_packed u8 [num_embeddings, embedding_dim/2], weight_absmax f32 [num_embeddings, ceil(dim/blocksize)],
 *   indices int64 [n_indices] (the reference casts to int32; int64 is torch's native index type), out [n_indices, dim].
 * Indices must lie in [0, num_embeddings): out-of-range rows are written as zeros (the reference raises on the host).
 * ------------------------------------------------------------------------- */
int mbnb_embedding_4bit(const int64_t *indices, int64_t n_indices, const uint8_t *weight_packed,
                        const float *weight_absmax, int64_t num_embeddings, int64_t embedding_dim, int blocksize,
                        int quant_type, int has_padding, int64_t padding_idx, int out_dtype, void *out, void *stream);

/* embedding_8bit — replaces `_C.embedding_8bit` (host mm:2390-2427, kernel mm:1277-1294) with the numerics of
 * Embedding8bit.forward's Python path (nn/embedding.py:255-268), arithmetic in `out_dtype` T:
 *   out[t, k] = RNE_T( float(W[row, k]) * float(RNE_T(scales[row] / 127)) ),  zeros for padding rows. */
int mbnb_embedding_8bit(const int64_t *indices, int64_t n_indices, const int8_t *weight_int8, const float *weight_scales,
                        int64_t num_embeddings, int64_t embedding_dim, int has_padding, int64_t padding_idx,
                        int out_dtype, void *out, void *stream);

/* ---------------------------------------------------------------------------
 * outlier_linear — OutlierAwareLinear.forward (nn/outlier_aware.py:84-146; the reference has no native binding
 * for it, the module is the boundary).  X [M,K], outlier_w [N, n_outliers], bias [N], out [M,N] in `dtype`:
 *   q, s   = quantize_rowwise(X with the outlier columns removed)                       (:127-131)
 *   main   = int32(q . W_i8[N,K]^T) * (s[m]/127) * (W_scales[n]/127)  rounded to dtype  (:133-138, on the int8 MFMA;
 *            the reference multiplies dtype-rounded dequantised operands instead: <= 4e-4 (f16) / 1.3e-3 (bf16) rel.)
 *   out    = RNE(RNE(main + RNE(X[:, outlier_idx] . outlier_w^T)) + bias)               (:141-143, :110-111)
 * n_outliers may be 0 (pure INT8 path, :100-105).  `workspace` (required) holds the int8 activations, their row scales and
 * the compact outlier activations [M, 16 * ceil(n_outliers / 16)]: mbnb_outlier_linear_workspace_bytes(M, K, n_outliers)
 * bytes let the 256 x 256 kernels fold ANY number of outlier columns (and the bias) into their epilogue (at most 64 columns
 * on >= 96 tiles: the four-wave kernel, one 16 x 16 x 32 MFMA per output fragment and 32 columns; otherwise the eight-wave
 * kernel, one MFMA per tile and 16 columns).  A workspace sized for fewer columns (at least the n_outliers = 0 query) is
 * valid: the outlier term then runs as a separate pass over the output (same results).
 * ------------------------------------------------------------------------- */
int64_t mbnb_outlier_linear_workspace_bytes(int64_t M, int64_t K, int64_t n_outliers);
int mbnb_outlier_linear(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W, const float *W_scales,
                        int64_t N, const int64_t *outlier_idx, int64_t n_outliers, const void *outlier_w,
                        const void *bias, void *out, void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * FP8 E4M3 in the reference's own format — replaces `_C.quantize_fp8_e4m3` / `_C.dequantize_fp8_e4m3` /
 * `_C.matmul_fp8_e4m3` (host mm:2157-2260, kernels mm:94-140, :1010-1180) with the numerics of the reference's Python path
 * (functional.py:643-673, :796-807, :1086-1215), bit-exact for quantize / dequantize:
 *   scales[r] = clamp(max|row| / 448, 1e-12);  byte = encode(clamp(x / scale, +-448))  with the reference's encoder
 *   (exponent = floor(log2 |v|) as torch evaluates it, mantissa = trunc((|v|/2^e - 1) * 8 + 0.5) without carry,
 *   subnormals flushed to signed zero, |v| >= 256 -> 0x77, NaN -> 0x7F) — not the OCP conversion.
 *   linear_fp8: out[M,N] = X[M,K] . round_dtype(decode(W[N,K]) * scales[n])^T + bias   (LinearFP8.forward,
 *   nn/linear_fp8.py:74-103); same kernels, workspace query and flags as mbnb_linear_int8.
 * ------------------------------------------------------------------------- */
int mbnb_quantize_fp8_e4m3(const void *A, int dtype, int64_t rows, int64_t cols, uint8_t *out, float *scales,
                           void *stream);
int mbnb_dequantize_fp8_e4m3(const uint8_t *q, const float *scales, int64_t rows, int64_t cols, int out_dtype,
                             void *out, void *stream);
int mbnb_linear_fp8(const void *X, int dtype, int64_t M, int64_t K, const uint8_t *W, const float *W_scales,
                    int64_t N, const void *bias, void *out, void *workspace, int64_t workspace_bytes, int flags,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MBNB_HIP_H */
