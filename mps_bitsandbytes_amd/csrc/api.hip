// api.hip — extern "C" entry points declared in include/mbnb_hip.h: argument validation,
// status / error-string convention, dispatch to the kernel launchers.  No device allocation,
// no synchronisation; the only process-wide state is the per-(device, kernel) record of
// idempotent function attributes below (mutex-protected), the error text is thread-local.
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace mbnb {

static thread_local char g_err[512] = "";
static thread_local const char *g_kernel = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void set_kernel_name(const char *name) { g_kernel = name; }

int ensure_dyn_lds(const void *func, int bytes, const char *what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error("%s: hipGetDevice failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> done;   // (kernel, device) -> largest limit set so far
    const uint64_t key = (reinterpret_cast<uint64_t>(func) << 8) ^ (uint64_t)(dev & 0xFF);
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return MBNB_OK;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed on device %d: %s", what, bytes, dev,
                  hipGetErrorString(e));
        return (int)e;
    }
    done[key] = bytes;
    return MBNB_OK;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return MBNB_OK;
}

// launchers (quant_kernels.hip, matmul4_kernels.hip, int8_kernels.hip)
int quantize_4bit_dispatch(const void *, int, int64_t, int64_t, int64_t, int, int, const float *, uint8_t *, float *, hipStream_t);
int quantize_4bit_dq_dispatch(const void *, int, int64_t, int64_t, int64_t, int, int, uint8_t *, int8_t *, float *, hipStream_t);
int dequantize_4bit_dispatch(const uint8_t *, const AbsmaxView &, int64_t, int64_t, int64_t, int, int, int, void *, hipStream_t, int store_policy = 0);
int quantize_blockwise_dispatch(const void *, int, int64_t, int, const float *, int8_t *, float *, hipStream_t);
int dequantize_blockwise_dispatch(const int8_t *, int64_t, const float *, int, int, void *, hipStream_t);
int dequant_absmax_dispatch(const void *, int, int64_t, int64_t, const float *, int64_t, int, float *, hipStream_t);
int quantize_rowwise_dispatch(const void *, int, int64_t, int64_t, int8_t *, float *, hipStream_t);
int dequantize_rowwise_dispatch(const int8_t *, const float *, int64_t, int64_t, int, void *, hipStream_t, int store_policy = 0);
int double_quant_dispatch(const void *, int, int64_t, int64_t, int8_t *, int8_t *, float *, float *, int, int, hipStream_t);
int matmul_4bit_dispatch(const void *, int64_t, int64_t, const uint8_t *, const AbsmaxView &, int64_t, int64_t, int, int, int, const void *, int, void *, void *, int64_t, int,
                         hipStream_t);
int quantize_fp8_dispatch(const void *, int, int64_t, int64_t, uint8_t *, float *, hipStream_t);
int dequantize_fp8_dispatch(const uint8_t *, const float *, int64_t, int64_t, int, void *, hipStream_t, int store_policy = 0);
int linear_fp8_dispatch(const void *, int, int64_t, int64_t, const uint8_t *, const float *, int64_t, const void *, void *, void *, int64_t, bool, hipStream_t);
int64_t matmul4_splitk_slices(int64_t, int64_t, int64_t);
int64_t gemm_mid_workspace_bytes(int64_t, int64_t, int64_t);
int64_t gemm_small_workspace_bytes(int64_t, int64_t, int64_t, int64_t);
int64_t gemm_small8_workspace_bytes(int64_t, int64_t, int64_t);
int64_t gemm_f32_workspace_bytes(int64_t, int64_t, int64_t, int64_t);
int64_t gemm_dense_workspace_bytes(int64_t, int64_t, int64_t, int64_t);
bool gemm_dense_shape(int64_t, int64_t, int64_t, int64_t);
int64_t gemm_dense_slices(int64_t, int64_t, int64_t);
int gemm_dense_direct(const void *, const void *, int, const void *, int, void *, int64_t, int64_t, int64_t, int64_t, float *, int64_t, int, int,
                      hipStream_t);
int matmul_int8_dispatch(const int8_t *, const int8_t *, const float *, const float *, int64_t, int64_t, int64_t, int, void *, void *, hipStream_t);
int64_t matmul_int8_workspace_bytes(int64_t, int64_t, int64_t);
int linear_int8_dispatch(const void *, int, int64_t, int64_t, const int8_t *, const float *, int64_t, const void *, void *, void *, int64_t, bool, hipStream_t);
int embedding_4bit_dispatch(const int64_t *, int64_t, const uint8_t *, const float *, int64_t, int64_t, int, int, int, int64_t, int, void *, hipStream_t);
int embedding_8bit_dispatch(const int64_t *, int64_t, const int8_t *, const float *, int64_t, int64_t, int, int64_t, int, void *, hipStream_t);
int outlier_linear_dispatch(const void *, int, int64_t, int64_t, const int8_t *, const float *, int64_t, const int64_t *, int64_t, const void *, const void *, void *, void *, int64_t, hipStream_t);
int64_t outlier_linear_workspace_bytes(int64_t, int64_t, int64_t);

static bool dtype_ok(int d) { return d == MBNB_F16 || d == MBNB_BF16 || d == MBNB_F32; }
static bool qt_ok(int q) { return q == MBNB_NF4 || q == MBNB_FP4; }
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int absmax_view(const mbnb_absmax *a, const char *who, AbsmaxView &v) {
    if (!a) return fail(MBNB_ERR_ARG, "%s: absmax descriptor is NULL", who);
    if (a->absmax_i8) {
        if (!a->absmax2 || a->blocksize2 <= 0)
            return fail(MBNB_ERR_ARG, "%s: int8 absmax needs absmax2 and blocksize2 > 0", who);
        v = AbsmaxView{nullptr, a->absmax_i8, a->absmax2, a->blocksize2};
    } else {
        if (!a->absmax_f32) return fail(MBNB_ERR_ARG, "%s: absmax_f32 and absmax_i8 are both NULL", who);
        v = AbsmaxView{a->absmax_f32, nullptr, nullptr, 1};
    }
    return MBNB_OK;
}

}  // namespace mbnb

using namespace mbnb;

extern "C" {

int mbnb_abi_version(void) { return MBNB_ABI_VERSION; }

const char *mbnb_last_error(void) { return g_err; }
const char *mbnb_last_kernel(void) { return g_kernel; }

int mbnb_quantize_4bit(const void *A, int dtype, int64_t rows, int64_t cols, int64_t cols_padded, int blocksize,
                       int quant_type, const float *absmax_in, uint8_t *packed, float *absmax_out, void *stream) {
    if (!dtype_ok(dtype) || !qt_ok(quant_type)) return fail(MBNB_ERR_ARG, "quantize_4bit: bad dtype/quant_type");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "quantize_4bit: negative size");
    if (!pow2(blocksize) || blocksize > 65536)
        return fail(MBNB_ERR_ARG, "quantize_4bit: blocksize must be a power of 2 in [1, 65536], got %d", blocksize);
    if (cols_padded < cols || cols_padded % blocksize || cols_padded % 2)
        return fail(MBNB_ERR_SHAPE, "quantize_4bit: cols_padded=%lld inconsistent with cols=%lld blocksize=%d",
                    (long long)cols_padded, (long long)cols, blocksize);
    if (rows == 0 || cols_padded == 0) return MBNB_OK;
    if (!A || !packed || !absmax_out) return fail(MBNB_ERR_ARG, "quantize_4bit: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed) & 3) return fail(MBNB_ERR_ARG, "quantize_4bit: packed must be 4-byte aligned");
    return quantize_4bit_dispatch(A, dtype, rows, cols, cols_padded, blocksize, quant_type, absmax_in, packed,
                                  absmax_out, static_cast<hipStream_t>(stream));
}

int mbnb_quantize_4bit_dq(const void *A, int dtype, int64_t rows, int64_t cols, int64_t cols_padded, int blocksize,
                          int quant_type, uint8_t *packed, int8_t *absmax_codes, float *absmax2, void *stream) {
    if (!dtype_ok(dtype) || !qt_ok(quant_type)) return fail(MBNB_ERR_ARG, "quantize_4bit_dq: bad dtype/quant_type");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "quantize_4bit_dq: negative size");
    if (!pow2(blocksize) || blocksize < 8 || blocksize > 512)
        return fail(MBNB_ERR_UNSUPPORTED, "quantize_4bit_dq: blocksize must be a power of 2 in [8, 512], got %d "
                                          "(use mbnb_quantize_4bit + mbnb_quantize_blockwise)", blocksize);
    if (cols_padded < cols || cols_padded % blocksize || cols_padded % 2)
        return fail(MBNB_ERR_SHAPE, "quantize_4bit_dq: cols_padded=%lld inconsistent with cols=%lld blocksize=%d",
                    (long long)cols_padded, (long long)cols, blocksize);
    if (rows == 0 || cols_padded == 0) return MBNB_OK;
    if (!A || !packed || !absmax_codes || !absmax2) return fail(MBNB_ERR_ARG, "quantize_4bit_dq: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed) & 3) return fail(MBNB_ERR_ARG, "quantize_4bit_dq: packed must be 4-byte aligned");
    return quantize_4bit_dq_dispatch(A, dtype, rows, cols, cols_padded, blocksize, quant_type, packed, absmax_codes, absmax2,
                                     static_cast<hipStream_t>(stream));
}

int mbnb_dequantize_4bit(const uint8_t *packed, const mbnb_absmax *absmax, int64_t rows, int64_t cols,
                         int64_t cols_padded, int blocksize, int quant_type, int out_dtype, void *out, void *stream) {
    if (!dtype_ok(out_dtype) || !qt_ok(quant_type)) return fail(MBNB_ERR_ARG, "dequantize_4bit: bad dtype/quant_type");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "dequantize_4bit: negative size");
    if (!pow2(blocksize) || blocksize > 65536) return fail(MBNB_ERR_ARG, "dequantize_4bit: bad blocksize %d", blocksize);
    if (cols_padded < cols || cols_padded % blocksize || cols_padded % 2)
        return fail(MBNB_ERR_SHAPE, "dequantize_4bit: cols_padded inconsistent");
    if (rows == 0 || cols == 0) return MBNB_OK;
    AbsmaxView v;
    if (int rc = absmax_view(absmax, "dequantize_4bit", v)) return rc;
    if (!packed || !out) return fail(MBNB_ERR_ARG, "dequantize_4bit: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed) & 3) return fail(MBNB_ERR_ARG, "dequantize_4bit: packed must be 4-byte aligned");
    return dequantize_4bit_dispatch(packed, v, rows, cols, cols_padded, blocksize, quant_type, out_dtype, out,
                                    static_cast<hipStream_t>(stream));
}

int mbnb_quantize_blockwise(const void *A, int dtype, int64_t numel, int blocksize, const float *absmax_in,
                            int8_t *out, float *absmax_out, void *stream) {
    if (!dtype_ok(dtype)) return fail(MBNB_ERR_ARG, "quantize_blockwise: bad dtype");
    if (numel < 0 || blocksize <= 0 || blocksize > 65536) return fail(MBNB_ERR_ARG, "quantize_blockwise: bad size");
    if (numel == 0) return MBNB_OK;
    if (!A || !out || !absmax_out) return fail(MBNB_ERR_ARG, "quantize_blockwise: NULL pointer");
    return quantize_blockwise_dispatch(A, dtype, numel, blocksize, absmax_in, out, absmax_out,
                                       static_cast<hipStream_t>(stream));
}

int mbnb_dequantize_blockwise(const int8_t *q, int64_t numel, const float *absmax, int blocksize, int out_dtype,
                              void *out, void *stream) {
    if (!dtype_ok(out_dtype)) return fail(MBNB_ERR_ARG, "dequantize_blockwise: bad dtype");
    if (numel < 0 || blocksize <= 0) return fail(MBNB_ERR_ARG, "dequantize_blockwise: bad size");
    if (numel == 0) return MBNB_OK;
    if (!q || !absmax || !out) return fail(MBNB_ERR_ARG, "dequantize_blockwise: NULL pointer");
    return dequantize_blockwise_dispatch(q, numel, absmax, blocksize, out_dtype, out, static_cast<hipStream_t>(stream));
}

int mbnb_dequant_absmax(const void *codes, int code_kind, int64_t rows, int64_t num_blocks, const float *scales,
                        int64_t dq_blocks, int blocksize, float *out, void *stream) {
    if (code_kind < 0 || code_kind > 2) return fail(MBNB_ERR_ARG, "dequant_absmax: code_kind must be 0 (int8), 1 (uint8) or 2 (f32)");
    if (rows < 0 || num_blocks < 0 || dq_blocks < 0 || blocksize <= 0) return fail(MBNB_ERR_ARG, "dequant_absmax: bad size");
    if (rows == 0 || num_blocks == 0) return MBNB_OK;
    if (!codes || !out || (dq_blocks > 0 && !scales)) return fail(MBNB_ERR_ARG, "dequant_absmax: NULL pointer");
    return dequant_absmax_dispatch(codes, code_kind, rows, num_blocks, scales, dq_blocks, blocksize, out,
                                   static_cast<hipStream_t>(stream));
}

int mbnb_quantize_rowwise(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out, float *scales,
                          void *stream) {
    if (!dtype_ok(dtype)) return fail(MBNB_ERR_ARG, "quantize_rowwise: bad dtype");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "quantize_rowwise: negative size");
    if (rows == 0) return MBNB_OK;
    if (!A || !out || !scales) return fail(MBNB_ERR_ARG, "quantize_rowwise: NULL pointer");
    return quantize_rowwise_dispatch(A, dtype, rows, cols, out, scales, static_cast<hipStream_t>(stream));
}

int mbnb_dequantize_rowwise(const int8_t *q, const float *scales, int64_t rows, int64_t cols, int out_dtype,
                            void *out, void *stream) {
    if (!dtype_ok(out_dtype)) return fail(MBNB_ERR_ARG, "dequantize_rowwise: bad dtype");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "dequantize_rowwise: negative size");
    if (rows == 0 || cols == 0) return MBNB_OK;
    if (!q || !scales || !out) return fail(MBNB_ERR_ARG, "dequantize_rowwise: NULL pointer");
    return dequantize_rowwise_dispatch(q, scales, rows, cols, out_dtype, out, static_cast<hipStream_t>(stream));
}

int mbnb_double_quant(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out_col, int8_t *out_row,
                      float *col_stats, float *row_stats, int col_given, int row_given, void *stream) {
    if (!dtype_ok(dtype)) return fail(MBNB_ERR_ARG, "double_quant: bad dtype");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "double_quant: negative size");
    if (rows == 0 || cols == 0) return MBNB_OK;
    if (!A || !out_col || !out_row || !col_stats || !row_stats) return fail(MBNB_ERR_ARG, "double_quant: NULL pointer");
    return double_quant_dispatch(A, dtype, rows, cols, out_col, out_row, col_stats, row_stats, col_given, row_given,
                                 static_cast<hipStream_t>(stream));
}

int mbnb_matmul_4bit(const void *A, int64_t M, int64_t K, const uint8_t *packed, const mbnb_absmax *absmax, int64_t N,
                     int64_t K_weight, int blocksize, int quant_type, int w_dtype, const void *bias, int out_dtype,
                     void *out, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    if (flags & ~MBNB_MATMUL_FUSED_ONLY) return fail(MBNB_ERR_ARG, "matmul_4bit: unknown flags 0x%x", flags);
    if (!dtype_ok(w_dtype) || !dtype_ok(out_dtype) || !qt_ok(quant_type))
        return fail(MBNB_ERR_ARG, "matmul_4bit: bad dtype/quant_type");
    if (M < 0 || N < 0 || K < 0) return fail(MBNB_ERR_ARG, "matmul_4bit: negative size");
    if (!pow2(blocksize) || blocksize > 65536) return fail(MBNB_ERR_ARG, "matmul_4bit: bad blocksize %d", blocksize);
    if (K_weight < K || K_weight % blocksize || K_weight % 2)
        return fail(MBNB_ERR_SHAPE, "matmul_4bit: K_weight=%lld inconsistent with K=%lld blocksize=%d",
                    (long long)K_weight, (long long)K, blocksize);
    if (workspace_bytes < 0) return fail(MBNB_ERR_ARG, "matmul_4bit: negative workspace size");
    if (M == 0 || N == 0) return MBNB_OK;
    AbsmaxView v;
    if (int rc = absmax_view(absmax, "matmul_4bit", v)) return rc;
    if (!A || !packed || !out) return fail(MBNB_ERR_ARG, "matmul_4bit: NULL pointer");
    // the workspace travels down the dispatch as an argument (no per-call state is kept anywhere)
    return matmul_4bit_dispatch(A, M, K, packed, v, N, K_weight, blocksize, quant_type, w_dtype, bias, out_dtype, out,
                                workspace, workspace ? workspace_bytes : 0, flags, static_cast<hipStream_t>(stream));
}

// the split-K share of the query: slices x tiles x 128 x 128 f32 (128 x 128 kernel) / slices x M x N f32 (k_gemm_mid, k_gemm_small)
static int64_t matmul4_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int64_t s = matmul4_splitk_slices(M, N, K);
    const int64_t a = s > 1 ? s * ((M + 127) / 128) * ((N + 127) / 128) * 65536 : 0;
    const int64_t b = gemm_mid_workspace_bytes(M, N, K);
    const int64_t c = gemm_small_workspace_bytes(M, N, K, K);
    const int64_t ab = a > b ? a : b;
    return ab > c ? ab : c;
}

int64_t mbnb_matmul_4bit_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight, int w_dtype, int flags) {
    if (M <= 0 || N <= 0 || K <= 0 || K_weight < K || !dtype_ok(w_dtype)) return 0;
    const bool fused_only = (flags & MBNB_MATMUL_FUSED_ONLY) != 0;
    if (w_dtype == MBNB_F32) return fused_only ? 0 : gemm_f32_workspace_bytes(M, N, K, K_weight);
    const int64_t ab = matmul4_splitk_workspace_bytes(M, N, K);
    if (fused_only) return ab;
    const int64_t c = gemm_dense_workspace_bytes(M, N, K, K_weight);   // the dequantised weight (+ split-K partials): large M
    return ab > c ? ab : c;
}

int64_t mbnb_linear_int8_workspace_bytes(int64_t M, int64_t N, int64_t K, int flags) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int64_t s = matmul4_splitk_slices(M, N, K);
    const int64_t a = s > 1 ? s * ((M + 127) / 128) * ((N + 127) / 128) * 65536 : 0;   // slices x tiles x 128 x 128 f32
    const int64_t d = gemm_small8_workspace_bytes(M, N, K);   // 32 < M <= 384: slices x M x N f32 (gemm_small8.h)
    const int64_t ad = a > d ? a : d;
    if (flags & MBNB_MATMUL_FUSED_ONLY) return ad;
    const int64_t c = gemm_dense_workspace_bytes(M, N, K, K);   // large M: the dequantised weight (+ split-K partials)
    return ad > c ? ad : c;
}

int mbnb_gemm_dense(const void *A, const void *W, int dtype, const void *bias, int out_dtype, void *out, int64_t M, int64_t N,
                    int64_t K, int64_t ldw, void *workspace, int64_t workspace_bytes, int slices, void *stream) {
    // slices: bits 0-7 the K slices; bits 8-15 the diagnostic tile code (0 the library's choice; 1 256 x 128 tiles, 2 256 x 256 tiles in uniform columns,
    // 3 128 x 128 tiles, 5 / 6 / 7 a forced column-balanced grid of 32 (code + 1) | 32 code wide columns, bits 16-31 how many of the wider)
    const int tile_m = ((slices >> 8) & 0xFF) * 128, forced_cols_a = (slices >> 16) & 0xFFFF;
    slices &= 0xFF;
    if (tile_m != 0 && tile_m != 128 && tile_m != 256 && tile_m != 384 && !(tile_m >= 640 && tile_m <= 896 && slices <= 1))
        return fail(MBNB_ERR_ARG, "gemm_dense: tile code must be 0, 1, 2, 3, or 5-7 with one slice");
    if (tile_m == 384 && (K < 192 || (slices & 0xFF) > 1)) return fail(MBNB_ERR_ARG, "gemm_dense: the 128 x 128 tile needs K >= 192 and takes no K slices");
    if ((dtype != MBNB_F16 && dtype != MBNB_BF16) || !dtype_ok(out_dtype)) return fail(MBNB_ERR_ARG, "gemm_dense: bad dtype");
    if (M <= 0 || N <= 0 || K < 128 || K % 64 || ldw < K || ldw % 8) return fail(MBNB_ERR_SHAPE, "gemm_dense: bad shape");
    if (256 * ldw * 2 >= ((int64_t)1 << 31)) return fail(MBNB_ERR_SHAPE, "gemm_dense: K too large");
    if (!A || !W || !out) return fail(MBNB_ERR_ARG, "gemm_dense: NULL pointer");
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15))
        return fail(MBNB_ERR_ARG, "gemm_dense: operands must be 16-byte aligned");
    if (slices == 0) {   // the library's plan (what mbnb_matmul_4bit_ws / mbnb_linear_int8_ws run), as far as the workspace allows
        slices = (int)gemm_dense_slices(M, N, K);
        if (slices > 1 && (!workspace || workspace_bytes < (int64_t)slices * M * N * 4)) slices = 1;
    }
    if (slices < 1 || slices > 16 || (int64_t)slices * 64 > K) return fail(MBNB_ERR_ARG, "gemm_dense: bad slice count");
    if (slices > 1 && (!workspace || workspace_bytes < (int64_t)slices * M * N * 4 || (reinterpret_cast<uintptr_t>(workspace) & 15)))
        return fail(MBNB_ERR_ARG, "gemm_dense: split-K needs slices * M * N * 4 bytes of 16-byte aligned workspace");
    return gemm_dense_direct(A, W, dtype, bias, out_dtype, out, M, N, K, ldw, static_cast<float *>(workspace), slices, tile_m, forced_cols_a,
                             static_cast<hipStream_t>(stream));
}

int mbnb_gemm_dense_applies(int64_t M, int64_t N, int64_t K, int64_t ldw) {
    return (M > 0 && N > 0 && K > 0 && ldw >= K && gemm_dense_shape(M, N, K, ldw)) ? 1 : 0;
}
int64_t mbnb_gemm_dense_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K < 128 || K % 64) return 0;
    const int64_t s = gemm_dense_slices(M, N, K);
    return s > 1 ? s * M * N * 4 : 0;
}

int64_t mbnb_matmul_int8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return matmul_int8_workspace_bytes(M, N, K);
}

int mbnb_matmul_int8(const int8_t *A, const int8_t *B, const float *A_scales, const float *B_scales, int64_t M,
                     int64_t N, int64_t K, int out_dtype, void *out, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!dtype_ok(out_dtype)) return fail(MBNB_ERR_ARG, "matmul_int8: bad dtype");
    if (M < 0 || N < 0 || K < 0) return fail(MBNB_ERR_ARG, "matmul_int8: negative size");
    if (M == 0 || N == 0) return MBNB_OK;
    if (!A || !B || !A_scales || !B_scales || !out) return fail(MBNB_ERR_ARG, "matmul_int8: NULL pointer");
    if (workspace != nullptr && workspace_bytes < N * K) workspace = nullptr;   // a short workspace costs the fast path, never the result
    return matmul_int8_dispatch(A, B, A_scales, B_scales, M, N, K, out_dtype, out, workspace,
                                static_cast<hipStream_t>(stream));
}

int mbnb_linear_int8(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W, const float *W_scales, int64_t N,
                     const void *bias, void *out, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    if (flags & ~MBNB_MATMUL_FUSED_ONLY) return fail(MBNB_ERR_ARG, "linear_int8: unknown flags 0x%x", flags);
    if (!dtype_ok(dtype)) return fail(MBNB_ERR_ARG, "linear_int8: bad dtype");
    if (M < 0 || N < 0 || K < 0 || workspace_bytes < 0) return fail(MBNB_ERR_ARG, "linear_int8: negative size");
    if (M == 0 || N == 0) return MBNB_OK;
    if (!X || !W || !W_scales || !out) return fail(MBNB_ERR_ARG, "linear_int8: NULL pointer");
    return linear_int8_dispatch(X, dtype, M, K, W, W_scales, N, bias, out, workspace, workspace ? workspace_bytes : 0,
                                (flags & MBNB_MATMUL_FUSED_ONLY) != 0, static_cast<hipStream_t>(stream));
}

int mbnb_embedding_4bit(const int64_t *indices, int64_t n_indices, const uint8_t *weight_packed, const float *weight_absmax,
                        int64_t num_embeddings, int64_t embedding_dim, int blocksize, int quant_type, int has_padding,
                        int64_t padding_idx, int out_dtype, void *out, void *stream) {
    if (!dtype_ok(out_dtype)) return fail(MBNB_ERR_ARG, "embedding_4bit: bad dtype");
    if (!qt_ok(quant_type)) return fail(MBNB_ERR_ARG, "embedding_4bit: bad quant_type");
    if (n_indices < 0 || num_embeddings <= 0 || embedding_dim <= 0 || blocksize <= 0)
        return fail(MBNB_ERR_ARG, "embedding_4bit: bad size");
    if (embedding_dim % 2 != 0) return fail(MBNB_ERR_ARG, "embedding_4bit: embedding_dim must be even");
    if (n_indices == 0) return MBNB_OK;
    if (!indices || !weight_packed || !weight_absmax || !out) return fail(MBNB_ERR_ARG, "embedding_4bit: NULL pointer");
    return embedding_4bit_dispatch(indices, n_indices, weight_packed, weight_absmax, num_embeddings, embedding_dim,
                                   blocksize, quant_type, has_padding, padding_idx, out_dtype, out,
                                   static_cast<hipStream_t>(stream));
}

int mbnb_embedding_8bit(const int64_t *indices, int64_t n_indices, const int8_t *weight_int8, const float *weight_scales,
                        int64_t num_embeddings, int64_t embedding_dim, int has_padding, int64_t padding_idx, int out_dtype,
                        void *out, void *stream) {
    if (!dtype_ok(out_dtype)) return fail(MBNB_ERR_ARG, "embedding_8bit: bad dtype");
    if (n_indices < 0 || num_embeddings <= 0 || embedding_dim <= 0) return fail(MBNB_ERR_ARG, "embedding_8bit: bad size");
    if (n_indices == 0) return MBNB_OK;
    if (!indices || !weight_int8 || !weight_scales || !out) return fail(MBNB_ERR_ARG, "embedding_8bit: NULL pointer");
    return embedding_8bit_dispatch(indices, n_indices, weight_int8, weight_scales, num_embeddings, embedding_dim,
                                   has_padding, padding_idx, out_dtype, out, static_cast<hipStream_t>(stream));
}

int64_t mbnb_outlier_linear_workspace_bytes(int64_t M, int64_t K, int64_t n_outliers) {
    if (M < 0 || K < 0 || n_outliers < 0) return 0;
    return outlier_linear_workspace_bytes(M, K, n_outliers > 16 ? n_outliers : 16);   // room for at least one chunk of 16 columns
}

int mbnb_outlier_linear(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W, const float *W_scales, int64_t N,
                        const int64_t *outlier_idx, int64_t n_outliers, const void *outlier_w, const void *bias, void *out,
                        void *workspace, int64_t workspace_bytes, void *stream) {
    if (!dtype_ok(dtype)) return fail(MBNB_ERR_ARG, "outlier_linear: bad dtype");
    if (M < 0 || N < 0 || K <= 0 || n_outliers < 0) return fail(MBNB_ERR_ARG, "outlier_linear: bad size");
    if (M == 0 || N == 0) return MBNB_OK;
    if (!X || !W || !W_scales || !out || !workspace) return fail(MBNB_ERR_ARG, "outlier_linear: NULL pointer");
    if (n_outliers > 0 && (!outlier_idx || !outlier_w)) return fail(MBNB_ERR_ARG, "outlier_linear: outliers without index/weight buffers");
    if (workspace_bytes < outlier_linear_workspace_bytes(M, K, 0))
        return fail(MBNB_ERR_ARG, "outlier_linear: workspace of %lld bytes is too small (need %lld)", (long long)workspace_bytes,
                    (long long)outlier_linear_workspace_bytes(M, K, n_outliers));
    return outlier_linear_dispatch(X, dtype, M, K, W, W_scales, N, outlier_idx, n_outliers, outlier_w, bias, out, workspace,
                                   workspace_bytes, static_cast<hipStream_t>(stream));
}

int mbnb_quantize_fp8_e4m3(const void *A, int dtype, int64_t rows, int64_t cols, uint8_t *out, float *scales, void *stream) {
    if (!dtype_ok(dtype)) return fail(MBNB_ERR_ARG, "quantize_fp8_e4m3: bad dtype");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "quantize_fp8_e4m3: negative size");
    if (rows == 0 || cols == 0) return MBNB_OK;
    if (!A || !out || !scales) return fail(MBNB_ERR_ARG, "quantize_fp8_e4m3: NULL pointer");
    return quantize_fp8_dispatch(A, dtype, rows, cols, out, scales, static_cast<hipStream_t>(stream));
}

int mbnb_dequantize_fp8_e4m3(const uint8_t *q, const float *scales, int64_t rows, int64_t cols, int out_dtype, void *out,
                             void *stream) {
    if (!dtype_ok(out_dtype)) return fail(MBNB_ERR_ARG, "dequantize_fp8_e4m3: bad dtype");
    if (rows < 0 || cols < 0) return fail(MBNB_ERR_ARG, "dequantize_fp8_e4m3: negative size");
    if (rows == 0 || cols == 0) return MBNB_OK;
    if (!q || !scales || !out) return fail(MBNB_ERR_ARG, "dequantize_fp8_e4m3: NULL pointer");
    return dequantize_fp8_dispatch(q, scales, rows, cols, out_dtype, out, static_cast<hipStream_t>(stream));
}

int mbnb_linear_fp8(const void *X, int dtype, int64_t M, int64_t K, const uint8_t *W, const float *W_scales, int64_t N,
                    const void *bias, void *out, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    if (flags & ~MBNB_MATMUL_FUSED_ONLY) return fail(MBNB_ERR_ARG, "linear_fp8: unknown flags 0x%x", flags);
    if (!dtype_ok(dtype)) return fail(MBNB_ERR_ARG, "linear_fp8: bad dtype");
    if (M < 0 || N < 0 || K < 0 || workspace_bytes < 0) return fail(MBNB_ERR_ARG, "linear_fp8: negative size");
    if (M == 0 || N == 0) return MBNB_OK;
    if (!X || !W || !W_scales || !out) return fail(MBNB_ERR_ARG, "linear_fp8: NULL pointer");
    return linear_fp8_dispatch(X, dtype, M, K, W, W_scales, N, bias, out, workspace, workspace ? workspace_bytes : 0,
                               (flags & MBNB_MATMUL_FUSED_ONLY) != 0, static_cast<hipStream_t>(stream));
}

}  // extern "C"
