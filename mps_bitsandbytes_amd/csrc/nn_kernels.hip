// nn_kernels.hip — gfx950 kernels for the quantized embedding lookups and the outlier-aware INT8 linear
// (reference: nn/embedding.py:83-138, :255-268, nn/outlier_aware.py:84-146).
//
//   k_embedding_4bit   gather + dequantize rows of a 4-bit table  -> 16-bit / f32 rows   (HBM-bound byte work)
//   k_embedding_8bit   gather + dequantize rows of an int8 table
//   k_quantize_rowwise_masked   quantize_rowwise over the non-outlier columns (outlier columns -> 0)
//   k_outlier_add      out = ((out + x[:, outlier_idx] . outlier_w^T) + bias), each step rounded as the reference
//
// The int8 contraction itself is the MFMA kernel of int8_kernels.hip (matmul_int8_nt_dispatch).
#include "common.h"

namespace mbnb {

int check_launch(const char *what);
void set_error(const char *fmt, ...);
void set_kernel_name(const char *name);
int matmul_int8_nt_dispatch(const int8_t *A, const int8_t *Bt, const float *sA, const float *sB, int64_t M, int64_t N,
                            int64_t K, int out_dtype, void *out, hipStream_t st);

// ------------------------------------------------------------------------------------ embeddings
// One workgroup per looked-up row.  A thread decodes 4 packed bytes (8 values) per trip: u32 load, table
// lookups (16-entry LDS table), value = code * absmax[row, k / blocksize] in f32 -> RNE to the output type
// (the reference's dequantize arithmetic, functional.py:388-416), one 16-byte store.  Rows equal to
// padding_idx are written as zeros (nn/embedding.py:133-136); so are rows whose index is out of range
// (the reference raises on the host; a device kernel cannot).
template <typename OutT, int QT>
__global__ __launch_bounds__(256) void k_embedding_4bit(const int64_t *__restrict__ idx, const uint8_t *__restrict__ packed,
                                                       const float *__restrict__ absmax, int64_t num, int64_t dim,
                                                       int blocksize, int has_pad, int64_t pad, OutT *__restrict__ out,
                                                       bool vec_ok) {
    __shared__ float lut[16];
    fill_code_lut<QT>(lut, threadIdx.x);
    __syncthreads();
    const int64_t t = blockIdx.x;
    const int64_t r = idx[t];
    const bool zero = r < 0 || r >= num || (has_pad && r == pad);
    const int64_t rr = zero ? 0 : r;
    const int64_t nblk = (dim + blocksize - 1) / blocksize;
    const uint8_t *prow = packed + rr * (dim >> 1);
    const float *arow = absmax + rr * nblk;
    OutT *orow = out + t * dim;
    const int64_t nvec = vec_ok ? dim / 8 : 0;
    for (int64_t g = threadIdx.x; g < nvec; g += blockDim.x) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(prow + g * 4);
        __attribute__((aligned(16))) OutT o[8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t k = g * 8 + 2 * j;
            const float a0 = arow[k / blocksize], a1 = arow[(k + 1) / blocksize];
            const uint32_t b = (w >> (8 * j)) & 0xFFu;
            o[2 * j] = from_f32<OutT>(zero ? 0.0f : lut[b & 15] * a0);
            o[2 * j + 1] = from_f32<OutT>(zero ? 0.0f : lut[b >> 4] * a1);
        }
        if constexpr (sizeof(OutT) == 2) {
            *reinterpret_cast<u32x4 *>(orow + g * 8) = *reinterpret_cast<const u32x4 *>(o);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) orow[g * 8 + j] = o[j];
        }
    }
    for (int64_t k = nvec * 8 + threadIdx.x; k < dim; k += blockDim.x) {
        const uint8_t b = prow[k >> 1];
        const int nib = (k & 1) ? (b >> 4) : (b & 15);
        orow[k] = from_f32<OutT>(zero ? 0.0f : lut[nib] * arow[k / blocksize]);
    }
}

// Embedding8bit.forward arithmetic (nn/embedding.py:259-262), in the output type T:
//   q.to(T) * (scale / 127.0).to(T)  =  RNE_T( float(q) * float(RNE_T(scale / 127.0f)) )
template <typename OutT>
__global__ __launch_bounds__(256) void k_embedding_8bit(const int64_t *__restrict__ idx, const int8_t *__restrict__ W,
                                                       const float *__restrict__ scales, int64_t num, int64_t dim,
                                                       int has_pad, int64_t pad, OutT *__restrict__ out, bool vec_ok) {
    const int64_t t = blockIdx.x;
    const int64_t r = idx[t];
    const bool zero = r < 0 || r >= num || (has_pad && r == pad);
    const int64_t rr = zero ? 0 : r;
    const float s = to_f32(from_f32<OutT>(scales[rr] / 127.0f));
    const int8_t *wrow = W + rr * dim;
    OutT *orow = out + t * dim;
    const int64_t nvec = vec_ok ? dim / 8 : 0;
    for (int64_t g = threadIdx.x; g < nvec; g += blockDim.x) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(wrow + g * 8);
        __attribute__((aligned(16))) OutT o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int q = (int)(int8_t)((w[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            o[j] = from_f32<OutT>(zero ? 0.0f : (float)q * s);
        }
        if constexpr (sizeof(OutT) == 2) {
            *reinterpret_cast<u32x4 *>(orow + g * 8) = *reinterpret_cast<const u32x4 *>(o);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) orow[g * 8 + j] = o[j];
        }
    }
    for (int64_t k = nvec * 8 + threadIdx.x; k < dim; k += blockDim.x)
        orow[k] = from_f32<OutT>(zero ? 0.0f : (float)wrow[k] * s);
}

int embedding_4bit_dispatch(const int64_t *idx, int64_t n_idx, const uint8_t *packed, const float *absmax, int64_t num,
                            int64_t dim, int blocksize, int qt, int has_pad, int64_t pad, int out_dtype, void *out,
                            hipStream_t st) {
    // vector path: 4 packed bytes per thread need 4-byte aligned rows, 16-byte stores need dim % 8 == 0
    const bool vec_ok = (dim % 8 == 0) && ((reinterpret_cast<uintptr_t>(packed) & 3) == 0) &&
                        ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const unsigned threads = dim >= 2048 ? 256 : (dim >= 1024 ? 128 : 64);
#define MBNB_E4(OT, QT) \
    hipLaunchKernelGGL((k_embedding_4bit<OT, QT>), dim3((unsigned)n_idx), dim3(threads), 0, st, idx, packed, absmax, num, dim, \
                       blocksize, has_pad, pad, static_cast<OT *>(out), vec_ok)
    if (qt == MBNB_NF4) {
        switch (out_dtype) {
            case MBNB_F16: MBNB_E4(f16_t, MBNB_NF4); break;
            case MBNB_BF16: MBNB_E4(bf16_t, MBNB_NF4); break;
            default: MBNB_E4(float, MBNB_NF4); break;
        }
    } else {
        switch (out_dtype) {
            case MBNB_F16: MBNB_E4(f16_t, MBNB_FP4); break;
            case MBNB_BF16: MBNB_E4(bf16_t, MBNB_FP4); break;
            default: MBNB_E4(float, MBNB_FP4); break;
        }
    }
#undef MBNB_E4
    set_kernel_name("embedding4");
    return check_launch("embedding_4bit");
}

int embedding_8bit_dispatch(const int64_t *idx, int64_t n_idx, const int8_t *W, const float *scales, int64_t num, int64_t dim,
                            int has_pad, int64_t pad, int out_dtype, void *out, hipStream_t st) {
    const bool vec_ok = (dim % 8 == 0) && ((reinterpret_cast<uintptr_t>(W) & 7) == 0) &&
                        ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const unsigned threads = dim >= 2048 ? 256 : (dim >= 1024 ? 128 : 64);
#define MBNB_E8(OT) \
    hipLaunchKernelGGL((k_embedding_8bit<OT>), dim3((unsigned)n_idx), dim3(threads), 0, st, idx, W, scales, num, dim, has_pad, \
                       pad, static_cast<OT *>(out), vec_ok)
    switch (out_dtype) {
        case MBNB_F16: MBNB_E8(f16_t); break;
        case MBNB_BF16: MBNB_E8(bf16_t); break;
        default: MBNB_E8(float); break;
    }
#undef MBNB_E8
    set_kernel_name("embedding8");
    return check_launch("embedding_8bit");
}

// ------------------------------------------------------------------------------------ outlier-aware linear
__global__ void k_set_mask(const int64_t *__restrict__ idx, int64_t n, int64_t K, uint8_t *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= 0 && idx[i] < K) mask[idx[i]] = 1;
}

// quantize_rowwise (functional.py:607-625) of x[:, non-outlier columns] (nn/outlier_aware.py:121-131), written at
// full width with zeros in the outlier columns: a zero neither raises the row absmax nor contributes to the
// integer dot product, so the contraction can run over the whole K with the stored int8 weight.
template <typename T>
__global__ __launch_bounds__(256) void k_quantize_rowwise_masked(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                                const uint8_t *__restrict__ mask, int8_t *__restrict__ out,
                                                                float *__restrict__ scales) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const T *row = A + r * cols;
    float am = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) am = fmaxf(am, mask[c] ? 0.0f : fabsf(to_f32(row[c])));
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);
    if (threadIdx.x == 0) scales[r] = am;
    const float s = rscale127(am);
    int8_t *orow = out + r * cols;
    for (int64_t c = threadIdx.x; c < cols; c += 256) orow[c] = mask[c] ? (int8_t)0 : quant_i8(to_f32(row[c]), s);
}

// out[m, n] <- RNE(RNE(out[m, n] + RNE(sum_j x[m, idx_j] * ow[n, j])) + bias[n])   (nn/outlier_aware.py:141-143, :110-111);
// without outliers only the bias add.  One thread per output element; the x values of a row are shared by the
// whole workgroup (one m per block row), ow[n, :] is contiguous.
template <typename T>
__global__ __launch_bounds__(256) void k_outlier_add(const T *__restrict__ X, int64_t M, int64_t K, int64_t N,
                                                    const int64_t *__restrict__ oidx, int64_t n_out,
                                                    const T *__restrict__ ow, const T *__restrict__ bias,
                                                    T *__restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t m = blockIdx.y;
    if (n >= N) return;
    float v = to_f32(out[m * N + n]);
    if (n_out > 0) {
        float o = 0.0f;
        for (int64_t j = 0; j < n_out; j++) o = fmaf(to_f32(X[m * K + oidx[j]]), to_f32(ow[n * n_out + j]), o);
        v = to_f32(from_f32<T>(v + to_f32(from_f32<T>(o))));
    }
    if (bias) v = to_f32(from_f32<T>(v + to_f32(bias[n])));
    out[m * N + n] = from_f32<T>(v);
}

template <typename T>
static int launch_outlier_linear(const void *X, int64_t M, int64_t K, const int8_t *W, const float *w_scales, int64_t N,
                                 const int64_t *oidx, int64_t n_out, const void *ow, const void *bias, void *out,
                                 void *workspace, int dtype, hipStream_t st) {
    // workspace: [x_q int8 M*K | pad to 256][x_scales f32 M | pad to 256][mask u8 K]
    char *ws = static_cast<char *>(workspace);
    int8_t *xq = reinterpret_cast<int8_t *>(ws);
    const int64_t off_s = (M * K + 255) & ~(int64_t)255;
    float *xs = reinterpret_cast<float *>(ws + off_s);
    const int64_t off_m = off_s + ((4 * M + 255) & ~(int64_t)255);
    uint8_t *mask = reinterpret_cast<uint8_t *>(ws + off_m);
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)K, st);
    if (e != hipSuccess) {
        set_error("outlier_linear: hipMemsetAsync failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    if (n_out > 0) hipLaunchKernelGGL(k_set_mask, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, oidx, n_out, K, mask);
    hipLaunchKernelGGL(k_quantize_rowwise_masked<T>, dim3((unsigned)M), dim3(256), 0, st, static_cast<const T *>(X), M, K, mask, xq, xs);
    int rc = check_launch("outlier_linear(quantize)");
    if (rc) return rc;
    rc = matmul_int8_nt_dispatch(xq, W, xs, w_scales, M, N, K, dtype, out, st);
    if (rc) return rc;
    if (n_out > 0 || bias) {
        dim3 grid((unsigned)((N + 255) / 256), (unsigned)M);
        hipLaunchKernelGGL(k_outlier_add<T>, grid, dim3(256), 0, st, static_cast<const T *>(X), M, K, N, oidx, n_out,
                           static_cast<const T *>(ow), static_cast<const T *>(bias), static_cast<T *>(out));
        rc = check_launch("outlier_linear(outlier add)");
    }
    return rc;
}

int outlier_linear_dispatch(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W, const float *w_scales, int64_t N,
                            const int64_t *oidx, int64_t n_out, const void *ow, const void *bias, void *out, void *workspace,
                            hipStream_t st) {
    switch (dtype) {
        case MBNB_F16: return launch_outlier_linear<f16_t>(X, M, K, W, w_scales, N, oidx, n_out, ow, bias, out, workspace, dtype, st);
        case MBNB_BF16: return launch_outlier_linear<bf16_t>(X, M, K, W, w_scales, N, oidx, n_out, ow, bias, out, workspace, dtype, st);
        default: return launch_outlier_linear<float>(X, M, K, W, w_scales, N, oidx, n_out, ow, bias, out, workspace, dtype, st);
    }
}

}  // namespace mbnb
