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
                            int64_t K, int out_dtype, void *out, hipStream_t st, const OutlierEpilogue *ep, bool *ep_done);

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
    const int nvec = vec_ok ? (int)(dim / 8) : 0;
    const bool one_am = (blocksize % 8) == 0;   // the 8 values of a group share one absmax block
    for (int g = threadIdx.x; g < nvec; g += blockDim.x) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(prow + g * 4);
        __attribute__((aligned(16))) OutT o[8];
        const float am = one_am ? arow[(g * 8) / blocksize] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = g * 8 + 2 * j;
            const float a0 = one_am ? am : arow[k / blocksize], a1 = one_am ? am : arow[(k + 1) / blocksize];
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
    for (int64_t k = (int64_t)nvec * 8 + threadIdx.x; k < dim; k += blockDim.x) {
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
// quantize_rowwise (functional.py:607-625) of x[:, non-outlier columns] (nn/outlier_aware.py:121-131), written at
// full width with zeros in the outlier columns: a zero neither raises the row absmax nor contributes to the
// integer dot product, so the contraction can run over the whole K with the stored int8 weight.
template <typename T>
__global__ __launch_bounds__(256) void k_quantize_rowwise_masked(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                                const int64_t *__restrict__ oidx, int64_t n_out,
                                                                int8_t *__restrict__ out, float *__restrict__ scales,
                                                                T *__restrict__ xo, int64_t ldxo, bool vec_ok) {
    // column mask of the outlier set, rebuilt per workgroup in LDS (cols bytes, rounded up to 8): cheaper than a
    // global mask + two extra launches for the handful of outlier columns of a layer
    extern __shared__ __attribute__((aligned(8))) uint8_t mask[];
    __shared__ float red[4];
    for (int64_t c = (int64_t)threadIdx.x * 8; c < cols; c += 256 * 8) *reinterpret_cast<u32x2 *>(mask + c) = u32x2{0u, 0u};
    __syncthreads();
    for (int64_t j = threadIdx.x; j < n_out; j += 256)
        if (oidx[j] >= 0 && oidx[j] < cols) mask[oidx[j]] = 1;
    __syncthreads();
    const int64_t r = blockIdx.x;
    const T *row = A + r * cols;
    auto load8m = [&](int64_t k0, float (&x)[8]) {   // 8 values with the outlier columns zeroed
        const u32x2 mk = *reinterpret_cast<const u32x2 *>(mask + k0);
        if constexpr (sizeof(T) == 2) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(row + k0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[2 * j] = unpack_lo<T>(v[j]);
                x[2 * j + 1] = unpack_hi<T>(v[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = to_f32(row[k0 + j]);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((mk[j >> 2] >> (8 * (j & 3))) & 0xFFu) x[j] = 0.0f;
    };
    float am = 0.0f;
    const int64_t nvec = vec_ok ? cols / 8 : 0;
    for (int64_t g = threadIdx.x; g < nvec; g += 256) {
        float x[8];
        load8m(g * 8, x);
#pragma unroll
        for (int j = 0; j < 8; j++) am = fmaxf(am, fabsf(x[j]));
    }
    for (int64_t c = nvec * 8 + threadIdx.x; c < cols; c += 256) am = fmaxf(am, mask[c] ? 0.0f : fabsf(to_f32(row[c])));
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);
    if (threadIdx.x == 0) scales[r] = am;
    // the row's outlier activations, compact and zero padded to a multiple of 16 (ldxo): the GEMM epilogue reads them as
    // 16-byte fragments, one chunk of 16 outliers per MFMA
    if (xo != nullptr)
        for (int64_t j = threadIdx.x; j < ldxo; j += 256) {
            const int64_t c = j < n_out ? oidx[j] : -1;
            xo[r * ldxo + j] = (c >= 0 && c < cols) ? row[c] : from_f32<T>(0.0f);
        }
    const float s = rscale127(am);
    int8_t *orow = out + r * cols;
    for (int64_t g = threadIdx.x; g < nvec; g += 256) {
        float x[8];
        load8m(g * 8, x);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lo |= (uint32_t)(uint8_t)quant_i8(x[j], s) << (8 * j);
            hi |= (uint32_t)(uint8_t)quant_i8(x[4 + j], s) << (8 * j);
        }
        *reinterpret_cast<u32x2 *>(orow + g * 8) = u32x2{lo, hi};
    }
    for (int64_t c = nvec * 8 + threadIdx.x; c < cols; c += 256) orow[c] = mask[c] ? (int8_t)0 : quant_i8(to_f32(row[c]), s);
}

// The same for 16-bit rows of up to 8192 columns on the vector path: the row's 16-byte pieces are requested BEFORE the column
// mask is built (its two barriers and the index loads sit under the row's HBM round trip) and stay in registers, with the
// outlier columns zeroed in place, for both the absmax pass and the quantise pass -- the row is read from memory once.
template <typename T>
__global__ __launch_bounds__(256) void k_quantize_rowwise_masked_regs(const T *__restrict__ A, int64_t cols,
                                                                     const int64_t *__restrict__ oidx, int64_t n_out,
                                                                     int8_t *__restrict__ out, float *__restrict__ scales,
                                                                     T *__restrict__ xo, int64_t ldxo) {
    static_assert(sizeof(T) == 2, "16-bit rows");
    extern __shared__ __attribute__((aligned(8))) uint8_t mask[];
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const T *row = A + r * cols;
    const int nvec = (int)(cols / 8);   // cols % 8 == 0, nvec <= 1024
    u32x4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int g = threadIdx.x + 256 * i;
        raw[i] = g < nvec ? *reinterpret_cast<const u32x4 *>(row + (int64_t)g * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    for (int64_t c = (int64_t)threadIdx.x * 8; c < cols; c += 256 * 8) *reinterpret_cast<u32x2 *>(mask + c) = u32x2{0u, 0u};
    __syncthreads();
    for (int64_t j = threadIdx.x; j < n_out; j += 256)
        if (oidx[j] >= 0 && oidx[j] < cols) mask[oidx[j]] = 1;
    __syncthreads();
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int g = threadIdx.x + 256 * i;
        if (g < nvec) {
            const u32x2 mk = *reinterpret_cast<const u32x2 *>(mask + (int64_t)g * 8);
#pragma unroll
            for (int j = 0; j < 4; j++) {   // dword j = elements 2j (low half) and 2j + 1 (high half)
                const uint32_t m2 = (mk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                if (m2 & 0x00FFu) raw[i][j] &= 0xFFFF0000u;
                if (m2 & 0xFF00u) raw[i][j] &= 0x0000FFFFu;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) am = fmaxf(am, fmaxf(fabsf(unpack_lo<T>(raw[i][j])), fabsf(unpack_hi<T>(raw[i][j]))));
    }
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);
    if (threadIdx.x == 0) scales[r] = am;
    if (xo != nullptr)
        for (int64_t j = threadIdx.x; j < ldxo; j += 256) {
            const int64_t c = j < n_out ? oidx[j] : -1;
            xo[r * ldxo + j] = (c >= 0 && c < cols) ? row[c] : from_f32<T>(0.0f);
        }
    const float s = rscale127(am);
    int8_t *orow = out + r * cols;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int g = threadIdx.x + 256 * i;
        if (g >= nvec) continue;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            lo |= (uint32_t)(uint8_t)quant_i8(unpack_lo<T>(raw[i][j]), s) << (16 * j);
            lo |= (uint32_t)(uint8_t)quant_i8(unpack_hi<T>(raw[i][j]), s) << (16 * j + 8);
            hi |= (uint32_t)(uint8_t)quant_i8(unpack_lo<T>(raw[i][2 + j]), s) << (16 * j);
            hi |= (uint32_t)(uint8_t)quant_i8(unpack_hi<T>(raw[i][2 + j]), s) << (16 * j + 8);
        }
        // write-through: the int8 rows are read next by the GEMM launch, on other XCDs (tools/exp/ab_outlier.py)
        const u32x2 pv = u32x2{lo, hi};
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(orow + (int64_t)g * 8), "v"(pv) : "memory");
    }
}

// out[m, n] <- RNE(RNE(out[m, n] + RNE(sum_j x[m, idx_j] * ow[n, j])) + bias[n])   (nn/outlier_aware.py:141-143, :110-111);
// without outliers only the bias add.  A workgroup owns 16 rows x 512 columns, a thread 16 rows x 2 consecutive
// columns (4-byte accesses to `out` for 16-bit types; its outlier weights come as 16-byte loads when n_out % 8 == 0).  The rows' outlier activations are gathered once into LDS
// (chunks of 16 outliers, broadcast ds_read_b128), the thread's outlier weights of the chunk live in registers;
// f32 accumulation.
template <typename T>
__global__ __launch_bounds__(256) void k_outlier_add(const T *__restrict__ X, int64_t M, int64_t K, int64_t N,
                                                    const int64_t *__restrict__ oidx, int64_t n_out,
                                                    const T *__restrict__ ow, const T *__restrict__ bias,
                                                    T *__restrict__ out, bool vec_ok, bool ow_vec) {
    constexpr int RM = 16, CN = 2, CH = 16;
    __shared__ __attribute__((aligned(16))) float xs[RM][CH];
    const int64_t n0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * CN;
    const int64_t m0 = (int64_t)blockIdx.y * RM;
    float acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int c = 0; c < CN; c++) acc[i][c] = 0.0f;
    for (int64_t j0 = 0; j0 < n_out; j0 += CH) {
        __syncthreads();
        {
            const int i = threadIdx.x / CH, j = threadIdx.x % CH;   // 256 threads = RM x CH values
            const int64_t m = m0 + i;
            xs[i][j] = (m < M && j0 + j < n_out) ? to_f32(X[m * K + oidx[j0 + j]]) : 0.0f;
        }
        __syncthreads();
        float w[CN][CH];
        if constexpr (sizeof(T) == 2) {
            if (ow_vec && n0 + CN <= N && j0 + CH <= n_out) {   // 2 x 16-byte loads per column
#pragma unroll
                for (int c = 0; c < CN; c++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(ow + (n0 + c) * n_out + j0 + 8 * h);
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            w[c][8 * h + 2 * e] = unpack_lo<T>(v[e]);
                            w[c][8 * h + 2 * e + 1] = unpack_hi<T>(v[e]);
                        }
                    }
            } else {
#pragma unroll
                for (int c = 0; c < CN; c++)
#pragma unroll
                    for (int j = 0; j < CH; j++)
                        w[c][j] = (n0 + c < N && j0 + j < n_out) ? to_f32(ow[(n0 + c) * n_out + j0 + j]) : 0.0f;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CN; c++)
#pragma unroll
                for (int j = 0; j < CH; j++)
                    w[c][j] = (n0 + c < N && j0 + j < n_out) ? to_f32(ow[(n0 + c) * n_out + j0 + j]) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < RM; i++)
#pragma unroll
            for (int q = 0; q < CH / 4; q++) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xs[i][4 * q]);   // one broadcast ds_read_b128
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int c = 0; c < CN; c++) acc[i][c] = fmaf(xv[e], w[c][4 * q + e], acc[i][c]);
            }
    }
    if (n0 >= N) return;
    float b[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) b[c] = (bias && n0 + c < N) ? to_f32(bias[n0 + c]) : 0.0f;
    const bool full = vec_ok && n0 + CN <= N;
#pragma unroll
    for (int i = 0; i < RM; i++) {
        const int64_t m = m0 + i;
        if (m >= M) break;
        T *p = out + m * N + n0;
        __attribute__((aligned(8))) T v[CN];
        if (full) {
            if constexpr (sizeof(T) == 2) *reinterpret_cast<uint32_t *>(v) = *reinterpret_cast<const uint32_t *>(p);
            else *reinterpret_cast<u32x2 *>(v) = *reinterpret_cast<const u32x2 *>(p);
        } else {
#pragma unroll
            for (int c = 0; c < CN; c++) v[c] = (n0 + c < N) ? p[c] : from_f32<T>(0.0f);
        }
#pragma unroll
        for (int c = 0; c < CN; c++) {
            float f = to_f32(v[c]);
            if (n_out > 0) f = to_f32(from_f32<T>(f + to_f32(from_f32<T>(acc[i][c]))));
            if (bias) f = to_f32(from_f32<T>(f + b[c]));
            v[c] = from_f32<T>(f);
        }
        if (full) {
            if constexpr (sizeof(T) == 2) *reinterpret_cast<uint32_t *>(p) = *reinterpret_cast<const uint32_t *>(v);
            else *reinterpret_cast<u32x2 *>(p) = *reinterpret_cast<const u32x2 *>(v);
        } else {
#pragma unroll
            for (int c = 0; c < CN; c++)
                if (n0 + c < N) p[c] = v[c];
        }
    }
}

// workspace: [x_q int8 M*K | pad to 256][x_scales f32 M | pad to 256][compact outlier activations M x ldx (16-bit types)]
int64_t outlier_linear_workspace_bytes(int64_t M, int64_t K, int64_t n_out) {
    const int64_t ldx = ((n_out + 15) / 16) * 16;
    return ((M * K + 255) & ~(int64_t)255) + ((4 * M + 255) & ~(int64_t)255) + ((2 * ldx * M + 255) & ~(int64_t)255);
}

template <typename T>
static int launch_outlier_linear(const void *X, int64_t M, int64_t K, const int8_t *W, const float *w_scales, int64_t N,
                                 const int64_t *oidx, int64_t n_out, const void *ow, const void *bias, void *out,
                                 void *workspace, int64_t ws_bytes, int dtype, hipStream_t st) {
    char *ws = static_cast<char *>(workspace);
    int8_t *xq = reinterpret_cast<int8_t *>(ws);
    const int64_t off_s = (M * K + 255) & ~(int64_t)255;
    float *xs = reinterpret_cast<float *>(ws + off_s);
    const int64_t off_x = off_s + ((4 * M + 255) & ~(int64_t)255);
    // the compact outlier activations exist when the workspace has room for all chunks of 16 (a caller that sized it with
    // the two-argument query has room for one chunk: more outliers then take the separate k_outlier_add pass)
    const int64_t ldx = ((n_out + 15) / 16) * 16;
    const bool room = ws_bytes >= outlier_linear_workspace_bytes(M, K, n_out);
    T *xo = (n_out > 0 && room && sizeof(T) == 2) ? reinterpret_cast<T *>(ws + off_x) : nullptr;
    const size_t mask_lds = (size_t)((K + 7) & ~(int64_t)7);
    if (mask_lds > 65536) {
        set_error("outlier_linear: in_features %lld too large for the LDS column mask", (long long)K);
        return MBNB_ERR_ARG;
    }
    const bool vec_ok = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    if constexpr (sizeof(T) == 2) {
        if (vec_ok && K <= 8192)   // the row fits the workgroup's registers: read once
            hipLaunchKernelGGL(k_quantize_rowwise_masked_regs<T>, dim3((unsigned)M), dim3(256), mask_lds, st, static_cast<const T *>(X), K,
                               oidx, n_out, xq, xs, xo, ldx);
        else
            hipLaunchKernelGGL(k_quantize_rowwise_masked<T>, dim3((unsigned)M), dim3(256), mask_lds, st, static_cast<const T *>(X), M, K,
                               oidx, n_out, xq, xs, xo, ldx, vec_ok);
    } else {
        hipLaunchKernelGGL(k_quantize_rowwise_masked<T>, dim3((unsigned)M), dim3(256), mask_lds, st, static_cast<const T *>(X), M, K,
                           oidx, n_out, xq, xs, xo, ldx, vec_ok);
    }
    int rc = check_launch("outlier_linear(quantize)");
    if (rc) return rc;
    // the 256 x 256 kernel folds the outlier columns (one MFMA per tile and chunk of 16 outliers) and the bias into its
    // epilogue; the other int8 kernels leave them to k_outlier_add
    const OutlierEpilogue ep{xo, ldx, oidx, n_out, ow, bias};
    bool fused = false;
    rc = matmul_int8_nt_dispatch(xq, W, xs, w_scales, M, N, K, dtype, out, st, ((n_out == 0 || xo != nullptr) && (n_out > 0 || bias)) ? &ep : nullptr, &fused);
    if (rc) return rc;
    if (!fused && (n_out > 0 || bias)) {
        dim3 grid((unsigned)((N + 511) / 512), (unsigned)((M + 15) / 16));
        const bool vec_out = (N % 2 == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
        const bool ow_vec = (n_out % 8 == 0) && ((reinterpret_cast<uintptr_t>(ow) & 15) == 0);
        hipLaunchKernelGGL(k_outlier_add<T>, grid, dim3(256), 0, st, static_cast<const T *>(X), M, K, N, oidx, n_out,
                           static_cast<const T *>(ow), static_cast<const T *>(bias), static_cast<T *>(out), vec_out, ow_vec);
        rc = check_launch("outlier_linear(outlier add)");
    }
    return rc;
}

int outlier_linear_dispatch(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W, const float *w_scales, int64_t N,
                            const int64_t *oidx, int64_t n_out, const void *ow, const void *bias, void *out, void *workspace,
                            int64_t ws_bytes, hipStream_t st) {
    switch (dtype) {
        case MBNB_F16: return launch_outlier_linear<f16_t>(X, M, K, W, w_scales, N, oidx, n_out, ow, bias, out, workspace, ws_bytes, dtype, st);
        case MBNB_BF16: return launch_outlier_linear<bf16_t>(X, M, K, W, w_scales, N, oidx, n_out, ow, bias, out, workspace, ws_bytes, dtype, st);
        default: return launch_outlier_linear<float>(X, M, K, W, w_scales, N, oidx, n_out, ow, bias, out, workspace, ws_bytes, dtype, st);
    }
}

}  // namespace mbnb
