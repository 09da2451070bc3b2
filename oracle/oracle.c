/*
 * oracle.c — CPU restatement of the reference's quantized-linear hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package
 * (mps_bitsandbytes_amd/) links, loads or calls this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline, never as the thing shipped.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit
 * (quantize / pack / dequantize / int8 paths) or to a stated tolerance
 * (matmuls) against outputs of the reference's own Python CPU path, captured
 * in this container by tests/golden/make_golden.py and committed under
 * tests/golden/ (see tests/test_oracle_golden.py).
 *
 * Each function cites the reference file:line it restates
 * (paths relative to the reference tree, mps_bitsandbytes/...).
 *
 * Element dtype codes: 0 = fp16, 1 = bf16, 2 = fp32 (raw little-endian bits).
 * Quant type codes:    0 = nf4,  1 = fp4.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- codebooks: functional.py:21-32 -------------------------------------- */
static const float NF4_CODE[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
static const float FP4_CODE[16] = {
    0.0f, 0.0625f, 0.125f, 0.25f, 0.375f, 0.5f, 0.75f, 1.0f,
    -0.0f, -0.0625f, -0.125f, -0.25f, -0.375f, -0.5f, -0.75f, -1.0f};

static const float *code_table(int quant_type) { return quant_type == 0 ? NF4_CODE : FP4_CODE; }

/* ---- scalar dtype conversion (IEEE, round-to-nearest-even) ----------------- */
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        /* subnormal: value = man * 2^-24 */
        float v = (float)man * 5.9604644775390625e-08f;
        return sign ? -v : v;
    }
    if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
    return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

static inline uint16_t float_to_half(float f) {
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u));
    }
    if (ax >= 0x477FF000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax < 0x38800000u) { /* result is subnormal or zero (|f| < 2^-14) */
        if (ax < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (ties at 2^-25 go to even = 0) */
        uint32_t e = ax >> 23;               /* biased exponent 102..112 */
        uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
        uint32_t shift = 126u - e;           /* 14..24 */
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1u);
        if (rem > halfway || (rem == halfway && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    uint32_t e = (ax >> 23) - 112u;
    uint32_t m = ax & 0x7FFFFFu;
    uint32_t q = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
    return (uint16_t)(sign | q);
}

static inline float bf16_to_float(uint16_t h) { return bits_f32((uint32_t)h << 16); }

static inline uint16_t float_to_bf16(float f) {
    uint32_t x = f32_bits(f);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x40u); /* quiet nan */
    uint32_t lsb = (x >> 16) & 1u;
    x += 0x7FFFu + lsb;
    return (uint16_t)(x >> 16);
}

static inline float load_elem(const void *p, int dtype, int64_t i) {
    if (dtype == 0) return half_to_float(((const uint16_t *)p)[i]);
    if (dtype == 1) return bf16_to_float(((const uint16_t *)p)[i]);
    return ((const float *)p)[i];
}

static inline void store_elem(void *p, int dtype, int64_t i, float v) {
    if (dtype == 0) ((uint16_t *)p)[i] = float_to_half(v);
    else if (dtype == 1) ((uint16_t *)p)[i] = float_to_bf16(v);
    else ((float *)p)[i] = v;
}

/* value after a round trip through `dtype` (what `.to(dtype).float()` gives) */
static inline float round_to(int dtype, float v) {
    if (dtype == 0) return half_to_float(float_to_half(v));
    if (dtype == 1) return bf16_to_float(float_to_bf16(v));
    return v;
}

/* `127.0 / tensor` in the reference is Python-scalar / Tensor, which torch evaluates as
 * tensor.reciprocal() * 127.0 (Tensor.__rtruediv__): TWO f32 roundings, not one division.
 * Measured in the build container: differs from 127.0f/x in ~25 % of inputs by one ulp.
 * Applies to functional.py:518, :621, :852, :859. */
static inline float rscale127(float absmax) { float r = 1.0f / absmax; return r * 127.0f; }

int orc_version(void) { return 1; }

/* ---------------------------------------------------------------------------
 * quantize_4bit — functional.py:163-303.
 * 2-D input [rows, cols]: each row is zero-padded to cols_padded
 * (functional.py:219-224) and cut into blocks of `blocksize`; a non-2-D input
 * is passed as rows = 1, cols = numel (functional.py:257-286: the same
 * algorithm over the flattened tensor).
 *   absmax = max|x| per block in f32, clamp(min=1e-8)        (:232 / :271)
 *   x_norm = x / absmax  (true f32 division)                  (:239 / :276)
 *   idx    = argmin_i |x_norm - code[i]|, first minimum wins  (:242-243)
 *   packed byte j of a row = idx[2j] | idx[2j+1] << 4         (:251 / :286)
 * absmax_in (nullable) is the caller-supplied `absmax=` argument (:231).
 * absmax_out [rows * cols_padded / blocksize] receives the absmax used.
 * ------------------------------------------------------------------------- */
int orc_quantize_4bit(const void *A, int dtype, int64_t rows, int64_t cols, int64_t cols_padded,
                      int blocksize, int quant_type, const float *absmax_in, uint8_t *packed,
                      float *absmax_out) {
    if (blocksize <= 0 || cols_padded % blocksize || cols_padded % 2 || cols_padded < cols) return -1;
    const float *code = code_table(quant_type);
    const int64_t nblk = cols_padded / blocksize;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        uint8_t *idx = (uint8_t *)malloc((size_t)blocksize);
        for (int64_t b = 0; b < nblk; b++) {
            const int64_t k0 = b * blocksize;
            float am;
            if (absmax_in) {
                am = absmax_in[r * nblk + b];
            } else {
                am = 0.0f;
                int has_nan = 0;
                for (int j = 0; j < blocksize; j++) {
                    int64_t k = k0 + j;
                    float v = (k < cols) ? fabsf(load_elem(A, dtype, r * cols + k)) : 0.0f;
                    if (v != v) has_nan = 1;
                    if (v > am) am = v;
                }
                if (am < 1e-8f) am = 1e-8f;
                /* torch's abs().max() propagates NaN (and clamp keeps it): functional.py:231-232.  The distances below
                 * are then all NaN, `d < bestd` is never true, and the index stays 0 -- torch.argmin's answer too. */
                if (has_nan) am = NAN;
            }
            absmax_out[r * nblk + b] = am;
            for (int j = 0; j < blocksize; j++) {
                int64_t k = k0 + j;
                float x = (k < cols) ? load_elem(A, dtype, r * cols + k) : 0.0f;
                float xn = x / am;
                int best = 0;
                float bestd = fabsf(xn - code[0]);
                for (int i = 1; i < 16; i++) {
                    float d = fabsf(xn - code[i]);
                    if (d < bestd) { bestd = d; best = i; }
                }
                idx[j] = (uint8_t)best;
            }
            if (blocksize >= 2) {
                for (int j = 0; j < blocksize; j += 2)
                    packed[(r * cols_padded + k0 + j) / 2] = (uint8_t)(idx[j] | (idx[j + 1] << 4));
            } else {
                /* blocksize 1: neighbouring blocks share a byte */
                int64_t k = k0;
                uint8_t *dst = &packed[(r * cols_padded + k) / 2];
                if ((k & 1) == 0) *dst = (uint8_t)((*dst & 0xF0u) | idx[0]);
                else *dst = (uint8_t)((*dst & 0x0Fu) | (idx[0] << 4));
            }
        }
        free(idx);
    }
    return 0;
}

/* ---------------------------------------------------------------------------
 * dequantize_4bit — functional.py:306-416.
 *   low nibble -> even k, high nibble -> odd k               (:360-366 / :390-396)
 *   value = code[idx] * absmax[row, k / blocksize]  in f32    (:375-376 / :404-405)
 *   sliced to [:, :cols], cast to out dtype (RNE)             (:379-382 / :410)
 * Flat (non-2-D) tensors are passed as rows = 1.
 * ------------------------------------------------------------------------- */
int orc_dequantize_4bit(const uint8_t *packed, const float *absmax, int64_t rows, int64_t cols,
                        int64_t cols_padded, int blocksize, int quant_type, int out_dtype, void *out) {
    if (blocksize <= 0 || cols_padded % blocksize || cols_padded % 2) return -1;
    const float *code = code_table(quant_type);
    const int64_t nblk = cols_padded / blocksize;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        for (int64_t k = 0; k < cols; k++) {
            uint8_t byte = packed[(r * cols_padded + k) / 2];
            int idx = ((r * cols_padded + k) & 1) ? (byte >> 4) : (byte & 0x0F);
            float v = code[idx] * absmax[r * nblk + k / blocksize];
            store_elem(out, out_dtype, r * cols + k, v);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------
 * quantize_blockwise — functional.py:469-539 (the int8 "double quant" of
 * absmax, called from quantize_4bit at :291 with blocksize 256).
 *   flat blocks; absmax per block clamp 1e-8 (:515); scale = 127.0/absmax (:518) == recip(absmax)*127
 *   q = clamp(round_half_even(x * scale), -127, 127) -> int8   (:520)
 * ------------------------------------------------------------------------- */
int orc_quantize_blockwise(const void *A, int dtype, int64_t numel, int blocksize,
                           const float *absmax_in, int8_t *out, float *absmax_out) {
    if (blocksize <= 0) return -1;
    const int64_t nblk = (numel + blocksize - 1) / blocksize;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblk; b++) {
        int64_t i0 = b * blocksize, i1 = i0 + blocksize;
        if (i1 > numel) i1 = numel;
        float am;
        if (absmax_in) am = absmax_in[b];
        else {
            am = 0.0f;
            for (int64_t i = i0; i < i1; i++) {
                float v = fabsf(load_elem(A, dtype, i));
                if (v > am) am = v;
            }
            if (am < 1e-8f) am = 1e-8f;
        }
        absmax_out[b] = am;
        float scale = rscale127(am);
        for (int64_t i = i0; i < i1; i++) {
            float q = rintf(load_elem(A, dtype, i) * scale);
            if (q < -127.0f) q = -127.0f;
            if (q > 127.0f) q = 127.0f;
            out[i] = (int8_t)q;
        }
    }
    return 0;
}

/* dequantize_blockwise — functional.py:542-600: q.float() * (absmax/127.0) -> dtype (:592-594) */
int orc_dequantize_blockwise(const int8_t *q, int64_t numel, const float *absmax, int blocksize,
                             int out_dtype, void *out) {
    if (blocksize <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < numel; i++) {
        float scale = absmax[i / blocksize] / 127.0f;
        store_elem(out, out_dtype, i, (float)q[i] * scale);
    }
    return 0;
}

/* ---------------------------------------------------------------------------
 * dequant_absmax, legacy (non-QuantState) form — functional.py:878-889.
 *   absmax = zeros(rows, num_blocks) f32;  for dqb < dq_blocks:
 *     absmax[:, dqb*bs : min((dqb+1)*bs, num_blocks)] = q[:, ...].float() * scales[:, dqb]
 * Codes past dq_blocks * blocksize keep the zero of zeros_like.  q_kind: 0 int8, 1 uint8, 2 f32
 * (any other code dtype is `.float()`-ed by the caller, which is what the reference does).
 * ------------------------------------------------------------------------- */
int orc_dequant_absmax(const void *q, int q_kind, int64_t rows, int64_t num_blocks, const float *scales,
                       int64_t dq_blocks, int blocksize, float *out) {
    if (blocksize <= 0 || rows < 0 || num_blocks < 0 || dq_blocks < 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++)
        for (int64_t j = 0; j < num_blocks; j++) {
            const int64_t i = r * num_blocks + j, dqb = j / blocksize;
            float v = 0.0f;
            if (dqb < dq_blocks) {
                float c = q_kind == 0 ? (float)((const int8_t *)q)[i] : q_kind == 1 ? (float)((const uint8_t *)q)[i]
                                                                                    : ((const float *)q)[i];
                v = c * scales[r * dq_blocks + dqb];
            }
            out[i] = v;
        }
    return 0;
}

/* ---------------------------------------------------------------------------
 * quantize_rowwise — functional.py:607-625.
 *   scales = max|x| per row (f32) clamp 1e-8 — the absmax itself  (:617-618)
 *   q = clamp(round(x * (127.0/scales)), -127, 127)              (:620-623)
 * ------------------------------------------------------------------------- */
int orc_quantize_rowwise(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out,
                         float *scales) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        float am = 0.0f;
        for (int64_t c = 0; c < cols; c++) {
            float v = fabsf(load_elem(A, dtype, r * cols + c));
            if (v > am) am = v;
        }
        if (am < 1e-8f) am = 1e-8f;
        scales[r] = am;
        float s = rscale127(am);
        for (int64_t c = 0; c < cols; c++) {
            float q = rintf(load_elem(A, dtype, r * cols + c) * s);
            if (q < -127.0f) q = -127.0f;
            if (q > 127.0f) q = 127.0f;
            out[r * cols + c] = (int8_t)q;
        }
    }
    return 0;
}

/* dequantize_rowwise — functional.py:628-636: q.float() * (scales/127.0) -> dtype */
int orc_dequantize_rowwise(const int8_t *q, const float *scales, int64_t rows, int64_t cols,
                           int out_dtype, void *out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        float s = scales[r] / 127.0f;
        for (int64_t c = 0; c < cols; c++)
            store_elem(out, out_dtype, r * cols + c, (float)q[r * cols + c] * s);
    }
    return 0;
}

/* ---------------------------------------------------------------------------
 * double_quant — functional.py:814-863 (LLM.int8 row + column statistics).
 * row_stats/col_stats are computed unless the *_given flag says the caller
 * supplied them (:844-847); out_row by row (:851-854), out_col by column (:858-861).
 * ------------------------------------------------------------------------- */
int orc_double_quant(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out_col,
                     int8_t *out_row, float *col_stats, float *row_stats, int col_given,
                     int row_given) {
    if (!row_given) {
        for (int64_t r = 0; r < rows; r++) {
            float am = 0.0f;
            for (int64_t c = 0; c < cols; c++) {
                float v = fabsf(load_elem(A, dtype, r * cols + c));
                if (v > am) am = v;
            }
            row_stats[r] = am < 1e-8f ? 1e-8f : am;
        }
    }
    if (!col_given) {
        for (int64_t c = 0; c < cols; c++) col_stats[c] = 0.0f;
        for (int64_t r = 0; r < rows; r++)
            for (int64_t c = 0; c < cols; c++) {
                float v = fabsf(load_elem(A, dtype, r * cols + c));
                if (v > col_stats[c]) col_stats[c] = v;
            }
        for (int64_t c = 0; c < cols; c++)
            if (col_stats[c] < 1e-8f) col_stats[c] = 1e-8f;
    }
    for (int64_t r = 0; r < rows; r++) {
        float sr = rscale127(row_stats[r]);
        for (int64_t c = 0; c < cols; c++) {
            float x = load_elem(A, dtype, r * cols + c);
            float q = rintf(x * sr);
            q = q < -127.0f ? -127.0f : (q > 127.0f ? 127.0f : q);
            out_row[r * cols + c] = (int8_t)q;
            float sc = rscale127(col_stats[c]);
            q = rintf(x * sc);
            q = q < -127.0f ? -127.0f : (q > 127.0f ? 127.0f : q);
            out_col[r * cols + c] = (int8_t)q;
        }
    }
    return 0;
}

/* dense helper lives in oracle_gemm.c (compiled with FMA contraction allowed) */
void orc_sgemm_nt(const float *A, const float *W, const float *bias, float *C, int64_t M,
                  int64_t N, int64_t K);
#define sgemm_nt orc_sgemm_nt

/* ---------------------------------------------------------------------------
 * matmul_4bit, CPU ("fallback") branch — functional.py:752-773:
 *   weight = dequantize_4bit(B, quant_state)         -> w_dtype (= quant_state.dtype)  (:756)
 *   A, bias cast to weight dtype                                                      (:764-766)
 *   output = F.linear(A, weight, bias)  in w_dtype (f32 accumulate, one rounding)     (:767)
 *   output.to(compute_dtype)                                                          (:773)
 * Double-quantised absmax (state2) is decoded first by the caller with
 * orc_dequantize_blockwise (functional.py:336-337).
 * K = activation width = quant_state.shape[1]; K_weight = padded row length.
 * ------------------------------------------------------------------------- */
int orc_matmul_4bit(const void *A, int a_dtype, int64_t M, int64_t K, const uint8_t *packed,
                    const float *absmax, int64_t N, int64_t K_weight, int blocksize, int quant_type,
                    int w_dtype, const void *bias, int bias_dtype, int out_dtype, void *out) {
    if (blocksize <= 0 || K_weight % blocksize || K_weight < K) return -1;
    const float *code = code_table(quant_type);
    const int64_t nblk = K_weight / blocksize;
    float *Wf = (float *)malloc(sizeof(float) * (size_t)(N * K));
    float *Af = (float *)malloc(sizeof(float) * (size_t)(M * K));
    float *Cf = (float *)malloc(sizeof(float) * (size_t)(M * N));
    float *bf = bias ? (float *)malloc(sizeof(float) * (size_t)N) : NULL;
    if (!Wf || !Af || !Cf || (bias && !bf)) { free(Wf); free(Af); free(Cf); free(bf); return -2; }
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++)
        for (int64_t k = 0; k < K; k++) {
            uint8_t byte = packed[(n * K_weight + k) / 2];
            int idx = (k & 1) ? (byte >> 4) : (byte & 0x0F);
            Wf[n * K + k] = round_to(w_dtype, code[idx] * absmax[n * nblk + k / blocksize]);
        }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * K; i++) Af[i] = round_to(w_dtype, load_elem(A, a_dtype, i));
    if (bias)
        for (int64_t n = 0; n < N; n++) bf[n] = round_to(w_dtype, load_elem(bias, bias_dtype, n));
    sgemm_nt(Af, Wf, bf, Cf, M, N, K);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * N; i++) store_elem(out, out_dtype, i, round_to(w_dtype, Cf[i]));
    free(Wf); free(Af); free(Cf); free(bf);
    return 0;
}

/* ---------------------------------------------------------------------------
 * matmul_int8 — functional.py:788-793:
 *   A_dequant = dequantize_rowwise(A[M,K], A_scales, dtype)
 *   B_dequant = dequantize_rowwise(B.T, B_scales, dtype).T      (B is [K,N]; scale per column)
 *   torch.matmul(A_dequant, B_dequant)  in `dtype` (f32 accumulate, one rounding)
 * ------------------------------------------------------------------------- */
int orc_matmul_int8(const int8_t *A, const int8_t *B, const float *A_scales, const float *B_scales,
                    int64_t M, int64_t N, int64_t K, int out_dtype, void *out) {
    float *Af = (float *)malloc(sizeof(float) * (size_t)(M * K));
    float *Wf = (float *)malloc(sizeof(float) * (size_t)(N * K));
    float *Cf = (float *)malloc(sizeof(float) * (size_t)(M * N));
    if (!Af || !Wf || !Cf) { free(Af); free(Wf); free(Cf); return -2; }
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; m++) {
        float s = A_scales[m] / 127.0f;
        for (int64_t k = 0; k < K; k++) Af[m * K + k] = round_to(out_dtype, (float)A[m * K + k] * s);
    }
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        float s = B_scales[n] / 127.0f;
        for (int64_t k = 0; k < K; k++) Wf[n * K + k] = round_to(out_dtype, (float)B[k * N + n] * s);
    }
    sgemm_nt(Af, Wf, NULL, Cf, M, N, K);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * N; i++) store_elem(out, out_dtype, i, Cf[i]);
    free(Af); free(Wf); free(Cf);
    return 0;
}

/* ---------------------------------------------------------------------------
 * Linear8bit.forward — nn/linear8bit.py:70-102:
 *   W = dequantize_rowwise(weight_int8[N,K], weight_scales[N], compute_dtype)
 *   F.linear(x, W, bias)   (x and bias already in compute_dtype)
 * ------------------------------------------------------------------------- */
int orc_linear_int8(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W,
                    const float *W_scales, int64_t N, const void *bias, void *out) {
    float *Af = (float *)malloc(sizeof(float) * (size_t)(M * K));
    float *Wf = (float *)malloc(sizeof(float) * (size_t)(N * K));
    float *Cf = (float *)malloc(sizeof(float) * (size_t)(M * N));
    float *bf = bias ? (float *)malloc(sizeof(float) * (size_t)N) : NULL;
    if (!Af || !Wf || !Cf || (bias && !bf)) { free(Af); free(Wf); free(Cf); free(bf); return -2; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * K; i++) Af[i] = load_elem(X, dtype, i);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        float s = W_scales[n] / 127.0f;
        for (int64_t k = 0; k < K; k++) Wf[n * K + k] = round_to(dtype, (float)W[n * K + k] * s);
    }
    if (bias)
        for (int64_t n = 0; n < N; n++) bf[n] = load_elem(bias, dtype, n);
    sgemm_nt(Af, Wf, bf, Cf, M, N, K);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * N; i++) store_elem(out, dtype, i, Cf[i]);
    free(Af); free(Wf); free(Cf); free(bf);
    return 0;
}

/* ---------------------------------------------------------------------------
 * Embedding4bit.forward — nn/embedding.py:83-138 (the Python path):
 *   row r = input[t]; QuantState(absmax[r], shape [dim], blocksize, dtype) -> dequantize_nf4/fp4(packed[r])
 *   = code[nibble] * absmax[r, k / blocksize] in f32 -> .to(dtype)   (functional.py:388-416, low nibble = even k)
 *   rows equal to padding_idx are masked to 0.0 (:133-136).  has_padding = 0: no masking.
 * weight_packed [num, dim/2] u8, weight_absmax [num, ceil(dim/blocksize)] f32 (:70-77).
 * ------------------------------------------------------------------------- */
int orc_embedding_4bit(const int64_t *idx, int64_t n_idx, const uint8_t *packed, const float *absmax,
                       int64_t num_embeddings, int64_t dim, int blocksize, int quant_type,
                       int has_padding, int64_t padding_idx, int out_dtype, void *out) {
    const float *code = code_table(quant_type);
    const int64_t nblk = (dim + blocksize - 1) / blocksize;
    for (int64_t t = 0; t < n_idx; t++)
        if (idx[t] < 0 || idx[t] >= num_embeddings) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n_idx; t++) {
        const int64_t r = idx[t];
        const int pad = has_padding && r == padding_idx;
        for (int64_t k = 0; k < dim; k++) {
            const uint8_t b = packed[r * (dim / 2) + (k >> 1)];
            const int nib = (k & 1) ? (b >> 4) : (b & 15);
            const float v = code[nib] * absmax[r * nblk + k / blocksize];
            store_elem(out, out_dtype, t * dim + k, pad ? 0.0f : v);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------
 * Embedding8bit.forward — nn/embedding.py:255-268 (the Python path), arithmetic in `dtype`:
 *   weight_int8[input].to(dtype) * (scales[input].unsqueeze(-1) / 127.0).to(dtype)
 *   i.e. the f32 quotient is rounded to dtype first, then one dtype multiply (computed in f32, rounded once).
 * ------------------------------------------------------------------------- */
int orc_embedding_8bit(const int64_t *idx, int64_t n_idx, const int8_t *W, const float *scales,
                       int64_t num_embeddings, int64_t dim, int has_padding, int64_t padding_idx,
                       int out_dtype, void *out) {
    for (int64_t t = 0; t < n_idx; t++)
        if (idx[t] < 0 || idx[t] >= num_embeddings) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n_idx; t++) {
        const int64_t r = idx[t];
        const int pad = has_padding && r == padding_idx;
        const float s = round_to(out_dtype, scales[r] / 127.0f);
        for (int64_t k = 0; k < dim; k++)
            store_elem(out, out_dtype, t * dim + k, pad ? 0.0f : (float)W[r * dim + k] * s);
    }
    return 0;
}

/* ---------------------------------------------------------------------------
 * OutlierAwareLinear.forward — nn/outlier_aware.py:84-146, arithmetic in `dtype` (compute_dtype):
 *   non-outlier columns of x: quantize_rowwise (functional.py:607-625) over those columns only  (:127-131)
 *   x_fp = q.to(dtype) * (x_scales/127).to(dtype);  w_fp = W.to(dtype) * (w_scales/127).to(dtype) (:135-136)
 *   output_main = mm(x_fp, w_fp.t())  (f32 accumulate, one rounding to dtype)                    (:138)
 *   output_outlier = mm(x[:, outlier_idx].to(dtype), outlier_weights.t())                        (:141)
 *   output = output_main + output_outlier  (dtype add; absent when there are no outliers, :100-105)
 *   output = output + bias                 (dtype add, :110-111)
 * W [N,K] int8; outlier_w [N, n_out] in dtype; x, bias in dtype.
 * ------------------------------------------------------------------------- */
int orc_outlier_linear(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W,
                       const float *w_scales, int64_t N, const int64_t *outlier_idx, int64_t n_out,
                       const void *outlier_w, const void *bias, void *out) {
    uint8_t *is_out = (uint8_t *)calloc((size_t)K, 1);
    float *Af = (float *)malloc(sizeof(float) * (size_t)(M * K));
    float *Wf = (float *)malloc(sizeof(float) * (size_t)(N * K));
    float *Cf = (float *)malloc(sizeof(float) * (size_t)(M * N));
    if (!is_out || !Af || !Wf || !Cf) { free(is_out); free(Af); free(Wf); free(Cf); return -2; }
    for (int64_t j = 0; j < n_out; j++) {
        if (outlier_idx[j] < 0 || outlier_idx[j] >= K) { free(is_out); free(Af); free(Wf); free(Cf); return -1; }
        is_out[outlier_idx[j]] = 1;
    }
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; m++) {
        float am = 0.0f;
        for (int64_t k = 0; k < K; k++) {
            if (is_out[k]) continue;
            float v = fabsf(load_elem(X, dtype, m * K + k));
            if (v > am) am = v;
        }
        if (am < 1e-8f) am = 1e-8f;
        const float r = rscale127(am);
        const float s = round_to(dtype, am / 127.0f);
        for (int64_t k = 0; k < K; k++) {
            if (is_out[k]) { Af[m * K + k] = 0.0f; continue; }   /* column not part of x_main */
            float q = rintf(load_elem(X, dtype, m * K + k) * r);
            if (q < -127.0f) q = -127.0f;
            if (q > 127.0f) q = 127.0f;
            Af[m * K + k] = round_to(dtype, q * s);
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        const float s = round_to(dtype, w_scales[n] / 127.0f);
        for (int64_t k = 0; k < K; k++)
            Wf[n * K + k] = is_out[k] ? 0.0f : round_to(dtype, (float)W[n * K + k] * s);
    }
    sgemm_nt(Af, Wf, NULL, Cf, M, N, K);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; m++)
        for (int64_t n = 0; n < N; n++) {
            float v = round_to(dtype, Cf[m * N + n]);
            if (n_out > 0) {
                float o = 0.0f;
                for (int64_t j = 0; j < n_out; j++)
                    o += load_elem(X, dtype, m * K + outlier_idx[j]) * load_elem(outlier_w, dtype, n * n_out + j);
                v = round_to(dtype, v + round_to(dtype, o));
            }
            if (bias) v = round_to(dtype, v + load_elem(bias, dtype, n));
            store_elem(out, dtype, m * N + n, v);
        }
    free(is_out); free(Af); free(Wf); free(Cf);
    return 0;
}

/* ---------------------------------------------------------------------------
 * FP8 E4M3 — functional.py:643-673 (entry points), :1086-1163 (encode), :1166-1215 (decode), :796-807 (matmul).
 *
 * The reference's encoder is NOT the OCP conversion; it is restated literally:
 *   scales = clamp(max|row| / 448, 1e-12)          normalized = clamp(x / scale, -448, 448)
 *   exp = floor(log2(|v|))  (torch.log2 in f32)     biased = exp + 7
 *   mant = trunc(clamp((|v| / 2^exp - 1) * 8 + 0.5, 0, 7))     (no carry into the exponent)
 *   biased <= 0 -> sign only (subnormals flushed)   biased >= 15 -> sign | 0x77 (so every |v| >= 256 becomes 240)
 *   zero -> sign, NaN -> 0x7F
 * torch.log2 is correctly rounded around the integers that matter here, which makes floor(log2(v)) jump to k already
 * a few ulps BELOW 2^k (log2(2^k - n ulp) rounds to k): n_k = 2 for k in [-7,-4], 1 for -3,-2, 0 for -1..2, 1 for 3,4,
 * 2 for 5..8, 5 for 9 (measured on torch 2.10 CPU by tests/golden/make_golden_fp8.py, which also pins this function).
 * fp8_exponent() applies that rule on the bits so the restatement does not depend on a libm.
 * ------------------------------------------------------------------------- */
static inline int fp8_round_up_ulps(int k) {   /* how many f32 values below 2^k already report exponent k */
    if (k >= 9) return 5;
    if (k >= 5) return 2;
    if (k >= 3) return 1;
    if (k >= -1) return 0;
    if (k >= -3) return 1;
    if (k >= -7) return 2;
    if (k >= -15) return 5;
    return 11;
}
static inline int fp8_exponent(float a) {      /* floor(torch.log2(a)) for finite a > 0 */
    uint32_t u = f32_bits(a);
    int e = (int)((u >> 23) & 0xFF) - 127;
    uint32_t mant = u & 0x7FFFFFu;
    if (((u >> 23) & 0xFF) == 0) {             /* f32 subnormal: far below the FP8 range, exponent only needs to be <= -8 */
        return -127;
    }
    if ((0x7FFFFFu - mant) < (uint32_t)fp8_round_up_ulps(e + 1)) e += 1;
    return e;
}
static inline uint8_t float_to_fp8_e4m3_ref(float v) {
    if (v != v) return 0x7F;
    uint8_t sign = v < 0.0f ? 0x80 : 0x00;
    float a = fabsf(v);
    if (a > 448.0f) a = 448.0f;
    if (a == 0.0f) return sign;
    int e = fp8_exponent(a);
    int biased = e + 7;
    if (biased >= 15) return (uint8_t)(sign | 0x77);
    if (biased <= 0) return sign;
    float m = a / ldexpf(1.0f, e) - 1.0f;
    float mb = m * 8.0f + 0.5f;
    if (mb < 0.0f) mb = 0.0f;
    if (mb > 7.0f) mb = 7.0f;
    return (uint8_t)(sign | (biased << 3) | (uint8_t)mb);
}
static inline float fp8_e4m3_to_float_ref(uint8_t b) {
    int sign = b >> 7, e = (b >> 3) & 0xF, m = b & 7;
    float r;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = ((float)m / 8.0f) * 0.015625f;                   /* (mant/8) * 2^-6 */
    else r = (1.0f + (float)m / 8.0f) * ldexpf(1.0f, e - 7);
    return sign ? -r : r;
}

int orc_quantize_fp8_e4m3(const void *A, int dtype, int64_t rows, int64_t cols, uint8_t *out, float *scales) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        float am = 0.0f;
        for (int64_t c = 0; c < cols; c++) {
            float v = fabsf(load_elem(A, dtype, r * cols + c));
            if (v > am || v != v) am = v;                                 /* torch.max propagates NaN */
        }
        float s = am / 448.0f;
        if (s < 1e-12f) s = 1e-12f;
        scales[r] = s;
        for (int64_t c = 0; c < cols; c++) {
            float n = load_elem(A, dtype, r * cols + c) / s;
            if (n < -448.0f) n = -448.0f;
            if (n > 448.0f) n = 448.0f;
            out[r * cols + c] = float_to_fp8_e4m3_ref(n);
        }
    }
    return 0;
}

int orc_dequantize_fp8_e4m3(const uint8_t *q, const float *scales, int64_t rows, int64_t cols, int out_dtype, void *out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++)
        for (int64_t c = 0; c < cols; c++)
            store_elem(out, out_dtype, r * cols + c, fp8_e4m3_to_float_ref(q[r * cols + c]) * scales[r]);
    return 0;
}

/* matmul_fp8_e4m3 / LinearFP8.forward — functional.py:796-807, nn/linear_fp8.py:74-103:
 *   W = dequantize_fp8_e4m3(weight[N,K], scales[N], dtype);  F.linear(input.to(dtype), W, bias) */
int orc_linear_fp8(const void *X, int dtype, int64_t M, int64_t K, const uint8_t *W, const float *W_scales, int64_t N,
                   const void *bias, void *out) {
    float *Af = (float *)malloc(sizeof(float) * (size_t)(M * K));
    float *Wf = (float *)malloc(sizeof(float) * (size_t)(N * K));
    float *Cf = (float *)malloc(sizeof(float) * (size_t)(M * N));
    float *bf = bias ? (float *)malloc(sizeof(float) * (size_t)N) : NULL;
    if (!Af || !Wf || !Cf || (bias && !bf)) { free(Af); free(Wf); free(Cf); free(bf); return -2; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * K; i++) Af[i] = load_elem(X, dtype, i);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++)
        for (int64_t k = 0; k < K; k++)
            Wf[n * K + k] = round_to(dtype, fp8_e4m3_to_float_ref(W[n * K + k]) * W_scales[n]);
    if (bias)
        for (int64_t n = 0; n < N; n++) bf[n] = load_elem(bias, dtype, n);
    sgemm_nt(Af, Wf, bf, Cf, M, N, K);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * N; i++) store_elem(out, dtype, i, Cf[i]);
    free(Af); free(Wf); free(Cf); free(bf);
    return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
