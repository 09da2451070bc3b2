/*
 * oracle_gemm.c — dense f32 GEMM helper for the CPU oracle (TEST INFRASTRUCTURE ONLY,
 * see oracle.c).  Kept in its own translation unit so it can be compiled with
 * -ffp-contract=fast (FMA) while the bit-exact quantize/dequantize restatements in
 * oracle.c are compiled with -ffp-contract=off.
 */
#include <stdint.h>
#include <stddef.h>

/* ---------------------------------------------------------------------------
 * Dense helper: C[M,N] = A[M,K] * W[N,K]^T (+ bias), f32 operands, f32
 * accumulation.  Cache-blocked, OpenMP over row blocks; the inner kernel is
 * written so gcc -O3 vectorises it.  Used by both matmul restatements.
 * ------------------------------------------------------------------------- */
void orc_sgemm_nt(const float *A, const float *W, const float *bias, float *C, int64_t M,
                     int64_t N, int64_t K) {
    const int64_t BM = 32, BN = 64;
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int64_t m0 = 0; m0 < M; m0 += BM) {
        for (int64_t n0 = 0; n0 < N; n0 += BN) {
            int64_t m1 = m0 + BM < M ? m0 + BM : M;
            int64_t n1 = n0 + BN < N ? n0 + BN : N;
            for (int64_t m = m0; m < m1; m++) {
                const float *a = A + m * K;
                for (int64_t n = n0; n < n1; n++) {
                    const float *w = W + n * K;
                    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    int64_t k = 0;
                    for (; k + 8 <= K; k += 8)
                        for (int j = 0; j < 8; j++) acc[j] += a[k + j] * w[k + j];
                    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
                    for (; k < K; k++) s += a[k] * w[k];
                    C[m * N + n] = s + (bias ? bias[n] : 0.0f);
                }
            }
        }
    }
}

