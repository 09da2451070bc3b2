// gemm256s.hip — launch of k_gemm256s (gemm256s.h): the shipping 256 x 256 fused 4-bit GEMM for blocksize 64 (absmax-by-4,
// byte-table decode, packed weights fetched two k-steps at a time).  Own translation unit: matmul4_kernels.hip only sees
// the declaration, and the four instantiations compile in parallel with it.
#include "gemm256s.h"

namespace mbnb {

constexpr int G256S_VAR = 1;   // RAW2 (gemm256s.h); STAG and ADMA measured slower at 4096^3 (profiles/r02_gemm256_ab.txt)

template <typename T, bool NESTED>
int launch_gemm256s(const T *x, const typename Q4ProducerRT<T, NESTED>::Params &wp, const T *bias, void *out, int out_dtype,
                    int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm256s<T, NESTED, G256S_VAR>;
    constexpr int lds = gemm256s_lds_bytes<NESTED, G256S_VAR>();
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(mfma256)")) return rc;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, x, wp, bias, out, out_dtype, M, N, K);
    return check_launch("matmul_4bit(mfma256)");
}

#define MBNB_INST(T, NESTED)                                                                                              \
    template int launch_gemm256s<T, NESTED>(const T *, const typename Q4ProducerRT<T, NESTED>::Params &, const T *, void *, \
                                            int, int64_t, int64_t, int64_t, hipStream_t);
MBNB_INST(f16_t, false)
MBNB_INST(f16_t, true)
MBNB_INST(bf16_t, false)
MBNB_INST(bf16_t, true)
#undef MBNB_INST

}  // namespace mbnb
