// gemm256q.hip — launch of k_gemm256q (gemm256q.h), the four-wave cut of the 256 x 256 fused 4-bit GEMM, in its own
// translation unit (the kernel is the longest compile of the library; matmul4_kernels.hip only sees the declaration).
#include <cstdlib>
#include <type_traits>
#include "gemm256q.h"

namespace mbnb {

template <typename T, bool NESTED>
int launch_gemm256q(const T *x, const typename Q4ProducerRT<T, NESTED>::Params &wp, const T *bias, void *out, int out_dtype,
                    int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kq = k_gemm256q<T, NESTED>;
#ifdef MBNB_Q_ABLATE
    if constexpr (std::is_same<T, bf16_t>::value && !NESTED) {
        static const int abl = getenv("MBNB_QABL") ? atoi(getenv("MBNB_QABL")) : 0;
        switch (abl) {
            case 64: kq = k_gemm256q<T, NESTED, 64>; break;
            default: break;
        }
    }
#endif
    constexpr int lds = gemm256q_lds_bytes<NESTED>();
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kq), lds, "matmul_4bit(mfma256q)")) return rc;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kq, dim3((unsigned)tiles), dim3(256), lds, st, x, wp, bias, out, out_dtype, M, N, K);
    return check_launch("matmul_4bit(mfma256q)");
}

#define MBNB_INST(T, NESTED)                                                                                              \
    template int launch_gemm256q<T, NESTED>(const T *, const typename Q4ProducerRT<T, NESTED>::Params &, const T *, void *, \
                                            int, int64_t, int64_t, int64_t, hipStream_t);
MBNB_INST(f16_t, false)
MBNB_INST(f16_t, true)
MBNB_INST(bf16_t, false)
MBNB_INST(bf16_t, true)
#undef MBNB_INST

}  // namespace mbnb
