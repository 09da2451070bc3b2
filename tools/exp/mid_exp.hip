// mid_exp.hip — harness for k_gemm_mid (diagnostic): bf16, plain absmax, slices forced by the caller.
#include <cstdarg>
#include <cstdio>
#include <type_traits>
#include "../../mps_bitsandbytes_amd/csrc/gemm_mid.hip"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
extern "C" int exp_mid_abl(int abl, int slices, const void *X, const uint8_t *packed, const float *absmax, void *out, void *ws, int64_t ws_bytes,
                           int64_t M, int64_t N, int64_t K, int64_t K_weight, void *stream) {
    AbsmaxView am{absmax, nullptr, nullptr, 1};
    Q4ProducerRT<bf16_t, false>::Params wp{packed, am, N, K_weight, K_weight / 64, 6, MBNB_NF4, 0, 8, 6};
#define ARGS static_cast<const bf16_t *>(X), wp, nullptr, static_cast<bf16_t *>(out), M, N, K, static_cast<float *>(ws), ws_bytes, slices, static_cast<hipStream_t>(stream)
    switch (abl) {
        case 1: return launch_gemm_mid<bf16_t, bf16_t, false, 1>(ARGS);
        case 2: return launch_gemm_mid<bf16_t, bf16_t, false, 2>(ARGS);
        case 3: return launch_gemm_mid<bf16_t, bf16_t, false, 3>(ARGS);
        case 4: return launch_gemm_mid<bf16_t, bf16_t, false, 4>(ARGS);
        case 6: return launch_gemm_mid<bf16_t, bf16_t, false, 6>(ARGS);
        case 7: return launch_gemm_mid<bf16_t, bf16_t, false, 7>(ARGS);
        case 9: return launch_gemm_mid<bf16_t, bf16_t, false, 9>(ARGS);
        case 15: return launch_gemm_mid<bf16_t, bf16_t, false, 15>(ARGS);
        default: return launch_gemm_mid<bf16_t, bf16_t, false, 0>(ARGS);
    }
#undef ARGS
}
extern "C" int exp_mid(int slices, const void *X, const uint8_t *packed, const float *absmax, void *out, void *ws, int64_t ws_bytes,
                       int64_t M, int64_t N, int64_t K, int64_t K_weight, void *stream) {
    AbsmaxView am{absmax, nullptr, nullptr, 1};
    Q4ProducerRT<bf16_t, false>::Params wp{packed, am, N, K_weight, K_weight / 64, 6, MBNB_NF4, 0, 8, 6};
    return launch_gemm_mid<bf16_t, bf16_t, false>(static_cast<const bf16_t *>(X), wp, nullptr, static_cast<bf16_t *>(out), M, N, K,
                                                  static_cast<float *>(ws), ws_bytes, slices, static_cast<hipStream_t>(stream));
}
