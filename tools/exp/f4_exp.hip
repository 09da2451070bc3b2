// f4_exp.hip — ablation harness for k_gemm_fused4 (diagnostic; not part of the product library): libf4_exp.so exports
// exp_f4(abl, ...) for the ablation variants of gemm_fused4.h; tools/exp/abl_fused4.py times them interleaved.
#include <cstdarg>
#include <cstdio>
#include "../../mps_bitsandbytes_amd/csrc/gemm_fused4.h"

namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;

template <int ABL> static int run(const bf16_t *x, Q4ProducerRT<bf16_t, false>::Params wp, void *out, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm_fused4<bf16_t, false, ABL>;
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GF_LDS) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GF_LDS, st, x, wp, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
    return (int)hipGetLastError();
}

#ifndef F4_VARIANTS
#define F4_VARIANTS X(0) X(1) X(2) X(4) X(7) X(15) X(31) X(63)
#endif

extern "C" int exp_f4(int abl, const void *X_, const uint8_t *packed, const float *absmax, void *out, int64_t M, int64_t N, int64_t K,
                      int64_t K_weight, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bf16_t *x = static_cast<const bf16_t *>(X_);
    AbsmaxView am{absmax, nullptr, nullptr, 1};
    Q4ProducerRT<bf16_t, false>::Params wp{packed, am, N, K_weight, K_weight / 64, 6, MBNB_NF4, 0, 8, 6};
    switch (abl) {
#define X(v) case v: return run<v>(x, wp, out, M, N, K, st);
        F4_VARIANTS
#undef X
        default: return -1;
    }
}
