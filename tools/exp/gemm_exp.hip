// gemm_exp.hip — A/B harness for the 256 x 256 fused 4-bit GEMM variants (diagnostic; not part of the product library).
// Builds into tools/exp/libgemm_exp.so; tools/exp/ab_gemm.py loads it, checks every variant bit for bit against the
// shipping kernel and times all of them interleaved in one process (cdna_hip_programming.md rule 24).
#include <cstdarg>
#include <cstdio>
#include "../../mps_bitsandbytes_amd/csrc/gemm256s.h"
#include "gemm256t.h"

namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;

template <int VAR> static int run_s(const bf16_t *x, Q4ProducerRT<bf16_t, false>::Params wp, void *out, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm256s<bf16_t, false, VAR>;
    constexpr int lds = gemm256s_lds_bytes<false, VAR>();
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, x, wp, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
    return (int)hipGetLastError();
}

extern "C" int exp_gemm256(int variant, const void *X, const uint8_t *packed, const float *absmax, void *out, int64_t M, int64_t N,
                           int64_t K, int64_t K_weight, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bf16_t *x = static_cast<const bf16_t *>(X);
    AbsmaxView am{absmax, nullptr, nullptr, 1};
    Q4ProducerRT<bf16_t, false>::Params wp{packed, am, N, K_weight, K_weight / 64, 6, MBNB_NF4, 0, 8, 6};
    if (variant < 0) {
        auto kern = k_gemm256p<bf16_t, false, 0, true, true>;
        constexpr int lds = gemm256p_lds_bytes<false>();
        static bool done = false;
        if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2; done = true; }
        const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, x, wp, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
        return (int)hipGetLastError();
    }
    if (variant == 100) {
        auto kern = k_gemm256t<bf16_t, false>;
        constexpr int lds = gemm256s_lds_bytes<false, 1>();
        static bool done = false;
        if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2; done = true; }
        const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, x, wp, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
        return (int)hipGetLastError();
    }
    switch (variant) {
        case 0: return run_s<0>(x, wp, out, M, N, K, st);
        case 1: return run_s<1>(x, wp, out, M, N, K, st);
        case 2: return run_s<2>(x, wp, out, M, N, K, st);
        case 3: return run_s<3>(x, wp, out, M, N, K, st);
        case 4: return run_s<4>(x, wp, out, M, N, K, st);
        case 5: return run_s<5>(x, wp, out, M, N, K, st);
        case 6: return run_s<6>(x, wp, out, M, N, K, st);
        case 7: return run_s<7>(x, wp, out, M, N, K, st);
        case 9: return run_s<9>(x, wp, out, M, N, K, st);
        case 12: return run_s<12>(x, wp, out, M, N, K, st);
        case 13: return run_s<13>(x, wp, out, M, N, K, st);
        default: return -1;
    }
}
