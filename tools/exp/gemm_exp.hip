// gemm_exp.hip — A/B harness for the 256 x 256 fused 4-bit GEMM variants (diagnostic; not part of the product library).
// Builds into tools/exp/libgemm_exp.so; tools/exp/ab_gemm.py loads it, checks every variant bit for bit against the
// shipping kernel and times all of them interleaved in one process (cdna_hip_programming.md rule 24).
#include <cstdarg>
#include <cstdio>
#include "parked/gemm256s.h"
#include "gemm256t.h"
#include "gemm256d.h"
#include "gemm256v.h"
#include "gemm256x.h"

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

template <int VAR> static int run_d(const bf16_t *x, const bf16_t *wd, void *out, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm256d<bf16_t, VAR>;
    constexpr int lds = gemm256d_lds_bytes();
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, x, wd, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
    return (int)hipGetLastError();
}

template <int VAR> static int run_v(const bf16_t *x, const bf16_t *wd, void *out, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm256v<bf16_t, VAR>;
    constexpr int lds = gemm256v_lds_bytes();
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, x, wd, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
    return (int)hipGetLastError();
}

template <int VAR> static int run_x(const bf16_t *x, const bf16_t *wd, void *out, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm256x<bf16_t, VAR>;
    constexpr int lds = gemm256v_lds_bytes();
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, x, wd, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
    return (int)hipGetLastError();
}

// dense variants: Wd = the dequantised weight [N, K] bf16
extern "C" int exp_gemm256d(int variant, const void *X, const void *Wd, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (variant) {
        case 0: return run_d<0>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 1: return run_d<1>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 3: return run_d<3>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 5: return run_d<5>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 7: return run_d<7>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 9: return run_d<9>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 15: return run_d<15>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 17: return run_d<17>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 100: return run_v<0>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 112: return run_v<12>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 120: return run_v<20>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 136: return run_v<36>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 160: return run_v<60>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 164: return run_v<64>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 165: return run_v<65>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 166: return run_v<66>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 232: return run_v<132>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 360: return run_v<260>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 616: return run_v<516>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 612: return run_v<512>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 1000: return run_x<0>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 1040: return run_x<40>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 1072: return run_x<72>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 1008: return run_x<8>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 1024: return run_x<24>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 1002: return run_x<2>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 101: return run_v<1>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 103: return run_v<3>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 104: return run_v<4>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 32: return run_d<32>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 33: return run_d<33>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        case 31: return run_d<31>(static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd), out, M, N, K, st);
        default: return -1;
    }
}
