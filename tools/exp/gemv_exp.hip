// gemv_exp.hip — A/B harness for the M = 1 GEMV decode flavours (gemv4.h DEC = 0 / 1 / 2), bf16, NF4, plain absmax.
#include <cstdarg>
#include <cstdio>
#include "../../mps_bitsandbytes_amd/csrc/gemv4_lean.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
template <int DEC, int NR> static int run(const bf16_t *x, const uint8_t *packed, AbsmaxView am, bf16_t *o, int64_t N, int64_t K, hipStream_t st) {
    const int64_t Kp = (K + 2047) & ~(int64_t)2047;
    dim3 grid((unsigned)((N + 4 * NR - 1) / (4 * NR)), 1);
    hipLaunchKernelGGL((k_gemv4<bf16_t, bf16_t, MBNB_NF4, false, 1, NR, 2, true, DEC>), grid, dim3(256), (size_t)Kp * 2, st, x, packed, am,
                       (const bf16_t *)nullptr, o, (int64_t)1, N, K, K, 6);
    return (int)hipGetLastError();
}
extern "C" int exp_gemv(int dec, int nr, const void *X, const uint8_t *packed, const float *absmax, void *out, int64_t N, int64_t K, void *stream) {
    AbsmaxView am{absmax, nullptr, nullptr, 1};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bf16_t *x = static_cast<const bf16_t *>(X);
    bf16_t *o = static_cast<bf16_t *>(out);
    if (dec == 10 || dec >= 100) {    // k_gemv4_lean (K = 4096); dec >= 100: with dec KiB of dynamic LDS per workgroup (fewer resident workgroups per CU)
        const size_t lds = dec >= 100 ? (size_t)(dec - 100) * 1024 : (size_t)K * 2;
        auto kern = k_gemv4_lean<bf16_t, bf16_t, MBNB_NF4, false, 2>;
        if (lds > 65536) {
            static size_t raised = 0;
            if (lds > raised) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3; raised = lds; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)((N + 3) / 4)), dim3(256), lds, st, x, packed, am, (const bf16_t *)nullptr, o, N, K);
        return (int)hipGetLastError();
    }
    if (nr == 1) { if (dec == 0) return run<0, 1>(x, packed, am, o, N, K, st); if (dec == 1) return run<1, 1>(x, packed, am, o, N, K, st); return run<2, 1>(x, packed, am, o, N, K, st); }
    if (dec == 0) return run<0, 2>(x, packed, am, o, N, K, st); if (dec == 1) return run<1, 2>(x, packed, am, o, N, K, st); return run<2, 2>(x, packed, am, o, N, K, st);
}
