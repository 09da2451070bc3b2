// dq_exp.hip — ablation harness for k_gemm_dq (diagnostic; not part of the product): libdq_exp.so exports exp_dq(abl, ...).
// The scratch must already hold the dequantised weight (the ablated variants skip part of the producer work).
#include <cstdarg>
#include <cstdio>
#include "parked/gemm_dq.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
template <int ABL> static int run(const bf16_t *x, const uint8_t *packed, const float *am, bf16_t *wd, uint32_t *sync, void *out, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm_dq<bf16_t, ABL>;
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GQ_LDS) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GQ_LDS, st, x, packed, am, (int)MBNB_NF4, wd, sync, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K);
    return (int)hipGetLastError();
}
#ifndef DQ_VARIANTS
#define DQ_VARIANTS X(0) X(1) X(2) X(4) X(24) X(32) X(62)
#endif
extern "C" int exp_dq(int abl, const void *X_, const uint8_t *packed, const float *absmax, void *wd, void *sync, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (abl) {
#define X(v) case v: return run<v>(static_cast<const bf16_t *>(X_), packed, absmax, static_cast<bf16_t *>(wd), static_cast<uint32_t *>(sync), out, M, N, K, st);
        DQ_VARIANTS
#undef X
        default: return -1;
    }
}
