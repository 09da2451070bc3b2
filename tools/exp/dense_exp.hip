// dense_exp.hip — A/B harness for compile-time variants of k_gemm_dense (diagnostic): built twice (-DGD_M0_GROUP=0 / 1) into
// libdense_exp0.so / libdense_exp1.so, each exporting exp_dense(X, Wd, out, M, N, K, stream): bf16, 256 x 256 tiles, unsplit.
#include "../../mps_bitsandbytes_amd/csrc/gemm_dense.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
extern "C" int exp_dense(const void *X, const void *Wd, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    auto kern = k_gemm_dense<bf16_t, false, 8>;
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GD_LDS) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd),
                       static_cast<const bf16_t *>(nullptr), out, (int)MBNB_BF16, static_cast<float *>(nullptr), M, N, K, K, K, static_cast<const float *>(nullptr),
                       static_cast<const float *>(nullptr), OutlierEpilogue{});
    return (int)hipGetLastError();
}
#if GD_STAMPS
extern "C" int exp_dense_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gd_stamps), sizeof(unsigned long long) * 16); }
#endif
