// d128_stamps.hip — the waits of k_gemm_dense128's k-step timed in place (G128_STAMPS = 1: the vmcnt wait, 2: the barrier), as dense_exp.hip does for
// k_gemm_dense.  exp_d128(X, Wd, out, M, N, K, stream); exp_d128_stamps(host[16]).
#include "../../mps_bitsandbytes_amd/csrc/gemm_dense128.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
extern "C" int exp_d128(const void *X, const void *Wd, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    auto kern = k_gemm_dense128<bf16_t>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS) != hipSuccess) return -2;
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), G128_LDS, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd),
                       static_cast<const bf16_t *>(nullptr), out, (int)MBNB_BF16, M, N, K, K);
    return (int)hipGetLastError();
}
extern "C" int exp_d128_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_d128_stamps), sizeof(unsigned long long) * 16); }
