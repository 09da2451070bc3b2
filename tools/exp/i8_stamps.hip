// i8_stamps.hip — cycles of k_gemm_i8_inplace's k-loop (GI8_STAMPS build): exp_i8(A, B, sA, sB, out, M, N, K, stream); exp_i8_stamps(host[8]).
#define GI8_STAMPS 1
#include "../../mps_bitsandbytes_amd/csrc/gemm_i8_inplace.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
extern "C" int exp_i8(const int8_t *A, const int8_t *B, const float *sA, const float *sB, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    auto kern = k_gemm_i8_inplace<bf16_t>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GD_LDS) != hipSuccess) return -2;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, static_cast<hipStream_t>(stream), A, B, sA, sB, static_cast<bf16_t *>(out), M, N, K);
    return (int)hipGetLastError();
}
extern "C" int exp_i8_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gi8_stamps), sizeof(unsigned long long) * (8 + 4 * 256)); }
