// i8_exp.hip — A/B harness for the 256 x 256 int8 GEMM kernels (diagnostic).
//   variant 0: k_transpose_i8_64 + k_gemm_i8_256 (round 1)     1: k_gemm_i8_p4<BNN = true> on B [K, N]
//   variant 2: k_gemm_i8_p4<BNN = false> on a pre-transposed B^T      3: k_gemm_i8_256 on a pre-transposed B^T
#include <cstdarg>
#include <cstdio>
#include "../../mps_bitsandbytes_amd/csrc/int8_kernels.hip"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
int64_t matmul4_splitk_slices(int64_t, int64_t, int64_t) { return 1; }
}  // namespace mbnb
using namespace mbnb;
extern "C" int exp_i8(int variant, const int8_t *A, const int8_t *B, const int8_t *Bt, const float *sA, const float *sB, void *out,
                      void *ws, int64_t M, int64_t N, int64_t K, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    OutlierEpilogue epv{nullptr, 0, nullptr, 0, nullptr, nullptr};
    f16_t *o = static_cast<f16_t *>(out);
    if (variant == 0 || variant == 3) {
        const int8_t *bt = Bt;
        if (variant == 0) {
            hipLaunchKernelGGL(k_transpose_i8_64, dim3((unsigned)(N / 64), (unsigned)(K / 64)), dim3(256), 0, st, B, static_cast<int8_t *>(ws), K, N);
            bt = static_cast<int8_t *>(ws);
        }
        auto kern = k_gemm_i8_256<f16_t>;
        ensure_dyn_lds(reinterpret_cast<const void *>(kern), 4 * P_IMG, "");
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), 4 * P_IMG, st, A, bt, sA, sB, o, M, N, K, epv);
    } else if (variant == 1) {
        auto kern = k_gemm_i8_p4<f16_t, true>;
        ensure_dyn_lds(reinterpret_cast<const void *>(kern), 131072, "");
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), 131072, st, A, B, sA, sB, o, M, N, K, epv);
    } else if (variant == 6) {
        auto kern = k_gemm_i8_256<f16_t, true>;
        ensure_dyn_lds(reinterpret_cast<const void *>(kern), 4 * P_IMG, "");
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), 4 * P_IMG, st, A, B, sA, sB, o, M, N, K, epv);
    } else if (variant == 4 || variant == 5) {
        if (variant == 4) {
            auto kern = k_gemm_i8_p4<f16_t, true, 1>;
            ensure_dyn_lds(reinterpret_cast<const void *>(kern), 131072, "");
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), 131072, st, A, B, sA, sB, o, M, N, K, epv);
        } else {
            auto kern = k_gemm_i8_p4<f16_t, true, 3>;
            ensure_dyn_lds(reinterpret_cast<const void *>(kern), 131072, "");
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), 131072, st, A, B, sA, sB, o, M, N, K, epv);
        }
    } else {
        auto kern = k_gemm_i8_p4<f16_t, false>;
        ensure_dyn_lds(reinterpret_cast<const void *>(kern), 131072, "");
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), 131072, st, A, Bt, sA, sB, o, M, N, K, epv);
    }
    return (int)hipGetLastError();
}
