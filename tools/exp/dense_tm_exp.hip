// dense_tm_exp.hip — VERDICT r3 item 3c: k_gemm_dense on a TILE-MAJOR private scratch (every LDS-DMA piece of the weight operand 1 KiB contiguous) against the
// row-major scratch dequantize_4bit writes.  Built twice (-DGD_B_TILE_MAJOR=0 / 1) into libdense_tm0.so / libdense_tm1.so; both export exp_dense (bf16, uniform
// 256 x 256 tiles) and exp_relayout (row-major [N, K] -> tile-major [ceil(N / 256)][K / 64][256][64], rows past N zero).  Diagnostic, not product.
#include "../../mps_bitsandbytes_amd/csrc/gemm_dense.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
__global__ void k_relayout(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int64_t N, int64_t K) {
    // one thread per 16-byte chunk (8 k) of the tile-major scratch
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x, kt_n = K >> 6, tiles_n = (N + 255) >> 8;
    if (c >= tiles_n * kt_n * 2048) return;
    const int64_t blk = c >> 11, in = c & 2047, row = in >> 3, ch = in & 7, tn = blk / kt_n, kt = blk % kt_n, n = tn * 256 + row;
    dst[c] = n < N ? src[(n * K + kt * 64 + ch * 8) >> 3] : uint4{0u, 0u, 0u, 0u};
}
extern "C" int exp_relayout(const void *src, void *dst, int64_t N, int64_t K, void *stream) {
    const int64_t chunks = ((N + 255) >> 8) * (K >> 6) * 2048;
    hipLaunchKernelGGL(k_relayout, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), N, K);
    return (int)hipGetLastError();
}
extern "C" int exp_tile_major(void) { return GD_B_TILE_MAJOR; }
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
