// beside_exp.hip — diagnostic harness for csrc/gemm_beside.h (not part of the product): libbeside_exp.so exports
//   exp_beside(mode, abl, ...):  mode 0 decoder + gated GEMM (any-order launch), 1 gated GEMM only, 2 decoder only, 3 decoder, then the
//   gated GEMM as an ORDINARY launch (ordered behind the decoder).  Built with GB_STAMPS: both kernels write s_memrealtime stamps into
//   the sync area from byte 16384 (the sync buffer must be >= 1 MiB).
#define GB_STAMPS 1
#include <hip/hip_ext.h>
#include "parked/gemm_beside.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
template <int ABL> static int run(int mode, const bf16_t *x, const uint8_t *packed, const float *am, bf16_t *wd, uint32_t *sync, void *out, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kg = k_gemm_gated<bf16_t, ABL>;
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, GD_LDS) != hipSuccess) return -2; done = true; }
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    AbsmaxView v{am, nullptr, nullptr, 0};
    if (mode == 6) {      // WaitValue: decoder on a side stream behind hipStreamWaitValue32(go == epoch); the GEMM publishes the epoch
        static hipStream_t side = nullptr;
        static uint32_t epoch = 0;
        if (!side) hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        ++epoch;
        uint32_t *go = sync + ((N + 255) / 256) * GB_COL_WORDS + 1;
        if (hipStreamWaitValue32(side, go, epoch, hipStreamWaitValueEq, 0xFFFFFFFFu) != hipSuccess) return -7;
        hipLaunchKernelGGL((k_decode_beside<bf16_t, false>), dim3((unsigned)((N + 15) / 16)), dim3(256), GB_DEC_LDS, side, packed, v, (int)MBNB_NF4, wd, sync, N, K, K, 0, (int)(K >> 9), 0u);
        hipLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, x, (const bf16_t *)wd, sync, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K, K, epoch);
        return (int)hipGetLastError();
    }
    if (mode == 4 || mode == 5) {      // two streams, fork / join of events; 5: GEMM launched first
        static hipStream_t side = nullptr;
        static hipEvent_t fk = nullptr, jn = nullptr;
        if (!side) { hipStreamCreateWithFlags(&side, hipStreamNonBlocking); hipEventCreateWithFlags(&fk, hipEventDisableTiming); hipEventCreateWithFlags(&jn, hipEventDisableTiming); }
        hipEventRecord(fk, st); hipStreamWaitEvent(side, fk, 0);
        if (mode == 5) hipLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, x, (const bf16_t *)wd, sync, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K, K, 0u);
        hipLaunchKernelGGL((k_decode_beside<bf16_t, false>), dim3((unsigned)((N + 15) / 16)), dim3(256), GB_DEC_LDS, side, packed, v, (int)MBNB_NF4, wd, sync, N, K, K, 0, (int)(K >> 9), 0u);
        if (mode == 4) hipLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, x, (const bf16_t *)wd, sync, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K, K, 0u);
        hipEventRecord(jn, side); hipStreamWaitEvent(st, jn, 0);
        return (int)hipGetLastError();
    }
    if (mode != 1)
        hipLaunchKernelGGL((k_decode_beside<bf16_t, false>), dim3((unsigned)((N + 15) / 16)), dim3(256), GB_DEC_LDS, st, packed, v, (int)MBNB_NF4, wd, sync, N, K, K, 0, (int)(K >> 9), 0u);
    if (mode == 0)
        hipExtLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, nullptr, nullptr, hipExtAnyOrderLaunch, x, (const bf16_t *)wd, sync, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K, K, 0u);
    else if (mode == 1 || mode == 3)
        hipLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, x, (const bf16_t *)wd, sync, (const bf16_t *)nullptr, out, (int)MBNB_BF16, M, N, K, K, 0u);
    return (int)hipGetLastError();
}
extern "C" int exp_beside(int mode, int abl, const void *X_, const uint8_t *packed, const float *absmax, void *wd, void *sync, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (abl) {
#define X(v) case v: return run<v>(mode, static_cast<const bf16_t *>(X_), packed, absmax, static_cast<bf16_t *>(wd), static_cast<uint32_t *>(sync), out, M, N, K, st);
        X(0) X(3) X(7)
#undef X
        default: return -1;
    }
}
