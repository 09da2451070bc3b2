// gemm_beside.hip — launch of k_decode_beside + k_gemm_gated (gemm_beside.h): large-M matmul_4bit with the dequantise pass on
// its own waves BESIDE the GEMM's (two launches on two streams, forked and joined by events inside the call).  Own translation unit.
#include <mutex>
#include <hip/hip_ext.h>
#include "gemm_beside.h"

namespace mbnb {

struct DensePlan { int fm; int64_t slices; };
DensePlan gemm_dense_plan(int64_t, int64_t, int64_t);
bool gemm_dense_shape(int64_t, int64_t, int64_t, int64_t);

// One side stream + two events per device, made on first use and kept for the life of the process.  The mutex also covers the
// record / wait / launch sequence of a call: the events are shared, and a second host thread recording `fork` between this
// thread's record and its wait would tie this call's decoder to the other thread's stream position.
struct BesideCtx { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool tried = false; };
static std::mutex g_beside_mu;
static BesideCtx g_beside[64];

static BesideCtx *beside_ctx() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    BesideCtx &c = g_beside[dev];
    if (!c.tried) {
        c.tried = true;
        if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            c.side = nullptr;
        }
    }
    return c.side ? &c : nullptr;
}

bool gemm_beside_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight, int blocksize) {
    if (blocksize != 64 || K_weight < K || K_weight % 64 != 0 || K % (64 * GB_SLAB) != 0 || K / (64 * GB_SLAB) > GB_MAX_SLABS || K < 1024) return false;
    if (!gemm_dense_shape(M, N, K, K_weight)) return false;
    const DensePlan plan = gemm_dense_plan(M, N, K);
    if (plan.fm != 8 || plan.slices != 1) return false;          // where the decode-once path runs unsplit 256 x 256 tiles
    if (N * K_weight * 2 >= ((int64_t)1 << 31)) return false;       // the decoder's 32-bit offsets
    return true;
}
int64_t gemm_beside_sync_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight, int blocksize) {
    if (!gemm_beside_shape(M, N, K, K_weight, blocksize)) return 0;
    return (gb_sync_bytes((N + 255) / 256) + 255) & ~(int64_t)255;
}

template <typename T>
static int launch_beside(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N, int64_t K_weight, int qt,
                         const void *bias, int out_dtype, void *out, void *ws, void *sync, hipStream_t st, int order) {
    auto kg = k_gemm_gated<T>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kg), GD_LDS, "matmul_4bit(beside)")) return rc;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const unsigned dgrid = (unsigned)((N + 15) / 16);
    const int nslab = (int)(K >> 9);
    if (!(order & 6)) {
        // ONE stream: the decoder is an ordinary launch (it starts behind everything the stream holds, the previous call's GEMM
        // included); the GEMM follows as an ANY-ORDER launch (no barrier bit on its packet: it may start while the decoder
        // runs -- its slab gates are what orders it behind the decoder's stores).  Whatever the caller launches next is an
        // ordinary launch again and waits for both.
        if (am.i8 != nullptr)
            hipLaunchKernelGGL((k_decode_beside<T, true>), dim3(dgrid), dim3(256), GB_DEC_LDS, st, packed, am, qt, static_cast<T *>(ws),
                               static_cast<uint32_t *>(sync), N, K, K_weight, 0, nslab, 0u);
        else
            hipLaunchKernelGGL((k_decode_beside<T, false>), dim3(dgrid), dim3(256), GB_DEC_LDS, st, packed, am, qt, static_cast<T *>(ws),
                               static_cast<uint32_t *>(sync), N, K, K_weight, 0, nslab, 0u);
        if (int rc = check_launch("matmul_4bit(beside decoder)")) return rc;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone)      // inside a capture: an ordinary kernel node behind the decoder's
            hipLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, static_cast<const T *>(A), static_cast<const T *>(ws),
                               static_cast<uint32_t *>(sync), static_cast<const T *>(bias), out, out_dtype, M, N, K, K_weight, 0u);
        else
            hipExtLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, nullptr, nullptr, hipExtAnyOrderLaunch, static_cast<const T *>(A),
                                  static_cast<const T *>(ws), static_cast<uint32_t *>(sync), static_cast<const T *>(bias), out, out_dtype, M, N, K, K_weight, 0u);
        return check_launch("matmul_4bit(beside)");
    }
    std::lock_guard<std::mutex> lk(g_beside_mu);
    BesideCtx *c = beside_ctx();
    if (c == nullptr) return MBNB_NOT_APPLICABLE;
    auto decode_range = [&](hipStream_t s, int u0, int u1) {
        if (am.i8 != nullptr)
            hipLaunchKernelGGL((k_decode_beside<T, true>), dim3(dgrid), dim3(256), GB_DEC_LDS, s, packed, am, qt, static_cast<T *>(ws),
                               static_cast<uint32_t *>(sync), N, K, K_weight, u0, u1, 0u);
        else
            hipLaunchKernelGGL((k_decode_beside<T, false>), dim3(dgrid), dim3(256), GB_DEC_LDS, s, packed, am, qt, static_cast<T *>(ws),
                               static_cast<uint32_t *>(sync), N, K, K_weight, u0, u1, 0u);
    };
    if (order & 4) {
        // SPLIT: a dependency between two queues costs ~8 us on this platform (profiles/r03_beside_timeline.txt), so the slabs the
        // k-loop needs first are decoded in the caller's stream, in front of the GEMM (ordinary launches), and the rest on the side
        // stream behind an event recorded at the call's start: by the time the side stream has reacted the GEMM is running, the
        // decoder's waves sit beside it, and its slabs are ready long before their gates.
        const int head = nslab > 2 ? 1 : nslab;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (head < nslab) {
            if (hipEventRecord(c->fork, st) != hipSuccess || hipStreamWaitEvent(c->side, c->fork, 0) != hipSuccess) return check_launch("matmul_4bit(beside fork)");
            decode_range(c->side, head, nslab);
        }
        decode_range(st, 0, head);
        hipLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, static_cast<const T *>(A), static_cast<const T *>(ws),
                           static_cast<uint32_t *>(sync), static_cast<const T *>(bias), out, out_dtype, M, N, K, K_weight, 0u);
        int rc = check_launch("matmul_4bit(beside split)");
        // No join outside a capture: the GEMM cannot finish before it has seen the flag of the decoder's last slab, i.e. before the
        // decoder's last store has left, and whatever follows in the caller's stream follows the GEMM.  (A capture needs the side
        // stream back in the caller's.)
        if (head < nslab && cap != hipStreamCaptureStatusNone)
            if (hipEventRecord(c->join, c->side) != hipSuccess || hipStreamWaitEvent(st, c->join, 0) != hipSuccess) return rc ? rc : check_launch("matmul_4bit(beside join)");
        return rc;
    }
    order &= 1;
    if (hipEventRecord(c->fork, st) != hipSuccess || hipStreamWaitEvent(c->side, c->fork, 0) != hipSuccess) return check_launch("matmul_4bit(beside fork)");
    auto decode = [&]() { decode_range(c->side, 0, nslab); };
    auto gemm = [&]() {
        hipLaunchKernelGGL(kg, dim3((unsigned)tiles), dim3(256), GD_LDS, st, static_cast<const T *>(A), static_cast<const T *>(ws),
                           static_cast<uint32_t *>(sync), static_cast<const T *>(bias), out, out_dtype, M, N, K, K_weight, 0u);
    };
    if (order == 0) { decode(); gemm(); } else { gemm(); decode(); }
    int rc = check_launch("matmul_4bit(beside)");
    // the join is made even after a failed launch: the caller's stream must not run ahead of the side stream
    if (hipEventRecord(c->join, c->side) != hipSuccess || hipStreamWaitEvent(st, c->join, 0) != hipSuccess) return rc ? rc : check_launch("matmul_4bit(beside join)");
    return rc;
}

// Returns MBNB_NOT_APPLICABLE when the path does not serve the call, otherwise the launch status.  ws: the Wd scratch
// (N * K_weight * 2 bytes, 256-byte aligned); sync: gemm_beside_sync_bytes() bytes, ZERO on entry (zero again when the call's work
// has finished).  order: 0 = one stream, the GEMM as an any-order launch behind the decoder; 2 = two streams (fork / join of events), decoder
// launched first; 3 = two streams, GEMM first (2 and 3: diagnostic); 4 = split: the first slab in the caller's stream in
// front of the GEMM, the rest on the side stream.
int matmul_4bit_beside_path(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N, int64_t K_weight,
                            int blocksize, int qt, int w_dtype, const void *bias, int out_dtype, void *out, void *ws, int64_t ws_bytes,
                            void *sync, int64_t sync_bytes, hipStream_t st, int order) {
    if (w_dtype != MBNB_F16 && w_dtype != MBNB_BF16) return MBNB_NOT_APPLICABLE;
    if (ws == nullptr || sync == nullptr) return MBNB_NOT_APPLICABLE;
    if (!gemm_beside_shape(M, N, K, K_weight, blocksize)) return MBNB_NOT_APPLICABLE;
    if (ws_bytes < N * K_weight * 2 || sync_bytes < gemm_beside_sync_bytes(M, N, K, K_weight, blocksize)) return MBNB_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(ws) & 255) || (reinterpret_cast<uintptr_t>(packed) & 15) ||
        (reinterpret_cast<uintptr_t>(sync) & 3))
        return MBNB_NOT_APPLICABLE;
    if (am.i8 != nullptr ? (am.am2 == nullptr || am.bs2 <= 0 || (am.bs2 & (am.bs2 - 1)) != 0) : (am.f32 == nullptr || (reinterpret_cast<uintptr_t>(am.f32) & 3))) return MBNB_NOT_APPLICABLE;
    int rc;
    if (w_dtype == MBNB_F16) rc = launch_beside<f16_t>(A, M, K, packed, am, N, K_weight, qt, bias, out_dtype, out, ws, sync, st, order);
    else rc = launch_beside<bf16_t>(A, M, K, packed, am, N, K_weight, qt, bias, out_dtype, out, ws, sync, st, order);
    if (rc == 0) set_kernel_name("decode_beside+gated");
    return rc;
}

}  // namespace mbnb
