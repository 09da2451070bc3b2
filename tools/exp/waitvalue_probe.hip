// waitvalue_probe.hip — how long after a RUNNING kernel writes a word does a kernel gated on that word by hipStreamWaitValue32 (other
// stream) start?  (Round 3: the cheapest way found to start a helper kernel beside a GEMM that is already running; an event between two
// queues costs 7-9 us, profiles/r03_beside_timeline.txt.)   hipcc -O3 --offload-arch=gfx950 -o waitvalue_probe waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

__global__ void k_main(uint32_t *go, uint32_t value, uint64_t *stamps, int spin_us) {
    // one workgroup: stamp, publish, keep running
    if (threadIdx.x == 0) {
        stamps[0] = wall_clock64();
        __hip_atomic_store(go, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        stamps[1] = wall_clock64();
        const uint64_t t_end = stamps[1] + (uint64_t)spin_us * 100;
        while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(8);
        stamps[3] = wall_clock64();
    }
}
__global__ void k_gated(uint64_t *stamps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2] = wall_clock64();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    for (int sig = 1; sig >= 0; sig--) {
        uint32_t *go; uint64_t *stamps;
        if (sig) { if (hipExtMallocWithFlags((void **)&go, 64, hipMallocSignalMemory) != hipSuccess) { printf("signal memory: allocation failed\n"); (void)hipGetLastError(); continue; } }
        else CK(hipMalloc(&go, 64));
        CK(hipMalloc(&stamps, 64));
        CK(hipMemset(go, 0, 64));
        hipStream_t s_main, s_side;
        CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
        std::vector<double> lat, total;
        for (int it = 1; it <= 30; it++) {
            CK(hipMemsetAsync(stamps, 0, 64, s_main));
            CK(hipStreamSynchronize(s_main));
            hipError_t e = hipStreamWaitValue32(s_side, go, (uint32_t)it, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { printf("hipStreamWaitValue32 (%s memory) -> %s\n", sig ? "signal" : "plain", hipGetErrorString(e)); (void)hipGetLastError(); break; }
            hipLaunchKernelGGL(k_gated, dim3(256), dim3(256), 0, s_side, stamps);
            hipLaunchKernelGGL(k_main, dim3(1), dim3(64), 0, s_main, go, (uint32_t)it, stamps, 60);
            CK(hipStreamSynchronize(s_side));
            CK(hipStreamSynchronize(s_main));
            uint64_t h[4];
            CK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
            if (it > 5) { lat.push_back((double)(h[2] - h[1]) / 100.0); total.push_back((double)(h[3] - h[0]) / 100.0); }
        }
        if (!lat.empty()) {
            std::sort(lat.begin(), lat.end());
            printf("%s memory: gated kernel starts %.2f us (median; min %.2f, max %.2f) after the running kernel's store\n", sig ? "signal" : "plain ", lat[lat.size() / 2], lat.front(), lat.back());
        }
        // reference: the same pair ordered by an EVENT recorded before k_main (the gated kernel may start at once)
        std::vector<double> ev;
        hipEvent_t fk; CK(hipEventCreateWithFlags(&fk, hipEventDisableTiming));
        for (int it = 0; it < 20; it++) {
            CK(hipMemsetAsync(stamps, 0, 64, s_main));
            CK(hipStreamSynchronize(s_main));
            hipLaunchKernelGGL(k_main, dim3(1), dim3(64), 0, s_main, go, 0u, stamps, 30);      // "previous call's GEMM"
            CK(hipEventRecord(fk, s_main)); CK(hipStreamWaitEvent(s_side, fk, 0));
            hipLaunchKernelGGL(k_gated, dim3(256), dim3(256), 0, s_side, stamps);
            CK(hipStreamSynchronize(s_side)); CK(hipStreamSynchronize(s_main));
            uint64_t h[4];
            CK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
            if (it > 3) ev.push_back((double)((int64_t)h[2] - (int64_t)h[3]) / 100.0);
        }
        std::sort(ev.begin(), ev.end());
        printf("event between two queues: gated kernel starts %.2f us (median; min %.2f, max %.2f) after the END of the kernel it waits for\n", ev[ev.size() / 2], ev.front(), ev.back());
        hipFree(go); hipFree(stamps);
    }
    return 0;
}
