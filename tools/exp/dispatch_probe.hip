// dispatch_probe.hip — which XCD does workgroup b of a grid LARGER than the chip land on, and when?  (diagnostic; round 4)
// Every workgroup (256 threads, LDS sized so that one fits per CU) records HW_REG_XCC_ID, its CU id, and s_memrealtime at start and end,
// and spins for a per-block duration: `uniform` (all equal) or `mixed` (the first 256 blocks 8/7 longer than the rest, like a column-balanced
// grid).  Questions: (1) is xcc(b) == (b + c) % 8 for EVERY b, also for b >= 256 (static round-robin), or do later blocks go wherever a CU
// frees up (dynamic)?  (2) in what order do the blocks of one XCD start?
//   hipcc -O3 --offload-arch=gfx950 tools/exp/dispatch_probe.hip -o tools/exp/dispatch_probe && tools/exp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct Rec { unsigned xcc, cu; unsigned long long t0, t1; };

__global__ __launch_bounds__(256, 1) void k_probe(Rec *rec, int base_us, int mixed) {
    extern __shared__ char smem[];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    int us = base_us;
    if (mixed && blockIdx.x < 256) us = base_us * 8 / 7;
    if (mixed == 2) us = base_us + (int)((blockIdx.x * 2654435761u) >> 28);   // +0..15 us of jitter
    const unsigned long long until = t0 + (unsigned long long)us * 100ull;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        smem[0] = 1;
        rec[blockIdx.x] = Rec{xcc & 15u, (hwid >> 8) & 0xFFu, t0, __builtin_amdgcn_s_memrealtime()};
    }
}

int main() {
    const int lds = 100 * 1024;     // > half of 160 KiB: one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int mode = 0; mode < 3; mode++) {
        for (int nwg : {256, 688, 1536}) {
            Rec *d;
            hipMalloc(&d, sizeof(Rec) * nwg);
            hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(256), lds, 0, d, 20, mode);
            hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(256), lds, 0, d, 20, mode);
            hipDeviceSynchronize();
            std::vector<Rec> h(nwg);
            hipMemcpy(h.data(), d, sizeof(Rec) * nwg, hipMemcpyDeviceToHost);
            hipFree(d);
            int c = (int)h[0].xcc, stat = 0, first_bad = -1;
            for (int b = 0; b < nwg; b++) {
                if ((int)h[b].xcc == (b + c) % 8) stat++;
                else if (first_bad < 0) first_bad = b;
            }
            unsigned long long tmin = h[0].t0, tmax = 0;
            for (auto &r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
            int per_xcc[8] = {0};
            for (auto &r : h) per_xcc[r.xcc & 7]++;
            // start order inside XCD of block 0: are the start times of its blocks monotone in b?
            int inversions = 0; unsigned long long prev = 0;
            for (int b = 0; b < nwg; b += 8) { if (h[b].t0 + 50 < prev) inversions++; prev = std::max(prev, h[b].t0); }
            printf("mode %d (%s)  nwg %4d: xcc(b) == (b + %d) %% 8 for %d of %d blocks (first mismatch at b = %d); blocks per XCD:", mode,
                   mode == 0 ? "uniform 20 us" : mode == 1 ? "first 256 blocks 8/7 longer" : "20 us + 0..15 us jitter", nwg, c, stat, nwg, first_bad);
            for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
            printf("; start-order inversions in block 0's XCD: %d; span %.1f us\n", inversions, (double)(tmax - tmin) / 100.0);
            if (nwg == 688 && mode != 1) {
                printf("   start times (us) of blocks 256..287: ");
                for (int b = 256; b < 288; b++) printf("%.1f/x%u ", (double)(h[b].t0 - tmin) / 100.0, h[b].xcc);
                printf("\n");
            }
        }
    }
    return 0;
}
