// tr8_probe.hip — what does ds_read_b64_tr_b8 deliver?  LDS holds 16-bit ids (two byte planes); every lane passes an address
// and prints the source byte index of each of its 8 result bytes.   hipcc --offload-arch=gfx950 -o tr8_probe tr8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(int stride, int mode, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) unsigned char lo[4096], hi[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) { lo[i] = i & 255; hi[i] = i >> 8; }
    __syncthreads();
    const int lane = threadIdx.x;
    // mode 0: lane l -> address l * stride;  mode 1: lane l -> (l / 2) * stride + (l % 2) * 8  (row per lane pair)
    const int off = mode == 0 ? lane * stride : (lane >> 1) * stride + (lane & 1) * 8;
    v2i a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(lo + off));
    v2i b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(hi + off));
    for (int e = 0; e < 8; e++) {
        const uint32_t l8 = ((uint32_t)a[e >> 2] >> (8 * (e & 3))) & 255, h8 = ((uint32_t)b[e >> 2] >> (8 * (e & 3))) & 255;
        out[lane * 8 + e] = l8 | (h8 << 8);
    }
}
int main() {
    uint32_t *d, h[512];
    hipMalloc(&d, sizeof(h));
    const int cfg[][2] = {{8, 0}, {64, 1}, {256, 1}, {16, 0}};
    for (auto &c : cfg) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d mode %d (source byte index of result bytes 0..7 per lane)\n", c[0], c[1]);
        for (int l = 0; l < 64; l++) {
            printf("  lane %2d:", l);
            for (int e = 0; e < 8; e++) printf(" %4u", h[l * 8 + e]);
            printf("\n");
        }
    }
    return 0;
}
