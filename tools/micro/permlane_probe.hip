// What v_permlane16_swap / v_permlane32_swap do on gfx950 (lane ids in, both results out).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned u = threadIdx.x, w = 100 + threadIdx.x;
    auto a = __builtin_amdgcn_permlane16_swap(u, w, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(u, w, false, false);
    o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"16swap[0]", "16swap[1]", "32swap[0]", "32swap[1]"};
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; i += 4) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
