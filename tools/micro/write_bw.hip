// Pure write / read / copy bandwidth with fully coalesced 16-B accesses over a 2.4 GB buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void wr(u32x4* p, size_t n) {
    u32x4 v = {1, 2, 3, 4};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void rd(const u32x4* p, size_t n, u32x4* sink) {
    u32x4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a.x == 0x12345678) sink[0] = a;
}
__global__ void cp(const u32x4* p, u32x4* q, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) q[i] = p[i];
}
int main() {
    const size_t bytes = 2400ull << 20, n = bytes / 16;
    u32x4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {2048, 8192}) {
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, a, n);
                if (k == 1) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, n, b);
                if (k == 2) hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("grid %d %s: %.1f us, %.2f TB/s\n", grid, k == 0 ? "write" : k == 1 ? "read" : "copy(r+w)", ms * 1e3,
                                (k == 2 ? 2.0 : 1.0) * bytes / ms / 1e9);
            }
        }
    }
    return 0;
}
