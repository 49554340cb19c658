// Micro-benchmark: how fast can a CU drain a 256x192 bf16 output tile, by store pattern?
//   P0: 16 rows x 64 B per wave instruction (the GEMM epilogue today)
//   P1:  8 rows x 128 B (whole lines)
//   P2:  2.67 rows x 384 B (the tile's whole row segment, 1 KB contiguous runs)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_bench.hip -o tools/micro/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(512) void k(char* out, int M, int N, int tiles_n, int ntiles, int reps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long)M * N * 2), 0x00020000);
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    for (int r = 0; r < reps; ++r)
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = (t / tiles_n) * 256 + wm * 64, n0 = (t % tiles_n) * 192 + wn * 96;
        if (P == 0) {
            const int i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((m0 + a * 16 + i) * N + n0 + 32 * c + 8 * g) * 2, 0, 0);
        } else if (P == 1) {
            // wave covers 64 rows x 192 B as 64-col (128 B) + 32-col pieces: 8 lanes per row for the
            // 128-B part, 4 lanes per row for the rest
            const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
            for (int a = 0; a < 8; ++a)
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((m0 + a * 8 + r8) * N + n0 + (wn ? 32 : 0) + 8 * c8) * 2, 0, 0);
            const int i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((m0 + a * 16 + i) * N + n0 + (wn ? 0 : 64) + 8 * g) * 2, 0, 0);
        } else if (P == 3) {
            // what the GEMM epilogue can reach with a lane-pair swap: lane = g*16 + i; even i stores
            // (row i or i+1, chunk g of column group cA), odd i the same row's chunk g of cB
            const int i = lane & 15, g = lane >> 4;
            const int cA = wn ? 1 : 0, cB = cA + 1, cH = wn ? 0 : 2;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        v, rs, ((m0 + a * 16 + (i & ~1) + h) * N + n0 + 32 * ((i & 1) ? cB : cA) + 8 * g) * 2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((m0 + a * 16 + i) * N + n0 + 32 * cH + 8 * g) * 2, 0, 0);
            }
        } else if (P == 4) {
            // wave-private transposition: the wave's 64 x 96 sub-tile as 192-B row segments,
            // adjacent lanes contiguous (12 lanes per row, 5.33 rows per instruction)
#pragma unroll
            for (int a = 0; a < 12; ++a) {
                const int idx = a * 64 + lane, row = idx / 12, ch = idx % 12;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((m0 + row) * N + n0 + 8 * ch) * 2, 0, 0);
            }
        } else {
            // wave pair (wn = 0,1) shares rows; here each wave takes 32 of the 64 rows, whole 384 B
            const int mrow = (t / tiles_n) * 256 + wm * 64 + wn * 32, ncol = (t % tiles_n) * 192;
#pragma unroll
            for (int a = 0; a < 12; ++a) {
                const int idx = a * 64 + lane;          // 16-B chunk index in the 32 x 24 chunk block
                const int row = idx / 24, ch = idx % 24;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((mrow + row) * N + ncol + 8 * ch) * 2, 0, 0);
            }
        }
    }
}

int main(int argc, char** argv) {
    const int M = 100864, N = argc > 1 ? atoi(argv[1]) : 1536;
    const int tiles_n = N / 192, ntiles = (M / 256) * tiles_n;
    char* out;
    hipMalloc(&out, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {256, 128, 32, 8}) {
        for (int P = 0; P < 5; ++P) {
            const int reps = 5;
            auto launch = [&]() {
                if (P == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, M, N, tiles_n, ntiles * grid / 256, reps);
                if (P == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, M, N, tiles_n, ntiles * grid / 256, reps);
                if (P == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, out, M, N, tiles_n, ntiles * grid / 256, reps);
                if (P == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, out, M, N, tiles_n, ntiles * grid / 256, reps);
                if (P == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, M, N, tiles_n, ntiles * grid / 256, reps);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)(ntiles * grid / 256) * 256 * 192 * 2 * reps;
            printf("N=%d grid=%3d P%d: %8.1f us  %7.1f GB/s total  %6.1f GB/s per CU\n", N, grid, P, ms * 1e3,
                   bytes / ms / 1e6, bytes / ms / 1e6 / grid);
        }
    }
    return 0;
}
