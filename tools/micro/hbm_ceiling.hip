// HBM streaming ceiling probe (round 3).
// Coalesced 16-B accesses, U of them in flight per lane, with and without the non-temporal hint, over buffers far
// larger than the 256 MiB Infinity Cache.  Patterns: read, write, copy (1r1w), add (2r1w), fan-out (1r2w: the
// shape of fc1 + saved derivative), rmw (read and write the same line: AdamW's shape).
// 12 repetitions each, median reported.  Build: hipcc -O3 --offload-arch=gfx950 hbm_ceiling.hip -o hbm_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// MODE 0 read, 1 write, 2 copy, 3 add (a + b -> c), 4 fan-out (a -> b, c), 5 rmw (a -> a)
// Each block-iteration covers blockDim * U consecutive 16-B words; a lane's U accesses are blockDim words apart, so
// every wave instruction is 1 KiB contiguous.
template <int MODE, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void stream(u32x4* __restrict__ a, u32x4* __restrict__ b, u32x4* __restrict__ c, size_t n) {
    const size_t span = (size_t)blockDim.x * U;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t base = blockIdx.x * span; base < n; base += (size_t)gridDim.x * span) {
        size_t i0 = base + threadIdx.x;
        u32x4 x[U], y[U];
        if constexpr (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld<NTL>(a + i0 + (size_t)u * blockDim.x);
        }
        if constexpr (MODE == 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = ld<NTL>(b + i0 + (size_t)u * blockDim.x);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = i0 + (size_t)u * blockDim.x;
            if constexpr (MODE == 0) acc += x[u];
            if constexpr (MODE == 1) st<NTS>(a + i, u32x4{1u, 2u, 3u, (unsigned)i});
            if constexpr (MODE == 2) st<NTS>(b + i, x[u]);
            if constexpr (MODE == 3) st<NTS>(c + i, x[u] + y[u]);
            if constexpr (MODE == 4) { st<NTS>(b + i, x[u]); st<NTS>(c + i, x[u] + 1u); }
            if constexpr (MODE == 5) st<NTS>(a + i, x[u] + 1u);
        }
    }
    if constexpr (MODE == 0) if (acc.x == 0x12345678u) c[0] = acc;
}

static hipEvent_t e0, e1;
template <int MODE, int U, bool NTL, bool NTS>
static void run(const char* name, double bytes_per_word, u32x4* a, u32x4* b, u32x4* c, size_t n, int grid) {
    std::vector<float> t;
    for (int rep = 0; rep < 13; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream<MODE, U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, a, b, c, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    double med = t[t.size() / 2], best = t[0];
    printf("%-8s U=%d ntl=%d nts=%d grid=%5d: median %8.1f us %5.2f TB/s   best %5.2f TB/s\n", name, U, (int)NTL, (int)NTS, grid,
           med * 1e3, bytes_per_word * 16.0 * n / med / 1e9, bytes_per_word * 16.0 * n / best / 1e9);
    fflush(stdout);
}

template <int U, bool NTL, bool NTS> static void sweep(u32x4* a, u32x4* b, u32x4* c, size_t n, int grid) {
    if (!NTS) run<0, U, NTL, NTS>("read", 1, a, b, c, n, grid);
    if (!NTL) run<1, U, NTL, NTS>("write", 1, a, b, c, n, grid);
    run<2, U, NTL, NTS>("copy", 2, a, b, c, n, grid);
    run<3, U, NTL, NTS>("add2r1w", 3, a, b, c, n, grid);
    run<4, U, NTL, NTS>("fan1r2w", 3, a, b, c, n, grid);
    run<5, U, NTL, NTS>("rmw", 2, a, b, c, n, grid);
}

int main(int argc, char** argv) {
    const size_t bytes = 1536ull << 20, n = bytes / 16;   // 3 x 1.5 GiB, each 6 x the Infinity Cache
    u32x4 *a, *b, *c;
    if (hipMalloc(&a, bytes) || hipMalloc(&b, bytes) || hipMalloc(&c, bytes)) { puts("alloc failed"); return 1; }
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 3, bytes);
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 2048, 4096}) {
        sweep<1, false, false>(a, b, c, n, grid);
        sweep<4, false, false>(a, b, c, n, grid);
        sweep<8, false, false>(a, b, c, n, grid);
        sweep<4, false, true>(a, b, c, n, grid);
        sweep<8, false, true>(a, b, c, n, grid);
        sweep<4, true, false>(a, b, c, n, grid);
        sweep<4, true, true>(a, b, c, n, grid);
        sweep<8, true, true>(a, b, c, n, grid);
    }
    return 0;
}
