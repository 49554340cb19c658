// Diagnostics that have no counterpart in the reference.
//
// dino_diag_occupy_cus: a kernel that does nothing but HOLD compute units for a given time -- n workgroups of 256
// threads and 32 KB of LDS each (so that, like an RCCL kernel and unlike a register-free sleeper, they do not fit
// beside a resident GEMM workgroup, nor it beside them) that sleep until `micros` have passed on the constant 100 MHz clock (bounded: at most ~4 x the request in
// sleep instructions, so the grid always drains).  It stands in, on a one-GPU box, for the kernels of another
// stream that the persistent GEMM / attention kernels meet on a multi-GPU node: RCCL's all-reduce of a gradient
// bucket running under backward.  Those kernels keep a whole CU per workgroup (152 KB of LDS, 3 x 168 registers per
// SIMD), nothing of another kernel fits beside one, so a workgroup whose CU is held starts late; bench.py
// --occupy n,us launches this where a gradient bucket's all-reduce would start and reports what the step loses
// (DESIGN.md section 5).
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void occupy_cus_kernel(long long ticks, int max_sleeps) {
    extern __shared__ char hold[];
    if (ticks < 0) hold[threadIdx.x] = 0;   // (keeps the allocation)
    const long long t0 = wall_clock64();
    for (int z = 0; z < max_sleeps; ++z) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(127);   // ~8 k cycles
    }
}
}  // namespace

extern "C" int dino_diag_occupy_cus(int workgroups, int micros, void* stream) {
    DINO_CHECK_ARG(workgroups > 0 && workgroups <= 1024 && micros > 0 && micros <= 100000);
    // wall_clock64: 100 MHz; one s_sleep(127) is ~3.4 us at 2.4 GHz: 4 x the request bounds the loop
    const long long ticks = (long long)micros * 100;
    const int max_sleeps = micros * 4 / 3 + 16;
    hipLaunchKernelGGL(occupy_cus_kernel, dim3(workgroups), dim3(256), 32 * 1024, (hipStream_t)stream, ticks, max_sleeps);
    return dino_launch_status();
}
