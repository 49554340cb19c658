// DINO loss, forward + backward in one entry point (dino/components/loss.py:38-79).
//
// Closed form used (T = 2*ncrops - 2 loss terms, x = s / tau_s, q_i = softmax((t_i - c)/tau_t)):
//   loss      = 1/(B T) * sum_{v,b} sum_k qs_v[b,k] * (lse(x_{v,b}) - x_{v,b,k}),
//   qs_v      = sum_{iq != v} q_iq          (q_1 for v=0, q_0 for v=1, q_0+q_1 for local crops)
//   dloss/ds  = (n_v * softmax(x) - qs_v) / (tau_s B T),   n_v = #{iq != v}
// Kernels: (1) teacher row statistics (log-sum-exp of the centred, sharpened logits),
// (2) per student row: log-sum-exp pass, then gradient + loss pass (the row re-read comes from
// L2 / Infinity Cache: a row is 128-256 KiB and was touched microseconds earlier),
// (3) column sums of the teacher logits for the centre EMA (rowops colsum), (4) finalize.
#include "common.h"

namespace {

constexpr int LT = 256;

struct MaxSum {
    float m, s;
};
DEVI MaxSum ms_combine(MaxSum a, MaxSum b) {
    const float m = fmaxf(a.m, b.m);
    MaxSum r;
    r.m = m;
    // threads that saw no element carry (-inf, 0): exp(-inf - -inf) would be NaN
    r.s = (a.s != 0.f ? a.s * __expf(a.m - m) : 0.f) + (b.s != 0.f ? b.s * __expf(b.m - m) : 0.f);
    return r;
}
// log-sum-exp over the block from per-thread (max, sum) pairs; all threads get the result
DEVI float block_lse(MaxSum v, float* red /* >= 2*nwaves floats */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MaxSum b;
        b.m = __shfl_xor(v.m, o, 64);
        b.s = __shfl_xor(v.s, o, 64);
        v = ms_combine(v, b);
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[2 * w] = v.m;
        red[2 * w + 1] = v.s;
    }
    __syncthreads();
    MaxSum r;
    r.m = red[0];
    r.s = red[1];
    for (int i = 1; i < nw; ++i) {
        MaxSum b;
        b.m = red[2 * i];
        b.s = red[2 * i + 1];
        r = ms_combine(r, b);
    }
    return r.m + __logf(r.s);
}

template <typename T>
__global__ __launch_bounds__(LT) void teacher_stats_kernel(const T* __restrict__ t,
                                                           const float* __restrict__ center,
                                                           float* __restrict__ lse_t, int K,
                                                           float inv_tt) {
    __shared__ float red[2 * LT / 64];
    const int64_t row = blockIdx.x;
    const T* tr = t + row * K;
    MaxSum acc;
    acc.m = -INFINITY;
    acc.s = 0.f;
    for (int k = threadIdx.x * 8; k < K; k += LT * 8) {
        float tv[8], cv[8];
        load8<T>(tr + k, tv);
        load8<float>(center + k, cv);
        float x[8], mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = (tv[e] - cv[e]) * inv_tt;
            mx = fmaxf(mx, x[e]);
        }
        const float m = fmaxf(acc.m, mx);
        float s = acc.s * __expf(acc.m - m);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf(x[e] - m);
        acc.m = m;
        acc.s = s;
    }
    const float lse = block_lse(acc, red);
    if (threadIdx.x == 0) lse_t[row] = lse;
}

template <typename T>
__global__ __launch_bounds__(LT) void student_loss_kernel(
    const T* __restrict__ s, const T* __restrict__ t, const float* __restrict__ center,
    const float* __restrict__ lse_t, T* __restrict__ ds, float* __restrict__ loss_parts, int B,
    int K, float inv_st, float inv_tt, float coef) {
    __shared__ float red[2 * LT / 64];
    const int row = blockIdx.x;  // = v * B + b (crop-major, dino/models/__init__.py:43)
    const int v = row / B, b = row % B;
    const T* sr = s + (int64_t)row * K;
    T* dr = ds + (int64_t)row * K;
    // pass 1: log-sum-exp of x = s / tau_s
    MaxSum acc;
    acc.m = -INFINITY;
    acc.s = 0.f;
    for (int k = threadIdx.x * 8; k < K; k += LT * 8) {
        float sv[8];
        load8<T>(sr + k, sv);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sv[e] *= inv_st;
            mx = fmaxf(mx, sv[e]);
        }
        const float m = fmaxf(acc.m, mx);
        float sum = acc.s * __expf(acc.m - m);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf(sv[e] - m);
        acc.m = m;
        acc.s = sum;
    }
    const float lse = block_lse(acc, red);
    // pass 2: gradient + loss
    const bool use0 = v != 0, use1 = v != 1;  // which teacher views contribute
    const float nv = (use0 ? 1.f : 0.f) + (use1 ? 1.f : 0.f);
    const T* t0 = t + (int64_t)b * K;
    const T* t1 = t + (int64_t)(B + b) * K;
    const float l0 = lse_t[b], l1 = lse_t[B + b];
    const float gscale = inv_st * coef;
    float part = 0.f;
    for (int k = threadIdx.x * 8; k < K; k += LT * 8) {
        float sv[8], cv[8], qs[8], g[8];
        load8<T>(sr + k, sv);
        load8<float>(center + k, cv);
#pragma unroll
        for (int e = 0; e < 8; ++e) qs[e] = 0.f;
        if (use0) {
            float tv[8];
            load8<T>(t0 + k, tv);
#pragma unroll
            for (int e = 0; e < 8; ++e) qs[e] += __expf((tv[e] - cv[e]) * inv_tt - l0);
        }
        if (use1) {
            float tv[8];
            load8<T>(t1 + k, tv);
#pragma unroll
            for (int e = 0; e < 8; ++e) qs[e] += __expf((tv[e] - cv[e]) * inv_tt - l1);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float lp = sv[e] * inv_st - lse;  // log softmax
            part -= qs[e] * lp;
            g[e] = (nv * __expf(lp) - qs[e]) * gscale;
        }
        store8_nt<T>(dr + k, g);
    }
    part = block_sum(part, red);
    if (threadIdx.x == 0) loss_parts[row] = part * coef;
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* __restrict__ x,
                                                          float* __restrict__ out, int rows, int K,
                                                          int slab) {
    // thread = 8 columns; grid.y = row slabs; atomics combine the slabs
    const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (col >= K) return;
    const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = r0; r < r1; ++r) {
        float v[8];
        load8<T>(x + (int64_t)r * K + col, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(out + col + e, acc[e]);
}

__global__ __launch_bounds__(256) void center_update_kernel(float* __restrict__ center,
                                                            float* __restrict__ center_acc, int K,
                                                            float momentum, float inv_count,
                                                            const float* __restrict__ loss_parts,
                                                            int nparts, float* __restrict__ loss_out) {
    __shared__ float red[4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (center != nullptr && k < K) {
        center[k] = center[k] * momentum + center_acc[k] * inv_count * (1.f - momentum);
        center_acc[k] = 0.f;
    }
    if (blockIdx.x == 0 && loss_out != nullptr) {
        float s = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) s += loss_parts[i];
        s = block_sum(s, red);
        if (threadIdx.x == 0) loss_out[0] = s;
    }
}

}  // namespace

extern "C" int dino_loss_fwd_bwd(int dtype, const void* student, const void* teacher,
                                 const float* center, void* dstudent, float* loss_parts,
                                 float* center_acc, float* row_stats, int B, int ncrops, int K,
                                 float inv_student_temp, float inv_teacher_temp, void* stream) {
    DINO_CHECK_ARG(student && teacher && center && dstudent && loss_parts && row_stats);
    DINO_CHECK_ARG(B > 0 && ncrops >= 2 && K > 0 && K % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    const float coef = 1.f / ((float)B * (float)(2 * ncrops - 2));
    const int slab = 32;
    dim3 cgrid((K / 8 + 255) / 256, (2 * B + slab - 1) / slab);
    if (dtype == DINO_BF16) {
        typedef bf16 T;
        hipLaunchKernelGGL(teacher_stats_kernel<T>, dim3(2 * B), dim3(LT), 0, st, (const T*)teacher,
                           center, row_stats, K, inv_teacher_temp);
        hipLaunchKernelGGL(student_loss_kernel<T>, dim3(ncrops * B), dim3(LT), 0, st,
                           (const T*)student, (const T*)teacher, center, row_stats, (T*)dstudent,
                           loss_parts, B, K, inv_student_temp, inv_teacher_temp, coef);
        if (center_acc)
            hipLaunchKernelGGL(colsum_rows_kernel<T>, cgrid, dim3(256), 0, st, (const T*)teacher,
                               center_acc, 2 * B, K, slab);
    } else if (dtype == DINO_F32) {
        typedef float T;
        hipLaunchKernelGGL(teacher_stats_kernel<T>, dim3(2 * B), dim3(LT), 0, st, (const T*)teacher,
                           center, row_stats, K, inv_teacher_temp);
        hipLaunchKernelGGL(student_loss_kernel<T>, dim3(ncrops * B), dim3(LT), 0, st,
                           (const T*)student, (const T*)teacher, center, row_stats, (T*)dstudent,
                           loss_parts, B, K, inv_student_temp, inv_teacher_temp, coef);
        if (center_acc)
            hipLaunchKernelGGL(colsum_rows_kernel<T>, cgrid, dim3(256), 0, st, (const T*)teacher,
                               center_acc, 2 * B, K, slab);
    } else {
        return DINO_ERR_BADARG;
    }
    return dino_launch_status();
}

extern "C" int dino_center_update(float* center, float* center_acc, int K, float momentum,
                                  float inv_count, const float* loss_parts, int nparts,
                                  float* loss_out, void* stream) {
    DINO_CHECK_ARG((center != nullptr) == (center_acc != nullptr) && (center || loss_out));
    if (center) DINO_CHECK_ARG(K > 0);
    if (loss_out) DINO_CHECK_ARG(loss_parts && nparts > 0);
    hipLaunchKernelGGL(center_update_kernel, dim3(center ? (K + 255) / 256 : 1), dim3(256), 0,
                       (hipStream_t)stream, center, center_acc, K, momentum, inv_count, loss_parts,
                       nparts, loss_out);
    return dino_launch_status();
}
