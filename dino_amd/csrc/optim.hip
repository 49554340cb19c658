// Flat-buffer optimiser-side kernels: per-tensor grad norms + clip coefficients, fused
// AdamW + teacher EMA + bf16 shadow refresh, stand-alone EMA, casts and batched transposes.
// All are pure HBM streams: 16-byte accesses, one chunk (<= DINO_CHUNK elements of ONE
// tensor, start aligned to 8 elements) per workgroup.
#include "common.h"

namespace {

struct Chunk {
    int64_t tensor, start, len;
};

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ grad,
                                                          const Chunk* __restrict__ chunks,
                                                          float* __restrict__ chunk_sq) {
    __shared__ float red[4];
    const Chunk c = chunks[blockIdx.x];
    const float* g = grad + c.start;
    float s = 0.f;
    const int64_t n4 = c.len & ~(int64_t)3;
    for (int64_t i = threadIdx.x * 4; i < n4; i += 256 * 4) {
        const f32x4 v = *(const f32x4*)(g + i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int64_t i = n4 + threadIdx.x; i < c.len; i += 256) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) chunk_sq[blockIdx.x] = s;
}

// one wave per tensor at a time (a tensor has up to a few thousand chunk partials: the 65 536-row
// prototype matrix), lane 0 keeps the wave's statistics, thread 0 folds them
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float* __restrict__ chunk_sq,
                                                         const int32_t* __restrict__ begin,
                                                         int ntensors,
                                                         const int32_t* __restrict__ has_grad,
                                                         float clip, float* __restrict__ norms,
                                                         float* __restrict__ coef,
                                                         float* __restrict__ stats) {
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    float tot = 0.f, mx = 0.f, nclip = 0.f;
    for (int t = wave; t < ntensors; t += nwave) {
        float s = 0.f;
        for (int c = begin[t] + lane; c < begin[t + 1]; c += 64) s += chunk_sq[c];
        s = wave_sum(s);
        if (lane == 0) {
            const float nrm = sqrtf(s);
            float cf = 1.f;
            const bool on = has_grad == nullptr || has_grad[t] != 0;
            if (on) {
                tot += s;
                mx = fmaxf(mx, nrm);
                if (clip > 0.f) {
                    const float k = clip / (nrm + 1e-6f);
                    if (k < 1.f) {
                        cf = k;
                        nclip += 1.f;
                    }
                }
            }
            norms[t] = on ? nrm : 0.f;
            coef[t] = cf;
        }
    }
    tot = block_sum(tot, red);
    mx = block_max(mx, red);
    nclip = block_sum(nclip, red);
    if (threadIdx.x == 0) {
        stats[0] = sqrtf(tot);
        stats[1] = mx;
        stats[2] = nclip;
    }
}

__global__ __launch_bounds__(256) void grad_scale_kernel(float* __restrict__ grad,
                                                         const Chunk* __restrict__ chunks,
                                                         const float* __restrict__ coef) {
    const Chunk c = chunks[blockIdx.x];
    const float k = coef[c.tensor];
    if (k == 1.f) return;
    float* g = grad + c.start;
    for (int64_t i = threadIdx.x; i < c.len; i += 256) g[i] *= k;
}

template <bool SHADOW>
__global__ __launch_bounds__(256) void adamw_ema_kernel(
    float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m1,
    float* __restrict__ m2, float* __restrict__ teacher, bf16* __restrict__ pshadow,
    bf16* __restrict__ tshadow, const Chunk* __restrict__ chunks, const float* __restrict__ hp,
    const float* __restrict__ coef, float beta1, float beta2, float eps, float ema_m,
    float ema_1m) {
    const Chunk c = chunks[blockIdx.x];
    const float* h = hp + c.tensor * 8;
    const float lr = h[0], wd = h[1], bc1 = h[2], bc2 = h[3];
    const bool active = h[4] != 0.f;
    const float gk = coef ? coef[c.tensor] : 1.f;
    const float step = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
    // chunk starts are 8-element aligned; lengths are padded by the host to multiples of 8
    for (int64_t i = c.start + threadIdx.x * 8; i < c.start + c.len; i += 256 * 8) {
        float p[8];
        load8<float>(param + i, p);
        if (active) {
            float g[8], a[8], b[8];
            load8<float>(grad + i, g);
            load8<float>(m1 + i, a);
            load8<float>(m2 + i, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ge = g[e] * gk;
                p[e] *= decay;
                a[e] = a[e] + (ge - a[e]) * (1.f - beta1);  // exp_avg.lerp_(grad, 1-beta1)
                b[e] = b[e] * beta2 + (1.f - beta2) * ge * ge;
                const float denom = sqrtf(b[e]) * inv_sqrt_bc2 + eps;
                p[e] -= step * (a[e] / denom);
            }
            store8<float>(m1 + i, a);
            store8<float>(m2 + i, b);
            store8<float>(param + i, p);
        }
        if (SHADOW) store8<bf16>(pshadow + i, p);
        if (teacher != nullptr) {
            float t[8];
            load8<float>(teacher + i, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = t[e] * ema_m + ema_1m * p[e];
            store8<float>(teacher + i, t);
            if (SHADOW) store8<bf16>(tshadow + i, t);
        }
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ teacher,
                                                  const float* __restrict__ student,
                                                  bf16* __restrict__ tshadow, int64_t n8, float m,
                                                  float one_m) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float t[8], s[8];
        load8<float>(teacher + i * 8, t);
        load8<float>(student + i * 8, s);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = t[e] * m + one_m * s[e];
        store8<float>(teacher + i * 8, t);
        if (tshadow) store8<bf16>(tshadow + i * 8, t);
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src,
                                                   bf16* __restrict__ dst, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<float>(src + i * 8, v);
        store8<bf16>(dst + i * 8, v);
    }
}

// LDS-tiled transpose fp32 [rows, cols] -> T [cols, rows]; blockIdx.y = matrix, blockIdx.x strides its tiles.
// 64 x 64 tiles, 16-B loads (256-B row pieces per 16 lanes) and 8-B / 16-B stores (whole 128-B / 256-B pieces of a
// destination row per 16 lanes) where the matrix allows it (offsets and both dimensions multiples of 4 -- every
// weight of the ViTs); anything else takes the element-wise 32 x 32 form.  (The 32 x 32 form alone moved the 22 M
// parameters of ViT-S at 1.5 TB/s; ViT-L has 300 M.)
struct Mat {
    int64_t src, dst, rows, cols;
};
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src_base,
                                                        T* __restrict__ dst_base,
                                                        const Mat* __restrict__ mats, int nmat) {
    __shared__ float tile[64][65];
    const Mat m = mats[blockIdx.y];
    const float* s = src_base + m.src;
    T* d = dst_base + m.dst;
    const bool wide = ((m.src | m.dst | m.rows | m.cols) & 3) == 0;
    if (wide) {
        const int tr = (int)((m.rows + 63) / 64), tc = (int)((m.cols + 63) / 64);
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16
        for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
            const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + ty + 16 * j, c = c0 + 4 * tx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < m.rows && c < m.cols) v = *(const f32x4*)(s + (int64_t)r * m.cols + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[ty + 16 * j][4 * tx + e] = v[e];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + ty + 16 * j, r = r0 + 4 * tx;   // dst[c, r .. r + 3]
                if (r < m.rows && c < m.cols) {
                    T o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(tile[4 * tx + e][ty + 16 * j]);
                    if constexpr (sizeof(T) == 2) *(u32x2*)(d + (int64_t)c * m.rows + r) = __builtin_bit_cast(u32x2, o);
                    else *(f32x4*)(d + (int64_t)c * m.rows + r) = __builtin_bit_cast(f32x4, o);
                }
            }
        }
        return;
    }
    const int tr = (int)((m.rows + 31) / 32), tc = (int)((m.cols + 31) / 32);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 8 * j, c = c0 + tx;
            tile[ty + 8 * j][tx] = (r < m.rows && c < m.cols) ? s[(int64_t)r * m.cols + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;  // dst[c, r]
            if (r < m.rows && c < m.cols) d[(int64_t)c * m.rows + r] = from_f32<T>(tile[tx][ty + 8 * j]);
        }
    }
}

}  // namespace

extern "C" int dino_abi_version(void) { return 10; }

extern "C" int dino_grad_sqnorm(const float* grad, const int64_t* chunks, int nchunks,
                                float* chunk_sq, void* stream) {
    DINO_CHECK_ARG(grad && chunks && chunk_sq && nchunks > 0);
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, grad,
                       (const Chunk*)chunks, chunk_sq);
    return dino_launch_status();
}

extern "C" int dino_clip_coef(const float* chunk_sq, const int32_t* tensor_chunk_begin, int ntensors,
                              const int32_t* has_grad, float clip, float* norms, float* coef,
                              float* stats, void* stream) {
    DINO_CHECK_ARG(chunk_sq && tensor_chunk_begin && norms && coef && stats && ntensors > 0);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, chunk_sq,
                       tensor_chunk_begin, ntensors, has_grad, clip, norms, coef, stats);
    return dino_launch_status();
}

extern "C" int dino_grad_scale(float* grad, const int64_t* chunks, int nchunks, const float* coef,
                               void* stream) {
    DINO_CHECK_ARG(grad && chunks && coef && nchunks > 0);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, grad,
                       (const Chunk*)chunks, coef);
    return dino_launch_status();
}

extern "C" int dino_adamw_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              float* teacher, void* param_shadow, void* teacher_shadow,
                              const int64_t* chunks, int nchunks, const float* hp, const float* coef,
                              float beta1, float beta2, float eps, float ema_m, float ema_one_minus_m,
                              void* stream) {
    DINO_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && chunks && hp && nchunks > 0);
    if (param_shadow && teacher) DINO_CHECK_ARG(teacher_shadow != nullptr);
    hipStream_t st = (hipStream_t)stream;
    if (param_shadow)
        hipLaunchKernelGGL(adamw_ema_kernel<true>, dim3(nchunks), dim3(256), 0, st, param, grad,
                           exp_avg, exp_avg_sq, teacher, (bf16*)param_shadow, (bf16*)teacher_shadow,
                           (const Chunk*)chunks, hp, coef, beta1, beta2, eps, ema_m, ema_one_minus_m);
    else
        hipLaunchKernelGGL(adamw_ema_kernel<false>, dim3(nchunks), dim3(256), 0, st, param, grad,
                           exp_avg, exp_avg_sq, teacher, (bf16*)nullptr, (bf16*)nullptr,
                           (const Chunk*)chunks, hp, coef, beta1, beta2, eps, ema_m, ema_one_minus_m);
    return dino_launch_status();
}

static int flat_grid(int64_t n8) {
    int64_t b = (n8 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int dino_ema(float* teacher, const float* student, void* teacher_shadow, int64_t n,
                        float m, float one_minus_m, void* stream) {
    DINO_CHECK_ARG(teacher && student && n > 0 && n % 8 == 0);
    hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, teacher,
                       student, (bf16*)teacher_shadow, n / 8, m, one_minus_m);
    return dino_launch_status();
}

extern "C" int dino_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    DINO_CHECK_ARG(src && dst && n > 0 && n % 8 == 0);
    hipLaunchKernelGGL(cast_kernel, dim3(flat_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16*)dst, n / 8);
    return dino_launch_status();
}

extern "C" int dino_transpose_batched(int dtype, const float* src_base, void* dst_base,
                                      const int64_t* table, int nmat, void* stream) {
    DINO_CHECK_ARG(src_base && dst_base && table && nmat > 0);
    dim3 grid(256, nmat);
    if (dtype == DINO_BF16)
        hipLaunchKernelGGL(transpose_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, src_base,
                           (bf16*)dst_base, (const Mat*)table, nmat);
    else if (dtype == DINO_F32)
        hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src_base,
                           (float*)dst_base, (const Mat*)table, nmat);
    else
        return DINO_ERR_BADARG;
    return dino_launch_status();
}
