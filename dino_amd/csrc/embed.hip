// Token assembly kernels: im2col for the patch-embedding GEMM, positional tables (bicubic
// resize as a fixed 16-tap gather), CLS rows, and the matching gradient scatter.
#include "common.h"

namespace {

// One thread = 8 consecutive k (same c, dy; dx..dx+7) of one patch row when P % 8 == 0;
// generic scalar path otherwise (P = 14).
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img,
                                                     T* __restrict__ out, int Bc, int H, int W,
                                                     int P, int gh, int gw, int Kp /* padded K */,
                                                     int cls_row) {
    const int KK = 3 * P * P;
    const int rpc = gh * gw + cls_row;  // rows per crop
    const int64_t rows = (int64_t)Bc * rpc;
    const int kch = Kp / 8;
    const int64_t total = rows * kch;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / kch;
        const int k0 = (int)(idx % kch) * 8;
        const int b = (int)(row / rpc), pr = (int)(row % rpc) - cls_row;
        const int py = pr / gw, px = pr % gw;
        float v[8];
        if (P % 8 == 0 && W % 4 == 0) {
            // fast path: the 8 values are consecutive pixels of one image row
            if (pr >= 0) {
                const int c = k0 / (P * P), r = k0 % (P * P), dy = r / P, dx = r % P;
                const float* src = img + (((int64_t)b * 3 + c) * H + py * P + dy) * W + px * P + dx;
                const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = lo[e];
                    v[4 + e] = hi[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            store8<T>(out + row * Kp + k0, v);
            continue;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float x = 0.f;
            if (k < KK && pr >= 0) {
                const int c = k / (P * P), r = k % (P * P), dy = r / P, dx = r % P;
                x = img[(((int64_t)b * 3 + c) * H + py * P + dy) * W + px * P + dx];
            }
            v[e] = x;
        }
        store8<T>(out + row * Kp + k0, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pos_table_kernel(const float* __restrict__ pos,
                                                        const int32_t* __restrict__ idx,
                                                        const float* __restrict__ w, int ntaps,
                                                        T* __restrict__ out, int n_dst, int D) {
    const int total = (n_dst + 1) * D;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
        const int i = o / D, d = o % D;
        float v;
        if (i == 0 || ntaps == 0) {
            v = pos[o];
        } else {
            v = 0.f;
            for (int t = 0; t < ntaps; ++t)
                v += w[(i - 1) * ntaps + t] * pos[(1 + idx[(i - 1) * ntaps + t]) * D + d];
        }
        out[o] = from_f32<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cls_rows_kernel(T* __restrict__ tokens,
                                                       const float* __restrict__ cls,
                                                       const float* __restrict__ pos0, int Bc, int N,
                                                       int D) {
    const int total = Bc * D;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
        const int b = o / D, d = o % D;
        tokens[(int64_t)b * N * D + d] = from_f32<T>(cls[d] + pos0[d]);
    }
}

__global__ __launch_bounds__(256) void pos_table_bwd_kernel(const float* __restrict__ dloc,
                                                            const int32_t* __restrict__ idx,
                                                            const float* __restrict__ w, int ntaps,
                                                            float* __restrict__ dpos, int n_dst,
                                                            int D) {
    const int total = (n_dst + 1) * D;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
        const int i = o / D, d = o % D;
        const float g = dloc[o];
        if (i == 0 || ntaps == 0) {
            atomicAdd(dpos + o, g);
        } else {
            for (int t = 0; t < ntaps; ++t)
                atomicAdd(dpos + (1 + idx[(i - 1) * ntaps + t]) * D + d, w[(i - 1) * ntaps + t] * g);
        }
    }
}

int flat_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int dino_im2col(int dtype, const float* img, void* out, int Bc, int H, int W, int P,
                           int cls_row, void* stream) {
    DINO_CHECK_ARG(img && out && Bc > 0 && P > 0 && H >= P && W >= P);
    const int gh = H / P, gw = W / P;
    const int Kp = (3 * P * P + 7) / 8 * 8;
    cls_row = cls_row ? 1 : 0;
    const int64_t total = (int64_t)Bc * (gh * gw + cls_row) * (Kp / 8);
    if (dtype == DINO_BF16)
        hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(flat_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, img, (bf16*)out, Bc, H, W, P, gh, gw, Kp, cls_row);
    else if (dtype == DINO_F32)
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(flat_grid(total)), dim3(256), 0,
                           (hipStream_t)stream, img, (float*)out, Bc, H, W, P, gh, gw, Kp, cls_row);
    else
        return DINO_ERR_BADARG;
    return dino_launch_status();
}

extern "C" int dino_pos_table(int dtype, const float* pos, const int32_t* idx, const float* w,
                              int ntaps, void* out, int n_dst, int D, void* stream) {
    DINO_CHECK_ARG(pos && out && n_dst > 0 && D > 0 && ntaps >= 0);
    if (ntaps > 0) DINO_CHECK_ARG(idx && w);
    const int grid = flat_grid((int64_t)(n_dst + 1) * D);
    if (dtype == DINO_BF16)
        hipLaunchKernelGGL(pos_table_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pos,
                           idx, w, ntaps, (bf16*)out, n_dst, D);
    else if (dtype == DINO_F32)
        hipLaunchKernelGGL(pos_table_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pos,
                           idx, w, ntaps, (float*)out, n_dst, D);
    else
        return DINO_ERR_BADARG;
    return dino_launch_status();
}

extern "C" int dino_cls_rows(int dtype, void* tokens, const float* cls, const float* pos0, int Bc,
                             int N, int D, void* stream) {
    DINO_CHECK_ARG(tokens && cls && pos0 && Bc > 0 && N > 0 && D > 0);
    const int grid = flat_grid((int64_t)Bc * D);
    if (dtype == DINO_BF16)
        hipLaunchKernelGGL(cls_rows_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (bf16*)tokens, cls, pos0, Bc, N, D);
    else if (dtype == DINO_F32)
        hipLaunchKernelGGL(cls_rows_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (float*)tokens, cls, pos0, Bc, N, D);
    else
        return DINO_ERR_BADARG;
    return dino_launch_status();
}

extern "C" int dino_pos_table_bwd(const float* dpos_local, const int32_t* idx, const float* w,
                                  int ntaps, float* dpos, int n_dst, int D, void* stream) {
    DINO_CHECK_ARG(dpos_local && dpos && n_dst >= 0 && D > 0 && ntaps >= 0);
    if (ntaps > 0) DINO_CHECK_ARG(idx && w);
    hipLaunchKernelGGL(pos_table_bwd_kernel, dim3(flat_grid((int64_t)(n_dst + 1) * D)), dim3(256), 0,
                       (hipStream_t)stream, dpos_local, idx, w, ntaps, dpos, n_dst, D);
    return dino_launch_status();
}

// ---------------------------------------------------------------------------------------------
// ToTensor + Normalize on the device (dino/data/augmentations.py:102-107): the loader ships uint8
// crops (a quarter of the PCIe bytes), this turns them into the fp32 NCHW crops the step reads.
// Arithmetic in torchvision's order, each operation rounded once: (float(u8) / 255 - mean) / std.
// ---------------------------------------------------------------------------------------------
namespace {

struct NormParams {
    float mean[4], std[4];
};

// one thread = VEC consecutive pixels of one image row, all channels
template <bool HWC, int VEC>
__global__ __launch_bounds__(256) void to_tensor_normalize_kernel(const uint8_t* __restrict__ src,
                                                                  float* __restrict__ dst,
                                                                  int64_t npix /* B*H*W */, int C,
                                                                  int64_t HW, NormParams np) {
    const int64_t total = npix / VEC;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int64_t p0 = idx * VEC;
        const int64_t b = p0 / HW, q = p0 % HW;
        for (int c = 0; c < C; ++c) {
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const uint8_t u = HWC ? src[(p0 + e) * C + c] : src[(b * C + c) * HW + q + e];
                v[e] = ((float)u / 255.0f - np.mean[c]) / np.std[c];
            }
            float* o = dst + (b * C + c) * HW + q;
            if (VEC == 4)
                __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4*)o);
            else
                o[0] = v[0];
        }
    }
}

}  // namespace

extern "C" int dino_to_tensor_normalize(const uint8_t* src, float* dst, int64_t B, int C, int H, int W,
                                        int src_hwc, const float* mean, const float* std,
                                        void* stream) {
    DINO_CHECK_ARG(src && dst && mean && std && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0);
    NormParams np{};
    for (int c = 0; c < C; ++c) {
        DINO_CHECK_ARG(std[c] != 0.f);
        np.mean[c] = mean[c];
        np.std[c] = std[c];
    }
    const int64_t HW = (int64_t)H * W, npix = B * HW;
    const bool v4 = HW % 4 == 0 && ((uintptr_t)dst & 15) == 0;
    const int grid = flat_grid(v4 ? npix / 4 : npix);
    hipStream_t st = (hipStream_t)stream;
    if (src_hwc) {
        if (v4)
            hipLaunchKernelGGL((to_tensor_normalize_kernel<true, 4>), dim3(grid), dim3(256), 0, st, src,
                               dst, npix, C, HW, np);
        else
            hipLaunchKernelGGL((to_tensor_normalize_kernel<true, 1>), dim3(grid), dim3(256), 0, st, src,
                               dst, npix, C, HW, np);
    } else {
        if (v4)
            hipLaunchKernelGGL((to_tensor_normalize_kernel<false, 4>), dim3(grid), dim3(256), 0, st,
                               src, dst, npix, C, HW, np);
        else
            hipLaunchKernelGGL((to_tensor_normalize_kernel<false, 1>), dim3(grid), dim3(256), 0, st,
                               src, dst, npix, C, HW, np);
    }
    return dino_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Stochastic depth (drop_path, dino/models/vision_transformer.py:52-62): scale = floor(keep + U[0,1)) / keep
// per (block, branch, crop), drawn on the device from a counter-based generator (splitmix64 of
// seed + counter: no state, any launch shape gives the same numbers), for the whole depth in one launch;
// optionally expanded to one value per token row (several resolution groups in one token matrix).
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t counter, float keep) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 bits: [0, 1)
    return floorf(keep + u) / keep;
}

__global__ __launch_bounds__(256) void drop_path_scales_kernel(uint64_t seed, const float* __restrict__ keep, int L,
                                                               int Bct, const int64_t* __restrict__ row_crop,
                                                               int64_t M, float* __restrict__ scales,
                                                               float* __restrict__ row_scales) {
    const int64_t per = (int64_t)Bct + (row_scales ? M : 0);
    const int64_t total = (int64_t)L * 2 * per;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t lw = idx / per, r = idx - lw * per;   // lw = block * 2 + branch
        const bool row = r >= Bct;
        const int64_t crop = row ? row_crop[r - Bct] : r;
        const float v = drop_scale(seed, (uint64_t)(lw * Bct + crop), keep[lw >> 1]);
        if (row) row_scales[lw * M + (r - Bct)] = v;
        else scales[lw * Bct + r] = v;
    }
}

}  // namespace

extern "C" int dino_drop_path_scales(uint64_t seed, const float* keep, int L, int Bct, const int64_t* row_crop,
                                     int64_t M, float* scales, float* row_scales, void* stream) {
    DINO_CHECK_ARG(keep && scales && L > 0 && Bct > 0);
    DINO_CHECK_ARG(!row_scales || (row_crop && M > 0));
    const int64_t total = (int64_t)L * 2 * ((int64_t)Bct + (row_scales ? M : 0));
    hipLaunchKernelGGL(drop_path_scales_kernel, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, seed, keep, L,
                       Bct, row_crop, M, scales, row_scales);
    return dino_launch_status();
}
