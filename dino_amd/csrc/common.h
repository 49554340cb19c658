// Shared device helpers for the dino_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dino_hip.h"
#include <stdlib.h>

// CUs that the one-workgroup-per-CU (persistent) kernels leave free: DINO_CU_MARGIN, default 0.  With several
// ranks per node the RCCL kernels of the gradient all-reduce need CUs of their own while a persistent kernel is
// resident; a workgroup of such a kernel that finds its CU taken starts a whole kernel late.  Set it to the number
// of RCCL channels (NCCL_MAX_NCHANNELS) to keep the two apart.
static inline int dino_usable_cus(int ncu) {
    static const int margin = getenv("DINO_CU_MARGIN") ? atoi(getenv("DINO_CU_MARGIN")) : 0;
    const int n = ncu - (margin > 0 ? margin : 0);
    return n < 32 ? 32 : n;
}

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64
#define DEVI __device__ __forceinline__

#define DINO_CHECK_ARG(cond) \
    do {                     \
        if (!(cond)) return DINO_ERR_BADARG; \
    } while (0)

static inline int dino_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DINO_OK : (int)e;
}

// ---------------------------------------------------------------------------
// scalar conversions
// ---------------------------------------------------------------------------
template <typename T> DEVI float to_f32(T x);
template <> DEVI float to_f32<float>(float x) { return x; }
template <> DEVI float to_f32<bf16>(bf16 x) { return (float)x; }
template <typename T> DEVI T from_f32(float x);
template <> DEVI float from_f32<float>(float x) { return x; }
template <> DEVI bf16 from_f32<bf16>(float x) { return (bf16)x; }  // v_cvt_pk_bf16_f32: RNE, NaN-safe

// ---------------------------------------------------------------------------
// 8-element vector access (16 B for bf16, 2 x 16 B for f32); p must be 16-B aligned
// ---------------------------------------------------------------------------
template <typename T> DEVI void load8(const T* p, float (&v)[8]);
template <> DEVI void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <> DEVI void load8<bf16>(const bf16* p, float (&v)[8]) {
    bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <typename T> DEVI void store8(T* p, const float (&v)[8]);
template <> DEVI void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *(f32x4*)p = a;
    *(f32x4*)(p + 4) = b;
}
template <> DEVI void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
    *(bf16x8*)p = a;
}
// the same with the non-temporal hint: streams that this kernel never re-reads
template <typename T> DEVI void load8_nt(const T* p, float (&v)[8]);
template <> DEVI void load8_nt<float>(const float* p, float (&v)[8]) {
    f32x4 a = __builtin_nontemporal_load((const f32x4*)p), b = __builtin_nontemporal_load((const f32x4*)(p + 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <> DEVI void load8_nt<bf16>(const bf16* p, float (&v)[8]) {
    bf16x8 a = __builtin_nontemporal_load((const bf16x8*)p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <typename T> DEVI void store8_nt(T* p, const float (&v)[8]);
template <> DEVI void store8_nt<float>(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    __builtin_nontemporal_store(a, (f32x4*)p);
    __builtin_nontemporal_store(b, (f32x4*)(p + 4));
}
template <> DEVI void store8_nt<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
    __builtin_nontemporal_store(a, (bf16x8*)p);
}

// 4-element vector access (8 B for bf16, 16 B for f32)
template <typename T> DEVI void load4(const T* p, float (&v)[4]);
template <> DEVI void load4<float>(const float* p, float (&v)[4]) {
    f32x4 a = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
}
template <> DEVI void load4<bf16>(const bf16* p, float (&v)[4]) {
    bf16x4 a = *(const bf16x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
template <typename T> DEVI void store4(T* p, const float (&v)[4]);
template <> DEVI void store4<float>(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    *(f32x4*)p = a;
}
template <> DEVI void store4<bf16>(bf16* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16)v[i];
    *(bf16x4*)p = a;
}

// ---------------------------------------------------------------------------
// wave (64-lane) and block reductions
// ---------------------------------------------------------------------------
// Cross-lane reductions on the DPP path (VALU operand swizzles: quad_perm, row_half_mirror,
// row_mirror) instead of ds_bpermute, which occupies the LDS pipeline the GEMM loaders need.
template <int CTRL> DEVI float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// sum / max over each aligned group of 16 lanes; every lane of the group gets the result
DEVI float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}
DEVI float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
DEVI float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
DEVI float wave_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
// Block-wide sum; `red` is LDS scratch of >= blockDim.x/64 floats. All threads get the result.
DEVI float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
DEVI float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

// W-wide (4 or 8) vector access
template <typename T, int W> DEVI void loadw(const T* p, float (&v)[W]) {
    if constexpr (W == 8) load8<T>(p, v); else load4<T>(p, v);
}
template <typename T, int W> DEVI void storew(T* p, const float (&v)[W]) {
    if constexpr (W == 8) store8<T>(p, v); else store4<T>(p, v);
}

// exact-erf GELU (nn.GELU() default) and its derivative
DEVI float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
DEVI float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16 path: the normal CDF by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 on erf, far below bf16
// resolution) on the raw v_rcp_f32 / v_exp_f32 units: with z = |x|/sqrt(2) and s = z*sqrt(log2 e),
//   h = 0.5*erfc(z) = 0.5 * poly(t) * 2^(-s^2),  t = 1/(1 + p*z),  Phi(x) = x < 0 ? h : 1 - h,
// and 2^(-s^2) = exp(-x^2/2) is what the derivative needs.  ~17 VALU ops for gelu AND gelu'
// (libm erff + a correctly rounded division were ~40).
DEVI float gelu_cdf_fast(float x, float& expmh) {
    const float s = fabsf(x) * 0.84932180028801904272f;                 // |x| * sqrt(log2(e)/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.27273748088f, s, 1.f));   // p / sqrt(log2 e) folded
    expmh = __builtin_amdgcn_exp2f(-s * s);
    const float poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f +
                       t * (-0.7265760135f + t * 0.5307027145f))));
    const float h = poly * expmh;
    return x < 0.f ? h : 1.f - h;
}
template <typename T> DEVI float gelu_t(float x);
template <> DEVI float gelu_t<float>(float x) { return gelu_f(x); }
template <> DEVI float gelu_t<bf16>(float x) {
    float ex;
    return x * gelu_cdf_fast(x, ex);
}
// gelu(x) and gelu'(x) from one erf / exp evaluation (forward epilogue that saves the derivative)
template <typename T> DEVI void gelu_both_t(float x, float& gl, float& dg);
template <> DEVI void gelu_both_t<float>(float x, float& gl, float& dg) {
    gl = gelu_f(x);
    dg = gelu_grad_f(x);
}
template <> DEVI void gelu_both_t<bf16>(float x, float& gl, float& dg) {
    float ex;
    const float cdf = gelu_cdf_fast(x, ex);
    gl = x * cdf;
    dg = fmaf(x * 0.39894228040143267794f, ex, cdf);
}
template <typename T> DEVI float gelu_grad_t(float x);
template <> DEVI float gelu_grad_t<float>(float x) { return gelu_grad_f(x); }
template <> DEVI float gelu_grad_t<bf16>(float x) {
    float ex;
    const float cdf = gelu_cdf_fast(x, ex);
    return fmaf(x * 0.39894228040143267794f, ex, cdf);
}

// XCD-aware bijective block remap (8 XCDs, round-robin dispatch): consecutive logical ids
// share an XCD's L2.  Speed only -- never correctness.
DEVI int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
