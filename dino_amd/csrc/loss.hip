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
#include <stdlib.h>

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

// column sums of the teacher logits (batch half of update_center).  Block = 64 column groups (8
// columns each) x 4 row phases; a thread sums CS_ROWS rows (rows y*4*CS_ROWS + phase, step 4) with all
// loads issued before the first add, the four phases meet in LDS, and one thread per column group adds
// the block's sums to `out`: rows / (4 * CS_ROWS) atomics per address instead of one per row slab.
constexpr int CS_ROWS = 16;
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* __restrict__ x,
                                                          float* __restrict__ out, int rows, int K) {
    __shared__ float part[3][64][9];
    const int cg = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cg) * 8;
    const int r0 = blockIdx.y * (4 * CS_ROWS) + ph;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < K) {
        float v[CS_ROWS][8];
#pragma unroll
        for (int i = 0; i < CS_ROWS; ++i) {
            const int r = r0 + 4 * i;
            if (r < rows) {
                load8_nt<T>(x + (int64_t)r * K + col, v[i]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < CS_ROWS; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[i][e];
    }
    if (ph > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[ph - 1][cg][e] = acc[e];
    }
    __syncthreads();
    if (ph == 0 && col < K) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            atomicAdd(out + col + e, acc[e] + part[0][cg][e] + part[1][cg][e] + part[2][cg][e]);
    }
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


// ------------------------------------------------------------------------------------------------
// bf16 fast path: ONE sweep over HBM.  A workgroup of 1024 threads owns image b (or a group of its
// crops when the batch is small): it turns the two teacher rows of the image into probabilities
// once and keeps them in registers (fp16 pairs scaled by 2^14: 64 VGPRs), then walks the student
// rows of that image.  A student row (<= 128 KiB of bf16) is fetched by LDS-DMA while the previous
// row is being worked on, moved LDS -> registers (32 VGPRs), and everything a row needs -- max,
// sum of exponentials, the dot with the teacher probabilities, the gradient -- is computed from
// those registers: the row crosses HBM once, the gradient row is written once, a teacher row is
// read once per crop group.  One exponential per element: exp(x - max) is kept (fp16, in place of
// the logits) between the normalisation pass and the gradient pass.
// ------------------------------------------------------------------------------------------------
#ifndef DINO_LOSS_FT
#define DINO_LOSS_FT 512
#endif
constexpr int FT = DINO_LOSS_FT;         // threads per workgroup (512: 8 waves, 2 per SIMD, <= 256 VGPRs each)
constexpr float QSCALE = 16384.f;        // teacher probabilities are stored as fp16(q * 2^14)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// sum / max over the workgroup of two values at once; every thread gets the result.  red: 2 x 32
// floats used alternately (`phase` flips per call), so ONE barrier per reduction suffices: a thread can
// only be writing half A again after every thread has passed the barrier of the reduction on half B,
// i.e. after all reads of A.  Raw s_barrier + lgkmcnt only: __syncthreads() would also wait for the
// LDS-DMA of the next row and for the gradient stores (vmcnt(0)) and serialise them with the arithmetic.
template <bool MAX>
DEVI void block_reduce2(float& a, float& b, float* red, int& phase) {
    if (MAX) { a = wave_max(a); b = wave_max(b); } else { a = wave_sum(a); b = wave_sum(b); }
    float* r = red + 32 * phase;
    phase ^= 1;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { r[w] = a; r[16 + w] = b; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float ra = r[0], rb = r[16];
#pragma unroll
    for (int i = 1; i < FT / 64; ++i) {
        if (MAX) { ra = fmaxf(ra, r[i]); rb = fmaxf(rb, r[16 + i]); }
        else { ra += r[i]; rb += r[16 + i]; }
    }
    a = ra; b = rb;
}

DEVI void unpack_bf16x8(const u32x4 q, float (&x)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        x[2 * r] = __builtin_bit_cast(float, q[r] << 16);
        x[2 * r + 1] = __builtin_bit_cast(float, q[r] & 0xffff0000u);
    }
}
DEVI void unpack_f16x8(const u32x4 q, float (&x)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned w = q[r];      // (bit_cast straight from the vector element reads element 0: clang)
        const f16x2 h = __builtin_bit_cast(f16x2, w);
        x[2 * r] = (float)h[0];
        x[2 * r + 1] = (float)h[1];
    }
}

// ONE: one workgroup per IMAGE (grid = B, chosen when B covers the chip): teacher view 1 -> global crop 0,
// teacher view 0 -> global crop 1, view 1 again (added: q0 + q1) -> the local crops.  A teacher row is then
// read 1.5 times instead of twice, its softmax runs 3 times per image instead of 4, the grid is one even
// round, and every teacher phase has the next student row's DMA in flight beside it.
// NT: the student rows (read exactly once by the whole grid) arrive with the non-temporal hint.

// ---- hand-picked instructions for the row passes.  The student-row loop is VALU-bound (timing-only build without
// the gradient stores: 162 of 193 us), so its instruction count IS its speed: hipcc spends ~13.7 VALU
// instructions per logit (separate f16 -> f32 conversions, canonicalising v_max before every fmaxf, scalar
// fma / add where the packed-f32 forms do two lanes' worth per issue); these forms need ~8.
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVI float max3_raw(float a, float b, float c) {           // no canonicalisation, NaNs not expected
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DEVI f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DEVI f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
DEVI f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// f16 half of e times f32 b PLUS the same f16 half of q
template <bool HI> DEVI float mix_h_f_h(unsigned e, float b, unsigned q) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(e), "v"(b), "v"(q));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(e), "v"(b), "v"(q));
    return r;
}
DEVI float dot2_h(unsigned a, unsigned b, float c) {      // a.lo * b.lo + a.hi * b.hi + c on fp16 pairs
    float r;
    asm("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DEVI f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// f16 half (HI: upper) of word h, times f32 b, plus f32 c -- no separate conversion
template <bool HI> DEVI float mix_h_f_f(unsigned h, float b, float c) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b), "v"(c));
    return r;
}
// f16 half of e times f32 b MINUS the same f16 half of q
template <bool HI> DEVI float mix_h_f_mh(unsigned e, float b, unsigned q) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(e), "v"(b), "v"(q));
    else asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(e), "v"(b), "v"(q));
    return r;
}
DEVI float exp2_raw(float x) {
    float r;
    asm("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
DEVI unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <int NCH, bool ONE, bool NT>   // 16-byte chunks per thread: K == FT * 8 * NCH
__global__ __launch_bounds__(FT) void loss_fused_bf16_kernel(
    const bf16* s, const bf16* t, const float* center, bf16* ds, float* loss_parts, int B, int ncrops,
    int K, int G, float inv_st, float inv_tt, float coef) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    float* red = (float*)(fsm + FT * NCH * 16);       // behind the row image
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups [0, B*G): image b = id % B, local crops 2 + g, 2 + g + G, ... (g = id / B), scored
    // against BOTH teacher views -> one probability array q0 + q1.  Workgroups [B*G, B*G + 2B): one
    // global crop v of image b, scored against the OTHER view only.  Either way a workgroup holds ONE
    // probability array (a second one does not fit the register file next to a 64 K row).
    const int nloc = B * G;
    const bool locals = ONE || (int)blockIdx.x < nloc;
    const int id = locals ? blockIdx.x : blockIdx.x - nloc;
    const int b = ONE ? (int)blockIdx.x : id % B, g = ONE ? 0 : id / B;   // globals: g = crop index (0 or 1)
    int phase = 0;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    // Every global access is a buffer operation: per-thread byte offset (tid * 16, one VGPR) + a scalar
    // offset for (row, chunk).  Registers are the scarce resource: a spilled one costs far more than its
    // reload, because scratch reloads wait on vmcnt and so drain the row prefetch and the gradient stores.
    const int vb16 = tid * 16, vb32 = tid * 32;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)s, 0, (int)((int64_t)ncrops * B * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
        (void*)ds, 0, (int)((int64_t)ncrops * B * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(
        (void*)t, 0, (int)((int64_t)2 * B * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)center, 0, K * 4, 0x00020000);
    auto issue_row = [&](int v) {
        const int so = (int)(((int64_t)v * B + b) * K * 2);
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lptr_t)(fsm + (j * FT + wv * 64) * 16), 16, vb16,
                                                     so + j * FT * 16, 0, NT ? 2 : 0);
    };
    const int v0 = ONE ? 0 : locals ? 2 + g : g, vstep = ONE ? 1 : locals ? G : ncrops;   // my crops: v0, v0 + vstep, ... < ncrops
    // (the first student row starts streaming into LDS right behind the first teacher row's loads and lands under
    // the teacher phase)

    // ---- teacher probabilities, fp16(q * 2^14) pairs in registers ---------------------------------
    // Per teacher row three sweeps over registers: max, sum of exponentials, probabilities.  The row stays
    // packed (bf16) in registers; the centre chunk is re-read for each sweep (256 KiB shared by every workgroup: L2 hits) so that no fp32
    // copy of the row has to be held.  (`vo` / tpk are made opaque between the sweeps: otherwise the
    // optimiser merges the sweeps' loads and unpacking and keeps the fp32 centred logits, 8 registers per
    // chunk, across the reductions.)
    u32x4 q[NCH];
    float qtot = 0.f;                                 // sum of the ROUNDED probabilities (* 2^14)
    const float a_t = inv_tt * LOG2E;
    auto center8 = [&](int vo, int j, float (&cv)[8]) {
        const f32x4 c0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsC, vo, j * FT * 32, 0));
        const f32x4 c1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsC, vo, j * FT * 32 + 16, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) { cv[e] = c0[e]; cv[4 + e] = c1[e]; }
    };
    // first_issue: the kernel's first student row is requested BEHIND this teacher row's loads (vmcnt retires in
    // order: the other way round the softmax below would wait for the 128 KiB row as well)
    auto teacher_row = [&](int r, bool accumulate, bool first_issue) __attribute__((always_inline)) {
        const int tso = (int)(((int64_t)r * B + b) * K * 2);
        u32x4 tpk[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) tpk[j] = __builtin_amdgcn_raw_buffer_load_b128(rsT, vb16, tso + j * FT * 16, 0);
        if (first_issue && v0 < ncrops) issue_row(v0);
        // sweep 1: max of (t - c)
        float mxp[2] = {-INFINITY, -INFINITY}, dummy = -INFINITY;
        int vo = vb32;
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float cv[8];
            center8(vo, j, cv);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned w = tpk[j][p];
                const f32x2 tv = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
                const f32x2 d = pk_sub(tv, f32x2{cv[2 * p], cv[2 * p + 1]});
                mxp[p & 1] = max3_raw(mxp[p & 1], d[0], d[1]);
            }
            if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);   // (bounds the centre loads in flight when NCH > 16)
        }
        float mx = fmaxf(mxp[0], mxp[1]) * a_t;
        block_reduce2<true>(mx, dummy, red, phase);
#ifdef LOSS_ABLATE_TEACHER
        // (timing-only build: no exponentials in the teacher phase; q is the packed row itself)
#pragma unroll
        for (int j = 0; j < NCH; ++j) q[j] = tpk[j];
        qtot = mx;
        return;
#endif
        // sweep 2: E = exp((t - c)/tau - max) * 2^14 replaces the logits in place (fp16 pairs), and its sum.
        // v_cvt_pkrtz truncates; the factor 1 + 2^-11 folded into the exponent turns that into rounding to
        // nearest up to ties (error within +-2^-11 relative, no bias).
        constexpr float RN = 14.000704269011246f;          // log2(2^14 * (1 + 2^-11))
        const f32x2 at2 = {a_t, a_t}, off2 = {RN - mx, RN - mx};
        f32x2 sum2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(tpk[j]));
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float cv[8];
            center8(vo, j, cv);
            u32x4 ev;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned w = tpk[j][p];
                const f32x2 tv = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
                const f32x2 y = pk_fma(pk_sub(tv, f32x2{cv[2 * p], cv[2 * p + 1]}), at2, off2);
                const f32x2 e = {exp2_raw(y[0]), exp2_raw(y[1])};
                ev[p] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e[0], e[1]));
                sum2[p & 1] = pk_add(sum2[p & 1], e);      // (carries the rounding factor: divided out in `sc`)
            }
            tpk[j] = ev;
            if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);
        }
        float sum = (sum2[0][0] + sum2[0][1]) + (sum2[1][0] + sum2[1][1]), z = 0.f;
        block_reduce2<false>(sum, z, red, phase);
        // sweep 3: q (* 2^14) = E * 2^14 / sum (+ the other view's q), rounded to fp16; `tot` = what the rounded
        // values really add up to
        const float sc = QSCALE * (1.f + 0x1p-11f) / sum;
        float tot2[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(tpk[j]));
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            u32x4 qv;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned we = tpk[j][p], wq = accumulate ? q[j][p] : 0u;
                const f16x2 h = {(_Float16)mix_h_f_h<false>(we, sc, wq), (_Float16)mix_h_f_h<true>(we, sc, wq)};   // RNE
                qv[p] = __builtin_bit_cast(unsigned, h);
                tot2[0] = mix_h_f_f<false>(qv[p], 1.f, tot2[0]);
                tot2[1] = mix_h_f_f<true>(qv[p], 1.f, tot2[1]);
            }
            q[j] = qv;
        }
        float tot = tot2[0] + tot2[1];
        z = 0.f;
        block_reduce2<false>(tot, z, red, phase);
        qtot = tot;                                   // (accumulate: the sum of the rounded q0 + q1)
    };
    // ---- student rows ---------------------------------------------------------------------------
    const float a_s = inv_st * LOG2E;
    const float gscale = inv_st * coef;
    // A wave's DMA pieces of a row were issued BEFORE the gradient stores of the previous row: waiting
    // for "at most NCH operations outstanding" retires the DMA and leaves those stores draining under
    // this row's arithmetic.
    // one student row: v (already streaming into LDS), vn = the row to prefetch behind it (< 0: none);
    // drained: nothing but this row's DMA may be outstanding (first row / first row after a teacher phase)
    auto student_row = [&](int v, int vn, float nv, bool drained) __attribute__((always_inline)) {
        u32x4 cur[NCH];
        if (drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");
        __builtin_amdgcn_s_barrier();                       // the whole row has landed in LDS
#pragma unroll
        for (int j = 0; j < NCH; ++j) cur[j] = *(const u32x4*)(fsm + j * FT * 16 + vb16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // ... and has left it again
        if (vn >= 0) issue_row(vn);                         // next row streams in under this one's work
        // pass 1: row maximum.  The upper bf16 of a word is compared as the raw word (its value plus < 1 bf16 ulp
        // from the other half's bits): the "maximum" may sit a fraction of an ulp off, which softmax, invariant
        // to the shift, does not see -- one shift + one max3 per pair.
        float mxp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, dummy = -INFINITY;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned w = cur[j][p];
                mxp[p] = max3_raw(mxp[p], __builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w));
            }
        }
        float mx = max3_raw(mxp[0], mxp[1], fmaxf(mxp[2], mxp[3]));
        block_reduce2<true>(mx, dummy, red, phase);
#pragma unroll
        for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(cur[j]));
        const float m2 = mx * a_s;
        // pass 2a: e = exp(x/tau - max) (kept, fp16, in place of the logits), its sum, and the dot of the
        // logits with the teacher probabilities (fp16 halves enter the fma directly: v_fma_mix)
        const f32x2 as2 = {a_s, a_s}, nm2 = {-m2, -m2};
        f32x2 sum2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        float dotp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            u32x4 qa = q[j], cx = cur[j], ep;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned wa = qa[p], wx = cx[p];
                const f32x2 x = {__builtin_bit_cast(float, wx << 16), __builtin_bit_cast(float, wx & 0xffff0000u)};
                const f32x2 y = pk_fma(x, as2, nm2);
                const f32x2 e = {exp2_raw(y[0]), exp2_raw(y[1])};
                dotp[2 * (p & 1)] = mix_h_f_f<false>(wa, x[0], dotp[2 * (p & 1)]);
                dotp[2 * (p & 1) + 1] = mix_h_f_f<true>(wa, x[1], dotp[2 * (p & 1) + 1]);
                sum2[p & 1] = pk_add(sum2[p & 1], e);
                ep[p] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e[0], e[1]));
            }
            cur[j] = ep;
            __builtin_amdgcn_sched_barrier(0);   // chunk by chunk: interleaving all chunks costs registers
        }
        float sum = (sum2[0][0] + sum2[0][1]) + (sum2[1][0] + sum2[1][1]);
        float dot = (dotp[0] + dotp[1]) + (dotp[2] + dotp[3]);
        block_reduce2<false>(sum, dot, red, phase);
        const float lse = (m2 + __builtin_amdgcn_logf(sum)) * LN2;      // natural-log lse of x / tau_s
        const int row = v * B + b;
#ifndef LOSS_DBG
        if (tid == 0) loss_parts[row] = coef * (lse * qtot - inv_st * dot) * (1.f / QSCALE);
#else   // (diagnostic build, tools/loss_debug.py: one ingredient of the row's loss instead of the loss)
        if (tid == 0) loss_parts[row] = LOSS_DBG == 1 ? lse : LOSS_DBG == 2 ? inv_st * dot / QSCALE : LOSS_DBG == 3 ? qtot / QSCALE : mx;
#endif
        // pass 2b: d loss / d s = (n_v * softmax - qs) / (tau_s B T) = cq * (e * (ce / cq) - q * 2^14)
        const float cq = gscale / QSCALE, cr = nv * QSCALE / sum;
        const f32x2 cq2 = {cq, cq};
        const int dso = (int)((int64_t)row * K * 2);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            u32x4 qa = q[j], cx = cur[j], gp;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned wa = qa[p], we = cx[p];
                const f32x2 t = {mix_h_f_mh<false>(we, cr, wa), mix_h_f_mh<true>(we, cr, wa)};
                const f32x2 g2 = pk_mul(t, cq2);
                gp[p] = cvt_pk_bf16(g2[0], g2[1]);
            }
#ifndef LOSS_ABLATE_STORE
            __builtin_amdgcn_raw_buffer_store_b128(gp, rsD, vb16, dso + j * FT * 16, 2 /* non-temporal */);
#else
            asm volatile("" ::"v"(gp));       // (timing-only build: the gradient is computed and dropped)
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (ONE) {
        teacher_row(1, false, true);                  // global crop 0 against view 1
        student_row(0, 1, 1.f, true);
        teacher_row(0, false, false);                 // global crop 1 against view 0
        student_row(1, ncrops > 2 ? 2 : -1, 1.f, true);
        if (ncrops > 2) {
            teacher_row(1, true, false);              // the local crops against both: q0 + q1
            bool drained = true;
            for (int v = 2; v < ncrops; ++v) {
                student_row(v, v + 1 < ncrops ? v + 1 : -1, 2.f, drained);
                drained = false;
            }
        }
    } else {
        if (locals) {
            teacher_row(0, false, true);
            teacher_row(1, true, false);
        } else {
            teacher_row(1 - g, false, true);          // global crop 0 against view 1, crop 1 against view 0
        }
        const float nv = locals ? 2.f : 1.f;
        bool drained = true;
        for (int v = v0; v < ncrops; v += vstep) {
            student_row(v, v + vstep < ncrops ? v + vstep : -1, nv, drained);
            drained = false;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Prototype layer + loss without the [ncrops*B, K] logits: the caller walks
// the K prototypes in chunks -- logits of one chunk from the GEMM, these kernels, next chunk -- once for the
// loss (running log-sum-exp and teacher dot product per student row) and once more for the gradient
// (the chunk's logits recomputed by the GEMM and overwritten by d loss / d logits, which the two
// prototype-gradient GEMMs consume before the next chunk).  Same arithmetic as student_loss_kernel:
//   loss_row = nv * lse(x) - sum_k qs[k] * x[k],   x = s / tau_s,  qs = sum of the other views' teacher
//   probabilities,   d loss / d s[k] = (nv * exp(x[k] - lse) - qs[k]) / tau_s * coef.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proto_run_init_kernel(float* __restrict__ run, int R) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R) {
        run[2 * r] = -INFINITY;
        run[2 * r + 1] = 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(LT) void proto_chunk_stats_kernel(
    const T* __restrict__ s, int64_t lds, const T* __restrict__ t, int64_t ldt, const float* __restrict__ center,
    const float* __restrict__ lse_t, float* __restrict__ run, int B, int Kc, float inv_st, float inv_tt) {
    __shared__ float red[2 * LT / 64];
    const int row = blockIdx.x;
    const int v = row / B, b = row % B;
    const T* sr = s + (int64_t)row * lds;
    const bool use0 = v != 0, use1 = v != 1;
    const T* t0 = t + (int64_t)b * ldt;
    const T* t1 = t + (int64_t)(B + b) * ldt;
    const float l0 = lse_t[b], l1 = lse_t[B + b];
    MaxSum acc;
    acc.m = -INFINITY;
    acc.s = 0.f;
    float dot = 0.f;
    for (int k = threadIdx.x * 8; k < Kc; k += LT * 8) {
        float sv[8], cv[8], qs[8];
        load8<T>(sr + k, sv);
        load8<float>(center + k, cv);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sv[e] *= inv_st;
            mx = fmaxf(mx, sv[e]);
            qs[e] = 0.f;
        }
        const float m = fmaxf(acc.m, mx);
        float sum = acc.s * __expf(acc.m - m);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf(sv[e] - m);
        acc.m = m;
        acc.s = sum;
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
        for (int e = 0; e < 8; ++e) dot += qs[e] * sv[e];
    }
    const float lse_c = block_lse(acc, red);
    dot = block_sum(dot, red);
    if (threadIdx.x == 0) {
        const float a = run[2 * row], m = fmaxf(a, lse_c);   // logaddexp(running, chunk)
        run[2 * row] = m + __logf(__expf(a - m) + __expf(lse_c - m));
        run[2 * row + 1] += dot;
    }
}

__global__ __launch_bounds__(256) void proto_loss_parts_kernel(const float* __restrict__ run, float* __restrict__ parts,
                                                               int R, int B, float coef) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R) {
        const float nv = (r / B) < 2 ? 1.f : 2.f;
        parts[r] = (nv * run[2 * r] - run[2 * r + 1]) * coef;
    }
}

template <typename T>
__global__ __launch_bounds__(LT) void proto_chunk_grad_kernel(
    T* __restrict__ s, int64_t lds, const T* __restrict__ t, int64_t ldt, const float* __restrict__ center,
    const float* __restrict__ lse_t, const float* __restrict__ run, int B, int Kc, float inv_st, float inv_tt,
    float coef, const float* __restrict__ gout) {
    const int row = blockIdx.x;
    const int v = row / B, b = row % B;
    T* sr = s + (int64_t)row * lds;
    const bool use0 = v != 0, use1 = v != 1;
    const float nv = (use0 ? 1.f : 0.f) + (use1 ? 1.f : 0.f);
    const T* t0 = t + (int64_t)b * ldt;
    const T* t1 = t + (int64_t)(B + b) * ldt;
    const float l0 = lse_t[b], l1 = lse_t[B + b], lse = run[2 * row];
    const float gscale = inv_st * coef * (gout ? gout[0] : 1.f);
    for (int k = threadIdx.x * 8; k < Kc; k += LT * 8) {
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
        for (int e = 0; e < 8; ++e) g[e] = (nv * __expf(sv[e] * inv_st - lse) - qs[e]) * gscale;
        store8<T>(sr + k, g);
    }
}

}  // namespace

// the fused kernel is instantiated for K = 4096 * {1, 2, 4, 8, 16} (FT = 512)
static bool fused_shape(int K) {
    if (K % (FT * 8)) return false;
    const int n = K / (FT * 8);
    return K <= 65536 && (n == 1 || n == 2 || n == 4 || n == 8 || n == 16);
}

extern "C" int dino_loss_fwd_bwd(int dtype, const void* student, const void* teacher,
                                 const float* center, void* dstudent, float* loss_parts,
                                 float* center_acc, float* row_stats, int B, int ncrops, int K,
                                 float inv_student_temp, float inv_teacher_temp, void* stream) {
    DINO_CHECK_ARG(student && teacher && center && dstudent && loss_parts && row_stats);
    DINO_CHECK_ARG(B > 0 && ncrops >= 2 && K > 0 && K % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    const float coef = 1.f / ((float)B * (float)(2 * ncrops - 2));
    dim3 cgrid((K / 8 + 63) / 64, (2 * B + 4 * CS_ROWS - 1) / (4 * CS_ROWS));
    static const bool fused_off = getenv("DINO_LOSS_FUSED") && getenv("DINO_LOSS_FUSED")[0] == '0';
    if (dtype == DINO_BF16 && !fused_off && fused_shape(K) &&
        (int64_t)ncrops * B * K * 2 < ((int64_t)1 << 31)) {
        // local-crop groups per image: enough workgroups to cover the chip when the batch is small
        const int nloc_crops = ncrops - 2;
        int G = nloc_crops > 0 ? (256 + B - 1) / B : 0;
        if (G > nloc_crops) G = nloc_crops;
        while (G > 0 && G < nloc_crops && nloc_crops % G) ++G;
        const int nch = K / (FT * 8);
        const int smem = FT * nch * 16 + 64 * 4;
        // one workgroup per image once the batch alone covers the chip (DINO_LOSS_ONE=0/1 forces either form)
        static const char* one_env = getenv("DINO_LOSS_ONE");
        static const char* nt_env = getenv("DINO_LOSS_NT");
        const bool one = one_env ? one_env[0] != '0' : B >= dino_usable_cus(256);
        const bool nt = nt_env ? nt_env[0] != '0' : true;
#define LOSS_FUSED_LAUNCH_V(N, ONE, NT)                                                                  \
    {                                                                                                    \
        static bool attr_set = false; /* (idempotent; a race sets it twice) */                           \
        if (!attr_set) {                                                                                 \
            hipError_t e = hipFuncSetAttribute((const void*)loss_fused_bf16_kernel<N, ONE, NT>,          \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);        \
            if (e != hipSuccess) return (int)e;                                                          \
            attr_set = true;                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((loss_fused_bf16_kernel<N, ONE, NT>), dim3(ONE ? B : B * G + 2 * B), dim3(FT), smem, st, \
                           (const bf16*)student, (const bf16*)teacher, center, (bf16*)dstudent,          \
                           loss_parts, B, ncrops, K, G, inv_student_temp, inv_teacher_temp, coef);       \
    }
#define LOSS_FUSED_LAUNCH(N)                                                                             \
    {                                                                                                    \
        if (one && nt) LOSS_FUSED_LAUNCH_V(N, true, true)                                                \
        else if (one) LOSS_FUSED_LAUNCH_V(N, true, false)                                                \
        else if (nt) LOSS_FUSED_LAUNCH_V(N, false, true)                                                 \
        else LOSS_FUSED_LAUNCH_V(N, false, false)                                                        \
    }
        switch (nch) {
            case 1: LOSS_FUSED_LAUNCH(1) break;
            case 2: LOSS_FUSED_LAUNCH(2) break;
            case 4: LOSS_FUSED_LAUNCH(4) break;
            case 8: LOSS_FUSED_LAUNCH(8) break;
            default:
                if constexpr (FT * 8 * 16 <= 65536) LOSS_FUSED_LAUNCH(16)
                break;
        }
#undef LOSS_FUSED_LAUNCH
#undef LOSS_FUSED_LAUNCH_V
        if (center_acc)
            hipLaunchKernelGGL(colsum_rows_kernel<bf16>, cgrid, dim3(256), 0, st, (const bf16*)teacher,
                               center_acc, 2 * B, K);
    } else if (dtype == DINO_BF16) {
        typedef bf16 T;
        hipLaunchKernelGGL(teacher_stats_kernel<T>, dim3(2 * B), dim3(LT), 0, st, (const T*)teacher,
                           center, row_stats, K, inv_teacher_temp);
        hipLaunchKernelGGL(student_loss_kernel<T>, dim3(ncrops * B), dim3(LT), 0, st,
                           (const T*)student, (const T*)teacher, center, row_stats, (T*)dstudent,
                           loss_parts, B, K, inv_student_temp, inv_teacher_temp, coef);
        if (center_acc)
            hipLaunchKernelGGL(colsum_rows_kernel<T>, cgrid, dim3(256), 0, st, (const T*)teacher,
                               center_acc, 2 * B, K);
    } else if (dtype == DINO_F32) {
        typedef float T;
        hipLaunchKernelGGL(teacher_stats_kernel<T>, dim3(2 * B), dim3(LT), 0, st, (const T*)teacher,
                           center, row_stats, K, inv_teacher_temp);
        hipLaunchKernelGGL(student_loss_kernel<T>, dim3(ncrops * B), dim3(LT), 0, st,
                           (const T*)student, (const T*)teacher, center, row_stats, (T*)dstudent,
                           loss_parts, B, K, inv_student_temp, inv_teacher_temp, coef);
        if (center_acc)
            hipLaunchKernelGGL(colsum_rows_kernel<T>, cgrid, dim3(256), 0, st, (const T*)teacher,
                               center_acc, 2 * B, K);
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

extern "C" int dino_proto_loss_chunk(int dtype, int phase, void* s_chunk, int64_t lds, const void* teacher,
                                     int64_t ldt, const float* center, float* teacher_lse, float* run, int B,
                                     int ncrops, int K, int k0, int Kc, float inv_student_temp,
                                     float inv_teacher_temp, const float* gout, float* loss_parts, void* stream) {
    DINO_CHECK_ARG(dtype == DINO_BF16 || dtype == DINO_F32);
    DINO_CHECK_ARG(teacher_lse && run && B > 0 && ncrops >= 2 && K > 0 && K % 8 == 0 && phase >= 0 && phase <= 3);
    hipStream_t st = (hipStream_t)stream;
    const int R = ncrops * B;
    const float coef = 1.f / ((float)B * (float)(2 * ncrops - 2));
    if (phase == 0) {          // teacher log-sum-exp per row, running student statistics cleared
        DINO_CHECK_ARG(teacher && center && ldt == K);
        if (dtype == DINO_BF16)
            hipLaunchKernelGGL(teacher_stats_kernel<bf16>, dim3(2 * B), dim3(LT), 0, st, (const bf16*)teacher, center,
                               teacher_lse, K, inv_teacher_temp);
        else
            hipLaunchKernelGGL(teacher_stats_kernel<float>, dim3(2 * B), dim3(LT), 0, st, (const float*)teacher,
                               center, teacher_lse, K, inv_teacher_temp);
        hipLaunchKernelGGL(proto_run_init_kernel, dim3((R + 255) / 256), dim3(256), 0, st, run, R);
        return dino_launch_status();
    }
    if (phase == 2) {          // per-row loss terms (folded by dino_center_update)
        DINO_CHECK_ARG(loss_parts);
        hipLaunchKernelGGL(proto_loss_parts_kernel, dim3((R + 255) / 256), dim3(256), 0, st, run, loss_parts, R, B,
                           coef);
        return dino_launch_status();
    }
    DINO_CHECK_ARG(s_chunk && teacher && center && k0 >= 0 && Kc > 0 && Kc % 8 == 0 && k0 % 8 == 0 && k0 + Kc <= K);
    DINO_CHECK_ARG(lds % 8 == 0 && lds >= Kc && ldt % 8 == 0 && ldt >= K);
#define PROTO_CHUNK(T)                                                                                            \
    if (phase == 1)                                                                                               \
        hipLaunchKernelGGL(proto_chunk_stats_kernel<T>, dim3(R), dim3(LT), 0, st, (const T*)s_chunk, lds,         \
                           (const T*)teacher + k0, ldt, center + k0, teacher_lse, run, B, Kc, inv_student_temp,   \
                           inv_teacher_temp);                                                                     \
    else                                                                                                          \
        hipLaunchKernelGGL(proto_chunk_grad_kernel<T>, dim3(R), dim3(LT), 0, st, (T*)s_chunk, lds,                \
                           (const T*)teacher + k0, ldt, center + k0, teacher_lse, run, B, Kc, inv_student_temp,   \
                           inv_teacher_temp, coef, gout);
    if (dtype == DINO_BF16) { PROTO_CHUNK(bf16) } else { PROTO_CHUNK(float) }
#undef PROTO_CHUNK
    return dino_launch_status();
}
