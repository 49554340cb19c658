// Fused multi-head attention (head_dim 64) forward and backward for short ViT sequences.
//
// A "problem" is one (crop, head).  Its K/V (forward, dQ) or Q/dO (dK/dV) rows are staged
// once into LDS, row-major with a padded row stride; every wave then owns 16-row tiles of the
// other side and keeps them in registers.  All products are written in the orientation whose
// accumulator (col = owner row on the lane, 4 consecutive contraction-side rows in registers)
// is directly the SECOND operand of the next MFMA, so probabilities never leave registers:
//   fwd : S^T = K Q^T  -> online softmax -> O^T += V^T P^T
//   dQ  : S^T, dP^T = V dO^T -> dS^T -> dQ^T += K^T dS^T
//   dKV : S = Q K^T, dP = dO V^T -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS
// "Transposed" first operands (V^T, K^T, dO^T, Q^T) are read from the row-major LDS image
// with ds_read_b64_tr_b16 (bf16) or scalar ds_read_b32 (f32: v_mfma_f32_16x16x4_f32 takes one
// element per lane).  The contraction order inside an MFMA is permuted identically for both
// operands (accumulator-as-operand rule), which leaves the sum unchanged.
// PW = waves per problem: 4 for N > 64 (one problem per workgroup), 1 for N <= 64 (four
// problems per workgroup, e.g. the 37-token local crops).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int HD = 64;

template <typename T> struct AT;
// LDS row images.  bf16: 128-B rows (no padding: three 197-token problems fit one CU), 16-B chunk
// index XOR key(row) = row & 6 -- conflict-free both for the ds_read_b128 row fragments (rows
// i = 0..15, chunks g / g+4) and for the ds_read_b64_tr_b16 column fragments (8 rows x 32 B per
// half-wave).  f32 (parity path): 256-B rows padded by 16 B, no swizzle.
template <> struct AT<bf16> {
    static constexpr int EPC = 8, NH = 2, TPS = 2, ROWB = 128;
    static DEVI int key(int row) { return row & 6; }
};
template <> struct AT<float> {
    static constexpr int EPC = 4, NH = 4, TPS = 1, ROWB = 272;
    static DEVI int key(int) { return 0; }
};

DEVI void mma(const u32x4& x, const u32x4& y, f32x4& acc, bf16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x),
                                                  __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0);
}
DEVI void mma(const u32x4& x, const u32x4& y, f32x4& acc, float) {
    const f32x4 xf = __builtin_bit_cast(f32x4, x), yf = __builtin_bit_cast(f32x4, y);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[e], yf[e], acc, 0, 0, 0);
}

// row fragment: lane (i, g) reads chunk c = g + 4h of row (r0 + i)
template <typename T> DEVI u32x4 row_frag(const char* base, int r0, int i, int c) {
    return *(const u32x4*)(base + (r0 + i) * AT<T>::ROWB + ((c ^ AT<T>::key(r0 + i)) << 4));
}
// column fragment: lane (i, g) gets column (c0 + i) of the contraction-side rows of one step
DEVI u32x4 col_frag(const char* base, int r0, int c0, int i, int g, bf16) {
    const int q = i >> 2, p = i & 3;
    const int row = r0 + 4 * g + q, lc = (c0 + 4 * p) >> 3, off = (p & 1) * 8;
    // rows row and row + 16 have the same key (it looks at row bits 1-2 only)
    const char* a = base + row * AT<bf16>::ROWB + ((lc ^ AT<bf16>::key(row)) << 4) + off;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * AT<bf16>::ROWB));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}
// the same for a step that holds 16 contraction-side rows only: the second half reads as zeros
DEVI u32x4 col_frag_lo(const char* base, int r0, int c0, int i, int g) {
    const int q = i >> 2, p = i & 3;
    const int row = r0 + 4 * g + q, lc = (c0 + 4 * p) >> 3, off = (p & 1) * 8;
    const char* a = base + row * AT<bf16>::ROWB + ((lc ^ AT<bf16>::key(row)) << 4) + off;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo);
    return u32x4{l2[0], l2[1], 0u, 0u};
}
DEVI u32x4 col_frag(const char* base, int r0, int c0, int i, int g, float) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        r[e] = *(const float*)(base + (r0 + 4 * g + e) * AT<float>::ROWB + (c0 + i) * 4);
    return __builtin_bit_cast(u32x4, r);
}
// accumulator tile(s) of one step -> second-operand chunk
DEVI u32x4 pack(const f32x4 (&t)[2], bf16) {
    bf16x8 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = (bf16)t[0][r];
        v[4 + r] = (bf16)t[1][r];
    }
    return __builtin_bit_cast(u32x4, v);
}
DEVI u32x4 pack(const f32x4 (&t)[1], float) { return __builtin_bit_cast(u32x4, t[0]); }

// One 16 x 64 accumulator tile (lane (i, g): row i, columns dt * 16 + 4 g + r) times `sc`, written as a row of the
// output.  bf16: lanes g and g ^ 1 (16 lanes apart) first exchange halves -- v_permlane16_swap -- so that each lane
// holds 8 consecutive columns and the row leaves in 16-B pieces: 64-B row segments per store instruction instead of
// 32-B ones, and half the store instructions.  The exchange is cross-lane: EVERY lane of the wave must get here;
// `ok` predicates the stores only.  (The swap's operands come from the conversions, never straight from an MFMA.)
DEVI void swap16(unsigned& odd_rows_of_a, unsigned& even_rows_of_b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(odd_rows_of_a), "+v"(even_rows_of_b));
}
template <typename T> DEVI void store_acc_row(T* row, const f32x4 (&o)[4], float sc, int g, bool ok);
// the same with a (scalar) base and a 32-bit element offset per lane: no 64-bit address arithmetic in vector registers
template <typename T> DEVI void store_acc_row(T* base, unsigned eoff, const f32x4 (&o)[4], float sc, int g, bool ok) {
    static_assert(sizeof(T) == 2, "bf16 rows");
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        bf16x4 a, b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r] = (bf16)(o[2 * pr][r] * sc);
            b[r] = (bf16)(o[2 * pr + 1][r] * sc);
        }
        u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
        unsigned a0 = ua[0], a1 = ua[1], b0 = ub[0], b1 = ub[1];
        swap16(a0, b0);
        swap16(a1, b1);
        if (ok) *(u32x4*)(base + (eoff + (unsigned)((2 * pr + (g & 1)) * 16 + 4 * (g & 2)))) = u32x4{a0, a1, b0, b1};
    }
}
template <typename T> DEVI void store_acc_row(T* row, const f32x4 (&o)[4], float sc, int g, bool ok) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            bf16x4 a, b;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = (bf16)(o[2 * pr][r] * sc);
                b[r] = (bf16)(o[2 * pr + 1][r] * sc);
            }
            u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            unsigned a0 = ua[0], a1 = ua[1], b0 = ub[0], b1 = ub[1];
            // even g: keeps its columns of tile 2 pr and gets those of lane g + 1; odd g: gets tile 2 pr + 1's columns of
            // lane g - 1 and keeps its own -- either way {a, b} are 8 consecutive columns of tile 2 pr + (g & 1)
            swap16(a0, b0);
            swap16(a1, b1);
            if (ok) *(u32x4*)(row + (2 * pr + (g & 1)) * 16 + 4 * (g & 2)) = u32x4{a0, a1, b0, b1};
        }
    } else {
        if (ok) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                float v[4] = {o[dt][0] * sc, o[dt][1] * sc, o[dt][2] * sc, o[dt][3] * sc};
                store4<T>(row + dt * 16 + 4 * g, v);
            }
        }
    }
}

// Stage `nvalid` rows (64 elements each, row stride ld in global) into a padded LDS image of
// `npad` rows; rows >= nvalid are zero-filled.  Executed by a group of gsize threads.
template <typename T>
DEVI void stage_rows(char* dst, const T* src, int64_t ld, int nvalid, int npad, int gt, int gsize) {
    constexpr int NCH = HD / AT<T>::EPC;
    constexpr int BATCH = 8;  // loads in flight per thread: one exposed memory latency per batch
    const int total = npad * NCH;
#pragma unroll 1
    for (int q0 = gt; q0 < total; q0 += gsize * BATCH) {
        u32x4 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int q = q0 + u * gsize, r = q / NCH, c = q % NCH;
            v[u] = u32x4{0u, 0u, 0u, 0u};
            if (q < total && r < nvalid) v[u] = *(const u32x4*)(src + (int64_t)r * ld + c * AT<T>::EPC);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int q = q0 + u * gsize, r = q / NCH, c = q % NCH;
            if (q < total) *(u32x4*)(dst + r * AT<T>::ROWB + ((c ^ AT<T>::key(r)) << 4)) = v[u];
        }
    }
}

// Two images of the same shape at once: the loads of both are in flight before the first LDS write, so a
// workgroup pays one memory latency for its K and V (Q and dO) instead of two.
template <typename T>
DEVI void stage_rows_pair(char* dst0, const T* src0, int64_t ld0, char* dst1, const T* src1, int64_t ld1,
                          int nvalid, int npad, int gt, int gsize) {
    constexpr int NCH = HD / AT<T>::EPC;
    constexpr int BATCH = 8;
    const int total = npad * NCH;
#pragma unroll 1
    for (int q0 = gt; q0 < total; q0 += gsize * BATCH) {
        u32x4 v[BATCH], w[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int q = q0 + u * gsize, r = q / NCH, c = q % NCH;
            v[u] = w[u] = u32x4{0u, 0u, 0u, 0u};
            if (q < total && r < nvalid) {
                v[u] = *(const u32x4*)(src0 + (int64_t)r * ld0 + c * AT<T>::EPC);
                w[u] = *(const u32x4*)(src1 + (int64_t)r * ld1 + c * AT<T>::EPC);
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int q = q0 + u * gsize, r = q / NCH, c = q % NCH;
            if (q < total) {
                const int off = r * AT<T>::ROWB + ((c ^ AT<T>::key(r)) << 4);
                *(u32x4*)(dst0 + off) = v[u];
                *(u32x4*)(dst1 + off) = w[u];
            }
        }
    }
}

template <typename T> DEVI u32x4 gload_frag(const T* rowptr, int c, bool ok) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *(const u32x4*)(rowptr + c * AT<T>::EPC);
    return v;
}

template <typename T> DEVI float dot_chunk(const u32x4& a, const u32x4& b);
template <> DEVI float dot_chunk<bf16>(const u32x4& a, const u32x4& b) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, a), y = __builtin_bit_cast(bf16x8, b);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)x[e] * (float)y[e];
    return s;
}
template <> DEVI float dot_chunk<float>(const u32x4& a, const u32x4& b) {
    const f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
    return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
}

struct Prob {
    bool valid;
    int b, head, gt, gw;  // crop, head, thread id in group, wave id in group
    char* lds;
};
template <int PW> DEVI Prob locate(char* smem, int group_bytes, int nprob, int H) {
    Prob p;
    // (scalar: the problem a wave works on -- crop, head, LDS image, its query-tile loop -- is then uniform to the
    // compiler too: base addresses in SGPRs, scalar loop control, the row scale through the scalar cache)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / PW;
    const int pid = blockIdx.x * (4 / PW) + grp;
    p.valid = pid < nprob;
    const int pc = p.valid ? pid : 0;
    p.b = pc / H;
    p.head = pc % H;
    p.gt = threadIdx.x - grp * PW * 64;
    p.gw = wave - grp * PW;
    p.lds = smem + grp * group_bytes;
    return p;
}

// ------------------------------------------------------------------------------ forward
constexpr float LOG2E = 1.44269504088896340736f;

// Forward, all scores of a query tile in registers (NS = number of 16*TPS-key steps, compile time):
// S^T for every key tile, ONE max / exp2 / sum pass, then O^T += V^T P^T.  Against the online
// form this drops the per-step rescale of O, the running-max bookkeeping and all but the last
// step's key masks; scale and log2(e) ride in the exp2 argument's FMA.  The kernel was VALU-bound
// (51 M vector instructions per launch against 36 M MFMA-busy cycles at N = 197).
template <typename T, int PW, int NS>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                       float* __restrict__ lse, int Bc, int N, int H,
                                                       int npad, const float* __restrict__ rowscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using A = AT<T>;
    constexpr int SR = 16 * A::TPS;
    const int D = H * HD;
    const int64_t ld = 3 * D;
    const int tile_bytes = npad * A::ROWB;
    const Prob pr = locate<PW>(smem, 2 * tile_bytes, Bc * H, H);
    char* sK = pr.lds;
    char* sV = pr.lds + tile_bytes;
    const T* base = qkv + (int64_t)pr.b * N * ld + pr.head * HD;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    u32x4 qnext[A::NH];
    if (pr.valid) {
        // the first query tile's fragments travel with the K / V images: one exposed latency per workgroup
#pragma unroll
        for (int h = 0; h < A::NH; ++h)
            qnext[h] = gload_frag<T>(base + (int64_t)(pr.gw * 16 + i) * ld, g + 4 * h, pr.gw * 16 + i < N);
        stage_rows_pair<T>(sK, base + D, ld, sV, base + 2 * D, ld, N, npad, pr.gt, PW * 64);
    }
    __syncthreads();
    if (!pr.valid) return;
    const float c2 = 0.125f * LOG2E;  // head_dim ** -0.5, in the exp2 domain
    const float osc = rowscale ? rowscale[pr.b] : 1.f;
    const int nqt = (N + 15) / 16;
    for (int qt = pr.gw; qt < nqt; qt += PW) {
        const int q = qt * 16 + i;
        const bool qok = q < N;
        u32x4 qf[A::NH];
#pragma unroll
        for (int h = 0; h < A::NH; ++h) qf[h] = qnext[h];
        {
            const int qn = (qt + PW) * 16 + i;
#pragma unroll
            for (int h = 0; h < A::NH; ++h)
                qnext[h] = gload_frag<T>(base + (int64_t)qn * ld, g + 4 * h, qt + PW < nqt && qn < N);
        }
        f32x4 st[NS][A::TPS];
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                st[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                // a key tile that is padding only (keys 48-63 of a 37-token crop: a quarter of the work;
                // keys 208-223 of a 197-token one) is not computed: its scores are masked below anyway
                if (s == NS - 1 && t > 0 && s * SR + 16 * t >= N) continue;
#pragma unroll
                for (int h = 0; h < A::NH; ++h)
                    mma(row_frag<T>(sK, s * SR + 16 * t, i, g + 4 * h), qf[h], st[s][t], T());
            }
            if (s == NS - 1) {  // only the last step holds keys >= N
#pragma unroll
                for (int t = 0; t < A::TPS; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (s * SR + 16 * t + 4 * g + r >= N) st[s][t][r] = -INFINITY;
            }
#pragma unroll
            for (int t = 0; t < A::TPS; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[s][t][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // finite: key 0 is always valid
        const float mc = mx * c2;
        float l = 0.f;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                if (s == NS - 1 && t > 0 && s * SR + 16 * t >= N) {   // padding-only tile: probabilities 0
                    st[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[s][t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[s][t][r], c2, -mc));
                    l += st[s][t][r];
                }
            }
            const u32x4 pc = pack(st[s], T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma(col_frag(sV, s * SR, dt * 16, i, g, T()), pc, o[dt], T());
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = osc / l;
        if (qok) {
            // (8-B pieces, 32-B row segments: the 16-B form of store_acc_row was measured here and lost -- 94 -> 106 us at
            // 197 tokens: the exchange sits on the tile loop's dependent path)
            T* orow = out + ((int64_t)pr.b * N + q) * D + pr.head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = o[dt][r] * inv;
                store4<T>(orow + dt * 16 + 4 * g, v);
            }
            if (g == 0) lse[((int64_t)pr.b * H + pr.head) * N + q] = mx * 0.125f + __logf(l);
        }
    }
}

// ------------------------------------------------------------------------------ backward: dQ
// NS > 0: the number of 16*TPS-row steps (hence the LDS image size) at compile time, so reads of the
// second image are immediate offsets from the first one's address registers (with a run-time size
// the step loop carried 28 address registers and incremented every one each step: 40 % of its
// vector instructions).  Fully unrolling the loop spills.  NS = 0: run-time count.
template <typename T, int PW, int NS>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const T* __restrict__ qkv,
                                                          const T* __restrict__ out,
                                                          const T* __restrict__ dout,
                                                          const float* __restrict__ lse,
                                                          T* __restrict__ dqkv, int Bc, int N, int H,
                                                          int npad, const float* __restrict__ rowscale,
                                                          float* __restrict__ delta_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using A = AT<T>;
    constexpr int SR = 16 * A::TPS;
    const int D = H * HD;
    const int64_t ld = 3 * D;
    // compile-time image size: the second image is an immediate offset away (staging keeps the
    // run-time npad: with a constant bound its loops unroll into spills)
    const int tile_bytes = (NS > 0 ? NS * SR : npad) * A::ROWB;
    const Prob pr = locate<PW>(smem, 2 * tile_bytes, Bc * H, H);
    char* sK = pr.lds;
    char* sV = pr.lds + tile_bytes;
    const T* base = qkv + (int64_t)pr.b * N * ld + pr.head * HD;
    if (pr.valid) {
        stage_rows_pair<T>(sK, base + D, ld, sV, base + 2 * D, ld, N, npad, pr.gt, PW * 64);
    }
    __syncthreads();
    if (!pr.valid) return;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const float scale = 0.125f, c2 = 0.125f * LOG2E;
    const float rs = rowscale ? rowscale[pr.b] : 1.f;
    const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
    const int nqt = (N + 15) / 16, nsteps = NS > 0 ? NS : npad / SR;
    // operands of the next query tile are requested before this tile's key loop
    u32x4 qn_[A::NH], don_[A::NH], on_[A::NH];
    float lqn_ = 0.f;
    auto fetch = [&](int qt) {
        const int q = qt * 16 + i;
        const bool ok = qt < nqt && q < N;
        const int64_t orow = ((int64_t)pr.b * N + q) * D + pr.head * HD;
#pragma unroll
        for (int h = 0; h < A::NH; ++h) {
            qn_[h] = gload_frag<T>(base + (int64_t)q * ld, g + 4 * h, ok);
            don_[h] = gload_frag<T>(dout + orow, g + 4 * h, ok);
            on_[h] = gload_frag<T>(out + orow, g + 4 * h, ok);
        }
        lqn_ = ok ? lse[((int64_t)pr.b * H + pr.head) * N + q] : 0.f;
    };
    fetch(pr.gw);
    for (int qt = pr.gw; qt < nqt; qt += PW) {
        const int q = qt * 16 + i;
        const bool qok = q < N;
        u32x4 qf[A::NH], dof[A::NH];
        float delta = 0.f;
#pragma unroll
        for (int h = 0; h < A::NH; ++h) {
            qf[h] = qn_[h];
            dof[h] = don_[h];
            delta += dot_chunk<T>(dof[h], on_[h]);
        }
        const float lqc = lqn_ * LOG2E;
        fetch(qt + PW);
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        delta *= inv_rs;
        if (delta_ws && g == 0 && qok) delta_ws[((int64_t)pr.b * H + pr.head) * N + q] = delta;
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const int key0 = s * SR;
            f32x4 ds[A::TPS];
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    mma(row_frag<T>(sK, key0 + 16 * t, i, g + 4 * h), qf[h], st, T());
                    mma(row_frag<T>(sV, key0 + 16 * t, i, g + 4 * h), dof[h], dp, T());
                }
                // no key mask: the staged K rows >= N are zero, so whatever dS holds there adds
                // nothing to dQ; the softmax scale is applied to dQ once, at the end
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ds[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -lqc)) * (dp[r] - delta);
            }
            const u32x4 dc = pack(ds, T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma(col_frag(sK, key0, dt * 16, i, g, T()), dc, dq[dt], T());
        }
        if (qok) {
            T* drow = dqkv + ((int64_t)pr.b * N + q) * ld + pr.head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                float v[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
                store4<T>(drow + dt * 16 + 4 * g, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------ backward: dK, dV
template <typename T, int PW, int NS>
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_kernel(const T* __restrict__ qkv,
                                                           const T* __restrict__ out,
                                                           const T* __restrict__ dout,
                                                           const float* __restrict__ lse,
                                                           T* __restrict__ dqkv, int Bc, int N, int H,
                                                           int npad, const float* __restrict__ rowscale,
                                                           const float* __restrict__ delta_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using A = AT<T>;
    constexpr int SR = 16 * A::TPS, NCH = HD / A::EPC;
    const int D = H * HD;
    const int64_t ld = 3 * D;
    // compile-time image size: the second image is an immediate offset away (staging keeps the
    // run-time npad: with a constant bound its loops unroll into spills)
    const int tile_bytes = (NS > 0 ? NS * SR : npad) * A::ROWB;
    const int group_bytes = 2 * tile_bytes + 2 * (NS > 0 ? NS * SR : npad) * 4;
    const Prob pr = locate<PW>(smem, group_bytes, Bc * H, H);
    char* sQ = pr.lds;
    char* sdO = pr.lds + tile_bytes;
    float* sL = (float*)(pr.lds + 2 * tile_bytes);
    float* sDl = sL + (NS > 0 ? NS * SR : npad);
    const T* base = qkv + (int64_t)pr.b * N * ld + pr.head * HD;
    const int64_t obase = (int64_t)pr.b * N * D + pr.head * HD;
    const float rs = rowscale ? rowscale[pr.b] : 1.f;
    const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
    if (pr.valid) {
        stage_rows_pair<T>(sQ, base, ld, sdO, dout + obase, D, N, npad, pr.gt, PW * 64);
        // delta[q] = rowsum(dO * O) / rowscale: left by the dQ kernel, or computed here (NCH
        // consecutive lanes share a row)
        if (delta_ws) {
            const int64_t rb = ((int64_t)pr.b * H + pr.head) * N;
            for (int r = pr.gt; r < npad; r += PW * 64) {
                sDl[r] = r < N ? delta_ws[rb + r] : 0.f;
                sL[r] = r < N ? lse[rb + r] * LOG2E : 0.f;
            }
        }
        constexpr int DB = 4;  // row chunks in flight per thread
#pragma unroll 1
        for (int q0 = 0; q0 < (delta_ws ? 0 : npad * NCH); q0 += PW * 64 * DB) {
            u32x4 a[DB], b[DB];
            float lv[DB];
#pragma unroll
            for (int u = 0; u < DB; ++u) {
                const int qq = q0 + u * PW * 64 + pr.gt;  // npad*NCH is a multiple of 64: whole waves stay active
                const int r = qq / NCH, c = qq % NCH;
                a[u] = b[u] = u32x4{0u, 0u, 0u, 0u};
                lv[u] = 0.f;
                if (r < N) {
                    a[u] = *(const u32x4*)(dout + obase + (int64_t)r * D + c * A::EPC);
                    b[u] = *(const u32x4*)(out + obase + (int64_t)r * D + c * A::EPC);
                    if (c == 0) lv[u] = lse[((int64_t)pr.b * H + pr.head) * N + r] * LOG2E;
                }
            }
#pragma unroll
            for (int u = 0; u < DB; ++u) {
                const int qq = q0 + u * PW * 64 + pr.gt;
                const int r = qq / NCH, c = qq % NCH;
                float part = dot_chunk<T>(a[u], b[u]);
#pragma unroll
                for (int o = NCH / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                if (c == 0 && r < npad) {
                    sDl[r] = part * inv_rs;
                    sL[r] = lv[u];
                }
            }
        }
    }
    __syncthreads();
    if (!pr.valid) return;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const float scale = 0.125f, c2 = 0.125f * LOG2E;
    const int nkt = (N + 15) / 16, nsteps = NS > 0 ? NS : npad / SR;
    u32x4 kn_[A::NH], vn_[A::NH];
    auto fetch = [&](int kt) {
        const int key = kt * 16 + i;
        const bool ok = kt < nkt && key < N;
#pragma unroll
        for (int h = 0; h < A::NH; ++h) {
            kn_[h] = gload_frag<T>(base + D + (int64_t)key * ld, g + 4 * h, ok);
            vn_[h] = gload_frag<T>(base + 2 * D + (int64_t)key * ld, g + 4 * h, ok);
        }
    };
    fetch(pr.gw);
    for (int kt = pr.gw; kt < nkt; kt += PW) {
        const int key = kt * 16 + i;
        const bool kok = key < N;
        u32x4 kf[A::NH], vf[A::NH];
#pragma unroll
        for (int h = 0; h < A::NH; ++h) {
            kf[h] = kn_[h];
            vf[h] = vn_[h];
        }
        fetch(kt + PW);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const int q0 = s * SR;
            f32x4 pt[A::TPS], ds[A::TPS];
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    mma(row_frag<T>(sQ, q0 + 16 * t, i, g + 4 * h), kf[h], st, T());
                    mma(row_frag<T>(sdO, q0 + 16 * t, i, g + 4 * h), vf[h], dp, T());
                }
                const int qb = q0 + 16 * t + 4 * g;
                const f32x4 l4 = *(const f32x4*)(sL + qb), d4 = *(const f32x4*)(sDl + qb);
                // no query mask: the staged Q / dO rows >= N are zero (and their delta), so those
                // rows add nothing to dK / dV; the softmax scale is applied to dK once, at the end
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -l4[r]));
                    pt[t][r] = p;
                    ds[t][r] = p * (dp[r] - d4[r]);
                }
            }
            const u32x4 pc = pack(pt, T()), dc = pack(ds, T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                mma(col_frag(sdO, q0, dt * 16, i, g, T()), pc, dv[dt], T());
                mma(col_frag(sQ, q0, dt * 16, i, g, T()), dc, dk[dt], T());
            }
        }
        if (kok) {
            T* drow = dqkv + ((int64_t)pr.b * N + key) * ld + pr.head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                float a[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
                float b[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
                store4<T>(drow + D + dt * 16 + 4 * g, a);
                store4<T>(drow + 2 * D + dt * 16 + 4 * g, b);
            }
        }
    }
}


// ------------------------------------------------------------------------------ long sequences
// N too long for a whole problem in LDS (785 tokens at ViT-S/8): workgroup = (crop, head, group of
// four 16-row tiles), one tile per wave, the other side streamed through LDS in chunks of CH rows.
// Same orientations and the same per-step arithmetic as the whole-problem kernels; the forward
// carries the online-softmax state (m, l, O) across chunks.
template <typename T> struct LongCfg;
template <> struct LongCfg<bf16> { static constexpr int CH = 256; };
template <> struct LongCfg<float> { static constexpr int CH = 128; };

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_long_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                               float* __restrict__ lse, int Bc, int N, int H,
                                                               const float* __restrict__ rowscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using A = AT<T>;
    constexpr int SR = 16 * A::TPS, CH = LongCfg<T>::CH;
    const int D = H * HD;
    const int64_t ld = 3 * D;
    char* sK = smem;
    char* sV = smem + CH * A::ROWB;
    const int b = blockIdx.x / H, head = blockIdx.x % H;
    const T* base = qkv + (int64_t)b * N * ld + head * HD;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int q = (blockIdx.y * 4 + wave) * 16 + i;
    const bool qok = q < N;
    const float c2 = 0.125f * LOG2E;
    const float osc = rowscale ? rowscale[b] : 1.f;
    u32x4 qf[A::NH];
#pragma unroll
    for (int h = 0; h < A::NH; ++h) qf[h] = gload_frag<T>(base + (int64_t)q * ld, g + 4 * h, qok);
    float m = -INFINITY, l = 0.f;  // m in raw-score units
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int nv = min(CH, N - c0), np = (nv + SR - 1) / SR * SR;
        __syncthreads();  // every wave has finished reading the previous chunk
        stage_rows<T>(sK, base + D + (int64_t)c0 * ld, ld, nv, np, threadIdx.x, 256);
        stage_rows<T>(sV, base + 2 * D + (int64_t)c0 * ld, ld, nv, np, threadIdx.x, 256);
        __syncthreads();
        for (int s = 0; s < np / SR; ++s) {
            const int key0 = s * SR;
            f32x4 st[A::TPS];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < A::NH; ++h)
                    mma(row_frag<T>(sK, key0 + 16 * t, i, g + 4 * h), qf[h], st[t], T());
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (key0 + 16 * t + 4 * g + r >= nv) st[t][r] = -INFINITY;
                    mx = fmaxf(mx, st[t][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);  // finite: every step holds at least one valid key
            const float alpha = __builtin_amdgcn_exp2f((m - mnew) * c2);
            const float mc = mnew * c2;
            float ls = 0.f;
#pragma unroll
            for (int t = 0; t < A::TPS; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][r], c2, -mc));
                    ls += st[t][r];
                }
            l = l * alpha + ls;
            m = mnew;
            const u32x4 pc = pack(st, T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
                mma(col_frag(sV, key0, dt * 16, i, g, T()), pc, o[dt], T());
            }
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qok) {
        const float inv = osc / l;
        T* orow = out + ((int64_t)b * N + q) * D + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = o[dt][r] * inv;
            store4<T>(orow + dt * 16 + 4 * g, v);
        }
        if (g == 0) lse[((int64_t)b * H + head) * N + q] = m * 0.125f + __logf(l);
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_long_kernel(const T* __restrict__ qkv,
                                                                  const T* __restrict__ out,
                                                                  const T* __restrict__ dout,
                                                                  const float* __restrict__ lse,
                                                                  T* __restrict__ dqkv, int Bc, int N, int H,
                                                                  const float* __restrict__ rowscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using A = AT<T>;
    constexpr int SR = 16 * A::TPS, CH = LongCfg<T>::CH;
    const int D = H * HD;
    const int64_t ld = 3 * D;
    char* sK = smem;
    char* sV = smem + CH * A::ROWB;
    const int b = blockIdx.x / H, head = blockIdx.x % H;
    const T* base = qkv + (int64_t)b * N * ld + head * HD;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int q = (blockIdx.y * 4 + wave) * 16 + i;
    const bool qok = q < N;
    const float scale = 0.125f, c2 = 0.125f * LOG2E;
    const float rs = rowscale ? rowscale[b] : 1.f;
    const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
    const int64_t orow = ((int64_t)b * N + q) * D + head * HD;
    u32x4 qf[A::NH], dof[A::NH];
    float delta = 0.f;
#pragma unroll
    for (int h = 0; h < A::NH; ++h) {
        qf[h] = gload_frag<T>(base + (int64_t)q * ld, g + 4 * h, qok);
        dof[h] = gload_frag<T>(dout + orow, g + 4 * h, qok);
        delta += dot_chunk<T>(dof[h], gload_frag<T>(out + orow, g + 4 * h, qok));
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    delta *= inv_rs;
    const float lqc = (qok ? lse[((int64_t)b * H + head) * N + q] : 0.f) * LOG2E;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int nv = min(CH, N - c0), np = (nv + SR - 1) / SR * SR;
        __syncthreads();
        stage_rows<T>(sK, base + D + (int64_t)c0 * ld, ld, nv, np, threadIdx.x, 256);
        stage_rows<T>(sV, base + 2 * D + (int64_t)c0 * ld, ld, nv, np, threadIdx.x, 256);
        __syncthreads();
        for (int s = 0; s < np / SR; ++s) {
            const int key0 = s * SR;
            f32x4 ds[A::TPS];
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    mma(row_frag<T>(sK, key0 + 16 * t, i, g + 4 * h), qf[h], st, T());
                    mma(row_frag<T>(sV, key0 + 16 * t, i, g + 4 * h), dof[h], dp, T());
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)  // (zero K rows beyond nv add nothing to dQ)
                    ds[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -lqc)) * (dp[r] - delta);
            }
            const u32x4 dc = pack(ds, T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma(col_frag(sK, key0, dt * 16, i, g, T()), dc, dq[dt], T());
        }
    }
    if (qok) {
        T* drow = dqkv + ((int64_t)b * N + q) * ld + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float v[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
            store4<T>(drow + dt * 16 + 4 * g, v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_long_kernel(const T* __restrict__ qkv,
                                                                   const T* __restrict__ out,
                                                                   const T* __restrict__ dout,
                                                                   const float* __restrict__ lse,
                                                                   T* __restrict__ dqkv, int Bc, int N, int H,
                                                                   const float* __restrict__ rowscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using A = AT<T>;
    constexpr int SR = 16 * A::TPS, CH = LongCfg<T>::CH, NCH = HD / A::EPC;
    const int D = H * HD;
    const int64_t ld = 3 * D;
    char* sQ = smem;
    char* sdO = smem + CH * A::ROWB;
    float* sL = (float*)(smem + 2 * CH * A::ROWB);
    float* sDl = sL + CH;
    const int b = blockIdx.x / H, head = blockIdx.x % H;
    const T* base = qkv + (int64_t)b * N * ld + head * HD;
    const int64_t obase = (int64_t)b * N * D + head * HD;
    const float rs = rowscale ? rowscale[b] : 1.f;
    const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int key = (blockIdx.y * 4 + wave) * 16 + i;
    const bool kok = key < N;
    const float scale = 0.125f, c2 = 0.125f * LOG2E;
    u32x4 kf[A::NH], vf[A::NH];
#pragma unroll
    for (int h = 0; h < A::NH; ++h) {
        kf[h] = gload_frag<T>(base + D + (int64_t)key * ld, g + 4 * h, kok);
        vf[h] = gload_frag<T>(base + 2 * D + (int64_t)key * ld, g + 4 * h, kok);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int nv = min(CH, N - c0), np = (nv + SR - 1) / SR * SR;
        __syncthreads();
        stage_rows<T>(sQ, base + (int64_t)c0 * ld, ld, nv, np, threadIdx.x, 256);
        stage_rows<T>(sdO, dout + obase + (int64_t)c0 * D, D, nv, np, threadIdx.x, 256);
        // delta[q] = rowsum(dO * O) / rowscale and log2-domain lse of the chunk's rows
        for (int qq = threadIdx.x; qq < np * NCH; qq += 256) {  // np*NCH is a multiple of 64
            const int r = qq / NCH, c = qq % NCH;
            float part = 0.f;
            if (r < nv) {
                const u32x4 a = *(const u32x4*)(dout + obase + (int64_t)(c0 + r) * D + c * A::EPC);
                const u32x4 bb = *(const u32x4*)(out + obase + (int64_t)(c0 + r) * D + c * A::EPC);
                part = dot_chunk<T>(a, bb);
            }
#pragma unroll
            for (int o = NCH / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            if (c == 0) {
                sDl[r] = part * inv_rs;
                sL[r] = r < nv ? lse[((int64_t)b * H + head) * N + c0 + r] * LOG2E : 0.f;
            }
        }
        __syncthreads();
        for (int s = 0; s < np / SR; ++s) {
            const int q0 = s * SR;
            f32x4 pt[A::TPS], ds[A::TPS];
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    mma(row_frag<T>(sQ, q0 + 16 * t, i, g + 4 * h), kf[h], st, T());
                    mma(row_frag<T>(sdO, q0 + 16 * t, i, g + 4 * h), vf[h], dp, T());
                }
                const int qb = q0 + 16 * t + 4 * g;
                const f32x4 l4 = *(const f32x4*)(sL + qb), d4 = *(const f32x4*)(sDl + qb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // (zero Q / dO rows beyond nv add nothing)
                    const float pr_ = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -l4[r]));
                    pt[t][r] = pr_;
                    ds[t][r] = pr_ * (dp[r] - d4[r]);
                }
            }
            const u32x4 pc = pack(pt, T()), dc = pack(ds, T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                mma(col_frag(sdO, q0, dt * 16, i, g, T()), pc, dv[dt], T());
                mma(col_frag(sQ, q0, dt * 16, i, g, T()), dc, dk[dt], T());
            }
        }
    }
    if (kok) {
        T* drow = dqkv + ((int64_t)b * N + key) * ld + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float a[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
            float bb[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
            store4<T>(drow + D + dt * 16 + 4 * g, a);
            store4<T>(drow + 2 * D + dt * 16 + 4 * g, bb);
        }
    }
}

// ------------------------------------------------------------------------------ long sequences, bf16 (round 3)
// The chunked kernels above stage every 256-row chunk through registers between two __syncthreads (four exposed
// memory latencies per chunk, nothing else in flight), re-stream K/V once per 64 query rows and run the online
// softmax once per 32 keys with two cross-lane exchanges each: at ViT-S/8 (785 tokens) they ran at 0.10-0.13
// MFMA-busy and were 40 % of the step.  These three replace them for bf16:
//   * workgroup = 8 waves = 128 rows of the side that stays in registers (one 16-row tile per wave), two workgroups
//     per CU; workgroups of one problem are neighbours on one XCD (xcd_remap), so the streamed side is an L2 hit for
//     all but the first of them;
//   * the streamed side arrives by LDS-DMA in chunks of 128 rows into a double buffer: chunk j + 1 is requested right
//     behind the one barrier of chunk j and lands under its arithmetic; the DMA lanes pick their global chunk so
//     that the image has the swizzle row_frag / col_frag expect; rows past N read as zeros (descriptor range check);
//   * forward: the scores of a whole chunk (8 key tiles) stay in registers -- one max, one pair of cross-lane
//     exchanges (v_permlane16/32_swap, no LDS traffic), one rescale of O per chunk instead of per 32 keys;
//   * backward: the dQ kernel leaves rowsum(dO * O) / rowscale in delta_ws, the dK/dV kernel gets it and lse by DMA
//     with the Q / dO rows (the old one recomputed it per workgroup from `out` and `dout`).
constexpr int SCH = 128;                 // streamed rows per chunk
constexpr int SIMG = SCH * 128;          // one bf16 image of a chunk: 16 KB
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// max / sum over the four 16-lane rows of a wave (lanes of equal i): every lane gets the result
template <bool MAX> DEVI float rows4_reduce(float v) {
    float x0 = v, x1 = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
    float t = MAX ? fmaxf(x0, x1) : x0 + x1;
    x0 = x1 = t;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
    return MAX ? fmaxf(x0, x1) : x0 + x1;
}

// Per-lane byte offsets of the two DMA instructions with which wave `wave` brings rows 16*wave .. 16*wave + 15 of a
// chunk image: lane -> (row = 8u + lane / 8, physical 16-B chunk lane % 8), which holds logical chunk ^ key(row).
DEVI void stream_voffs(int wave, int lane, int pitch_bytes, int (&voff)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = 16 * wave + 8 * u + (lane >> 3);
        voff[u] = row * pitch_bytes + ((((lane & 7) ^ AT<bf16>::key(row))) << 4);
    }
}
// The DMA is issued from inline asm: told about it (the builtin), hipcc puts `s_waitcnt vmcnt(0)` in front of every
// ds_read_b64_tr_b16 that follows a request -- the intrinsic carries no address information, so it "may alias" the
// image being filled -- and the arithmetic of chunk j would wait for chunk j + 1.  The kernels wait themselves
// (vmcnt(0) before the barrier that publishes a chunk).  M0 is written here and nowhere else in these kernels.
DEVI u32x4 stream_rsrc(const void* p, int bytes) {
    const uint64_t a = (uint64_t)p;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));   // stride 0
    r[2] = __builtin_amdgcn_readfirstlane(bytes);                 // range check: rows past the end read as zeros
    r[3] = 0x00020000;
    return r;
}
DEVI unsigned lds_offset(const char* p) { return (unsigned)(uintptr_t)(lds_ptr_t)p; }
DEVI void dma16(unsigned lds, int voff, const u32x4& rs, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
DEVI void dma4(unsigned lds, int voff, const u32x4& rs, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// (the chunk's offset rides in the VGPR offset: the descriptor's range check covers VGPR + immediate offsets only)
DEVI void stream_issue(const u32x4& rs, unsigned img, int wave, const int (&voff)[2], int chunk_off) {
#ifdef ATTN_ABLATE_DMA   // timing only: chunks after the first are not fetched
    if (chunk_off) return;
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) dma16(img + (16 * wave + 8 * u) * 128, voff[u] + chunk_off, rs, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// Plain vector expressions, not inline asm: these read MFMA results, and the MFMA -> VALU wait states are inserted by
// hipcc's hazard recognizer only for instructions it can see (asm helpers here returned stale accumulators).
DEVI f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
DEVI f32x2 pk_mul2(f32x2 a, f32x2 b) { return a * b; }
DEVI f32x2 pk_sub2(f32x2 a, f32x2 b) { return a - b; }
DEVI float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
DEVI f32x2 lo2(const f32x4& v) { return f32x2{v[0], v[1]}; }
DEVI f32x2 hi2(const f32x4& v) { return f32x2{v[2], v[3]}; }
DEVI void scale4(f32x4& v, float a) {
    const f32x2 l = pk_mul2(lo2(v), f32x2{a, a}), h = pk_mul2(hi2(v), f32x2{a, a});
    v = f32x4{l[0], l[1], h[0], h[1]};
}
#define BF16_ONES 0x3F803F80u   // two bf16 1.0

// (Ten waves per workgroup -- 785 tokens are 50 tiles = 5 x 10 -- measured 375 us against 237: a workgroup's waves go
// to the SIMDs 3/3/2/2, and two such workgroups do not fit a CU's register files side by side.)
constexpr int W = 8;   // waves = 16-row tiles per workgroup
__global__ __launch_bounds__(64 * W, 4) void attn_fwd_stream_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                    float* __restrict__ lse, int Bc, int N, int H,
                                                                    int ny, const float* __restrict__ rowscale) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K image | V image]
    const int D = H * HD, ld = 3 * D;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int prob = lid / ny, yb = lid - prob * ny;
    const int b = prob / H, head = prob - b * H;
    const bf16* base = qkv + (int64_t)b * N * ld + head * HD;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = (yb * W + wave) * 16 + i;
    const bool qok = q < N;
    const bool active = (yb * W + wave) * 16 < N;          // wave-uniform: this wave owns at least one query
    const int recs = (int)(((int64_t)(N - 1) * ld + HD) * 2);
    const u32x4 rsK = stream_rsrc(base + D, recs), rsV = stream_rsrc(base + 2 * D, recs);
    const unsigned lds0 = lds_offset(smem);
    int voff[2];
    stream_voffs(wave, lane, ld * 2, voff);
    auto issue = [&](int j) {
        const unsigned img = lds0 + (j & 1) * 2 * SIMG;
        stream_issue(rsK, img, wave, voff, j * SCH * ld * 2);
        stream_issue(rsV, img + SIMG, wave, voff, j * SCH * ld * 2);
    };
    issue(0);
    const float c2 = 0.125f * LOG2E;
    const float osc = rowscale ? rowscale[b] : 1.f;
    u32x4 qf[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) qf[h] = gload_frag<bf16>(base + (int64_t)q * ld, g + 4 * h, qok);
    float m = -INFINITY;              // running max, raw-score units
    f32x4 o[4], osum = {0.f, 0.f, 0.f, 0.f};   // osum: P times a tile of ones -- the row sums, on the MFMA pipe
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = {BF16_ONES, BF16_ONES, BF16_ONES, BF16_ONES};
    const int nch = (N + SCH - 1) / SCH;
    for (int j = 0; j < nch; ++j) {
        // chunk j has landed (mine: vmcnt; everybody's: the barrier) and everybody has finished chunk j - 1, whose
        // buffer the next request overwrites
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): as the builtin, so that hipcc knows its own loads have landed too
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 1 < nch) issue(j + 1);
        if (!active) continue;
        const char* sK = smem + (j & 1) * 2 * SIMG;
        const char* sV = sK + SIMG;
        const int nv = min(SCH, N - j * SCH);
        f32x4 st[8];
        float mx;
        if (nv == SCH) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h) mma(row_frag<bf16>(sK, 16 * t, i, g + 4 * h), qf[h], st[t], bf16());
            }
            mx = max3f(st[0][0], st[0][1], st[0][2]);
            mx = max3f(mx, st[0][3], st[1][0]);
            mx = max3f(mx, st[1][1], st[1][2]);
#pragma unroll
            for (int t = 2; t < 8; t += 2) {
                mx = max3f(mx, st[t - 1][3], st[t][0]);
                mx = max3f(mx, st[t][1], st[t][2]);
                mx = max3f(mx, st[t][3], st[t + 1][0]);
                mx = max3f(mx, st[t + 1][1], st[t + 1][2]);
            }
            mx = fmaxf(mx, st[7][3]);
        } else {
            mx = -INFINITY;
            const int nt = (nv + 15) >> 4;      // key tiles that hold a valid key
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                st[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if (t < nt) {
                    st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < 2; ++h) mma(row_frag<bf16>(sK, 16 * t, i, g + 4 * h), qf[h], st[t], bf16());
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (16 * t + 4 * g + r >= nv) st[t][r] = -INFINITY;
                        mx = fmaxf(mx, st[t][r]);
                    }
                }
            }
        }
        mx = rows4_reduce<true>(mx);           // finite: a chunk holds at least one valid key
        const float mnew = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f((m - mnew) * c2);
        const float mc = mnew * c2;
        m = mnew;
        const f32x2 c22 = {c2, c2}, mc2 = {-mc, -mc};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x2 a0 = pk_fma2(lo2(st[t]), c22, mc2), a1 = pk_fma2(hi2(st[t]), c22, mc2);
            st[t] = f32x4{__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1]),    // exp2(-inf) = 0
                          __builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) scale4(o[dt], alpha);
        scale4(osum, alpha);
        auto pv = [&](int s) {
            const f32x4 pr2[2] = {st[2 * s], st[2 * s + 1]};
            const u32x4 pc = pack(pr2, bf16());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma(col_frag(sV, 32 * s, dt * 16, i, g, bf16()), pc, o[dt], bf16());
            mma(ones, pc, osum, bf16());
        };
        if (nv == SCH) {
            pv(0); pv(1); pv(2); pv(3);
        } else {
            const int ns = (nv + 31) >> 5;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < ns) pv(s);
        }
    }
    if (!active) return;
    if (qok) {
        const float l = osum[0];               // (every row of the ones tile holds the same sum)
        const float inv = osc / l;
        bf16* orow = out + ((int64_t)b * N + q) * D + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = o[dt][r] * inv;
            store4<bf16>(orow + dt * 16 + 4 * g, v);
        }
        if (g == 0) lse[((int64_t)b * H + head) * N + q] = m * 0.125f + __logf(l);
    }
#endif
}

__global__ __launch_bounds__(64 * W, 4) void attn_bwd_dq_stream_kernel(
    const bf16* __restrict__ qkv, const bf16* __restrict__ out, const bf16* __restrict__ dout,
    const float* __restrict__ lse, bf16* __restrict__ dqkv, int Bc, int N, int H, int ny,
    const float* __restrict__ rowscale, float* __restrict__ delta_ws) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K image | V image]
    const int D = H * HD, ld = 3 * D;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int prob = lid / ny, yb = lid - prob * ny;
    const int b = prob / H, head = prob - b * H;
    const bf16* base = qkv + (int64_t)b * N * ld + head * HD;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = (yb * W + wave) * 16 + i;
    const bool qok = q < N;
    const bool active = (yb * W + wave) * 16 < N;
    const int recs = (int)(((int64_t)(N - 1) * ld + HD) * 2);
    const u32x4 rsK = stream_rsrc(base + D, recs), rsV = stream_rsrc(base + 2 * D, recs);
    const unsigned lds0 = lds_offset(smem);
    int voff[2];
    stream_voffs(wave, lane, ld * 2, voff);
    auto issue = [&](int j) {
        const unsigned img = lds0 + (j & 1) * 2 * SIMG;
        stream_issue(rsK, img, wave, voff, j * SCH * ld * 2);
        stream_issue(rsV, img + SIMG, wave, voff, j * SCH * ld * 2);
    };
    issue(0);
    const float scale = 0.125f, c2 = 0.125f * LOG2E;
    const float rs = rowscale ? rowscale[b] : 1.f;
    const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
    const int64_t orow = ((int64_t)b * N + q) * D + head * HD;
    u32x4 qf[2], dof[2];
    float delta = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        qf[h] = gload_frag<bf16>(base + (int64_t)q * ld, g + 4 * h, qok);
        dof[h] = gload_frag<bf16>(dout + orow, g + 4 * h, qok);
        delta += dot_chunk<bf16>(dof[h], gload_frag<bf16>(out + orow, g + 4 * h, qok));
    }
    delta = rows4_reduce<false>(delta) * inv_rs;
    if (qok && g == 0) delta_ws[((int64_t)b * H + head) * N + q] = delta;
    const float lqc = (qok ? lse[((int64_t)b * H + head) * N + q] : 0.f) * LOG2E;
    const f32x2 c22 = {c2, c2}, nl2 = {-lqc, -lqc}, dl2 = {delta, delta};
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nch = (N + SCH - 1) / SCH;
    for (int j = 0; j < nch; ++j) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 1 < nch) issue(j + 1);
        if (!active) continue;
        const char* sK = smem + (j & 1) * 2 * SIMG;
        const char* sV = sK + SIMG;
        const int nv = min(SCH, N - j * SCH);
        auto step = [&](int s) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    mma(row_frag<bf16>(sK, 32 * s + 16 * t, i, g + 4 * h), qf[h], st, bf16());
                    mma(row_frag<bf16>(sV, 32 * s + 16 * t, i, g + 4 * h), dof[h], dp, bf16());
                }
                // dS^T = P * (dP - delta)   (zero K rows beyond N add nothing to dQ)
                const f32x2 a0 = pk_fma2(lo2(st), c22, nl2), a1 = pk_fma2(hi2(st), c22, nl2);
                const f32x2 p0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
                const f32x2 p1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
                const f32x2 e0 = pk_mul2(p0, pk_sub2(lo2(dp), dl2)), e1 = pk_mul2(p1, pk_sub2(hi2(dp), dl2));
                ds[t] = f32x4{e0[0], e0[1], e1[0], e1[1]};
            }
            const u32x4 dc = pack(ds, bf16());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma(col_frag(sK, 32 * s, dt * 16, i, g, bf16()), dc, dq[dt], bf16());
        };
        if (nv == SCH) {
            step(0); step(1); step(2); step(3);
        } else {
            const int ns = (nv + 31) >> 5;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < ns) step(s);
        }
    }
    if (qok) {
        bf16* drow = dqkv + ((int64_t)b * N + q) * ld + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float v[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
            store4<bf16>(drow + dt * 16 + 4 * g, v);
        }
    }
#endif
}

__global__ __launch_bounds__(64 * W, 4) void attn_bwd_dkv_stream_kernel(
    const bf16* __restrict__ qkv, const bf16* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta_ws, bf16* __restrict__ dqkv, int Bc, int N, int H, int ny) {
#if defined(__HIP_DEVICE_COMPILE__)
    // [2 buffers][Q image | dO image | lse (128 floats) | delta (128 floats)]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = 2 * SIMG + 2 * SCH * 4;
    const int D = H * HD, ld = 3 * D;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int prob = lid / ny, yb = lid - prob * ny;
    const int b = prob / H, head = prob - b * H;
    const bf16* base = qkv + (int64_t)b * N * ld + head * HD;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int key = (yb * W + wave) * 16 + i;
    const bool kok = key < N;
    const bool active = (yb * W + wave) * 16 < N;
    const u32x4 rsQ = stream_rsrc(base, (int)(((int64_t)(N - 1) * ld + HD) * 2));
    const u32x4 rsO = stream_rsrc(dout + (int64_t)b * N * D + head * HD, (int)(((int64_t)(N - 1) * D + HD) * 2));
    const u32x4 rsL = stream_rsrc(lse + ((int64_t)b * H + head) * N, N * 4);
    const u32x4 rsD = stream_rsrc(delta_ws + ((int64_t)b * H + head) * N, N * 4);
    const unsigned lds0 = lds_offset(smem);
    int voffQ[2], voffO[2];
    stream_voffs(wave, lane, ld * 2, voffQ);
    stream_voffs(wave, lane, D * 2, voffO);
    auto issue = [&](int j) {
        const unsigned img = lds0 + (j & 1) * BUF;
        stream_issue(rsQ, img, wave, voffQ, j * SCH * ld * 2);
        stream_issue(rsO, img + SIMG, wave, voffO, j * SCH * D * 2);
        // waves 0-1: lse of rows 0-63 / 64-127 of the chunk; waves 2-3: delta (rows past N read as zeros)
        if (wave < 4) {
            const int half = wave & 1;
            const unsigned dst = img + 2 * SIMG + (wave >> 1) * SCH * 4 + half * 256;
            const int off = (j * SCH + half * 64) * 4;
            if (wave < 2) dma4(dst, lane * 4 + off, rsL, 0);
            else dma4(dst, lane * 4 + off, rsD, 0);
        }
    };
    issue(0);
    const float scale = 0.125f, c2 = 0.125f * LOG2E;
    const f32x2 c22 = {c2, c2}, nl2e = {-LOG2E, -LOG2E};
    u32x4 kf[2], vf[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        kf[h] = gload_frag<bf16>(base + D + (int64_t)key * ld, g + 4 * h, kok);
        vf[h] = gload_frag<bf16>(base + 2 * D + (int64_t)key * ld, g + 4 * h, kok);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nch = (N + SCH - 1) / SCH;
    for (int j = 0; j < nch; ++j) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 1 < nch) issue(j + 1);
        if (!active) continue;
        const char* sQ = smem + (j & 1) * BUF;
        const char* sdO = sQ + SIMG;
        const float* sL = (const float*)(sQ + 2 * SIMG);
        const float* sDl = sL + SCH;
        const int nv = min(SCH, N - j * SCH);
        auto step = [&](int s) {
            f32x4 pt[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    mma(row_frag<bf16>(sQ, 32 * s + 16 * t, i, g + 4 * h), kf[h], st, bf16());
                    mma(row_frag<bf16>(sdO, 32 * s + 16 * t, i, g + 4 * h), vf[h], dp, bf16());
                }
                const int qb = 32 * s + 16 * t + 4 * g;
                const f32x4 l4 = *(const f32x4*)(sL + qb), d4 = *(const f32x4*)(sDl + qb);
                // (query rows past N: Q = dO = 0, lse = delta = 0: P = 1, dS = 0, and both add nothing)
                const f32x2 a0 = pk_fma2(lo2(st), c22, pk_mul2(lo2(l4), nl2e));
                const f32x2 a1 = pk_fma2(hi2(st), c22, pk_mul2(hi2(l4), nl2e));
                const f32x2 p0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
                const f32x2 p1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
                const f32x2 e0 = pk_mul2(p0, pk_sub2(lo2(dp), lo2(d4))), e1 = pk_mul2(p1, pk_sub2(hi2(dp), hi2(d4)));
                pt[t] = f32x4{p0[0], p0[1], p1[0], p1[1]};
                ds[t] = f32x4{e0[0], e0[1], e1[0], e1[1]};
            }
            const u32x4 pc = pack(pt, bf16()), dc = pack(ds, bf16());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                mma(col_frag(sdO, 32 * s, dt * 16, i, g, bf16()), pc, dv[dt], bf16());
                mma(col_frag(sQ, 32 * s, dt * 16, i, g, bf16()), dc, dk[dt], bf16());
            }
        };
        if (nv == SCH) {
            step(0); step(1); step(2); step(3);
        } else {
            const int ns = (nv + 31) >> 5;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < ns) step(s);
        }
    }
    if (kok) {
        bf16* drow = dqkv + ((int64_t)b * N + key) * ld + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float a[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
            float bb[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
            store4<bf16>(drow + D + dt * 16 + 4 * g, a);
            store4<bf16>(drow + 2 * D + dt * 16 + 4 * g, bb);
        }
    }
#endif
}

// ------------------------------------------------------------------------------ backward, N <= 64
// Short sequences (the 37-token local crops: 15 360 problems per layer at ViT-S/16, batch 256) are
// bound by HBM traffic, not by MFMA work: the dQ kernel and the dK/dV kernel each read q, k, v and dO.
// Here ONE wave owns a whole (crop, head) problem and runs both sweeps back to back on two LDS images:
// K, V staged -> dQ sweep (S^T, dP^T orientation; Q / dO / O rows as fragments straight from memory),
// then the same two slots are refilled with Q, dO -> dK/dV sweep (S, dP orientation; the K / V row
// fragments were lifted into registers before the refill).  Each operand comes from HBM once; the
// second touch of Q and dO (the refill) is an L2 hit a few microseconds after the first.  Workgroup =
// one wave, 16.5 KB of LDS: nine problems in flight per CU (a single-staging variant with all four
// images resident ran at four per CU and was slower than the two kernels it replaced: 248 vs 235 us;
// this one 173 us).  The same fusion for 197-token problems (four waves per problem, barrier between
// the sweeps) measured 314 us against 290 us for the two launches and was dropped: those kernels are
// bound by their dependent MFMA / exp chains, not by HBM.
template <typename T, int NT>   // NT = 16-row tiles of the problem (3: up to 48 tokens, 4: up to 64)
__global__ __launch_bounds__(64) void attn_bwd_small_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                            const T* __restrict__ dout,
                                                            const float* __restrict__ lse, T* __restrict__ dqkv,
                                                            int Bc, int N, int H, int npad,
                                                            const float* __restrict__ rowscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using A = AT<T>;
    constexpr int SR = 16 * A::TPS, NCH = HD / A::EPC, CPT = NT * 16 * NCH / 64;   // chunks per thread and image (npad = 16 NT rows)
    constexpr int NFULL = NT / 2, HALF = NT & 1;   // 32-row steps, and a last step of 16 rows
    const int D = H * HD;
    const int64_t ld = 3 * D;
    const int tile_bytes = npad * A::ROWB;
    char* sX = smem;                                  // K, then Q
    char* sY = smem + tile_bytes;                     // V, then dO
    float* sL = (float*)(smem + 2 * tile_bytes);      // lse * log2(e) per query
    float* sDl = sL + npad;                           // rowsum(dO * O) / rowscale per query
    const int pid = blockIdx.x;
    const int b = pid / H, head = pid % H;
    const T* base = qkv + (int64_t)b * N * ld + head * HD;
    const int64_t obase = (int64_t)b * N * D + head * HD;
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int total = npad * NCH;
    // coalesced image loads: lane + 64 u -> (row, chunk); both images' loads in flight together
    auto load2 = [&](const T* s0, int64_t l0, const T* s1, int64_t l1, u32x4 (&v)[2][CPT]) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int q = lane + u * 64, r = q / NCH, c = q % NCH;
            v[0][u] = v[1][u] = u32x4{0u, 0u, 0u, 0u};
            if (q < total && r < N) {
                v[0][u] = *(const u32x4*)(s0 + (int64_t)r * l0 + c * A::EPC);
                v[1][u] = *(const u32x4*)(s1 + (int64_t)r * l1 + c * A::EPC);
            }
        }
    };
    auto store2 = [&](const u32x4 (&v)[2][CPT]) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int q = lane + u * 64, r = q / NCH, c = q % NCH;
            if (q < total) {
                const int off = r * A::ROWB + ((c ^ A::key(r)) << 4);
                *(u32x4*)(sX + off) = v[0][u];
                *(u32x4*)(sY + off) = v[1][u];
            }
        }
    };
    const float scale = 0.125f, c2 = 0.125f * LOG2E;
    const float rs = rowscale ? rowscale[b] : 1.f;
    const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
    const int nt = (N + 15) / 16;
    u32x4 img[2][CPT];
    load2(base + D, ld, base + 2 * D, ld, img);       // K, V
    // Q / dO / O rows of every (<= 4) query tile as MFMA fragments, and their lse: requested now too
    u32x4 qfr[NT][A::NH], dofr[NT][A::NH], ofr[NT][A::NH];
    float lq_[NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int q = qt * 16 + i;
        const bool ok = qt < nt && q < N;
#pragma unroll
        for (int h = 0; h < A::NH; ++h) {
            qfr[qt][h] = gload_frag<T>(base + (int64_t)q * ld, g + 4 * h, ok);
            dofr[qt][h] = gload_frag<T>(dout + obase + (int64_t)q * D, g + 4 * h, ok);
            ofr[qt][h] = gload_frag<T>(out + obase + (int64_t)q * D, g + 4 * h, ok);
        }
        lq_[qt] = ok ? lse[((int64_t)b * H + head) * N + q] : 0.f;
    }
    store2(img);
    __syncthreads();   // (one wave: orders the LDS writes above before the reads below)
    // ---- dQ sweep: per query tile, S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T -------------------
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        if (qt >= nt) break;
        const int q = qt * 16 + i;
        const bool qok = q < N;
        float delta = 0.f;
#pragma unroll
        for (int h = 0; h < A::NH; ++h) delta += dot_chunk<T>(dofr[qt][h], ofr[qt][h]);
        const float lqc = lq_[qt] * LOG2E;
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        delta *= inv_rs;
        if (g == 0) {
            sL[q] = lqc;
            sDl[q] = delta;
        }
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NFULL + HALF; ++s) {
            const int key0 = s * SR;
            const bool both = s < NFULL;   // the last step of an odd tile count holds 16 keys
            f32x4 ds[A::TPS];
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t == 1 && !both) continue;
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    mma(row_frag<T>(sX, key0 + 16 * t, i, g + 4 * h), qfr[qt][h], st, T());
                    mma(row_frag<T>(sY, key0 + 16 * t, i, g + 4 * h), dofr[qt][h], dp, T());
                }
                // no key mask: the staged K rows >= N are zero, so whatever dS holds there adds nothing
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ds[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -lqc)) * (dp[r] - delta);
            }
            const u32x4 dc = pack(ds, T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                mma(both ? col_frag(sX, key0, dt * 16, i, g, T()) : col_frag_lo(sX, key0, dt * 16, i, g), dc, dq[dt], T());
        }
        if (qok) {
            T* drow = dqkv + ((int64_t)b * N + q) * ld + head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                float v[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
                store4<T>(drow + dt * 16 + 4 * g, v);
            }
        }
    }
    for (int r = nt * 16 + lane; r < npad; r += 64) {   // query rows of the padding: zero Q / dO rows
        sL[r] = 0.f;
        sDl[r] = 0.f;
    }
    // ---- refill: K / V row fragments of every key tile into registers, Q / dO into the two slots ----
    load2(base, ld, dout + obase, (int64_t)D, img);   // (L2 hits: both were read as fragments above)
    u32x4 kfr[NT][A::NH], vfr[NT][A::NH];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int h = 0; h < A::NH; ++h) {
            kfr[kt][h] = row_frag<T>(sX, (kt < nt ? kt : 0) * 16, i, g + 4 * h);
            vfr[kt][h] = row_frag<T>(sY, (kt < nt ? kt : 0) * 16, i, g + 4 * h);
        }
    __syncthreads();   // every read of K / V has returned
    store2(img);
    __syncthreads();
    // ---- dK / dV sweep: per key tile, S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS --------
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (kt >= nt) break;
        const int key = kt * 16 + i;
        const bool kok = key < N;
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NFULL + HALF; ++s) {
            const int q0 = s * SR;
            const bool both = s < NFULL;
            f32x4 pt[A::TPS], ds[A::TPS];
#pragma unroll
            for (int t = 0; t < A::TPS; ++t) {
                pt[t] = ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t == 1 && !both) continue;
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    mma(row_frag<T>(sX, q0 + 16 * t, i, g + 4 * h), kfr[kt][h], st, T());
                    mma(row_frag<T>(sY, q0 + 16 * t, i, g + 4 * h), vfr[kt][h], dp, T());
                }
                const int qb = q0 + 16 * t + 4 * g;
                const f32x4 l4 = *(const f32x4*)(sL + qb), d4 = *(const f32x4*)(sDl + qb);
                // no query mask: the staged Q / dO rows >= N are zero (and their delta)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -l4[r]));
                    pt[t][r] = p;
                    ds[t][r] = p * (dp[r] - d4[r]);
                }
            }
            const u32x4 pc = pack(pt, T()), dc = pack(ds, T());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                mma(both ? col_frag(sY, q0, dt * 16, i, g, T()) : col_frag_lo(sY, q0, dt * 16, i, g), pc, dv[dt], T());
                mma(both ? col_frag(sX, q0, dt * 16, i, g, T()) : col_frag_lo(sX, q0, dt * 16, i, g), dc, dk[dt], T());
            }
        }
        if (kok) {
            T* drow = dqkv + ((int64_t)b * N + key) * ld + head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                float a[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
                float v[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
                store4<T>(drow + D + dt * 16 + 4 * g, a);
                store4<T>(drow + 2 * D + dt * 16 + 4 * g, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------ backward, fused (65..208 tokens)
// The two-kernel backward moves 12 x the size of one [tokens, 64] operand per head (K, V staged + Q, dO, O read
// + dQ written; then Q, dO staged + K, V read + dK, dV written), and with its whole key loop removed the dQ
// kernel still takes 108 of its 127 us: at 197 tokens the backward is paced by bytes, not by its arithmetic
// (profiles/r02_attention_backward_ablations.txt).  Here every operand crosses the memory path once (8 x):
// one persistent 8-wave workgroup per CU keeps all four images of a problem (K, V, Q, dO: 104 KiB) in LDS,
// runs the 13 dQ row tiles and the 13 dK/dV row tiles of the problem as 26 independent tasks over its waves
// (same orientations and per-step arithmetic as the two kernels), and holds the NEXT problem's operands --
// requested before the tasks start -- in registers (20 x 16 B per thread) while it works, so that the memory
// path is busy during the arithmetic and the only exposed load is the first problem's.
// LDS reads through explicit 32-bit bases + compile-time offsets.  With generic pointer arithmetic the step loops of
// the 208-row form re-derived ~25 addresses per step in VALU instructions: the images past the first 64 KB (dO at
// 3 * 26 KB, lse / delta behind it) are out of reach of the 16-bit offset field from the array's base, and there are
// no registers to keep their addresses.  Here each task keeps one running base per lane-dependent pattern (7 for a
// dK/dV tile, 6 for a dQ tile), taken relative to the first image ITS loop reads, so that everything else is an
// immediate offset, and advances them by one step's 4 KB.
typedef __attribute__((address_space(3))) const char* lds_cptr;
DEVI lds_cptr to_lds(const char* p) { return (lds_cptr)p; }
DEVI u32x4 lds128(lds_cptr p, int off) { return *(const __attribute__((address_space(3))) u32x4*)(p + off); }
DEVI f32x4 lds128f(lds_cptr p, int off) { return *(const __attribute__((address_space(3))) f32x4*)(p + off); }
template <bool BOTH> DEVI u32x4 lds_tr(lds_cptr p, int off) {   // col_frag / col_frag_lo at p + off
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const u32x2 l2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + off)));
    if (!BOTH) return u32x4{l2[0], l2[1], 0u, 0u};
    const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + off + 16 * AT<bf16>::ROWB)));
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}
// lane offsets of the two fragment kinds inside an image whose first row is a multiple of 32 (the swizzle key looks
// at row bits 1-2 only, so it does not depend on the step)
DEVI int row_frag_off(int i, int c) { return i * AT<bf16>::ROWB + ((c ^ AT<bf16>::key(i)) << 4); }
DEVI int col_frag_off(int c0, int i, int g) {
    const int q = i >> 2, p = i & 3, row = 4 * g + q, lc = (c0 + 4 * p) >> 3;
    return row * AT<bf16>::ROWB + ((lc ^ AT<bf16>::key(row)) << 4) + (p & 1) * 8;
}
constexpr int FUSED_ROWS = 208;   // rows of an image in the 8-wave form: 13 tiles of 16
// Template: ROWS = rows of an image (a multiple of 16 >= N), THREADS = workgroup size.  <208, 512> is the
// description above; <48, 64> serves the 37-token local crops with ONE wave per problem (6 tasks), six
// workgroups per CU by LDS (24.4 KiB), every operand loaded once as whole 128-B row segments -- the
// two-slot kernel above it re-reads Q and dO for its second sweep, and that refill was measured as real traffic
// (696 MB per launch against 582 MB of operands).
template <int ROWS, int THREADS, bool PERSIST>
__global__ __launch_bounds__(THREADS, 2) void attn_bwd_fused_kernel(
    const bf16* __restrict__ qkv, const bf16* __restrict__ out, const bf16* __restrict__ dout,
    const float* __restrict__ lse, bf16* __restrict__ dqkv, int Bc, int N, int H, const float* __restrict__ rowscale) {
    constexpr int FUSED_ROWS = ROWS, FUSED_THREADS = THREADS;
    constexpr int FUSED_CPT = (ROWS * 8 + THREADS - 1) / THREADS;   // 16-B chunks per thread and image (208 / 512: 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = bf16;
    using A = AT<T>;
    constexpr int IMG = FUSED_ROWS * A::ROWB, NCH = 8;
    char* sK = smem;
    char* sV = smem + IMG;
    char* sQ = smem + 2 * IMG;
    char* sdO = smem + 3 * IMG;
    float* sL = (float*)(smem + 4 * IMG);
    float* sDl = sL + FUSED_ROWS;
    int* sNext = (int*)(sDl + FUSED_ROWS);   // task counter of the problem in work
    const int D = H * HD;
    const int64_t ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int nt = (N + 15) / 16;               // 16-row tiles (<= 13)
    const int nfull = nt >> 1, half = nt & 1;   // 32-row steps; a last step of 16 rows
    const int nprob = Bc * H;
    const float scale = 0.125f, c2 = 0.125f * LOG2E;

    // ---- operands of one problem -> registers (chunk q = tid + 512 u of each image: row q / 8, chunk q % 8)
    u32x4 pk[FUSED_CPT], pv[FUSED_CPT], pq[FUSED_CPT], pdo[FUSED_CPT], po[FUSED_CPT];
    float plse = 0.f;
    auto request = [&](int pid) {
        const int b = pid / H, head = pid % H;
        const bf16* base = qkv + (int64_t)b * N * ld + head * HD;
        const int64_t obase = (int64_t)b * N * D + head * HD;
#pragma unroll
        for (int u = 0; u < FUSED_CPT; ++u) {
            const int q = tid + u * FUSED_THREADS, r = q / NCH, c = q % NCH;
            pk[u] = pv[u] = pq[u] = pdo[u] = po[u] = u32x4{0u, 0u, 0u, 0u};
            if (r < N) {
                const bf16* row = base + (int64_t)r * ld + c * 8;
                pq[u] = *(const u32x4*)(row);
                pk[u] = *(const u32x4*)(row + D);
                pv[u] = *(const u32x4*)(row + 2 * D);
                pdo[u] = *(const u32x4*)(dout + obase + (int64_t)r * D + c * 8);
                po[u] = *(const u32x4*)(out + obase + (int64_t)r * D + c * 8);
            }
        }
        plse = tid < N ? lse[((int64_t)b * H + head) * N + tid] : 0.f;   // (scaled at its use: no wait here)
    };
    int pid = blockIdx.x;
    if (pid < nprob) request(pid);
    for (; pid < nprob; pid += PERSIST ? gridDim.x : nprob) {
        const int b = pid / H, head = pid % H;
        const float rs = rowscale ? rowscale[b] : 1.f;
        const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
        // ---- registers -> LDS images; delta[row] = rowsum(dO * O) / rowscale (the 8 lanes of a row are adjacent)
#pragma unroll
        for (int u = 0; u < FUSED_CPT; ++u) {
            const int q = tid + u * FUSED_THREADS, r = q / NCH, c = q % NCH;
            float part = dot_chunk<T>(pdo[u], po[u]);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 1, 64);
            if (r < FUSED_ROWS) {
                const int off = r * A::ROWB + ((c ^ A::key(r)) << 4);
                *(u32x4*)(sK + off) = pk[u];
                *(u32x4*)(sV + off) = pv[u];
                *(u32x4*)(sQ + off) = pq[u];
                *(u32x4*)(sdO + off) = pdo[u];
                if (c == 0) sDl[r] = part * inv_rs;
            }
        }
        if (tid < FUSED_ROWS) sL[tid] = plse * LOG2E;
        if (tid == 0) *sNext = 0;
        // (raw barriers: __syncthreads() would also wait for the output stores and the next problem's loads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the next problem's operands travel while this one is worked on (PERSIST; the one-wave form has no
        // registers for them -- 30 x 16 B -- and runs one problem per workgroup, six workgroups per CU)
        if (PERSIST && pid + (int)gridDim.x < nprob) request(pid + gridDim.x);

        // 2 nt tasks, taken from a counter (13 + 13 tasks do not divide over 8 waves, and a dK/dV tile is a third
        // more work than a dQ tile): the heavier dK/dV tiles first, so that the tail is made of the light ones
        for (;;) {
            int tk = 0;
            if (lane == 0) tk = atomicAdd(sNext, 1);
            tk = __builtin_amdgcn_readfirstlane(tk);
#if defined(FUSED_ABLATE_TASKS)
            if (tk >= 0) break;
#endif
            if (tk >= 2 * nt) break;
            const int task = tk < nt ? tk + nt : tk - nt;   // counter 0..nt-1 -> dK/dV tiles, nt..2nt-1 -> dQ tiles
            if (task < nt) {
                // ---- dQ of query tile `task`: S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T
                const int q = task * 16 + i;
                u32x4 qf[A::NH], dof[A::NH];
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    qf[h] = row_frag<T>(sQ, task * 16, i, g + 4 * h);
                    dof[h] = row_frag<T>(sdO, task * 16, i, g + 4 * h);
                }
                const float lqc = sL[q], delta = sDl[q];
                f32x4 dq[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // running bases relative to the K image (V is IMG further: an immediate)
                lds_cptr bR[2], bC[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) bR[h] = to_lds(sK) + row_frag_off(i, g + 4 * h);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) bC[dt] = to_lds(sK) + col_frag_off(dt * 16, i, g);
                const f32x2 c22 = {c2, c2}, nl2 = {-lqc, -lqc}, dl2 = {delta, delta};
                auto step = [&](auto both_c) {
                    constexpr bool both = decltype(both_c)::value;
                    f32x4 ds[A::TPS];
#pragma unroll
                    for (int t = 0; t < A::TPS; ++t) {
                        ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (t == 1 && !both) continue;
                        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int h = 0; h < A::NH; ++h) {
                            mma(lds128(bR[h], t * 16 * A::ROWB), qf[h], st, T());
                            mma(lds128(bR[h], IMG + t * 16 * A::ROWB), dof[h], dp, T());
                        }
                        // no key mask: the staged K rows >= N are zero, so whatever dS holds there adds nothing
                        const f32x2 a0 = pk_fma2(lo2(st), c22, nl2), a1 = pk_fma2(hi2(st), c22, nl2);
                        const f32x2 p0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
                        const f32x2 p1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
                        const f32x2 e0 = pk_mul2(p0, pk_sub2(lo2(dp), dl2)), e1 = pk_mul2(p1, pk_sub2(hi2(dp), dl2));
                        ds[t] = f32x4{e0[0], e0[1], e1[0], e1[1]};
                    }
                    const u32x4 dc = pack(ds, T());
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) mma(lds_tr<both>(bC[dt], 0), dc, dq[dt], T());
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bR[h] += 32 * A::ROWB;
                        asm volatile("" : "+v"(bR[h]));
                    }
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        bC[dt] += 32 * A::ROWB;
                        asm volatile("" : "+v"(bC[dt]));
                    }
                };
#pragma unroll 1
                for (int s = 0; s < nfull; ++s) step(std::true_type());
                if (half) step(std::false_type());
                store_acc_row<T>(dqkv + ((int64_t)b * N + q) * ld + head * HD, dq, scale, g, q < N);
            } else {
                // ---- dK, dV of key tile `task - nt`: S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
                const int kt = task - nt, key = kt * 16 + i;
                u32x4 kf[A::NH], vf[A::NH];
#pragma unroll
                for (int h = 0; h < A::NH; ++h) {
                    kf[h] = row_frag<T>(sK, kt * 16, i, g + 4 * h);
                    vf[h] = row_frag<T>(sV, kt * 16, i, g + 4 * h);
                }
                f32x4 dk[4], dv[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // running bases relative to the Q image: dO is IMG further, lse / delta 2 * IMG -- all immediates
                lds_cptr bR[2], bC[4], bL = to_lds(sQ) + 16 * g;
#pragma unroll
                for (int h = 0; h < 2; ++h) bR[h] = to_lds(sQ) + row_frag_off(i, g + 4 * h);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) bC[dt] = to_lds(sQ) + col_frag_off(dt * 16, i, g);
                const f32x2 c22 = {c2, c2};
                auto step = [&](auto both_c) {
                    constexpr bool both = decltype(both_c)::value;
                    f32x4 pt[A::TPS], ds[A::TPS];
#pragma unroll
                    for (int t = 0; t < A::TPS; ++t) {
                        pt[t] = ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (t == 1 && !both) continue;
                        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int h = 0; h < A::NH; ++h) {
                            mma(lds128(bR[h], t * 16 * A::ROWB), kf[h], st, T());
                            mma(lds128(bR[h], IMG + t * 16 * A::ROWB), vf[h], dp, T());
                        }
                        const f32x4 l4 = lds128f(bL, 2 * IMG + t * 64), d4 = lds128f(bL, 2 * IMG + FUSED_ROWS * 4 + t * 64);
                        // no query mask: the staged Q / dO rows >= N are zero (and their delta)
                        const f32x2 a0 = pk_fma2(lo2(st), c22, -lo2(l4)), a1 = pk_fma2(hi2(st), c22, -hi2(l4));
                        const f32x2 p0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
                        const f32x2 p1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
                        const f32x2 e0 = pk_mul2(p0, pk_sub2(lo2(dp), lo2(d4))), e1 = pk_mul2(p1, pk_sub2(hi2(dp), hi2(d4)));
                        pt[t] = f32x4{p0[0], p0[1], p1[0], p1[1]};
                        ds[t] = f32x4{e0[0], e0[1], e1[0], e1[1]};
                    }
                    const u32x4 pc = pack(pt, T()), dc = pack(ds, T());
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        mma(lds_tr<both>(bC[dt], IMG), pc, dv[dt], T());
                        mma(lds_tr<both>(bC[dt], 0), dc, dk[dt], T());
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bR[h] += 32 * A::ROWB;
                        asm volatile("" : "+v"(bR[h]));
                    }
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        bC[dt] += 32 * A::ROWB;
                        asm volatile("" : "+v"(bC[dt]));
                    }
                    bL += 32 * 4;
                    asm volatile("" : "+v"(bL));
                };
#pragma unroll 1
                for (int s = 0; s < nfull; ++s) step(std::true_type());
                if (half) step(std::false_type());
                {
                    bf16* drow = dqkv + ((int64_t)b * N + key) * ld + head * HD;
                    store_acc_row<T>(drow + D, dk, scale, g, key < N);
                    store_acc_row<T>(drow + 2 * D, dv, 1.f, g, key < N);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every read of the images is done before the next problem overwrites them
        asm volatile("" ::: "memory");
    }
}

// ------------------------------------------------------------------------------ backward, one pass (65..208 tokens)
// EXPERIMENT, not built by default (-DDINO_ATTN_ONEPASS_EXPERIMENT, then DINO_ATTN_ONEPASS=1 selects it): correct (the
// attention tests pass with it) and SLOWER than attn_bwd_fused_kernel -- 197 tokens, 3 072 problems: 1 500 us against
// 247 us.  ds_add_f32 costs about 170 cycles per wave instruction here (1 130 of the 1 500 us; the LDS adds lanes one
// after another), and with plain stores in their place (wrong results, timing only) the kernel still takes 370 us:
// 44 MFMAs per 32 x 32 block instead of 56 and half the exp2 do not help a loop that is bound by its dependent chain
// at two waves per SIMD, and this block's chain is longer (S -> P -> pack -> transpose MFMA -> pack -> dQ MFMA -> LDS).
// profiles/r03_attention_onepass_experiment.log; kept as the record of VERDICT r2 item 5's suggestion.
#ifdef DINO_ATTN_ONEPASS_EXPERIMENT
// attn_bwd_fused_kernel forms every S / dP tile twice: once with queries on the lanes (dK, dV contract over
// queries) and once with keys on the lanes (dQ contracts over keys) -- 56 MFMAs, 32 exp2 per lane and two rounds of
// softmax arithmetic per 32 x 32 block of the score matrix.  Here a block is formed ONCE, by the wave that owns its
// 32 keys (dK^T, dV^T of those keys stay in its accumulators for the whole problem):
//   S = Q K^T, dP = dO V^T -> P, dS (rows = queries on the register side, columns = keys on the lanes)
//   dV^T += dO^T P, dK^T += Q^T dS                                  (as in the fused kernel)
//   dS^T = dS-as-operand x identity: an MFMA whose second operand is a one-hot tile returns the first operand
//          transposed in accumulator layout (exact: each output is one bf16 value times 1.0), which packs into the
//          second operand of
//   dQ^T = K^T dS^T, added to an fp32 image of dQ^T in LDS ([d][q], row pitch 212 dwords: the 4 x 16 lanes of an add
//          fall on 32 different banks, and so do the 8 x 4 lanes of a read-out instruction).
// 44 MFMAs, 16 exp2 and one round of softmax arithmetic per block.  Order of the additions: wave w works on query
// block (w + s) mod npair in step s, with a workgroup barrier after every step, so no two waves touch one dQ block at
// a time, step 0 STORES (nothing to clear) and every dQ element is summed in a fixed order: results do not change
// from run to run.  Staging, prefetch of the next problem (registers) and persistence as in the fused kernel.
constexpr int OP_DQP = 212;
constexpr int OP_DQ_BYTES = (64 * OP_DQP + 8 * 4) * 4;
DEVI u32x2 pack_half(const f32x2& a, const f32x2& b) {
    bf16x4 v;
    v[0] = (bf16)a[0];
    v[1] = (bf16)a[1];
    v[2] = (bf16)b[0];
    v[3] = (bf16)b[1];
    return __builtin_bit_cast(u32x2, v);
}
typedef __attribute__((address_space(3))) float* lds_fptr;
// (the scheduler may not move instructions across the ends of a block's phases: left alone it starts every LDS read of
// a block at its top, and the operands of phases that come later stand in registers the earlier ones need)
#ifdef OP_NO_PHASE
#define OP_PHASE()
#else
#define OP_PHASE() __builtin_amdgcn_sched_barrier(0)
#endif
__global__ __launch_bounds__(512, 2) void attn_bwd_onepass_kernel(
    const bf16* __restrict__ qkv, const bf16* __restrict__ out, const bf16* __restrict__ dout,
    const float* __restrict__ lse, bf16* __restrict__ dqkv, int Bc, int N, int H, const float* __restrict__ rowscale) {
    constexpr int ROWS = FUSED_ROWS, THREADS = 512, CPT = (ROWS * 8 + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = bf16;
    using A = AT<T>;
    constexpr int IMG = ROWS * A::ROWB, NCH = 8;
    char* sK = smem;
    char* sV = smem + IMG;
    char* sQ = smem + 2 * IMG;
    char* sdO = smem + 3 * IMG;
    float* sL = (float*)(smem + 4 * IMG);
    float* sDl = sL + ROWS;
    char* sDQ = (char*)(sDl + ROWS);
    const int D = H * HD;
    const int64_t ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = (N + 15) / 16;                        // 16-row tiles (<= 13)
    const int npair = (nt + 1) >> 1, odd = nt & 1;       // 32-row blocks; the last one holds 16 rows when nt is odd
    const int nprob = Bc * H;
    const float scale = 0.125f, c2 = 0.125f * LOG2E;

    u32x4 pk[CPT], pv[CPT], pq[CPT], pdo[CPT], po[CPT];
    float plse = 0.f;
    // operands of a problem -> registers, through buffer descriptors of the problem (scalar registers): the lane part
    // of every address is one of two 32-bit offsets, and rows >= N read as zeros by the descriptors' range check
    const int voffA = (tid >> 3) * (int)ld * 2 + (tid & 7) * 16, voffB = (tid >> 3) * D * 2 + (tid & 7) * 16;
    const int recA = (int)(((int64_t)(N - 1) * ld + HD) * 2), recB = (int)(((int64_t)(N - 1) * D + HD) * 2);
    auto request = [&](int pid) {
        const int b = pid / H, head = pid % H;
        const bf16* base = qkv + (int64_t)b * N * ld + head * HD;
        const int64_t obase = (int64_t)b * N * D + head * HD;
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, recA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(base + D), 0, recA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(base + 2 * D), 0, recA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsdO = __builtin_amdgcn_make_buffer_rsrc((void*)(dout + obase), 0, recB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(out + obase), 0, recB, 0x00020000);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int va = voffA + u * (THREADS / NCH) * (int)ld * 2, vb = voffB + u * (THREADS / NCH) * D * 2;
            pq[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsQ, va, 0, 0));
            pk[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, va, 0, 0));
            pv[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, va, 0, 0));
            pdo[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsdO, vb, 0, 0));
            po[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsO, vb, 0, 0));
        }
        plse = tid < N ? lse[((int64_t)b * H + head) * N + tid] : 0.f;
    };
    // lane patterns that do not depend on the block
    int offR[2], offC[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) offR[h] = row_frag_off(i, g + 4 * h);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) offC[dt] = col_frag_off(dt * 16, i, g);
    // one-hot tile: lane (i, g) holds contraction slots 4 g .. 4 g + 3 of row i
    const unsigned w0 = (i >> 2) == g ? ((i & 3) == 0 ? 0x3F80u : (i & 3) == 1 ? 0x3F800000u : 0u) : 0u;
    const unsigned w1 = (i >> 2) == g ? ((i & 3) == 2 ? 0x3F80u : (i & 3) == 3 ? 0x3F800000u : 0u) : 0u;
    int pid = blockIdx.x;
    if (pid < nprob) request(pid);
    for (; pid < nprob; pid += gridDim.x) {
#if defined(OP_ABLATE_PREFETCH)
        if (pid != (int)blockIdx.x) request(pid);
#endif
        const int b = pid / H, head = pid % H;
        const float rs = rowscale ? rowscale[b] : 1.f;
        const float inv_rs = rs != 0.f ? 1.f / rs : 0.f;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int q = tid + u * THREADS, r = q / NCH, c = q % NCH;
            float part = dot_chunk<T>(pdo[u], po[u]);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 1, 64);
            if (r < ROWS) {
                const int off = r * A::ROWB + ((c ^ A::key(r)) << 4);
                *(u32x4*)(sK + off) = pk[u];
                *(u32x4*)(sV + off) = pv[u];
                *(u32x4*)(sQ + off) = pq[u];
                *(u32x4*)(sdO + off) = pdo[u];
                if (c == 0) sDl[r] = part * inv_rs;
            }
        }
        if (tid < ROWS) sL[tid] = plse * LOG2E;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if !defined(OP_ABLATE_PREFETCH)   // (timing only: the next problem's operands are requested when they are needed)
        if (pid + (int)gridDim.x < nprob) request(pid + gridDim.x);
#endif

        const bool owner = wave < npair;                       // (scalar) this wave owns keys 32 wave .. 32 wave + 31
        const bool kboth = !(odd && wave == npair - 1);
        f32x4 dk[2][4], dv[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dk[t][dt] = dv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x2 c22 = {c2, c2};
        auto block = [&](int j, bool first, auto qb_c) {
            constexpr bool QB = decltype(qb_c)::value;
            constexpr int NQ = QB ? 2 : 1;
            const int joff = j * (32 * A::ROWB);
            // (bases of the wave's own K / V rows: formed here, per block, from a value the compiler cannot see through
            // -- hoisted out of the step loop they are six registers that the loop does not have)
            int wofs = wave * (32 * A::ROWB);
            asm volatile("" : "+s"(wofs));
            lds_cptr bKR[2], bKC[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) bKR[h] = to_lds(sK) + offR[h] + wofs;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) bKC[dt] = to_lds(sK) + offC[dt] + wofs;
            const lds_cptr bL = to_lds(sQ) + 16 * g + j * 128;
            u32x2 ph[2][2], dh[2][2];   // [query tile][key tile]: P and dS, packed halves of the second operands
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                if (qt >= NQ) {
                    ph[qt][0] = ph[qt][1] = dh[qt][0] = dh[qt][1] = u32x2{0u, 0u};
                    continue;
                }
                u32x4 qf[2], dof[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    qf[h] = lds128(to_lds(sQ) + offR[h] + joff, qt * 16 * A::ROWB);
                    dof[h] = lds128(to_lds(sQ) + offR[h] + joff, IMG + qt * 16 * A::ROWB);
                }
                const f32x4 l4 = lds128f(bL, 2 * IMG + qt * 64), d4 = lds128f(bL, 2 * IMG + ROWS * 4 + qt * 64);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // (the wave's own K / V rows come from LDS every time: 32 registers that the accumulators and the
                    // next problem's operands need more; key rows past the image read the next image -- finite values,
                    // and whatever they give is dropped below)
                    f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        mma(qf[h], lds128(bKR[h], t * 16 * A::ROWB), st, T());
                        mma(dof[h], lds128(bKR[h], IMG + t * 16 * A::ROWB), dp, T());
                    }
                    const f32x2 a0 = pk_fma2(lo2(st), c22, -lo2(l4)), a1 = pk_fma2(hi2(st), c22, -hi2(l4));
                    const f32x2 p0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
                    const f32x2 p1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
                    const f32x2 e0 = pk_mul2(p0, pk_sub2(lo2(dp), lo2(d4))), e1 = pk_mul2(p1, pk_sub2(hi2(dp), hi2(d4)));
                    ph[qt][t] = pack_half(p0, p1);
                    dh[qt][t] = pack_half(e0, e1);
                    OP_PHASE();
                }
            }
            u32x4 pc[2], dc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pc[t] = u32x4{ph[0][t][0], ph[0][t][1], ph[1][t][0], ph[1][t][1]};
                dc[t] = u32x4{dh[0][t][0], dh[0][t][1], dh[1][t][0], dh[1][t][1]};
            }
            if (!kboth) pc[1] = dc[1] = u32x4{0u, 0u, 0u, 0u};   // keys past the last tile: nothing into dQ
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 dOT = lds_tr<QB>(to_lds(sQ) + offC[dt] + joff, IMG);
                const u32x4 QT = lds_tr<QB>(to_lds(sQ) + offC[dt] + joff, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    mma(dOT, pc[t], dv[t][dt], T());
                    mma(QT, dc[t], dk[t][dt], T());
                }
                OP_PHASE();
            }
            u32x4 dsT[NQ];
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt) {
                const u32x4 one = qt == 0 ? u32x4{w0, w1, 0u, 0u} : u32x4{0u, 0u, w0, w1};
                f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
                mma(dc[0], one, t0, T());
                mma(dc[1], one, t1, T());
                const u32x2 h0 = pack_half(lo2(t0), hi2(t0)), h1 = pack_half(lo2(t1), hi2(t1));
                dsT[qt] = u32x4{h0[0], h0[1], h1[0], h1[1]};
            }
            OP_PHASE();
            const lds_fptr dqp = (lds_fptr)(to_lds(sDQ) + ((4 * g * OP_DQP + (g >> 1) * 4 + i) * 4 + j * 128));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 KT = lds_tr<true>(bKC[dt], 0);
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) {
                    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
                    mma(KT, dsT[qt], dq, T());
#if defined(OP_ABLATE_ADD)      // timing only: stores in place of the additions
                    if (true) {
#elif defined(OP_ABLATE_DQ)     // timing only: nothing leaves the dQ tiles
                    if (dq[0] == 12345.f) {
#else
                    if (first) {
#endif
#pragma unroll
                        for (int r = 0; r < 4; ++r) dqp[(dt * 16 + r) * OP_DQP + dt * 8 + qt * 16] = dq[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            __hip_atomic_fetch_add(dqp + ((dt * 16 + r) * OP_DQP + dt * 8 + qt * 16), dq[r], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                OP_PHASE();
            }
        };
#pragma unroll 1
        for (int s = 0; s < npair; ++s) {
            if (owner) {
                int j = wave + s;
                if (j >= npair) j -= npair;
                if (odd && j == npair - 1) block(j, s == 0, std::false_type());
                else block(j, s == 0, std::true_type());
            }
#if defined(OP_ABLATE_STEPBAR)   // timing only
            if (s == npair - 1)
#endif
            {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        bf16* const pbase = dqkv + (int64_t)b * N * ld + head * HD;   // (scalar) the problem's rows of dqkv
        if (owner) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int key = wave * 32 + t * 16 + i;
                const unsigned eoff = (unsigned)(key * (int)ld + D);
                store_acc_row<T>(pbase, eoff, dk[t], scale, g, key < N);
                store_acc_row<T>(pbase, eoff + D, dv[t], 1.f, g, key < N);
            }
        }
        // dQ: [d][q] fp32 image -> rows of the output
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int it = tid + u * THREADS, q = it >> 3, c = it & 7;
            if (q < N) {
                const lds_fptr src = (lds_fptr)(to_lds(sDQ) + ((c * 8 * OP_DQP + c * 4 + q) * 4));
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)(src[e * OP_DQP] * scale);
                *(u32x4*)((char*)pbase + (unsigned)(voffA + u * (THREADS / NCH) * (int)ld * 2)) = __builtin_bit_cast(u32x4, v);
            }
        }
        // (the next problem's staging writes the operand images only; its first dQ store comes after that
        // staging's barrier, which every wave reaches after its own read-out)
    }
}
#endif   // DINO_ATTN_ONEPASS_EXPERIMENT

constexpr int LDS_LIMIT = 160 * 1024;

template <typename T> int pad_rows(int N) {
    constexpr int SR = 16 * AT<T>::TPS;
    return (N + SR - 1) / SR * SR;
}

int num_cus_attn() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        n = dino_usable_cus(n);
    }
    return n;
}

// DINO_ATTN_STREAM=0: long bf16 sequences by the older register-staged chunk kernels
// DINO_ATTN_STREAM_MIN=n: bf16 sequences of at least n tokens (n > 64) take the streamed kernels even when a whole
// problem would fit in LDS (measurement switch; unset: only sequences that do not fit)
int stream_min() {
    static const int n = getenv("DINO_ATTN_STREAM_MIN") ? atoi(getenv("DINO_ATTN_STREAM_MIN")) : 1 << 30;
    return n < 65 ? 65 : n;
}
bool stream_off() {
    static const bool off = getenv("DINO_ATTN_STREAM") && getenv("DINO_ATTN_STREAM")[0] == '0';
    return off;
}

template <typename K> int set_smem(K kernel, int bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e == hipSuccess ? 0 : (int)e;
}

template <typename T>
int fwd_impl(const void* qkv, void* out, float* lse, int Bc, int N, int H, const float* rowscale,
             hipStream_t st) {
    const int npad = pad_rows<T>(N);
    const int per = 2 * npad * AT<T>::ROWB;
    const int nprob = Bc * H;
    const int nsteps = npad / (16 * AT<T>::TPS);
#define ATTN_FWD_LAUNCH(KERNEL, GRID, SMEM)                                                          \
    {                                                                                                \
        if (int e = set_smem(KERNEL, SMEM)) return e;                                                \
        hipLaunchKernelGGL((KERNEL), dim3(GRID), dim3(256), SMEM, st, (const T*)qkv, (T*)out, lse, Bc, N, H, \
                           npad, rowscale);                                                          \
    }
#define ATTN_FWD_CASE(PWV, NSV, GRID, SMEM) \
    case NSV: ATTN_FWD_LAUNCH((attn_fwd_kernel<T, PWV, NSV>), GRID, SMEM) break;
    if (N <= 64) {
        switch (nsteps) {
            ATTN_FWD_CASE(1, 1, (nprob + 3) / 4, 4 * per)
            ATTN_FWD_CASE(1, 2, (nprob + 3) / 4, 4 * per)
            ATTN_FWD_CASE(1, 3, (nprob + 3) / 4, 4 * per)
            ATTN_FWD_CASE(1, 4, (nprob + 3) / 4, 4 * per)
            default: return DINO_ERR_UNSUPPORTED;
        }
    } else if ((per > LDS_LIMIT || nsteps > 16 || N >= stream_min()) && sizeof(T) == 2 && !stream_off()) {
        // bf16: LDS-DMA streamed kernels, 8 query tiles per workgroup
        const int ny = ((N + 15) / 16 + W - 1) / W, smem = 4 * SIMG;
        if (int e = set_smem(attn_fwd_stream_kernel, smem)) return e;
        hipLaunchKernelGGL(attn_fwd_stream_kernel, dim3(nprob * ny), dim3(64 * W), smem, st, (const bf16*)qkv,
                           (bf16*)out, lse, Bc, N, H, ny, rowscale);
    } else if (per > LDS_LIMIT || nsteps > 16) {
        constexpr int CH = LongCfg<T>::CH;
        const int smem = 2 * CH * AT<T>::ROWB;
        if (int e = set_smem(attn_fwd_long_kernel<T>, smem)) return e;
        hipLaunchKernelGGL((attn_fwd_long_kernel<T>), dim3(nprob, ((N + 15) / 16 + 3) / 4), dim3(256), smem, st,
                           (const T*)qkv, (T*)out, lse, Bc, N, H, rowscale);
    } else {
        switch (nsteps) {
            ATTN_FWD_CASE(4, 3, nprob, per)
            ATTN_FWD_CASE(4, 4, nprob, per)
            ATTN_FWD_CASE(4, 5, nprob, per)
            ATTN_FWD_CASE(4, 6, nprob, per)
            ATTN_FWD_CASE(4, 7, nprob, per)
            ATTN_FWD_CASE(4, 8, nprob, per)
            ATTN_FWD_CASE(4, 9, nprob, per)
            ATTN_FWD_CASE(4, 10, nprob, per)
            ATTN_FWD_CASE(4, 11, nprob, per)
            ATTN_FWD_CASE(4, 12, nprob, per)
            ATTN_FWD_CASE(4, 13, nprob, per)
            ATTN_FWD_CASE(4, 14, nprob, per)
            ATTN_FWD_CASE(4, 15, nprob, per)
            ATTN_FWD_CASE(4, 16, nprob, per)
            default: return DINO_ERR_UNSUPPORTED;
        }
    }
#undef ATTN_FWD_CASE
#undef ATTN_FWD_LAUNCH
    return dino_launch_status();
}

template <typename T>
int bwd_impl(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int Bc,
             int N, int H, const float* rowscale, float* delta_ws, hipStream_t st) {
    const int npad = pad_rows<T>(N);
    const int per_q = 2 * npad * AT<T>::ROWB;
    const int per_kv = per_q + 2 * npad * 4;
    const int nprob = Bc * H;
    const int nsteps = npad / (16 * AT<T>::TPS);
#define ATTN_BWD_LAUNCH(PWV, NSV, GRID, MULT)                                                           \
    {                                                                                                   \
        if (int e = set_smem(attn_bwd_dq_kernel<T, PWV, NSV>, MULT * per_q)) return e;                  \
        if (int e = set_smem(attn_bwd_dkv_kernel<T, PWV, NSV>, MULT * per_kv)) return e;                \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, PWV, NSV>), dim3(GRID), dim3(256), MULT * per_q, st,  \
                           (const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, Bc, N, H, npad, \
                           rowscale, delta_ws);                                                         \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, PWV, NSV>), dim3(GRID), dim3(256), MULT * per_kv, st, \
                           (const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, Bc, N, H, npad, \
                           rowscale, (const float*)delta_ws);                                           \
    }
    constexpr bool FAST = sizeof(T) == 2;  // unrolled step counts for the bf16 path's common shapes
    static const bool small_off = getenv("DINO_ATTN_SMALL") && getenv("DINO_ATTN_SMALL")[0] == '0';
    static const bool fused_off = getenv("DINO_ATTN_FUSED") && getenv("DINO_ATTN_FUSED")[0] == '0';
    if (N <= 64 && !small_off && FAST) {
        // one wave per problem, Q / K / V / dO staged once (see attn_bwd_small_kernel)
        // images of 16-row granularity: 48 rows for the 37-token crops (a quarter less to stage, to keep in
        // registers and to multiply than at 64, and two waves per SIMD instead of one)
        if (N <= 48 && !fused_off) {
            // all four images at once, one wave per problem (two waves per problem: 24 spilled registers at three
            // waves per SIMD, 171 us against 158 us)
            if constexpr (FAST) {
                const int smem = 4 * 48 * AT<bf16>::ROWB + 2 * 48 * 4 + 16;
                if (int e = set_smem(attn_bwd_fused_kernel<48, 64, false>, smem)) return e;
                hipLaunchKernelGGL((attn_bwd_fused_kernel<48, 64, false>), dim3(nprob), dim3(64), smem, st,
                                   (const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, Bc, N, H, rowscale);
            }
        } else if (N <= 48) {
            const int np = 48, smem = 2 * np * AT<T>::ROWB + 2 * np * 4;
            if (int e = set_smem(attn_bwd_small_kernel<T, 3>, smem)) return e;
            hipLaunchKernelGGL((attn_bwd_small_kernel<T, 3>), dim3(nprob), dim3(64), smem, st, (const T*)qkv,
                               (const T*)out, (const T*)dout, lse, (T*)dqkv, Bc, N, H, np, rowscale);
        } else {
            const int smem = 2 * npad * AT<T>::ROWB + 2 * npad * 4;
            if (int e = set_smem(attn_bwd_small_kernel<T, 4>, smem)) return e;
            hipLaunchKernelGGL((attn_bwd_small_kernel<T, 4>), dim3(nprob), dim3(64), smem, st, (const T*)qkv,
                               (const T*)out, (const T*)dout, lse, (T*)dqkv, Bc, N, H, npad, rowscale);
        }
    } else if (N <= 64) {
        if (FAST && nsteps == 2) ATTN_BWD_LAUNCH(1, 2, (nprob + 3) / 4, 4)
        else ATTN_BWD_LAUNCH(1, 0, (nprob + 3) / 4, 4)
    } else if ((per_kv > LDS_LIMIT || N >= stream_min() || N >= 384) && FAST && delta_ws && !stream_off()) {
        // (from about 400 tokens the streamed pair is faster than the whole-problem dq + dkv kernels: 577 tokens,
        // 108 against 188 us; at 257 tokens 157 against 146)
        const int smem_q = 4 * SIMG, smem_kv = 2 * (2 * SIMG + 2 * SCH * 4);
        const int ny = ((N + 15) / 16 + W - 1) / W;
        if (int e = set_smem(attn_bwd_dq_stream_kernel, smem_q)) return e;
        if (int e = set_smem(attn_bwd_dkv_stream_kernel, smem_kv)) return e;
        hipLaunchKernelGGL(attn_bwd_dq_stream_kernel, dim3(nprob * ny), dim3(64 * W), smem_q, st, (const bf16*)qkv,
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, Bc, N, H, ny, rowscale, delta_ws);
        hipLaunchKernelGGL(attn_bwd_dkv_stream_kernel, dim3(nprob * ny), dim3(64 * W), smem_kv, st, (const bf16*)qkv,
                           (const bf16*)dout, lse, (const float*)delta_ws, (bf16*)dqkv, Bc, N, H, ny);
    } else if (per_kv > LDS_LIMIT) {
        constexpr int CH = LongCfg<T>::CH;
        const int smem = 2 * CH * AT<T>::ROWB + 2 * CH * 4;
        const dim3 grid(nprob, ((N + 15) / 16 + 3) / 4);
        if (int e = set_smem(attn_bwd_dq_long_kernel<T>, smem)) return e;
        if (int e = set_smem(attn_bwd_dkv_long_kernel<T>, smem)) return e;
        hipLaunchKernelGGL((attn_bwd_dq_long_kernel<T>), grid, dim3(256), smem, st, (const T*)qkv,
                           (const T*)out, (const T*)dout, lse, (T*)dqkv, Bc, N, H, rowscale);
        hipLaunchKernelGGL((attn_bwd_dkv_long_kernel<T>), grid, dim3(256), smem, st, (const T*)qkv,
                           (const T*)out, (const T*)dout, lse, (T*)dqkv, Bc, N, H, rowscale);
    } else if (FAST && N <= FUSED_ROWS && !fused_off) {
        // every operand once through the memory path (see attn_bwd_fused_kernel)
        if constexpr (FAST) {
            const int grid = nprob < num_cus_attn() ? nprob : num_cus_attn();
#ifdef DINO_ATTN_ONEPASS_EXPERIMENT
            static const bool onepass = getenv("DINO_ATTN_ONEPASS") && getenv("DINO_ATTN_ONEPASS")[0] == '1';
            if (onepass) {
                const int smem1 = 4 * FUSED_ROWS * AT<bf16>::ROWB + 2 * FUSED_ROWS * 4 + OP_DQ_BYTES;
                if (int e = set_smem(attn_bwd_onepass_kernel, smem1)) return e;
                hipLaunchKernelGGL(attn_bwd_onepass_kernel, dim3(grid), dim3(512), smem1, st, (const bf16*)qkv,
                                   (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, Bc, N, H, rowscale);
                return dino_launch_status();
            }
#endif
            const int smem = 4 * FUSED_ROWS * AT<bf16>::ROWB + 2 * FUSED_ROWS * 4 + 16;
            if (int e = set_smem(attn_bwd_fused_kernel<FUSED_ROWS, 512, true>, smem)) return e;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<FUSED_ROWS, 512, true>), dim3(grid), dim3(512), smem, st, (const bf16*)qkv,
                               (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, Bc, N, H, rowscale);
        }
    } else {
        if (FAST && nsteps == 7) ATTN_BWD_LAUNCH(4, 7, nprob, 1)        // 197 tokens (ViT-S/B/16 @ 224)
        else if (FAST && nsteps == 5) ATTN_BWD_LAUNCH(4, 5, nprob, 1)   // 145 tokens (ViT-S/8 @ 96)
        else if (FAST && nsteps == 9) ATTN_BWD_LAUNCH(4, 9, nprob, 1)   // 257 tokens (ViT-L/14 @ 224)
        else ATTN_BWD_LAUNCH(4, 0, nprob, 1)
    }
#undef ATTN_BWD_LAUNCH
    return dino_launch_status();
}

// ---- attention probabilities of one block (Block.forward(return_attention=True),
// vision_transformer.py:131-132,220-226): softmax(q k^T * 64^-1/2) written out as [Bc, H, N, N].
// An inspection / visualisation helper (get_last_selfattention), not on the training path: one
// wave per query row, lane j owns keys j, j+64, ...; plain fp32 dot products.
constexpr int PROBS_MAXJ = 16;   // keys per lane: N <= 1024
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(const T* __restrict__ qkv, T* __restrict__ probs,
                                                         int N, int H) {
    __shared__ float qs[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    const int h = blockIdx.y, b = blockIdx.z;
    const int64_t ld = 3 * (int64_t)H * 64;
    const bool live = i < N;
    if (live) qs[wave][lane] = to_f32(qkv[((int64_t)b * N + i) * ld + h * 64 + lane]);
    __syncthreads();
    if (!live) return;
    float sc[PROBS_MAXJ];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < PROBS_MAXJ; ++t) {
        const int j = t * 64 + lane;
        sc[t] = -INFINITY;
        if (j < N) {
            const T* k = qkv + ((int64_t)b * N + j) * ld + (int64_t)H * 64 + h * 64;
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < 64; d += 8) {
                float kv[8];
                load8<T>(k + d, kv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf(qs[wave][d + e], kv[e], acc);
            }
            sc[t] = acc * 0.125f;
            mx = fmaxf(mx, sc[t]);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < PROBS_MAXJ; ++t) {
        sc[t] = (t * 64 + lane < N) ? expf(sc[t] - mx) : 0.f;
        sum += sc[t];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    T* out = probs + (((int64_t)b * H + h) * N + i) * N;
#pragma unroll
    for (int t = 0; t < PROBS_MAXJ; ++t) {
        const int j = t * 64 + lane;
        if (j < N) out[j] = from_f32<T>(sc[t] * inv);
    }
}

// ---- attention of the CLS query alone -------------------------------------------------------------------------
// VisionTransformer.forward returns x[:, 0] of the LAST block (vision_transformer.py:382-383): in that block only the
// CLS row of attention / proj / MLP is ever consumed, and only the CLS row of their outputs carries a gradient.  So
// the last block's attention has ONE query per (crop, head): scores against every key, softmax, one output row; its
// backward gives dq for that row and dk_j, dv_j for every key.  Memory-bound on K and V; one wave per problem,
// 8 lanes per token row (16-B pieces of a 128-B row: whole rows per request), no MFMA.
template <typename T> DEVI void load8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const bf16x8 x = __builtin_bit_cast(bf16x8, *(const u32x4*)p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
    } else {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = a[e];
            v[4 + e] = b[e];
        }
    }
}
template <typename T> DEVI void store8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 x;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (bf16)v[e];
        *(u32x4*)p = __builtin_bit_cast(u32x4, x);
    } else {
        *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}
DEVI float sum8lanes(float v) {   // over the 8 adjacent lanes of a token row
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}
DEVI float sum_rows(float v) {    // over the 8 row groups of a wave (lanes 8 apart)
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
DEVI float max_rows(float v) {
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// out_cls [Bc, H*64] (times rowscale[b]), lse_cls fp32 [Bc, H]; LDS: 4 waves x npad floats
template <typename T>
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out_cls,
                                                           float* __restrict__ lse_cls, int nprob, int N, int H,
                                                           int npad, const float* __restrict__ rowscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 3, c = lane & 7;
    const int pid = blockIdx.x * 4 + wave;
    if (pid >= nprob) return;
    float* sp = (float*)smem + wave * npad;
    const int b = pid / H, head = pid % H, D = H * HD;
    const int64_t ld = 3 * D;
    const T* base = qkv + (int64_t)b * N * ld + head * HD + c * 8;
    float q[8];
    load8<T>(base, q);
    float mx = -INFINITY;
    for (int j0 = 0; j0 < N; j0 += 8) {
        const int j = j0 + r;
        float s = 0.f;
        if (j < N) {
            float k[8];
            load8<T>(base + (int64_t)j * ld + D, k);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(q[e], k[e], s);
        }
        s = sum8lanes(s) * 0.125f;
        if (j < N) {
            if (c == 0) sp[j] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = max_rows(mx);   // (every lane of a row holds the row's score; rows past N never raised it)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, l = 0.f;
    for (int j0 = 0; j0 < N; j0 += 8) {
        const int j = j0 + r;
        if (j < N) {
            const float pj = __expf(sp[j] - mx);   // (written by this wave's lane c == 0 of the same row group: same wave, in order)
            float v[8];
            load8<T>(base + (int64_t)j * ld + 2 * D, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, v[e], acc[e]);
            l += pj;
        }
    }
    l = sum_rows(l);
    const float sc = (rowscale ? rowscale[b] : 1.f) / l;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = sum_rows(acc[e]) * sc;
    if (r == 0) store8<T>(out_cls + (int64_t)b * D + head * HD + c * 8, acc);
    if (lane == 0) lse_cls[(int64_t)b * H + head] = mx + __logf(l);
}

// dqkv [Bc*N, 3*H*64]: this head's columns of every row -- dq in row 0 (zeros in the other rows), dk_j, dv_j.
// dout_cls [Bc, H*64] already carries the drop-path factor (the contract of dino_attn_bwd).
template <typename T>
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout_cls,
                                                           const float* __restrict__ lse_cls, T* __restrict__ dqkv,
                                                           int nprob, int N, int H, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 3, c = lane & 7;
    const int pid = blockIdx.x * 4 + wave;
    if (pid >= nprob) return;
    float* sp = (float*)smem + wave * 2 * npad;
    float* sdp = sp + npad;
    const int b = pid / H, head = pid % H, D = H * HD;
    const int64_t ld = 3 * D;
    const T* base = qkv + (int64_t)b * N * ld + head * HD + c * 8;
    T* dbase = dqkv + (int64_t)b * N * ld + head * HD + c * 8;
    float q[8], dout[8];
    load8<T>(base, q);
    load8<T>(dout_cls + (int64_t)b * D + head * HD + c * 8, dout);
    const float lse = lse_cls[(int64_t)b * H + head];
    float delta = 0.f;
    for (int j0 = 0; j0 < N; j0 += 8) {
        const int j = j0 + r;
        float s = 0.f, dp = 0.f;
        if (j < N) {
            float k[8], v[8];
            load8<T>(base + (int64_t)j * ld + D, k);
            load8<T>(base + (int64_t)j * ld + 2 * D, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s = fmaf(q[e], k[e], s);
                dp = fmaf(dout[e], v[e], dp);
            }
        }
        s = sum8lanes(s) * 0.125f;
        dp = sum8lanes(dp);
        if (j < N) {
            const float pj = __expf(s - lse);
            if (c == 0) {
                sp[j] = pj;
                sdp[j] = dp;
                delta = fmaf(pj, dp, delta);
            }
        }
    }
    delta = sum_rows(delta);                       // lanes c == 0 of the eight row groups ...
    delta = __shfl(delta, lane & ~7, 64);          // ... handed to the other lanes of their rows
    float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < N; j0 += 8) {
        const int j = j0 + r;
        if (j < N) {
            const float pj = sp[j], ds = pj * (sdp[j] - delta) * 0.125f;
            float k[8], dk[8], dv[8];
            load8<T>(base + (int64_t)j * ld + D, k);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dq[e] = fmaf(ds, k[e], dq[e]);
                dk[e] = ds * q[e];
                dv[e] = pj * dout[e];
            }
            T* row = dbase + (int64_t)j * ld;
            if (j > 0) store8<T>(row, zero);
            store8<T>(row + D, dk);
            store8<T>(row + 2 * D, dv);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] = sum_rows(dq[e]);
    if (r == 0) store8<T>(dbase, dq);
}

}  // namespace

extern "C" int dino_attn_probs(int dtype, const void* qkv, void* probs, int Bc, int N, int H, void* stream) {
    DINO_CHECK_ARG(qkv && probs && Bc > 0 && N > 0 && H > 0 && Bc <= 65535 && H <= 65535);
    if (N > 64 * PROBS_MAXJ) return DINO_ERR_UNSUPPORTED;
    const dim3 grid((N + 3) / 4, H, Bc);
    if (dtype == DINO_BF16)
        hipLaunchKernelGGL(attn_probs_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv,
                           (bf16*)probs, N, H);
    else if (dtype == DINO_F32)
        hipLaunchKernelGGL(attn_probs_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv,
                           (float*)probs, N, H);
    else
        return DINO_ERR_BADARG;
    return dino_launch_status();
}

extern "C" int dino_attn_cls_fwd(int dtype, const void* qkv, void* out_cls, float* lse_cls, int Bc, int N, int H,
                                 const float* rowscale, void* stream) {
    DINO_CHECK_ARG(qkv && out_cls && lse_cls && Bc > 0 && N > 0 && H > 0);
    const int nprob = Bc * H, npad = (N + 7) / 8 * 8, smem = 4 * npad * 4;
    if (smem > 64 * 1024) return DINO_ERR_UNSUPPORTED;
    const dim3 grid((nprob + 3) / 4);
    if (dtype == DINO_BF16)
        hipLaunchKernelGGL(attn_cls_fwd_kernel<bf16>, grid, dim3(256), smem, (hipStream_t)stream, (const bf16*)qkv,
                           (bf16*)out_cls, lse_cls, nprob, N, H, npad, rowscale);
    else if (dtype == DINO_F32)
        hipLaunchKernelGGL(attn_cls_fwd_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, (const float*)qkv,
                           (float*)out_cls, lse_cls, nprob, N, H, npad, rowscale);
    else
        return DINO_ERR_BADARG;
    return dino_launch_status();
}

extern "C" int dino_attn_cls_bwd(int dtype, const void* qkv, const void* dout_cls, const float* lse_cls, void* dqkv,
                                 int Bc, int N, int H, void* stream) {
    DINO_CHECK_ARG(qkv && dout_cls && lse_cls && dqkv && Bc > 0 && N > 0 && H > 0);
    const int nprob = Bc * H, npad = (N + 7) / 8 * 8, smem = 4 * 2 * npad * 4;
    if (smem > 64 * 1024) return DINO_ERR_UNSUPPORTED;
    const dim3 grid((nprob + 3) / 4);
    if (dtype == DINO_BF16)
        hipLaunchKernelGGL(attn_cls_bwd_kernel<bf16>, grid, dim3(256), smem, (hipStream_t)stream, (const bf16*)qkv,
                           (const bf16*)dout_cls, lse_cls, (bf16*)dqkv, nprob, N, H, npad);
    else if (dtype == DINO_F32)
        hipLaunchKernelGGL(attn_cls_bwd_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, (const float*)qkv,
                           (const float*)dout_cls, lse_cls, (float*)dqkv, nprob, N, H, npad);
    else
        return DINO_ERR_BADARG;
    return dino_launch_status();
}

extern "C" int dino_attn_fwd(int dtype, const void* qkv, void* out, float* lse, int Bc, int N, int H,
                             const float* rowscale, void* stream) {
    DINO_CHECK_ARG(qkv && out && lse && Bc > 0 && N > 0 && H > 0);
    if (dtype == DINO_BF16) return fwd_impl<bf16>(qkv, out, lse, Bc, N, H, rowscale, (hipStream_t)stream);
    if (dtype == DINO_F32) return fwd_impl<float>(qkv, out, lse, Bc, N, H, rowscale, (hipStream_t)stream);
    return DINO_ERR_BADARG;
}

extern "C" int dino_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout,
                             const float* lse, void* dqkv, int Bc, int N, int H, const float* rowscale,
                             float* delta_ws, void* stream) {
    DINO_CHECK_ARG(qkv && out && dout && lse && dqkv && Bc > 0 && N > 0 && H > 0);
    if (dtype == DINO_BF16)
        return bwd_impl<bf16>(qkv, out, dout, lse, dqkv, Bc, N, H, rowscale, delta_ws, (hipStream_t)stream);
    if (dtype == DINO_F32)
        return bwd_impl<float>(qkv, out, dout, lse, dqkv, Bc, N, H, rowscale, delta_ws, (hipStream_t)stream);
    return DINO_ERR_BADARG;
}
