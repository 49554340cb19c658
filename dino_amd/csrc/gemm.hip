// MFMA GEMM family with fused epilogues (see include/dino_hip.h: dino_gemm).
//
// One kernel template <T, TN>: 128x128 output tile per 256-thread workgroup (2x2 waves, each
// wave 64x64 = 4x4 MFMA 16x16 fragments), K-tile of 128 bytes per row (64 bf16 / 32 f32),
// register-staged global->LDS double buffering with ONE barrier per K-tile (issue the next
// tile's global loads before the MFMA block, write them to the other LDS buffer after it).
//   bf16: v_mfma_f32_16x16x32_bf16     f32: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain)
// NT tiles live in LDS as [row][8 x 16-B chunk] with chunk ^= (row & 7) (conflict-free
// ds_read_b128 fragment reads).  TN tiles (wgrad: contraction over the token dim) keep the
// global [k][m] orientation with a padded row stride and are consumed with
// ds_read_b64_tr_b16 (bf16) / scalar ds_read_b32 (f32) -- no transposing stores.
// The MFMA operands are swapped (first = B fragment, second = A fragment) so each lane ends
// up with 4 CONSECUTIVE n of one output row m: 8/16-byte epilogue accesses.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

template <typename T> struct GT;
template <> struct GT<bf16> {
    static constexpr int EPC = 8;      // elements per 16-byte chunk
    static constexpr int BK = 64;      // K-tile (128 bytes per NT row)
    static constexpr int TNROW = 288;  // TN LDS row stride in bytes (256 + 32: tr-read conflict-free)
};
template <> struct GT<float> {
    static constexpr int EPC = 4;
    static constexpr int BK = 32;
    static constexpr int TNROW = 528;  // 512 + 16
};

template <typename T, bool TN> struct TileBytes {
    static constexpr int value = TN ? GT<T>::BK * GT<T>::TNROW : BM * 128;
};

DEVI void mma_frag(const u32x4& x, const u32x4& y, f32x4& acc, bf16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x),
                                                  __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0);
}
DEVI void mma_frag(const u32x4& x, const u32x4& y, f32x4& acc, float) {
    const f32x4 xf = __builtin_bit_cast(f32x4, x), yf = __builtin_bit_cast(f32x4, y);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[e], yf[e], acc, 0, 0, 0);
}

// Fragment of 16 rows starting at `rbase` for K-chunk c = g + 4h of an NT tile.
DEVI u32x4 frag_nt(const char* tile, int rbase, int i, int c) {
    const int row = rbase + i;
    return *(const u32x4*)(tile + row * 128 + ((c ^ (row & 7)) << 4));
}
// B-side NT fragment.  Operand row rho = i of fragment ni is tile row 16*(i>>2) + 4*ni + (i&3)
// (instead of 16*ni + i): after the MFMA each lane then owns 16 CONSECUTIVE output columns
// n = 16*g + 4*ni + r of its row m, i.e. two 16-byte bf16 (four fp32) epilogue accesses.
// The swizzle key keeps the permuted ds_read_b128 conflict-free.
DEVI int bkey(int row) { return ((row >> 1) & 1) + 2 * ((row >> 4) & 3); }
DEVI u32x4 frag_nt_b(const char* tile, int rbase, int ni, int i, int c) {
    const int row = rbase + 16 * (i >> 2) + 4 * ni + (i & 3);
    return *(const u32x4*)(tile + row * 128 + ((c ^ bkey(row)) << 4));
}
// Same fragment from a TN tile ([k][m] orientation).
DEVI u32x4 frag_tn(const char* tile, int cbase, int i, int g, int h, bf16) {
    // lane 4q+p of each 16-lane group addresses row (k0 + q), columns cbase + 4p .. +3;
    // lane i of the group receives column cbase + i of rows k0 .. k0+3.
    const int q = i >> 2, p = i & 3;
    const int k0 = 32 * h + 8 * g;
    const char* a0 = tile + (k0 + q) * GT<bf16>::TNROW + (cbase + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * GT<bf16>::TNROW));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r = {l2[0], l2[1], h2[0], h2[1]};
    return r;
}
DEVI u32x4 frag_tn(const char* tile, int cbase, int i, int g, int h, float) {
    const int c = g + 4 * h;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        r[e] = *(const float*)(tile + (4 * c + e) * GT<float>::TNROW + (cbase + i) * 4);
    return __builtin_bit_cast(u32x4, r);
}

template <typename T, bool TN>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const dino_gemm_args p, int tiles_n, int kper) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = GT<T>::EPC, BK = GT<T>::BK, TB = TileBytes<T, TN>::value;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 15, g = lane >> 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int kbeg = blockIdx.y * kper;
    const int kend = min(p.K, kbeg + kper);
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ B = (const T*)p.B;

    u32x4 ra[4], rb[4];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + NT * u;
            if (!TN) {
                const int row = q >> 3, c = q & 7;
                const int gk = k0 + c * EPC;
                const bool kin = gk < kend;
                ra[u] = (kin && m0 + row < p.M) ? *(const u32x4*)(A + (int64_t)(m0 + row) * p.lda + gk) : zero4;
                rb[u] = (kin && n0 + row < p.N) ? *(const u32x4*)(B + (int64_t)(n0 + row) * p.ldb + gk) : zero4;
            } else {
                constexpr int CPR = 128 / EPC;
                const int kk = q / CPR, mc = q % CPR;
                const int gk = k0 + kk;
                const bool kin = gk < kend;
                ra[u] = (kin && m0 + mc * EPC < p.M) ? *(const u32x4*)(A + (int64_t)gk * p.lda + m0 + mc * EPC) : zero4;
                rb[u] = (kin && n0 + mc * EPC < p.N) ? *(const u32x4*)(B + (int64_t)gk * p.ldb + n0 + mc * EPC) : zero4;
            }
        }
    };
    auto sstore = [&](int stage) {
        char* sa = smem + stage * 2 * TB;
        char* sb = sa + TB;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + NT * u;
            int off, offb;
            if (!TN) {
                const int row = q >> 3, c = q & 7;
                off = row * 128 + ((c ^ (row & 7)) << 4);
                offb = row * 128 + ((c ^ bkey(row)) << 4);
            } else {
                constexpr int CPR = 128 / EPC;
                off = offb = (q / CPR) * GT<T>::TNROW + (q % CPR) * 16;
            }
            *(u32x4*)(sa + off) = ra[u];
            *(u32x4*)(sb + offb) = rb[u];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kbeg + (kt + 1) * BK);
        const char* sa = smem + (kt & 1) * 2 * TB;
        const char* sb = sa + TB;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!TN) {
                    fa[t] = frag_nt(sa, wm * 64 + t * 16, i, g + 4 * h);
                    fb[t] = frag_nt_b(sb, wn * 64, t, i, g + 4 * h);
                } else {
                    fa[t] = frag_tn(sa, wm * 64 + t * 16, i, g, h, T());
                    fb[t] = frag_tn(sb, wn * 64 + t * 16, i, g, h, T());
                }
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) mma_frag(fb[ni], fa[mi], acc[mi][ni], T());
        }
        if (more) sstore((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------
    // NT: lane holds, for row m = .. + i, the 16 consecutive columns n = .. + 16g + (0..15)
    //     (acc[mi][ni][r] <-> column 4*ni + r): processed as two 8-wide chunks.
    // TN: lane holds 4 consecutive columns n = .. + 16*ni + 4g + (0..3) per fragment.
    constexpr int W = TN ? 4 : 8;        // chunk width
    constexpr int NCHK = TN ? 4 : 2;     // chunks per (lane, mi)
    T* outT = (T*)p.out;
    float* outF = (float*)p.out;
    const T* res = (const T*)p.residual;
    T* aux = (T*)p.aux;
    float csum[NCHK][W];
#pragma unroll
    for (int c = 0; c < NCHK; ++c)
#pragma unroll
        for (int r = 0; r < W; ++r) csum[c][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + i;
        if (m >= p.M) continue;
        const float rs = p.rowscale ? p.rowscale[m / p.rows_per_group] : 1.f;
        const int64_t orow = p.row_map_n ? (int64_t)m + m / p.row_map_n + 1 : m;
        const int64_t rrow = p.res_mod ? m % p.res_mod : (p.row_map_n ? 1 + m % p.row_map_n : m);
#pragma unroll
        for (int c = 0; c < NCHK; ++c) {
            const int n = TN ? n0 + wn * 64 + c * 16 + 4 * g : n0 + wn * 64 + 16 * g + 8 * c;
            if (n >= p.N) continue;
            float v[W];
            if (TN) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mi][c][r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[mi][2 * c][r];
                    v[(4 + r) % W] = acc[mi][(2 * c + 1) % 4][r];
                }
            }
            if (p.bias) {
                float b[W];
                loadw<float, W>(p.bias + n, b);
                const float bs = p.bias_rowscale ? rs : 1.f;
#pragma unroll
                for (int r = 0; r < W; ++r) v[r] += b[r] * bs;
            }
            if (p.act == 1) {
                if (aux) storew<T, W>(aux + (int64_t)m * p.ldaux + n, v);
#pragma unroll
                for (int r = 0; r < W; ++r) v[r] = gelu_t<T>(v[r]);
            } else if (p.act == 2) {
                float pre[W];
                loadw<T, W>(aux + (int64_t)m * p.ldaux + n, pre);
#pragma unroll
                for (int r = 0; r < W; ++r) v[r] *= gelu_grad_t<T>(pre[r]);
            } else if (p.act == 3) {
                float dg[W];
#pragma unroll
                for (int r = 0; r < W; ++r) gelu_both_t<T>(v[r], v[r], dg[r]);
                if (aux) storew<T, W>(aux + (int64_t)m * p.ldaux + n, dg);
            } else if (p.act == 4) {
                float dg[W];
                loadw<T, W>(aux + (int64_t)m * p.ldaux + n, dg);
#pragma unroll
                for (int r = 0; r < W; ++r) v[r] *= dg[r];
            }
            if (p.out_rowscale) {
#pragma unroll
                for (int r = 0; r < W; ++r) v[r] *= rs;
            }
            if (res) {
                float rv[W];
                loadw<T, W>(res + rrow * p.ldres + n, rv);
#pragma unroll
                for (int r = 0; r < W; ++r) v[r] += rv[r];
            }
            if (p.colsum) {
#pragma unroll
                for (int r = 0; r < W; ++r) csum[c][r] += v[r];
            }
            if (p.out_mode == 0) {
                storew<T, W>(outT + orow * p.ldo + n, v);
            } else if (p.out_mode == 1) {
                storew<float, W>(outF + orow * p.ldo + n, v);
            } else {
#pragma unroll
                for (int r = 0; r < W; ++r) atomicAdd(outF + orow * p.ldo + n + r, v[r]);
            }
        }
    }
    if (p.colsum) {
        // column sums of the stored tile (bias gradient of the layer that produced A's partner):
        // reduce over the 16 lanes that share g (xor 1,2,4,8), one atomic per column per wave
#pragma unroll
        for (int c = 0; c < NCHK; ++c) {
            const int n = TN ? n0 + wn * 64 + c * 16 + 4 * g : n0 + wn * 64 + 16 * g + 8 * c;
#pragma unroll
            for (int r = 0; r < W; ++r) {
                float x = csum[c][r];
                x = row16_sum(x);
                if (i == 0 && n + r < p.N) atomicAdd(p.colsum + n + r, x);
            }
        }
    }
}

template <typename T, bool TN>
int launch(const dino_gemm_args& a, hipStream_t st) {
    constexpr int BK = GT<T>::BK;
    constexpr int smem = 4 * TileBytes<T, TN>::value;
    static bool configured = false;  // benign race: idempotent attribute set
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<T, TN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int splitk = a.splitk > 1 ? a.splitk : 1;
    int kper = (a.K + splitk - 1) / splitk;
    kper = (kper + BK - 1) / BK * BK;
    const int ny = (a.K + kper - 1) / kper;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)ny);
    hipLaunchKernelGGL((gemm_kernel<T, TN>), grid, dim3(NT), smem, st, a, tiles_n, kper);
    return dino_launch_status();
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, int nslab, int64_t n4,
                                                        float* __restrict__ out, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = accumulate ? ((const f32x4*)out)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= nslab; s += 4) {  // four loads in flight
        const f32x4 v0 = ((const f32x4*)slabs)[(int64_t)s * n4 + i], v1 = ((const f32x4*)slabs)[(int64_t)(s + 1) * n4 + i];
        const f32x4 v2 = ((const f32x4*)slabs)[(int64_t)(s + 2) * n4 + i], v3 = ((const f32x4*)slabs)[(int64_t)(s + 3) * n4 + i];
        a += (v0 + v1) + (v2 + v3);
    }
    for (; s < nslab; ++s) a += ((const f32x4*)slabs)[(int64_t)s * n4 + i];
    ((f32x4*)out)[i] = a;
}

}  // namespace

bool dino_gemm_nt_v2_supported(const dino_gemm_args& a);
int dino_gemm_nt_v2(const dino_gemm_args& a, hipStream_t st);
bool dino_gemm_nt_v3_supported(const dino_gemm_args& a);
int dino_gemm_nt_v3(const dino_gemm_args& a, hipStream_t st);
bool dino_gemm_tn_v2_supported(const dino_gemm_args& a);
int dino_gemm_tn_v2(const dino_gemm_args& a, hipStream_t st);
int dino_gemm_tn_group_prefix(const dino_gemm_args* a, int count);
int dino_gemm_tn_group_v2(const dino_gemm_args* a, int count, hipStream_t st);

extern "C" int dino_gemm(const dino_gemm_args* args, void* stream) {
    DINO_CHECK_ARG(args != nullptr);
    const dino_gemm_args& a = *args;
    DINO_CHECK_ARG(a.dtype == DINO_F32 || a.dtype == DINO_BF16);
    DINO_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0);
    DINO_CHECK_ARG(a.A && a.B && a.out);
    DINO_CHECK_ARG(aligned16(a.A) && aligned16(a.B) && aligned16(a.out));
    DINO_CHECK_ARG(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldo % 8 == 0 && a.N % 8 == 0);
    if (a.trans) {
        DINO_CHECK_ARG(a.M % 8 == 0 && a.N % 8 == 0);
    } else {
        DINO_CHECK_ARG(a.K % 8 == 0);
    }
    DINO_CHECK_ARG(a.out_mode >= 0 && a.out_mode <= 3);
    DINO_CHECK_ARG(a.act >= 0 && a.act <= 4);
    if (a.act == 2 || a.act == 4) DINO_CHECK_ARG(a.aux != nullptr);
    if (a.aux) DINO_CHECK_ARG(aligned16(a.aux) && a.ldaux % 8 == 0);
    if (a.residual) DINO_CHECK_ARG(aligned16(a.residual) && a.ldres % 8 == 0);
    if (a.colsum && !a.trans) DINO_CHECK_ARG(a.splitk <= 1);
    if (a.bias) DINO_CHECK_ARG(aligned16(a.bias));
    if (a.rowscale) DINO_CHECK_ARG(a.rows_per_group > 0);
    if (a.splitk > 1)
        DINO_CHECK_ARG((a.out_mode == 2 || a.out_mode == 3) && !a.bias && a.act == 0 && !a.residual);
    hipStream_t st = (hipStream_t)stream;
    static const bool use_v2 = getenv("DINO_GEMM_V1") == nullptr;
    static const bool use_v3 = getenv("DINO_GEMM_NO_V3") == nullptr;
    // split-K slabs (out_mode 3) exist in the bf16 NT LDS-DMA kernel only: never let another kernel take the
    // call (the register-staged kernel's epilogue would add every split into slab 0 of an uninitialised buffer)
    if (a.out_mode == 3) {
        DINO_CHECK_ARG(!a.trans && !a.row_map_n);
        if (!use_v2 || !dino_gemm_nt_v2_supported(a)) return DINO_ERR_UNSUPPORTED;
        return dino_gemm_nt_v2(a, st);
    }
    if (use_v2 && use_v3 && dino_gemm_nt_v3_supported(a)) return dino_gemm_nt_v3(a, st);
    if (a.prescale || a.ln_stats || a.ln_stats_in) return DINO_ERR_UNSUPPORTED;   // (the folded-LayerNorm epilogues: wide tile only)
    if (use_v2 && dino_gemm_nt_v2_supported(a)) {
        // Small problems (the projection head, the last block's CLS rows: M = crops) are latency-bound, not
        // bandwidth-bound: with fewer than ~160 of its 256 x 128 tiles the LDS-DMA kernel leaves most CUs idle, and the
        // 128 x 128 register-staged kernel below is 20-35 % faster (tools/bench_small_gemm.py: M = 3072, N = 384,
        // K = 1536 38 -> 29 us; N = 256, K = 2048 49 -> 36 us).  DINO_V2_MIN_ITEMS overrides the threshold.
        static const int64_t min_items = getenv("DINO_V2_MIN_ITEMS") ? atoi(getenv("DINO_V2_MIN_ITEMS")) : 160;
        const int64_t items = (int64_t)((a.M + 255) / 256) * ((a.N + 127) / 128) * (a.splitk > 1 ? a.splitk : 1);
        if (items >= min_items) return dino_gemm_nt_v2(a, st);
    }
    if (use_v2 && dino_gemm_tn_v2_supported(a)) return dino_gemm_tn_v2(a, st);
    if (a.trans && a.colsum) {
        // TN: colsum[m] += sum_k A[k, m].  Only the LDS-DMA kernel fuses it; here it is its own pass.
        if (int e = dino_colsum(a.dtype, a.A, a.lda, a.colsum, a.K, a.M, nullptr, 0, stream)) return e;
        dino_gemm_args b = a;
        b.colsum = nullptr;
        if (a.dtype == DINO_BF16) return launch<bf16, true>(b, st);
        return launch<float, true>(b, st);
    }
    if (a.dtype == DINO_BF16)
        return a.trans ? launch<bf16, true>(a, st) : launch<bf16, false>(a, st);
    return a.trans ? launch<float, true>(a, st) : launch<float, false>(a, st);
}

extern "C" int dino_gemm_wide_tile_supported(const dino_gemm_args* args) {
    if (args == nullptr) return 0;
    static const bool on = getenv("DINO_GEMM_V1") == nullptr && getenv("DINO_GEMM_NO_V3") == nullptr;
    const dino_gemm_args& a = *args;
    if (!on || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % 8 != 0 || a.N % 8 != 0) return 0;
    return dino_gemm_nt_v3_supported(a) ? 1 : 0;
}

extern "C" int dino_sum_slabs(const float* slabs, int nslab, int64_t n, float* out, int accumulate, void* stream) {
    DINO_CHECK_ARG(slabs && out && nslab > 0 && n > 0 && n % 4 == 0);
    DINO_CHECK_ARG(aligned16(slabs) && aligned16(out));
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs,
                       nslab, n4, out, accumulate);
    return dino_launch_status();
}

// `count` independent GEMMs, same result as `count` dino_gemm calls.  Weight-gradient (TN, split
// over the tokens) problems that share the token dimension run as one launch, so the split-K
// machinery fills the chip once for all of them (see gemm_tn_v2.hip).
extern "C" int dino_gemm_group(const dino_gemm_args* args, int count, void* stream) {
    DINO_CHECK_ARG(args != nullptr && count > 0);
    static const bool use_v2 = getenv("DINO_GEMM_V1") == nullptr;
    static const bool use_group = getenv("DINO_GEMM_NO_GROUP") == nullptr;
    int q = 0;
    while (q < count) {
        int n = 0;
        if (use_v2 && use_group) {
            n = dino_gemm_tn_group_prefix(args + q, count - q);
            for (int j = 0; j < n; ++j) {  // the checks dino_gemm would make
                const dino_gemm_args& a = args[q + j];
                if (!(a.A && a.B && a.out && aligned16(a.A) && aligned16(a.B) && aligned16(a.out) &&
                      a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldo % 8 == 0 && a.M > 0 && a.N > 0 && a.K > 0))
                    n = 0;
            }
        }
        if (n >= 2) {
            if (int e = dino_gemm_tn_group_v2(args + q, n, (hipStream_t)stream)) return e;
            q += n;
        } else {
            if (int e = dino_gemm(args + q, stream)) return e;
            q += 1;
        }
    }
    return DINO_OK;
}
