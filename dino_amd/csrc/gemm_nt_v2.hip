// Persistent, LDS-DMA pipelined NT GEMM for bf16 (the forward / dgrad work-horse).
//
// Why a second kernel: on the ViT-S shapes (K = 384) the 128x128 tile loop is not MFMA-bound
// but latency-bound -- one K-tile of prefetch per workgroup leaves ~64 KB in flight per CU and
// every output tile restarts the pipeline.  Here
//   * global -> LDS goes through global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass),
//     S stages of 128x32 (A) + 128x32 (B) bf16 = 16 KB each form a ring, S-1 stages in flight;
//   * the workgroup is persistent: it walks its list of (tile, k-split) items as ONE stream of
//     stages, so the next tile's loads are already in flight during an epilogue;
//   * waits are counted (s_waitcnt vmcnt(4*(S-2))) with a raw s_barrier: one barrier per stage,
//     never a full drain in steady state;
//   * the bias vector sits in LDS (a regular VMEM load in the epilogue would make hipcc drain
//     the DMA queue);
//   * LDS rows are 64 B (4 x 16-B chunks), chunk ^= f(row) on the SOURCE address and on the
//     read (LDS-DMA writes lane-linear), conflict-free for ds_read_b128; the B fragment rows are
//     permuted so each lane ends with 16 consecutive output columns (see gemm.hip).
// Two workgroups per CU (72 KB each) let one workgroup's VALU epilogue (GELU) overlap the
// other's MFMA stream.
#include <stdlib.h>

#include "common.h"

namespace {

// 256 x 128 output tile per 256-thread workgroup (2x2 waves of 128x64): 85 FLOP per staged byte.
// Measured: the 128x128 tile (64 FLOP/B) saturates the L2 -> CU path at ~37 GB/s per CU (~620 TF).
constexpr int TM = 256, TN_ = 128, BK = 32, NTHR = 256;
constexpr int MI = TM / 32;                          // 16-row fragments per wave along m
constexpr int NA = TM / 64;                          // A DMA pieces per thread per stage
constexpr int NDMA = NA + 2;                         // DMAs per thread per stage
constexpr int STAGE_BYTES = (TM + TN_) * BK * 2;  // 24 KiB
constexpr int MAX_BIAS = 2048;

DEVI int fsw(int x) { return (4 - (x & 3)) & 3; }  // [0,3,2,1]

template <int N> DEVI void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Item {
    int m0, n0, k0, nkt, split;  // tile origin, first k, number of 32-deep stages, k-split index
};

DEVI Item locate(int w, int tiles_n, int nsplit, int kper, int K) {
    const int tile = w / nsplit, split = w - tile * nsplit;
    Item it;
    it.m0 = (tile / tiles_n) * TM;
    it.n0 = (tile % tiles_n) * TN_;
    it.k0 = split * kper;
    it.split = split;
    const int kend = min(K, it.k0 + kper);
    it.nkt = (kend - it.k0) / BK;
    return it;
}

// LONGB: the bias has more than MAX_BIAS columns (ViT-L's fc1: 4096) and is read from memory in
// the epilogue (L1/L2 hits) instead of from LDS -- its own instantiation, so that the common
// variant's epilogue is untouched.
template <int S, int DBG, bool LONGB = false>
__global__ __launch_bounds__(NTHR, 2) void gemm_nt_bf16_v2(const dino_gemm_args p, int tiles_n,
                                                           int nwork, int nsplit, int kper) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    float* sbias = (float*)(smem + S * STAGE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 15, g = lane >> 4;
    const int G = gridDim.x;
    const int b = xcd_remap(blockIdx.x, G);

    if (p.bias && !LONGB) {
        for (int n = tid; n < p.N; n += NTHR) sbias[n] = p.bias[n];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int nmine = b < nwork ? (nwork - b + G - 1) / G : 0;
    if (nmine == 0) return;

    // ---- issue side: buffer_load ... lds with a per-thread CONSTANT voffset and a scalar
    // soffset = (tile row origin * ld + k) * 2, so a stage costs no address VALU at all.
    // Rows beyond M / N fall outside the descriptor's range and are dropped by the hardware
    // range check (their accumulator rows / columns are never stored).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.B, 0, (int)(((int64_t)(p.N - 1) * p.ldb + p.K) * 2), 0x00020000);
    int voffA[NA], voffB[2];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int L = tid + NTHR * u, r = L >> 2, c = L & 3;
        voffA[u] = (r * (int)p.lda + ((c ^ fsw(r >> 2)) << 3)) * 2;
        if (u < 2) voffB[u] = (r * (int)p.ldb + ((c ^ fsw(r >> 4)) << 3)) * 2;
    }
    int ij = 0, ikt = 0;                       // item / k-stage being issued
    Item iit = locate(b, tiles_n, nsplit, kper, p.K);
    int islot = 0;
    auto issue = [&]() {
        if (DBG == 1) return;   // timing experiment: no DMA
        const int sa = islot * STAGE_BYTES + wv * 1024;   // wave-uniform LDS byte offsets
        const int k = iit.k0 + ikt * BK;
        const int soA = (iit.m0 * (int)p.lda + k) * 2, soB = (iit.n0 * (int)p.ldb + k) * 2;
#pragma unroll
        for (int u = 0; u < NA; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + sa + u * 4096), 16, voffA[u],
                                                     soA, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(smem + sa + TM * BK * 2 + u * 4096),
                                                     16, voffB[u], soB, 0, 0);
        islot = islot + 1 == S ? 0 : islot + 1;
        if (++ikt == iit.nkt) {
            ikt = 0;
            ++ij;
            if (ij < nmine) iit = locate(b + ij * G, tiles_n, nsplit, kper, p.K);
        }
    };

    int total = 0;
    for (int j = 0; j < nmine; ++j) total += locate(b + j * G, tiles_n, nsplit, kper, p.K).nkt;

    int issued = 0;
#pragma unroll
    for (int q = 0; q < S - 1; ++q) {
        if (issued < total) {
            issue();
            ++issued;
        }
    }

    // ---- consume side -----------------------------------------------------------------
    f32x4 acc[MI][4];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cj = 0, ckt = 0, cslot = 0;
    Item cit = locate(b, tiles_n, nsplit, kper, p.K);

    // fragment byte offsets inside a stage: one VGPR per operand, the four fragments are at
    // constant strides (1024 B for A rows +16, 256 B for the permuted B rows +4)
    const int ra0 = wm * (TM / 2) + i, rb0 = wn * 64 + 16 * (i >> 2) + (i & 3);
    const int offA = ra0 * 64 + ((g ^ fsw(ra0 >> 2)) << 4);
    const int offB = TM * BK * 2 + rb0 * 64 + ((g ^ fsw(rb0 >> 4)) << 4);

    for (int s = 0; s < total; ++s) {
        // stage s has landed once at most (S-2) younger stages (NDMA DMAs each) are outstanding
        if (total - 1 - s >= S - 2) wait_vmcnt<NDMA * (S - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < total) {  // refill the slot consumed in the previous iteration
            issue();
            ++issued;
        }
        const char* st = smem + cslot * STAGE_BYTES;
        u32x4 fa[MI], fb[4];
        if (DBG == 3) {  // timing experiment: no LDS reads
#pragma unroll
            for (int t = 0; t < MI; ++t) fa[t] = u32x4{(unsigned)s, 1u, 2u, 3u};
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[t] = u32x4{(unsigned)s, 1u, 2u, 3u};
        } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = *(const u32x4*)(st + offB + t * 256);
#pragma unroll
        for (int t = 0; t < MI; ++t) fa[t] = *(const u32x4*)(st + offA + t * 1024);
        }
        if (DBG == 2) {  // timing experiment: no MFMA (keep the LDS reads alive)
#pragma unroll
            for (int t = 0; t < MI; ++t) asm volatile("" ::"v"(fa[t]));
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(fb[t]));
        } else
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, fb[ni]), __builtin_bit_cast(bf16x8, fa[mi]),
                    acc[mi][ni], 0, 0, 0);
        cslot = cslot + 1 == S ? 0 : cslot + 1;
        if (++ckt == cit.nkt) {
            // ---- epilogue of this (tile, split) -------------------------------------------
            bf16* outT = (bf16*)p.out;
            float* outF = (float*)p.out;
            const bf16* res = (const bf16*)p.residual;
            bf16* aux = (bf16*)p.aux;
            float csum[2][8];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 8; ++r) csum[c][r] = 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = cit.m0 + wm * (TM / 2) + mi * 16 + i;
                if (m < p.M) {
                    const float rs = p.rowscale ? p.rowscale[m / p.rows_per_group] : 1.f;
                    const int64_t orow = p.row_map_n ? (int64_t)m + m / p.row_map_n + 1 : m;
                    const int64_t rrow =
                        p.res_mod ? m % p.res_mod : (p.row_map_n ? 1 + m % p.row_map_n : m);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int n = cit.n0 + wn * 64 + 16 * g + 8 * c;
                        if (n < p.N) {
                            float v[8];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = acc[mi][2 * c][r];
                                v[4 + r] = acc[mi][2 * c + 1][r];
                            }
                            if (p.bias) {
                                const float bs = p.bias_rowscale ? rs : 1.f;
                                const float* bsrc = LONGB ? p.bias : sbias;
                                const f32x4 b0 = *(const f32x4*)(bsrc + n), b1 = *(const f32x4*)(bsrc + n + 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    v[r] += b0[r] * bs;
                                    v[4 + r] += b1[r] * bs;
                                }
                            }
                            if (p.act == 1) {
                                if (aux) store8<bf16>(aux + (int64_t)m * p.ldaux + n, v);
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] = gelu_t<bf16>(v[r]);
                            } else if (p.act == 2) {
                                float pre[8];
                                load8<bf16>(aux + (int64_t)m * p.ldaux + n, pre);
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] *= gelu_grad_t<bf16>(pre[r]);
                            } else if (p.act == 3) {
                                float dg[8];
#pragma unroll
                                for (int r = 0; r < 8; ++r) gelu_both_t<bf16>(v[r], v[r], dg[r]);
                                if (aux) store8<bf16>(aux + (int64_t)m * p.ldaux + n, dg);
                            } else if (p.act == 4) {
                                float dg[8];
                                load8<bf16>(aux + (int64_t)m * p.ldaux + n, dg);
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] *= dg[r];
                            }
                            if (p.out_rowscale) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] *= rs;
                            }
                            if (res) {
                                float rv[8];
                                load8<bf16>(res + rrow * p.ldres + n, rv);
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] += rv[r];
                            }
                            if (p.colsum) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) csum[c][r] += v[r];
                            }
                            if (p.out_mode == 0) {
                                store8<bf16>(outT + orow * p.ldo + n, v);
                            } else if (p.out_mode == 1) {
                                store8<float>(outF + orow * p.ldo + n, v);
                            } else if (p.out_mode == 3) {  // split s owns slab s: plain stores
                                store8<float>(outF + ((int64_t)cit.split * p.M + orow) * p.ldo + n, v);
                            } else {
#pragma unroll
                                for (int r = 0; r < 8; ++r) atomicAdd(outF + orow * p.ldo + n + r, v[r]);
                            }
                        }
                    }
                }
            }
            if (p.colsum) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int n = cit.n0 + wn * 64 + 16 * g + 8 * c;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float x = csum[c][r];
                        x = row16_sum(x);
                        if (i == 0 && n + r < p.N) atomicAdd(p.colsum + n + r, x);
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            ckt = 0;
            ++cj;
            if (cj < nmine) cit = locate(b + cj * G, tiles_n, nsplit, kper, p.K);
        }
        // my LDS reads of this slot are complete before any wave can pass the next barrier and
        // re-fill it by DMA
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

// Internal entry (same translation-unit family as dino_gemm, which validates the arguments).
bool dino_gemm_nt_v2_supported(const dino_gemm_args& a) {
    const int64_t ba = ((int64_t)a.M * a.lda + a.K) * 2, bb = ((int64_t)a.N * a.ldb + a.K) * 2;
    return a.dtype == DINO_BF16 && !a.trans && a.K % BK == 0 && (!a.bias || a.N <= MAX_BIAS || a.N % 8 == 0) &&
           ba < (1ll << 31) && bb < (1ll << 31);
}

template <int S, int DBG, bool LONGB = false>
static int launch_v2(const dino_gemm_args& a, hipStream_t st) {
    const int smem = S * STAGE_BYTES + MAX_BIAS * 4;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_bf16_v2<S, DBG, LONGB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int tiles_m = (a.M + TM - 1) / TM, tiles_n = (a.N + TN_ - 1) / TN_;
    const int splitk = a.splitk > 1 ? a.splitk : 1;
    int kper = (a.K + splitk - 1) / splitk;
    kper = (kper + BK - 1) / BK * BK;
    const int nsplit = (a.K + kper - 1) / kper;
    const int nwork = tiles_m * tiles_n * nsplit;
    if (a.out_mode == 3 && nsplit < splitk) {  // slabs no split writes read as zeros
        hipError_t e = hipMemsetAsync((float*)a.out + (int64_t)nsplit * a.M * a.ldo, 0,
                                      (size_t)(splitk - nsplit) * a.M * a.ldo * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    const int grid = nwork < 2 * num_cus() ? nwork : 2 * num_cus();
    hipLaunchKernelGGL((gemm_nt_bf16_v2<S, DBG, LONGB>), dim3(grid), dim3(NTHR), smem, st, a, tiles_n, nwork,
                       nsplit, kper);
    return dino_launch_status();
}

int dino_gemm_nt_v2(const dino_gemm_args& a, hipStream_t st) {
    static const int dbg = getenv("DINO_GEMM_DBG") ? atoi(getenv("DINO_GEMM_DBG")) : 0;
    if (a.bias && a.N > MAX_BIAS) return launch_v2<3, 0, true>(a, st);
    switch (dbg) {
        case 1: return launch_v2<3, 1>(a, st);
        case 2: return launch_v2<3, 2>(a, st);
        case 3: return launch_v2<3, 3>(a, st);
        default: return launch_v2<3, 0>(a, st);
    }
}
