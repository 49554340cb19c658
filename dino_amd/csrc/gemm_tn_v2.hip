// LDS-DMA pipelined TN GEMM for bf16 weight gradients: out[m,n] += sum_k A[k,m] * B[k,n]
// (A = dY [tokens, out_f], B = X [tokens, in_f]; the contraction runs over ~1e5 tokens).
//
// The 128x128 register-staged kernel (gemm.hip) is L2->CU bound on these shapes: every token
// panel of dY is streamed once per 128-column tile of X and vice versa (64 FLOP per staged
// byte).  Here one 512-thread workgroup (8 waves) owns a 128x384 (or 384x128) output tile, so
// the 384-wide operand -- the ViT-S embed dim -- is streamed exactly once per tile row and the
// staged traffic drops by a third (96 FLOP/B); tokens are split over workgroups and the
// partial tiles go to a workspace slab with plain stores, summed by a second tiny kernel
// (no float atomics on the critical path, bitwise reproducible).
//   * stages of 32 tokens: A [32][TM] + B [32][TN] bf16 = 32 KB, a ring of four (all requested up front),
//     buffer_load ... lds with constant per-lane voffset + scalar soffset, counted vmcnt;
//   * the MFMA waves read the fragments of stage s + 1 while they multiply stage s, into the registers whose
//     fragments have just had their last use (round 3: -0.3 ms per step);
//   * tiles keep the global [k][col] orientation; MFMA fragments are read with
//     ds_read_b64_tr_b16; 16-byte chunks are XOR-swizzled with a key of the token row so the
//     eight rows a half-wave touches hit disjoint banks (rows are multiples of 256 B).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

#ifndef TN_S
#define TN_S 4   // ring depth (stages of 32 KB for the 128 x 384 tile; 5 = the whole 160 KB of LDS)
#endif
constexpr int BK = 32, NTHR = 768, S = TN_S;
[[maybe_unused]] constexpr int NCONS = 512, NLOAD = 4;   // (device pass only: 8 MFMA waves + 4 loader waves)

DEVI int tkey(int row) { return 2 * ((row & 3) | (((row >> 3) & 1) << 2)); }

// Up to MAXG weight-gradient problems that share the token dimension run as ONE launch: the split
// over tokens only has to fill the chip once for all of them, so the partial-tile slabs (256
// tiles of 196 KB written and read back per launch, ~40 us) are paid once per transformer block
// instead of once per linear layer.
constexpr int MAXG = 4;
struct TnProb {
    const void* A;
    const void* B;
    float* out;
    float* colsum;            // nullable, fp32 [M]: += sum_k A[k, m] (the bias gradient beside dW)
    int lda, ldb, ldo, M, N;
    int tile_begin, tiles_n;  // first global tile index of this problem; its tiles per row
};
struct TnGroup {
    TnProb prob[MAXG];
    int count, K;
};

template <int N> DEVI void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WM x WN waves, each MI x NI fragments of 16x16:  tile = (WM*MI*16) x (WN*NI*16)
template <int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(NTHR) void gemm_tn_bf16_v2(const TnGroup grp, float* __restrict__ slab,
                                                           int nsplit, int kper) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
    constexpr int TM = WM * MI * 16, TN = WN * NI * 16;
    constexpr int ROWA = TM * 2, ROWB = TN * 2;              // LDS row bytes
    constexpr int ABYTES = BK * ROWA, BBYTES = BK * ROWB, STAGE = ABYTES + BBYTES;
    constexpr int PA = ABYTES / 1024, PB = BBYTES / 1024;    // 1-KiB DMA pieces per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups that read the same token range (same split, different output tile) should share
    // an XCD's L2: logical id = contiguous chunk per XCD, tile index fastest.
    const int ntile_ = gridDim.x / nsplit;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid / ntile_, tile = lid - split * ntile_;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < MAXG; ++q)
        if (q < grp.count && tile >= grp.prob[q].tile_begin) pi = q;
    const TnProb p = grp.prob[pi];
    const int ltile = tile - p.tile_begin;
    const int m0 = (ltile / p.tiles_n) * TM, n0 = (ltile % p.tiles_n) * TN;
    const int kbeg = split * kper, kend = min(grp.K, kbeg + kper);
    // a ragged last stage reads token rows >= K: outside both descriptors, so the DMA writes zeros
    const int nst = (kend - kbeg + BK - 1) / BK;

    if (wv >= NCONS / 64) {
        // ===== loader waves (one per SIMD): nothing but buffer_load ... lds and counted waits =====
        constexpr int NP = (PA + PB) / NLOAD;  // pieces per loader per stage
        static_assert((PA + PB) % NLOAD == 0, "pieces must split evenly over the loaders");
        const int w = wv - NCONS / 64;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.A, 0, (int)(((int64_t)(grp.K - 1) * p.lda + p.M) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.B, 0, (int)(((int64_t)(grp.K - 1) * p.ldb + p.N) * 2), 0x00020000);
        // piece q (0 .. PA+PB-1) = 64 consecutive 16-B chunks of the stage image; loader w moves
        // pieces w*NP .. w*NP+NP-1.  Per-lane source offset is constant for the whole kernel.
        int voff[NP];
        bool isA[NP];
        int ldsoff[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int q = w * NP + u;
            if (q < PA) {
                constexpr int CPR = ROWA / 16;
                const int lc = q * 64 + lane, r = lc / CPR, c = lc % CPR;
                voff[u] = (r * (int)p.lda + ((c ^ tkey(r)) << 3)) * 2;
                ldsoff[u] = q * 1024;
                isA[u] = true;
            } else {
                constexpr int CPR = ROWB / 16;
                const int lc = (q - PA) * 64 + lane, r = lc / CPR, c = lc % CPR;
                voff[u] = (r * (int)p.ldb + ((c ^ tkey(r)) << 3)) * 2;
                ldsoff[u] = ABYTES + (q - PA) * 1024;
                isA[u] = false;
            }
        }
        auto issue = [&](int st, int slot) {
            const int k = kbeg + st * BK;
            const int soA = (k * (int)p.lda + m0) * 2, soB = (k * (int)p.ldb + n0) * 2;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                char* dst = smem + slot * STAGE + ldsoff[u];
#ifndef TN_ABLATE_DMA   // (timing-only builds: -DTN_ABLATE_DMA issues no loads, -DTN_ABLATE_MFMA no MFMAs)
                if (isA[u]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)dst, 16, voff[u], soA, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)dst, 16, voff[u], soB, 0, 0);
#else
                (void)dst; (void)soA; (void)soB;
#endif
            }
        };
        // Protocol (consumers read the fragments of stage s + 1 while they multiply stage s, see below): every slot
        // is requested up front; barrier P tells the consumers that stage 0 has landed, barrier s that stage s + 1 has,
        // and -- every consumer having waited for its own LDS reads before it -- that nobody reads stage s's slot
        // any more, so stage s + S goes there.  vmcnt retires in order: "stage t landed" = at most (number of
        // stages requested after t) * NP requests outstanding.
        auto wait_stages = [&](int younger) {
            if (younger >= 4) wait_vmcnt<NP * 4>();
            else if (younger == 3) wait_vmcnt<NP * 3>();
            else if (younger == 2) wait_vmcnt<NP * 2>();
            else if (younger == 1) wait_vmcnt<NP>();
            else wait_vmcnt<0>();
        };
#pragma unroll
        for (int q = 0; q < S; ++q)
            if (q < nst) issue(q, q);
        wait_stages(min(nst, S) - 1);
        __builtin_amdgcn_s_barrier();   // P
        for (int s = 0; s < nst; ++s) {
            wait_stages(max(0, min(nst - 1, s + S - 1) - (s + 1)));
            __builtin_amdgcn_s_barrier();
            if (s + S < nst) issue(s + S, s % S);
        }
        return;
    }

    // ===== consumer waves: ds_read_b64_tr_b16 + MFMA =====
    const int wm = wv / WN, wn = wv % WN, i = lane & 15, g = lane >> 4;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // tr-read addresses: lane (g, q = i>>2, pp = i&3) addresses token row 8g+q (+4), columns
    // cbase + 4pp .. +3 of a 16-column fragment
    const int q4 = i >> 2, pp = i & 3;
    const int row0 = 8 * g + q4;
    // (rows row0 and row0 + 4 have the same key: it looks at row bits 0-1 and 3, and row0 & 7 < 4 -- the second half of a
    // fragment is a constant 4 rows further)
    const int key = tkey(row0);
    int aLo[MI], bLo[NI];
#pragma unroll
    for (int t = 0; t < MI; ++t) {
        const int ch = (wm * MI * 16 + t * 16) / 8 + (pp >> 1);
        aLo[t] = row0 * ROWA + ((ch ^ key) << 4) + (pp & 1) * 8;
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int ch = (wn * NI * 16 + t * 16) / 8 + (pp >> 1);
        bLo[t] = ABYTES + row0 * ROWB + ((ch ^ key) << 4) + (pp & 1) * 8;
    }
    auto frag = [&](const char* st, int off, int rowbytes) {
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + off)));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + off + 4 * rowbytes)));
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };

    // Column sums of A (a bias gradient) ride along as one more MFMA per stage against a fragment of
    // ones: the column tiles with n0 == 0 take them, wave (wm, wn) for its A fragments mi == wn mod WN.
    const bool do_cs = p.colsum != nullptr && n0 == 0;
    f32x4 acs[(MI + WN - 1) / WN];
#pragma unroll
    for (int a = 0; a < (MI + WN - 1) / WN; ++a) acs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    // The fragments of stage s + 1 are read WHILE stage s is multiplied, into the registers of the fragments that
    // have just had their last use (an A fragment after its row of MFMAs, the B fragments along the last row): no
    // second set of 40 registers, and no LDS latency between a stage's barrier and its first MFMA -- all twelve
    // waves leave that barrier together, so the two MFMA waves of a SIMD used to wait for their reads side by side.
    u32x4 fa[MI], fb[NI];
    __builtin_amdgcn_s_barrier();   // P: stage 0 has landed
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < MI; ++t) fa[t] = frag(smem, aLo[t], ROWA);
#pragma unroll
    for (int t = 0; t < NI; ++t) fb[t] = frag(smem, bLo[t], ROWB);
    auto stage = [&](const char* nx, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        if (do_cs) {   // (first: the A fragments are overwritten row by row below)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                if (mi % WN == wn)
                    acs[mi / WN] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, ones), acs[mi / WN], 0, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#ifndef TN_ABLATE_MFMA
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]), acc[mi][ni], 0, 0, 0);
#else
                acc[mi][ni][0] += __builtin_bit_cast(float, fa[mi][0] ^ fb[ni][0]);
#endif
                if (more && mi == MI - 1) {
                    fb[ni] = frag(nx, bLo[ni], ROWB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (more) {
                fa[mi] = frag(nx, aLo[mi], ROWA);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int s = 0; s < nst; ++s) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // (one code path: after the last stage the reads fetch a slot nobody needs -- a second, read-free copy of the
        // body behind a branch costs hipcc's register allocator the 96 accumulators: 300 spilled registers)
        stage(smem + ((s + 1) % S) * STAGE, std::true_type());
    }

    // lane holds D[m = .. + 4g + r][n = .. + i].  A slab is private to this kernel and its reduce
    // kernel, so it keeps the accumulator layout [wave][mi][ni][lane][r]: one 16-byte store per
    // lane and accumulator tile, 1 KB contiguous per instruction (row-major would be 96 four-byte
    // stores per lane in 64-byte segments).
    if (do_cs && i == 0) {  // D[m = 4g + r][n = i]: every column holds the sums, column 0 adds them
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            if (mi % WN == wn) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(p.colsum + m0 + wm * MI * 16 + mi * 16 + 4 * g + r, acs[mi / WN][r]);
            }
    }
    if (slab) {
        f32x4* dst = (f32x4*)(slab + ((int64_t)tile * nsplit + split) * TM * TN) + (wv * MI * NI) * 64 + lane;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) dst[(mi * NI + ni) * 64] = acc[mi][ni];
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lm = wm * MI * 16 + mi * 16 + 4 * g + r, ln = wn * NI * 16 + ni * 16 + i;
                    atomicAdd(p.out + (int64_t)(m0 + lm) * p.ldo + n0 + ln, acc[mi][ni][r]);
                }
    }
#endif
}

template __global__ void gemm_tn_bf16_v2<2, 4, 4, 6>(const TnGroup, float* __restrict__, int, int);
template __global__ void gemm_tn_bf16_v2<4, 2, 6, 4>(const TnGroup, float* __restrict__, int, int);

// ---------------------------------------------------------------------------------------------------------------
// EXPERIMENT, opt-in (DINO_TN_V3=1), measured SLOWER than v2: the block's grouped launch 790-820 us against 735-760 us,
// the step 47.2-47.4 ms against 46.6-46.7 (profiles/r03_tn_pipelined_and_v3.log).  Kept, tested, because it says what
// the weight-gradient kernel is NOT bound by: a quarter fewer staged bytes per FLOP buy nothing once the MFMA waves
// have to issue the requests themselves.
// v3: the same contraction with 192 x 384 output tiles on EIGHT waves (two per SIMD, 256 registers each: 144
// accumulator registers per lane) that request their own stages.  v2's 128 x 384 tile is what 96 accumulator
// registers allow at three waves per SIMD (two MFMA waves + a loader); its K-loop runs at the rate the fill path
// delivers stages, about twice its MFMA time, so the lever is bytes per FLOP: 192 x 384 stages 128 FLOP per byte
// against 96.  Eight waves share the 36 one-KiB pieces of a stage (waves 0-3 five, 4-7 four; inline asm, so that
// hipcc does not put vmcnt(0) in front of every LDS read that follows a request -- see attention.hip), each waits
// for ITS pieces of stage s + 1 before barrier s, the fragments of stage s + 1 are read while stage s is multiplied
// (as in v2), and stage s + 4 is requested into stage s's slot right behind barrier s.
// A rows are 384 B here, not a multiple of 256 B: consecutive rows start 32 banks apart, and the chunk key has to
// stay inside groups of 8 chunks (24 per row), so A uses key3 (bits from row bits 1 and 3); B rows (768 B) keep tkey.
constexpr int NTHR3 = 512, S3 = 4;
DEVI int key3(int row) { return 2 * (((row >> 1) & 1) | (((row >> 3) & 1) << 1)); }
template <int ROWBYTES> DEVI int rkey(int row) { return ROWBYTES % 256 == 0 ? tkey(row) : key3(row); }
DEVI u32x4 tn_rsrc(const void* p, int bytes) {
    const uint64_t a = (uint64_t)p;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}
DEVI void tn_dma16(unsigned lds, int voff, const u32x4& rs, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
DEVI void tn_wait_vm(int n) {   // n requests may stay outstanding (n is one of the products below)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(NTHR3, 2) void gemm_tn_bf16_v3(const TnGroup grp, float* __restrict__ slab,
                                                              int nsplit, int kper) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = WM * MI * 16, TN = WN * NI * 16;
    constexpr int ROWA = TM * 2, ROWB = TN * 2;
    constexpr int ABYTES = BK * ROWA, BBYTES = BK * ROWB, STAGE = ABYTES + BBYTES;
    constexpr int PA = ABYTES / 1024, PB = BBYTES / 1024, NPC = PA + PB, NPW = (NPC + 7) / 8;
    static_assert(NPW <= 5 && NPC > 8 * (NPW - 1) && NPW <= MI, "piece counts the waits below know; one request per row of MFMAs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile_ = gridDim.x / nsplit;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid / ntile_, tile = lid - split * ntile_;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < MAXG; ++q)
        if (q < grp.count && tile >= grp.prob[q].tile_begin) pi = q;
    const TnProb p = grp.prob[pi];
    const int ltile = tile - p.tile_begin;
    const int m0 = (ltile / p.tiles_n) * TM, n0 = (ltile % p.tiles_n) * TN;
    const int kbeg = split * kper, kend = min(grp.K, kbeg + kper);
    const int nst = (kend - kbeg + BK - 1) / BK;

    // ---- this wave's pieces of a stage: q = wv, wv + 8, ... (< NPC)
    const u32x4 rsA = tn_rsrc(p.A, (int)(((int64_t)(grp.K - 1) * p.lda + p.M) * 2));
    const u32x4 rsB = tn_rsrc(p.B, (int)(((int64_t)(grp.K - 1) * p.ldb + p.N) * 2));
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const int npw = (NPC - wv + 7) / 8;   // (scalar) NPW or NPW - 1
    int voff[NPW];
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
        const int q = wv + 8 * u;
        if (q < PA) {
            constexpr int CPR = ROWA / 16;
            const int lc = q * 64 + lane, r = lc / CPR, c = lc % CPR;
            voff[u] = (r * (int)p.lda + ((c ^ rkey<ROWA>(r)) << 3)) * 2;
        } else {
            constexpr int CPR = ROWB / 16;
            const int lc = (q - PA) * 64 + lane, r = lc / CPR, c = lc % CPR;
            voff[u] = (r * (int)p.ldb + ((c ^ rkey<ROWB>(r)) << 3)) * 2;
        }
    }
    // piece u of this wave for stage st (the caller has checked st < nst)
    auto issue_piece = [&](int st, int slot, int u) {
        const int k = kbeg + st * BK;
        const int soA = (k * (int)p.lda + m0) * 2, soB = (k * (int)p.ldb + n0) * 2;
#ifndef TN_ABLATE_DMA
        const int q = wv + 8 * u;
        if (q < NPC) {
            if (q < PA) tn_dma16(lds0 + slot * STAGE + q * 1024, voff[u], rsA, soA);
            else tn_dma16(lds0 + slot * STAGE + q * 1024, voff[u], rsB, soB);
        }
#else
        (void)soA; (void)soB;
#endif
    };
    auto issue = [&](int st, int slot) {
#pragma unroll
        for (int u = 0; u < NPW; ++u) issue_piece(st, slot, u);
    };

    const int wm = wv / WN, wn = wv % WN, i = lane & 15, g = lane >> 4;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q4 = i >> 2, pp = i & 3;
    const int row0 = 8 * g + q4;   // (rows row0 and row0 + 4 have the same key, under both keys)
    int aLo[MI], bLo[NI];
#pragma unroll
    for (int t = 0; t < MI; ++t) {
        const int ch = (wm * MI * 16 + t * 16) / 8 + (pp >> 1);
        aLo[t] = row0 * ROWA + ((ch ^ rkey<ROWA>(row0)) << 4) + (pp & 1) * 8;
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int ch = (wn * NI * 16 + t * 16) / 8 + (pp >> 1);
        bLo[t] = ABYTES + row0 * ROWB + ((ch ^ rkey<ROWB>(row0)) << 4) + (pp & 1) * 8;
    }
    auto frag = [&](const char* st, int off, int rowbytes) {
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + off)));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + off + 4 * rowbytes)));
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    const bool do_cs = p.colsum != nullptr && n0 == 0;
    f32x4 acs[(MI + WN - 1) / WN];
#pragma unroll
    for (int a = 0; a < (MI + WN - 1) / WN; ++a) acs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    // "stage t has landed for this wave's pieces" = at most npw * (stages requested after t) requests outstanding
#pragma unroll
    for (int q = 0; q < S3; ++q)
        if (q < nst) issue(q, q);
    tn_wait_vm(npw * (min(nst, S3) - 1));
    __builtin_amdgcn_s_barrier();   // P: stage 0 is complete
    asm volatile("" ::: "memory");
    u32x4 fa[MI], fb[NI];
#pragma unroll
    for (int t = 0; t < MI; ++t) fa[t] = frag(smem, aLo[t], ROWA);
#pragma unroll
    for (int t = 0; t < NI; ++t) fb[t] = frag(smem, bLo[t], ROWB);
    for (int s = 0; s < nst; ++s) {
        tn_wait_vm(npw * max(0, min(nst - 1, s + S3 - 1) - (s + 1)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // stage s + 1 is complete; nobody reads stage s's slot any more
        asm volatile("" ::: "memory");
        // stage s + S3 goes into the slot stage s has left; its requests are spread over the rows of MFMAs below -- a
        // request holds the wave for 60-180 cycles, and issued in a burst behind the barrier by both waves of a SIMD
        // they left the matrix pipe idle for that long in every stage
        const bool refill = s + S3 < nst;
        const char* nx = smem + ((s + 1) % S3) * STAGE;
        if (do_cs) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                if (mi % WN == wn)
                    acs[mi / WN] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, ones), acs[mi / WN], 0, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#ifndef TN_ABLATE_MFMA
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]), acc[mi][ni], 0, 0, 0);
#else
                acc[mi][ni][0] += __builtin_bit_cast(float, fa[mi][0] ^ fb[ni][0]);
#endif
                if (mi == MI - 1) {
                    fb[ni] = frag(nx, bLo[ni], ROWB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fa[mi] = frag(nx, aLo[mi], ROWA);
            __builtin_amdgcn_sched_barrier(0);
            if (mi < NPW && refill) issue_piece(s + S3, s % S3, mi);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    if (do_cs && i == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            if (mi % WN == wn) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(p.colsum + m0 + wm * MI * 16 + mi * 16 + 4 * g + r, acs[mi / WN][r]);
            }
    }
    if (slab) {
        f32x4* dst = (f32x4*)(slab + ((int64_t)tile * nsplit + split) * TM * TN) + (wv * MI * NI) * 64 + lane;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) dst[(mi * NI + ni) * 64] = acc[mi][ni];
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lm = wm * MI * 16 + mi * 16 + 4 * g + r, ln = wn * NI * 16 + ni * 16 + i;
                    atomicAdd(p.out + (int64_t)(m0 + lm) * p.ldo + n0 + ln, acc[mi][ni][r]);
                }
    }
#endif
}

template __global__ void gemm_tn_bf16_v3<2, 4, 6, 6>(const TnGroup, float* __restrict__, int, int);

// out[m0+lm, n0+ln] += sum_s slab[tile*nsplit + s][wave][mi][ni][lane][r]  (accumulator layout, see
// the kernel's epilogue): lm = wm*MI*16 + mi*16 + 4*(lane>>4) + r, ln = wn*NI*16 + ni*16 + (lane&15)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, const TnGroup grp,
                                                          int TM, int TN, int nsplit,
                                                          int WN, int MI, int NI) {
    const int tile = blockIdx.y;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < MAXG; ++q)
        if (q < grp.count && tile >= grp.prob[q].tile_begin) pi = q;
    const TnProb p = grp.prob[pi];
    float* __restrict__ out = p.out;
    const int64_t ldo = p.ldo;
    const int ltile = tile - p.tile_begin;
    const int m0 = (ltile / p.tiles_n) * TM, n0 = (ltile % p.tiles_n) * TN;
    const int per = TM * TN;
    for (int e = (blockIdx.x * 256 + threadIdx.x) * 4; e < per; e += gridDim.x * 256 * 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const float* base = slab + (int64_t)tile * nsplit * per + e;
        for (int k = 0; k < nsplit; ++k) {
            const f32x4 v = *(const f32x4*)(base + (int64_t)k * per);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        const int lane = (e >> 2) & 63, t = e >> 8;
        const int ni = t % NI, mi = (t / NI) % MI, wv = t / (NI * MI), wm = wv / WN, wn = wv - wm * WN;
        const int lm = wm * MI * 16 + mi * 16 + 4 * (lane >> 4), ln = wn * NI * 16 + ni * 16 + (lane & 15);
        float* o = out + (int64_t)(m0 + lm) * ldo + n0 + ln;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * ldo] += s[r];
    }
}

int num_cus_tn() {
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

}  // namespace

extern "C" int dino_gemm_work_queue(int on);   // (gemm_nt_v3.hip)

template <int WM, int WN, int MI, int NI, bool V3 = false>
static int launch_tn(const dino_gemm_args* a, int count, hipStream_t st) {
    constexpr int TM = WM * MI * 16, TN = WN * NI * 16;
    constexpr int smem = (V3 ? S3 : S) * BK * (TM + TN) * 2;
    const void* kernel;
    if constexpr (V3) kernel = (const void*)gemm_tn_bf16_v3<WM, WN, MI, NI>;
    else kernel = (const void*)gemm_tn_bf16_v2<WM, WN, MI, NI>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    TnGroup grp;
    grp.count = count;
    grp.K = (int)a[0].K;
    int tiles = 0;
    for (int q = 0; q < count; ++q) {
        TnProb& pr = grp.prob[q];
        pr.A = a[q].A;
        pr.B = a[q].B;
        pr.out = (float*)a[q].out;
        pr.colsum = a[q].colsum;
        pr.lda = (int)a[q].lda;
        pr.ldb = (int)a[q].ldb;
        pr.ldo = (int)a[q].ldo;
        pr.M = (int)a[q].M;
        pr.N = (int)a[q].N;
        pr.tile_begin = tiles;
        pr.tiles_n = pr.N / TN;
        tiles += (pr.M / TM) * pr.tiles_n;
    }
    for (int q = count; q < MAXG; ++q) grp.prob[q] = grp.prob[0];
    // split the tokens so the grid fills the chip once; every split a multiple of BK tokens.
    // With dino_gemm_work_queue on (multi-rank runs: another stream's kernel -- RCCL -- may hold CUs when this one
    // starts) twice as many, half as long: one workgroup per CU and exactly one round meant that every held CU's
    // workgroup ran after all the others (1.4 x with 8 of 256 CUs held); with two rounds' worth of shorter
    // workgroups the dispatcher hands them to whichever CU is free (tools/bench_occupied_others.py).
    int nsplit = num_cus_tn() / tiles;
    // (... sized for 16 CUs fewer, so that the two rounds stay two rounds with up to ~20 CUs held)
    if (dino_gemm_work_queue(-1) && num_cus_tn() > 64) nsplit = 2 * ((num_cus_tn() - 16) / tiles);
    if (nsplit < 1) nsplit = 1;
    const int maxsplit = grp.K / (BK * 8) > 0 ? grp.K / (BK * 8) : 1;
    if (nsplit > maxsplit) nsplit = maxsplit;
    int kper = (grp.K + nsplit - 1) / nsplit;
    kper = (kper + BK - 1) / BK * BK;
    nsplit = (grp.K + kper - 1) / kper;
    const int64_t need = (int64_t)tiles * nsplit * TM * TN * 4;
    float* slab = (a[0].workspace && a[0].workspace_bytes >= need && nsplit > 1) ? (float*)a[0].workspace : nullptr;
    if constexpr (V3)
        hipLaunchKernelGGL((gemm_tn_bf16_v3<WM, WN, MI, NI>), dim3(tiles * nsplit), dim3(NTHR3), smem, st, grp,
                           slab, nsplit, kper);
    else
        hipLaunchKernelGGL((gemm_tn_bf16_v2<WM, WN, MI, NI>), dim3(tiles * nsplit), dim3(NTHR), smem, st, grp,
                           slab, nsplit, kper);
    if (slab)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(TM * TN / 1024, tiles), dim3(256), 0, st, slab, grp,
                           TM, TN, nsplit, WN, MI, NI);
    return dino_launch_status();
}

bool dino_gemm_tn_v2_supported(const dino_gemm_args& a) {
    if (a.dtype != DINO_BF16 || !a.trans || a.out_mode != 2) return false;
    if (a.bias || a.act || a.residual || a.rowscale || a.row_map_n || a.res_mod) return false;
    if (a.K < 8 * BK) return false;   // any token count: rows past K are zero-filled by the descriptors
    const int64_t ba = ((int64_t)a.K * a.lda) * 2, bb = ((int64_t)a.K * a.ldb) * 2;
    if (ba >= (1ll << 31) || bb >= (1ll << 31)) return false;
    return (a.M % 128 == 0 && a.N % 384 == 0) || (a.M % 384 == 0 && a.N % 128 == 0) ||
           (a.M % 128 == 0 && a.N % 256 == 0);
}

// DINO_TN_V3=1: 192 x 384 tiles on eight waves (gemm_tn_bf16_v3) wherever every problem of a launch divides into them
static bool tn_v3_on() {
    static const bool on = getenv("DINO_TN_V3") && getenv("DINO_TN_V3")[0] == '1';
    return on;
}
static bool tn_v3_fits(const dino_gemm_args* a, int count) {
    for (int q = 0; q < count; ++q)
        if (a[q].M % 192 != 0 || a[q].N % 384 != 0) return false;
    return true;
}

int dino_gemm_tn_v2(const dino_gemm_args& a, hipStream_t st) {
    if (tn_v3_on() && tn_v3_fits(&a, 1)) return launch_tn<2, 4, 6, 6, true>(&a, 1, st);          // 192 x 384
    if (a.M % 128 == 0 && a.N % 384 == 0) return launch_tn<2, 4, 4, 6>(&a, 1, st);  // 128 x 384
    if (a.M % 384 == 0 && a.N % 128 == 0) return launch_tn<4, 2, 6, 4>(&a, 1, st);  // 384 x 128
    // widths that are multiples of 256 but not of 384 (ViT-L: 1024, 4096): 128 x 256 tiles, 16
    // instead of 24 MFMAs per wave and stage against 16 instead of 20 fragment reads
    return launch_tn<2, 4, 4, 4>(&a, 1, st);
}

// 128 x 384 output tiles of a problem the grouped launch can take (0: it cannot)
int dino_gemm_tn_group_tiles(const dino_gemm_args& a) {
    if (!dino_gemm_tn_v2_supported(a) || a.M % 128 != 0 || a.N % 384 != 0 || !a.workspace) return 0;
    return (a.M / 128) * (a.N / 384);
}

// Longest prefix of args[0..count) that should share one launch: same token count, and few enough
// tiles that the split over tokens still has >= 4 ranges (a group of 144 tiles -- ViT-B's four
// weight gradients -- would run unsplit on 144 of 256 CUs).  0: args[0] is not groupable at all.
int dino_gemm_tn_group_prefix(const dino_gemm_args* a, int count) {
    const int cap = num_cus_tn() / 4;
    int n = 0, tiles = 0;
    while (n < count && n < MAXG) {
        const int t = dino_gemm_tn_group_tiles(a[n]);
        if (t == 0 || a[n].K != a[0].K || (n > 0 && tiles + t > cap)) break;
        tiles += t;
        ++n;
    }
    return n;
}

int dino_gemm_tn_group_v2(const dino_gemm_args* a, int count, hipStream_t st) {
    if (tn_v3_on() && tn_v3_fits(a, count)) return launch_tn<2, 4, 6, 6, true>(a, count, st);
    return launch_tn<2, 4, 4, 6>(a, count, st);
}
