// Warp-specialised persistent NT GEMM for bf16: 256 x 192 output tile, full-line (BK = 64)
// LDS-DMA, 8 consumer waves + 4 loader waves per workgroup, one workgroup per CU.
//
// Why a third kernel: gemm_nt_v2 stages 32-deep slabs, so every DMA wave-instruction touches 16
// half-used 128-B lines and the texture-address path, not the MFMA, sets the pace; and because
// vmcnt retires in issue order, a wave that both issues DMA and stores its output tile cannot
// learn that its next operands have landed before its stores have drained.  Here
//   * a 256 x 192 tile (110 FLOP per staged byte; every ViT width -- 192/384/768/1152/1536/
//     2304/3072 -- is a multiple of 192), K-tiles 64 deep: LDS rows are 128 B = one full cache
//     line, a DMA piece is 8 whole lines;
//   * waves 8-11 (one per SIMD) do nothing but issue buffer_load ... lds and wait for them with
//     a counted s_waitcnt; waves 0-7 (two per SIMD, 64 x 96 each) do nothing but ds_read, MFMA
//     and the epilogue, and never wait on vmcnt for an operand: their output stores drain in the
//     background under the next tile's MFMAs, and a loader's ~100-cycle DMA issue stalls never
//     sit in an MFMA wave's instruction stream.  One s_barrier per K-tile joins the two roles;
//   * the A operand (activations, first touch comes from HBM) has a 3-slot ring, the B operand
//     (weights, always L2-resident) a 2-slot ring: 144 KB of LDS, issue order B(t+1), A(t+2);
//   * B rows are stored in LDS in fragment order (the permutation is applied on the DMA source
//     address), so both operands read with constant immediate offsets and the four lanes that
//     share an output row hold 32 consecutive columns per store (64-B segments);
//   * 16-B chunks of a row are XOR-swizzled with key = (row >> 1) & 7 on the source address and
//     on the read: conflict-free for ds_read_b128's four 16-lane groups.
// The workgroup is persistent and walks its tiles as ONE stream of K-tiles, so the next tile's
// operands are in flight during an epilogue.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace {

constexpr int BM = 256, BN = 192, BK = 64, NTHR = 768, NCONS = 512;
constexpr int A_SLOT = BM * BK * 2;  // 32 KiB
constexpr int B_SLOT = BN * BK * 2;  // 24 KiB
// Ring depths.  -DNT3_NAS / -DNT3_NBS beyond 3 / 2 are TIMING-ONLY builds (tools/ab_nt_ring.sh): the extra slots
// alias the first ones (results are garbage), which is how deeper prefetch was measured before finding LDS for it.
#ifndef NT3_NAS
#define NT3_NAS 3
#endif
#ifndef NT3_NBS
#define NT3_NBS 2
#endif
constexpr int NAS = NT3_NAS, NBS = NT3_NBS;
constexpr int NAS_REAL = NAS > 3 ? 3 : NAS, NBS_REAL = NBS > 2 ? 2 : NBS;
constexpr int B_BASE = NAS_REAL * A_SLOT;
DEVI int a_slot_off(int s) { return (NAS > NAS_REAL ? s % NAS_REAL : s) * A_SLOT; }
DEVI int b_slot_off(int s) { return (NBS > NBS_REAL ? s % NBS_REAL : s) * B_SLOT; }
// DMA pieces that may still be outstanding when K-tile t + 1 must have landed (issue order per K-tile t:
// B(t + NBS - 1), A(t + NAS - 1); vmcnt retires in issue order)
constexpr int NPA_ = 8, NPB_ = 6;
constexpr int AFTER_A = (NAS - 2) * (NPA_ + NPB_), AFTER_B = NPA_ + (NBS - 2) * (NPA_ + NPB_);
constexpr int VM_STEADY = AFTER_A < AFTER_B ? AFTER_A : AFTER_B;
static_assert(VM_STEADY <= 63, "vmcnt is six bits");
constexpr int BIAS_BASE = B_BASE + NBS_REAL * B_SLOT;  // 144 KiB
// The bias vector sits in LDS for the life of the workgroup.  Up to 2048 columns the allocation is
// 152.75 KiB; a longer bias (ViT-B's qkv/fc1, ViT-L's fc1: up to 4096 columns) takes it to exactly
// the CU's 160 KiB and lies over the column-sum accumulators, so a launch that wants a long bias AND
// column sums (none on the DINO path) goes to the 256x128 kernel.  The small allocation is kept
// for the common case on purpose: with all 160 KiB taken, no kernel of a side stream that needs
// LDS can share the CU, and the ViT-S/16 step was 10-14 ms slower.
constexpr int MAX_BIAS_SMALL = 2048, MAX_BIAS = 4096;
constexpr int COLSUM_BASE = BIAS_BASE + MAX_BIAS_SMALL * 4;  // 152 KiB
constexpr int SMEM_BYTES = COLSUM_BASE + BN * 4;
constexpr int SMEM_BYTES_LONG_BIAS = BIAS_BASE + MAX_BIAS * 4;  // 160 KiB
static_assert(SMEM_BYTES_LONG_BIAS <= 160 * 1024 && SMEM_BYTES <= SMEM_BYTES_LONG_BIAS, "LDS budget");
// GELU by table (ACT 5 / 6 = ACT 1 / 3 with the table): the epilogue's erf / exp arithmetic (~24 issue slots per
// element, two of them quarter-rate transcendentals, VALU time that nothing overlaps) becomes one LDS read.
// The pre-activation is rounded to bf16 (which the reference's autocast does too, to fp16) and its magnitude
// indexes Phi(-|x|) and gelu'(-|x|) as an fp16 pair for |x| in [2^-9, 8): 12 binades x 128 mantissas; the
// other sign is 1 - value (both are small on the negative side, so nothing cancels).  gelu = x * Phi uses
// the unrounded x.
constexpr int LUT_LO = (127 - 9) << 7, LUT_N = 12 * 128, LUT_BASE = SMEM_BYTES;
constexpr int SMEM_BYTES_LUT = SMEM_BYTES + LUT_N * 4;
static_assert(SMEM_BYTES_LUT <= 160 * 1024, "LDS budget");
// Output tiles are written once and not re-read by this GEMM: non-temporal stores, so that they do
// not push the A/B panels (re-read by every tile of a row / column) out of L2: fc1-shape 147 -> 123 us.
// (The same hint on the epilogue's operand loads made things worse.)
constexpr int NT_AUX = 2;
constexpr int NPA = 8, NPB = 6;  // DMA pieces per loader wave per K-tile

// EXPERIMENT, off (-DNT3_PF_DIST=n to build it): first touch of the epilogue's operand tile (residual / saved GELU',
// 96 KB per output tile) n K-tiles before the epilogue that reads it -- one dword per 128-B line through the LDS-DMA
// path into a dump area, issued by the MFMA waves from inline asm.  Measured slower at every distance
// (profiles/r03_nt_ring_and_touch_experiments.log: dgrad x GELU' 331 -> 343 / 352 / 371 us at n = 1 / 2 / 6, the
// residual variants +3..+10 us): these epilogues are paced by the bytes of their operand stream, not by its latency,
// and the touches are more requests on the same path.
#ifndef NT3_PF_DIST
#define NT3_PF_DIST 0
#endif
DEVI u32x4 raw_rsrc(const void* p, int bytes) {
    const uint64_t a = (uint64_t)p;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}

template <int N> DEVI void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

DEVI void unpack8(const u32x4& q, float (&x)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        x[2 * r] = __builtin_bit_cast(float, q[r] << 16);
        x[2 * r + 1] = __builtin_bit_cast(float, q[r] & 0xffff0000u);
    }
}

// gl[r] = gelu(v[r]); dg[r] = gelu'(v[r]) (BOTH) from the table at lut_adj = table base - 4 * LUT_LO.
// With t = Phi(-|x|) and d = gelu'(-|x|) from the table (fp16 halves of one word):
//   gelu(x)  = max(x, 0) - |x| * t            (x > 0: x - x t = x Phi(x);  x < 0: x t)
//   gelu'(x) = step(x) - sgn(x) * d           (x > 0: 1 - d;  x < 0: d)
// -- no compare / select, and each is ONE v_fma_mix_f32 that takes the fp16 half directly.  The two indices of a
// bf16 pair are clamped together (v_pk_max_u16 / v_pk_min_u16).  13.5 -> ~10 VALU instructions per element; the
// epilogue's instructions are all exposed time.
// (Hand-picked instructions: written as C++, hipcc converts the fp16 halves separately, canonicalises before every
// max, materialises |x| for its packed fma and extracts the two indices with shifts and masks -- no gain over the
// compare / select form.  The asm operands are never raw MFMA results: x is acc * prescale + bias or acc + bias,
// formed by instructions the compiler emits itself.)
// One asm block per element (between separate asm statements hipcc pads every dependent pair with an s_nop).
template <bool BOTH> DEVI void lut_apply(unsigned ent, float x, float& gl, float& dg) {
    if (BOTH) {
        float t;
        asm("v_max_f32 %0, 0, %4\n\t"                                                     // relu(x)
            "v_bfi_b32 %1, %5, 1.0, %4\n\t"                                               // sgn(x) = +-1
            "v_fma_mix_f32 %0, %3, -|%4|, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"        // relu - |x| * t
            "v_max_f32 %2, 0, %1\n\t"                                                     // step(x)
            "v_fma_mix_f32 %1, %3, -%1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"              // step - sgn * d
            : "=&v"(gl), "=&v"(dg), "=&v"(t)
            : "v"(ent), "v"(x), "s"(0x7fffffffu));
    } else {
        asm("v_max_f32 %0, 0, %2\n\t"
            "v_fma_mix_f32 %0, %1, -|%2|, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(gl)
            : "v"(ent), "v"(x));
    }
}
template <bool HI> DEVI unsigned lut_addr(unsigned w, unsigned lut_adj) {   // 4 * half of w + lut_adj
    unsigned r;
    if (HI) asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "s"(lut_adj));
    else asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[0,0,0,0]" : "=v"(r) : "v"(w), "s"(lut_adj));
    return r;
}
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const unsigned* lds_cu32_t;
template <bool BOTH> DEVI void gelu_lut8(const float (&v)[8], const char* lut_adj, float (&gl)[8], float (&dg)[8]) {
    const unsigned adj = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)lut_adj);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bf16x2 h = {(bf16)v[2 * r], (bf16)v[2 * r + 1]};
        u16x2 w = __builtin_bit_cast(u16x2, __builtin_bit_cast(unsigned, h) & 0x7fff7fffu);
        w = __builtin_elementwise_min(__builtin_elementwise_max(w, u16x2{(unsigned short)LUT_LO, (unsigned short)LUT_LO}),
                                      u16x2{(unsigned short)(LUT_LO + LUT_N - 1), (unsigned short)(LUT_LO + LUT_N - 1)});
        const unsigned wp = __builtin_bit_cast(unsigned, w);
        const unsigned e0 = *(lds_cu32_t)(uintptr_t)lut_addr<false>(wp, adj);
        const unsigned e1 = *(lds_cu32_t)(uintptr_t)lut_addr<true>(wp, adj);
        const float x0 = v[2 * r], x1 = v[2 * r + 1];   // (gl may be v itself)
        lut_apply<BOTH>(e0, x0, gl[2 * r], dg[2 * r]);
        lut_apply<BOTH>(e1, x1, gl[2 * r + 1], dg[2 * r + 1]);
    }
}

DEVI u32x4 pack8(const float (&x)[8]) {
    u32x4 q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bf16x2 h = {(bf16)x[2 * r], (bf16)x[2 * r + 1]};
        q[r] = __builtin_bit_cast(unsigned, h);
    }
    return q;
}

struct Cursor {
    int tile, kt, m0, n0, half;  // tile = work-item index (< 0: not looked up yet); half: a 128-row item of the last round
    int j;                       // ordinal of the item in this workgroup's sequence
};

// Work queue (`sched` != nullptr): a workgroup's FIRST item is its own (blockIdx -> tile, XCD-aware as before), every
// further one is drawn from a counter of its XCD -- the XCD's items in the static order (base + i + round * grid), so
// the tiles that share an A panel still meet in one L2.  Why: the kernel holds a whole CU per workgroup (152 KB of
// LDS, 3 x 168 registers per SIMD); when another stream's kernel -- RCCL's all-reduce of a gradient bucket under
// backward, the side-stream weight gradients -- sits on n CUs at launch, n workgroups start a whole kernel late, and
// with a static share of 1/grid of the tiles each they doubled the kernel's duration.  Drawn from a queue, the late
// ones find it (nearly) empty and the others absorb their share: grid / (grid - n) instead of 2 x.
// Slot layout (16 uints): [0..7] per-XCD draw counters, [8] workgroups that have finished (the last one clears the slot).
constexpr int SCHED_SLOT = 16, SCHED_SLOTS = 1024;
__device__ unsigned nt3_sched_mem[SCHED_SLOT * SCHED_SLOTS];

// LN: LayerNorm folded into the GEMMs around it (include/dino_hip.h).  1: `prescale` -- the accumulator times a
// per-row fp32 factor (rstd) before the bias; 2: `ln_stats` -- per output row and 96-column wave slice, (sum, sum
// of squares) of the values written, for the NEXT LayerNorm's statistics; 3: like 1, but the factor is formed here
// from such partials (`ln_stats_in`: no finalize launch between the two GEMMs).
// QUEUE: the work-queue form (a template parameter, not a run-time branch: the static form keeps its leaner loops --
// as a run-time switch the queue's cursors cost the static step 0.15 ms)
template <int ACT, bool HASLD, bool COLSUM, int LN = 0, bool QUEUE = false>
__global__ __launch_bounds__(NTHR) void gemm_nt_bf16_v3(const dino_gemm_args p, int tiles_n, int nfull, int nhalf,
                                                        unsigned* sched) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    float* sbias = (float*)(smem + BIAS_BASE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int b = xcd_remap(blockIdx.x, G);
    // this XCD's share of the items in the static order: base + i + round * G, i < xcnt (see xcd_remap)
    const int x8 = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
    const int xcnt = q8 + (x8 < r8 ? 1 : 0);
    const int xbase = x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8;
    constexpr int SEQ_BASE = (ACT == 5 || ACT == 6) ? SMEM_BYTES_LUT : SMEM_BYTES;   // 8 ints behind everything else
    // The ring is read and written from asm volatile statements: ordered against the barriers (both have side
    // effects) without a "memory" clobber, which in the loaders' loop made hipcc reload kernel arguments per K-tile.
    const unsigned seq_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + SEQ_BASE);
    auto seq_get = [&](int j) -> int {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(seq_lds + 4u * (unsigned)(j & 7)));
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto seq_put = [&](int j, int v) {   // (lane 0 of loader wave 0; the caller waits for lgkmcnt before publishing)
        asm volatile("ds_write_b32 %0, %1" ::"v"(seq_lds + 4u * (unsigned)(j & 7)), "v"(v));
    };

    if (p.bias) {
        for (int n = tid; n < p.N; n += NTHR) sbias[n] = p.bias[n];
    }
    // Fused column sums: all tiles of a workgroup share their columns when the grid is a multiple
    // of the tiles per row, so the sums gather in LDS and reach memory once per workgroup -- every
    // row tile adding straight into the same 192 addresses made the atomics the bottleneck.
    float* scol = (float*)(smem + COLSUM_BASE);
    const bool persist_colsum = COLSUM && G % tiles_n == 0;
    if (COLSUM && tid < BN) scol[tid] = 0.f;   // (shares LDS with bias columns >= 2048)
    if (ACT == 5 || ACT == 6) {
        unsigned* lut = (unsigned*)(smem + LUT_BASE);
        for (int q = tid; q < LUT_N; q += NTHR) {
            const float a = __builtin_bit_cast(float, (unsigned)(LUT_LO + q) << 16);
            float ex;
            const float t = gelu_cdf_fast(-a, ex);                          // Phi(-a)
            const float d = fmaf(-a * 0.39894228040143267794f, ex, t);      // gelu'(-a)
            lut[q] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)t) |
                     ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)d) << 16);
        }
    }
    const char* lut_adj = smem + LUT_BASE - 4 * LUT_LO;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // Work items: nfull whole 256-row tiles, then the nhalf tiles of the partly filled last round
    // as 2*nhalf half tiles (rows 0-127 / 128-255 by the waves of rows 0-127 -- one MFMA wave per
    // SIMD, so a half tile takes about half the time and the last round costs half a round).
    const int ntiles = nfull + 2 * nhalf;
    auto leave = [&]() {   // queue bookkeeping of a workgroup that is done (one lane): the last one clears the slot
        if (__hip_atomic_fetch_add(sched + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G - 1)) {
#pragma unroll
            for (int z = 0; z < 9; ++z) __hip_atomic_store(sched + z, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (b >= ntiles) {
        if (QUEUE && tid == 0) leave();
        return;
    }
#ifdef NT3_STAGGER_PHASES   // experiment: workgroups start out of phase, so that their store bursts do not coincide
    for (int z = 0; z < (blockIdx.x / 8 % NT3_STAGGER_PHASES) * NT3_STAGGER_SLEEPS; ++z) __builtin_amdgcn_s_sleep(127);
#endif
    const int nkt = p.K / BK;
    const int lda = (int)p.lda, ldb = (int)p.ldb;
    // j-th item of this workgroup: static stride, or the ring that loader wave 0 fills two items ahead (entry j is
    // written before the barrier of the first K-tile of item j - 2 and read after it)
    auto item_at = [&](int j) -> int {
        if (!QUEUE) return b + j * G;
        return j == 0 ? b : seq_get(j);
    };

    auto locate = [&](Cursor& c, int w) {
        c.tile = w;
        c.kt = 0;
        int tile = w, hh = 0;
        c.half = 0;
        if (w >= nfull) {  // (nhalf is a multiple of tiles_n whenever the grid is: n0 stays per-workgroup)
            const int h = w - nfull;
            hh = h >= nhalf ? 1 : 0;
            tile = nfull + h - hh * nhalf;
            c.half = 1;
        }
        c.m0 = (tile / tiles_n) * BM + hh * (BM / 2);
        c.n0 = (tile % tiles_n) * BN;
    };
    auto start = [&](Cursor& c) {
        c.j = 0;
        locate(c, b);
    };
    // the next item is looked up when it is first needed (resolve), not when the last K-tile of this one is passed:
    // its ring entry may not be published yet at that point
    auto advance = [&](Cursor& c) {
        if (++c.kt == nkt) {
            ++c.j;
            if (QUEUE) {
                c.kt = 0;
                c.tile = -1;
            } else {
                locate(c, b + c.j * G);
            }
        }
    };
    auto resolve = [&](Cursor& c) {
        if (c.tile < 0) locate(c, item_at(c.j));
    };

    if (wv >= NCONS / 64) {
        // ================= loader waves =================
        // A piece = 8 LDS rows x 128 B; lane -> (row = lane >> 3, physical chunk = lane & 7),
        // logical chunk = physical ^ key(row).  Loader w moves A pieces 8w..8w+7 and B pieces
        // 6w..6w+5.  B's LDS row rho = wnB*96 + t*16 + ii holds W row
        // wnB*96 + 32*(t >> 1) + 8*(ii >> 2) + 4*(t & 1) + (ii & 3)  (fragment order, see header).
        const int w = wv - NCONS / 64;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.B, 0, (int)(((int64_t)(p.N - 1) * p.ldb + p.K) * 2), 0x00020000);
        int voffA[NPA], voffB[NPB];
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int row = 8 * (NPA * w + u) + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            voffA[u] = (row * lda + 8 * lc) * 2;
        }
#pragma unroll
        for (int u = 0; u < NPB; ++u) {
            const int rho = 8 * (NPB * w + u) + (lane >> 3);
            const int lc = (lane & 7) ^ ((rho >> 1) & 7);
            const int wnB = rho / 96, within = rho % 96, t = within >> 4, ii = within & 15;
            const int src = wnB * 96 + 32 * (t >> 1) + 8 * (ii >> 2) + 4 * (t & 1) + (ii & 3);
            voffB[u] = (src * ldb + 8 * lc) * 2;
        }
        Cursor ca, cb, cl;       // next A / B K-tile to request; cl: the K-tile the consumers are on
        start(ca);
        start(cb);
        start(cl);
        int sa_i = 0, sb_i = 0;  // slots being filled
        auto issueA = [&]() {
            const int so = (ca.m0 * lda + ca.kt * BK) * 2;
            const int base = a_slot_off(sa_i) + w * (NPA * 1024);
#pragma unroll
            for (int u = 0; u < NPA; ++u)
#ifndef NT3_ABLATE_DMA
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + base + u * 1024), 16, voffA[u], so, 0, 0);
#else
                ;
#endif
            sa_i = sa_i + 1 == NAS ? 0 : sa_i + 1;
            advance(ca);
        };
        auto issueB = [&]() {
            const int so = (cb.n0 * ldb + cb.kt * BK) * 2;
            const int base = B_BASE + b_slot_off(sb_i) + w * (NPB * 1024);
#pragma unroll
            for (int u = 0; u < NPB; ++u)
#ifndef NT3_ABLATE_DMA
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(smem + base + u * 1024), 16, voffB[u], so, 0, 0);
#else
                ;
#endif
            sb_i = sb_i + 1 == NBS ? 0 : sb_i + 1;
            advance(cb);
        };
        // B(0), A(0), A(1); then at K-tile t: B(t+1), A(t+2) -- the slots the consumers left at
        // the previous barrier -- and a wait that leaves only A(t+2) outstanding
        // (general ring depths: B(0..NBS-2), A(0..NAS-2) in the steady state's order, then B(t+NBS-1), A(t+NAS-1))
        constexpr int DA = NAS - 1, DB = NBS - 1, DMAX = DA > DB ? DA : DB;
        // Work queue: loader wave 0 draws the items (lane 0), entry j + 2 at the first K-tile of item j.  The atomic
        // is issued in FRONT of that K-tile's requests and its value used behind the counted wait that the K-tile
        // ends with anyway (vmcnt retires in order: the wait that proves B(t + 1) has landed proves the draw has
        // returned), so it costs the loader nothing; the K-tile's barrier publishes the entry.  Issued from inline
        // asm: given the builtin, hipcc waits for the returned value on the spot (vmcnt(0): the whole pipeline of
        // requests drained once per item, +2..5 % per GEMM).  The register is not touched before the wait.
        auto draw_issue = [&]() -> unsigned {
            unsigned raw;
            asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(raw) : "v"(x8 * 4), "v"(1u), "s"(sched));
            return raw;
        };
        const float xinv = 1.f / (float)xcnt;
        auto draw_item = [&](unsigned raw) -> int {      // k-th item of the XCD's queue -> work-item index
            const unsigned kk = (unsigned)xcnt + raw;    // (the XCD's first xcnt items are the workgroups' own)
            int q = (int)((float)kk * xinv), r = (int)kk - q * xcnt;     // kk / xcnt, exact after one correction
            if (r < 0) {
                --q;
                r += xcnt;
            } else if (r >= xcnt) {
                ++q;
                r -= xcnt;
            }
            return xbase + r + q * G;
        };
        const bool drawer = QUEUE && w == 0;
        int newest = b;                                  // the newest ring entry (uniform; >= ntiles: the queue is empty)
        unsigned raw = 0;
        int pend = 0;
        if (drawer && lane == 0) raw = draw_issue();     // entry 1
        // (the prologue's requests all belong to item 0: the work queue is used from two K-tiles per item on)
        bool a_last = false;
#pragma unroll
        for (int j = -DMAX; j < 0; ++j) {
            if (j + DB >= 0) {
                resolve(cb);
                if (cb.tile < ntiles) issueB();
            }
            if (j + DA >= 0) {
                resolve(ca);
                a_last = ca.tile < ntiles;
                if (a_last) issueA();
            }
        }
        if (a_last && NAS == 3 && NBS == 2) wait_vmcnt<NPA>();   // (A(1) may still be in flight)
        else wait_vmcnt<0>();
        if (drawer) {
            if (lane == 0) {
                pend = draw_item(raw);
                seq_put(1, pend);
            }
            newest = __builtin_amdgcn_readfirstlane(pend);
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        __builtin_amdgcn_s_barrier();
        // (the LDS round trips of the queue stay off the path from one K-tile's requests to the next: a cursor that
        // has just left an item looks the next one up BEHIND this K-tile's requests, under the wait for them -- from
        // three K-tiles per item on, where that entry is always published already -- and a drawn entry is written
        // behind the wait and only waited for before the NEXT K-tile's barrier)
        const bool eager = QUEUE && nkt > 2;
        bool put_open = false;
        int cl_next = 0;
        while (cl.tile < ntiles) {
            const bool drawing = drawer && cl.kt == 0 && newest < ntiles;
            if (drawing && lane == 0) raw = draw_issue();    // entry cl.j + 2
            resolve(cb);
            if (cb.tile < ntiles) issueB();
            resolve(ca);
            const bool a_now = ca.tile < ntiles;
            if (a_now) issueA();
            if (eager) {
                resolve(cb);
                resolve(ca);
                if (cl.kt == 1) cl_next = item_at(cl.j + 1);
            }
            if (put_open) {
                asm volatile("s_waitcnt lgkmcnt(0)");
                put_open = false;
            }
            if (a_now) wait_vmcnt<VM_STEADY>();
            else wait_vmcnt<0>();
            if (drawing) {
                if (lane == 0) {
                    pend = draw_item(raw);
                    seq_put(cl.j + 2, pend);
                }
                newest = __builtin_amdgcn_readfirstlane(pend);
                if (eager) put_open = true;
                else asm volatile("s_waitcnt lgkmcnt(0)");
            }
            // tile end: the consumers' "all reads of K-tile t's slots are done" barrier (they
            // stage their output through the A slot; nothing is issued into it before the next one)
            advance(cl);
            if (cl.kt == 0) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
                if (eager) locate(cl, cl_next);
                else resolve(cl);     // (its ring entry was published at the first K-tile of the previous item)
            } else {
                __builtin_amdgcn_s_barrier();
            }
        }
        if (drawer && lane == 0) leave();
        return;
    }

    // ================= consumer waves =================
    const int wm = wv >> 1, wn = wv & 1, i = lane & 15, g = lane >> 4;
    const void* ldp = !HASLD ? nullptr : (p.residual ? p.residual : p.aux);
    const int ldl = p.residual ? (int)p.ldres : (int)p.ldaux;
    const __amdgpu_buffer_rsrc_t rsLd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)ldp, 0, ldp ? (int)(((int64_t)((p.res_mod ? p.res_mod : p.M) - 1) * ldl + p.N) * 2) : 0, 0x00020000);
    constexpr bool TOUCH = HASLD && !COLSUM && NT3_PF_DIST > 0;   // (the column-sum variants use the dump area)
    const int ld_bytes = ldp ? (int)(((int64_t)((p.res_mod ? p.res_mod : p.M) - 1) * ldl + p.N) * 2) : 0;
    const u32x4 rsTouch = raw_rsrc(ldp, ld_bytes);
    // (not with a bias longer than 2048 columns, which lies over the dump area, nor with repeating residual rows)
    const bool touch_on = TOUCH && !p.res_mod && !(p.bias && p.N > MAX_BIAS_SMALL);
    const bool has_aux_store = (ACT == 1 || ACT == 3 || ACT == 5 || ACT == 6) && p.aux;
    const int ldo = (int)p.ldo, ldaux = (int)p.ldaux;
    const __amdgpu_buffer_rsrc_t rsOut = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (int)(((int64_t)(p.M - 1) * p.ldo + p.N) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsAux = __builtin_amdgcn_make_buffer_rsrc(
        has_aux_store ? p.aux : nullptr, 0,
        has_aux_store ? (int)(((int64_t)(p.M - 1) * p.ldaux + p.N) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsRs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.rowscale, 0,
        p.rowscale ? (int)(((p.M + p.rows_per_group - 1) / p.rows_per_group) * 4) : 0, 0x00020000);
    const int lnparts = LN == 3 ? (p.ln_dim + 95) / 96 : 0;
    const __amdgpu_buffer_rsrc_t rsPs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(LN == 1 ? p.prescale : LN == 3 ? p.ln_stats_in : nullptr), 0,
        LN == 1 ? p.M * 4 : LN == 3 ? (int)((int64_t)lnparts * p.M * 8) : 0, 0x00020000);
    Cursor cc;
    start(cc);
    f32x4 acc[4][6];

    const int key = (i >> 1) & 7;
    const int offA0 = (wm * 64 + i) * 128 + ((g ^ key) << 4), offA1 = offA0 ^ 64;
    const int offB0 = B_BASE + (wn * 96 + i) * 128 + ((g ^ key) << 4), offB1 = offB0 ^ 64;
    int sa_c = 0, sb_c = 0;

    // The K-loop is software-pipelined by hand.  A K-tile is two 32-deep steps; each step runs as two
    // phases of twelve MFMAs -- columns 0-47 (fbL) and 48-95 (fbH) of the wave's 64 x 96 -- and every
    // fragment register is refilled for the NEXT step right after its last use in this one: fbL after
    // phase 0, fa[a] after row a of phase 1, fbH after phase 1.  Every ds_read then has >= 9 MFMAs
    // (>= 144 matrix-pipe cycles) between issue and first use; hipcc's own schedule of the plain loop
    // exposed the LDS latency four times per K-tile, for both waves of a SIMD at once (they leave the
    // barrier together).  The barrier sits after phase 0 of the second step -- every read of this
    // K-tile's slots was issued twelve MFMAs earlier -- so the last phase already refills from the
    // next K-tile's slots.  At the end of an output tile nothing is prefetched (the registers belong
    // to the epilogue); the next tile starts with one exposed load.
    u32x4 fa[4], fbL[3], fbH[3];
    auto lds16 = [&](int off) { return *(const u32x4*)(smem + off); };
    // (same issue order as the refills in the loop: the counted waits at the loop head are the
    // stricter of the two ways in)
    auto cold_load = [&](int aoff, int boff) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 3; ++c) fbL[c] = lds16(boff + offB0 + c * 2048);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            fa[a] = lds16(aoff + offA0 + a * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) fbH[c] = lds16(boff + offB0 + (3 + c) * 2048);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma3 = [&](const u32x4 (&fbx)[3], int a, int c0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#ifndef NT3_ABLATE_MFMA
            acc[a][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, fbx[c]), __builtin_bit_cast(bf16x8, fa[a]), acc[a][c0 + c], 0, 0, 0);
#else
            asm volatile("" : "+v"(acc[a][c0 + c]) : "v"(fbx[c]), "v"(fa[a]));
#endif
    };
    // phase 0: twelve MFMAs on fbL, then (refill) fbL of the next step from nb
    auto phase0 = [&](bool refill, int nb) {
#pragma unroll
        for (int a = 0; a < 4; ++a) mma3(fbL, a, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (refill) {
#pragma unroll
            for (int c = 0; c < 3; ++c) fbL[c] = lds16(nb + c * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // phase 1: three MFMAs per row on fbH, the row's fa of the next step (from na) behind them, then fbH (from nb)
    auto phase1 = [&](bool refill, int na, int nb) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            mma3(fbH, a, 3);
            __builtin_amdgcn_sched_barrier(0);
            if (refill) {
                fa[a] = lds16(na + a * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (refill) {
#pragma unroll
            for (int c = 0; c < 3; ++c) fbH[c] = lds16(nb + (3 + c) * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    __builtin_amdgcn_s_barrier();  // K-tile 0 has landed
    asm volatile("" ::: "memory");
    cold_load(0, 0);
    // Tiles outside, K-tiles inside: the inner loop of an MFMA wave is one straight line that carries
    // the accumulators and the fragments and nothing else (with the epilogue inside the same loop the
    // register allocator copied and spilled accumulators around its branches).
    while (cc.tile < ntiles) {
        // (the next item is looked up here, where the wave waits for the first fragments anyway: at the end of the
        // epilogue the LDS round trip of the work queue's ring would be exposed once per item)
        const int cc_next = item_at(cc.j + 1);
        const bool idle = cc.half && wm >= 2;  // half tile: rows 128-255 of the slot belong to nobody
        int sa_l = sa_c, sb_l = sb_c;          // slots of the tile's last K-tile: the epilogue stages through them
        // LN == 1: the per-row factors are requested NOW and arrive under the K-loop -- asked for at the top of the
        // epilogue (which otherwise issues no load in these variants) they cost a memory latency per tile, 13-25 us
        // per launch
        float psv[(LN == 1 || LN == 3) ? 4 : 1];
        float2 pst[LN == 3 ? 4 : 1];
        if (LN == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a)   // (rows >= M read 0: their stores are dropped anyway)
                psv[a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                             rsPs, (cc.m0 + wm * 64 + a * 16 + i) * 4, 0, 0));
        }
        if (LN == 3) {
            // the four g-lanes of a row each fetch a quarter of the row's partials (slices g, g + 4, ...); the
            // epilogue adds them across the lanes.  Rows >= M read zeros (range-checked descriptor).
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int m = cc.m0 + wm * 64 + a * 16 + i;
                // (at most four slices, ln_dim <= 384: one load per lane and row, nothing waited for here; a slice
                // that does not exist is asked for outside the descriptor and reads zeros)
                const auto w2 = __builtin_amdgcn_raw_buffer_load_b64(
                    rsPs, g < lnparts ? (g * p.M + m) * 8 : 0x7ffffff0, 0, 0);
                const unsigned w0 = w2[0], w1 = w2[1];   // (bit_cast straight from a vector element reads element 0: clang)
                pst[a].x = __builtin_bit_cast(float, w0);
                pst[a].y = __builtin_bit_cast(float, w1);
            }
        }
        // (cleared on every path, so that no accumulator is carried around the tile loop)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The barrier inside: every read of this K-tile's slots has returned (the loaders refill them
        // behind it, or, on the last K-tile of an output tile, the epilogue stages the output through
        // them), and the next K-tile has landed.  The refills behind it run on the last K-tile too
        // (what they fetch is dropped: the cold load after the epilogue replaces it).
        if (idle) {
            for (int kt = 0; kt < nkt; ++kt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                sa_l = sa_c;
                sb_l = sb_c;
                sa_c = sa_c + 1 == NAS ? 0 : sa_c + 1;
                sb_c = sb_c + 1 == NBS ? 0 : sb_c + 1;
            }
        } else {
            const int kt_touch = !touch_on ? -1 : nkt > NT3_PF_DIST ? nkt - NT3_PF_DIST : 0;
#pragma unroll 1
            for (int kt = 0; kt < nkt; ++kt) {
                if (TOUCH && kt == kt_touch) {
                    // this wave's 64 rows x 192 B of the operand tile = two lines per row: lane = row
                    int lane_t = lane;
                    asm volatile("" : "+v"(lane_t));
                    const int voff = ((cc.m0 + wm * 64 + lane_t) * ldl + cc.n0 + wn * 96) * 2;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                                 "buffer_load_dword %1, %2, 0 offen lds\n\t"
                                 "buffer_load_dword %1, %2, 0 offen offset:128 lds"
                                 :: "s"(COLSUM_BASE), "v"(voff), "s"(rsTouch) : "memory");
                }
                const int aoff = a_slot_off(sa_c), boff = b_slot_off(sb_c);
                const int sa_n = sa_c + 1 == NAS ? 0 : sa_c + 1, sb_n = sb_c + 1 == NBS ? 0 : sb_c + 1;
                phase0(true, boff + offB1);
                phase1(true, aoff + offA1, boff + offB1);
                phase0(false, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int nb = b_slot_off(sb_n) + offB0;
#pragma unroll
                for (int c = 0; c < 3; ++c) fbL[c] = lds16(nb + c * 2048);
                __builtin_amdgcn_sched_barrier(0);
                phase1(true, a_slot_off(sa_n) + offA0, nb);
                sa_l = sa_c;
                sb_l = sb_c;
                sa_c = sa_n;
                sb_c = sb_n;
            }
        }
        const char* ap = smem + a_slot_off(sa_l);
        const char* bp = smem + b_slot_off(sb_l);
        const bool last = true;
        if (last && idle) {
            ++cc.j;
            locate(cc, cc_next);
        } else if (last) {
            // the epilogue's lane-derived offsets are recomputed per tile from an opaque copy of the lane id:
            // hoisted out of the K-loop they would sit in ~25 registers that the pipelined loop needs
            int lane_op = lane;
            asm volatile("" : "+v"(lane_op));
            const int lane = lane_op, i = lane & 15, g = lane >> 4;
            char* stage = const_cast<char*>(ap) + wv * 4096;   // 16 rows x 208 B (A slot of this K-tile)
            char* stage2 = const_cast<char*>(bp) + B_BASE + wv * 3072;  // 16 rows x 192 B (its B slot): second output
            // ---- epilogue: lane (i, g) holds rows wm*64 + a*16 + i, columns wn*96 + 32c + 8g + 0..7,
            // c = 0..2: the four g-lanes of a row store 64 contiguous bytes.  Every operand load
            // (row scales, residual or saved pre-activation) goes out first, through range-checked
            // buffer descriptors (rows >= M read 0), so the tile pays one memory latency, not four.
            float rsv[4];
            u32x4 L[2][3];
            if (LN == 3) {
                const float invd = 1.f / (float)p.ln_dim;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float sm[2] = {pst[a].x, pst[a].y};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float x0 = sm[e], x1 = x0;
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
                        float t = x0 + x1;
                        x0 = x1 = t;
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
                        sm[e] = x0 + x1;
                    }
                    const float mu = sm[0] * invd;
                    const float rstd = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mu, mu, sm[1] * invd), 0.f) + p.ln_eps);
                    psv[a] = rstd;
                    const int m = cc.m0 + wm * 64 + a * 16 + i;
                    if (cc.n0 == 0 && wn == 0 && g == 0 && m < p.M) {
                        if (p.ln_mean_out) p.ln_mean_out[m] = mu;
                        if (p.ln_rstd_out) p.ln_rstd_out[m] = rstd;
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int m = cc.m0 + wm * 64 + a * 16 + i;
                rsv[a] = p.rowscale ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                          rsRs, (m / p.rows_per_group) * 4, 0, 0))
                                    : 1.f;
            }
            // two row fragments of operands in flight (48 more registers would not fit beside
            // the accumulators at three waves per SIMD)
            // the operand tile arrives the way the output leaves: 192-B row segments, adjacent lanes
            // adjacent in memory, and is turned into the accumulator layout through stage2
            // (the column-sum variants have no registers left for it and load in accumulator layout)
            auto load_frag = [&](int a, u32x4 (&dst)[3]) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int idx = j * 64 + lane, row = idx / 12, ch = idx - row * 12;
                    const int r = COLSUM ? i : row, col = COLSUM ? 32 * j + 8 * g : 8 * ch;
                    int mr = cc.m0 + wm * 64 + a * 16 + r;
                    if (p.res_mod) mr %= p.res_mod;   // residual rows repeat (the position table under every crop)
                    dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rsLd, (mr * ldl + cc.n0 + wn * 96 + col) * 2, 0, 0);
                }
            };
            auto turn_frag = [&](const u32x4 (&src)[3]) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int idx = j * 64 + lane, row = idx / 12, ch = idx - row * 12;
                    *(u32x4*)(stage2 + (COLSUM ? i * 192 + 64 * j + 16 * g : row * 192 + ch * 16)) = src[j];
                }
            };
            auto turned = [&](int a, int c) {
                return COLSUM ? L[a & 1][c] : *(const u32x4*)(stage2 + i * 192 + 64 * c + 16 * g);
            };
            if (HASLD) {
                load_frag(0, L[0]);
                load_frag(1, L[1]);
            }
            // Output transposition: a row fragment (16 rows x 192 B) goes through the wave's staging
            // area (row pitch 208 B: conflict-free both ways) and leaves as 192-B row segments with
            // adjacent lanes adjacent in memory -- 16 rows x 64-B segments per store instruction
            // (four lanes per row, one row per lane quad-group) drain 4x slower per CU.  LDS
            // operations of one wave execute in order: no wait between the write and the read.
            auto stage_put = [&](char* st, int pitch, int c, const u32x4& pk) {
                *(u32x4*)(st + i * pitch + 64 * c + 16 * g) = pk;
            };
            // read-back: 64 lanes x 16 B = 5.33 rows of 192 B per instruction (3 instructions); the
            // column-sum variants are out of registers and take lanes 0-47 as 4 rows x 12 chunks,
            // so that their four instructions differ by immediates / a scalar offset only
            const int trow = lane / 12, tch = lane - trow * 12;
            const int ncol = (int)p.N - cc.n0 - wn * 96;   // how many of this wave's 96 columns exist
            // LN == 2: the rows written here are the NEXT GEMM's A operand (no LayerNorm kernel in between any more):
            // default cache policy (DINO_V3_LN2_NT=1 at build time keeps the hint), so that they are still in the
            // Infinity Cache when it starts
#ifdef DINO_V3_LN2_NT
            constexpr int OUT_AUX = NT_AUX;
#else
            constexpr int OUT_AUX = LN == 2 ? 0 : NT_AUX;
#endif
            auto stage_flush = [&](const char* st, int pitch, const __amdgpu_buffer_rsrc_t& rs, int ld, int a) {
                if (COLSUM) {
                    if (lane < 48) {
                        const char* src = st + trow * pitch + tch * 16;
                        const int voff = (trow * ld + wn * 96 + 8 * tch) * 2;
                        const int soff = ((cc.m0 + wm * 64 + a * 16) * ld + cc.n0) * 2;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#ifndef NT3_ABLATE_STORE
                            __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)(src + 4 * j * pitch), rs, voff,
                                                                   soff + 4 * j * ld * 2, NT_AUX);
#else
                            ;
#endif
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int idx = j * 64 + lane, row = idx / 12, ch = idx - row * 12;
                        // columns >= N (last column tile of a width that is no multiple of 192): the
                        // offset is pushed outside the descriptor and the store is dropped, like rows >= M
                        const int off = ((cc.m0 + wm * 64 + a * 16 + row) * ld + cc.n0 + wn * 96 + 8 * ch) * 2;
#ifndef NT3_ABLATE_STORE
                        __builtin_amdgcn_raw_buffer_store_b128(
                            *(const u32x4*)(st + row * pitch + ch * 16), rs, 8 * ch < ncol ? off : 0x7fffff00, 0, OUT_AUX);
#else
                        (void)off;
#endif
                    }
                }
            };
            float csum[3][8];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 8; ++r) csum[c][r] = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int m = cc.m0 + wm * 64 + a * 16 + i;
                {   // rows >= M: the stores fall outside their descriptors and are dropped
                    const float rs = rsv[a];
                    const float ors = p.out_rowscale ? rs : 1.f;
                    float st_s = 0.f, st_q = 0.f;
                    if (HASLD && !COLSUM) turn_frag(L[a & 1]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int n = cc.n0 + wn * 96 + 32 * c + 8 * g;
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc[a][2 * c][r];
                            v[4 + r] = acc[a][2 * c + 1][r];
                        }
                        if (LN == 1 || LN == 3) {
                            // rstd * acc + bias as ONE fma per element: every VALU instruction of this epilogue is
                            // exposed time (96 of them per lane and tile = +10 us per launch)
                            const f32x4 b0 = p.bias ? *(const f32x4*)(sbias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                            const f32x4 b1 = p.bias ? *(const f32x4*)(sbias + n + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = fmaf(v[r], psv[a], b0[r]);
                                v[4 + r] = fmaf(v[4 + r], psv[a], b1[r]);
                            }
                        } else if (p.bias) {
                            const float bs = p.bias_rowscale ? rs : 1.f;
                            const f32x4 b0 = *(const f32x4*)(sbias + n), b1 = *(const f32x4*)(sbias + n + 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] += b0[r] * bs;
                                v[4 + r] += b1[r] * bs;
                            }
                        }
                        if (ACT == 1) {
                            if (has_aux_store) stage_put(stage2, 192, c, pack8(v));
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] = gelu_t<bf16>(v[r]);
                        } else if (ACT == 2) {
                            float pre[8];
                            unpack8(turned(a, c), pre);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] *= gelu_grad_t<bf16>(pre[r]);
                        } else if (ACT == 3) {
                            float dg[8];
#pragma unroll
                            for (int r = 0; r < 8; ++r) gelu_both_t<bf16>(v[r], v[r], dg[r]);
                            if (has_aux_store) stage_put(stage2, 192, c, pack8(dg));
                        } else if (ACT == 5) {
                            if (has_aux_store) stage_put(stage2, 192, c, pack8(v));
                            float dg[8];
                            gelu_lut8<false>(v, lut_adj, v, dg);
                        } else if (ACT == 6) {
                            float dg[8];
                            gelu_lut8<true>(v, lut_adj, v, dg);
                            if (has_aux_store) stage_put(stage2, 192, c, pack8(dg));
                        } else if (ACT == 4) {
                            float dg[8];
                            unpack8(turned(a, c), dg);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] *= dg[r];
                        }
                        {   // (unconditional: as `if (p.out_rowscale)` it became a multiply AND a select per element)
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] *= ors;
                        }
                        if (HASLD && ACT != 2 && ACT != 4) {
                            float rv[8];
                            unpack8(turned(a, c), rv);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] += rv[r];
                        }
                        if (COLSUM) {
                            const bool live = m < p.M;
#pragma unroll
                            for (int r = 0; r < 8; ++r) csum[c][r] += live ? v[r] : 0.f;
                        }
                        if (LN == 2) {
                            if (ncol >= 96) {            // (wave-uniform: all of this wave's 96 columns exist)
#pragma unroll
                                for (int r = 0; r < 8; ++r) {
                                    st_s += v[r];
                                    st_q = fmaf(v[r], v[r], st_q);
                                }
                            } else {                     // a ragged last column tile
#pragma unroll
                                for (int r = 0; r < 8; ++r) {
                                    const float x = 32 * c + 8 * g + r < ncol ? v[r] : 0.f;
                                    st_s += x;
                                    st_q = fmaf(x, x, st_q);
                                }
                            }
                        }
                        stage_put(stage, 208, c, pack8(v));
                    }
                    if (LN == 2) {
                        // the four g-lanes of a row hold 24 columns each: add across the 16-lane rows of the wave
                        // (v_permlane16_swap / v_permlane32_swap: no LDS traffic), lanes g == 0 store the pair
                        float2 pr = {st_s, st_q};
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            // (inline asm: given the same value for both operands, hipcc 7.2 folds the builtin's
                            // two results into one -- it emitted s16[0] + s16[0]; the nops cover the VALU -> swap
                            // hazard the compiler would otherwise pad)
                            float x0 = e ? pr.y : pr.x, x1 = x0;
                            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
                            float t = x0 + x1;
                            x0 = x1 = t;
                            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
                            t = x0 + x1;
                            if (e) pr.y = t; else pr.x = t;
                        }
                        const int q = (cc.n0 + wn * 96) / 96;
                        if (g == 0 && m < p.M && ncol > 0)
                            *(float2*)(p.ln_stats + ((int64_t)q * p.M + m) * 2) = pr;
                    }
                    if ((ACT == 1 || ACT == 3 || ACT == 5 || ACT == 6) && has_aux_store) stage_flush(stage2, 192, rsAux, ldaux, a);
                    stage_flush(stage, 208, rsOut, ldo, a);
                }
                if (HASLD && a + 2 < 4) load_frag(a + 2, L[a & 1]);
            }
            if (COLSUM) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int n = cc.n0 + wn * 96 + 32 * c + 8 * g;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float x = csum[c][r];
                        x = row16_sum(x);
                        if (i == 0) {
                            if (persist_colsum) atomicAdd(scol + wn * 96 + 32 * c + 8 * g + r, x);
                            else atomicAdd(p.colsum + n + r, x);
                        }
                    }
                }
            }
            ++cc.j;
            locate(cc, cc_next);
        }
        // the staging areas are free again: the loaders refill the slots behind this barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cold_load(a_slot_off(sa_c), b_slot_off(sb_c));   // next tile's K-tile 0 (landed before the barrier above)
    }
    if (persist_colsum) {  // the loaders have left: this barrier joins the consumer waves only
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < BN) atomicAdd(p.colsum + (b % tiles_n) * BN + tid, scol[tid]);
    }
#endif
}

int num_cus_v3() {
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

bool dino_gemm_nt_v3_supported(const dino_gemm_args& a) {
    const int64_t ba = ((int64_t)a.M * a.lda + a.K) * 2, bb = ((int64_t)a.N * a.ldb + a.K) * 2;
    if (a.dtype != DINO_BF16 || a.trans || a.splitk > 1 || a.out_mode != 0) return false;
    // widths that are no multiple of 192 (ViT-L: 1024, 4096): the last column tile is ragged -- W rows
    // past N load as zeros, output columns past N are dropped; not in the column-sum variants
    if (a.K % BK != 0 || a.N % 8 != 0 || (a.N % BN != 0 && a.colsum)) return false;
    const bool pres = a.prescale || a.ln_stats_in;
    if (a.prescale && a.ln_stats_in) return false;
    if (a.ln_stats_in && (a.ln_dim <= 0 || a.ln_dim > 384 || (int64_t)((a.ln_dim + 95) / 96) * a.M * 8 >= (1ll << 31))) return false;
    if (pres && (a.colsum || a.residual || a.act == 2 || a.act == 4 || a.ln_stats)) return false;
    // (GELU table + a bias longer than 2048 columns: no LDS for both -- dino_gemm_nt_v3 runs such a launch as column ranges)
    if (a.ln_stats && (a.colsum || a.act != 0)) return false;
    if (a.bias && a.N > (a.colsum ? MAX_BIAS_SMALL : MAX_BIAS)) return false;
    if (ba >= (1ll << 31) || bb >= (1ll << 31)) return false;
    // epilogue operands are read through 32-bit buffer offsets, rows map one to one
    if (a.row_map_n || (a.res_mod && !a.residual) || (a.act != 0 && a.residual) || ((a.act == 2 || a.act == 4) && !a.aux)) return false;
    const int64_t ldl = a.residual ? a.ldres : a.ldaux;
    if ((a.residual || a.act == 2 || a.act == 4) && (ldl % 8 != 0 || ((int64_t)a.M + BM) * ldl * 2 >= (1ll << 31))) return false;
    if (((int64_t)a.M + BM) * a.ldo * 2 >= (1ll << 31) || a.ldo % 8 != 0) return false;
    if (a.aux && (((int64_t)a.M + BM) * a.ldaux * 2 >= (1ll << 31) || a.ldaux % 8 != 0)) return false;
    // small problems keep the 256-thread kernel (two workgroups per CU, finer tiles)
    const int64_t tiles = (int64_t)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    // from half a chip's worth of tiles on, the wide tile beats the 256-thread kernel (ViT-S at 64
    // images per GPU: +16 % step throughput)
    static const int min_tiles = getenv("DINO_V3_MIN_TILES") ? atoi(getenv("DINO_V3_MIN_TILES")) : num_cus_v3() / 2;
    return tiles >= min_tiles;
}

// Work queue on / off (dino_gemm_work_queue; env DINO_NT_QUEUE=0/1 sets the initial state, default off: with the
// chip to itself a launch is 1-2 % faster on static shares -- step 48.1 against 48.3 ms --, with another stream's
// kernel on even 8 CUs it is 1.5 x slower, see tools/bench_nt_occupied.py and include/dino_hip.h for when it pays).
static std::atomic<int> nt3_queue_on{(getenv("DINO_NT_QUEUE") && atoi(getenv("DINO_NT_QUEUE")) != 0) ? 1 : 0};

extern "C" int dino_gemm_work_queue(int on) {
    const int was = nt3_queue_on.load();
    if (on >= 0) nt3_queue_on.store(on ? 1 : 0);
    return was;
}

// One queue slot per launch, taken round robin (a slot is cleared by the last workgroup of the launch that used it;
// 1024 launches later nothing of that launch is left).
static unsigned* nt3_sched_slot() {
    if (!nt3_queue_on.load(std::memory_order_relaxed)) return nullptr;
    static std::mutex mu;
    static unsigned* base[16] = {};
    static std::atomic<unsigned> next{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!base[dev]) {
        std::lock_guard<std::mutex> lk(mu);
        if (!base[dev]) {
            void* ptr = nullptr;
            if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(nt3_sched_mem)) != hipSuccess) return nullptr;
            base[dev] = (unsigned*)ptr;
        }
    }
    return base[dev] + (size_t)(next.fetch_add(1) % SCHED_SLOTS) * SCHED_SLOT;
}

template <int ACT, bool HASLD, bool COLSUM, int LN = 0>
static int launch_v3(const dino_gemm_args& a, hipStream_t st) {
    static bool configured = false;  // (one device per process)
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_bf16_v3<ACT, HASLD, COLSUM, LN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES_LONG_BIAS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int grid = ntiles < num_cus_v3() ? ntiles : num_cus_v3();
    // a last round that fills at most half the CUs runs as half tiles
    const int rem = ntiles % grid;
    const int nhalf = (ntiles > grid && rem > 0 && 2 * rem <= grid) ? rem : 0;
    const bool long_bias = !(ACT == 5 || ACT == 6) && a.bias && a.N > MAX_BIAS_SMALL;
    // the work queue: not with column sums gathered per workgroup (they rely on a fixed column tile), not where the
    // ring has no LDS left (long bias), not below two K-tiles per item (the prologue would need item 1), and not
    // when every workgroup has one item anyway
    unsigned* sched = (COLSUM || long_bias || a.K < 2 * BK || ntiles <= grid) ? nullptr : nt3_sched_slot();
    const int smem = ((ACT == 5 || ACT == 6) ? SMEM_BYTES_LUT : long_bias ? SMEM_BYTES_LONG_BIAS : SMEM_BYTES) +
                     (sched ? 64 : 0);
    if constexpr (!COLSUM) {
        if (sched) {
            static bool configured_q = false;
            if (!configured_q) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_bf16_v3<ACT, HASLD, COLSUM, LN, true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES_LONG_BIAS);
                if (e != hipSuccess) return (int)e;
                configured_q = true;
            }
            hipLaunchKernelGGL((gemm_nt_bf16_v3<ACT, HASLD, COLSUM, LN, true>), dim3(grid), dim3(NTHR), smem, st, a,
                               tiles_n, ntiles - nhalf, nhalf, sched);
            return dino_launch_status();
        }
    }
    hipLaunchKernelGGL((gemm_nt_bf16_v3<ACT, HASLD, COLSUM, LN, false>), dim3(grid), dim3(NTHR), smem, st, a, tiles_n,
                       ntiles - nhalf, nhalf, nullptr);
    return dino_launch_status();
}

// GELU by table AND a bias longer than 2048 columns (ViT-B's / ViT-L's fc1: 3072 / 4096) do not fit the 160 KiB beside
// the operand rings.  Such a launch runs as two or three column ranges of at most 1920 columns (multiples of the
// 192-column tile), each an ordinary launch on its own rows of W, bias, output and saved GELU': the token matrix is
// read once per range, which costs less than the arithmetic GELU epilogue and -- for the consumer of a folded
// LayerNorm, which only exists with the table -- than a LayerNorm pass over the tokens.
static bool nt3_lut_on() {
    static const bool on = !(getenv("DINO_GELU_LUT") && atoi(getenv("DINO_GELU_LUT")) == 0);
    return on;
}
static bool nt3_split_columns(const dino_gemm_args& a) {
    return a.bias && a.N > MAX_BIAS_SMALL && (a.act == 1 || a.act == 3) && !a.residual && !a.ln_stats &&
           (nt3_lut_on() || a.prescale || a.ln_stats_in);
}

int dino_gemm_nt_v3(const dino_gemm_args& a, hipStream_t st) {
    static const bool lut_on = nt3_lut_on();
    if (nt3_split_columns(a)) {
        int nch = (a.N + 1919) / 1920, per = 0;
        for (;; ++nch) {
            per = ((a.N + nch - 1) / nch + BN - 1) / BN * BN;
            if (per <= 1920) break;
        }
        for (int n0 = 0; n0 < a.N; n0 += per) {
            dino_gemm_args b = a;
            b.N = a.N - n0 < per ? a.N - n0 : per;
            b.B = (const char*)a.B + (int64_t)n0 * a.ldb * 2;
            b.out = (char*)a.out + (int64_t)n0 * 2;
            b.bias = a.bias + n0;
            if (a.aux) b.aux = (char*)a.aux + (int64_t)n0 * 2;
            if (a.colsum) b.colsum = a.colsum + n0;
            if (int e = dino_gemm_nt_v3(b, st)) return e;
        }
        return 0;
    }
    if (a.prescale) {   // consumer of a folded LayerNorm: plain (qkv) or GELU by table (fc1), no column sums
        if (a.bias && a.N > MAX_BIAS_SMALL && a.act != 0) return DINO_ERR_UNSUPPORTED;
        switch (a.act) {
            case 0: return launch_v3<0, false, false, 1>(a, st);
            case 1: return launch_v3<5, false, false, 1>(a, st);
            case 3: return launch_v3<6, false, false, 1>(a, st);
            default: return DINO_ERR_UNSUPPORTED;
        }
    }
    if (a.ln_stats_in) {   // the same, the factor formed from the producer's partials in the epilogue
        if (a.bias && a.N > MAX_BIAS_SMALL && a.act != 0) return DINO_ERR_UNSUPPORTED;
        switch (a.act) {
            case 0: return launch_v3<0, false, false, 3>(a, st);
            case 1: return launch_v3<5, false, false, 3>(a, st);
            case 3: return launch_v3<6, false, false, 3>(a, st);
            default: return DINO_ERR_UNSUPPORTED;
        }
    }
    if (a.ln_stats) return a.residual ? launch_v3<0, true, false, 2>(a, st) : launch_v3<0, false, false, 2>(a, st);
    if (lut_on && (a.act == 1 || a.act == 3) && !(a.bias && a.N > MAX_BIAS_SMALL)) {
        if (a.act == 1) return a.colsum ? launch_v3<5, false, true>(a, st) : launch_v3<5, false, false>(a, st);
        return a.colsum ? launch_v3<6, false, true>(a, st) : launch_v3<6, false, false>(a, st);
    }
    switch (a.act) {
        case 1: return a.colsum ? launch_v3<1, false, true>(a, st) : launch_v3<1, false, false>(a, st);
        case 2: return a.colsum ? launch_v3<2, true, true>(a, st) : launch_v3<2, true, false>(a, st);
        case 3: return a.colsum ? launch_v3<3, false, true>(a, st) : launch_v3<3, false, false>(a, st);
        case 4: return a.colsum ? launch_v3<4, true, true>(a, st) : launch_v3<4, true, false>(a, st);
        default: break;
    }
    if (a.residual) return a.colsum ? launch_v3<0, true, true>(a, st) : launch_v3<0, true, false>(a, st);
    return a.colsum ? launch_v3<0, false, true>(a, st) : launch_v3<0, false, false>(a, st);
}
