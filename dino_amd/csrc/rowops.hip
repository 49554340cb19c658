// HBM-bound row kernels: LayerNorm fwd/bwd, L2-normalise fwd/bwd, weight-norm fwd/bwd,
// column sums (bias grads), row folding (pos/cls grads), scalar scaling.
// One 64-lane wave per row, 16-byte accesses, two-pass statistics held in registers.
#include <mutex>

#include "common.h"

namespace {

constexpr int MAXC = 4;  // 16-B chunks per lane -> D <= 64*4*8 = 2048

DEVI void load8f(const float* p, float (&v)[8]) { load8<float>(p, v); }

// ------------------------------------------------------------------ LayerNorm forward
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int64_t xs,
                                                     T* __restrict__ y, int64_t ys,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows,
                                                     int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int nch = D >> 3;
    const float invD = 1.f / (float)D;
    for (int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); row < rows;
         row += (int64_t)gridDim.x * 4) {
        float v[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                load8<T>(x + row * xs + ch * 8, v[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[c][e];
            }
        }
        const float mu = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (lane + 64 * c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[c][e] - mu;
                    q += d * d;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(q) * invD + eps);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float gm[8], bt[8], o[8];
                load8f(gamma + ch * 8, gm);
                load8f(beta + ch * 8, bt);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mu) * rs * gm[e] + bt[e];
                store8<T>(y + row * ys + ch * 8, o);
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// D <= 512 (one 16-byte chunk per lane): each wave takes R consecutive rows per iteration with all
// R loads in flight (one 768-byte row per wave is too little outstanding traffic to fill HBM),
// gamma / beta stay in registers.  Same per-row arithmetic (and summation order) as ln_fwd_kernel.
template <typename T> struct Raw8;
template <> struct Raw8<bf16> {
    u32x4 v;
    DEVI void ld(const bf16* p) { v = *(const u32x4*)p; }
    DEVI void zero() { v = u32x4{0u, 0u, 0u, 0u}; }
    DEVI void get(float (&o)[8]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[2 * r] = __builtin_bit_cast(float, v[r] << 16);
            o[2 * r + 1] = __builtin_bit_cast(float, v[r] & 0xffff0000u);
        }
    }
    // the compiler must not keep the unpacked copy of every row alive from one use to the next
    DEVI void opaque() { asm volatile("" : "+v"(v)); }
};
template <> struct Raw8<float> {
    f32x4 a, b;
    DEVI void ld(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    DEVI void zero() { a = b = f32x4{0.f, 0.f, 0.f, 0.f}; }
    DEVI void opaque() {}
    DEVI void get(float (&o)[8]) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
    }
};

template <typename T, int R>
__global__ __launch_bounds__(256) void ln_fwd_small_kernel(const T* __restrict__ x, int64_t xs,
                                                           T* __restrict__ y, int64_t ys,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ mean,
                                                           float* __restrict__ rstd, int64_t rows,
                                                           int D, float eps) {
    const int lane = threadIdx.x & 63;
    const bool act = lane < (D >> 3);
    const float invD = 1.f / (float)D;
    float gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gm[e] = bt[e] = 0.f;
    if (act) {
        load8f(gamma + lane * 8, gm);
        load8f(beta + lane * 8, bt);
    }
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * R; row0 < rows;
         row0 += (int64_t)gridDim.x * 4 * R) {
        Raw8<T> raw[R];   // rows stay as loaded (half the registers in bf16: twice the rows in flight)
        float mu[R], rs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            raw[r].zero();
            if (act && row0 + r < rows) raw[r].ld(x + (row0 + r) * xs + lane * 8);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float v[8], s = 0.f;
            raw[r].get(v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e];
            mu[r] = wave_sum(s) * invD;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float q = 0.f;
            raw[r].opaque();
            if (act) {
                float v[8];
                raw[r].get(v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[e] - mu[r];
                    q += d * d;
                }
            }
            rs[r] = rsqrtf(wave_sum(q) * invD + eps);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            raw[r].opaque();
            if (act && row0 + r < rows) {
                float v[8], o[8];
                raw[r].get(v);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu[r]) * rs[r] * gm[e] + bt[e];
                store8<T>(y + (row0 + r) * ys + lane * 8, o);
            }
            if (lane == r && row0 + r < rows) {
                mean[row0 + r] = mu[r];
                rstd[row0 + r] = rs[r];
            }
        }
    }
}

// 512 < D <= 512*NC (ViT-B 768, ViT-L 1024): NC 16-byte chunks per lane, R rows in flight per wave,
// gamma / beta in registers; per-row arithmetic and summation order of ln_fwd_kernel (chunk-major).
// (One row per wave-iteration left the generic kernel at 1.9 TB/s on 1.5-KB rows.)
template <typename T, int NC, int R>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(const T* __restrict__ x, int64_t xs,
                                                          T* __restrict__ y, int64_t ys,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ mean,
                                                          float* __restrict__ rstd, int64_t rows,
                                                          int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int nch = D >> 3;
    const float invD = 1.f / (float)D;
    float gm[NC][8], bt[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gm[c][e] = bt[c][e] = 0.f;
        if (lane + 64 * c < nch) {
            load8f(gamma + (lane + 64 * c) * 8, gm[c]);
            load8f(beta + (lane + 64 * c) * 8, bt[c]);
        }
    }
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * R; row0 < rows;
         row0 += (int64_t)gridDim.x * 4 * R) {
        Raw8<T> raw[R][NC];   // rows stay as loaded (see ln_fwd_small_kernel)
        float mu[R], rs[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                raw[r][c].zero();
                if (lane + 64 * c < nch && row0 + r < rows)
                    raw[r][c].ld(x + (row0 + r) * xs + (lane + 64 * c) * 8);
            }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float v[8];
                raw[r][c].get(v);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
            }
            mu[r] = wave_sum(s) * invD;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                raw[r][c].opaque();
                if (lane + 64 * c < nch) {
                    float v[8];
                    raw[r][c].get(v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = v[e] - mu[r];
                        q += d * d;
                    }
                }
            }
            rs[r] = rsqrtf(wave_sum(q) * invD + eps);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                raw[r][c].opaque();
                if (lane + 64 * c < nch && row0 + r < rows) {
                    float v[8], o[8];
                    raw[r][c].get(v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu[r]) * rs[r] * gm[c][e] + bt[c][e];
                    store8<T>(y + (row0 + r) * ys + (lane + 64 * c) * 8, o);
                }
            }
            if (lane == r && row0 + r < rows) {
                mean[row0 + r] = mu[r];
                rstd[row0 + r] = rs[r];
            }
        }
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// NC = 16-byte chunks per lane (D <= 512*NC).  Each wave walks rows two at a time (two rows'
// loads in flight); dgamma / dbeta / column sums of dx (the bias gradient of the linear layer
// that produced this residual stream) are kept in registers, combined across the block's four
// waves through LDS and added with one atomic per column per block.
// XN: also write the normalised rows xn = xhat * gamma + beta (dino_layernorm_bwd_xn)
#ifndef LNB_R      // experiment switches (tools/bench_ln_xn.py): rows per wave-iteration and waves per SIMD asked of
#define LNB_R 2    // the register allocator for the D <= 512 xn-writing form
#endif
#ifndef LNB_OCC
#define LNB_OCC 4
#endif
template <typename T, int NC, int R, bool XN = false>
__global__ __launch_bounds__(256, (XN && NC == 1) ? LNB_OCC : 1) void ln_bwd_kernel(
    const T* __restrict__ dy, int64_t dys, const T* __restrict__ x, int64_t xs,
    const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, const T* __restrict__ dres, int64_t drs, T* __restrict__ dx,
    int64_t dxs, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dxsum, const float* __restrict__ rowscale, int rpg, int64_t rows, int D,
    float* __restrict__ part, T* __restrict__ xn, int64_t xns, const float* __restrict__ beta) {
    extern __shared__ __attribute__((aligned(16))) char ln_smem[];
    float* red = (float*)ln_smem;  // [3 kinds][4 waves][nch*8]
    // (the wave index as a scalar: a wave's rows are then uniform, and the per-row mean / rstd / row scale come through
    // the scalar cache into SGPRs instead of as three or four vector-memory instructions per row)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = D >> 3;
    const float invD = 1.f / (float)D;
#ifndef LNB_BETA_GLOBAL
    float* sbeta = red + 12 * D;           // (behind the reduction scratch; the launch adds D floats)
    if (XN) {
        for (int o = threadIdx.x; o < D; o += 256) sbeta[o] = beta[o];
        __syncthreads();
    }
#endif
    float ag[NC][8], ab[NC][8], as[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) ag[c][e] = ab[c][e] = as[c][e] = 0.f;
    float gm[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) load8f(gamma + ch * 8, gm[c]);
    }
    const int64_t stride = (int64_t)gridDim.x * 4 * R;
    for (int64_t row0 = (int64_t)blockIdx.x * 4 * R + wave * R; row0 < rows; row0 += stride) {
        float xh[R][NC][8];
        Raw8<T> dyr[R][NC];  // dy stays as loaded (gamma * dy is recomputed below: half the registers in bf16)
        Raw8<T> rr[R][NC];   // the residual-stream gradient: in flight with x and dy, not after the reductions
        constexpr bool PRE = sizeof(T) == 2;   // (fp32 rows: the registers are worth more than the overlap)
        float s1[R], s2[R];
        float rsd[R], scl[R];   // rstd and the bias-gradient row scale: loaded once, with the row (they were re-read
                                // in front of the stores: two more memory instructions per row on the path to them)
#pragma unroll
        for (int u = 0; u < R; ++u) {
            s1[u] = s2[u] = 0.f;
            rsd[u] = 0.f;
            scl[u] = 1.f;
            const int64_t row = row0 + u;
            if (row < rows) {
                const float mu = mean[row], rs = rstd[row];
                rsd[u] = rs;
                if (dxsum && rowscale) scl[u] = rowscale[row / rpg];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int ch = lane + 64 * c;
                    if (ch < nch) {
                        float xv[8], dv[8];
                        load8<T>(x + row * xs + ch * 8, xv);
                        dyr[u][c].ld(dy + row * dys + ch * 8);
                        if constexpr (PRE) { if (dres) rr[u][c].ld(dres + row * drs + ch * 8); }
                        dyr[u][c].get(dv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) xh[u][c][e] = (xv[e] - mu) * rs;
                        if (XN) {   // the normalised row itself (LayerNorm folded into the GEMMs: the forward
                            float bt[8], o[8];   // pass never wrote it, the weight gradient wants it)
#ifndef LNB_BETA_GLOBAL   // beta from an LDS copy: as two cached global loads per row and lane it cost 23 of 185 us
                            {
                                const f32x4 b0 = *(const f32x4*)(sbeta + ch * 8), b1 = *(const f32x4*)(sbeta + ch * 8 + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { bt[e] = b0[e]; bt[4 + e] = b1[e]; }
                            }
#else
                            load8f(beta + ch * 8, bt);
#endif
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = fmaf(xh[u][c][e], gm[c][e], bt[e]);
                            store8<T>(xn + row * xns + ch * 8, o);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float wde = dv[e] * gm[c][e];
                            s1[u] += wde * xh[u][c][e];
                            s2[u] += wde;
                            ag[c][e] += dv[e] * xh[u][c][e];
                            ab[c][e] += dv[e];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t row = row0 + u;
            const float c1 = wave_sum(s1[u]) * invD, c2 = wave_sum(s2[u]) * invD;
            if (row < rows) {
                const float rs = rsd[u];
                const float sc = scl[u];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int ch = lane + 64 * c;
                    if (ch < nch) {
                        float o[8], dv[8];
                        dyr[u][c].get(dv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = rs * (dv[e] * gm[c][e] - c2 - xh[u][c][e] * c1);
                        if (dres) {
                            float r[8];
                            if constexpr (PRE) rr[u][c].get(r);
                            else load8<T>(dres + row * drs + ch * 8, r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] += r[e];
                        }
                        store8<T>(dx + row * dxs + ch * 8, o);
                        if (dxsum) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) as[c][e] += o[e] * sc;
                        }
                    }
                }
            }
        }
    }
    if (dgamma == nullptr && dxsum == nullptr) return;
    const int per = nch * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(0 * 4 + wave) * per + ch * 8 + e] = ag[c][e];
                red[(1 * 4 + wave) * per + ch * 8 + e] = ab[c][e];
                red[(2 * 4 + wave) * per + ch * 8 + e] = as[c][e];
            }
        }
    }
    __syncthreads();
    if (part) {  // per-workgroup partial sums, combined by ln_bwd_reduce_kernel: no same-address atomics
        float* mine = part + (int64_t)blockIdx.x * 3 * per;
        for (int o = threadIdx.x; o < 3 * per; o += 256) {
            const int kind = o / per, col = o - kind * per;
            const float* r = red + kind * 4 * per + col;
            mine[o] = r[0] + r[per] + r[2 * per] + r[3 * per];
        }
        return;
    }
    for (int o = threadIdx.x; o < per; o += 256) {
        if (dgamma) {
            atomicAdd(dgamma + o, red[0 * per + o] + red[1 * per + o] + red[2 * per + o] + red[3 * per + o]);
            atomicAdd(dbeta + o, red[4 * per + o] + red[5 * per + o] + red[6 * per + o] + red[7 * per + o]);
        }
        if (dxsum)
            atomicAdd(dxsum + o, red[8 * per + o] + red[9 * per + o] + red[10 * per + o] + red[11 * per + o]);
    }
}

// part[nblk][3][per] -> dgamma / dbeta / dxsum (+=).  blockIdx.y splits the workgroup partials 64 ways.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int per,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta,
                                                            float* __restrict__ dxsum) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= 3 * per) return;
    const int kind = o / per, col = o - kind * per;
    float* dst = kind == 0 ? dgamma : (kind == 1 ? dbeta : dxsum);
    if (dst == nullptr) return;
    const int chunk = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * chunk, b1 = min(nblk, b0 + chunk);
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += part[(int64_t)b * 3 * per + o];
    if (b1 > b0) atomicAdd(dst + col, s);
}

// ------------------------------------------------------------------ L2 normalise
template <typename T>
__global__ __launch_bounds__(256) void l2n_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                      float* __restrict__ norm, int64_t rows, int D) {
    const int lane = threadIdx.x & 63, nch = D >> 3;
    for (int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); row < rows;
         row += (int64_t)gridDim.x * 4) {
        float v[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                load8<T>(x + row * D + ch * 8, v[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[c][e] * v[c][e];
            }
        }
        const float nrm = sqrtf(wave_sum(s));
        const float inv = 1.f / fmaxf(nrm, 1e-12f);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[c][e] * inv;
                store8<T>(y + row * D + ch * 8, o);
            }
        }
        if (lane == 0) norm[row] = nrm;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void l2n_bwd_kernel(const float* __restrict__ dy,
                                                      const T* __restrict__ y,
                                                      const float* __restrict__ norm,
                                                      T* __restrict__ dx, int64_t rows, int D) {
    const int lane = threadIdx.x & 63, nch = D >> 3;
    for (int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); row < rows;
         row += (int64_t)gridDim.x * 4) {
        float yv[MAXC][8], dv[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                load8<T>(y + row * D + ch * 8, yv[c]);
                load8f(dy + row * D + ch * 8, dv[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += yv[c][e] * dv[c][e];
            }
        }
        const float dot = wave_sum(s);
        const float inv = 1.f / fmaxf(norm[row], 1e-12f);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (dv[c][e] - yv[c][e] * dot) * inv;
                store8<T>(dx + row * D + ch * 8, o);
            }
        }
    }
}

// ------------------------------------------------------------------ weight norm
// Block = 32 rows of v.  Phase 1: wave w handles rows 8w..8w+7 (row norm, scaled row -> w and
// LDS).  Phase 2: transposed, coalesced write of the 32 x D tile into wt[D, K].
template <typename T>
__global__ __launch_bounds__(256) void wn_fwd_kernel(const float* __restrict__ g,
                                                     const float* __restrict__ v,
                                                     T* __restrict__ w, T* __restrict__ wt,
                                                     float* __restrict__ vnorm, int K, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = (T*)smem_raw;  // [32][D + 2]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k0 = blockIdx.x * 32;
    const int ldt = D + 2;
    for (int r = 0; r < 8; ++r) {
        const int lr = wave * 8 + r, k = k0 + lr;
        if (k >= K) break;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float a = v[(int64_t)k * D + d];
            s += a * a;
        }
        const float nrm = sqrtf(wave_sum(s));
        const float sc = g[k] / nrm;
        for (int d = lane; d < D; d += 64) {
            const T o = from_f32<T>(v[(int64_t)k * D + d] * sc);
            w[(int64_t)k * D + d] = o;
            tile[lr * ldt + d] = o;
        }
        if (lane == 0) vnorm[k] = nrm;
    }
    __syncthreads();
    if (wt == nullptr) return;
    const int kk = threadIdx.x & 31;
    if (k0 + kk < K)
        for (int d = threadIdx.x >> 5; d < D; d += 8) wt[(int64_t)d * K + k0 + kk] = tile[kk * ldt + d];
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dW,
                                                     const float* __restrict__ g,
                                                     const float* __restrict__ v,
                                                     const float* __restrict__ vnorm,
                                                     float* __restrict__ dg, float* __restrict__ dv,
                                                     int K, int D) {
    const int lane = threadIdx.x & 63;
    for (int k = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); k < K; k += gridDim.x * 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += dW[(int64_t)k * D + d] * v[(int64_t)k * D + d];
        const float dot = wave_sum(s);
        const float inv = 1.f / vnorm[k];
        const float gk = g[k];
        if (dg && lane == 0) dg[k] += dot * inv;
        const float a = gk * inv, b = gk * dot * inv * inv * inv;
        for (int d = lane; d < D; d += 64)
            dv[(int64_t)k * D + d] += a * dW[(int64_t)k * D + d] - b * v[(int64_t)k * D + d];
    }
}

// ------------------------------------------------------------------ column sums
// block = 32 column-chunks (256 columns) x 8 row lanes; grid.y row slabs of `slab` rows.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ldx,
                                                     float* __restrict__ out, int64_t M, int N,
                                                     const float* __restrict__ rowscale, int rpg,
                                                     int slab) {
    __shared__ float red[8][32][8];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + tx) * 8;
    const int64_t r0 = (int64_t)blockIdx.y * slab;
    const int64_t r1 = r0 + slab < M ? r0 + slab : M;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        // four rows in flight per thread (one load at a time left the kernel latency-bound)
        for (int64_t r = r0 + ty; r < r1; r += 32) {
            float v[4][8], rs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + 8 * u;
                const bool ok = rr < r1;
                if (ok) load8_nt<T>(x + rr * ldx + col, v[u]);
                rs[u] = !ok ? 0.f : (rowscale ? rowscale[rr / rpg] : 1.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += rs[u] != 0.f ? v[u][e] * rs[u] : 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx][e] = acc[e];
    __syncthreads();
    // thread t sums column (t) of the block's 256 columns
    const int c = threadIdx.x;
    const int gc = blockIdx.x * 256 + c;
    if (gc < N) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += red[j][c >> 3][c & 7];
        atomicAdd(out + gc, s);
    }
}

// out[i, d] += sum_b x[(b*N + i), d];  grid (N, bsplit), block = D/8 threads (<= 256)
template <typename T>
__global__ void fold_rows_kernel(const T* __restrict__ x, float* __restrict__ out, int Bc, int N,
                                 int D) {
    __shared__ float sm[2048];
    const int i = blockIdx.x, ch = threadIdx.x;
    const bool live = ch * 8 < D;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // eight crops per round, every load of a round issued before the first add (one load in flight per wave ran
    // this kernel at 1.2 TB/s)
    for (int b0 = blockIdx.y; b0 < Bc; b0 += gridDim.y * 8) {
        float v[8][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u * gridDim.y;
            if (b < Bc && live) {
                load8<T>(x + ((int64_t)b * N + i) * D + ch * 8, v[u]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[u][e];
    }
    // the adds leave as whole 256-B pieces of the output row (lane = column): eight columns per lane made every
    // atomic instruction touch twelve 128-B lines with four lanes each, and the kernel ran at the atomics' rate
    if (live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[ch * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) atomicAdd(out + (int64_t)i * D + d, sm[d]);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ x, int64_t n8,
                                                    const float* __restrict__ scalar) {
    const float s = *scalar;
    if (s == 1.f) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= s;
        store8<T>(x + i * 8, v);
    }
}

int row_grid(int64_t rows) {
    int64_t b = (rows + 3) / 4;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

#define DISPATCH_T(dtype, ...)                 \
    if ((dtype) == DINO_BF16) { typedef bf16 T; __VA_ARGS__; } \
    else if ((dtype) == DINO_F32) { typedef float T; __VA_ARGS__; } \
    else return DINO_ERR_BADARG;

extern "C" int dino_layernorm_fwd(int dtype, const void* x, int64_t xs, void* y, int64_t ys,
                                  const float* gamma, const float* beta, float* mean, float* rstd,
                                  int64_t rows, int D, float eps, void* stream) {
    DINO_CHECK_ARG(x && y && gamma && beta && mean && rstd && rows > 0);
    DINO_CHECK_ARG(D % 8 == 0 && D > 0 && D <= 2048 && xs % 8 == 0 && ys % 8 == 0);
    if (D <= 512) {
        constexpr int R = 4;
        const int64_t g = (rows + 4 * R - 1) / (4 * R);
        const int grid = (int)(g > 2048 ? 2048 : g);
        DISPATCH_T(dtype, hipLaunchKernelGGL((ln_fwd_small_kernel<T, R>), dim3(grid), dim3(256), 0,
                                              (hipStream_t)stream, (const T*)x, xs, (T*)y, ys, gamma,
                                              beta, mean, rstd, rows, D, eps));
        return dino_launch_status();
    }
    if (D <= 1024) {
        constexpr int R = 2;
        const int64_t g = (rows + 4 * R - 1) / (4 * R);
        const int grid = (int)(g > 2048 ? 2048 : g);
        DISPATCH_T(dtype, hipLaunchKernelGGL((ln_fwd_wide_kernel<T, 2, R>), dim3(grid), dim3(256), 0,
                                              (hipStream_t)stream, (const T*)x, xs, (T*)y, ys, gamma,
                                              beta, mean, rstd, rows, D, eps));
        return dino_launch_status();
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(ln_fwd_kernel<T>, dim3(row_grid(rows)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, xs, (T*)y, ys, gamma,
                                          beta, mean, rstd, rows, D, eps));
    return dino_launch_status();
}

static int ln_bwd_launch(int dtype, const void* dy, int64_t dys, const void* x, int64_t xs,
                         const float* gamma, const float* mean, const float* rstd,
                         const void* dres, int64_t drs, void* dx, int64_t dxs,
                         float* dgamma, float* dbeta, float* dx_colsum,
                         const float* rowscale, int rows_per_group, int64_t rows, int D,
                         void* workspace, int64_t workspace_bytes, void* stream,
                         void* xn = nullptr, int64_t xns = 0, const float* beta = nullptr) {
    DINO_CHECK_ARG(dy && x && gamma && mean && rstd && dx && rows > 0);
    if (xn) DINO_CHECK_ARG(beta && xns % 8 == 0 && xns >= D);
    DINO_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr));
    DINO_CHECK_ARG(D % 8 == 0 && D > 0 && D <= 2048 && xs % 8 == 0 && dys % 8 == 0 && dxs % 8 == 0 &&
                   drs % 8 == 0);
    if (rowscale) DINO_CHECK_ARG(rows_per_group > 0);
    const int R = D <= 512 ? (xn ? LNB_R : 2) : 1;
    int64_t g = (rows + 4 * R - 1) / (4 * R);
    const int cap = D <= 512 ? (xn ? 256 * LNB_OCC : 1024) : (D <= 1024 ? 768 : 256);  // resident workgroups (VGPR-limited)
    const int grid = (int)(g < 1 ? 1 : (g > cap ? cap : g));
    const int smem = 13 * D * (int)sizeof(float);   // reduction scratch + a copy of beta
    if (smem > 64 * 1024) return DINO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // with a workspace the column sums leave the kernel as per-workgroup partials
    float* part = nullptr;
    if ((dgamma || dx_colsum) && workspace && workspace_bytes >= (int64_t)grid * 3 * D * 4 && grid > 16)
        part = (float*)workspace;
#define LN_BWD_LAUNCH(TT, NC, RR)                                                                   \
    if (xn)                                                                                         \
        hipLaunchKernelGGL((ln_bwd_kernel<TT, NC, RR, true>), dim3(grid), dim3(256), smem, st, (const TT*)dy, dys, \
                       (const TT*)x, xs, gamma, mean, rstd, (const TT*)dres, drs, (TT*)dx, dxs,    \
                       dgamma, dbeta, dx_colsum, rowscale, rows_per_group, rows, D, part, (TT*)xn, xns, beta); \
    else                                                                                            \
        hipLaunchKernelGGL((ln_bwd_kernel<TT, NC, RR, false>), dim3(grid), dim3(256), smem, st, (const TT*)dy, dys, \
                       (const TT*)x, xs, gamma, mean, rstd, (const TT*)dres, drs, (TT*)dx, dxs,    \
                       dgamma, dbeta, dx_colsum, rowscale, rows_per_group, rows, D, part, (TT*)nullptr, 0, nullptr)
#define LN_BWD_NC(TT)                                  \
    if (D <= 512 && xn) { LN_BWD_LAUNCH(TT, 1, LNB_R); } \
    else if (D <= 512) { LN_BWD_LAUNCH(TT, 1, 2); }    \
    else if (D <= 1024) { LN_BWD_LAUNCH(TT, 2, 1); }   \
    else { LN_BWD_LAUNCH(TT, 4, 1); }
    if (dtype == DINO_BF16) { LN_BWD_NC(bf16) }
    else if (dtype == DINO_F32) { LN_BWD_NC(float) }
    else return DINO_ERR_BADARG;
    if (part)
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((3 * D + 255) / 256, 64), dim3(256), 0, st, part, grid, D,
                           dgamma, dbeta, dx_colsum);
    return dino_launch_status();
}

extern "C" int dino_layernorm_bwd(int dtype, const void* dy, int64_t dys, const void* x, int64_t xs,
                                  const float* gamma, const float* mean, const float* rstd,
                                  const void* dres, int64_t drs, void* dx, int64_t dxs,
                                  float* dgamma, float* dbeta, float* dx_colsum,
                                  const float* rowscale, int rows_per_group, int64_t rows, int D,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
    return ln_bwd_launch(dtype, dy, dys, x, xs, gamma, mean, rstd, dres, drs, dx, dxs, dgamma, dbeta,
                         dx_colsum, rowscale, rows_per_group, rows, D, workspace, workspace_bytes, stream);
}

extern "C" int dino_layernorm_bwd_xn(int dtype, const void* dy, int64_t dys, const void* x, int64_t xs,
                                     const float* gamma, const float* beta, const float* mean, const float* rstd,
                                     const void* dres, int64_t drs, void* dx, int64_t dxs, void* xn, int64_t xns,
                                     float* dgamma, float* dbeta, float* dx_colsum, const float* rowscale,
                                     int rows_per_group, int64_t rows, int D, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    DINO_CHECK_ARG(xn && beta);
    return ln_bwd_launch(dtype, dy, dys, x, xs, gamma, mean, rstd, dres, drs, dx, dxs, dgamma, dbeta,
                         dx_colsum, rowscale, rows_per_group, rows, D, workspace, workspace_bytes, stream, xn, xns,
                         beta);
}

// ---- LayerNorm folded into the GEMMs around it: statistics and folded weights ----------------------------
namespace {
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ stats, int nparts,
                                                                int64_t rows, float invD, float eps,
                                                                float* __restrict__ mean, float* __restrict__ rstd) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= rows) return;
    float s = 0.f, q = 0.f;
    for (int p = 0; p < nparts; ++p) {
        const float2 v = *(const float2*)(stats + ((int64_t)p * rows + m) * 2);
        s += v.x;
        q += v.y;
    }
    const float mu = s * invD;
    const float var = fmaxf(fmaf(-mu, mu, q * invD), 0.f);
    mean[m] = mu;
    rstd[m] = rsqrtf(var + eps);
}

// one wave per row, the row in registers (D <= 2048), two-pass like ln_fwd_kernel
template <typename T>
__global__ __launch_bounds__(256) void ln_stats_kernel(const T* __restrict__ x, int64_t xs, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int nch = D >> 3;
    const float invD = 1.f / (float)D;
    for (int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        float v[4][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                load8<T>(x + row * xs + ch * 8, v[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[c][e];
            }
        }
        const float mu = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (lane + 64 * c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[c][e] - mu;
                    q += d * d;
                }
            }
        }
        const float r = rsqrtf(wave_sum(q) * invD + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = r;
        }
    }
}

// wave per weight row n: wfold[n,:] = bf16(gamma * W[n,:] - mean_k(gamma * W[n,:])), bfold[n] = bias[n] + beta . W[n,:]
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const float* __restrict__ master,
                                                              const int64_t* __restrict__ table,
                                                              bf16* __restrict__ wfold, float* __restrict__ bfold) {
    const int64_t* t = table + (int64_t)blockIdx.y * 8;
    const int N = (int)t[6], K = (int)t[7];
    const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const float* W = master + t[0] + (int64_t)n * K;
    const float* gamma = master + t[1];
    const float* beta = master + t[2];
    float sg = 0.f, sb = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 w = *(const f32x4*)(W + k), g = *(const f32x4*)(gamma + k), b = *(const f32x4*)(beta + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sg = fmaf(g[e], w[e], sg);
            sb = fmaf(b[e], w[e], sb);
        }
    }
    sg = wave_sum(sg) / (float)K;
    sb = wave_sum(sb);
    bf16* dst = wfold + t[4] + (int64_t)n * K;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 w = *(const f32x4*)(W + k), g = *(const f32x4*)(gamma + k);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(fmaf(g[e], w[e], -sg));
        *(bf16x4*)(dst + k) = o;
    }
    if (lane == 0) bfold[t[5] + n] = master[t[3] + n] + sb;
}
}  // namespace

extern "C" int dino_ln_stats_finalize(const float* stats, int nparts, int64_t rows, int D, float eps, float* mean,
                                      float* rstd, void* stream) {
    DINO_CHECK_ARG(stats && mean && rstd && nparts > 0 && rows > 0 && D > 0);
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, stats, nparts, rows, 1.f / (float)D, eps, mean, rstd);
    return dino_launch_status();
}

extern "C" int dino_layernorm_stats(int dtype, const void* x, int64_t xs, float* mean, float* rstd, int64_t rows, int D,
                                    float eps, void* stream) {
    DINO_CHECK_ARG(x && mean && rstd && rows > 0 && D % 8 == 0 && D > 0 && D <= 2048 && xs % 8 == 0);
    DISPATCH_T(dtype, hipLaunchKernelGGL(ln_stats_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, (hipStream_t)stream,
                                          (const T*)x, xs, mean, rstd, rows, D, eps));
    return dino_launch_status();
}

extern "C" int dino_ln_fold_weights(const float* master, const int64_t* table, int count, void* wfold, float* bfold,
                                    int maxN, void* stream) {
    DINO_CHECK_ARG(master && table && wfold && bfold && count > 0 && maxN > 0);
    hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((maxN + 3) / 4, count), dim3(256), 0, (hipStream_t)stream, master,
                       table, (bf16*)wfold, bfold);
    return dino_launch_status();
}

// ---- the allocation-free variant: the per-workgroup partials live in a device array of the library
// (1024 workgroups x 3 column-sum kinds x DINO_LN_STATIC_MAX_D floats = 12 MiB per GPU, held until
// the process ends).  `ln_scratch` below is the HOST-side handle of that array: the device address
// comes from hipGetSymbolAddress, once per device, and only that pointer is ever handed to a kernel.
static __device__ __align__(256) float ln_scratch[1024 * 3 * DINO_LN_STATIC_MAX_D];
static float* ln_scratch_ptr[64];          // per device; written under ln_scratch_mu
static std::mutex ln_scratch_mu;

static int ln_scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return DINO_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(ln_scratch_mu);
    if (ln_scratch_ptr[dev] == nullptr) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(ln_scratch));   // this device's copy
        if (e != hipSuccess) return (int)e;
        if (p == nullptr) return DINO_ERR_UNSUPPORTED;
        ln_scratch_ptr[dev] = p;
    }
    *out = ln_scratch_ptr[dev];
    return DINO_OK;
}

extern "C" int dino_layernorm_bwd_static(int dtype, const void* dy, int64_t dys, const void* x,
                                         int64_t xs, const float* gamma, const float* mean,
                                         const float* rstd, const void* dres, int64_t drs, void* dx,
                                         int64_t dxs, float* dgamma, float* dbeta, float* dx_colsum,
                                         const float* rowscale, int rows_per_group, int64_t rows,
                                         int D, void* stream) {
    if (D > DINO_LN_STATIC_MAX_D) return DINO_ERR_TOO_LARGE;   // the static array is sized for D <= 1024
    float* scratch = nullptr;
    const int rc = ln_scratch_for_current_device(&scratch);
    if (rc != DINO_OK) return rc;
    // every workgroup overwrites its own 3*D partials before the reduce kernel reads them: nothing in
    // the array has to start at zero, whatever the previous call left there
    return ln_bwd_launch(dtype, dy, dys, x, xs, gamma, mean, rstd, dres, drs, dx, dxs, dgamma, dbeta,
                         dx_colsum, rowscale, rows_per_group, rows, D, scratch,
                         (int64_t)sizeof(float) * 1024 * 3 * DINO_LN_STATIC_MAX_D, stream);
}

extern "C" int dino_l2norm_fwd(int dtype, const void* x, void* y, float* norm, int64_t rows, int D,
                               void* stream) {
    DINO_CHECK_ARG(x && y && norm && rows > 0 && D % 8 == 0 && D > 0 && D <= 2048);
    DISPATCH_T(dtype, hipLaunchKernelGGL(l2n_fwd_kernel<T>, dim3(row_grid(rows)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, (T*)y, norm, rows, D));
    return dino_launch_status();
}

extern "C" int dino_l2norm_bwd(int dtype, const float* dy, const void* y, const float* norm,
                               void* dx, int64_t rows, int D, void* stream) {
    DINO_CHECK_ARG(dy && y && norm && dx && rows > 0 && D % 8 == 0 && D > 0 && D <= 2048);
    DISPATCH_T(dtype, hipLaunchKernelGGL(l2n_bwd_kernel<T>, dim3(row_grid(rows)), dim3(256), 0,
                                          (hipStream_t)stream, dy, (const T*)y, norm, (T*)dx, rows, D));
    return dino_launch_status();
}

extern "C" int dino_weightnorm_fwd(int dtype, const float* g, const float* v, void* w, void* wt,
                                   float* vnorm, int K, int D, void* stream) {
    DINO_CHECK_ARG(g && v && w && vnorm && K > 0 && D > 0 && D <= 4096);
    DISPATCH_T(dtype, hipLaunchKernelGGL(wn_fwd_kernel<T>, dim3((K + 31) / 32), dim3(256),
                                          32 * (D + 2) * sizeof(T), (hipStream_t)stream, g, v,
                                          (T*)w, (T*)wt, vnorm, K, D));
    return dino_launch_status();
}

extern "C" int dino_weightnorm_bwd(const float* dW, const float* g, const float* v,
                                   const float* vnorm, float* dg, float* dv, int K, int D,
                                   void* stream) {
    DINO_CHECK_ARG(dW && g && v && vnorm && dv && K > 0 && D > 0);
    int grid = (K + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(wn_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dW, g, v, vnorm,
                       dg, dv, K, D);
    return dino_launch_status();
}

extern "C" int dino_colsum(int dtype, const void* x, int64_t ldx, float* out, int64_t M, int N,
                           const float* rowscale, int rows_per_group, void* stream) {
    DINO_CHECK_ARG(x && out && M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0);
    if (rowscale) DINO_CHECK_ARG(rows_per_group > 0);
    const int gx = (N + 255) / 256;
    int slab = 512;
    int64_t gy = (M + slab - 1) / slab;
    if (gy > 1024) {
        gy = 1024;
        slab = (int)((M + gy - 1) / gy);
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_kernel<T>, dim3(gx, (unsigned)gy), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, ldx, out, M, N, rowscale,
                                          rows_per_group, slab));
    return dino_launch_status();
}

extern "C" int dino_fold_rows(int dtype, const void* x, float* out, int Bc, int N, int D,
                              void* stream) {
    DINO_CHECK_ARG(x && out && Bc > 0 && N > 0 && D > 0 && D % 8 == 0 && D <= 2048);
    const int bsplit = Bc < 32 ? (Bc < 16 ? Bc : 16) : (Bc / 16 < 64 ? Bc / 16 : 64);   // >= 16 crops per workgroup
    DISPATCH_T(dtype, hipLaunchKernelGGL(fold_rows_kernel<T>, dim3(N, bsplit), dim3((D / 8 + 63) / 64 * 64),
                                          0, (hipStream_t)stream, (const T*)x, out, Bc, N, D));
    return dino_launch_status();
}

extern "C" int dino_scale_by_scalar(int dtype, void* x, int64_t n, const float* scalar, void* stream) {
    DINO_CHECK_ARG(x && scalar && n > 0 && n % 8 == 0);
    int64_t b = (n / 8 + 255) / 256;
    if (b > 4096) b = 4096;
    DISPATCH_T(dtype, hipLaunchKernelGGL(scale_kernel<T>, dim3((unsigned)b), dim3(256), 0,
                                          (hipStream_t)stream, (T*)x, n / 8, scalar));
    return dino_launch_status();
}
