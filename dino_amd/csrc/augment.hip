// Multi-crop augmentation on the device: the image operations behind
// dino/data/augmentations.py:75-157 -- RandomResizedCrop (bicubic), horizontal flip, ColorJitter
// (brightness / contrast / saturation / hue), RandomGrayscale, GaussianBlur, Solarization -- on uint8
// [B, H, W, 3] source images that already sit in HBM.  The reference runs them as torchvision -> Pillow
// calls on PIL images in DataLoader workers; every kernel here follows the Pillow C routine the call
// lands in, operation by operation (fixed-point taps, uint8 intermediates, float / double promotion of
// each expression), so the results are bit-identical to Pillow's (oracle/augment_oracle.py, pinned against
// Pillow 12.2 itself).  The random draws stay on the host (dino_amd/data/augment.py) and arrive as one
// int32 parameter row per crop; nothing here is random.
//
// All of it is HBM-bound byte work: one thread per output pixel, crops along blockIdx.y so that every
// per-crop parameter is wave-uniform (scalar loads).
#include "common.h"

// Pillow is built without FMA contraction; a fused multiply-add would change the last bit of the taps
// and of Image.blend.
#pragma clang fp contract(off)

namespace {

constexpr int AP = DINO_AUG_PARAM_WORDS;
constexpr int PRECISION_BITS = 32 - 8 - 2;   // Resample.c

// parameter row (int32 words): see include/dino_hip.h
enum { P_SRC = 0, P_TOP, P_LEFT, P_H, P_W, P_FLIP, P_OP0 = 6, P_FAC0 = 12, P_BLUR = 18, P_RADIUS, P_WW, P_FW };

__device__ double bicubic_filter(double x) {   // Resample.c, a = -0.5
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for both axes of every crop: thread = (crop, axis,
// output index).  bounds[(crop*2 + axis)*S + xx] = {first input index, tap count}; taps in 10.22 fixed point.
__global__ __launch_bounds__(256) void resize_coeffs_kernel(const int* __restrict__ params, int ncrops, int S,
                                                            int kmax, int2* __restrict__ bounds,
                                                            int* __restrict__ coeffs) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= ncrops * 2 * S) return;
    const int xx = idx % S, axis = (idx / S) & 1, c = idx / (2 * S);
    const int in_size = params[c * AP + (axis ? P_H : P_W)];
    const double scale = (double)in_size / (double)S;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > kmax) xmax = kmax;   // (the host sized kmax from the largest box: never taken)
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
    int* k = coeffs + (int64_t)idx * kmax;
    for (int x = 0; x < kmax; ++x) {
        double w = 0.0;
        if (x < xmax) {
            w = bicubic_filter((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
        }
        k[x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
    }
    bounds[idx] = make_int2(xmin, xmax);
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Horizontal pass of ImagingResample on the cropped box: tmp[crop][y][xx] for the box's rows y < h.
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, int Hs, int Ws,
                                                       const int* __restrict__ params, int S, int kmax,
                                                       const int2* __restrict__ bounds,
                                                       const int* __restrict__ coeffs, uint8_t* __restrict__ tmp,
                                                       int tmp_rows) {
    const int c = blockIdx.y;
    const int* P = params + c * AP;
    const int h = P[P_H];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * S) return;
    const int y = idx / S, xx = idx - y * S;
    const int2 bd = bounds[(c * 2 + 0) * S + xx];
    const int* k = coeffs + (int64_t)((c * 2 + 0) * S + xx) * kmax;
    const uint8_t* row = src + (((int64_t)P[P_SRC] * Hs + P[P_TOP] + y) * Ws + P[P_LEFT] + bd.x) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < bd.y; ++x) {
        const int kk = k[x];
        a0 += row[3 * x] * kk;
        a1 += row[3 * x + 1] * kk;
        a2 += row[3 * x + 2] * kk;
    }
    uint8_t* o = tmp + (((int64_t)c * tmp_rows + y) * S + xx) * 3;
    o[0] = clip8(a0);
    o[1] = clip8(a1);
    o[2] = clip8(a2);
}

// Vertical pass; the flip (img.transpose(FLIP_LEFT_RIGHT) after the resize) is the output column.
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ tmp, int tmp_rows,
                                                       const int* __restrict__ params, int S, int kmax,
                                                       const int2* __restrict__ bounds,
                                                       const int* __restrict__ coeffs, uint8_t* __restrict__ out) {
    const int c = blockIdx.y;
    const int* P = params + c * AP;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    const int yy = idx / S, xx = idx - yy * S;
    const int2 bd = bounds[(c * 2 + 1) * S + yy];
    const int* k = coeffs + (int64_t)((c * 2 + 1) * S + yy) * kmax;
    const uint8_t* col = tmp + (((int64_t)c * tmp_rows + bd.x) * S + xx) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < bd.y; ++y) {
        const int kk = k[y];
        const uint8_t* p = col + (int64_t)y * S * 3;
        a0 += p[0] * kk;
        a1 += p[1] * kk;
        a2 += p[2] * kk;
    }
    const int ox = P[P_FLIP] ? S - 1 - xx : xx;
    uint8_t* o = out + (((int64_t)c * S + yy) * S + ox) * 3;
    o[0] = clip8(a0);
    o[1] = clip8(a1);
    o[2] = clip8(a2);
}

__device__ __forceinline__ int luma(int r, int g, int b) {   // Convert.c rgb2l (L24 / 65536, rounded)
    return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16;
}

// sums[crop] += sum of convert("L") over the crop (ImageStat.Stat(...).sum of ImageEnhance.Contrast)
__global__ __launch_bounds__(256) void gray_sums_kernel(const uint8_t* __restrict__ img, int npix,
                                                        unsigned* __restrict__ sums,
                                                        const int* __restrict__ params, int slot) {
    // only the crops whose op in this slot is contrast read their sum
    if (params && params[blockIdx.y * AP + P_OP0 + slot] != 2) return;
    __shared__ unsigned part;
    if (threadIdx.x == 0) part = 0;
    __syncthreads();
    const uint8_t* p = img + (int64_t)blockIdx.y * npix * 3;
    unsigned s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256)
        s += (unsigned)luma(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    atomicAdd(&part, s);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + blockIdx.y, part);
}

// Blend.c ImagingBlend(degenerate, image, alpha): single precision, truncation; clipping only when
// alpha is outside [0, 1].
__device__ __forceinline__ int blend1(int deg, int px, float alpha) {
    const float t = (float)deg + alpha * (float)(px - deg);
    if (alpha >= 0.f && alpha <= 1.0f) return (int)t & 255;
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
    return (int)t;
}

// Convert.c rgb2hsv_row / hsv2rgb: float variables, double where a double constant promotes the expression
__device__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) {
        uh = 0;
        us = 0;
        return;
    }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    const double hd = (double)h / 6.0 + 1.0;
    h = (float)(hd - floor(hd));   // fmod(hd, 1.0), hd > 0
    const int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
    uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
    us = is < 0 ? 0 : (is > 255 ? 255 : is);
}

__device__ __forceinline__ int round8(double x) {   // CLIP8(round(x)), x >= 0
    const int v = (int)floor(x + 0.5);
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__device__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) {
        r = g = b = v;
        return;
    }
    const double hd = (double)(float)h * 6.0 / 255.0;
    const double fi = floor(hd);
    const int i = (int)fi;
    const float f = (float)(hd - fi);
    const float fs = (float)((double)(float)s / 255.0);
    const double vf = (double)(float)v;
    const int p = round8(vf * (1.0 - (double)fs));
    const int q = round8(vf * (1.0 - (double)(fs * f)));
    const int t = round8(vf * (1.0 - (double)fs * (1.0 - (double)f)));
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// One slot of the per-crop operation list, in place.  op: 0 none, 1 brightness, 2 contrast, 3 saturation
// (factor = float), 4 hue (factor word = the uint8 shift), 5 grey-scale, 6 solarize.
__global__ __launch_bounds__(256) void color_op_kernel(uint8_t* __restrict__ img, int npix,
                                                       const int* __restrict__ params, int slot,
                                                       const unsigned* __restrict__ sums) {
    const int c = blockIdx.y;
    const int op = params[c * AP + P_OP0 + slot];
    if (op == 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int fbits = params[c * AP + P_FAC0 + slot];
    const float fac = __builtin_bit_cast(float, fbits);
    uint8_t* p = img + ((int64_t)c * npix + i) * 3;
    int r = p[0], g = p[1], b = p[2];
    if (op == 1) {
        r = blend1(0, r, fac);
        g = blend1(0, g, fac);
        b = blend1(0, b, fac);
    } else if (op == 2) {
        // int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5): a Python float (double) division
        const int mean = (int)((double)sums[c] / (double)npix + 0.5);
        r = blend1(mean, r, fac);
        g = blend1(mean, g, fac);
        b = blend1(mean, b, fac);
    } else if (op == 3) {
        const int l = luma(r, g, b);
        r = blend1(l, r, fac);
        g = blend1(l, g, fac);
        b = blend1(l, b, fac);
    } else if (op == 4) {
        int h, s, v;
        rgb2hsv(r, g, b, h, s, v);
        h = (h + fbits) & 255;   // numpy uint8 wrap-around add (torchvision _functional_pil.adjust_hue)
        hsv2rgb(h, s, v, r, g, b);
    } else if (op == 5) {
        r = g = b = luma(r, g, b);
    } else if (op == 6) {
        r = r < 128 ? r : 255 - r;
        g = g < 128 ? g : 255 - g;
        b = b < 128 ? b : 255 - b;
    }
    p[0] = (uint8_t)r;
    p[1] = (uint8_t)g;
    p[2] = (uint8_t)b;
}

// One pass of BoxBlur.c ImagingHorizontalBoxBlur along x (axis 0) or y (axis 1), edge pixels replicated:
// out = (ww * sum_{|j| <= r} in[x + j] + fw * (in[x - r - 1] + in[x + r + 1]) + 2^23) >> 24  (UINT32).
__global__ __launch_bounds__(256) void box_blur_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       int S, const int* __restrict__ params, int axis) {
    const int c = blockIdx.y;
    const int* P = params + c * AP;
    if (!P[P_BLUR]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const int y = i / S, x = i - y * S;
    const int radius = P[P_RADIUS];
    const unsigned ww = (unsigned)P[P_WW], fw = (unsigned)P[P_FW];
    const uint8_t* base = src + (int64_t)c * S * S * 3;
    const int pos = axis ? y : x, stride = axis ? S * 3 : 3;
    const uint8_t* line = base + (axis ? x * 3 : y * S * 3);
    unsigned a0 = 0, a1 = 0, a2 = 0;
    for (int j = -radius; j <= radius; ++j) {
        int q = pos + j;
        q = q < 0 ? 0 : (q > S - 1 ? S - 1 : q);
        const uint8_t* p = line + q * stride;
        a0 += p[0];
        a1 += p[1];
        a2 += p[2];
    }
    int ql = pos - radius - 1, qr = pos + radius + 1;
    ql = ql < 0 ? 0 : ql;
    qr = qr > S - 1 ? S - 1 : qr;
    const uint8_t *pl = line + ql * stride, *prr = line + qr * stride;
    uint8_t* o = dst + ((int64_t)c * S * S + i) * 3;
    o[0] = (uint8_t)((a0 * ww + ((unsigned)pl[0] + prr[0]) * fw + (1u << 23)) >> 24);
    o[1] = (uint8_t)((a1 * ww + ((unsigned)pl[1] + prr[1]) * fw + (1u << 23)) >> 24);
    o[2] = (uint8_t)((a2 * ww + ((unsigned)pl[2] + prr[2]) * fw + (1u << 23)) >> 24);
}

int cdiv(int64_t a, int b) { return (int)((a + b - 1) / b); }

}  // namespace

extern "C" int dino_aug_resized_crop(const uint8_t* src, int B, int Hs, int Ws, const int32_t* params, int ncrops,
                                     int S, int kmax, int32_t* bounds, int32_t* coeffs, uint8_t* tmp, int tmp_rows,
                                     uint8_t* out, void* stream) {
    DINO_CHECK_ARG(src && params && bounds && coeffs && tmp && out);
    DINO_CHECK_ARG(B > 0 && Hs > 0 && Ws > 0 && ncrops > 0 && ncrops <= 65535 && S > 0 && S <= 4096);
    DINO_CHECK_ARG(kmax > 0 && kmax <= 256 && tmp_rows > 0 && tmp_rows <= Hs);
    DINO_CHECK_ARG((int64_t)B * Hs * Ws * 3 < (1ll << 40) && ((uintptr_t)bounds & 7) == 0);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(cdiv((int64_t)ncrops * 2 * S, 256)), dim3(256), 0, st, params,
                       ncrops, S, kmax, (int2*)bounds, coeffs);
    hipLaunchKernelGGL(resize_h_kernel, dim3(cdiv((int64_t)tmp_rows * S, 256), ncrops), dim3(256), 0, st, src, Hs, Ws,
                       params, S, kmax, (const int2*)bounds, coeffs, tmp, tmp_rows);
    hipLaunchKernelGGL(resize_v_kernel, dim3(cdiv((int64_t)S * S, 256), ncrops), dim3(256), 0, st, tmp, tmp_rows,
                       params, S, kmax, (const int2*)bounds, coeffs, out);
    return dino_launch_status();
}

extern "C" int dino_aug_gray_sums(const uint8_t* img, int ncrops, int S, uint32_t* sums, const int32_t* params,
                                  int slot, void* stream) {
    DINO_CHECK_ARG(img && sums && ncrops > 0 && ncrops <= 65535 && S > 0 && S <= 4096);
    DINO_CHECK_ARG(!params || (slot >= 0 && slot < DINO_AUG_SLOTS));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)ncrops * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    const int npix = S * S;
    int gx = cdiv(npix, 256 * 8);
    hipLaunchKernelGGL(gray_sums_kernel, dim3(gx, ncrops), dim3(256), 0, st, img, npix, sums, params, slot);
    return dino_launch_status();
}

extern "C" int dino_aug_color_op(uint8_t* img, int ncrops, int S, const int32_t* params, int slot,
                                 const uint32_t* sums, void* stream) {
    DINO_CHECK_ARG(img && params && ncrops > 0 && ncrops <= 65535 && S > 0 && S <= 4096);
    DINO_CHECK_ARG(slot >= 0 && slot < DINO_AUG_SLOTS && sums);
    hipLaunchKernelGGL(color_op_kernel, dim3(cdiv((int64_t)S * S, 256), ncrops), dim3(256), 0, (hipStream_t)stream, img,
                       S * S, params, slot, sums);
    return dino_launch_status();
}

extern "C" int dino_aug_box_blur(const uint8_t* src, uint8_t* dst, int ncrops, int S, const int32_t* params,
                                 int axis, void* stream) {
    DINO_CHECK_ARG(src && dst && src != dst && params && ncrops > 0 && ncrops <= 65535 && S > 0 && S <= 4096);
    DINO_CHECK_ARG(axis == 0 || axis == 1);
    hipLaunchKernelGGL(box_blur_kernel, dim3(cdiv((int64_t)S * S, 256), ncrops), dim3(256), 0, (hipStream_t)stream, src,
                       dst, S, params, axis);
    return dino_launch_status();
}
