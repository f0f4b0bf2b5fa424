// MV-guided bilinear alignment (the reference's VOSAlignment = flow_warp:
// mmedit/models/backbones/sr_backbones/iconvsr_mv.py:12-18,
// mmedit/models/common/flow_warp.py:6-50 -> F.grid_sample(bilinear, zeros, align_corners=True)).
//
// HBM-bound gather.  Algorithmic bytes per output pixel (C = 64, fp32):
//   8 (flow) + 256 (each source byte once) + 256 (write) = 520 B.
//
// Pixel-major layout: one pixel's 64 channels are 256 contiguous bytes, so every bilinear
// tap is one fully-used 256-B segment; 16 lanes x float4 cover a pixel, a wave covers 4
// pixels per instruction, and since codec MVs are constant over >= 8x8 blocks neighbouring
// pixels' taps are neighbouring segments (L2/TA friendly).  No LDS is needed: there is no
// reuse beyond what a 4-tap footprint shares through L1/L2.
#include "warp.h"
#include "f16_util.h"
#ifndef WARP_RPT
#define WARP_RPT 2
#endif

namespace {

// Coordinate arithmetic kept in the reference's order (normalise to [-1,1], flow_warp.py:41-42,
// then ATen's align_corners=True un-normalise) so results track the CPU path to ~1e-7.
__device__ __forceinline__ void sample_coords(float x, float y, float fx, float fy, int H, int W, float& ix, float& iy) {
    const float px = x + fx, py = y + fy;
    const float wm1 = (float)(W - 1 > 1 ? W - 1 : 1), hm1 = (float)(H - 1 > 1 ? H - 1 : 1);
    const float nx = 2.0f * px / wm1 - 1.0f, ny = 2.0f * py / hm1 - 1.0f;
    ix = ((nx + 1.0f) / 2.0f) * (float)(W - 1);
    iy = ((ny + 1.0f) / 2.0f) * (float)(H - 1);
    // keep the int conversion defined for wild vectors; anything beyond [-1, size] has no valid tap
    ix = fminf(fmaxf(ix, -2.0f), (float)W + 1.0f);
    iy = fminf(fmaxf(iy, -2.0f), (float)H + 1.0f);
}

__device__ __forceinline__ void tap_setup(float x, float y, float fx, float fy, int H, int W,
                                          int& x0, int& y0, float& wx1, float& wy1) {
    float ix, iy;
    sample_coords(x, y, fx, fy, H, W, ix, iy);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    x0 = (int)fx0;
    y0 = (int)fy0;
    wx1 = ix - fx0;
    wy1 = iy - fy0;
}

// feat/out: [H][W][C4*4] ; fxp/fyp: [H][W] planes (two channel planes of the NCHW mvs tensor)
// OUT16: the aligned map is written as fp16 (8 B per lane).  On the fp16-operand conv path it is read only as an MFMA A
// operand of the input conv, which would round it (saturating, round-to-nearest-even) on its way into LDS: rounding it
// here instead is bit-identical and halves the bytes written and re-read.
// NEAREST (flow_inter='nearest', flow_warp.py:47 -> ATen grid_sampler_2d Nearest): the pixel at nearbyint of the un-normalised
// coordinate (ties to even = rintf in the default rounding mode) if it lies in the image, else 0 -- one tap, 264 + 256 B per pixel.
template <bool OUT16, bool NEAREST = false>
__global__ __launch_bounds__(256) void mv_warp_nhwc_kernel(const float* __restrict__ feat,
                                                           const float* __restrict__ fxp,
                                                           const float* __restrict__ fyp,
                                                           void* __restrict__ out, int H, int W, int C4,
                                                           long total) {
    const f32x4* f4 = reinterpret_cast<const f32x4*>(feat);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    h4* o16 = reinterpret_cast<h4*>(out);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C4;
        const int c4 = (int)(i - pix * C4);
        const int y = (int)(pix / W), x = (int)(pix - (long)y * W);
        if (NEAREST) {
            float ix, iy;
            sample_coords((float)x, (float)y, fxp[pix], fyp[pix], H, W, ix, iy);
            const int xn = (int)rintf(ix), yn = (int)rintf(iy);
            const bool ok = (xn >= 0) & (xn < W) & (yn >= 0) & (yn < H);
            f32x4 v = ok ? f4[((long)yn * W + xn) * C4 + c4] : (f32x4)(0.f);
            if (OUT16) {
                v = __builtin_elementwise_min(__builtin_elementwise_max(v, (f32x4)(-65504.f)), (f32x4)(65504.f));
                o16[i] = __builtin_convertvector(v, h4);
            } else {
                o4[i] = v;
            }
            continue;
        }
        int x0, y0;
        float wx1, wy1;
        tap_setup((float)x, (float)y, fxp[pix], fyp[pix], H, W, x0, y0, wx1, wy1);
        const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;   // == (x0+1) - ix
        const bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x0 + 1 >= 0) & (x0 + 1 < W);
        const bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y0 + 1 >= 0) & (y0 + 1 < H);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const long r0 = ((long)y0 * W + x0) * C4 + c4, r1 = r0 + (long)W * C4;
        const f32x4 v00 = (vx0 & vy0) ? f4[r0] : z;
        const f32x4 v01 = (vx1 & vy0) ? f4[r0 + C4] : z;
        const f32x4 v10 = (vx0 & vy1) ? f4[r1] : z;
        const f32x4 v11 = (vx1 & vy1) ? f4[r1 + C4] : z;
        f32x4 v = v00 * (wx0 * wy0) + v01 * (wx1 * wy0) + v10 * (wx0 * wy1) + v11 * (wx1 * wy1);
        if (OUT16) {
            v = __builtin_elementwise_min(__builtin_elementwise_max(v, (f32x4)(-65504.f)), (f32x4)(65504.f));
            o16[i] = __builtin_convertvector(v, h4);
        } else {
            o4[i] = v;
        }
    }
}

// The 64-channel case (every call of the generator): 16 lanes per pixel, RPT pixel rows per thread, no index division (grid.x walks a
// row in groups of 16 pixels, grid.y the row groups) and every tap through a buffer descriptor -- a tap outside the image gets the
// offset OOB and loads zeros, so the 4 x RPT loads of a thread are in flight together with no exec-mask juggling in between.
// r04: 97 us -> see DESIGN.md 3.2 (the grid-stride kernel above spent two 64-bit divisions per float4 and issued its taps under
// four divergent branches).
template <bool OUT16, bool NEAREST, int RPT>
__global__ __launch_bounds__(256) void mv_warp_nhwc64_kernel(const float* __restrict__ feat, const float* __restrict__ fxp,
                                                             const float* __restrict__ fyp, void* __restrict__ out, int H, int W) {
    const int c4 = threadIdx.x & 15, x = blockIdx.x * 16 + (threadIdx.x >> 4);
    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(feat, map_bytes);
    const __amdgpu_buffer_rsrc_t r_fx = make_rsrc(fxp, (unsigned)H * (unsigned)W * 4u), r_fy = make_rsrc(fyp, (unsigned)H * (unsigned)W * 4u);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(out, OUT16 ? map_bytes / 2 : map_bytes);
    const bool col_ok = x < W;
    float fx[RPT], fy[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int y = blockIdx.y * RPT + r;
        const unsigned po = (col_ok & (y < H)) ? ((unsigned)y * (unsigned)W + (unsigned)x) * 4u : OOB;
        fx[r] = buf_load1(r_fx, po);
        fy[r] = buf_load1(r_fy, po);
    }
    f32x4 v[RPT][4];
    float wt[RPT][4];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int y = blockIdx.y * RPT + r;
        if (NEAREST) {
            float ix, iy;
            sample_coords((float)x, (float)y, fx[r], fy[r], H, W, ix, iy);
            const int xn = (int)rintf(ix), yn = (int)rintf(iy);
            const bool ok = (xn >= 0) & (xn < W) & (yn >= 0) & (yn < H);
            v[r][0] = buf_load4(r_in, ok ? ((unsigned)yn * (unsigned)W + (unsigned)xn) * 256u + (unsigned)c4 * 16u : OOB);
        } else {
            int x0, y0;
            float wx1, wy1;
            tap_setup((float)x, (float)y, fx[r], fy[r], H, W, x0, y0, wx1, wy1);
            const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;   // == (x0+1) - ix
            const bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x0 + 1 >= 0) & (x0 + 1 < W);
            const bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y0 + 1 >= 0) & (y0 + 1 < H);
            const unsigned r0 = (unsigned)(y0 * W + x0) * 256u + (unsigned)c4 * 16u, r1 = r0 + (unsigned)W * 256u;
            v[r][0] = buf_load4(r_in, (vx0 & vy0) ? r0 : OOB);
            v[r][1] = buf_load4(r_in, (vx1 & vy0) ? r0 + 256u : OOB);
            v[r][2] = buf_load4(r_in, (vx0 & vy1) ? r1 : OOB);
            v[r][3] = buf_load4(r_in, (vx1 & vy1) ? r1 + 256u : OOB);
            wt[r][0] = wx0 * wy0, wt[r][1] = wx1 * wy0, wt[r][2] = wx0 * wy1, wt[r][3] = wx1 * wy1;
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int y = blockIdx.y * RPT + r;
        // the same expression, in the same order, as the general kernel above (bit-identical results)
        f32x4 o = NEAREST ? v[r][0] : v[r][0] * wt[r][0] + v[r][1] * wt[r][1] + v[r][2] * wt[r][2] + v[r][3] * wt[r][3];
        const unsigned pix = (unsigned)y * (unsigned)W + (unsigned)x;
        const bool ok = col_ok & (y < H);
        if (OUT16) {
            const h4 hv = to_h4(o);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), r_out, ok ? (int)(pix * 128u + (unsigned)c4 * 8u) : (int)OOB, 0, 0);
        } else {
            buf_store4(r_out, ok ? pix * 256u + (unsigned)c4 * 16u : OOB, o);
        }
    }
}

// Drop-in for flow_warp(x, flow): x (n,c,h,w) NCHW, flow (n,h,w,2) = (dx,dy) pixels.
__global__ __launch_bounds__(256) void flow_warp_nchw_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ flow,
                                                             float* __restrict__ out, int N, int C, int H, int W,
                                                             int nearest) {
    const long hw = (long)H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)N * hw) return;
    const int n = (int)(p / hw);
    const long pix = p - (long)n * hw;
    const int yy = (int)(pix / W), xx = (int)(pix - (long)yy * W);
    int x0, y0;
    float wx1, wy1;
    tap_setup((float)xx, (float)yy, flow[p * 2], flow[p * 2 + 1], H, W, x0, y0, wx1, wy1);
    const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    const bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x0 + 1 >= 0) & (x0 + 1 < W);
    const bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y0 + 1 >= 0) & (y0 + 1 < H);
    float w00 = (vx0 & vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 & vy0) ? wx1 * wy0 : 0.f;
    float w10 = (vx0 & vy1) ? wx0 * wy1 : 0.f, w11 = (vx1 & vy1) ? wx1 * wy1 : 0.f;
    if (nearest) {      // the one pixel at nearbyint(ix), nearbyint(iy) (ties to even) with weight 1, as tap 00
        float ix, iy;
        sample_coords((float)xx, (float)yy, flow[p * 2], flow[p * 2 + 1], H, W, ix, iy);
        x0 = (int)rintf(ix);
        y0 = (int)rintf(iy);
        w00 = ((x0 >= 0) & (x0 < W) & (y0 >= 0) & (y0 < H)) ? 1.f : 0.f;
        w01 = w10 = w11 = 0.f;
    }
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
    const long o00 = (long)cy0 * W + cx0, o01 = (long)cy0 * W + cx1;
    const long o10 = (long)cy1 * W + cx0, o11 = (long)cy1 * W + cx1;
    const float* xb = x + (long)n * C * hw;
    float* ob = out + (long)n * C * hw + pix;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float* xc = xb + (long)c * hw;
        ob[(long)c * hw] = xc[o00] * w00 + xc[o01] * w01 + xc[o10] * w10 + xc[o11] * w11;
    }
}

}  // namespace

int launch_mv_warp_nhwc(const float* feat, const float* fx, const float* fy, void* out, int H, int W, int C,
                        hipStream_t stream, bool out_f16, bool nearest) {
    if (C % 4) return PNP_ERR_BAD_ARG;
    if (C == 64 && (long)H * W * 256 < (1L << 32)) {        // 32-bit byte offsets through buffer descriptors
        constexpr int RPT = WARP_RPT;
        auto k64 = out_f16 ? (nearest ? mv_warp_nhwc64_kernel<true, true, RPT> : mv_warp_nhwc64_kernel<true, false, RPT>)
                           : (nearest ? mv_warp_nhwc64_kernel<false, true, RPT> : mv_warp_nhwc64_kernel<false, false, RPT>);
        hipLaunchKernelGGL(k64, dim3((unsigned)((W + 15) / 16), (unsigned)((H + RPT - 1) / RPT)), dim3(256), 0, stream, feat, fx, fy, out,
                           H, W);
        return (int)hipGetLastError();
    }
    const long total = (long)H * W * (C / 4);
    long blocks = (total + 255) / 256;
    const long cap = 256L * 32;            // 32 blocks per CU worth of grid, grid-stride beyond
    if (blocks > cap) blocks = cap;
    auto kern = out_f16 ? (nearest ? mv_warp_nhwc_kernel<true, true> : mv_warp_nhwc_kernel<true, false>)
                        : (nearest ? mv_warp_nhwc_kernel<false, true> : mv_warp_nhwc_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, feat, fx, fy, out, H, W, C / 4, total);
    return (int)hipGetLastError();
}

int launch_flow_warp_nchw(const float* x, const float* flow, float* out, int N, int C, int H, int W,
                          hipStream_t stream, bool nearest) {
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(flow_warp_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, flow,
                       out, N, C, H, W, nearest ? 1 : 0);
    return (int)hipGetLastError();
}
