// conv_last (64 -> 3 channels, 3x3) + the low-quality frame, on the VECTOR ALUs (fp32 path).
//   iconvsr_ipb_par.py:140-146:  out = conv_last(hr) + lr            (out_mode 2)
//                                out = conv_last(hr) + bilinear_x4(lr)   (out_mode 3, the x4 heads)
//
// Why not the matrix cores: with 3 output channels a 32-wide MFMA N tile does 10.7x the useful work
// (conv3x3_mfma_kernel<4,1,1,1>: 327 us per 720p frame, 10 TFLOP/s of useful FLOPs).  The useful work is only
// 3.2 GFLOP per frame: one thread per output pixel, 3 accumulators, the 1728 weights as SCALAR operands
// (wave-uniform loads -> SGPRs, `v_fmac v, s, v`), the pixel's 9 x 64 inputs from an LDS halo tile as
// conflict-free ds_read_b128 (pixel stride 272 B as in conv_mfma.hip).  12 FMAs per LDS read: VALU-bound,
// ~6900 issue cycles per wave.
//
// Block = 128 threads = one 8x16 output tile; LDS 10 x 18 x 272 B = 48960 B -> 3 blocks per CU.
#include "conv_mfma.h"

namespace {

constexpr int TH = 8, TW = 16, PW = 18, PIX = (TH + 2) * PW, PSTR = 17;
constexpr int LDS_BYTES = PIX * PSTR * 16;
constexpr int SIT = (PIX * 16 + 127) / 128;          // float4 halo loads per thread

// wv: [9 taps][64 input channels][4] = (co 0, co 1, co 2, 0), made by pack_last_valu_kernel
// KS = 1: one thread per pixel (128 threads).  KS = 4 (small frames: fewer tiles than CUs x 3, a block's 1728-FMA chain and its LDS
// latency are the launch): 512 threads, thread group g = t >> 7 contracts input channels 16 g .. 16 g + 15 of every tap for pixel
// t & 127 (the weights stay wave-uniform scalars), the four partial sums meet in LDS -- 16 -> 6 us per 128x128 frame; the
// summation order differs from KS = 1 in the last bits (both are held to 2e-6 against the matrix-core form).
template <int KS>
__global__ __launch_bounds__(128 * KS) void conv_last_valu_kernel(const ConvArgs a, const float* __restrict__ wv) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sA = reinterpret_cast<f32x4*>(smem_raw);
    const int t = KS == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 127), grp = KS == 1 ? 0 : (int)(threadIdx.x >> 7);      // (KS = 1: compile-time bounds below)
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    int tile;
    {   // XCD-aware: blocks b, b+8, ... (one XCD) walk a contiguous band of tiles
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const float* sp = a.src[0];

    // ---- halo tile: all loads in flight, then into LDS
    f32x4 reg[SIT];
#pragma unroll
    for (int k = 0; k < SIT; ++k) {
        if (KS > 1 && (k & (KS - 1)) != grp) continue;          // (the groups share the halo loads)
        const int i = t + 128 * k;
        const int pix = (i >> 4) < PIX ? (i >> 4) : PIX - 1, c16 = i & 15;
        const int ry = pix / PW, rx = pix - ry * PW;
        const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + ((long)cy * W + cx) * 64 + c16 * 4);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        reg[k] = inb ? v : z;
    }
    // the frame to add (requested now, used at the end)
    const int py = t >> 4, px = t & 15;
    const int gy = ty0 + py, gx = tx0 + px;
    const bool inb = gy < H && gx < W;
    float base[3] = {0.f, 0.f, 0.f};
    if (inb) {
        if (a.out_mode == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) base[c] = a.lr[c * a.lr_plane + (long)gy * W + gx];
        } else {
            // F.interpolate(scale_factor=4, bilinear, align_corners=False) (iconvsr_ipb_par.py:41,140):
            // src = (dst + 0.5) / 4 - 0.5, clamped at 0
            const int lh = H >> 2, lw = W >> 2;
            float sy = (gy + 0.5f) * 0.25f - 0.5f, sx = (gx + 0.5f) * 0.25f - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < lh - 1 ? 1 : 0), x1 = x0 + (x0 < lw - 1 ? 1 : 0);
            const float ly = sy - y0, lx = sx - x0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* p = a.lr + c * a.lr_plane;
                const float v00 = p[(long)y0 * lw + x0], v01 = p[(long)y0 * lw + x1];
                const float v10 = p[(long)y1 * lw + x0], v11 = p[(long)y1 * lw + x1];
                base[c] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < SIT; ++k) {
        const int i = t + 128 * k;
        if ((KS == 1 || (k & (KS - 1)) == grp) && i < PIX * 16) sA[(i >> 4) * PSTR + (i & 15)] = reg[k];
    }
    __syncthreads();

    // ---- 9 x 64 x 3 FMAs per pixel; weights are wave-uniform -> scalar loads
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    const f32x4* wv4 = reinterpret_cast<const f32x4*>(wv);
    const f32x4* xp = sA + (py * PW + px) * PSTR;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const f32x4* xt = xp + (dy * PW + dx) * PSTR;
#pragma unroll 4
        for (int c4 = (16 / KS) * grp; c4 < (16 / KS) * (grp + 1); ++c4) {
            const f32x4 x = xt[c4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 wq = wv4[(tap * 64 + c4 * 4 + j)];      // uniform address: s_load
                acc0 = __builtin_fmaf(x[j], wq[0], acc0);
                acc1 = __builtin_fmaf(x[j], wq[1], acc1);
                acc2 = __builtin_fmaf(x[j], wq[2], acc2);
            }
        }
    }
    if constexpr (KS > 1) {
        __syncthreads();                                       // every group is done with the halo tile: its first bytes hold the partial sums
        float* red = reinterpret_cast<float*>(smem_raw);
        if (grp > 0) {
            red[((grp - 1) * 128 + t) * 3 + 0] = acc0;
            red[((grp - 1) * 128 + t) * 3 + 1] = acc1;
            red[((grp - 1) * 128 + t) * 3 + 2] = acc2;
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 0; g < KS - 1; ++g) {
            acc0 += red[(g * 128 + t) * 3 + 0];
            acc1 += red[(g * 128 + t) * 3 + 1];
            acc2 += red[(g * 128 + t) * 3 + 2];
        }
    }
    if (inb) {
        const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
        const long o = (long)gy * W + gx, plane = (long)H * W;
        float v[3] = {acc0 + a.bias[0], acc1 + a.bias[1], acc2 + a.bias[2]};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float u = fmaxf(v[c], 0.f) + neg_slope * fminf(v[c], 0.f);
            a.out[c * plane + o] = u + base[c];
        }
    }
}

// OIHW (3, 64, 3, 3) -> [tap][ci][4]
__global__ __launch_bounds__(256) void pack_last_valu_kernel(const float* __restrict__ w, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // over 9 * 64 * 4
    if (i >= 9 * 64 * 4) return;
    const int co = i & 3, ci = (i >> 2) & 63, tap = i >> 8;
    dst[i] = co < 3 ? w[(co * 64 + ci) * 9 + tap] : 0.f;
}

}  // namespace

int launch_pack_last_valu(const float* w_oihw, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(pack_last_valu_kernel, dim3(9), dim3(256), 0, stream, w_oihw, dst);
    return (int)hipGetLastError();
}

bool conv_last_valu_eligible(const ConvArgs& a, int cfg, int grid_y) {
    return cfg == CONV_CFG_RGB && grid_y == 1 && a.wvalu && a.nsrc == 1 && a.src_c[0] == 64 &&
           (a.out_mode == 2 || a.out_mode == 3) && a.lr && a.bias && !a.wpar && !a.residual && !a.gamma && !a.src_f16 &&
           !a.out_f16;
}

int launch_conv_last_valu(const ConvArgs& a, hipStream_t stream) {
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    if (tiles < 768) hipLaunchKernelGGL(conv_last_valu_kernel<4>, dim3(tiles), dim3(512), LDS_BYTES, stream, a, a.wvalu);
    else hipLaunchKernelGGL(conv_last_valu_kernel<1>, dim3(tiles), dim3(128), LDS_BYTES, stream, a, a.wvalu);
    return (int)hipGetLastError();
}
