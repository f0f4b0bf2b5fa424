// 3x3 convolution (+ fused per-pixel-weighted 1x1 branches) as an implicit GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD on gfx950).
//
// Restates, as ONE kernel family, the reference's
//   * input_conv + LeakyReLU           mmedit/models/backbones/sr_backbones/basicvsr_net.py:484,515
//     over the *virtual* concat [lr, key_warp, neighbor(, bwd)]  iconvsr_ipb_par.py:90,125
//   * BAE block front half             mmedit/models/common/sr_backbone_utils.py:310-311
//       relu( gamma * (conv3x3(x; Wagg) + bagg) + sum_j par_j * conv1x1_j(x) )
//   * BAE block back half              sr_backbone_utils.py:313,329   x + conv1(o)
//   * conv_hr / conv_last heads        iconvsr_ipb_par.py:144-146
//   * PixelShufflePack convs           mmedit/models/common/upsample.py:49-50
//
// GEMM view:  M = pixels, N = output channels, K = (source, tap, input channel).
// Feature maps are pixel-major (NHWC, 64 fp32 = 256 B per pixel) so that a halo tile is a
// handful of fully coalesced row reads and an A fragment is one ds_read_b128.
//
// Block = 256 threads = 4 waves.  Each wave owns one 32-pixel M tile (2 rows x 16 columns)
// and NT 32-channel N tiles.  Per q-step a lane reads ONE float4 of A (4 consecutive input
// channels of its pixel) and NT float4 of B and issues 4*NT MFMAs: the K order inside a
// q-step is permuted identically for A and B (common.h) so no shuffles are needed.
//
// LDS:  A tile  (TH+2) x 18 pixels x 256 B, 16-byte slots XOR-swizzled with the tile column
//       (conflict-free ds_read_b128 across 16 consecutive columns);
//       B chunks 2 x (8 q-steps x NTB x 1 KiB), double-buffered through registers
//       (global_load early, ds_write after the chunk's MFMAs, one barrier per chunk).
//       8x16 tile: 46080 + 32768 = 78848 B  -> two blocks per CU (2 waves per SIMD), so one
//       block's halo fill / epilogue overlaps the other's MFMA stream.
#include "conv_mfma.h"
#include <mutex>

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int WM, int WN, int NT, int NTB>
struct ConvCfg {
    static constexpr int TH = 2 * WM;
    static constexpr int TW = 16;
    static constexpr int PW = TW + 2;
    static constexpr int PIX = (TH + 2) * PW;
    static constexpr int CH4 = PNP_CHUNK_Q * NTB * 64;      // float4 per chunk
    static constexpr int LDS_BYTES = (PIX * 16 + 2 * CH4) * 16;
};

template <int WM, int WN, int NT, int NTB>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvArgs a) {
    using C = ConvCfg<WM, WN, NT, NTB>;
    constexpr int TH = C::TH, TW = C::TW, PW = C::PW, PIX = C::PIX, CH4 = C::CH4;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(NT * WN == NTB, "N tiles");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sA = reinterpret_cast<f32x4*>(smem_raw);
    f32x4* sB = sA + PIX * 16;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15;
    const int H = a.H, W = a.W;

    // ---- tile id, remapped so that each XCD (blocks b, b+8, ...) walks a contiguous band
    const int tiles_x = (W + TW - 1) / TW;
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const int yimg = blockIdx.y;

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- epilogue operands are fetched NOW so that their HBM/L2 latency hides under the K loop
    //      (the residual may alias `out` element-for-element, which would otherwise serialise
    //      every load behind the previous store).
    const int n0 = lane & 31;
    float bco[NT], gco[NT], pv[3] = {0.f, 0.f, 0.f};
    float res[NT][16];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = (wn * NT + j) * 32 + n0;
        bco[j] = a.bias ? a.bias[yimg * a.bias_ystride + co] : 0.f;
        gco[j] = a.gamma ? a.gamma[co] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int gy = ty0 + 2 * wm + (mm >> 4), gx = tx0 + (mm & 15);
            res[j][r] = (a.residual && gy < H && gx < W) ? a.residual[((long)gy * W + gx) * 64 + co] : 0.f;
        }
    }
    if (a.wpar) {
        const int gy = ty0 + 2 * wm + my, gx = tx0 + mx;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            pv[jj] = (gy < H && gx < W) ? a.par[jj * a.par_plane + (long)gy * W + gx] : 0.f;
    }

    // chunk bookkeeping: chunks of source s are wsrc[s] + tap*CH4*4 floats
    int nchunks = 0;
    for (int s = 0; s < a.nsrc; ++s) nchunks += (a.src_c[s] == 64) ? 9 : 1;
    const int nmain = nchunks;
    if (a.wpar) nchunks += 3;
    const long yoff = (long)yimg * a.w_ystride;

    auto chunk_ptr = [&](int c) -> const f32x4* {
        // wave-uniform walk over at most 4 sources
        int base = 0;
        for (int s = 0; s < a.nsrc; ++s) {
            const int n = (a.src_c[s] == 64) ? 9 : 1;
            if (c < base + n) return reinterpret_cast<const f32x4*>(a.wsrc[s] + yoff) + (long)(c - base) * CH4;
            base += n;
        }
        return reinterpret_cast<const f32x4*>(a.wpar + yoff) + (long)(c - base) * CH4;
    };

    constexpr int BPT = CH4 / 256;
    f32x4 breg[BPT];
    {   // chunk 0 -> sB[0]
        const f32x4* g = chunk_ptr(0);
#pragma unroll
        for (int i = 0; i < BPT; ++i) breg[i] = g[t + 256 * i];
#pragma unroll
        for (int i = 0; i < BPT; ++i) sB[t + 256 * i] = breg[i];
    }

    int c = 0;   // global chunk counter
    for (int s = 0; s < a.nsrc; ++s) {
        const float* sp = a.src[s];
        const bool wide = (a.src_c[s] == 64);
        // ------------------------------------------------------------ stage the halo tile
        if (wide) {
            constexpr int AIT = (PIX * 16 + 255) / 256;
            f32x4 areg[AIT];
#pragma unroll
            for (int k = 0; k < AIT; ++k) {
                const int i = t + 256 * k;
                const int pix = i >> 4, c16 = i & 15;
                const int ry = pix / PW, rx = pix - ry * PW;
                const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (i < PIX * 16 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v = *reinterpret_cast<const f32x4*>(sp + ((long)gy * W + gx) * 64 + c16 * 4);
                areg[k] = v;
            }
#pragma unroll
            for (int k = 0; k < AIT; ++k) {
                const int i = t + 256 * k;
                const int pix = i >> 4, c16 = i & 15;
                const int ry = pix / PW, rx = pix - ry * PW;
                if (i < PIX * 16) sA[pix * 16 + (c16 ^ (rx & 15))] = areg[k];
            }
        } else {
            for (int i = t; i < PIX; i += 256) {
                const int ry = i / PW, rx = i - ry * PW;
                const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v = *reinterpret_cast<const f32x4*>(sp + ((long)gy * W + gx) * 4);
                sA[i] = v;
            }
        }
        __syncthreads();

        const int ntap = wide ? 9 : 1;
        for (int tap = 0; tap < ntap; ++tap) {
            const bool more = (c + 1 < nchunks);
            if (more) {
                const f32x4* g = chunk_ptr(c + 1);
#pragma unroll
                for (int i = 0; i < BPT; ++i) breg[i] = g[t + 256 * i];
            }
            const f32x4* bb = sB + (c & 1) * CH4 + (wn * NT) * 64 + lane;
            if (wide) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const int apix = (2 * wm + my + dy) * PW + mx + dx;
                const int sw = (mx + dx) & 15;
                // fragments of q-step q+1 are read while the MFMAs of q-step q issue
                f32x4 av = sA[apix * 16 + (h ^ sw)];
                f32x4 bv[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = bb[j * 64];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x4 an = av;
                    f32x4 bn[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bn[j] = bv[j];
                    if (q < 7) {
                        an = sA[apix * 16 + ((2 * (q + 1) + h) ^ sw)];
#pragma unroll
                        for (int j = 0; j < NT; ++j) bn[j] = bb[((q + 1) * NTB + j) * 64];
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[j] = mfma32(av[kk], bv[j][kk], acc[j]);
                    av = an;
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[j] = bn[j];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int tp0 = 2 * q, tp1 = (2 * q + 1 > 8) ? 8 : 2 * q + 1;
                    const int tp = h ? tp1 : tp0;
                    const int dy = tp / 3, dx = tp - dy * 3;
                    const f32x4 av = sA[(2 * wm + my + dy) * PW + mx + dx];
                    f32x4 bv[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[j] = bb[(q * NTB + j) * 64];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[j] = mfma32(av[kk], bv[j][kk], acc[j]);
                }
            }
            if (more) {
                f32x4* d = sB + ((c + 1) & 1) * CH4;
#pragma unroll
                for (int i = 0; i < BPT; ++i) d[t + 256 * i] = breg[i];
            }
            __syncthreads();
            ++c;
        }
    }

#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] + bco[j]) * gco[j];

    // ---- fused 1x1 partition branches: K-extension with A scaled by par_j(pixel)
    if (a.wpar) {
        const int apix = (2 * wm + my + 1) * PW + mx + 1;
        const int sw = (mx + 1) & 15;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const bool more = (c + 1 < nchunks);
            if (more) {
                const f32x4* g = chunk_ptr(c + 1);
#pragma unroll
                for (int i = 0; i < BPT; ++i) breg[i] = g[t + 256 * i];
            }
            const f32x4* bb = sB + (c & 1) * CH4 + (wn * NT) * 64 + lane;
            const float ps = pv[jj];
            f32x4 av = sA[apix * 16 + (h ^ sw)];
            f32x4 bv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = bb[j * 64];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 an = av;
                f32x4 bn[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bn[j] = bv[j];
                if (q < 7) {
                    an = sA[apix * 16 + ((2 * (q + 1) + h) ^ sw)];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bn[j] = bb[((q + 1) * NTB + j) * 64];
                }
                av *= ps;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = mfma32(av[kk], bv[j][kk], acc[j]);
                av = an;
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = bn[j];
            }
            if (more) {
                f32x4* d = sB + ((c + 1) & 1) * CH4;
#pragma unroll
                for (int i = 0; i < BPT; ++i) d[t + 256 * i] = breg[i];
            }
            __syncthreads();
            ++c;
        }
    }
    (void)nmain;

    // ---- epilogue: activation, residual, store.  Accumulator register r of lane (n0,h)
    //      is pixel m = (r&3) + 8*(r>>2) + 4*h of the wave's M tile, channel n0 of N tile j.
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = (wn * NT + j) * 32 + n0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int gy = ty0 + 2 * wm + (mm >> 4), gx = tx0 + (mm & 15);
            if (gy >= H || gx >= W) continue;
            float v = acc[j][r];
            if (a.act == 1) v = fmaxf(v, 0.f);
            else if (a.act == 2) v = v > 0.f ? v : 0.1f * v;
            if (a.out_mode == 0) {
                a.out[((long)gy * W + gx) * 64 + co] = v + res[j][r];
            } else if (a.out_mode == 1) {
                const int oy = 2 * gy + (yimg >> 1), ox = 2 * gx + (yimg & 1);
                a.out[((long)oy * (2 * W) + ox) * 64 + co] = v;
            } else if (co < 3) {
                float base;
                if (a.out_mode == 2) {
                    base = a.lr[co * a.lr_plane + (long)gy * W + gx];
                } else {
                    // F.interpolate(scale_factor=4, bilinear, align_corners=False)
                    // (iconvsr_ipb_par.py:41,140): src = (dst + 0.5) / 4 - 0.5, clamped at 0
                    const int lh = H >> 2, lw = W >> 2;
                    float sy = (gy + 0.5f) * 0.25f - 0.5f, sx = (gx + 0.5f) * 0.25f - 0.5f;
                    sy = sy < 0.f ? 0.f : sy;
                    sx = sx < 0.f ? 0.f : sx;
                    const int y0 = (int)sy, x0 = (int)sx;
                    const int y1 = y0 + (y0 < lh - 1 ? 1 : 0), x1 = x0 + (x0 < lw - 1 ? 1 : 0);
                    const float ly = sy - y0, lx = sx - x0;
                    const float* p = a.lr + co * a.lr_plane;
                    const float v00 = p[(long)y0 * lw + x0], v01 = p[(long)y0 * lw + x1];
                    const float v10 = p[(long)y1 * lw + x0], v11 = p[(long)y1 * lw + x1];
                    base = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                }
                a.out[(long)co * H * W + (long)gy * W + gx] = v + base;
            }
        }
    }
}

template <int WM, int WN, int NT, int NTB>
int launch_cfg(const ConvArgs& a, int grid_y, hipStream_t stream) {
    using C = ConvCfg<WM, WN, NT, NTB>;
    auto kern = conv3x3_mfma_kernel<WM, WN, NT, NTB>;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((a.W + C::TW - 1) / C::TW) * ((a.H + C::TH - 1) / C::TH);
    hipLaunchKernelGGL(kern, dim3(tiles, grid_y), dim3(256), C::LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}

}  // namespace

int conv_pick_cfg(int H, int W) {
    // 8x16 tiles need >= 2 blocks per CU (512 tiles) to fill the chip; below that use 4x16 tiles.
    const long tiles_big = (long)((W + 15) / 16) * ((H + 7) / 8);
    return tiles_big >= 1024 ? CONV_CFG_BIG : CONV_CFG_SMALL;
}

int launch_conv3x3(const ConvArgs& a, int cfg, int grid_y, hipStream_t stream) {
    if (a.nsrc < 1 || a.nsrc > 4) return PNP_ERR_BAD_ARG;
    if (a.wpar && (a.nsrc != 1 || a.src_c[0] != 64 || !a.par)) return PNP_ERR_BAD_ARG;
    for (int s = 0; s < a.nsrc; ++s)
        if (a.src_c[s] != 64 && a.src_c[s] != 4) return PNP_ERR_BAD_ARG;
    switch (cfg) {
        case CONV_CFG_BIG: return launch_cfg<4, 1, 2, 2>(a, grid_y, stream);
        case CONV_CFG_SMALL: return launch_cfg<2, 2, 1, 2>(a, grid_y, stream);
        case CONV_CFG_RGB: return launch_cfg<4, 1, 1, 1>(a, grid_y, stream);
    }
    return PNP_ERR_BAD_ARG;
}
