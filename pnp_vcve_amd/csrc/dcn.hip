// Flow-guided modulated deformable alignment: the reference's optional aligners
// BasiceformableAlignment / FVCDeformableAlignment (mmedit/models/backbones/sr_backbones/iconvsr_mv.py:21-84)
// call mmcv.ops.modulated_deform_conv2d(x, offset, mask, W(64,64,3,3), bias, stride 1, pad 1, dil 1,
// groups 1, deform_groups 16).  mmcv-full is not vendored in the reference: the arithmetic below restates
// its published semantics (modulated_deformable_im2col + dmcn_im2col_bilinear: a sample is taken only if
// -1 < h_im < H and -1 < w_im < W, each of the four corners contributes only if it lies inside the image;
// out = W . (mask * sample) + bias) -- PARITY UNPINNED against mmcv itself, checked against oracle/cpu_ref.py.
//
// Structure: the deformable im2col never exists in memory.  For each of the 9 taps the block gathers the
// modulated samples of its 128 pixels x 16 deform groups (one float4 = the 4 channels of a group, thanks to
// the pixel-major layout) straight into the LDS A chunk of the fp32-MFMA GEMM used by conv_mfma.hip, then
// runs the 8 q-steps of that tap against the packed weights.  Gather-bound (9 x 16 x 4 corner float4 per
// pixel through L1/TA), MFMA only for the 64x576 contraction.
#include "dcn.h"
#include <mutex>

namespace {

constexpr int PSTR = 17;                 // float4 per LDS pixel (256 B + 16 B pad, as in conv_mfma.hip)
constexpr int CH4 = PNP_CHUNK_Q * 2 * 64;
constexpr int LDS_BYTES = (128 * PSTR + 2 * CH4) * 16;

__global__ __launch_bounds__(256, 2) void dcn_mfma_kernel(const DcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sG = reinterpret_cast<f32x4*>(smem_raw);      // 128 px x 16 groups (+pad)
    f32x4* sB = sG + 128 * PSTR;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + 15) / 16;
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * 8, tx0 = (tile % tiles_x) * 16;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
    const f32x4* wimg = reinterpret_cast<const f32x4*>(a.w);
    const f32x4* a_lane = sG + (wave * 32 + m) * PSTR + h;
    const int g = t & 15;                      // this thread's deform group
    f32x4 breg[4];

    for (int k = 0; k < 9; ++k) {
        // ---- weights of tap k: global -> registers (written to LDS after the gather)
#pragma unroll
        for (int i = 0; i < 4; ++i) breg[i] = wimg[(long)k * CH4 + t + 256 * i];
        // ---- gather: 8 (pixel, group) items per thread; 16 consecutive lanes = the 16 groups of one pixel
        const int ky = k / 3 - 1, kx = k % 3 - 1;
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
            const int p = (t >> 4) + 16 * it;            // pixel of the tile
            const int gy = ty0 + (p >> 4), gx = tx0 + (p & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < H && gx < W) {
                const float* rec = a.om + ((long)gy * W + gx) * 448;
                const float2 off = *reinterpret_cast<const float2*>(rec + k * 32 + g * 2);
                const float mraw = rec[288 + k * 16 + g];
                float dy = off.x, dx = off.y;
                if (a.fx) {                              // 'basic': offset + flow.flip(1) (iconvsr_mv.py:77)
                    dy += a.fy[(long)gy * W + gx];
                    dx += a.fx[(long)gy * W + gx];
                }
                const float hi = (float)(gy + ky) + dy, wi = (float)(gx + kx) + dx;
                if (hi > -1.f && wi > -1.f && hi < (float)H && wi < (float)W) {
                    const float hl = floorf(hi), wl = floorf(wi);
                    const int h0 = (int)hl, w0 = (int)wl, h1 = h0 + 1, w1 = w0 + 1;
                    const float lh = hi - hl, lw = wi - wl, hh = 1.f - lh, hw = 1.f - lw;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 v1 = (h0 >= 0 && w0 >= 0) ? x4[((long)h0 * W + w0) * 16 + g] : z;
                    const f32x4 v2 = (h0 >= 0 && w1 <= W - 1) ? x4[((long)h0 * W + w1) * 16 + g] : z;
                    const f32x4 v3 = (h1 <= H - 1 && w0 >= 0) ? x4[((long)h1 * W + w0) * 16 + g] : z;
                    const f32x4 v4 = (h1 <= H - 1 && w1 <= W - 1) ? x4[((long)h1 * W + w1) * 16 + g] : z;
                    const float mask = 1.f / (1.f + expf(-mraw));      // torch.sigmoid (iconvsr_mv.py:79)
                    v = (v1 * (hh * hw) + v2 * (hh * lw) + v3 * (lh * hw) + v4 * (lh * lw)) * mask;
                }
            }
            sG[p * PSTR + g] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sB[(k & 1) * CH4 + t + 256 * i] = breg[i];
        __syncthreads();
        // ---- 8 q-steps of this tap
        const f32x4* bb = sB + (k & 1) * CH4 + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 av = a_lane[2 * q];
            const f32x4 b0 = bb[(q * 2) * 64], b1 = bb[(q * 2 + 1) * 64];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], acc[1], 0, 0, 0);
            }
        }
        __syncthreads();       // sG is rewritten by the next tap's gather
    }

    // ---- bias, row-wise store through the (now free) gather buffer
    const int n0 = lane & 31;
    float* sT = reinterpret_cast<float*>(sG) + wave * (32 * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float b = a.bias[j * 32 + n0];
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + n0] = acc[j][r] + b;
    }
    asm volatile("" ::: "memory");
    const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
    const int ec = lane & 15, ep = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = ep + i * 4;                        // pixel inside the wave's 32 (rows 2*wave, 2*wave+1)
        const int gy = ty0 + 2 * wave + (p >> 4), gx = tx0 + (p & 15);
        if (gy < H && gx < W)
            *reinterpret_cast<f32x4*>(a.out + ((long)gy * W + gx) * 64 + ec * 4) = sT4[p * 16 + ec];
    }
}

}  // namespace

int launch_dcn(const DcnArgs& a, hipStream_t stream) {
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_mfma_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((a.W + 15) / 16) * ((a.H + 7) / 8);
    hipLaunchKernelGGL(dcn_mfma_kernel, dim3(tiles), dim3(256), LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}
