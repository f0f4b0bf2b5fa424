// On-device PSNR statistic (the step right after the hot path).
//
// Reference: BasicVSR.evaluate (mmedit/models/restorers/basicvsr.py:119-153) moves every frame
// to the host, tensor2img (mmedit/core/misc.py:51-71: clamp to [0,1], *255, round -> uint8) and
// psnr (mmedit/core/evaluation/metrics.py:200-215: mean squared uint8 difference).
// Here: one pass over the two frames in HBM, rounding exactly as numpy does (half to even), the
// squared differences summed as 64-bit INTEGERS (exact, order-independent, deterministic), so the
// host only sees 8 bytes per frame instead of 2 x 3*H*W*4.  HBM-bound: 8 B per element.
#include "prep.h"
#include <cmath>

namespace {

__device__ __forceinline__ int to_u8(float v) {
    v = fminf(fmaxf(v, 0.f), 1.f);
    return (int)rintf(v * 255.0f);
}

// a, b: (frames, C, H, W) fp32; sse: (frames) u64, zeroed by the launcher
__global__ __launch_bounds__(256) void psnr_sse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       unsigned long long* __restrict__ sse, int C, int H, int W,
                                                       int crop) {
    const int frame = blockIdx.y;
    const long plane = (long)H * W, per = plane * C;
    const float* fa = a + frame * per;
    const float* fb = b + frame * per;
    unsigned long long acc = 0;
    if (crop == 0 && (per & 3) == 0) {
        const f32x4* a4 = reinterpret_cast<const f32x4*>(fa);
        const f32x4* b4 = reinterpret_cast<const f32x4*>(fb);
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per / 4; i += (long)gridDim.x * blockDim.x) {
            const f32x4 x = a4[i], y = b4[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = to_u8(x[k]) - to_u8(y[k]);
                acc += (unsigned)(d * d);
            }
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
            const long p = i % plane;
            const int yy = (int)(p / W), xx = (int)(p - (long)yy * W);
            if (yy < crop || yy >= H - crop || xx < crop || xx >= W - crop) continue;
            const int d = to_u8(fa[i]) - to_u8(fb[i]);
            acc += (unsigned)(d * d);
        }
    }
    // wave reduction, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&sse[frame], acc);
}

// ---------------------------------------------------------------------------------------------
// SSIM (mmedit/core/evaluation/metrics.py:266-355): per channel, 11x11 Gaussian (sigma 1.5) 'valid' window
// over the uint8-rounded frames, in fp64 like the reference.  The reference filters with cv2.filter2D
// (absent here -> parity unpinned; checked against the numpy restatement in pnp_vcve_amd/metrics.py).
// One block = a 16x32 tile of the valid map of one (frame, channel): the two u8 tiles (26x42) go to LDS,
// a horizontal pass leaves the five windowed moments of 26x32 positions in LDS, the vertical pass finishes
// them and forms the SSIM ratio; each block writes ONE partial sum (deterministic; the host adds them).
struct SsimArgs {
    double g[11];
    const float* a;
    const float* b;
    double* partial;       // [frames*C][blocks_per_plane]
    int H, W, crop, oh, ow, tiles_x, blocks_per_plane;
};

__global__ __launch_bounds__(256) void ssim_kernel(const SsimArgs s) {
    __shared__ float ta[26 * 42], tb[26 * 42];
    __shared__ double hm[5][26 * 32];
    __shared__ double red[4];
    const int plane = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
    const int ty0 = (blk / s.tiles_x) * 16, tx0 = (blk % s.tiles_x) * 32;        // in valid-map coordinates
    const float* pa = s.a + (long)plane * s.H * s.W;
    const float* pb = s.b + (long)plane * s.H * s.W;
    for (int i = t; i < 26 * 42; i += 256) {
        const int r = i / 42, c = i - r * 42;
        const int y = s.crop + ty0 + r, x = s.crop + tx0 + c;
        float va = 0.f, vb = 0.f;
        if (y < s.H - s.crop && x < s.W - s.crop) {
            va = (float)to_u8(pa[(long)y * s.W + x]);
            vb = (float)to_u8(pb[(long)y * s.W + x]);
        }
        ta[i] = va;
        tb[i] = vb;
    }
    __syncthreads();
    for (int i = t; i < 26 * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const double x = ta[r * 42 + c + k], y = tb[r * 42 + c + k], gk = s.g[k];
            m1 += gk * x;
            m2 += gk * y;
            s11 += gk * x * x;
            s22 += gk * y * y;
            s12 += gk * x * y;
        }
        hm[0][i] = m1;
        hm[1][i] = m2;
        hm[2][i] = s11;
        hm[3][i] = s22;
        hm[4][i] = s12;
    }
    __syncthreads();
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    double acc = 0;
    for (int i = t; i < 16 * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        if (ty0 + r >= s.oh || tx0 + c >= s.ow) continue;
        double v[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 11; ++k)
#pragma unroll
            for (int q = 0; q < 5; ++q) v[q] += s.g[k] * hm[q][(r + k) * 32 + c];
        const double mu1 = v[0], mu2 = v[1];
        const double sg1 = v[2] - mu1 * mu1, sg2 = v[3] - mu2 * mu2, sg12 = v[4] - mu1 * mu2;
        acc += ((2 * mu1 * mu2 + C1) * (2 * sg12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (sg1 + sg2 + C2));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) s.partial[(long)plane * s.blocks_per_plane + blk] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace

extern "C" int pnp_ssim_blocks(int h, int w, int crop_border) {
    const int oh = h - 2 * crop_border - 10, ow = w - 2 * crop_border - 10;
    if (oh < 1 || ow < 1) return 0;
    return ((oh + 15) / 16) * ((ow + 31) / 32);
}

extern "C" int pnp_ssim_partials_f32(const float* a, const float* b, double* partials, int frames, int c, int h, int w,
                                     int crop_border, void* stream) {
    const int nb = pnp_ssim_blocks(h, w, crop_border);
    if (frames < 1 || c < 1 || nb < 1 || crop_border < 0) return PNP_ERR_BAD_ARG;
    SsimArgs s;
    double sum = 0;
    for (int i = 0; i < 11; ++i) {                       // cv2.getGaussianKernel(11, 1.5)
        s.g[i] = exp(-((i - 5.0) * (i - 5.0)) / (2 * 1.5 * 1.5));
        sum += s.g[i];
    }
    for (int i = 0; i < 11; ++i) s.g[i] /= sum;
    s.a = a;
    s.b = b;
    s.partial = partials;
    s.H = h;
    s.W = w;
    s.crop = crop_border;
    s.oh = h - 2 * crop_border - 10;
    s.ow = w - 2 * crop_border - 10;
    s.tiles_x = (s.ow + 31) / 32;
    s.blocks_per_plane = nb;
    hipLaunchKernelGGL(ssim_kernel, dim3(nb, frames * c), dim3(256), 0, (hipStream_t)stream, s);
    return (int)hipGetLastError();
}

extern "C" int pnp_psnr_sse_f32(const float* a, const float* b, unsigned long long* sse, int frames, int c, int h,
                                int w, int crop_border, void* stream) {
    if (frames < 1 || c < 1 || h < 1 || w < 1 || crop_border < 0 || 2 * crop_border >= h || 2 * crop_border >= w)
        return PNP_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int ze = launch_zero_words(sse, 2L * frames, st);      // a kernel, not a memset node (prep.h)
    if (ze != PNP_OK) return ze;
    const long per = (long)c * h * w;
    long bx = (per / 4 + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(psnr_sse_kernel, dim3((unsigned)bx, frames), dim3(256), 0, st, a, b, sse, c, h, w, crop_border);
    return (int)hipGetLastError();
}

// tensor2img for the write-back (mmedit/core/misc.py:51-71 + mmcv.imwrite, basicvsr.py:205-231): (frames,3,h,w) fp32
// planes -> (frames,h,w,3) uint8 RGB, clamp to [0,1], * 255, round half to even -- a quarter of the D2H bytes.
namespace {
__global__ __launch_bounds__(256) void frames_to_rgb8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                             long hw, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // over frames * hw pixels
    if (i >= total) return;
    const long f = i / hw, p = i - f * hw;
    const float* s = x + f * 3 * hw + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fminf(fmaxf(s[c * hw], 0.f), 1.f) * 255.0f;
        out[i * 3 + c] = (unsigned char)rintf(v);
    }
}
}  // namespace

extern "C" int pnp_frames_to_rgb8(const float* frames, unsigned char* out, int nframes, int h, int w, void* stream) {
    if (nframes < 1 || h < 1 || w < 1 || !frames || !out) return PNP_ERR_BAD_ARG;
    const long hw = (long)h * w, total = hw * nframes;
    hipLaunchKernelGGL(frames_to_rgb8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       frames, out, hw, total);
    return (int)hipGetLastError();
}
