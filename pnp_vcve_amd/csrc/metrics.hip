// On-device PSNR statistic (the step right after the hot path).
//
// Reference: BasicVSR.evaluate (mmedit/models/restorers/basicvsr.py:119-153) moves every frame
// to the host, tensor2img (mmedit/core/misc.py:51-71: clamp to [0,1], *255, round -> uint8) and
// psnr (mmedit/core/evaluation/metrics.py:200-215: mean squared uint8 difference).
// Here: one pass over the two frames in HBM, rounding exactly as numpy does (half to even), the
// squared differences summed as 64-bit INTEGERS (exact, order-independent, deterministic), so the
// host only sees 8 bytes per frame instead of 2 x 3*H*W*4.  HBM-bound: 8 B per element.
#include "common.h"

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

}  // namespace

extern "C" int pnp_psnr_sse_f32(const float* a, const float* b, unsigned long long* sse, int frames, int c, int h,
                                int w, int crop_border, void* stream) {
    if (frames < 1 || c < 1 || h < 1 || w < 1 || crop_border < 0 || 2 * crop_border >= h || 2 * crop_border >= w)
        return PNP_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sse, 0, sizeof(unsigned long long) * frames, st);
    if (e != hipSuccess) return (int)e;
    const long per = (long)c * h * w;
    long bx = (per / 4 + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(psnr_sse_kernel, dim3((unsigned)bx, frames), dim3(256), 0, st, a, b, sse, c, h, w, crop_border);
    return (int)hipGetLastError();
}
