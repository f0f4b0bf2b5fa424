// Upper bound for a Winograd F(4x4,3x3) 64 -> 64 channel 3x3 conv on the fp32 matrix pipe of MI355X, next to the
// F(2x2,3x3) loop of ub_winograd.hip (211 effective TFLOP/s as a first build, 246 on static LDS).
//
// F(4x4,3x3): a 4x4 output tile from a 6x6 patch with 36 channel contractions instead of 144 tap products: 4x fewer matrix FLOPs than
// the direct form, 1.78x fewer than F(2x2).  What it costs on this chip:
//   * accumulators: 36 positions x N tiles x 4 registers.  With the F(2x2) kernel's 4 N tiles (a wave owns all 64 output channels, so
//     the input transform is computed once) that is 576 registers -- a lane has 512.  TWO N tiles (32 channels per wave): 288, i.e. more
//     than the 256 AGPRs: 8 accumulator quads have to live in arch VGPRs beside V and the fragments.
//   * the input transform B^T d B has multiplies (4, 5, 2): 12 FMA/add per 6-vector, 12 vectors per (tile, channel quad) = 144 float4
//     ops = 576 vector-ALU instructions per step of 288 MFMAs (F(2x2): 128 per 256), and with two N tiles per wave it is computed
//     twice per pixel.  fp32 MFMAs execute on the vector ALUs (ub_valu_gap.hip): every one of them is matrix time lost.
// Measured here, per wave on LDS-resident data (no global traffic, no barriers: an upper bound of the K loop only):
//   mode 0: the 288 MFMAs of a step back to back on register operands (does the register file hold 288 accumulators at all?)
//   mode 1: + B fragments from LDS (72 ds_read_b128 per step)
//   mode 2: + the input transform from 36 patch reads (two rows of V at a time)
// Prints cycles per step and the effective direct-conv TFLOP/s it would correspond to for the chip (256 CUs x 4 waves), to compare with
// the F(2x2) loop measured the same way (ub_winograd mode 1 / 3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int LDS_F4 = 40 * 1024;      // float4 elements: 22x22-pixel halo slab of 16 channels (1936 float4 x 4 = 31 KB) + a B chunk

template <int MODE>
__global__ __launch_bounds__(256, 1) void wino_f4(const f32x4* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < LDS_F4 / 4; i += 256) smem[i] = src[i & 4095];
    __syncthreads();
    const int m = lane & 15, kq = lane >> 4;
    // tile m of the wave's 4x4 tiles of 4x4 pixels: patch origin (4 (m >> 2), 4 (m & 3)) in a 22-pixel-wide halo, 4 float4 per pixel
    const f32x4* dbase = smem + ((4 * (m >> 2)) * 22 + 4 * (m & 3)) * 4 + kq;
    const f32x4* bbase = smem + 8192 / 4 + lane;
    f32x4 acc[36][2];
#pragma unroll
    for (int p = 0; p < 36; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 V[2][6];          // two position rows of V at a time (a row = 6 positions x 4 k-steps)
#pragma unroll
    for (int i = 0; i < 12; ++i) V[i / 6][i % 6] = src[t + 256 * i];
    f32x4 bf[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bf[0][n] = bf[1][n] = src[t + 64 * n];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {            // position row pr: 6 positions x 2 N tiles x 4 k-steps = 48 MFMAs
                if (MODE >= 2) {
                    // V row (pr + 1) % 6 of this / the next step: the row combination of the patch rows it needs (B^T has <= 4 nonzeros per
                    // row), then the column transform -- 6 columns x (<= 4 reads + 3 ops) + 12 ops
                    const int r = (pr + 1) % 6;
                    f32x4 tt[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        const f32x4* col = dbase + c * 4 + ((s + (pr == 5)) & 3) * 0;
                        const f32x4 d0 = col[0 * 88], d1 = col[1 * 88], d2 = col[2 * 88], d3 = col[3 * 88], d4 = col[4 * 88], d5 = col[5 * 88];
                        tt[c] = r == 0 ? 4.f * d0 - 5.f * d2 + d4
                              : r == 1 ? (d4 - 4.f * d2) + (d3 - 4.f * d1)
                              : r == 2 ? (d4 - 4.f * d2) - (d3 - 4.f * d1)
                              : r == 3 ? (d4 - d2) + 2.f * (d3 - d1)
                              : r == 4 ? (d4 - d2) - 2.f * (d3 - d1)
                                       : 4.f * d1 - 5.f * d3 + d5;
                    }
                    f32x4* v = V[r & 1];
                    const f32x4 a = tt[4] - 4.f * tt[2], b = tt[3] - 4.f * tt[1], c2 = tt[4] - tt[2], e = tt[3] - tt[1];
                    v[0] = 4.f * tt[0] - 5.f * tt[2] + tt[4];
                    v[1] = a + b;
                    v[2] = a - b;
                    v[3] = c2 + 2.f * e;
                    v[4] = c2 - 2.f * e;
                    v[5] = 4.f * tt[1] - 5.f * tt[3] + tt[5];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pj = 0; pj < 6; ++pj) {
                    if (MODE >= 1) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) bf[(pj + 1) & 1][n] = bbase[((pr * 6 + pj) & 7) * 128 + n * 64];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[pr * 6 + pj][n] = mfma16(V[pr & 1][pj][k], bf[pj & 1][n][k], acc[pr * 6 + pj][n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 36; ++p) sum += acc[p][0] + acc[p][1];
    reinterpret_cast<f32x4*>(out)[blockIdx.x * 256 + t] = sum;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const f32x4* src, float* out, unsigned long long* cyc) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(wino_f4<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F4 * 4);
    const int blocks = 256, iters = 400;
    hipLaunchKernelGGL(wino_f4<MODE>, dim3(blocks), dim3(256), LDS_F4 * 4, 0, src, out, 2, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(wino_f4<MODE>, dim3(blocks), dim3(256), LDS_F4 * 4, 0, src, out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(blocks);
        hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double cs = 0;
        for (auto v : hc) cs += (double)v;
        cs /= blocks;
        // one "iteration" of a wave = 4 steps = 64 input channels for 16 tiles x 16 pixels x 32 output channels
        const double units = (double)blocks * 4 * iters;                                      // wave-iterations
        const double eff = units * 256.0 * 32 * 576 * 2 / (ms * 1e-3) / 1e12, exe = units * 4 * 288 * 2048 / (ms * 1e-3) / 1e12;
        printf("%-52s rep %d: %7.2f ms  effective %6.1f TFLOP/s  executed %6.1f TFLOP/s  %6.0f cycles/step (MFMA floor 9216)  clock %.2f GHz\n",
               name, rep, ms, eff, exe, cs / iters / 4, cs / (ms * 1e-3) / 1e9);
    }
}

int main() {
    f32x4* src;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&src, 65536 * 16);
    hipMalloc(&out, 256 * 256 * 16);
    hipMalloc(&cyc, 256 * 8);
    std::vector<float> h(65536 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<0>("F(4x4,3x3) 288 MFMAs per step on register operands", src, out, cyc);
        run<1>("  + B fragments from LDS", src, out, cyc);
        run<2>("  + input transform from 36 patch reads per V row", src, out, cyc);
    }
    return 0;
}
