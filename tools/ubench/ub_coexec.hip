// Does VALU work of a co-resident wave steal time from an fp32 MFMA stream (v_mfma_f32_32x32x2_f32)?
// 512 blocks of 256 threads (2 waves per SIMD).  Even blocks: MFMA loop.  Odd blocks: VALU loop of a given kind
// (or idle).  Reports cycles per MFMA for the MFMA blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>   // 0 idle, 1 v_fma stream, 2 int VALU, 3 LDS reads, 4 global loads (L2-resident 16 B/lane), 5 global stores 16 B/lane, 6 ds_write_b128
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* cyc, unsigned long long* hw, int iters, f32x4* gbuf) {
    __shared__ float lds[4096];
    const int t = threadIdx.x;
    lds[t] = t;
    __syncthreads();
    // which role? blocks are dealt round-robin to XCDs then CUs; pairs (b, b + 256) share a CU if the dispatcher
    // fills one block per CU first -- the host prints both pairings
    const bool mfma_role = (blockIdx.x < 256);
    if (mfma_role) {
        f32x16 a0 = {0}, a1 = {0};
        float x = t * 1e-3f, y = 1.0f;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
        out[blockIdx.x * 256 + t] = s;
        if ((t & 63) == 0) cyc[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
        if (t == 0) hw[blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xFF00);
    } else {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        float v0 = t, v1 = 1.0001f, v2 = 0.5f;
        int w0 = t, w1 = 7;
        float acc = 0;
        // run roughly as long as the MFMA role: iters * 2 MFMA * 64 cycles
        const int n = KIND == 0 ? 0 : iters * 8;
        for (int i = 0; i < n; ++i) {
            if (KIND == 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v0 = __builtin_fmaf(v0, v1, v2);
            } else if (KIND == 2) {
#pragma unroll
                for (int u = 0; u < 8; ++u) w0 = (w0 ^ w1) + (w0 >> 3);
            } else if (KIND == 3) {
#pragma unroll
                for (int u = 0; u < 2; ++u) acc += lds[(t + i + u * 64) & 4095];
            } else if (KIND == 4) {
                f32x4 v = gbuf[(size_t)blockIdx.x * 4096 + ((t + i * 256) & 4095)];
                acc += v[0];
            } else if (KIND == 5) {
                f32x4 v = {acc, v0, v1, v2};
                gbuf[(size_t)blockIdx.x * 4096 + ((t + i * 256) & 4095)] = v;
            } else {
                f32x4 v = {acc, v0, v1, v2};
                reinterpret_cast<f32x4*>(lds)[(t + i * 7) & 1023] = v;
            }
        }
        out[blockIdx.x * 256 + t] = v0 + w0 + acc;
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((t & 63) == 0) cyc[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
        if (t == 0) hw[blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xFF00);
    }
}

template <int KIND>
void run(const char* name) {
    const int blocks = 512, iters = 4000;
    float* out; unsigned long long *cyc, *hw; f32x4* gbuf;
    hipMalloc(&gbuf, (size_t)blocks * 4096 * 16);
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8); hipMalloc(&hw, blocks * 8);
    hipMemset(cyc, 0, blocks * 4 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, hw, iters, gbuf);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) { sum += h[b * 4 + w]; ++n; }
    std::vector<unsigned long long> hh(blocks);
    hipMemcpy(hh.data(), hw, blocks * 8, hipMemcpyDeviceToHost);
    int shared = 0;
    for (int b = 0; b < 256; ++b) for (int c = 256; c < 512; ++c) if (hh[b] == hh[c]) { ++shared; break; }
    double vs = 0; int vn = 0;
    for (int b = 256; b < 512; ++b) for (int w = 0; w < 4; ++w) { vs += h[b * 4 + w]; ++vn; }
    printf("%-32s : %.1f cycles per MFMA ; partner loop %.1f cycles per iteration ; MFMA blocks with a partner on their CU: %d/256\n",
           name, sum / n / (iters * 2.0), KIND ? vs / vn / (iters * 8.0) : 0.0, shared);
    hipFree(out); hipFree(cyc); hipFree(gbuf);
}

int main() {
    run<0>("partner idle");
    run<1>("partner v_fma_f32 stream");
    run<2>("partner integer VALU stream");
    run<3>("partner LDS read stream");
    run<4>("partner global loads 16B/lane (1/iter)");
    run<5>("partner global stores 16B/lane (1/iter)");
    run<6>("partner ds_write_b128 (1/iter)");
    return 0;
}
