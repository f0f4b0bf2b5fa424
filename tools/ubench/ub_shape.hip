// Which fp32 MFMA shape sustains the higher WALL-CLOCK rate on MI355X under a realistic operand feed?
//   v_mfma_f32_32x32x2_f32 (64 cycles)  vs  v_mfma_f32_16x16x4_f32 (32 cycles); same FLOP per cycle on paper,
// but the clock the chip holds under load can depend on the shape (MI355X_MICROARCH.md, DVFS give-back 7).
// Random operands, 2 blocks x 4 waves per CU, LDS-fed fragments (12 dwords per 32 KFLOP like the conv kernel),
// ~100 ms per variant, hipEvent wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters) {
    __shared__ f32x4 lds[3 * 1024];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 3 * 1024; i += 256) lds[i] = reinterpret_cast<const f32x4*>(src)[i];
    __syncthreads();
    float s = 0;
    if (SHAPE == 0) {
        f32x16 a0 = {0}, a1 = {0};
        f32x4 av = lds[lane], b0 = lds[1024 + lane], b1 = lds[2048 + lane];
        for (int it = 0; it < iters; ++it) {
            const int o = ((it + 1) * 64) & 1023;
            const f32x4 an = lds[o + lane], bn0 = lds[1024 + o + lane], bn1 = lds[2048 + o + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], a1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            av = an; b0 = bn0; b1 = bn1;
        }
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    } else {
        // same wave tile (32 px x 64 ch) and K (8) per iteration: 2 M tiles x 4 N tiles of 16x16, 2 k4-steps
        f32x4 acc[2][4];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        f32x4 av = lds[lane], b0 = lds[1024 + lane], b1 = lds[2048 + lane];
        for (int it = 0; it < iters; ++it) {
            const int o = ((it + 1) * 64) & 1023;
            const f32x4 an = lds[o + lane], bn0 = lds[1024 + o + lane], bn1 = lds[2048 + o + lane];
            __builtin_amdgcn_sched_barrier(0);
            // 12 operand dwords -> A: av[0..3] = (M tile 0, k step 0/1), (M tile 1, k step 0/1); B: b0,b1 = 4 N tiles x 2 k steps
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float bb = (j < 2) ? b0[j * 2 + ks] : b1[(j - 2) * 2 + ks];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i * 2 + ks], bb, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
            av = an; b0 = bn0; b1 = bn1;
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    out[blockIdx.x * 256 + t] = s;
}

template <int SHAPE>
void run(const char* name, const float* src, float* out) {
    const int blocks = 512;
    int iters = 20000;                     // 8 x 4096 (or 16 x 2048) MAC-pairs = 32768 FLOP... per wave-iteration: 65536 FLOP
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, src, out, 2000);
    hipDeviceSynchronize();
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 4; ++l) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 4.0 * blocks * 4 * (double)iters * 32768.0;   // 8 x 32x32x2 (or 16 x 16x16x4) MFMA = 32768 FLOP
        const double tf = flop / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
        printf("%-28s rep %d: %.1f ms  %.1f TFLOP/s\n", name, rep, ms, tf);
    }
}

int main() {
    std::vector<float> h(3 * 1024 * 4);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    for (auto& v : h) v = d(g);
    float *src, *out;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<0>("mfma_f32_32x32x2", src, out);
        run<1>("mfma_f32_16x16x4", src, out);
    }
    return 0;
}
