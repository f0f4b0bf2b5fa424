// Which fp16 MFMA shape sustains the higher WALL-CLOCK rate on MI355X in the split-fp16 K loop (conv_f16x3.hip)?
//   v_mfma_f32_32x32x16_f16 (32 cycles)  vs  v_mfma_f32_16x16x32_f16 (16 cycles): the same FLOP per cycle on paper, but the kernel is
// POWER-limited (r04 A/B: 10 % fewer cycles per tile came back as 5 % lower clock), and the clock the chip holds under load depends
// on the MFMA shape (MI355X_MICROARCH.md, DVFS give-back 7: bf16 16x16x32 ~1.12-1.15x the FLOP/s of 32x32x16 on random data).
// Both loops: a wave owns 32 pixels x 64 channels, operands split hi / lo, three MFMAs per product, every fragment re-read from LDS by
// ds_read_b128 dealt into the MFMA gaps (1 KiB of reads per 32 cycles of MFMA in both), random data, 2 blocks x 4 waves per CU,
// hipEvent wall time.      hipcc --offload-arch=gfx950 -O3 ub_shape16.hip -o ub_shape16 && ./ub_shape16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const _Float16* __restrict__ src, float* out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 64 * 1024 / 16; i += 256) reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(src)[i];
    __syncthreads();
    const char* lin = smem + lane * 16 + wave * 1024;
    float s = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 0) {
        // per 16-deep k-step: A hi, A lo, B hi N0/N1, B lo N0/N1 -> 6 MFMAs of 32 cycles
        f32x16 ah0 = {0}, ah1 = {0}, al0 = {0}, al1 = {0};
        h8 c[6], n[6];
        for (int j = 0; j < 6; ++j) c[j] = *reinterpret_cast<const h8*>(lin + j * 4096);
#define STEP(cur, nxt, off)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) nxt[j] = *reinterpret_cast<const h8*>(lin + ((j * 4096 + off) & 0xffff)); \
        ah0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[2], ah0, 0, 0, 0);                            \
        ah1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[3], ah1, 0, 0, 0);                            \
        al0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[1], cur[2], al0, 0, 0, 0);                            \
        al1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[1], cur[3], al1, 0, 0, 0);                            \
        al0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[4], al0, 0, 0, 0);                            \
        al1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[5], al1, 0, 0, 0);                            \
        _Pragma("unroll") for (int g = 0; g < 6; ++g) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);
        for (int it = 0; it < iters; it += 2) {
            STEP(c, n, 16 * 1024 + (it & 63) * 16)
            STEP(n, c, 32 * 1024 + (it & 63) * 16)
        }
#undef STEP
        for (int r = 0; r < 16; ++r) s += ah0[r] + ah1[r] + al0[r] + al1[r];
    } else {
        // per 32-deep k-step: A hi M0/M1, A lo M0/M1, B hi N0..3, B lo N0..3 -> 24 MFMAs of 16 cycles (the same 32 px x 64 ch tile)
        f32x4 ah[2][4], al[2][4];
        for (int m = 0; m < 2; ++m)
            for (int j = 0; j < 4; ++j) ah[m][j] = al[m][j] = (f32x4)(0.f);
        h8 c[12], n[12];
        for (int j = 0; j < 12; ++j) c[j] = *reinterpret_cast<const h8*>(lin + ((j * 4096) & 0xffff));
#define STEP(cur, nxt, off)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int j = 0; j < 12; ++j) nxt[j] = *reinterpret_cast<const h8*>(lin + ((j * 4096 + off) & 0xffff)); \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int j = 0; j < 4; ++j)              \
            ah[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[m], cur[4 + j], ah[m][j], 0, 0, 0);           \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int j = 0; j < 4; ++j)              \
            al[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[2 + m], cur[4 + j], al[m][j], 0, 0, 0);       \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int j = 0; j < 4; ++j)              \
            al[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[m], cur[8 + j], al[m][j], 0, 0, 0);           \
        _Pragma("unroll") for (int g = 0; g < 12; ++g) {                                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);
        for (int it = 0; it < iters; it += 4) {          // one STEP = two 16-deep k-steps of the other shape
            STEP(c, n, 16 * 1024 + (it & 63) * 16)
            STEP(n, c, 32 * 1024 + (it & 63) * 16)
        }
#undef STEP
        for (int m = 0; m < 2; ++m)
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 4; ++r) s += ah[m][j][r] + al[m][j][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + t] = s;
    if (t == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE>
void run(const _Float16* src, float* out, unsigned long long* stamps, const char* name) {
    const int blocks = 512, iters = 40000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 64 * 1024, 0, src, out, stamps, iters);   // ~1 s of load first
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 64 * 1024, 0, src, out, stamps, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) cyc += h[2 * b], real += h[2 * b + 1];
    // per 16-deep k-step and wave: 6 MFMAs of 32x32x16 = 6 * 32768 FLOP
    const double flop = (double)reps * blocks * 4 * iters * 6.0 * 32768.0;
    printf("%-28s %8.1f TFLOP/s (fp16 MFMA FLOPs, wall clock)   %6.1f cycles per 16-deep k-step and wave (ideal 2 waves/SIMD: 384)   in-kernel clock %.3f GHz\n",
           name, flop / (ms * 1e-3) / 1e12, cyc / blocks / iters, cyc / real / 10.0);
}

int main() {
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<_Float16> h(64 * 1024 / 2);
    for (auto& v : h) v = (_Float16)nd(rng);
    _Float16* src;
    float* out;
    unsigned long long* stamps;
    (void)hipMalloc(&src, 64 * 1024);
    (void)hipMalloc(&out, 512 * 256 * 4);
    (void)hipMalloc(&stamps, 512 * 16);
    (void)hipMemcpy(src, h.data(), 64 * 1024, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(src, out, stamps, "v_mfma_f32_32x32x16_f16");
        run<1>(src, out, stamps, "v_mfma_f32_16x16x32_f16");
    }
    return 0;
}
