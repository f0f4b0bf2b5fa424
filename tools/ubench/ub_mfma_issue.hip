// How fast does ONE wave per SIMD issue v_mfma_f32_32x32x16_f16 -- alone, with NACC independent accumulator chains, and beside R
// software-pipelined ds_read_b128 per MFMA (data consumed one iteration later, counted waits)?  Question behind it (DESIGN.md 3.4 /
// 3.6): every fp16 K loop of this repo runs 60-70 cycles per MFMA and wave whatever its prefetch depth; two waves per SIMD reach 32.
//   hipcc --offload-arch=gfx950 -O3 ub_mfma_issue.hip -o ub_mfma_issue && ./ub_mfma_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NACC, int READS>     // READS: ds_read_b128 per group of 4 MFMAs (0, 2, 4, 6)
__global__ __launch_bounds__(512) void k(unsigned long long* stamps, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 64 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1e-3f * (i & 255);
    __syncthreads();
    const char* lin = smem + lane * 16 + (wave & 3) * 8192;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h8 cur[6], nxt[6];
    for (int j = 0; j < 6; ++j) cur[j] = *reinterpret_cast<const h8*>(lin + j * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int o = ((it + 1) & 3) * 1024 * 8 * 0 + ((it + 1) & 7) * 16;     // stays inside the wave's 8 KiB
#pragma unroll
        for (int j = 0; j < READS; ++j) nxt[j] = *reinterpret_cast<const h8*>(lin + (j * 1024 + o) % 8192);
        __builtin_amdgcn_sched_barrier(0);
        acc[0 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[1], acc[0 % NACC], 0, 0, 0);
        acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[2 % 6], acc[1 % NACC], 0, 0, 0);
        acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[3], cur[4], acc[2 % NACC], 0, 0, 0);
        acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[3], cur[5], acc[3 % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < READS; ++j) cur[j] = nxt[j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + t] = s;
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

// the same loop with the reads DEALT into the MFMA gaps (sched_group_barrier: 1 MFMA, then READS/4 or so ds_reads, ...) instead of
// issued as one burst in front of the four MFMAs
template <int NACC, int READS>
__global__ __launch_bounds__(512) void k_il(unsigned long long* stamps, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 64 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1e-3f * (i & 255);
    __syncthreads();
    const char* lin = smem + lane * 16 + (wave & 3) * 8192;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h8 bufA[6], bufB[6];
    for (int j = 0; j < 6; ++j) bufA[j] = bufB[j] = *reinterpret_cast<const h8*>(lin + j * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define DEAL(g)                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
        if constexpr ((READS * (g + 1)) / 4 - (READS * g) / 4 == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
        if constexpr ((READS * (g + 1)) / 4 - (READS * g) / 4 == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#define STEP(cur, nxt, off)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        _Pragma("unroll") for (int j = 0; j < READS; ++j) nxt[j] = *reinterpret_cast<const h8*>(lin + j * 1024 + off); \
        acc[0 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[1], acc[0 % NACC], 0, 0, 0);   \
        acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[0], cur[2 % 6], acc[1 % NACC], 0, 0, 0); \
        acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[3], cur[4], acc[2 % NACC], 0, 0, 0);   \
        acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[3], cur[5], acc[3 % NACC], 0, 0, 0);   \
        DEAL(0) DEAL(1) DEAL(2) DEAL(3)                                                               \
        __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it += 2) {
        STEP(bufA, bufB, 16)
        STEP(bufB, bufA, 32)
    }
#undef STEP
#undef DEAL
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + t] = s;
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NACC, int READS, bool IL = false>
void run(int threads, unsigned long long* stamps, float* out) {
    const int blocks = 256, iters = 4000;
    void (*kern)(unsigned long long*, float*, int) = k<NACC, READS>;
    if (IL) kern = k_il<NACC, READS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 64 * 1024, 0, stamps, out, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    const int waves = threads / 64;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) cyc.push_back((double)h[b * 8 + w]);
    std::sort(cyc.begin(), cyc.end());
    const double med = cyc[cyc.size() / 2];
    printf("%s%d accumulator chains, %d ds_read_b128 per 4 MFMAs, %d wave(s) per SIMD: %6.1f cycles per MFMA and wave, %6.1f per MFMA and SIMD\n",
           IL ? "[reads dealt into the MFMA gaps] " : "", NACC, READS, waves / 4, med / (4.0 * iters), med / (4.0 * iters * (waves / 4)));
}

int main() {
    unsigned long long* stamps;
    float* out;
    hipMalloc(&stamps, 256 * 8 * 8);
    hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<4, 0>(threads, stamps, out);
        run<2, 0>(threads, stamps, out);
        run<1, 0>(threads, stamps, out);
        run<4, 2>(threads, stamps, out);
        run<4, 4>(threads, stamps, out);
        run<4, 6>(threads, stamps, out);
        run<2, 6>(threads, stamps, out);
        run<4, 2, true>(threads, stamps, out);
        run<4, 4, true>(threads, stamps, out);
        run<4, 6, true>(threads, stamps, out);
        run<2, 6, true>(threads, stamps, out);
    }
    return 0;
}
