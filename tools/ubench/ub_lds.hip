// How many bytes per clock does a CU's LDS deliver to ds_read_b128 -- alone, and beside v_mfma_f32_32x32x16_f16?
// Question behind it: the fp16 conv kernels read 1.0-1.5 KiB of fragments per MFMA; are they LDS-bound?
//   mode 0: every wave issues conflict-free lane-linear ds_read_b128 back to back (8 in flight), no MFMA
//   mode 1: the conv kernel's access pattern (A fragment: 144-B pixel stride; B fragment: lane-linear), no MFMA
//   mode 2: R reads per MFMA pair, R = 3 (32 px x 64 ch wave tile) / 2 (64x64) / 1.5 (128x64), MFMAs consuming the data
// Reported: bytes per shader clock per CU (s_memtime around the loop, median over blocks) and cycles per MFMA in mode 2.
// One 256-thread block per CU (1 wave / SIMD) or 512 threads (2 waves / SIMD).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE, int RPM2>     // RPM2: ds_read_b128 per 2 MFMAs, times 2 (6 = 3 reads per MFMA pair, 4 = 2, 3 = 1.5)
__global__ __launch_bounds__(512) void k(unsigned long long* stamps, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 96 * 1024 / 16; i += blockDim.x) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const int m = lane & 31, h = lane >> 5;
    const char* lin = smem + lane * 16 + (wave & 3) * 1024;
    const char* apat = smem + ((2 * (wave & 3) + (m >> 4)) * 2816 + (m & 15) * 144 + 16 * h);
    f32x4 s = {0, 0, 0, 0};
    f32x16 acc0 = {0}, acc1 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE < 2) {
        for (int it = 0; it < iters; ++it) {
            const int o = (it & 7) * 8192;
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = *reinterpret_cast<const f32x4*>((MODE == 0 || (j % 3)) ? lin + o + j * 1024 : apat + (j & 3) * 32 + (o & 0x3fff));
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
    } else {
        // per iteration: 4 MFMAs (2 pairs) and RPM2 reads
        for (int it = 0; it < iters; ++it) {
            const int o = (it & 7) * 8192;
            h8 f[6];
#pragma unroll
            for (int j = 0; j < RPM2; ++j)
                f[j] = *reinterpret_cast<const h8*>((j % 3) ? lin + o + j * 1024 : apat + (j & 3) * 32 + (o & 0x3fff));
#pragma unroll
            for (int j = RPM2; j < 6; ++j) f[j] = f[j % RPM2];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], f[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], f[2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], f[4], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], f[5], acc1, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) s[0] += acc0[r] + acc1[r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + t] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int RPM2>
void run(const char* name, int threads, unsigned long long* stamps, float* out) {
    const int blocks = 256, iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, RPM2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, RPM2>), dim3(blocks), dim3(threads), 96 * 1024, 0, stamps, out, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    const int waves = threads / 64;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) cyc.push_back((double)h[b * 8 + w]);
    std::sort(cyc.begin(), cyc.end());
    const double med = cyc[cyc.size() / 2];
    const double reads = (MODE < 2 ? 8.0 : (double)RPM2) * iters * waves;         // wave-instructions of 1 KiB per block
    printf("%-58s %d waves/CU: %8.0f cycles  %6.1f B/clk/CU", name, waves, med, reads * 1024.0 / med);
    if (MODE == 2) printf("   %5.1f cycles per MFMA per SIMD", med / (4.0 * iters * (waves / 4)));
    printf("\n");
}

int main() {
    unsigned long long* stamps;
    float* out;
    hipMalloc(&stamps, 256 * 8 * 8);
    hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<0, 6>("ds_read_b128 lane-linear, no MFMA", threads, stamps, out);
        run<1, 6>("ds_read_b128 conv pattern (1/3 A 144-B stride, 2/3 B linear)", threads, stamps, out);
        run<2, 6>("MFMA f16 32x32x16 + 3 reads / 2 MFMA (32x64 wave tile)", threads, stamps, out);
        run<2, 4>("MFMA f16 32x32x16 + 2 reads / 2 MFMA (64x64 wave tile)", threads, stamps, out);
        run<2, 3>("MFMA f16 32x32x16 + 1.5 reads / 2 MFMA (128x64 wave tile)", threads, stamps, out);
    }
    return 0;
}
