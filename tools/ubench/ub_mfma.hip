// Micro-benchmark: what limits one wave's v_mfma_f32_32x32x2_f32 stream on gfx950?
// Variants (template V): 0 bare MFMAs, 2 accumulators alternating
//                        1 + 3 ds_read_b128 per 8 MFMAs, prefetched one step ahead (sched_barrier pinned)
//                        2 + the VALU address arithmetic of the conv kernel
//                        3 = 1 but ONE accumulator chain (NT=1 style)
//                        4 = 0 with 4 accumulators
// build: hipcc --offload-arch=gfx950 -O3 ub_mfma.hip -o ub_mfma ; run: ./ub_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* cyc, int iters, int waves_active) {
    __shared__ f32x4 lds[4096];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 4096; i += 256) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f} * 1e-3f;
    __syncthreads();
    if ((t >> 6) >= waves_active) return;
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    f32x4 av = lds[lane], b0 = lds[64 + lane], b1 = lds[128 + lane];
    int idx = lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 an = av, bn0 = b0, bn1 = b1;
        if (V == 1 || V == 2 || V == 3) {
            int a_idx = idx;
            if (V == 2) a_idx = ((idx * 18 + it) & 1023) * 2 + (((it & 7) * 2 + (lane >> 5)) ^ (lane & 15)) % 2;
            an = lds[(a_idx + it * 64) & 4095];
            bn0 = lds[(idx + 64 + it * 128) & 4095];
            bn1 = lds[(idx + 128 + it * 128) & 4095];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (V == 3) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], acc0, 0, 0, 0);
            } else if (V == 4) {
                if (kk & 1) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], acc3, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], acc1, 0, 0, 0);
                }
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], acc1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        av = an; b0 = bn0; b1 = bn1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + t] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
}

template <int V>
void run(const char* name, int blocks, int waves_active) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    hipMemset(cyc, 0, blocks * 4 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, waves_active);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, waves_active);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (auto v : h) if (v) { sum += v; ++n; }
    printf("%-52s blocks %4d waves/blk %d : %.1f cycles per MFMA\n", name, blocks, waves_active, sum / n / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0>("bare, 2 accumulators", blocks, 4);
        run<4>("bare, 4 accumulators", blocks, 4);
        run<1>("+3 ds_read_b128 per 8 MFMA (prefetched)", blocks, 4);
        run<2>("+ds_reads +VALU address math", blocks, 4);
        run<3>("+3 ds_read, ONE accumulator chain", blocks, 4);
    }
    run<0>("bare, 2 acc, 1 wave per CU", 256, 1);
    return 0;
}
