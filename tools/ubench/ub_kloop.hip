// Micro-benchmark mirroring the conv kernel's K loop: per chunk 8 q-steps x (3 ds_read_b128 prefetched one
// q-step ahead + 8 MFMA 32x32x2 f32).  Variants add the per-chunk barrier, the B-image global load +
// ds_write, and the swizzled-address VALU.  Prints cycles per MFMA (ideal 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// F bit0: barrier per chunk; bit1: global load + ds_write of next B chunk; bit2: conv-style A address math
template <int F>
__global__ __launch_bounds__(256, 2) void k(const f32x4* __restrict__ w, float* out, unsigned long long* cyc, int chunks) {
    extern __shared__ f32x4 lds[];      // 2880 (A) + 2*1024 (B)
    f32x4* sA = lds;
    f32x4* sB = lds + 2880;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 2880 + 2048; i += 256) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f} * 1e-3f;
    __syncthreads();
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15;
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4 breg[4];
    int cbuf = 0;
    f32x4 av = sA[lane], b0 = sB[lane], b1 = sB[64 + lane];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        const int tap = c % 9;
        const int dy = tap / 3, dx = tap - dy * 3;
        if (F & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) breg[i] = w[(size_t)((c + 1) % 9) * 1024 + t + 256 * i];
        }
        const f32x4* bb = sB + cbuf * 1024 + lane;
        const f32x4* bnb = sB + (cbuf ^ 1) * 1024 + lane;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4 an, bn0, bn1;
            if (q < 7) {
                if (F & 4) an = sA[((2 * wave + my + dy) * 18 + mx + dx) * 16 + ((2 * (q + 1) + h) ^ ((mx + dx) & 15))];
                else an = sA[(lane + (q + 1) * 64) & 2047];
                bn0 = bb[((q + 1) * 2) * 64];
                bn1 = bb[((q + 1) * 2 + 1) * 64];
            } else {
                if (F & 1) __syncthreads();
                if (F & 4) an = sA[((2 * wave + my + dy) * 18 + mx + dx) * 16 + (h ^ ((mx + dx) & 15))];
                else an = sA[lane];
                bn0 = bnb[0];
                bn1 = bnb[64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], acc1, 0, 0, 0);
            }
            if (q == 6 && (F & 2)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sB[(cbuf ^ 1) * 1024 + t + 256 * i] = breg[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            av = an; b0 = bn0; b1 = bn1;
        }
        cbuf ^= 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + t] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int F>
void run(const char* name, int blocks) {
    float* out; unsigned long long* cyc; f32x4* w;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8); hipMalloc(&w, 9 * 1024 * 16);
    hipMemset(w, 0, 9 * 1024 * 16);
    const int chunks = 90;
    const int lds = (2880 + 2048) * 16;
    hipFuncSetAttribute((const void*)k<F>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256), lds, 0, w, out, cyc, chunks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += v;
    printf("%-60s blocks %4d : %.1f cycles per MFMA per wave\n", name, blocks, sum / h.size() / (chunks * 64.0));
    hipFree(out); hipFree(cyc); hipFree(w);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0>("reads prefetched, no barrier", blocks);
        run<4>("+ conv A address math", blocks);
        run<1>("+ barrier per chunk", blocks);
        run<3>("+ barrier + B global load/ds_write", blocks);
        run<7>("+ barrier + B load/store + A address math (= conv K loop)", blocks);
    }
    return 0;
}
