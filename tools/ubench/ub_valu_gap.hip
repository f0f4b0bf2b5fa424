// What does vector-ALU work between fp32 MFMAs cost, and how should it be grouped?  One wave per SIMD (the Winograd kernel's
// occupancy), a stream of independent v_mfma_f32_16x16x4_f32 (16 accumulators), and in the gaps:
//   0 nothing | 1: one v_add_f32 per gap | 2: four per gap | 3: eight every 2nd gap | 4: sixteen every 4th gap | 5: 64 every 16th gap
//   6: two v_pk_add_f32 per gap (the same 4 adds) | 7: four v_pk_add_f32 every 2nd gap | 8: four dependent v_add_f32 per gap
// Reports cycles per MFMA and checks the adds' results (packed ops beside MFMAs gave wrong values in three kernels of this repo).
//   hipcc -O3 --offload-arch=gfx950 -o ub_valu_gap ub_valu_gap.hip && ./ub_valu_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(i) "v_mfma_f32_16x16x4_f32 %" #i ", %16, %17, %" #i "\n\t"

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters, const float* gsrc) {
    extern __shared__ char smem[];
    const int t = threadIdx.x;
    f32x4 c[16];
    for (int i = 0; i < 16; ++i) c[i] = f32x4{0, 0, 0, 0};
    float a = 1.0f + (t & 15) * 0.125f, b = 0.5f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)(t + i);
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f32x2{(float)(t + 2 * i), (float)(t + 2 * i + 1)};
    const float one = 1.0f;
    f32x4 ld[4] = {};
    const unsigned laddr = (unsigned)(t * 16);
    const unsigned gaddr = (unsigned)(t * 16);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, 1 << 20, 0x00020000);
    const f32x2 one2 = {1.0f, 1.0f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define GAP_V4(r0, r1, r2, r3) asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(v[r0]), "+v"(v[r1]), "+v"(v[r2]), "+v"(v[r3]) : "v"(one));
#define GAP_V1(r0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r0]) : "v"(one));
#define GAP_D4(r0) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(v[r0]) : "v"(one));
#define GAP_P2(r0, r1) asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2" : "+v"(p[r0]), "+v"(p[r1]) : "v"(one2));
#define ONE_MFMA(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ONE_MFMA(i)
            if (KIND == 1) { GAP_V1(i) }
            if (KIND == 2) { GAP_V4((4 * i) & 15, (4 * i + 1) & 15, (4 * i + 2) & 15, (4 * i + 3) & 15) }
            if (KIND == 3 && (i & 1)) { GAP_V4(0, 1, 2, 3) GAP_V4(4, 5, 6, 7) }
            if (KIND == 4 && (i & 3) == 3) { GAP_V4(0, 1, 2, 3) GAP_V4(4, 5, 6, 7) GAP_V4(8, 9, 10, 11) GAP_V4(12, 13, 14, 15) }
            if (KIND == 5 && i == 15) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { GAP_V4(0, 1, 2, 3) GAP_V4(4, 5, 6, 7) GAP_V4(8, 9, 10, 11) GAP_V4(12, 13, 14, 15) }
            }
            if (KIND == 6) { GAP_P2((2 * i) & 7, (2 * i + 1) & 7) }
            if (KIND == 7 && (i & 1)) { GAP_P2(0, 1) GAP_P2(2, 3) }
            if (KIND == 8) { GAP_D4(i) }
            if (KIND == 9) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[i & 3]) : "v"(laddr));
            if (KIND == 10 && (i & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(i >> 2) & 3]) : "v"(laddr));
            if (KIND == 11) asm volatile("s_nop 0");
            if (KIND == 12) asm volatile("s_waitcnt lgkmcnt(8)");
            if (KIND == 13 && (i & 7) == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[(i >> 3) & 1]) : "v"(gaddr), "s"(rs));
            if (KIND == 14 && (i & 7) == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(laddr), "v"(ld[0]));
            if (KIND == 15 && i == 15) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier");
            if (KIND == 16) { asm volatile("ds_read_b128 %0, %1" : "=v"(ld[i & 3]) : "v"(laddr)); if ((i & 3) == 3) asm volatile("s_waitcnt lgkmcnt(2)"); }
            if (KIND == 17 && (i & 3) == 0) asm volatile("v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %5\n\tv_add_f32 %2, %2, %5\n\tv_add_f32 %3, %3, %5\n\tds_read_b128 %4, %6" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=v"(ld[i >> 2]) : "v"(one), "v"(laddr));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += ld[i][0] * 1e-30f;
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    float sv = 0;
    for (int i = 0; i < 16; ++i) sv += v[i] - (float)(t + i);
    for (int i = 0; i < 8; ++i) sv += (p[i][0] - (float)(t + 2 * i)) + (p[i][1] - (float)(t + 2 * i + 1));
    out[(blockIdx.x * 256 + t) * 2] = s;
    out[(blockIdx.x * 256 + t) * 2 + 1] = sv;      // = number of adds performed per lane (exact while < 2^24)
    if ((t & 63) == 0) cyc[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int adds_per_16, float* out, unsigned long long* cyc, int iters) {
    static float* gsrc = nullptr;
    if (!gsrc) { (void)hipMalloc(&gsrc, 1 << 20); (void)hipMemset(gsrc, 0, 1 << 20); }
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    k<KIND><<<256, 256, 150 * 1024>>>(out, cyc, iters, gsrc);
    k<KIND><<<256, 256, 150 * 1024>>>(out, cyc, iters, gsrc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    std::vector<float> o(256 * 256 * 2);
    hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto x : h) sum += (double)x;
    const double per = sum / 1024 / ((double)iters * 16);
    // s_memtime ticks at 100 MHz on this part?  report ticks per MFMA and the ratio to KIND 0 (printed by the caller)
    long bad = 0;
    const float want = (float)((double)adds_per_16 * iters);
    for (size_t i = 0; i < o.size() / 2; ++i)
        if (o[2 * i + 1] != want) ++bad;
    printf("%-44s ticks/MFMA %8.4f   adds/lane %g (want %g)  wrong lanes %ld\n", name, per, (double)o[1], (double)want, bad);
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 2 * 4);
    hipMalloc(&cyc, 1024 * 8);
    const int iters = 4096;
    run<0>("0 MFMAs alone", 0, out, cyc, iters);
    run<1>("1 one v_add per gap", 16, out, cyc, iters);
    run<2>("2 four v_add per gap", 64, out, cyc, iters);
    run<3>("3 eight v_add every 2nd gap", 64, out, cyc, iters);
    run<4>("4 sixteen v_add every 4th gap", 64, out, cyc, iters);
    run<5>("5 sixty-four v_add every 16th gap", 64, out, cyc, iters);
    run<6>("6 two v_pk_add per gap (4 adds)", 64, out, cyc, iters);
    run<7>("7 four v_pk_add every 2nd gap (4 adds/gap)", 64, out, cyc, iters);
    run<8>("8 four DEPENDENT v_add per gap", 64, out, cyc, iters);
    run<9>("9 one ds_read_b128 per gap", 0, out, cyc, iters);
    run<10>("10 one ds_read_b128 every 4th gap", 0, out, cyc, iters);
    run<11>("11 s_nop 0 per gap", 0, out, cyc, iters);
    run<12>("12 satisfied s_waitcnt per gap", 0, out, cyc, iters);
    run<13>("13 buffer_load_dwordx4 (L2 hit) every 8th gap", 0, out, cyc, iters);
    run<14>("14 ds_write_b128 every 8th gap", 0, out, cyc, iters);
    run<15>("15 lgkmcnt(0) + s_barrier every 16th gap", 0, out, cyc, iters);
    run<16>("16 ds_read per gap + counted wait every 4th", 0, out, cyc, iters);
    run<17>("17 four v_add + ds_read in every 4th gap", 16, out, cyc, iters);
    return 0;
}
