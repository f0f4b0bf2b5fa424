// Minimal repro for the hazard behind csrc/dcn.hip's fp16 note (found in round 2, root-caused in round 3):
//
//   On gfx950 (MI355X, ROCm 7.2) a v_mfma_f32_32x32x16_f16 that is still in flight when EXEC is narrowed produces wrong
//   results for the MFMA rows of lanes 16..31 / 48..63 of the A operand (the half of the 32x32 tile the later passes compute).
//   hipcc's hazard recognizer puts no wait states between an MFMA and a following EXEC write (s_and_saveexec / s_and_b64 exec);
//   any divergent branch placed right behind a queue of MFMAs can therefore corrupt them -- non-deterministically, because
//   whether the MFMA has drained depends on what else the SIMD is doing.
//
// The kernel issues NQ back-to-back MFMAs (independent accumulators, so they queue in the matrix pipe), then D wait states of
// s_nop, then narrows EXEC to the even lanes for one VALU instruction and restores it.  Every accumulator is compared with a
// run that waits 64 x 16 wait states before touching EXEC.  Reported per (NQ, D): wrong elements, and which MFMA rows they
// are in.  Expectation if the hypothesis holds: errors for small D, growing with NQ, none from some D on (~ 8 passes per queued
// MFMA), all of them in rows 16..31.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NQ, int D, bool NARROW>
__global__ __launch_bounds__(64) void k(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float* __restrict__ out, float* side) {
    const int lane = threadIdx.x;
    const h8 av = *reinterpret_cast<const h8*>(a + lane * 8);
    h8 bv[4];
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const h8*>(b + (j * 64 + lane) * 8);
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float x = (float)lane;
    unsigned long long saved;
    const unsigned long long mask = NARROW ? 0x5555555555555555ull : ~0ull;
    // one asm block: the compiler cannot schedule anything into it or add wait states inside it
    asm volatile(
        "v_mfma_f32_32x32x16_f16 %0, %5, %6, %0\n\t"
        ".if %c10 > 1\n\tv_mfma_f32_32x32x16_f16 %1, %5, %7, %1\n\t.endif\n\t"
        ".if %c10 > 2\n\tv_mfma_f32_32x32x16_f16 %2, %5, %8, %2\n\t.endif\n\t"
        ".if %c10 > 3\n\tv_mfma_f32_32x32x16_f16 %3, %5, %9, %3\n\t.endif\n\t"
        ".rept %c11\n\ts_nop 0\n\t.endr\n\t"
        "s_mov_b64 %4, exec\n\t"
        "s_and_b64 exec, exec, %12\n\t"
        "v_add_f32 %13, %13, %13\n\t"
        "s_mov_b64 exec, %4\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&s"(saved)
        : "v"(av), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "n"(NQ), "n"(D), "s"(mask), "v"(x)
        : "memory");
    for (int j = 0; j < NQ; ++j)
        for (int r = 0; r < 16; ++r) out[(j * 16 + r) * 64 + lane] = acc[j][r];
    side[lane] = x;
}

template <int NQ, int D>
void run(const _Float16* a, const _Float16* b, float* out, float* side, const std::vector<float>& ref) {
    int worst = 0, rows_lo = 0, rows_hi = 0;
    for (int rep = 0; rep < 200; ++rep) {
        hipMemsetAsync(out, 0, 4 * 16 * 64 * 4, 0);
        hipLaunchKernelGGL((k<NQ, D, true>), dim3(1024), dim3(64), 0, 0, a, b, out, side);   // 1024 waves: the chip is busy, timing varies
        std::vector<float> h(4 * 16 * 64);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int j = 0; j < NQ; ++j)
            for (int r = 0; r < 16; ++r)
                for (int l = 0; l < 64; ++l)
                    if (h[(j * 16 + r) * 64 + l] != ref[(j * 16 + r) * 64 + l]) {
                        ++bad;
                        // accumulator register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the 32x32 tile, column l & 31
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                        (row < 16 ? rows_lo : rows_hi)++;
                    }
        if (bad > worst) worst = bad;
    }
    printf("queued MFMAs %d, wait states before the EXEC write %3d: worst wrong elements per wave-output %5d of %d; over 200 launches wrong in rows 0..15: %d, rows 16..31: %d\n",
           NQ, D, worst, NQ * 1024, rows_lo, rows_hi);
}

int main() {
    std::vector<_Float16> ha(64 * 8), hb(4 * 64 * 8);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 9) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& v : ha) v = (_Float16)rnd();
    for (auto& v : hb) v = (_Float16)rnd();
    _Float16 *a, *b;
    float *out, *side;
    hipMalloc(&a, ha.size() * 2); hipMalloc(&b, hb.size() * 2); hipMalloc(&out, 4 * 16 * 64 * 4); hipMalloc(&side, 64 * 4);
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    // reference: the same MFMAs with EXEC left alone (NARROW = false) and a long wait
    hipLaunchKernelGGL((k<4, 64, false>), dim3(1), dim3(64), 0, 0, a, b, out, side);
    std::vector<float> ref(4 * 16 * 64);
    hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0;
    for (float v : ref) cs += v;
    printf("reference checksum %.6f (EXEC untouched)\n", cs);
    run<1, 0>(a, b, out, side, ref);  run<1, 2>(a, b, out, side, ref);  run<1, 4>(a, b, out, side, ref);  run<1, 8>(a, b, out, side, ref);
    run<1, 12>(a, b, out, side, ref); run<1, 16>(a, b, out, side, ref);
    run<2, 0>(a, b, out, side, ref);  run<2, 8>(a, b, out, side, ref);  run<2, 16>(a, b, out, side, ref); run<2, 24>(a, b, out, side, ref);
    run<4, 0>(a, b, out, side, ref);  run<4, 8>(a, b, out, side, ref);  run<4, 16>(a, b, out, side, ref); run<4, 24>(a, b, out, side, ref);
    run<4, 32>(a, b, out, side, ref); run<4, 40>(a, b, out, side, ref); run<4, 48>(a, b, out, side, ref);
    return 0;
}
