// Minimal repro, second hypothesis (the one the DCN diagnostics point to: only accumulator registers 8..15 of a 16-register MFMA
// destination were wrong, i.e. MFMA rows 16..31, and only where the compiler copied accumulators with v_mov_b64 from the TOP
// register pair downwards right behind the MFMAs):
//
//   how many wait states after v_mfma_f32_32x32x16_f16 until a VALU instruction may READ the destination registers -- the
//   lowest one, a middle one, the highest one -- and how does that change when NQ independent MFMAs were issued right before it?
//
// hipcc (ROCm 7.2) separates an XDL write from a VALU read of it by the 8-pass figure (s_nop 9 + the instructions in between =
// 11-12 wait states in the DCN kernel).  Here the whole sequence is ONE asm block with fixed registers, so neither the compiler's
// scheduler nor its hazard recognizer can add anything: D is exactly the number of s_nop 0 between the last MFMA and the reads.
// The accumulator starts at 1000.0 in every register: a register read before the MFMA wrote it returns exactly 1000.0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NQ, int D, int F = 0>
__global__ __launch_bounds__(64) void k(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const h8 av = *reinterpret_cast<const h8*>(a + lane * 8);
    const h8 bv = *reinterpret_cast<const h8*>(b + lane * 8);
    const float init = 1000.f;
    float hi, mid, lo;
    asm volatile(
        ".irp r,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103\n\t"
        "v_mov_b32 v\\r, %5\n\t"
        ".endr\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        ".if %c6 > 3\n\tv_mfma_f32_32x32x16_f16 v[88:103], %3, %4, v[88:103]\n\t.endif\n\t"
        ".if %c6 > 2\n\tv_mfma_f32_32x32x16_f16 v[72:87], %3, %4, v[72:87]\n\t.endif\n\t"
        ".if %c6 > 1\n\tv_mfma_f32_32x32x16_f16 v[56:71], %3, %4, v[56:71]\n\t.endif\n\t"
        "v_mfma_f32_32x32x16_f16 v[40:55], %3, %4, v[40:55]\n\t"
        ".if %c8 == 1 || %c8 == 3\n\ts_waitcnt lgkmcnt(0)\n\t.endif\n\t"
        ".if %c8 == 2 || %c8 == 3\n\tv_mfma_f32_32x32x16_f16 v[56:71], %3, %4, v[56:71]\n\t.endif\n\t"
        ".if %c8 == 4\n\ts_mov_b32 s40, 0\n\t.endif\n\t"
        ".if %c8 == 5\n\tv_mov_b32 v72, v73\n\t.endif\n\t"
        ".rept %c7\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b32 %0, v55\n\t"
        "v_mov_b32 %1, v48\n\t"
        "v_mov_b32 %2, v40\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        : "=&v"(hi), "=&v"(mid), "=&v"(lo)
        : "v"(av), "v"(bv), "v"(init), "n"(NQ), "n"(D), "n"(F)
        : "memory", "s40", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
          "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",
          "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",
          "v99", "v100", "v101", "v102", "v103");
    out[(blockIdx.x * 3 + 0) * 64 + lane] = hi;
    out[(blockIdx.x * 3 + 1) * 64 + lane] = mid;
    out[(blockIdx.x * 3 + 2) * 64 + lane] = lo;
}

static std::vector<float> ref;
template <int NQ, int D, int F = 0>
void run(const _Float16* a, const _Float16* b, float* out) {
    const int blocks = 1024;
    long stale[3] = {0, 0, 0}, wrong[3] = {0, 0, 0};
    for (int rep = 0; rep < 50; ++rep) {
        hipLaunchKernelGGL((k<NQ, D, F>), dim3(blocks), dim3(64), 0, 0, a, b, out);
        std::vector<float> h((size_t)blocks * 3 * 64);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        if (ref.empty()) ref.assign(h.begin(), h.begin() + 3 * 64);
        for (int blk = 0; blk < blocks; ++blk)
            for (int w = 0; w < 3; ++w)
                for (int l = 0; l < 64; ++l) {
                    const float v = h[((size_t)blk * 3 + w) * 64 + l];
                    if (v != ref[w * 64 + l]) {
                        ++wrong[w];
                        if (v == 1000.f) ++stale[w];
                    }
                }
    }
    static const char* fill[] = {"", " + s_waitcnt lgkmcnt(0)", " + an independent MFMA", " + s_waitcnt + an independent MFMA (the DCN kernel's sequence)",
                                 " + one SALU instruction", " + one independent VALU instruction"};
    if (F) printf("after the MFMA:%s, then  ", fill[F]);
    printf("MFMAs issued back to back %d, s_nop 0 between the last one and the reads %3d:  wrong reads of dst[15] %8ld (stale 1000.0: %8ld)   dst[8] %8ld (%8ld)   dst[0] %8ld (%8ld)   of %d each\n",
           NQ, D, wrong[0], stale[0], wrong[1], stale[1], wrong[2], stale[2], 50 * blocks * 64);
}

int main() {
    std::vector<_Float16> ha(64 * 8), hb(64 * 8);
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 9) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& v : ha) v = (_Float16)rnd();
    for (auto& v : hb) v = (_Float16)rnd();
    _Float16 *a, *b;
    float* out;
    hipMalloc(&a, ha.size() * 2); hipMalloc(&b, hb.size() * 2); hipMalloc(&out, (size_t)1024 * 3 * 64 * 4);
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    run<1, 64>(a, b, out);          // reference: a long wait (first call fills `ref`)
    run<1, 0>(a, b, out);  run<1, 2>(a, b, out);  run<1, 4>(a, b, out);  run<1, 6>(a, b, out);  run<1, 8>(a, b, out);  run<1, 10>(a, b, out);
    run<1, 11>(a, b, out); run<1, 12>(a, b, out); run<1, 14>(a, b, out); run<1, 16>(a, b, out); run<1, 18>(a, b, out); run<1, 20>(a, b, out);
    run<2, 8>(a, b, out);  run<2, 11>(a, b, out); run<2, 12>(a, b, out); run<2, 16>(a, b, out); run<2, 20>(a, b, out); run<2, 24>(a, b, out);
    run<4, 8>(a, b, out);  run<4, 11>(a, b, out); run<4, 12>(a, b, out); run<4, 16>(a, b, out); run<4, 20>(a, b, out); run<4, 24>(a, b, out);
    run<4, 28>(a, b, out); run<4, 32>(a, b, out); run<4, 40>(a, b, out);
    // what does each kind of instruction between the MFMA and the read count for?  hipcc's hazard recognizer counts every one of them
    // as one wait state and tops up with s_nop to 12 in total.
    run<1, 11, 1>(a, b, out); run<1, 10, 1>(a, b, out);
    run<1, 11, 2>(a, b, out); run<1, 10, 2>(a, b, out); run<1, 9, 2>(a, b, out);  run<1, 4, 2>(a, b, out);
    run<1, 10, 3>(a, b, out); run<1, 9, 3>(a, b, out);
    run<1, 11, 4>(a, b, out); run<1, 10, 4>(a, b, out);
    run<1, 11, 5>(a, b, out); run<1, 10, 5>(a, b, out);
    return 0;
}
