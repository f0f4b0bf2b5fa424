// Minimal repro, third question: after v_mfma_f32_32x32x16_f16 has ISSUED, how soon may a VALU instruction OVERWRITE one of its
// source operands (SrcA, SrcB, or SrcC when C != D)?  hipcc treats SrcA / SrcB as read at issue (0 wait states) and SrcC with the
// XDL-read-SrcC figure.  One asm block, fixed registers, D = number of s_nop 0 between the MFMA and the overwriting v_mov;
// NQ independent MFMAs in front of it so that the probed one waits in the queue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// WHAT: 0 SrcA (v[8:11]), 1 SrcB (v[12:15]), 2 SrcC (v[40:55], D = v[56:71])
template <int NQ, int D, int WHAT>
__global__ __launch_bounds__(64) void k(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const h8 av = *reinterpret_cast<const h8*>(a + lane * 8);
    const h8 bv = *reinterpret_cast<const h8*>(b + lane * 8);
    const float init = 3.f;
    asm volatile(
        "v_mov_b32 v8, %0\n\tv_mov_b32 v9, %1\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "v_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v14, %6\n\tv_mov_b32 v15, %7\n\t"
        ".irp r,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119\n\t"
        "v_mov_b32 v\\r, %8\n\t"
        ".endr\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        ".if %c9 > 2\n\tv_mfma_f32_32x32x16_f16 v[104:119], v[8:11], v[12:15], v[104:119]\n\t.endif\n\t"
        ".if %c9 > 1\n\tv_mfma_f32_32x32x16_f16 v[88:103], v[8:11], v[12:15], v[88:103]\n\t.endif\n\t"
        ".if %c9 > 0\n\tv_mfma_f32_32x32x16_f16 v[72:87], v[8:11], v[12:15], v[72:87]\n\t.endif\n\t"
        "v_mfma_f32_32x32x16_f16 v[56:71], v[8:11], v[12:15], v[40:55]\n\t"
        ".rept %c10\n\ts_nop 0\n\t.endr\n\t"
        ".if %c11 == 0\n\tv_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\t.endif\n\t"
        ".if %c11 == 1\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t.endif\n\t"
        ".if %c11 == 2\n\t"
        ".irp r,55,54,53,52,51,50,49,48,47,46,45,44,43,42,41,40\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t"
        ".endif\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        :
        : "v"(__builtin_bit_cast(int4, av).x), "v"(__builtin_bit_cast(int4, av).y), "v"(__builtin_bit_cast(int4, av).z), "v"(__builtin_bit_cast(int4, av).w),
          "v"(__builtin_bit_cast(int4, bv).x), "v"(__builtin_bit_cast(int4, bv).y), "v"(__builtin_bit_cast(int4, bv).z), "v"(__builtin_bit_cast(int4, bv).w),
          "v"(init), "n"(NQ), "n"(D), "n"(WHAT)
        : "memory", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",
          "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92",
          "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
          "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    // read D = v[56:71] back through a second asm (plain moves)
    float r[16];
    asm volatile(
        "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\tv_mov_b32 %4, v60\n\tv_mov_b32 %5, v61\n\tv_mov_b32 %6, v62\n\tv_mov_b32 %7, v63\n\t"
        "v_mov_b32 %8, v64\n\tv_mov_b32 %9, v65\n\tv_mov_b32 %10, v66\n\tv_mov_b32 %11, v67\n\tv_mov_b32 %12, v68\n\tv_mov_b32 %13, v69\n\tv_mov_b32 %14, v70\n\tv_mov_b32 %15, v71\n\t"
        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]),
          "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
        :
        : "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    for (int i = 0; i < 16; ++i) out[(blockIdx.x * 16 + i) * 64 + lane] = r[i];
}

static std::vector<float> ref;
template <int NQ, int D, int WHAT>
void run(const _Float16* a, const _Float16* b, float* out) {
    const int blocks = 512;
    long wrong_lo = 0, wrong_hi = 0;
    for (int rep = 0; rep < 30; ++rep) {
        hipLaunchKernelGGL((k<NQ, D, WHAT>), dim3(blocks), dim3(64), 0, 0, a, b, out);
        std::vector<float> h((size_t)blocks * 16 * 64);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        if (ref.empty()) ref.assign(h.begin(), h.begin() + 16 * 64);
        for (int blk = 0; blk < blocks; ++blk)
            for (int i = 0; i < 16; ++i)
                for (int l = 0; l < 64; ++l)
                    if (h[((size_t)blk * 16 + i) * 64 + l] != ref[i * 64 + l]) (i < 8 ? wrong_lo : wrong_hi)++;
    }
    static const char* what[] = {"SrcA", "SrcB", "SrcC (C != D)"};
    printf("%-14s overwritten %3d wait states after the MFMA issued, %d MFMAs queued in front: wrong results in dst[0..7] (rows 0..15) %9ld, dst[8..15] (rows 16..31) %9ld of %d each\n",
           what[WHAT], D, NQ, wrong_lo, wrong_hi, 30 * blocks * 8 * 64);
}

int main() {
    std::vector<_Float16> ha(64 * 8), hb(64 * 8);
    unsigned s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 9) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& v : ha) v = (_Float16)rnd();
    for (auto& v : hb) v = (_Float16)rnd();
    _Float16 *a, *b;
    float* out;
    hipMalloc(&a, ha.size() * 2); hipMalloc(&b, hb.size() * 2); hipMalloc(&out, (size_t)512 * 16 * 64 * 4);
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    run<0, 64, 0>(a, b, out);          // reference (first call fills `ref`): overwrite long after the MFMA
    run<0, 0, 0>(a, b, out); run<0, 1, 0>(a, b, out); run<0, 2, 0>(a, b, out); run<0, 4, 0>(a, b, out); run<0, 8, 0>(a, b, out);
    run<3, 0, 0>(a, b, out); run<3, 2, 0>(a, b, out); run<3, 4, 0>(a, b, out); run<3, 8, 0>(a, b, out);
    run<0, 0, 1>(a, b, out); run<0, 1, 1>(a, b, out); run<0, 2, 1>(a, b, out); run<0, 4, 1>(a, b, out); run<0, 8, 1>(a, b, out);
    run<3, 0, 1>(a, b, out); run<3, 2, 1>(a, b, out); run<3, 4, 1>(a, b, out); run<3, 8, 1>(a, b, out);
    run<0, 0, 2>(a, b, out); run<0, 2, 2>(a, b, out); run<0, 4, 2>(a, b, out); run<0, 6, 2>(a, b, out); run<0, 7, 2>(a, b, out); run<0, 8, 2>(a, b, out);
    run<0, 10, 2>(a, b, out); run<0, 12, 2>(a, b, out);
    run<3, 0, 2>(a, b, out); run<3, 4, 2>(a, b, out); run<3, 7, 2>(a, b, out); run<3, 8, 2>(a, b, out); run<3, 12, 2>(a, b, out);
    return 0;
}
