// Does an MFMA still read its SrcA / SrcB registers when the NEXT vector-ALU instruction overwrites them?
//
// Same question as store_x4_then_wide_valu.hip, asked of the matrix instructions: hipcc treats SrcA / SrcB as read at issue (no wait
// state before a VALU write of them).  r03's mfma_war_hazard.hip asked it with ONE wave and `v_mov v8, v9, v10, v11` in that order --
// the two things that hid the store hazard (element 0 first; an idle SIMD).  Here: 131072 threads, the overwriting instruction
// aimed at the LAST / FIRST register of the operand, 32-bit, 64-bit and packed forms, 0-3 wait states, with and without MFMAs queued
// in front.  Operands are all ones, the overwriting value is NaN: any output that is not exactly K is a read of the new value.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_src_then_valu mfma_src_then_valu.hip && ./mfma_src_then_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W3 "s_nop 2\n\t"
#define Q0 ""
#define Q2_F16 "v_mfma_f32_32x32x16_f16 v[120:135], v[108:111], v[112:115], 0\n\tv_mfma_f32_32x32x16_f16 v[136:151], v[108:111], v[112:115], 0\n\t"
#define Q2_F32 "v_mfma_f32_16x16x4_f32 v[120:123], v108, v112, 0\n\tv_mfma_f32_16x16x4_f32 v[136:139], v108, v112, 0\n\t"
#define M_32x32x16_F16 "v_mfma_f32_32x32x16_f16 %[d], v[100:103], v[104:107], 0\n\t"
#define M_16x16x32_F16 "v_mfma_f32_16x16x32_f16 %[d], v[100:103], v[104:107], 0\n\t"
#define M_16x16x4_F32 "v_mfma_f32_16x16x4_f32 %[d], v100, v104, 0\n\t"
#define M_32x32x2_F32 "v_mfma_f32_32x32x2_f32 %[d], v100, v104, 0\n\t"
// the instruction behind the MFMA (v[116:117] hold NaN patterns)
#define C_A3 "v_mov_b32 v103, v116\n\t"
#define C_A0 "v_mov_b32 v100, v116\n\t"
#define C_A23_PK "v_pk_mul_f32 v[102:103], v[116:117], v[116:117]\n\t"
#define C_A23_M64 "v_mov_b64 v[102:103], v[116:117]\n\t"
#define C_A01_PK "v_pk_mul_f32 v[100:101], v[116:117], v[116:117]\n\t"
#define C_B3 "v_mov_b32 v107, v116\n\t"
#define C_B0 "v_mov_b32 v104, v116\n\t"
#define C_B23_PK "v_pk_mul_f32 v[106:107], v[116:117], v[116:117]\n\t"
#define C_NONE ""

#define PROBE(NAME, DT, ONE, EXPECT, QUEUE, MFMA, WAIT, CLOB)                                                                    \
    __global__ __launch_bounds__(256, 2) void NAME(unsigned* out, int iters) {                                                    \
        unsigned bad = 0;                                                                                                         \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            DT d;                                                                                                                 \
            const unsigned one = ONE, nan = 0x7fc07e00u + (unsigned)(it & 1) * 0;                                                  \
            asm volatile(".irp r,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115\n\tv_mov_b32 v\\r, %[one]\n\t.endr\n\t"  \
                         "v_mov_b32 v116, %[nan]\n\tv_mov_b32 v117, %[nan]\n\ts_nop 15\n\t" QUEUE MFMA WAIT CLOB "s_nop 15\n\ts_nop 15\n\t" \
                         : [d] "=&v"(d)                                                                                           \
                         : [one] "v"(one), [nan] "v"(nan)                                                                         \
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
                           "v114", "v115", "v116", "v117", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", \
                           "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", \
                           "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "memory");                             \
            bool ok = true;                                                                                                       \
            for (int k = 0; k < (int)(sizeof(DT) / 4); ++k) ok &= (d[k] == EXPECT);                                               \
            bad += !ok;                                                                                                           \
        }                                                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = bad;                                                                                \
    }
#define ALLW(BASE, DT, ONE, EXPECT, QUEUE, MFMA, CLOB)            \
    PROBE(BASE##_w0, DT, ONE, EXPECT, QUEUE, MFMA, W0, CLOB)      \
    PROBE(BASE##_w1, DT, ONE, EXPECT, QUEUE, MFMA, W1, CLOB)      \
    PROBE(BASE##_w2, DT, ONE, EXPECT, QUEUE, MFMA, W2, CLOB)      \
    PROBE(BASE##_w3, DT, ONE, EXPECT, QUEUE, MFMA, W3, CLOB)
#define H1 0x3c003c00u      /* two fp16 ones */
#define F1 0x3f800000u      /* one fp32 one */
ALLW(h32_none, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_NONE)
ALLW(h32_a3, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_A3)
ALLW(h32_a0, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_A0)
ALLW(h32_a23pk, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_A23_PK)
ALLW(h32_a23m64, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_A23_M64)
ALLW(h32_a01pk, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_A01_PK)
ALLW(h32_b3, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_B3)
ALLW(h32_b0, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_B0)
ALLW(h32_b23pk, f32x16, H1, 16.f, Q0, M_32x32x16_F16, C_B23_PK)
ALLW(h32q_a3, f32x16, H1, 16.f, Q2_F16, M_32x32x16_F16, C_A3)
ALLW(h32q_a23pk, f32x16, H1, 16.f, Q2_F16, M_32x32x16_F16, C_A23_PK)
ALLW(h32q_b3, f32x16, H1, 16.f, Q2_F16, M_32x32x16_F16, C_B3)
ALLW(h32q_b23pk, f32x16, H1, 16.f, Q2_F16, M_32x32x16_F16, C_B23_PK)
ALLW(h16_a3, f32x4, H1, 32.f, Q0, M_16x16x32_F16, C_A3)
ALLW(h16_a23pk, f32x4, H1, 32.f, Q0, M_16x16x32_F16, C_A23_PK)
ALLW(h16_b3, f32x4, H1, 32.f, Q0, M_16x16x32_F16, C_B3)
ALLW(h16q_a23pk, f32x4, H1, 32.f, Q2_F16, M_16x16x32_F16, C_A23_PK)
ALLW(h16q_b23pk, f32x4, H1, 32.f, Q2_F16, M_16x16x32_F16, C_B23_PK)
ALLW(f16_a0, f32x4, F1, 4.f, Q0, M_16x16x4_F32, C_A0)
ALLW(f16_a01pk, f32x4, F1, 4.f, Q0, M_16x16x4_F32, C_A01_PK)
ALLW(f16_b0, f32x4, F1, 4.f, Q0, M_16x16x4_F32, C_B0)
ALLW(f16q_a0, f32x4, F1, 4.f, Q2_F32, M_16x16x4_F32, C_A0)
ALLW(f16q_a01pk, f32x4, F1, 4.f, Q2_F32, M_16x16x4_F32, C_A01_PK)
ALLW(f16q_b0, f32x4, F1, 4.f, Q2_F32, M_16x16x4_F32, C_B0)
ALLW(f32_a0, f32x16, F1, 2.f, Q0, M_32x32x2_F32, C_A0)
ALLW(f32_b0, f32x16, F1, 2.f, Q0, M_32x32x2_F32, C_B0)

// The shape found in the fp16 DCN kernel (round 6, tools/repro/dcn_f16_listing_bisect.py: an s_nop behind every MFMA makes that kernel
// right): two MFMAs sharing SrcA, the second one waiting for its SrcB from LDS, then the next A fragment is converted INTO SrcA,
// last register first.
#define C_CVT4 "v_cvt_pk_f16_f32 v103, v116, v117\n\tv_cvt_pk_f16_f32 v102, v116, v117\n\tv_cvt_pk_f16_f32 v101, v116, v117\n\tv_cvt_pk_f16_f32 v100, v116, v117\n\t"
#define C_CVT1 "v_cvt_pk_f16_f32 v103, v116, v117\n\t"
#define PROBE_DCN(NAME, WAIT, CLOB, CHAIN)                                                                                       \
    __global__ __launch_bounds__(256, 2) void NAME(unsigned* out, int iters) {                                                    \
        __shared__ unsigned lds[4096];                                                                                            \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = H1;                                                                \
        __syncthreads();                                                                                                          \
        const unsigned la = (unsigned)(unsigned long long)(&lds[0]) + threadIdx.x * 16u;                                          \
        unsigned bad = 0;                                                                                                         \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            f32x16 d0, d1;                                                                                                        \
            const unsigned one = H1, nan = 0x7fc00000u;                                                                           \
            asm volatile(".irp r,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115\n\tv_mov_b32 v\\r, %[one]\n\t.endr\n\t"  \
                         "v_mov_b32 v116, %[nan]\n\tv_mov_b32 v117, %[nan]\n\t"                                                   \
                         ".irp r,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151\n\tv_mov_b32 v\\r, 0\n\t.endr\n\t" \
                         "s_nop 15\n\t"                                                                                           \
                         ".if " #CHAIN "\n\tv_mfma_f32_32x32x16_f16 v[120:135], v[112:115], v[112:115], v[120:135]\n\tv_mfma_f32_32x32x16_f16 v[136:151], v[112:115], v[112:115], v[136:151]\n\t.endif\n\t" \
                         "ds_read_b128 v[104:107], %[la]\n\tds_read_b128 v[108:111], %[la] offset:4096\n\t"                        \
                         "s_waitcnt lgkmcnt(1)\n\t"                                                                               \
                         "v_mfma_f32_32x32x16_f16 v[120:135], v[100:103], v[104:107], v[120:135]\n\t"                             \
                         "s_waitcnt lgkmcnt(0)\n\t"                                                                               \
                         "v_mfma_f32_32x32x16_f16 v[136:151], v[100:103], v[108:111], v[136:151]\n\t" WAIT CLOB                    \
                         "s_nop 15\n\ts_nop 15\n\t"                                                                               \
                         ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\t.endr\n\t"                                              \
                         "v_mov_b32 %[a], v120\n\tv_mov_b32 %[b], v135\n\tv_mov_b32 %[c], v136\n\tv_mov_b32 %[e], v151\n\t"        \
                         : [a] "=&v"(d0[0]), [b] "=&v"(d0[1]), [c] "=&v"(d1[0]), [e] "=&v"(d1[1])                                  \
                         : [one] "v"(one), [nan] "v"(nan), [la] "v"(la)                                                           \
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
                           "v114", "v115", "v116", "v117", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", \
                           "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", \
                           "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "memory");                             \
            const float want = CHAIN ? 32.f : 16.f;                                                                               \
            bad += !(d0[0] == want && d0[1] == want && d1[0] == want && d1[1] == want);                                           \
        }                                                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = bad;                                                                                \
    }
#define ALLW_DCN(BASE, CLOB, CHAIN)            \
    PROBE_DCN(BASE##_w0, W0, CLOB, CHAIN)      \
    PROBE_DCN(BASE##_w1, W1, CLOB, CHAIN)      \
    PROBE_DCN(BASE##_w2, W2, CLOB, CHAIN)      \
    PROBE_DCN(BASE##_w3, W3, CLOB, CHAIN)
ALLW_DCN(dcn_cvt4, C_CVT4, 0)
ALLW_DCN(dcn_cvt1, C_CVT1, 0)
ALLW_DCN(dcn_mov3, C_A3, 0)
ALLW_DCN(dcnq_cvt4, C_CVT4, 1)
ALLW_DCN(dcnq_mov3, C_A3, 1)

typedef void (*kern_t)(unsigned*, int);
struct Case { const char* what; kern_t k[4]; };
#define CASE(BASE, WHAT) {WHAT, {BASE##_w0, BASE##_w1, BASE##_w2, BASE##_w3}}

int main() {
    const int blocks = 512, iters = 16;
    unsigned* out;
    if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess) return 1;
    std::vector<unsigned> h(blocks * 256);
    const Case cases[] = {
        CASE(h32_none, "v_mfma_f32_32x32x16_f16 (A v[100:103], B v[104:107]), nothing behind it (control)"),
        CASE(h32_a3, "v_mfma_f32_32x32x16_f16 ; v_mov_b32 v103 (last register of A)"),
        CASE(h32_a0, "v_mfma_f32_32x32x16_f16 ; v_mov_b32 v100 (first register of A)"),
        CASE(h32_a23pk, "v_mfma_f32_32x32x16_f16 ; v_pk_mul_f32 v[102:103]"),
        CASE(h32_a23m64, "v_mfma_f32_32x32x16_f16 ; v_mov_b64 v[102:103]"),
        CASE(h32_a01pk, "v_mfma_f32_32x32x16_f16 ; v_pk_mul_f32 v[100:101]"),
        CASE(h32_b3, "v_mfma_f32_32x32x16_f16 ; v_mov_b32 v107 (last register of B)"),
        CASE(h32_b0, "v_mfma_f32_32x32x16_f16 ; v_mov_b32 v104 (first register of B)"),
        CASE(h32_b23pk, "v_mfma_f32_32x32x16_f16 ; v_pk_mul_f32 v[106:107]"),
        CASE(h32q_a3, "2 MFMAs queued ; v_mfma_f32_32x32x16_f16 ; v_mov_b32 v103"),
        CASE(h32q_a23pk, "2 MFMAs queued ; v_mfma_f32_32x32x16_f16 ; v_pk_mul_f32 v[102:103]"),
        CASE(h32q_b3, "2 MFMAs queued ; v_mfma_f32_32x32x16_f16 ; v_mov_b32 v107"),
        CASE(h32q_b23pk, "2 MFMAs queued ; v_mfma_f32_32x32x16_f16 ; v_pk_mul_f32 v[106:107]"),
        CASE(h16_a3, "v_mfma_f32_16x16x32_f16 ; v_mov_b32 v103"),
        CASE(h16_a23pk, "v_mfma_f32_16x16x32_f16 ; v_pk_mul_f32 v[102:103]"),
        CASE(h16_b3, "v_mfma_f32_16x16x32_f16 ; v_mov_b32 v107"),
        CASE(h16q_a23pk, "2 MFMAs queued ; v_mfma_f32_16x16x32_f16 ; v_pk_mul_f32 v[102:103]"),
        CASE(h16q_b23pk, "2 MFMAs queued ; v_mfma_f32_16x16x32_f16 ; v_pk_mul_f32 v[106:107]"),
        CASE(f16_a0, "v_mfma_f32_16x16x4_f32 (A v100, B v104) ; v_mov_b32 v100"),
        CASE(f16_a01pk, "v_mfma_f32_16x16x4_f32 ; v_pk_mul_f32 v[100:101]"),
        CASE(f16_b0, "v_mfma_f32_16x16x4_f32 ; v_mov_b32 v104"),
        CASE(f16q_a0, "2 MFMAs queued ; v_mfma_f32_16x16x4_f32 ; v_mov_b32 v100"),
        CASE(f16q_a01pk, "2 MFMAs queued ; v_mfma_f32_16x16x4_f32 ; v_pk_mul_f32 v[100:101]"),
        CASE(f16q_b0, "2 MFMAs queued ; v_mfma_f32_16x16x4_f32 ; v_mov_b32 v104"),
        CASE(f32_a0, "v_mfma_f32_32x32x2_f32 ; v_mov_b32 v100"),
        CASE(f32_b0, "v_mfma_f32_32x32x2_f32 ; v_mov_b32 v104"),
        CASE(dcn_cvt4, "2 x 32x32x16_f16 sharing A, B from LDS (waitcnt) ; v_cvt_pk_f16_f32 v103, v102, v101, v100"),
        CASE(dcn_cvt1, "2 x 32x32x16_f16 sharing A, B from LDS (waitcnt) ; v_cvt_pk_f16_f32 v103"),
        CASE(dcn_mov3, "2 x 32x32x16_f16 sharing A, B from LDS (waitcnt) ; v_mov_b32 v103"),
        CASE(dcnq_cvt4, "2 MFMAs queued into the same accumulators ; the same ; v_cvt_pk_f16_f32 v103 .. v100"),
        CASE(dcnq_mov3, "2 MFMAs queued into the same accumulators ; the same ; v_mov_b32 v103"),
    };
    printf("%d threads x %d MFMAs per probe (two blocks of four waves per CU); wrong = lanes whose outputs are not all exactly K\n", blocks * 256, iters);
    printf("%-92s %12s %12s %12s %12s\n", "MFMA ; next vector-ALU instruction", "0 wait", "1 (s_nop 0)", "2 (s_nop 1)", "3 (s_nop 2)");
    for (const Case& c : cases) {
        char line[512];
        int n = snprintf(line, sizeof line, "%-92s", c.what);
        for (int w = 0; w < 4; ++w) {
            (void)hipMemset(out, 0xff, blocks * 256 * 4);
            hipLaunchKernelGGL(c.k[w], dim3(blocks), dim3(256), 0, 0, out, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", c.what); return 1; }
            (void)hipMemcpy(h.data(), out, blocks * 256 * 4, hipMemcpyDeviceToHost);
            unsigned long long bad = 0;
            for (unsigned v : h) bad += v;
            n += snprintf(line + n, sizeof line - n, " %12llu", bad);
        }
        printf("%s\n", line);
    }
    return 0;
}
