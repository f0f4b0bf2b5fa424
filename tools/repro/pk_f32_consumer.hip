// Is the result of a packed fp32 VALU op (v_pk_add_f32) safe to consume by the very next instruction -- and is a freshly written
// register safe to feed INTO one?  Three kernels of this repo produced wrong values in fixed (lane, register) slots whenever packed
// fp32 ops were in their epilogues (profiles/r03_dcn_hazard_report.txt, r04_f16x3_resplit_hazard_report.txt,
// r05_wino_packed_f32_hazard.txt).  Each probe is ONE asm block on fixed registers: producer, N wait states, consumer; the same block
// with 16 wait states is the reference; results are compared bit for bit, per lane, over many iterations and changing data, with and
// without fp32 MFMAs issued right in front.
//   hipcc -O3 --offload-arch=gfx950 -o pk_f32_consumer pk_f32_consumer.hip && ./pk_f32_consumer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define MF0 ""
#define MF1 "v_mfma_f32_16x16x4_f32 v[40:43], v30, v31, v[40:43]\n\tv_mfma_f32_16x16x4_f32 v[44:47], v30, v31, v[44:47]\n\t"
#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W4 "s_nop 3\n\t"
#define WREF "s_nop 15\n\t"
// producer / consumer pairs on v[10:11] (pair), v[12:13] (increment), result -> v20
#define P_PK "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n\t"
#define P_PKMUL "v_pk_mul_f32 v[10:11], v[10:11], v[12:13]\n\t"
#define C_ADD_HI "v_add_f32 v20, 0, v11\n\t"
#define C_ADD_LO "v_add_f32 v20, 0, v10\n\t"
#define C_MAX_HI "v_max_f32 v20, v11, v11\n\t"
#define C_LDS_HI "ds_write_b32 v21, v11\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 v20, v21\n\ts_waitcnt lgkmcnt(0)\n\t"
#define C_ACCW_HI "v_accvgpr_write_b32 a0, v11\n\ts_nop 7\n\tv_accvgpr_read_b32 v20, a0\n\t"
#define C_MFMA_HI "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"
// reverse direction: plain VALU / accvgpr_read writes v11, then the packed op reads the pair
#define R_MOV "v_add_f32 v11, v11, v13\n\t"
#define R_ACCR "v_accvgpr_write_b32 a1, v11\n\ts_nop 7\n\tv_accvgpr_read_b32 v11, a1\n\t"
#define R_PKSUM "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n\ts_nop 15\n\tv_add_f32 v20, 0, v11\n\t"

#define PROBE(NAME, MF, PROD, WAIT, CONS)                                                                                       \
    __global__ __launch_bounds__(256, 1) void NAME(unsigned* out, int iters) {                                                    \
        __shared__ float ldsbuf[256];                                                                                             \
        const int t = threadIdx.x;                                                                                                \
        ldsbuf[t] = 0.f;                                                                                                          \
        __syncthreads();                                                                                                          \
        unsigned bad = 0;                                                                                                         \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            float x0 = (float)(t * 3 + it), x1 = (float)(t * 7 + 2 * it + 1), i0 = 1.0f + (it & 7), i1 = 2.0f + (it & 3);           \
            float r, ref;                                                                                                         \
            unsigned la = (unsigned)(t * 4);                                                                                      \
            asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\tv_mov_b32 v21, %5\n\t" \
                         "v_mov_b32 v30, 1.0\n\tv_mov_b32 v31, 1.0\n\ts_nop 15\n\t" MF PROD WAIT CONS "s_nop 15\n\tv_mov_b32 %0, v20" \
                         : "=v"(r)                                                                                                \
                         : "v"(x0), "v"(x1), "v"(i0), "v"(i1), "v"(la)                                                            \
                         : "v10", "v11", "v12", "v13", "v20", "v21", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", \
                           "v47", "v48", "v49", "v50", "v51", "a0", "a1", "memory");                                              \
            asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\tv_mov_b32 v21, %5\n\t" \
                         "v_mov_b32 v30, 1.0\n\tv_mov_b32 v31, 1.0\n\ts_nop 15\n\t" MF PROD WREF CONS "s_nop 15\n\tv_mov_b32 %0, v20" \
                         : "=v"(ref)                                                                                              \
                         : "v"(x0), "v"(x1), "v"(i0), "v"(i1), "v"(la)                                                            \
                         : "v10", "v11", "v12", "v13", "v20", "v21", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", \
                           "v47", "v48", "v49", "v50", "v51", "a0", "a1", "memory");                                              \
            bad += (__builtin_bit_cast(unsigned, r) != __builtin_bit_cast(unsigned, ref));                                        \
        }                                                                                                                         \
        out[blockIdx.x * 256 + t] = bad + (ldsbuf[t] == 12345.f);                                                                 \
    }

#define FAMILY(TAG, PROD, CONS)          \
    PROBE(TAG##_m0_w0, MF0, PROD, W0, CONS) \
    PROBE(TAG##_m0_w1, MF0, PROD, W1, CONS) \
    PROBE(TAG##_m1_w0, MF1, PROD, W0, CONS) \
    PROBE(TAG##_m1_w1, MF1, PROD, W1, CONS) \
    PROBE(TAG##_m1_w2, MF1, PROD, W2, CONS) \
    PROBE(TAG##_m1_w4, MF1, PROD, W4, CONS)

FAMILY(pk_add_hi, P_PK, C_ADD_HI)
FAMILY(pk_add_lo, P_PK, C_ADD_LO)
FAMILY(pk_max_hi, P_PK, C_MAX_HI)
FAMILY(pkmul_max_hi, P_PKMUL, C_MAX_HI)
FAMILY(pk_lds_hi, P_PK, C_LDS_HI)
FAMILY(pk_accw_hi, P_PK, C_ACCW_HI)
FAMILY(valu_then_pk, R_MOV, R_PKSUM)
FAMILY(accr_then_pk, R_ACCR, R_PKSUM)

typedef void (*kern_t)(unsigned*, int);
static void run(const char* name, kern_t k, unsigned* out) {
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, 2000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(256 * 256);
    (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long bad = 0, lanes = 0;
    unsigned lanemask[64] = {};
    for (size_t i = 0; i < h.size(); ++i)
        if (h[i]) {
            bad += h[i];
            ++lanes;
            lanemask[i & 63] = 1;
        }
    int nl = 0;
    for (int i = 0; i < 64; ++i) nl += lanemask[i];
    printf("%-22s wrong results %10llu in %6llu threads, %2d distinct lane ids\n", name, bad, lanes, nl);
}
#define RUNF(TAG) \
    run(#TAG " m0 w0", TAG##_m0_w0, out); run(#TAG " m0 w1", TAG##_m0_w1, out); run(#TAG " m1 w0", TAG##_m1_w0, out); \
    run(#TAG " m1 w1", TAG##_m1_w1, out); run(#TAG " m1 w2", TAG##_m1_w2, out); run(#TAG " m1 w4", TAG##_m1_w4, out);
int main() {
    unsigned* out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    RUNF(pk_add_hi) RUNF(pk_add_lo) RUNF(pk_max_hi) RUNF(pkmul_max_hi) RUNF(pk_lds_hi) RUNF(pk_accw_hi) RUNF(valu_then_pk) RUNF(accr_then_pk)
    return 0;
}
