// Does a transcendental VALU op (quarter rate: v_exp_f32, v_rcp_f32, v_rsq_f32, v_sqrt_f32, v_log_f32, v_sin_f32) still read its source
// when the NEXT vector-ALU instruction overwrites it?  (gfx940+ runs a trans op beside the following non-trans VALU instruction; LLVM pads
// the read-after-write side of that -- a VALU use of a trans result -- but assumes the source is read at issue.)  Same form as
// store_x4_then_wide_valu.hip: 131072 threads, the overwriting instruction writes NaN into the source, 0-3 wait states; a result that
// differs from the one computed with 16 wait states in between is a late read.
//   hipcc -O3 --offload-arch=gfx950 -o trans_src_then_valu trans_src_then_valu.hip && ./trans_src_then_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W3 "s_nop 2\n\t"
#define WREF "s_nop 15\n\t"
#define C_MOV "v_mov_b32 v100, v104\n\t"
#define C_PK "v_pk_mul_f32 v[100:101], v[104:105], v[104:105]\n\t"
#define C_B64 "v_mov_b64 v[100:101], v[104:105]\n\t"
#define C_FMA "v_fma_f32 v100, v104, v104, v104\n\t"

#define PROBE(NAME, OP, WAIT, CLOB)                                                                                              \
    __global__ __launch_bounds__(256, 2) void NAME(unsigned* out, int iters) {                                                    \
        unsigned bad = 0;                                                                                                         \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            const float x = 0.5f + 0.001f * (float)((threadIdx.x * 7 + it * 13 + blockIdx.x) & 1023);                              \
            const unsigned nan = 0x7fc00000u;                                                                                     \
            unsigned r, ref;                                                                                                      \
            asm volatile("v_mov_b32 v100, %[x]\n\tv_mov_b32 v101, %[x]\n\tv_mov_b32 v104, %[n]\n\tv_mov_b32 v105, %[n]\n\ts_nop 15\n\t"     \
                         OP " v110, v100\n\t" WAIT CLOB "s_nop 15\n\tv_mov_b32 %[r], v110\n\t"                                    \
                         "v_mov_b32 v100, %[x]\n\tv_mov_b32 v101, %[x]\n\ts_nop 15\n\t"                                            \
                         OP " v110, v100\n\t" WREF CLOB "s_nop 15\n\tv_mov_b32 %[f], v110\n\t"                                    \
                         : [r] "=&v"(r), [f] "=&v"(ref)                                                                           \
                         : [x] "v"(x), [n] "v"(nan)                                                                               \
                         : "v100", "v101", "v104", "v105", "v110", "memory");                                                     \
            bad += r != ref;                                                                                                      \
        }                                                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = bad;                                                                                \
    }
#define ALLW(BASE, OP, CLOB)            \
    PROBE(BASE##_w0, OP, W0, CLOB)      \
    PROBE(BASE##_w1, OP, W1, CLOB)      \
    PROBE(BASE##_w2, OP, W2, CLOB)      \
    PROBE(BASE##_w3, OP, W3, CLOB)
ALLW(exp_mov, "v_exp_f32", C_MOV)
ALLW(exp_pk, "v_exp_f32", C_PK)
ALLW(exp_b64, "v_exp_f32", C_B64)
ALLW(exp_fma, "v_exp_f32", C_FMA)
ALLW(rcp_mov, "v_rcp_f32", C_MOV)
ALLW(rcp_pk, "v_rcp_f32", C_PK)
ALLW(rsq_pk, "v_rsq_f32", C_PK)
ALLW(sqrt_pk, "v_sqrt_f32", C_PK)
ALLW(log_pk, "v_log_f32", C_PK)
ALLW(sin_pk, "v_sin_f32", C_PK)
ALLW(cvt_pk, "v_cvt_f16_f32", C_PK)
ALLW(add_pk, "v_floor_f32", C_PK)

typedef void (*kern_t)(unsigned*, int);
struct Case { const char* what; kern_t k[4]; };
#define CASE(BASE, WHAT) {WHAT, {BASE##_w0, BASE##_w1, BASE##_w2, BASE##_w3}}
int main() {
    const int blocks = 512, iters = 16;
    unsigned* out;
    if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess) return 1;
    std::vector<unsigned> h(blocks * 256);
    const Case cases[] = {
        CASE(exp_mov, "v_exp_f32 v110, v100 ; v_mov_b32 v100"), CASE(exp_pk, "v_exp_f32 v110, v100 ; v_pk_mul_f32 v[100:101]"),
        CASE(exp_b64, "v_exp_f32 v110, v100 ; v_mov_b64 v[100:101]"), CASE(exp_fma, "v_exp_f32 v110, v100 ; v_fma_f32 v100"),
        CASE(rcp_mov, "v_rcp_f32 v110, v100 ; v_mov_b32 v100"), CASE(rcp_pk, "v_rcp_f32 v110, v100 ; v_pk_mul_f32 v[100:101]"),
        CASE(rsq_pk, "v_rsq_f32 v110, v100 ; v_pk_mul_f32 v[100:101]"), CASE(sqrt_pk, "v_sqrt_f32 v110, v100 ; v_pk_mul_f32 v[100:101]"),
        CASE(log_pk, "v_log_f32 v110, v100 ; v_pk_mul_f32 v[100:101]"), CASE(sin_pk, "v_sin_f32 v110, v100 ; v_pk_mul_f32 v[100:101]"),
        CASE(cvt_pk, "v_cvt_f16_f32 v110, v100 ; v_pk_mul_f32 v[100:101]  (not a trans op: control)"),
        CASE(add_pk, "v_floor_f32 v110, v100 ; v_pk_mul_f32 v[100:101]  (not a trans op: control)"),
    };
    printf("%d threads x %d ops per probe; wrong = results that differ from the same op with 16 wait states before the overwrite\n", blocks * 256, iters);
    printf("%-86s %12s %12s %12s %12s\n", "op ; next vector-ALU instruction (writes the op's source)", "0 wait", "1 (s_nop 0)", "2 (s_nop 1)", "3 (s_nop 2)");
    for (const Case& c : cases) {
        char line[512];
        int n = snprintf(line, sizeof line, "%-86s", c.what);
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
