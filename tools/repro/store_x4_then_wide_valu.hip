// Does a 16-byte vector-memory store still read its data registers when the NEXT vector-ALU instruction overwrites them?
//
// Round 6 found the wrong values of the packed-fp32 Winograd builds (profiles/r05_wino_packed_f32_hazard.txt,
// r06_wino_pk_add_probe.txt) in one place of the Winograd epilogue's code object:
//       buffer_store_dwordx4 v[34:37], v55, s[24:27], s8 offen
//       v_pk_add_f32 v[34:35], v[230:231], v[198:199]          <- the next tile's output transform, into the store's data registers
// and the wrong output values are exactly the HIGH half of the overwritten pair (element 1 of the float4), in lanes 12-15 of each row
// of 16.  LLVM's hazard recogniser knows "a VMEM store of more than 64 bits followed by a VALU write of its data VGPRs needs 1 wait
// state (2 on gfx940+)" but exempts MUBUF stores whose soffset operand is an SGPR (GCNHazardRecognizer::createsVALUHazard), so it
// puts no s_nop there; scalar code gets away with it because v_add_f32 v34 / v_add_f32 v35 overwrite element 0 first.
//
// Each probe is ONE asm block on fixed registers: data into v[100:103], the store, N wait states, the overwriting instruction;
// every iteration stores to its own address and the host compares the whole buffer with the expected pattern afterwards.
//   hipcc -O3 --offload-arch=gfx950 -o store_x4_then_wide_valu store_x4_then_wide_valu.hip && ./store_x4_then_wide_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W3 "s_nop 2\n\t"
// stores of v[100:103] (%[v] = byte offset VGPR, %[r] = resource, %[s] = an SGPR holding 0, %[p] = 64-bit address pair)
#define ST_BUF_S "buffer_store_dwordx4 v[100:103], %[v], %[r], %[s] offen\n\t"       // soffset in an SGPR: the exempted form
#define ST_BUF_0 "buffer_store_dwordx4 v[100:103], %[v], %[r], 0 offen\n\t"          // soffset = constant 0: the form LLVM pads
#define ST_BUF3_S "buffer_store_dwordx3 v[100:102], %[v], %[r], %[s] offen\n\t"
#define ST_BUF2_S "buffer_store_dwordx2 v[100:101], %[v], %[r], %[s] offen\n\t"
#define ST_GLB "global_store_dwordx4 %[p], v[100:103], off\n\t"
// a queue of stores in front (to the scratch half of the buffer), so that the probed store waits for issue slots like the epilogue's
#define BUSY0 ""
#define BUSY4                                                                                                                    \
    "buffer_store_dwordx4 v[108:111], %[v], %[r], %[s2] offen\n\tbuffer_store_dwordx4 v[108:111], %[v], %[r], %[s2] offen offset:16\n\t" \
    "buffer_store_dwordx4 v[108:111], %[v], %[r], %[s2] offen offset:32\n\tbuffer_store_dwordx4 v[108:111], %[v], %[r], %[s2] offen offset:48\n\t"
// the instruction behind the store (v[104:107] hold the poison 0x7fc0dead)
#define C_PK01 "v_pk_add_f32 v[100:101], v[104:105], v[106:107]\n\t"
#define C_PK23 "v_pk_add_f32 v[102:103], v[104:105], v[106:107]\n\t"
#define C_PKMUL23 "v_pk_mul_f32 v[102:103], v[104:105], v[106:107]\n\t"
#define C_MOV64_01 "v_mov_b64 v[100:101], v[104:105]\n\t"
#define C_MOV64_23 "v_mov_b64 v[102:103], v[104:105]\n\t"
#define C_S0 "v_mov_b32 v100, v104\n\t"
#define C_S1 "v_mov_b32 v101, v104\n\t"
#define C_S2 "v_mov_b32 v102, v104\n\t"
#define C_S3 "v_mov_b32 v103, v104\n\t"
#define C_S0123 "v_mov_b32 v100, v104\n\tv_mov_b32 v101, v104\n\tv_mov_b32 v102, v104\n\tv_mov_b32 v103, v104\n\t"      // compiler order
#define C_S3210 "v_mov_b32 v103, v104\n\tv_mov_b32 v102, v104\n\tv_mov_b32 v101, v104\n\tv_mov_b32 v100, v104\n\t"
#define C_ACC3 "v_accvgpr_read_b32 v103, a0\n\t"
#define C_NONE ""

#define PROBE(NAME, BUSY, STORE, WAIT, CLOB)                                                                                     \
    __global__ __launch_bounds__(256, 1) void NAME(unsigned* buf, unsigned long long bytes, int iters) {                          \
        const unsigned gt = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;                                               \
        const unsigned long long pa = (unsigned long long)buf;                                                                    \
        i32x4 rs = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)(unsigned)bytes, 0x00020000};                              \
        rs[0] = __builtin_amdgcn_readfirstlane(rs[0]);                                                                            \
        rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);                                                                            \
        rs[2] = __builtin_amdgcn_readfirstlane(rs[2]);                                                                            \
        rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);                                                                            \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            const unsigned slot = (unsigned)it * nthr + gt, off = slot * 16u;                                                     \
            const unsigned d0 = slot * 4u + 0x10000000u, d1 = d0 + 1u, d2 = d0 + 2u, d3 = d0 + 3u, poison = 0x7fc0deadu;          \
            const int s0 = __builtin_amdgcn_readfirstlane(it * 0), s2 = __builtin_amdgcn_readfirstlane((int)(bytes / 2) + it * 0);  \
            const unsigned long long ga = pa + off;                                                                               \
            asm volatile("v_mov_b32 v100, %[d0]\n\tv_mov_b32 v101, %[d1]\n\tv_mov_b32 v102, %[d2]\n\tv_mov_b32 v103, %[d3]\n\t"    \
                         "v_mov_b32 v104, %[po]\n\tv_mov_b32 v105, %[po]\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"            \
                         "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"                    \
                         "v_accvgpr_write_b32 a0, %[po]\n\ts_nop 15\n\t" BUSY STORE WAIT CLOB "s_nop 15\n\t"                       \
                         :                                                                                                        \
                         : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [po] "v"(poison), [v] "v"(off), [r] "s"(rs),     \
                           [s] "s"(s0), [s2] "s"(s2), [p] "v"(ga)                                                                  \
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "a0",  \
                           "memory");                                                                                             \
        }                                                                                                                         \
    }

#define ALLW(BASE, BUSY, STORE, CLOB)              \
    PROBE(BASE##_w0, BUSY, STORE, W0, CLOB)        \
    PROBE(BASE##_w1, BUSY, STORE, W1, CLOB)        \
    PROBE(BASE##_w2, BUSY, STORE, W2, CLOB)        \
    PROBE(BASE##_w3, BUSY, STORE, W3, CLOB)

ALLW(bs_pk01, BUSY0, ST_BUF_S, C_PK01)
ALLW(bs_pk23, BUSY0, ST_BUF_S, C_PK23)
ALLW(bs_pkmul23, BUSY0, ST_BUF_S, C_PKMUL23)
ALLW(bs_mov64_01, BUSY0, ST_BUF_S, C_MOV64_01)
ALLW(bs_mov64_23, BUSY0, ST_BUF_S, C_MOV64_23)
ALLW(bs_s0, BUSY0, ST_BUF_S, C_S0)
ALLW(bs_s1, BUSY0, ST_BUF_S, C_S1)
ALLW(bs_s2, BUSY0, ST_BUF_S, C_S2)
ALLW(bs_s3, BUSY0, ST_BUF_S, C_S3)
ALLW(bs_s0123, BUSY0, ST_BUF_S, C_S0123)
ALLW(bs_s3210, BUSY0, ST_BUF_S, C_S3210)
ALLW(bs_acc3, BUSY0, ST_BUF_S, C_ACC3)
ALLW(bs_none, BUSY0, ST_BUF_S, C_NONE)
ALLW(b0_pk01, BUSY0, ST_BUF_0, C_PK01)
ALLW(b0_pk23, BUSY0, ST_BUF_0, C_PK23)
ALLW(b0_s3, BUSY0, ST_BUF_0, C_S3)
ALLW(b3_s2, BUSY0, ST_BUF3_S, C_S2)
ALLW(b2_pk01, BUSY0, ST_BUF2_S, C_PK01)
ALLW(gl_pk01, BUSY0, ST_GLB, C_PK01)
ALLW(gl_pk23, BUSY0, ST_GLB, C_PK23)
ALLW(gl_s3, BUSY0, ST_GLB, C_S3)
ALLW(qs_pk01, BUSY4, ST_BUF_S, C_PK01)
ALLW(qs_pk23, BUSY4, ST_BUF_S, C_PK23)
ALLW(qs_mov64_23, BUSY4, ST_BUF_S, C_MOV64_23)
ALLW(qs_s1, BUSY4, ST_BUF_S, C_S1)
ALLW(qs_s3, BUSY4, ST_BUF_S, C_S3)
ALLW(qs_s0123, BUSY4, ST_BUF_S, C_S0123)
ALLW(qs_s3210, BUSY4, ST_BUF_S, C_S3210)
ALLW(q0_pk01, BUSY4, ST_BUF_0, C_PK01)
ALLW(q0_s3, BUSY4, ST_BUF_0, C_S3)

// the same question for LDS stores: ds_write_b128 / ds_write_b64 of v[100:103], the overwriting instruction, then the value read back
#define ST_DS128 "ds_write_b128 %[la], v[100:103]\n\t"
#define ST_DS64X2 "ds_write2_b64 %[la], v[100:101], v[102:103] offset1:1\n\t"
#define PROBE_DS(NAME, STORE, WAIT, CLOB)                                                                                        \
    __global__ __launch_bounds__(256, 1) void NAME(unsigned* buf, unsigned long long bytes, int iters) {                          \
        __shared__ unsigned lds[1024];                                                                                            \
        lds[threadIdx.x] = 0;                                                                                                     \
        __syncthreads();                                                                                                          \
        const unsigned gt = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;                                               \
        const unsigned long long pa = (unsigned long long)buf;                                                                    \
        i32x4 rs = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)(unsigned)bytes, 0x00020000};                              \
        rs[0] = __builtin_amdgcn_readfirstlane(rs[0]);                                                                            \
        rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);                                                                            \
        rs[2] = __builtin_amdgcn_readfirstlane(rs[2]);                                                                            \
        rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);                                                                            \
        const unsigned la = (unsigned)(unsigned long long)(&lds[0]) + threadIdx.x * 16u;                                          \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            const unsigned slot = (unsigned)it * nthr + gt, off = slot * 16u;                                                     \
            const unsigned d0 = slot * 4u + 0x10000000u, d1 = d0 + 1u, d2 = d0 + 2u, d3 = d0 + 3u, poison = 0x7fc0deadu;          \
            asm volatile("v_mov_b32 v100, %[d0]\n\tv_mov_b32 v101, %[d1]\n\tv_mov_b32 v102, %[d2]\n\tv_mov_b32 v103, %[d3]\n\t"    \
                         "v_mov_b32 v104, %[po]\n\tv_mov_b32 v105, %[po]\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"            \
                         "v_accvgpr_write_b32 a0, %[po]\n\ts_nop 15\n\t" STORE WAIT CLOB "s_nop 15\n\ts_waitcnt lgkmcnt(0)\n\t"    \
                         "ds_read_b128 v[108:111], %[la]\n\ts_waitcnt lgkmcnt(0)\n\t"                                             \
                         "buffer_store_dwordx4 v[108:111], %[v], %[r], 0 offen\n\ts_nop 15\n\t"                                   \
                         :                                                                                                        \
                         : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [po] "v"(poison), [v] "v"(off), [r] "s"(rs),     \
                           [la] "v"(la)                                                                                            \
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "a0",  \
                           "memory");                                                                                             \
        }                                                                                                                         \
    }
#define ALLW_DS(BASE, STORE, CLOB)            \
    PROBE_DS(BASE##_w0, STORE, W0, CLOB)      \
    PROBE_DS(BASE##_w1, STORE, W1, CLOB)      \
    PROBE_DS(BASE##_w2, STORE, W2, CLOB)      \
    PROBE_DS(BASE##_w3, STORE, W3, CLOB)
ALLW_DS(ds_pk01, ST_DS128, C_PK01)
ALLW_DS(ds_pk23, ST_DS128, C_PK23)
ALLW_DS(ds_s0, ST_DS128, C_S0)
ALLW_DS(ds_s3, ST_DS128, C_S3)
ALLW_DS(ds_mov64_23, ST_DS128, C_MOV64_23)
ALLW_DS(d2_pk23, ST_DS64X2, C_PK23)
ALLW_DS(d2_s3, ST_DS64X2, C_S3)

// ... and for stores to scratch (what a register spill is): scratch_store_dword / x2 / x4 of v[100:103] to the lane's private
// memory, the overwriting instruction, then the value read back
#define ST_SCR1 "scratch_store_dword %[pa], v100, off\n\t"
#define ST_SCR2 "scratch_store_dwordx2 %[pa], v[100:101], off\n\t"
#define ST_SCR4 "scratch_store_dwordx4 %[pa], v[100:103], off\n\t"
#define PROBE_SCR(NAME, STORE, WAIT, CLOB)                                                                                       \
    __global__ __launch_bounds__(256, 1) void NAME(unsigned* buf, unsigned long long bytes, int iters) {                          \
        volatile unsigned priv[8];                                                                                                \
        priv[threadIdx.x & 7] = 0;                                                                                                \
        const unsigned gt = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;                                               \
        const unsigned long long pa64 = (unsigned long long)buf;                                                                  \
        i32x4 rs = {(int)(unsigned)pa64, (int)(unsigned)(pa64 >> 32), (int)(unsigned)bytes, 0x00020000};                          \
        rs[0] = __builtin_amdgcn_readfirstlane(rs[0]);                                                                            \
        rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);                                                                            \
        rs[2] = __builtin_amdgcn_readfirstlane(rs[2]);                                                                            \
        rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);                                                                            \
        const unsigned pa = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(5))) char*)(&priv[0]);                      \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            const unsigned slot = (unsigned)it * nthr + gt, off = slot * 16u;                                                     \
            const unsigned d0 = slot * 4u + 0x10000000u, d1 = d0 + 1u, d2 = d0 + 2u, d3 = d0 + 3u, poison = 0x7fc0deadu;          \
            asm volatile("v_mov_b32 v100, %[d0]\n\tv_mov_b32 v101, %[d1]\n\tv_mov_b32 v102, %[d2]\n\tv_mov_b32 v103, %[d3]\n\t"    \
                         "v_mov_b32 v104, %[po]\n\tv_mov_b32 v105, %[po]\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"            \
                         "v_mov_b32 v108, %[d0]\n\tv_mov_b32 v109, %[d1]\n\tv_mov_b32 v110, %[d2]\n\tv_mov_b32 v111, %[d3]\n\t"    \
                         "v_accvgpr_write_b32 a0, %[po]\n\ts_nop 15\n\t" STORE WAIT CLOB "s_nop 15\n\ts_waitcnt vmcnt(0)\n\t"      \
                         "scratch_load_dwordx4 v[108:111], %[pa], off\n\ts_waitcnt vmcnt(0)\n\t"                                  \
                         "buffer_store_dwordx4 v[108:111], %[v], %[r], 0 offen\n\ts_nop 15\n\t"                                   \
                         :                                                                                                        \
                         : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [po] "v"(poison), [v] "v"(off), [r] "s"(rs),     \
                           [pa] "v"(pa)                                                                                            \
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "a0",  \
                           "memory");                                                                                             \
        }                                                                                                                         \
        if (priv[3] == 0x12345u) buf[0] = 1;                                                                                      \
    }
#define ALLW_SCR(BASE, STORE, CLOB)            \
    PROBE_SCR(BASE##_w0, STORE, W0, CLOB)      \
    PROBE_SCR(BASE##_w1, STORE, W1, CLOB)      \
    PROBE_SCR(BASE##_w2, STORE, W2, CLOB)      \
    PROBE_SCR(BASE##_w3, STORE, W3, CLOB)
ALLW_SCR(sc1_s0, ST_SCR1, C_S0)
ALLW_SCR(sc1_pk01, ST_SCR1, C_PK01)
ALLW_SCR(sc2_s0, ST_SCR2, C_S0)
ALLW_SCR(sc2_s1, ST_SCR2, C_S1)
ALLW_SCR(sc2_pk01, ST_SCR2, C_PK01)
ALLW_SCR(sc2_mov64, ST_SCR2, C_MOV64_01)
ALLW_SCR(sc4_pk23, ST_SCR4, C_PK23)
ALLW_SCR(sc4_s3, ST_SCR4, C_S3)
// 8-byte and 4-byte global / buffer stores behind a packed op (LLVM pads nothing at or below 64 bits)
#define ST_BUF1_S "buffer_store_dword v100, %[v], %[r], %[s] offen\n\t"
#define ST_GLB2 "global_store_dwordx2 %[p], v[100:101], off\n\t"
ALLW(b1_s0, BUSY0, ST_BUF1_S, C_S0)
ALLW(b1_pk01, BUSY0, ST_BUF1_S, C_PK01)
ALLW(g2_pk01, BUSY0, ST_GLB2, C_PK01)
ALLW(q2_pk01, BUSY4, ST_BUF2_S, C_PK01)

typedef void (*kern_t)(unsigned*, unsigned long long, int);
struct Case { const char* name; const char* what; int elems; kern_t k[4]; };
#define CASE(BASE, WHAT, ELEMS) {#BASE, WHAT, ELEMS, {BASE##_w0, BASE##_w1, BASE##_w2, BASE##_w3}}

int main() {
    const int blocks = 512, iters = 16;
    const unsigned nthr = blocks * 256;
    const unsigned long long half = (unsigned long long)nthr * iters * 16, bytes = 2 * half;      // second half: the busy queue's scratch
    unsigned* buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    std::vector<unsigned> h(half / 4);
    const Case cases[] = {
        CASE(bs_none, "buffer_store_dwordx4 (SGPR soffset), nothing behind it (control)", 4),
        CASE(bs_pk01, "buffer_store_dwordx4 (SGPR soffset) ; v_pk_add_f32 v[100:101]", 4),
        CASE(bs_pk23, "buffer_store_dwordx4 (SGPR soffset) ; v_pk_add_f32 v[102:103]", 4),
        CASE(bs_pkmul23, "buffer_store_dwordx4 (SGPR soffset) ; v_pk_mul_f32 v[102:103]", 4),
        CASE(bs_mov64_01, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b64 v[100:101]", 4),
        CASE(bs_mov64_23, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b64 v[102:103]", 4),
        CASE(bs_s0, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v100", 4),
        CASE(bs_s1, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v101", 4),
        CASE(bs_s2, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v102", 4),
        CASE(bs_s3, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v103", 4),
        CASE(bs_s0123, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v100, v101, v102, v103", 4),
        CASE(bs_s3210, "buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v103, v102, v101, v100", 4),
        CASE(bs_acc3, "buffer_store_dwordx4 (SGPR soffset) ; v_accvgpr_read_b32 v103", 4),
        CASE(b0_pk01, "buffer_store_dwordx4 (soffset 0) ; v_pk_add_f32 v[100:101]", 4),
        CASE(b0_pk23, "buffer_store_dwordx4 (soffset 0) ; v_pk_add_f32 v[102:103]", 4),
        CASE(b0_s3, "buffer_store_dwordx4 (soffset 0) ; v_mov_b32 v103", 4),
        CASE(b3_s2, "buffer_store_dwordx3 (SGPR soffset) ; v_mov_b32 v102", 3),
        CASE(b2_pk01, "buffer_store_dwordx2 (SGPR soffset) ; v_pk_add_f32 v[100:101]", 2),
        CASE(gl_pk01, "global_store_dwordx4 ; v_pk_add_f32 v[100:101]", 4),
        CASE(gl_pk23, "global_store_dwordx4 ; v_pk_add_f32 v[102:103]", 4),
        CASE(gl_s3, "global_store_dwordx4 ; v_mov_b32 v103", 4),
        CASE(qs_pk01, "4 stores queued ; buffer_store_dwordx4 (SGPR soffset) ; v_pk_add_f32 v[100:101]", 4),
        CASE(qs_pk23, "4 stores queued ; buffer_store_dwordx4 (SGPR soffset) ; v_pk_add_f32 v[102:103]", 4),
        CASE(qs_mov64_23, "4 stores queued ; buffer_store_dwordx4 (SGPR soffset) ; v_mov_b64 v[102:103]", 4),
        CASE(qs_s1, "4 stores queued ; buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v101", 4),
        CASE(qs_s3, "4 stores queued ; buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v103", 4),
        CASE(qs_s0123, "4 stores queued ; buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v100..v103", 4),
        CASE(qs_s3210, "4 stores queued ; buffer_store_dwordx4 (SGPR soffset) ; v_mov_b32 v103..v100", 4),
        CASE(q0_pk01, "4 stores queued ; buffer_store_dwordx4 (soffset 0) ; v_pk_add_f32 v[100:101]", 4),
        CASE(q0_s3, "4 stores queued ; buffer_store_dwordx4 (soffset 0) ; v_mov_b32 v103", 4),
        CASE(b1_s0, "buffer_store_dword (SGPR soffset) ; v_mov_b32 v100", 1),
        CASE(b1_pk01, "buffer_store_dword (SGPR soffset) ; v_pk_add_f32 v[100:101]", 1),
        CASE(g2_pk01, "global_store_dwordx2 ; v_pk_add_f32 v[100:101]", 2),
        CASE(q2_pk01, "4 stores queued ; buffer_store_dwordx2 (SGPR soffset) ; v_pk_add_f32 v[100:101]", 2),
        CASE(sc1_s0, "scratch_store_dword ; v_mov_b32 v100   (value read back from scratch)", 1),
        CASE(sc1_pk01, "scratch_store_dword ; v_pk_add_f32 v[100:101]", 1),
        CASE(sc2_s0, "scratch_store_dwordx2 ; v_mov_b32 v100", 2),
        CASE(sc2_s1, "scratch_store_dwordx2 ; v_mov_b32 v101", 2),
        CASE(sc2_pk01, "scratch_store_dwordx2 ; v_pk_add_f32 v[100:101]", 2),
        CASE(sc2_mov64, "scratch_store_dwordx2 ; v_mov_b64 v[100:101]", 2),
        CASE(sc4_pk23, "scratch_store_dwordx4 ; v_pk_add_f32 v[102:103]", 4),
        CASE(sc4_s3, "scratch_store_dwordx4 ; v_mov_b32 v103", 4),
        CASE(ds_pk01, "ds_write_b128 ; v_pk_add_f32 v[100:101]   (value read back from LDS)", 4),
        CASE(ds_pk23, "ds_write_b128 ; v_pk_add_f32 v[102:103]", 4),
        CASE(ds_s0, "ds_write_b128 ; v_mov_b32 v100", 4),
        CASE(ds_s3, "ds_write_b128 ; v_mov_b32 v103", 4),
        CASE(ds_mov64_23, "ds_write_b128 ; v_mov_b64 v[102:103]", 4),
        CASE(d2_pk23, "ds_write2_b64 ; v_pk_add_f32 v[102:103]", 4),
        CASE(d2_s3, "ds_write2_b64 ; v_mov_b32 v103", 4),
    };
    printf("%u threads x %d stores per probe; wrong = stored dwords that are not the value the store was issued with\n", nthr, iters);
    printf("%-88s %12s %12s %12s %12s\n", "store ; next vector-ALU instruction", "0 wait", "1 (s_nop 0)", "2 (s_nop 1)", "3 (s_nop 2)");
    for (const Case& c : cases) {
        char line[512];
        int n = snprintf(line, sizeof line, "%-88s", c.what);
        unsigned long long elem[4] = {0, 0, 0, 0}, lanes[4] = {0, 0, 0, 0};
        for (int w = 0; w < 4; ++w) {
            (void)hipMemset(buf, 0xff, bytes);
            hipLaunchKernelGGL(c.k[w], dim3(blocks), dim3(256), 0, 0, buf, bytes, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", c.name); return 1; }
            (void)hipMemcpy(h.data(), buf, half, hipMemcpyDeviceToHost);
            unsigned long long bad = 0;
            for (unsigned long long slot = 0; slot < (unsigned long long)nthr * iters; ++slot)
                for (int e = 0; e < c.elems; ++e)
                    if (h[slot * 4 + e] != (unsigned)slot * 4u + 0x10000000u + e) {
                        ++bad;
                        if (w == 0) { ++elem[e]; ++lanes[((slot % nthr) % 16) / 4]; }
                    }
            n += snprintf(line + n, sizeof line - n, " %12llu", bad);
        }
        printf("%s\n", line);
        if (elem[0] + elem[1] + elem[2] + elem[3])
            printf("      (0 wait) by element of the store: %llu %llu %llu %llu   by lane group (lane %% 16) / 4: %llu %llu %llu %llu\n", elem[0], elem[1],
                   elem[2], elem[3], lanes[0], lanes[1], lanes[2], lanes[3]);
    }
    return 0;
}
