// Does a memory instruction still read its ADDRESS registers when the next vector-ALU instruction overwrites them?
//
// Companion of store_x4_then_wide_valu.hip (there: the DATA registers of a store of more than 64 bits are read late).  Here the address
// operand of loads and stores -- the 64-bit pair of global_load / global_store, the voffset of buffer_load / buffer_store, the
// address of ds_read / ds_write -- is overwritten by the very next instruction with the address of ANOTHER slot (valid memory,
// different contents), 131072 threads, 0-3 wait states.  A load that returns the other slot's value, or a store that lands in the
// other slot, is a late read of the address.
//   hipcc -O3 --offload-arch=gfx950 -o mem_addr_then_valu mem_addr_then_valu.hip && ./mem_addr_then_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W3 "s_nop 2\n\t"
// v[100:101] = address of the lane's slot, v[102:103] = address of the other slot, v104 / v105 = byte offsets of the two, v106 / v107 =
// LDS addresses of the two; the loaded value ends in v110, stores write v[112:115]
#define L_GLB1 "global_load_dword v110, v[100:101], off\n\t"
#define L_GLB4 "global_load_dwordx4 v[108:111], v[100:101], off\n\t"
#define L_BUF1 "buffer_load_dword v110, v104, %[r], 0 offen\n\t"
#define L_BUF4 "buffer_load_dwordx4 v[108:111], v104, %[r], 0 offen\n\t"
#define L_DS1 "ds_read_b32 v110, v106\n\t"
#define L_DS4 "ds_read_b128 v[108:111], v106\n\t"
#define S_GLB4 "global_store_dwordx4 v[100:101], v[112:115], off\n\t"
#define S_GLB1 "global_store_dword v[100:101], v112, off\n\t"
#define S_BUF4 "buffer_store_dwordx4 v[112:115], v104, %[r], 0 offen\n\t"
#define S_DS4 "ds_write_b128 v106, v[112:115]\n\t"
#define C_LO "v_mov_b32 v100, v102\n\t"
#define C_B64 "v_mov_b64 v[100:101], v[102:103]\n\t"
#define C_ADD64 "v_lshl_add_u64 v[100:101], v[102:103], 0, 0\n\t"
#define C_PK "v_pk_mov_b32 v[100:101], v[102:103], v[102:103]\n\t"
#define C_OFF "v_mov_b32 v104, v105\n\t"
#define C_LDS "v_mov_b32 v106, v107\n\t"

// loads: slot i of `buf` holds i (host); the lane's slot is `gt`, the other one `gt ^ 1`.  out[gt] = number of iterations whose loaded
// value was not the own slot's.  stores: each iteration stores (marker, slot, it, 0) to the own slot of a per-iteration region; the host
// looks for markers that landed in the neighbour's slot (slot field != position).
#define PROBE_LOAD(NAME, LOAD, WAIT, CLOB, LDSINIT)                                                                              \
    __global__ __launch_bounds__(256, 2) void NAME(unsigned* buf, unsigned long long bytes, unsigned* out, int iters) {           \
        __shared__ unsigned lds[1024];                                                                                            \
        const unsigned gt = blockIdx.x * 256 + threadIdx.x;                                                                       \
        for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = blockIdx.x * 256 + (i >> 2);                                        \
        __syncthreads();                                                                                                          \
        const unsigned long long pa = (unsigned long long)buf;                                                                    \
        i32x4 rs = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)(unsigned)bytes, 0x00020000};                              \
        rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);                             \
        rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);                             \
        const unsigned o0 = gt * 16u, o1 = (gt ^ 1u) * 16u;                                                                       \
        const unsigned long long a0 = pa + o0, a1 = pa + o1;                                                                      \
        const unsigned l0 = (unsigned)(unsigned long long)(&lds[0]) + threadIdx.x * 16u, l1 = (unsigned)(unsigned long long)(&lds[0]) + (threadIdx.x ^ 1u) * 16u; \
        unsigned bad = 0;                                                                                                         \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            unsigned v;                                                                                                           \
            asm volatile("v_mov_b32 v100, %[a0l]\n\tv_mov_b32 v101, %[a0h]\n\tv_mov_b32 v102, %[a1l]\n\tv_mov_b32 v103, %[a1h]\n\t" \
                         "v_mov_b32 v104, %[o0]\n\tv_mov_b32 v105, %[o1]\n\tv_mov_b32 v106, %[l0]\n\tv_mov_b32 v107, %[l1]\n\ts_nop 15\n\t" \
                         LOAD WAIT CLOB "s_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_mov_b32 %[v], v110\n\t"                    \
                         : [v] "=v"(v)                                                                                            \
                         : [a0l] "v"((unsigned)a0), [a0h] "v"((unsigned)(a0 >> 32)), [a1l] "v"((unsigned)a1), [a1h] "v"((unsigned)(a1 >> 32)), \
                           [o0] "v"(o0), [o1] "v"(o1), [l0] "v"(l0), [l1] "v"(l1), [r] "s"(rs)                                      \
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "memory"); \
            bad += v != gt;                                                                                                       \
        }                                                                                                                         \
        out[gt] = bad;                                                                                                            \
    }
#define PROBE_STORE(NAME, STORE, WAIT, CLOB, TOLDS)                                                                              \
    __global__ __launch_bounds__(256, 2) void NAME(unsigned* buf, unsigned long long bytes, unsigned* out, int iters) {           \
        __shared__ unsigned lds[1024];                                                                                            \
        const unsigned gt = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;                                               \
        lds[threadIdx.x] = 0;                                                                                                     \
        __syncthreads();                                                                                                          \
        const unsigned long long pa = (unsigned long long)buf;                                                                    \
        i32x4 rs = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)(unsigned)bytes, 0x00020000};                              \
        rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);                             \
        rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);                             \
        const unsigned l0 = (unsigned)(unsigned long long)(&lds[0]) + threadIdx.x * 16u, l1 = (unsigned)(unsigned long long)(&lds[0]) + (threadIdx.x ^ 1u) * 16u; \
        unsigned bad = 0;                                                                                                         \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            const unsigned o0 = ((unsigned)it * nthr + gt) * 16u, o1 = ((unsigned)it * nthr + (gt ^ 1u)) * 16u;                     \
            const unsigned long long a0 = pa + o0, a1 = pa + o1;                                                                  \
            unsigned back;                                                                                                        \
            const unsigned val = gt + (unsigned)it * nthr;                                                                        \
            asm volatile("v_mov_b32 v100, %[a0l]\n\tv_mov_b32 v101, %[a0h]\n\tv_mov_b32 v102, %[a1l]\n\tv_mov_b32 v103, %[a1h]\n\t" \
                         "v_mov_b32 v104, %[o0]\n\tv_mov_b32 v105, %[o1]\n\tv_mov_b32 v106, %[l0]\n\tv_mov_b32 v107, %[l1]\n\t"    \
                         "v_mov_b32 v112, %[gt]\n\tv_mov_b32 v113, %[gt]\n\tv_mov_b32 v114, %[gt]\n\tv_mov_b32 v115, %[gt]\n\ts_nop 15\n\t" \
                         STORE WAIT CLOB "s_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\t"                               \
                         "v_mov_b32 v106, %[l0]\n\tds_read_b32 %[bk], v106\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"               \
                         : [bk] "=v"(back)                                                                                        \
                         : [a0l] "v"((unsigned)a0), [a0h] "v"((unsigned)(a0 >> 32)), [a1l] "v"((unsigned)a1), [a1h] "v"((unsigned)(a1 >> 32)), \
                           [o0] "v"(o0), [o1] "v"(o1), [l0] "v"(l0), [l1] "v"(l1), [gt] "v"(val), [r] "s"(rs)                       \
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v112", "v113", "v114", "v115", "memory"); \
            if (TOLDS) bad += back != val;                                                                                         \
        }                                                                                                                         \
        out[gt] = bad;                                                                                                            \
    }
#define ALL_L(BASE, LOAD, CLOB)                    \
    PROBE_LOAD(BASE##_w0, LOAD, W0, CLOB, 0)       \
    PROBE_LOAD(BASE##_w1, LOAD, W1, CLOB, 0)       \
    PROBE_LOAD(BASE##_w2, LOAD, W2, CLOB, 0)       \
    PROBE_LOAD(BASE##_w3, LOAD, W3, CLOB, 0)
#define ALL_S(BASE, STORE, CLOB, TOLDS)            \
    PROBE_STORE(BASE##_w0, STORE, W0, CLOB, TOLDS) \
    PROBE_STORE(BASE##_w1, STORE, W1, CLOB, TOLDS) \
    PROBE_STORE(BASE##_w2, STORE, W2, CLOB, TOLDS) \
    PROBE_STORE(BASE##_w3, STORE, W3, CLOB, TOLDS)
ALL_L(lg1_lo, L_GLB1, C_LO)
ALL_L(lg1_b64, L_GLB1, C_B64)
ALL_L(lg1_add64, L_GLB1, C_ADD64)
ALL_L(lg1_pk, L_GLB1, C_PK)
ALL_L(lg4_lo, L_GLB4, C_LO)
ALL_L(lg4_b64, L_GLB4, C_B64)
ALL_L(lb1_off, L_BUF1, C_OFF)
ALL_L(lb4_off, L_BUF4, C_OFF)
ALL_L(ld1, L_DS1, C_LDS)
ALL_L(ld4, L_DS4, C_LDS)
ALL_S(sg4_lo, S_GLB4, C_LO, 0)
ALL_S(sg4_b64, S_GLB4, C_B64, 0)
ALL_S(sg1_b64, S_GLB1, C_B64, 0)
ALL_S(sb4_off, S_BUF4, C_OFF, 0)
ALL_S(sd4, S_DS4, C_LDS, 1)

typedef void (*kern_t)(unsigned*, unsigned long long, unsigned*, int);
struct Case { const char* what; int kind; kern_t k[4]; };      // kind 0: load, 1: store to global (host check), 2: store to LDS (device check)
#define CASE(BASE, KIND, WHAT) {WHAT, KIND, {BASE##_w0, BASE##_w1, BASE##_w2, BASE##_w3}}

int main() {
    const int blocks = 512, iters = 16;
    const unsigned nthr = blocks * 256;
    const unsigned long long bytes = (unsigned long long)nthr * iters * 16;
    unsigned *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, nthr * 4) != hipSuccess) return 1;
    std::vector<unsigned> h(bytes / 4), ho(nthr);
    const Case cases[] = {
        CASE(lg1_lo, 0, "global_load_dword v[100:101] ; v_mov_b32 v100"),
        CASE(lg1_b64, 0, "global_load_dword v[100:101] ; v_mov_b64 v[100:101]"),
        CASE(lg1_add64, 0, "global_load_dword v[100:101] ; v_lshl_add_u64 v[100:101]"),
        CASE(lg1_pk, 0, "global_load_dword v[100:101] ; v_pk_mov_b32 v[100:101]"),
        CASE(lg4_lo, 0, "global_load_dwordx4 v[100:101] ; v_mov_b32 v100"),
        CASE(lg4_b64, 0, "global_load_dwordx4 v[100:101] ; v_mov_b64 v[100:101]"),
        CASE(lb1_off, 0, "buffer_load_dword voffset v104 ; v_mov_b32 v104"),
        CASE(lb4_off, 0, "buffer_load_dwordx4 voffset v104 ; v_mov_b32 v104"),
        CASE(ld1, 0, "ds_read_b32 v106 ; v_mov_b32 v106"),
        CASE(ld4, 0, "ds_read_b128 v106 ; v_mov_b32 v106"),
        CASE(sg4_lo, 1, "global_store_dwordx4 v[100:101] ; v_mov_b32 v100"),
        CASE(sg4_b64, 1, "global_store_dwordx4 v[100:101] ; v_mov_b64 v[100:101]"),
        CASE(sg1_b64, 1, "global_store_dword v[100:101] ; v_mov_b64 v[100:101]"),
        CASE(sb4_off, 1, "buffer_store_dwordx4 voffset v104 ; v_mov_b32 v104"),
        CASE(sd4, 2, "ds_write_b128 v106 ; v_mov_b32 v106"),
    };
    printf("%u threads x %d accesses per probe; wrong = accesses that went to the OTHER slot (the address written by the next instruction)\n", nthr, iters);
    printf("%-72s %12s %12s %12s %12s\n", "access ; next vector-ALU instruction (writes the address)", "0 wait", "1 (s_nop 0)", "2 (s_nop 1)", "3 (s_nop 2)");
    for (const Case& c : cases) {
        char line[512];
        int n = snprintf(line, sizeof line, "%-72s", c.what);
        for (int w = 0; w < 4; ++w) {
            if (c.kind == 0) {
                for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i >> 2);
                (void)hipMemcpy(buf, h.data(), (size_t)nthr * 16, hipMemcpyHostToDevice);
            } else {
                (void)hipMemset(buf, 0xff, bytes);
            }
            hipLaunchKernelGGL(c.k[w], dim3(blocks), dim3(256), 0, 0, buf, bytes, out, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", c.what); return 1; }
            unsigned long long bad = 0;
            if (c.kind == 1) {
                (void)hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost);
                for (unsigned long long s = 0; s < (unsigned long long)nthr * iters; ++s)
                    bad += h[s * 4] != (unsigned)s;
            } else {
                (void)hipMemcpy(ho.data(), out, nthr * 4, hipMemcpyDeviceToHost);
                for (unsigned v : ho) bad += v;
            }
            n += snprintf(line + n, sizeof line - n, " %12llu", bad);
        }
        printf("%s\n", line);
    }
    return 0;
}
