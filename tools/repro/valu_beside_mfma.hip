// Fourth probe for the fp16 DCN hazard.  The DCN diagnostics (tools/repro/dcn_f16_hazard.py) show that what gets corrupted is the
// A operand -- the gathered SAMPLES, i.e. results of ordinary VALU / LDS work -- of lanes 16..31 and 48..63, and only in builds
// where that work runs while v_mfma_f32_32x32x16_f16 instructions are in flight on the same SIMD (same wave right behind its
// MFMAs, or the partner wave w+4 of the anti-phase pair).  The fp32 instantiation never fails: v_mfma_f32_32x32x2_f32 executes
// on the vector ALUs themselves, so nothing runs beside it.
//
// Here: a 512-thread block per CU.  Waves 4..7 (partner of waves 0..3 on the same SIMD) run a self-checking VALU loop of one
// instruction class; waves 0..3 either idle (QUIET) or issue fp16 MFMAs back to back.  SAME = 1 instead puts 4 MFMAs and then
// the VALU sequence into the SAME wave each iteration (what arm B of the bad kernel shape does).  Every result is compared with
// the value the same instruction sequence gives in a quiet run; mismatches are counted per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int CLS>
__device__ __forceinline__ float work(float x, int it, const f32x4* lds) {
    if (CLS == 0) {                       // transcendental: the kernel's sigmoid
        return __builtin_amdgcn_rcpf(1.f + __expf(-(x + 0.001f * it)));
    } else if (CLS == 1) {                // packed fp32 (what SLP makes of the bilinear blend)
        f32x2 a = {x, x * 0.5f}, b = {0.25f + it * 0.01f, 1.5f}, c = {1.f, 2.f};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
        return a[0] + a[1];
    } else if (CLS == 2) {                // plain fp32 fma chain
        float a = x;
        for (int i = 0; i < 4; ++i) a = __builtin_fmaf(a, 0.75f, 0.125f + it * 0.01f);
        return a;
    } else if (CLS == 3) {                // fp32 -> fp16 pack (the A operand conversion)
        f32x4 v = {x, x + 1.f, x * 2.f, x + it * 0.01f};
        const h4 q = __builtin_convertvector(v, h4);
        return (float)q[0] + (float)q[1] + (float)q[2] + (float)q[3];
    } else if (CLS == 5) {                // 64-bit register copies (what hipcc uses for accumulator / sample copies on gfx950)
        f32x2 a = {x + it, x * 3.f}, b, c;
        asm volatile("v_mov_b64 %0, %1" : "=v"(b) : "v"(a));
        asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "v"(b));
        return c[0] * 2.f + c[1];
    } else if (CLS == 6) {                // DPP / cross-lane (quad_perm swap), used by the fp16 mirror epilogue
        const int v = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x + it), 0xB1, 0xF, 0xF, true);
        return __builtin_bit_cast(float, v) + x;
    } else if (CLS == 7) {                // the operand forms SLP gives the bilinear blend in dcn.hip
        f32x2 a = {x + it * 0.25f, x * 0.5f + 1.f}, b = {0.25f + it * 0.01f, 1.5f + x}, c = {1.f + x, 2.f}, d, e, f, g;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(e) : "v"(d), "v"(c));
        asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(f) : "v"(e));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(g) : "v"(f), "v"(b), "v"(a));
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(g), "v"(c));
        return d[0] * 3.f + d[1];
    } else if (CLS == 8) {                // trans result consumed as ONE half of a packed operand (mask * sample)
        f32x2 a = {x + it * 0.25f, x * 0.5f + 1.f}, d;
        const float m = __builtin_amdgcn_rcpf(1.f + __expf(-(x + 0.001f * it)));
        f32x2 mm = {m, 0.5f};
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(mm));
        return d[0] + 2.f * d[1];
    } else {                              // LDS window reads (ds_read_b128) + blend
        const f32x4 p0 = lds[(threadIdx.x * 7 + it) & 1023], p1 = lds[(threadIdx.x * 13 + it * 3) & 1023];
        return (p0[0] + p0[1] + p0[2] + p0[3]) * 0.5f + p1[2] * x;
    }
}

template <int CLS, int SAME>
__global__ __launch_bounds__(512) void k(const _Float16* __restrict__ ab, const float* __restrict__ expect, float* __restrict__ out_expect,
                                          unsigned* __restrict__ bad, int iters, int quiet) {
    __shared__ f32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = f32x4{(float)i, i * 0.5f, 1.f, (float)(i & 7)};
    __syncthreads();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const h8 av = *reinterpret_cast<const h8*>(ab + lane * 8), bv = *reinterpret_cast<const h8*>(ab + 512 + lane * 8);
    f32x16 acc0 = {0}, acc1 = {0};
    const float x = 0.01f * lane + 0.1f * (wave & 3);
    unsigned nbad = 0;
    if (SAME) {
        if (wave >= 4) return;
        for (int it = 0; it < iters; ++it) {
            if (!quiet) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc1, 0, 0, 0);
            }
            const float y = work<CLS>(x, it & 63, lds);
            const int slot = ((blockIdx.x & 0) * 64 + (it & 63)) * 256 + (wave & 3) * 64 + lane;
            if (quiet) out_expect[slot] = y;
            else if (y != expect[slot]) ++nbad;
        }
    } else {
        if (wave < 4) {
            if (quiet) return;
            for (int it = 0; it < iters * 2; ++it) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc1, 0, 0, 0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                const float y = work<CLS>(x, it & 63, lds);
                const int slot = (it & 63) * 256 + (wave & 3) * 64 + lane;
                if (quiet) out_expect[slot] = y;
                else if (y != expect[slot]) ++nbad;
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out_expect[0] = s;           // keep the MFMAs alive
    if (nbad) atomicAdd(&bad[lane], nbad);
}

template <int CLS, int SAME>
void run(const char* name, const _Float16* ab, float* expect, unsigned* bad) {
    const int blocks = 256, iters = 20000;
    hipMemset(bad, 0, 64 * 4);
    hipLaunchKernelGGL((k<CLS, SAME>), dim3(1), dim3(512), 0, 0, ab, expect, expect, bad, 64, 1);          // quiet run: expected values
    hipDeviceSynchronize();
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<CLS, SAME>), dim3(blocks), dim3(512), 0, 0, ab, expect, expect, bad, iters, 0);
    hipDeviceSynchronize();
    unsigned h[64];
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
    printf("%-34s %-46s wrong results in lanes 0-15: %8lu  16-31: %8lu  32-47: %8lu  48-63: %8lu   (of %.3g per lane group)\n", name,
           SAME ? "same wave, right behind 4 fp16 MFMAs" : "partner wave on the SIMD of an MFMA-issuing wave", q[0], q[1], q[2], q[3],
           5.0 * blocks * iters * 4 * 16);
}

int main() {
    std::vector<_Float16> h(1024);
    unsigned s = 5;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 9) & 0xFFFF) / 65536.f - 0.5f); }
    _Float16* ab;
    float* expect;
    unsigned* bad;
    hipMalloc(&ab, 2048); hipMalloc(&expect, 64 * 256 * 4); hipMalloc(&bad, 256);
    hipMemcpy(ab, h.data(), 2048, hipMemcpyHostToDevice);
    run<0, 1>("v_exp_f32 + v_rcp_f32 (sigmoid)", ab, expect, bad);   run<0, 0>("v_exp_f32 + v_rcp_f32 (sigmoid)", ab, expect, bad);
    run<1, 1>("v_pk_fma_f32 + v_pk_mul_f32", ab, expect, bad);       run<1, 0>("v_pk_fma_f32 + v_pk_mul_f32", ab, expect, bad);
    run<2, 1>("v_fma_f32 chain", ab, expect, bad);                   run<2, 0>("v_fma_f32 chain", ab, expect, bad);
    run<3, 1>("v_cvt_pk_f16_f32", ab, expect, bad);                  run<3, 0>("v_cvt_pk_f16_f32", ab, expect, bad);
    run<4, 1>("ds_read_b128 + blend", ab, expect, bad);              run<4, 0>("ds_read_b128 + blend", ab, expect, bad);
    run<5, 1>("v_mov_b64 copies", ab, expect, bad);                  run<5, 0>("v_mov_b64 copies", ab, expect, bad);
    run<6, 1>("v_mov_b32 dpp quad_perm", ab, expect, bad);           run<6, 0>("v_mov_b32 dpp quad_perm", ab, expect, bad);
    run<7, 1>("v_pk_* with op_sel / neg / 1.0", ab, expect, bad);   run<7, 0>("v_pk_* with op_sel / neg / 1.0", ab, expect, bad);
    run<8, 1>("trans -> v_pk_mul (half a pair)", ab, expect, bad);  run<8, 0>("trans -> v_pk_mul (half a pair)", ab, expect, bad);
    return 0;
}
