// How long after it has issued does a 32x32x16 f16 MFMA still read SrcA / SrcB?  mfma_src_then_valu.hip overwrites the sources 0-3 wait
// states behind the MFMA and finds nothing; the fp16 DCN kernel (tools/repro/dcn_f16_listing_bisect.py) goes right when an s_nop is put
// behind two MFMAs whose SrcB is overwritten 7-8 v_mov_b64 later by the accumulator copies of its run-time-ordered loop.  This sweep
// puts K wait states (0 .. 96) between the MFMA and the overwrite, with 0 / 1 / 2 independent MFMAs issued right in front of it.
// Operands are all ones, the overwriting value is NaN: an output that is not exactly 16 is a late read.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_src_window mfma_src_window.hip && ./mfma_src_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define H1 0x3c003c00u

template <int NQ, int K, int WHAT>      // WHAT 0: v_mov_b64 v[106:107] (SrcB hi)  1: v_mov_b64 v[104:105] (SrcB lo)  2: SrcA hi  3: v_mov_b32 v107
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int iters) {
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x16 d;
        const unsigned one = H1, nan = 0x7fc07e00u;
        asm volatile(".irp r,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115\n\tv_mov_b32 v\\r, %[one]\n\t.endr\n\t"
                     "v_mov_b32 v116, %[nan]\n\tv_mov_b32 v117, %[nan]\n\ts_nop 15\n\ts_nop 15\n\t"
                     ".if %c[nq] > 1\n\tv_mfma_f32_32x32x16_f16 v[136:151], v[108:111], v[112:115], 0\n\t.endif\n\t"
                     ".if %c[nq] > 0\n\tv_mfma_f32_32x32x16_f16 v[120:135], v[108:111], v[112:115], 0\n\t.endif\n\t"
                     "v_mfma_f32_32x32x16_f16 %[d], v[100:103], v[104:107], 0\n\t"
                     ".rept %c[k]\n\ts_nop 0\n\t.endr\n\t"
                     ".if %c[what] == 0\n\tv_mov_b64 v[106:107], v[116:117]\n\t.endif\n\t"
                     ".if %c[what] == 1\n\tv_mov_b64 v[104:105], v[116:117]\n\t.endif\n\t"
                     ".if %c[what] == 2\n\tv_mov_b64 v[102:103], v[116:117]\n\t.endif\n\t"
                     ".if %c[what] == 3\n\tv_mov_b32 v107, v116\n\t.endif\n\t"
                     "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
                     : [d] "=&v"(d)
                     : [one] "v"(one), [nan] "v"(nan), [nq] "n"(NQ), [k] "n"(K), [what] "n"(WHAT)
                     : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
                       "v114", "v115", "v116", "v117", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",
                       "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143",
                       "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "memory");
        bool ok = true;
        for (int k = 0; k < 16; ++k) ok &= d[k] == 16.f;
        bad += !ok;
    }
    out[blockIdx.x * 256 + threadIdx.x] = bad;
}

typedef void (*kern_t)(unsigned*, int);
template <int NQ, int WHAT>
static void sweep(const char* what, unsigned* out, std::vector<unsigned>& h) {
    static const kern_t ks[] = {probe<NQ, 0, WHAT>,  probe<NQ, 1, WHAT>,  probe<NQ, 2, WHAT>,  probe<NQ, 3, WHAT>,  probe<NQ, 4, WHAT>,  probe<NQ, 6, WHAT>,
                                probe<NQ, 8, WHAT>,  probe<NQ, 12, WHAT>, probe<NQ, 16, WHAT>, probe<NQ, 20, WHAT>, probe<NQ, 24, WHAT>, probe<NQ, 28, WHAT>,
                                probe<NQ, 32, WHAT>, probe<NQ, 40, WHAT>, probe<NQ, 48, WHAT>, probe<NQ, 56, WHAT>, probe<NQ, 64, WHAT>, probe<NQ, 80, WHAT>,
                                probe<NQ, 96, WHAT>};
    printf("%-64s", what);
    for (kern_t k : ks) {
        (void)hipMemset(out, 0xff, h.size() * 4);
        hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, out, 16);
        if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return; }
        (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long bad = 0;
        for (unsigned v : h) bad += v;
        printf(" %7llu", bad);
    }
    printf("\n");
}

int main() {
    unsigned* out;
    std::vector<unsigned> h(512 * 256);
    if (hipMalloc(&out, h.size() * 4) != hipSuccess) return 1;
    printf("131072 threads x 16 MFMAs per cell; wrong lanes when the source register is overwritten K wait states behind v_mfma_f32_32x32x16_f16\n");
    printf("%-64s", "MFMAs in front ; overwritten operand \\ K =");
    for (int k : {0, 1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96}) printf(" %7d", k);
    printf("\n");
    sweep<0, 0>("none ; v_mov_b64 v[106:107] (SrcB, high pair)", out, h);
    sweep<0, 1>("none ; v_mov_b64 v[104:105] (SrcB, low pair)", out, h);
    sweep<0, 2>("none ; v_mov_b64 v[102:103] (SrcA, high pair)", out, h);
    sweep<0, 3>("none ; v_mov_b32 v107", out, h);
    sweep<1, 0>("1 MFMA in front ; v_mov_b64 v[106:107] (SrcB, high pair)", out, h);
    sweep<1, 1>("1 MFMA in front ; v_mov_b64 v[104:105] (SrcB, low pair)", out, h);
    sweep<1, 2>("1 MFMA in front ; v_mov_b64 v[102:103] (SrcA, high pair)", out, h);
    sweep<2, 0>("2 MFMAs in front ; v_mov_b64 v[106:107] (SrcB, high pair)", out, h);
    sweep<2, 1>("2 MFMAs in front ; v_mov_b64 v[104:105] (SrcB, low pair)", out, h);
    sweep<2, 2>("2 MFMAs in front ; v_mov_b64 v[102:103] (SrcA, high pair)", out, h);
    return 0;
}
