// Host-only stand-in for the few HIP runtime names the CLIP SCHEDULER (csrc/generator.hip) uses, selected by -DPNP_HOST_STUB.
//
// Purpose: compile the 1.2 k lines of host C++ in generator.hip (workspace carving, key-frame selection, expert dedup, event
// pool, side streams) with a plain host compiler under AddressSanitizer / UBSan on the CPU box (GPU ASan is not available on
// this pool) and run the schedule against recording launchers (tests/host/sched_stub.cpp).  Nothing here is used by the
// product build; kernels are not executed -- a launch is a record.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidDevice = 101;
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;

struct pnp_stub_stream { int id; };
struct pnp_stub_event { int id; int recorded_on; };     // recorded_on: stream id, -1 = never recorded
typedef pnp_stub_stream* hipStream_t;
typedef pnp_stub_event* hipEvent_t;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;      // referenced by the two tiny __global__ helpers in generator.hip

// A kernel launch is recorded, not executed: name, stream and the arguments as tagged scalars (tests/host/sched_stub.cpp knows
// what each of the scheduler's own little kernels reads and writes).
struct PnpStubArg {
    int kind;            // 0 pointer, 1 floating point, 2 integer
    const void* p;
    double f;
    long long i;
};
template <class T>
inline PnpStubArg pnp_stub_arg(T* v) { return PnpStubArg{0, (const void*)v, 0.0, 0}; }
inline PnpStubArg pnp_stub_arg(float v) { return PnpStubArg{1, nullptr, (double)v, 0}; }
inline PnpStubArg pnp_stub_arg(double v) { return PnpStubArg{1, nullptr, v, 0}; }
inline PnpStubArg pnp_stub_arg(int v) { return PnpStubArg{2, nullptr, 0.0, v}; }
inline PnpStubArg pnp_stub_arg(long v) { return PnpStubArg{2, nullptr, 0.0, v}; }
inline PnpStubArg pnp_stub_arg(long long v) { return PnpStubArg{2, nullptr, 0.0, v}; }
void pnp_stub_kernel_launch_impl(const char* name, dim3 grid, dim3 block, hipStream_t stream, const PnpStubArg* args, int nargs);
template <class... A>
inline void pnp_stub_kernel_launch(const char* name, dim3 grid, dim3 block, hipStream_t stream, A... args) {
    const PnpStubArg v[] = {pnp_stub_arg(args)..., PnpStubArg{2, nullptr, 0.0, 0}};
    pnp_stub_kernel_launch_impl(name, grid, block, stream, v, (int)sizeof...(A));
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    pnp_stub_kernel_launch(#kernel, (grid), (block), (stream), ##__VA_ARGS__)

hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipGetLastError();
hipError_t hipGetDevice(int* dev);
hipError_t hipFuncSetAttribute(const void* fn, int attr, int value);
