// FP64 peak of the device this library runs on, measured, not quoted ("FP64 peak is the AMD datasheet figure, not in
// the local guides; microbench it").  Two kernels, every CU filled with 8 wavefronts per SIMD:
//   * dependent-free chains of v_fma_f64 (16 independent accumulators per lane),
//   * dependent-free chains of v_mfma_f64_16x16x4_f64 (8 independent 16 x 16 accumulator tiles per wavefront),
// each timed with HIP events over a launch long enough (~2 ms) for the launch overhead to vanish.  bench.py calls it on the box the
// headline is measured on and prices the solve kernel's FP64 rate against the better of the two (`roofline_fp64`).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../include/tcv.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

template <int ITERS>
__global__ void __launch_bounds__(256) fma_f64_kernel(double *out, double a0, double b0) {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (double)(threadIdx.x + i);
    const double a = a0, b = b0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    if (s == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = s;      // (never true for these inputs: keeps the chains alive)
}

template <int ITERS>
__global__ void __launch_bounds__(256) mfma_f64_kernel(double *out, double a0, double b0) {
    v4d acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
int timed(F launch, double *ms_out) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return TCV_ERR_HIP;
    launch();      // warm-up (code object load, clocks)
    if (hipDeviceSynchronize() != hipSuccess) return TCV_ERR_HIP;
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, nullptr);
        launch();
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess) return TCV_ERR_HIP;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_out = best;
    return TCV_OK;
}

}  // namespace

// out[0] = v_fma_f64 TFLOP/s, out[1] = v_mfma_f64_16x16x4 TFLOP/s, out[2] = CUs, out[3] = shader clock the runtime reports [MHz]
extern "C" int tcv_microbench_fp64(double *out4) {
    if (!out4) return TCV_ERR_INVALID;
    int dev = 0, n_cu = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TCV_ERR_NO_DEVICE;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return TCV_ERR_HIP;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    double *d = nullptr;
    constexpr int ITERS = 4096;
    const int grid = n_cu * 8;      // 8 workgroups x 4 wavefronts per CU = 8 wavefronts per SIMD
    if (hipMalloc(&d, sizeof(double) * (size_t)grid * 256) != hipSuccess) return TCV_ERR_HIP;
    double ms_f = 0, ms_m = 0;
    int rc = timed([&] { hipLaunchKernelGGL(fma_f64_kernel<ITERS>, dim3(grid), dim3(256), 0, nullptr, d, 0.999999, 1e-7); }, &ms_f);
    if (rc == TCV_OK) rc = timed([&] { hipLaunchKernelGGL(mfma_f64_kernel<ITERS>, dim3(grid), dim3(256), 0, nullptr, d, 0.5, 0.25); }, &ms_m);
    (void)hipFree(d);
    if (rc != TCV_OK) return rc;
    const double fl_f = 2.0 * 16 * 4 * (double)ITERS * 256.0 * grid;                    // FMA = 2 flops
    const double fl_m = 2.0 * 16 * 16 * 4 * 8 * 4 * (double)ITERS * 4.0 * grid;          // per wavefront and instruction 16 x 16 x 4 FMAs
    out4[0] = fl_f / (ms_f * 1e-3) / 1e12;
    out4[1] = fl_m / (ms_m * 1e-3) / 1e12;
    out4[2] = n_cu;
    out4[3] = khz / 1000.0;
    return TCV_OK;
}
