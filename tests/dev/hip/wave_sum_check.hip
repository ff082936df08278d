// Developer check (GPU): the VALU-only wave reduction (v_permlane32_swap, v_permlane16_swap, DPP row_shl) against the __shfl_down tree it
// replaces -- lane 0 must get the same bits.   hipcc --offload-arch=gfx950 -O3 tests/dev/hip/wave_sum_check.hip -o /tmp/wsc && /tmp/wsc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__device__ __forceinline__ double from_halves(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ double down32(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return from_halves(a[1], b[1]);
}
__device__ __forceinline__ double down16(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return from_halves(a[1], b[1]);
}
template <int CTRL>
__device__ __forceinline__ double down_dpp(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const unsigned a = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false), b = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return from_halves(a, b);
}
__device__ __forceinline__ double wave_sum_valu(double v) {
    v += down32(v); v += down16(v);
    v += down_dpp<0x108>(v); v += down_dpp<0x104>(v); v += down_dpp<0x102>(v); v += down_dpp<0x101>(v);
    return v;
}
__global__ void k(const double *in, double *o1, double *o2) {
    double v = in[blockIdx.x * 64 + threadIdx.x], w = v;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    w = wave_sum_valu(w);
    if (threadIdx.x == 0) { o1[blockIdx.x] = v; o2[blockIdx.x] = w; }
}
int main() {
    const int nb = 4096;
    double *h = (double *)malloc(sizeof(double) * nb * 64);
    srand(7);
    for (int i = 0; i < nb * 64; i++) h[i] = ((double)rand() / RAND_MAX - 0.5) * ((i % 7) ? 1.0 : 1e9) + ((i % 13) ? 0.0 : 1e-9);
    double *d, *o1, *o2;
    hipMalloc(&d, sizeof(double) * nb * 64); hipMalloc(&o1, sizeof(double) * nb); hipMalloc(&o2, sizeof(double) * nb);
    hipMemcpy(d, h, sizeof(double) * nb * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d, o1, o2);
    double *a = (double *)malloc(sizeof(double) * nb), *b = (double *)malloc(sizeof(double) * nb);
    hipMemcpy(a, o1, sizeof(double) * nb, hipMemcpyDeviceToHost); hipMemcpy(b, o2, sizeof(double) * nb, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < nb; i++) if (memcmp(&a[i], &b[i], 8)) bad++;
    printf("wave_sum_valu vs __shfl_down tree: %d of %d sums differ\n", bad, nb);
    return bad != 0;
}
