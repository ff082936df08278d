// Does a short command sequence on one HIP stream wait for a long kernel on ANOTHER stream of the same host thread?  (round 5: the pipelined lock-step
// replay -- host side of one group against the kernels of another on the thread's second stream -- was slower than the plain one, with the second
// group's small round trips taking as long as the first group's solve kernel.)   hipcc --offload-arch=gfx950 -O2 stream_overlap.hip -o stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long ticks, int *out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1; }
__global__ void spin_scratch(long long ticks, double *out, int n) {      // the same with a private array that lives in scratch memory
    double a[200];
    for (int i = 0; i < 200; i++) a[i] = i * 0.5 + threadIdx.x;
    const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {}
    double s = 0; for (int i = 0; i < 200; i++) s += a[(i * 7 + n) % 200];
    if (out && s == 12345.678) *out = s;
}
__global__ void tiny(int *p) { if (threadIdx.x == 0) p[blockIdx.x] = blockIdx.x; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const long long ticks = (long long)khz * 2;      // 2 ms
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    int *d, *h; double *dd; hipMalloc(&d, 4096); hipMalloc(&dd, 4096); hipHostMalloc(&h, 4096);
    for (int variant = 0; variant < 4; variant++) {
        const char *name[4] = {"long kernel: 32 workgroups, no scratch, no LDS", "long kernel: 32 workgroups with 160 KiB of LDS each", "long kernel: 32 workgroups with scratch", "long kernel: 256 workgroups with 160 KiB of LDS each"};
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            const double t0 = now_ms();
            if (variant == 0) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, s1, ticks, (int *)nullptr);
            if (variant == 1) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 160 * 1024, s1, ticks, (int *)nullptr);
            if (variant == 2) hipLaunchKernelGGL(spin_scratch, dim3(32), dim3(256), 0, s1, ticks, (double *)nullptr, rep);
            if (variant == 3) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 160 * 1024, s1, ticks, (int *)nullptr);
            const double t1 = now_ms();
            hipMemcpyAsync(d, h, 1024, hipMemcpyHostToDevice, s2);
            hipLaunchKernelGGL(tiny, dim3(4), dim3(64), 0, s2, d);
            hipMemcpyAsync(h, d, 1024, hipMemcpyDeviceToHost, s2);
            hipStreamSynchronize(s2);
            const double t2 = now_ms();
            hipStreamSynchronize(s1);
            const double t3 = now_ms();
            printf("%-58s launch %.3f ms | round trip on the OTHER stream %.3f ms | long kernel done after %.3f ms\n", name[variant], t1 - t0, t2 - t1, t3 - t0);
        }
    }
    return 0;
}
