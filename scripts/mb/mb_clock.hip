// diagnostic: what does s_memtime count?  scratch tool
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out)
{
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
    do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); } while (t1 - t0 < ticks);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    out[0] = t1 - t0; out[1] = r1 - r0;
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 24000000ull, d);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("s_memtime ticks %llu, s_memrealtime ticks %llu, elapsed %.3f ms -> memtime %.1f MHz, realtime %.1f MHz\n", h[0], h[1], ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
    }
    return 0;
}
