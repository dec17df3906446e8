// One wave per SIMD: cycles per v_mfma_f64_16x16x4_f64 as a function of the number of independent accumulators (dependent distance)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0, unsigned long long* cyc)
{
    d4 c[NACC];
#pragma unroll
    for (int q = 0; q < NACC; q++) c[q] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < NACC; q++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; q++) s += c[q][q & 3];
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC> void run(int blocks, int iters)
{
    double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9, cyc);
    hipEventRecord(a); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9, cyc); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d accumulators, %d blocks of 4 waves (1 wave per SIMD), %d MFMAs per wave: %.3f ms, %.1f TFLOP/s, %.1f shader cycles per MFMA (s_memtime), %.1f ns per MFMA\n", NACC, blocks, iters * NACC, ms,
           (double)blocks * 4 * iters * NACC * 2048.0 / ms / 1e9, (double)h / (iters * (double)NACC), ms * 1e6 / (iters * (double)NACC));
    hipFree(out); hipFree(cyc);
}
int main() { run<1>(256, 40000); run<2>(256, 20000); run<4>(256, 10000); run<8>(256, 5000); run<16>(256, 2500); return 0; }
