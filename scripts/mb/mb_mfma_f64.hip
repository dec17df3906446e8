// FP64 MFMA peak on this device: back-to-back v_mfma_f64_16x16x4_f64, 4 independent accumulators per wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k(double* out, int iters, double a0, double b0)
{
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int W> void run(int blocks, int iters)
{
    double* out; hipMalloc(&out, sizeof(double) * blocks * 64 * W);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64 * W), 0, 0, out, iters, 1.0, 1e-9);
    hipEventRecord(a); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64 * W), 0, 0, out, iters, 1.0, 1e-9); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * W * iters * 4 * 2048.0;
    printf("blocks %d x %d waves, %d iters: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz if 1 wave per SIMD)\n", blocks, W, iters, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * 4.0 * (blocks * W / 1024.0)));
    hipFree(out);
}
int main() { run<4>(256, 20000); run<4>(512, 20000); run<8>(256, 20000); run<4>(1024, 10000); return 0; }
