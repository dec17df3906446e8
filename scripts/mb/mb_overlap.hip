// diagnostic: do independent v_fmac_f64_dpp issue in the latency shadow of a dependent FP64 chain?  One wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T> __device__ __forceinline__ unsigned long long now(T& dep) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory"); return t; }
#define REP 32
// K independent fmacs per chain iteration, one after each chain link (round robin)
template <int K, int DPP> __global__ void k(double* out, unsigned long long* ts, double x0, double eps, int slot)
{
    double x = x0 + threadIdx.x * 1e-3, src = x * 0.25, nt0 = x * 1e-3;
    double v[16];
    for (int q = 0; q < 16; q++) v[q] = x + q;
    unsigned long long t0 = now(x);
#pragma unroll
    for (int r = 0; r < REP; r++) {
        double D, rc, e;
        int used = 0;
        auto fill = [&](int n) {
#pragma unroll
            for (int q = 0; q < n; q++) if (used < K) {
                if (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(v[used & 15]) : "v"(src), "v"(nt0));
                else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v[used & 15]) : "v"(src), "v"(nt0));
                used++;
            }
        };
        asm volatile("v_max_f64 %0, %1, |%2|" : "=v"(D) : "v"(eps), "v"(x)); fill(2);
        asm volatile("v_rcp_f64 %0, %1" : "=v"(rc) : "v"(D)); fill(3);
        asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(D), "v"(rc)); fill(2);
        asm volatile("v_fmac_f64 %0, %1, %0" : "+v"(rc) : "v"(e)); fill(2);
        asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(e) : "v"(x), "v"(rc)); fill(2);
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(src), "v"(e)); fill(K);
    }
    unsigned long long t1 = now(x);
    if (threadIdx.x == 0) ts[slot] = t1 - t0;
    double acc = x; for (int q = 0; q < 16; q++) acc += v[q];
    out[threadIdx.x] = acc;
}
int main()
{
    double* out; unsigned long long* ts;
    hipMalloc(&out, 64 * 8); hipMalloc(&ts, 32 * 8);
    unsigned long long h[32];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13, 0);
        hipLaunchKernelGGL((k<4, 1>), dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13, 1);
        hipLaunchKernelGGL((k<8, 1>), dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13, 2);
        hipLaunchKernelGGL((k<11, 1>), dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13, 3);
        hipLaunchKernelGGL((k<16, 1>), dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13, 4);
        hipLaunchKernelGGL((k<8, 0>), dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13, 5);
        hipLaunchKernelGGL((k<16, 0>), dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13, 6);
        hipDeviceSynchronize();
        hipMemcpy(h, ts, sizeof h, hipMemcpyDeviceToHost);
    }
    const char* nm[] = { "chain only", "+4 fmac_dpp in the slots", "+8 fmac_dpp", "+11 fmac_dpp", "+16 fmac_dpp (5 after the chain)", "+8 plain fmac", "+16 plain fmac" };
    for (int i = 0; i < 7; i++) printf("%-36s %6.1f cycles/iteration\n", nm[i], h[i] / (double)REP);
    return 0;
}
