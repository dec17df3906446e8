// diagnostic: which link of the GMW pivot chain is slow?  One wave, dependent iterations; variants drop one link each.
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T> __device__ __forceinline__ unsigned long long now(T& dep) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory"); return t; }
#define REP 32
template <int V> __global__ void k(double* out, unsigned long long* ts, double x0, double eps)
{
    __shared__ double lds[256];
    double x = x0 + threadIdx.x * 1e-3, acc = x * 0.5, src = x;
    const unsigned la = threadIdx.x * 8;
    unsigned long long t0 = now(x);
#pragma unroll
    for (int r = 0; r < REP; r++) {
        double d, D, rc, nt;
        if (V == 1) d = x;                                      // no DPP broadcast
        else asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(x));
        if (V == 2) D = d; else asm volatile("v_max_f64 %0, %1, |%2|" : "=v"(D) : "v"(eps), "v"(d));
        if (V == 3) rc = D; else asm volatile("v_rcp_f64 %0, %1" : "=v"(rc) : "v"(D));
        if (V != 4) {
            double e;
            asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(D), "v"(rc));
            asm volatile("v_fmac_f64 %0, %1, %0" : "+v"(rc) : "v"(e));
            asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(D), "v"(rc));
            asm volatile("v_fmac_f64 %0, %1, %0" : "+v"(rc) : "v"(e));
        }
        asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(nt) : "v"(x), "v"(rc));
        if (V != 5) {
            asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(nt) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:2048" :: "v"(la), "v"(D) : "memory");
        }
        if (V == 6) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(src), "v"(nt));
        else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(nt));
        x = acc;
    }
    unsigned long long t1 = now(x);
    if (threadIdx.x == 0) ts[V] = t1 - t0;
    out[threadIdx.x] = x + lds[threadIdx.x];
}
int main()
{
    double* out; unsigned long long* ts;
    hipMalloc(&out, 64 * 8); hipMalloc(&ts, 16 * 8);
    unsigned long long h[16];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13);
        hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13);
        hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13);
        hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, out, ts, 1.37, 1e-13);
        hipDeviceSynchronize();
        hipMemcpy(h, ts, sizeof h, hipMemcpyDeviceToHost);
    }
    const char* nm[] = { "full chain", "no mov_dpp", "no max", "no rcp", "no Newton", "no LDS publish", "plain fmac instead of fmac_dpp" };
    for (int i = 0; i < 7; i++) printf("%-32s %6.1f cycles/pivot\n", nm[i], h[i] / (double)REP);
    return 0;
}
