// diagnostic: issue cost of DPP64 / permlane-swap / readlane instruction mixes (one wave, gfx950).  scratch tool
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T> __device__ __forceinline__ unsigned long long now(T& dep) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory"); return t; }
template <int N> __device__ __forceinline__ void fmac_bcast16(double& acc, double src, double nt)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(nt), "n"(N));
}
__global__ void k(double* out, unsigned long long* ts, double x0)
{
    double x = x0 + threadIdx.x * 1e-9, nt = x * 0.001;
    double v[16];
    for (int q = 0; q < 16; q++) v[q] = x + q;
    unsigned long long t0, t1; int i = 0;
    // 1. 64 independent-ish fmac_dpp (16 accumulators x 4 rounds)
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        fmac_bcast16<0>(v[0], x, nt); fmac_bcast16<1>(v[1], x, nt); fmac_bcast16<2>(v[2], x, nt); fmac_bcast16<3>(v[3], x, nt);
        fmac_bcast16<4>(v[4], x, nt); fmac_bcast16<5>(v[5], x, nt); fmac_bcast16<6>(v[6], x, nt); fmac_bcast16<7>(v[7], x, nt);
        fmac_bcast16<8>(v[8], x, nt); fmac_bcast16<9>(v[9], x, nt); fmac_bcast16<10>(v[10], x, nt); fmac_bcast16<11>(v[11], x, nt);
        fmac_bcast16<12>(v[12], x, nt); fmac_bcast16<13>(v[13], x, nt); fmac_bcast16<14>(v[14], x, nt); fmac_bcast16<15>(v[15], x, nt);
    }
    t1 = now(v[15]); ts[i++] = t1 - t0;
    // 2. 64 plain v_fmac_f64 (same shape)
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 16; q++) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(v[q]) : "v"(x), "v"(nt));
    t1 = now(v[15]); ts[i++] = t1 - t0;
    // 3. 64 v_mov_b64_dpp
    double m[16];
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 16; q++) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(m[q]) : "v"(x));
    t1 = now(m[15]); ts[i++] = t1 - t0;
    // 4. 64 v_permlane16_swap
    unsigned a = threadIdx.x, b = threadIdx.x * 3;
    t0 = now(a);
#pragma unroll
    for (int r = 0; r < 64; r++) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    t1 = now(a); ts[i++] = t1 - t0;
    // 5. 64 fmac with SGPR multiplier (readlane once)
    double s = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 16; q++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[q]) : "s"(s), "v"(nt));
    t1 = now(v[15]); ts[i++] = t1 - t0;
    // 6. 64 x (2 readlane + fma)
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            int lo, hi;
            asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4" : "=s"(lo), "=s"(hi) : "v"(__double2loint(x)), "v"(__double2hiint(x)), "n"(5));
            double ss = __hiloint2double(hi, lo);
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[q]) : "s"(ss), "v"(nt));
        }
    t1 = now(v[15]); ts[i++] = t1 - t0;
    double acc = 0;
    for (int q = 0; q < 16; q++) acc += v[q] + m[q];
    out[threadIdx.x] = acc + a + b;
}
int main()
{
    double* out; unsigned long long* ts;
    hipMalloc(&out, 64 * 8); hipMalloc(&ts, 16 * 8);
    unsigned long long h[16];
    const char* nm[] = { "v_fmac_f64_dpp", "v_fmac_f64", "v_mov_b64_dpp", "v_permlane16_swap", "v_fma_f64 sgpr", "2 readlane + fma" };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, ts, 1.37);
        hipDeviceSynchronize();
        hipMemcpy(h, ts, sizeof h, hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 6; i++) printf("%-22s %6.1f cycles/op\n", nm[i], h[i] / 64.0);
    return 0;
}
