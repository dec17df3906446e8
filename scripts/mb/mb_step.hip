// diagnostic: where does one k_gmw_step spend its cycles (block (0,0), wave 0)?  scratch tool
#define SRUKF_STAMPS 1
#include "../../cv-monoslam_amd/csrc/srukf_factor.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int n = 1204, ld = 1216;
    std::vector<double> h((size_t)ld * ld);
    for (int r = 0; r < ld; r++) for (int c = 0; c < ld; c++) h[(size_t)r * ld + c] = (r == c) ? 2.0 + 0.001 * r : 0.3 / (1 + abs(r - c));
    double *G, *D, *S; GmwPanel* pan[2];
    hipMalloc(&G, sizeof(double) * ld * ld); hipMalloc(&S, sizeof(double) * ld * ld); hipMalloc(&D, sizeof(double) * ld);
    hipMalloc(&pan[0], sizeof(GmwPanel)); hipMalloc(&pan[1], sizeof(GmwPanel));
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(G, h.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice);
        hipEventRecord(a, st);
        srukf_launch_gmw_first(st, n, ld, 1e-13, G, pan[0], D, S);
        int pb = 0; 
        for (int j0 = 0; j0 + 32 < ld; j0 += 32, pb ^= 1) srukf_launch_gmw_step(st, n, ld, j0, 1e-13, G, pan[pb], pan[pb ^ 1], D, S, nullptr);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("whole factorisation: %.1f us (%d launches)\n", ms * 1000, 1 + (ld / 32 - 1));
    }
    // stamps of ONE step in the middle (j0 = 512)
    hipMemcpy(G, h.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice);
    srukf_launch_gmw_first(st, n, ld, 1e-13, G, pan[0], D, S);
    int pb = 0;
    for (int j0 = 0; j0 <= 512; j0 += 32, pb ^= 1) srukf_launch_gmw_step(st, n, ld, j0, 1e-13, G, pan[pb], pan[pb ^ 1], D, S, nullptr);
    hipStreamSynchronize(st);
    unsigned long long hs[16];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(srukf_stamps), sizeof hs);
    const char* nm[] = { "start->prefetch issued", "phase A: quarter slab -> LDS", "barrier", "phase B: quarter update + exchange", "factor + outputs" };
    for (int i = 0; i < 5; i++) printf("%-28s %6llu ticks (100 MHz realtime? s_memtime = shader clock)\n", nm[i], hs[i + 1] - hs[i]);
    printf("total in-kernel %llu ticks\n", hs[5] - hs[0]);

    return 0;
}
