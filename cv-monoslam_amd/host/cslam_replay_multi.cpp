// cslam_replay_multi — native multi-GPU replay host: independent sequences (Monte-Carlo runs of one map) sharded one per GPU of a node, RCCL over xGMI for
// the two exchanges the path has (BASELINE.json north_star): a broadcast of the shared initial map (X0: n doubles, S0: n x n doubles) from device 0
// before the first frame and an all-gather of the trajectories (F x 8 doubles per device) after the last; nothing per frame.  One process, one host thread
// and one srukf_ctx per device (a context is not thread-safe, independent contexts are fully concurrent: include/srukf.h), single-process RCCL
// (ncclCommInitAll + group calls).  The reference has no counterpart: its host runs ONE CSLAM on the UI thread (MonoSLAMView.cpp:526-572 is the loop a
// Monte-Carlo run repeats per sequence).  bench.py --gpus N is the same split with one PROCESS per GPU through torch.distributed; this is the C++ product's own.
//   cslam_replay_multi scene.bin odometry.txt devices=0,1,... [frames=K] [warmup=W] [traj=out.bin]
// scene.bin: int32 N, int32 F, double a1..a4, double X0[n], double S0[n*n], double z[F][2N]   (the file cslam_replay reads); device d > 0 adds its own
// measurement noise to z (0.5 px, deterministic per device), device 0 replays the file's z unchanged.
// Prints ONE JSON line with bench.py's keys (metric, value, n_gpus, rccl_world_size, per_rank_frames_per_s, slowest_rank, map_broadcast, ...).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/srukf.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Barrier {
    std::mutex m; std::condition_variable cv; int n, waiting = 0, gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() { std::unique_lock<std::mutex> lk(m); const int g = gen; if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g; }); }
};

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLOK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s scene.bin odometry.txt devices=0,1,... [frames=K] [warmup=W] [traj=out.bin]\n", argv[0]); return 2; }
    std::vector<int> devs;
    int K = 100, W = 10;
    std::string traj_out;
    for (int a = 3; a < argc; a++) {
        if (!strncmp(argv[a], "devices=", 8)) { for (char* t = strtok(argv[a] + 8, ","); t; t = strtok(nullptr, ",")) devs.push_back(atoi(t)); }
        else if (!strncmp(argv[a], "frames=", 7)) K = atoi(argv[a] + 7);
        else if (!strncmp(argv[a], "warmup=", 7)) W = atoi(argv[a] + 7);
        else if (!strncmp(argv[a], "traj=", 5)) traj_out = argv[a] + 5;
    }
    if (devs.empty()) devs.push_back(0);
    const int G = (int)devs.size();
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int N = 0, F = 0; double a4[4];
    if (fread(&N, 4, 1, f) != 1 || fread(&F, 4, 1, f) != 1 || fread(a4, 8, 4, f) != 4) return 2;
    const int n = 6 * N + 4;
    std::vector<double> X0(n), S0((size_t)n * n), z((size_t)F * 2 * N);
    if (fread(X0.data(), 8, n, f) != (size_t)n || fread(S0.data(), 8, (size_t)n * n, f) != (size_t)n * n || fread(z.data(), 8, z.size(), f) != z.size()) { fprintf(stderr, "short scene file\n"); return 2; }
    fclose(f);
    if (W + K > F) { fprintf(stderr, "scene has %d frames, need %d\n", F, W + K); return 2; }
    std::vector<double> odo;
    {
        FILE* o = fopen(argv[2], "r");
        if (!o) { perror(argv[2]); return 2; }
        char line[500];
        while (fgets(line, sizeof line, o)) { int id; double x, y, th; if (sscanf(line, "%d : %*f %lf %lf %lf", &id, &x, &y, &th) == 4) { odo.push_back(x); odo.push_back(y); odo.push_back(th); } }
        fclose(o);
    }
    if ((int)odo.size() / 3 < F + 1) { fprintf(stderr, "odometry has %d poses, need %d\n", (int)odo.size() / 3, F + 1); return 2; }
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    for (int d : devs) if (d < 0 || d >= ndev) { fprintf(stderr, "device %d: this node has %d\n", d, ndev); return 2; }

    // ---- RCCL: one communicator per device, the map broadcast from device devs[0] ----
    std::vector<ncclComm_t> comms(G);
    NCCLOK(ncclCommInitAll(comms.data(), G, devs.data()));
    std::vector<hipStream_t> cstream(G);
    std::vector<double*> dX(G), dS(G), dtraj(G), dall(G);
    const size_t Ftot = (size_t)(W + K);
    for (int g = 0; g < G; g++) {
        HIPOK(hipSetDevice(devs[g]));
        HIPOK(hipStreamCreateWithFlags(&cstream[g], hipStreamNonBlocking));
        HIPOK(hipMalloc((void**)&dX[g], sizeof(double) * n)); HIPOK(hipMalloc((void**)&dS[g], sizeof(double) * (size_t)n * n));
        HIPOK(hipMalloc((void**)&dtraj[g], sizeof(double) * 8 * Ftot)); HIPOK(hipMalloc((void**)&dall[g], sizeof(double) * 8 * Ftot * G));
        HIPOK(hipMemset(dtraj[g], 0, sizeof(double) * 8 * Ftot));
        if (g == 0) { HIPOK(hipMemcpy(dX[0], X0.data(), sizeof(double) * n, hipMemcpyHostToDevice)); HIPOK(hipMemcpy(dS[0], S0.data(), sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice)); }
        else { HIPOK(hipMemset(dX[g], 0, sizeof(double) * n)); HIPOK(hipMemset(dS[g], 0, sizeof(double) * (size_t)n * n)); }
    }
    const double tb0 = now_s();
    NCCLOK(ncclGroupStart());
    for (int g = 0; g < G; g++) { NCCLOK(ncclBroadcast(dX[g], dX[g], n, ncclDouble, 0, comms[g], cstream[g])); NCCLOK(ncclBroadcast(dS[g], dS[g], (size_t)n * n, ncclDouble, 0, comms[g], cstream[g])); }
    NCCLOK(ncclGroupEnd());
    for (int g = 0; g < G; g++) { HIPOK(hipSetDevice(devs[g])); HIPOK(hipStreamSynchronize(cstream[g])); }
    const double bcast_ms = (now_s() - tb0) * 1e3;

    // ---- one host thread and one filter per device ----
    srukf_params p;
    srukf_default_params(&p);
    p.a1 = a4[0]; p.a2 = a4[1]; p.a3 = a4[2]; p.a4 = a4[3];
    std::vector<double> walls(G, 0.0);
    std::vector<int> rcs(G, 0), nulls(G, 0);
    std::vector<std::string> errs(G);
    Barrier bar(G);
    auto worker = [&](int g) {
        srukf_ctx* c = nullptr;
        int rc = srukf_create(&c, N, &p, devs[g], nullptr);
        if (rc) { rcs[g] = rc; errs[g] = srukf_last_error(nullptr); }
        std::vector<double> zg(z.begin(), z.begin() + Ftot * 2 * N);
        if (g > 0) {                                           // own measurement stream: + N(0, 0.5^2) px, deterministic per device (sum of 12 uniforms)
            unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(g + 1);
            for (auto& v : zg) { double u = 0; for (int q = 0; q < 12; q++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; u += (double)(s >> 11) / 9007199254740992.0; } v += 0.5 * (u - 6.0); }
        }
        std::vector<int> m(Ftot * N, 1);
        if (!rc) rc = srukf_set_state_device(c, dX[g], dS[g], n);
        if (!rc) rc = srukf_stage_sequence(c, (int)Ftot, odo.data(), zg.data(), m.data());
        if (!rc) rc = srukf_prepare_frames(c, K);
        if (!rc && W > 0) { rc = srukf_run_frames_async(c, 0, W, SRUKF_UPDATE_BATCHED, dtraj[g]); if (!rc) rc = srukf_synchronize(c); }
        if (rc && !rcs[g]) { rcs[g] = rc; errs[g] = c ? srukf_last_error(c) : ""; }
        bar.wait();                                            // every device starts its timed block together (the barrier of bench.py's contract)
        const double t0 = now_s();
        if (!rcs[g]) { rc = srukf_run_frames_async(c, W, K, SRUKF_UPDATE_BATCHED, dtraj[g] + (size_t)8 * W); if (!rc) rc = srukf_synchronize(c); if (rc) { rcs[g] = rc; errs[g] = srukf_last_error(c); } }
        walls[g] = now_s() - t0;
        bar.wait();
        if (c) { nulls[g] = srukf_null_directions(c); srukf_destroy(c); }
    };
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) th.emplace_back(worker, g);
    for (auto& t : th) t.join();
    for (int g = 0; g < G; g++) if (rcs[g]) { fprintf(stderr, "device %d: srukf error %d: %s\n", devs[g], rcs[g], errs[g].c_str()); return 1; }

    // ---- all-gather of the trajectories (end of run) ----
    NCCLOK(ncclGroupStart());
    for (int g = 0; g < G; g++) NCCLOK(ncclAllGather(dtraj[g], dall[g], 8 * Ftot, ncclDouble, comms[g], cstream[g]));
    NCCLOK(ncclGroupEnd());
    for (int g = 0; g < G; g++) { HIPOK(hipSetDevice(devs[g])); HIPOK(hipStreamSynchronize(cstream[g])); }
    std::vector<double> all((size_t)8 * Ftot * G), own((size_t)8 * Ftot);
    HIPOK(hipSetDevice(devs[0]));
    HIPOK(hipMemcpy(all.data(), dall[0], sizeof(double) * all.size(), hipMemcpyDeviceToHost));
    bool gathered_ok = true;                                   // what device 0 gathered from device g is what device g computed
    for (int g = 0; g < G; g++) {
        HIPOK(hipSetDevice(devs[g]));
        HIPOK(hipMemcpy(own.data(), dtraj[g], sizeof(double) * own.size(), hipMemcpyDeviceToHost));
        gathered_ok = gathered_ok && memcmp(own.data(), all.data() + (size_t)8 * Ftot * g, sizeof(double) * own.size()) == 0;
    }
    if (!traj_out.empty()) { FILE* o = fopen(traj_out.c_str(), "wb"); if (o) { fwrite(all.data(), 8, all.size(), o); fclose(o); } }
    double wmax = 0; int slow = 0;
    for (int g = 0; g < G; g++) if (walls[g] > wmax) { wmax = walls[g]; slow = g; }
    int ver = 0; ncclGetVersion(&ver);
    printf("{\"metric\": \"srukf_updates_per_sec\", \"value\": %.2f, \"unit\": \"frames/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.5f, \"higher_is_better\": true, "
           "\"scaling\": \"weak\", \"dtype\": \"f64\", \"data\": \"synthetic\", \"host\": \"cslam_replay_multi (C++, one thread and one srukf_ctx per device, single-process RCCL)\", "
           "\"collectives\": \"rccl\", \"rccl_version\": %d, \"rccl_world_size\": %d, \"landmarks\": %d, \"null_directions_skipped\": %d, \"per_rank_frames_per_s\": [",
           G * K / wmax, G, K, W, wmax / K * 1e3, ver, G, N, nulls[0]);
    for (int g = 0; g < G; g++) printf("%s%.2f", g ? ", " : "", K / walls[g]);
    printf("], \"slowest_rank\": %d, \"map_broadcast\": {\"bytes\": %zu, \"ms\": %.3f}, \"trajectory_allgather_ok\": %s, \"final_pose_per_rank\": [", slow, sizeof(double) * ((size_t)n + (size_t)n * n), bcast_ms,
           gathered_ok ? "true" : "false");
    for (int g = 0; g < G; g++) { const double* r = all.data() + (size_t)8 * Ftot * g + 8 * (Ftot - 1); printf("%s[%.12g, %.12g, %.12g, %.12g]", g ? ", " : "", r[0], r[1], r[2], r[3]); }
    printf("]}\n");
    for (int g = 0; g < G; g++) { hipSetDevice(devs[g]); hipFree(dX[g]); hipFree(dS[g]); hipFree(dtraj[g]); hipFree(dall[g]); hipStreamDestroy(cstream[g]); ncclCommDestroy(comms[g]); }
    return gathered_ok ? 0 : 1;
}
