// cslam_step_bench — wall-clock frame rate of the DROP-IN path: what a host that binds this library the way the MFC view binds the reference's
// CSLAM gets per frame (SLAM.cpp:87-112, called from MonoSLAMView.cpp:499-572): predict, host (or device) data association, update — one frame at a
// time, with the host round trips in between.  bench.py's headline is the staged replay (inputs resident in HBM, no host in the loop); this is the other number.
//   cslam_step_bench scene.bin odometry.txt mode=<capi|facade|assoc> [frames=K] [warmup=W] [hint=0|1] [set=key:value ...]
//     capi   : srukf_predict_motion -> srukf_predict_measurement (D->H h, Si, visible) -> host association (matched = visible, z from the scene)
//              -> srukf_update (H->D) -> srukf_get_robot (the pose and robot block RobotPath.txt records, SLAM.cpp:3539-3556)
//     facade : monoslam::CSLAM::SLAM() with the same association as a callback (what cslam_replay does), display refresh and mirrors included
//     assoc  : capi with srukf_associate (wrapPatch + dataAssociation on the device, SLAM.cpp:1803-2009) on a 640 x 480 gray frame between predict and
//              update.  The frame is a static synthetic texture the landmarks' init patches were cut from, so the templates stop correlating as the
//              robot moves: the launches and their D->H copy are what is timed, the filter itself is driven by the scene's z / matched (stated in the output).
//     hint=1 : capi / assoc announce the NEXT frame's odometry with srukf_predict_motion_next before every update (a host that has its odometry
//              file loaded, as the reference has: loadOdometryData reads it whole, SLAM.cpp:363-496)
//     set=key:value : srukf_debug_set(ctx, key, value) after the context exists (capi / assoc; measurement switches, e.g. set=step_fuse_export:0)
//     churn=P (facade): the map changes while the filter runs, through the reference's OWN policy (SLAM.cpp:2443-2460 deletions in updateFeaturesInformation,
//              552-562 additions through the addFeatures callback): every P frames the host stops matching one landmark and zeroes its match count, so that the
//              policy's "predicted often, matched rarely" rule (nPredictTimes > 2 nMatchTimes, >= 10 predictions) removes it in that frame's
//              updateFeaturesInformation -> deleteOneFeature, and sets isAdding, so that the same frame's addFeatures hands one key point to
//              integrateFeaturesInformation (joint initialisation on the device); the next frame's KalmanUpdate then runs FLAG_4_NEED_REORDER.  N stays where it
//              was.  Every landmark is "found" at its predicted pixel + 0.5 px of noise (new landmarks have no entry in the scene's z).  Prints frames/s,
//              the map changes and their wall time per operation, and how many frames ran on the step-wise fast path / on the other one
// scene.bin: int32 N, int32 F, double a1..a4, double X0[n], double S0[n*n], double z[F][2N]   (the file cslam_replay reads)
// Prints ONE JSON object.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>
#include "cslam.hpp"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s scene.bin odometry.txt mode=<capi|facade|assoc> [frames=K] [warmup=W] [hint=0|1]\n", argv[0]); return 2; }
    std::string mode = "capi";
    int K = 200, W = 20, hint = 0, churn = 0;
    std::vector<std::pair<std::string, int>> sets;
    for (int a = 3; a < argc; a++) {
        if (!strncmp(argv[a], "mode=", 5)) mode = argv[a] + 5;
        else if (!strncmp(argv[a], "frames=", 7)) K = atoi(argv[a] + 7);
        else if (!strncmp(argv[a], "warmup=", 7)) W = atoi(argv[a] + 7);
        else if (!strncmp(argv[a], "hint=", 5)) hint = atoi(argv[a] + 5);
        else if (!strncmp(argv[a], "churn=", 6)) churn = atoi(argv[a] + 6);
        else if (!strncmp(argv[a], "set=", 4)) { const char* q = strchr(argv[a] + 4, ':'); if (!q) { fprintf(stderr, "set=key:value\n"); return 2; } sets.emplace_back(std::string((const char*)argv[a] + 4, (size_t)(q - (argv[a] + 4))), atoi(q + 1)); }
    }
    for (auto& kv : sets) (void)srukf_debug_set(nullptr, kv.first.c_str(), kv.second);      // process-wide keys (e.g. set=timing:1) apply in every mode; per-filter keys: mode=capi, below
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int N = 0, F = 0; double a4[4];
    if (fread(&N, 4, 1, f) != 1 || fread(&F, 4, 1, f) != 1 || fread(a4, 8, 4, f) != 4) return 2;
    const int n = 6 * N + 4;
    std::vector<double> X0(n), S0((size_t)n * n), z((size_t)F * 2 * N);
    if (fread(X0.data(), 8, n, f) != (size_t)n || fread(S0.data(), 8, (size_t)n * n, f) != (size_t)n * n || fread(z.data(), 8, z.size(), f) != z.size()) { fprintf(stderr, "short scene file\n"); return 2; }
    fclose(f);
    if (W + K > F) { fprintf(stderr, "scene has %d frames, need %d\n", F, W + K); return 2; }
    // odometry: the reference's text format "%d : %*lf %lf %lf %lf" (SLAM.cpp:462-496)
    std::vector<double> odo;
    {
        FILE* o = fopen(argv[2], "r");
        if (!o) { perror(argv[2]); return 2; }
        char line[500];
        while (fgets(line, sizeof line, o)) { int id; double x, y, th; if (sscanf(line, "%d : %*f %lf %lf %lf", &id, &x, &y, &th) == 4) { odo.push_back(x); odo.push_back(y); odo.push_back(th); } }
        fclose(o);
    }
    if ((int)odo.size() / 3 < F + 1) { fprintf(stderr, "odometry has %d poses, need %d\n", (int)odo.size() / 3, F + 1); return 2; }

    double t_timed = 0.0, pose[4] = { 0, 0, 0, 0 }, P4[16] = { 0 };
    double tcall[5] = { 0, 0, 0, 0, 0 };                      // capi / assoc: host time inside predict_motion, predict_measurement, the association, update, get_robot
    long long matches_dev = 0;
    long long flag_ticks = -1;                                       // last frame: start of the frame's first launch -> h / Si / visible flagged to the host (10 ns ticks)
    char churn_json[900] = "";
    if (mode == "facade") {
        monoslam::CSLAM SLAM;
        SLAM.m_params.a1 = a4[0]; SLAM.m_params.a2 = a4[1]; SLAM.m_params.a3 = a4[2]; SLAM.m_params.a4 = a4[3];
        if (!SLAM.setMap(N, X0.data(), S0.data(), nullptr)) { fprintf(stderr, "%s\n", SLAM.lastError.c_str()); return 1; }
        SLAM.MIN_STEP_X = SLAM.MIN_STEP_Y = 0.0;
        if (!SLAM.loadOdometryData(argv[2])) { fprintf(stderr, "%s\n", SLAM.lastError.c_str()); return 1; }
        unsigned long long rs = 0x9E3779B97F4A7C15ull;
        auto rnd = [&rs]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; };      // xorshift, [0, 1)
        auto gauss = [&rnd]() { double s = 0; for (int q = 0; q < 12; q++) s += rnd(); return s - 6.0; };
        int victim_turn = 0;
        if (churn > 0) {
            SLAM.dataAssociation = [&](monoslam::CSLAM& s) {
                const int fr = s.m_frame.counter - 1;
                const bool change = fr >= 10 && fr % churn == 0;
                int idx = 0, victim = -1;
                if (change) {
                    // the next landmark (from a rotating start) that has been predicted often enough for the rule to remove it in THIS frame
                    const int start = (victim_turn * 7) % s.m_nMapFeatures; victim_turn++;
                    for (int q = 0; q < s.m_nMapFeatures && victim < 0; q++) { const int k = (start + q) % s.m_nMapFeatures; if (s.map[k].nPredictTimes >= 10) victim = k; }
                    s.isAdding = true;
                }
                for (monoslam::PointsMap* mp = s.map; NULL != mp; mp = mp->next, idx++) {
                    mp->isMatching = mp->isVisible;
                    mp->matchLocation.x = mp->predictLocation.x + 0.5 * gauss(); mp->matchLocation.y = mp->predictLocation.y + 0.5 * gauss();
                    if (idx == victim && mp->nPredictTimes >= 10) { mp->isMatching = false; mp->nMatchTimes = 0; }      // the policy's rule 2443: it leaves in this frame
                }
            };
            SLAM.addFeatures = [&](monoslam::CSLAM& s, std::vector<double>& kp) {
                s.isAdding = false;
                kp.assign({ 60.0 + 520.0 * rnd(), 60.0 + 360.0 * rnd() });                                  // one key point where the detector "found" a corner
                return 1;
            };
        } else
        SLAM.dataAssociation = [&](monoslam::CSLAM& s) {
            const int fr = s.m_frame.counter - 1;
            for (monoslam::PointsMap* mp = s.map; NULL != mp; mp = mp->next) {
                const double* zz = &z[(size_t)fr * 2 * N + 2 * ((mp->ID - 1) % N)];
                mp->isMatching = mp->isVisible; mp->matchLocation.x = zz[0]; mp->matchLocation.y = zz[1];
            }
        };
        for (int fr = 0; fr < W; fr++) SLAM.SLAM();
        long long fast0 = 0, slow0 = 0, fast1 = 0, slow1 = 0;
        srukf_debug_get(SLAM.context(), "step_fast", &fast0); srukf_debug_get(SLAM.context(), "step_slow", &slow0);
        const double add0 = SLAM.m_addTime, del0 = SLAM.m_deleteTime; const int na0 = SLAM.m_nAddCalls, nd0 = SLAM.m_nDeleteCalls;
        const double t0 = now_s();
        double by_phase[4] = { 0, 0, 0, 0 }; int n_phase[4] = { 0, 0, 0, 0 };      // churn: frames by distance from the last map change (0: the frame that changes the map, 1: the NEED_REORDER frame, 2, 3+)
        for (int fr = 0; fr < K; fr++) {
            const int counter = SLAM.m_frame.counter - 1;
            const double f0 = now_s();
            SLAM.SLAM();
            if (churn > 0) { const int ph = counter >= 10 ? (counter % churn < 3 ? counter % churn : 3) : 3; by_phase[ph] += now_s() - f0; n_phase[ph]++; }
        }
        t_timed = now_s() - t0;
        // (the counters live in the context, and a map change rebuilds the context behind the handle: they restart with it — so the split is read from the facade's
        //  own bookkeeping where the library cannot give it: frames since the last rebuild)
        srukf_debug_get(SLAM.context(), "step_fast", &fast1); srukf_debug_get(SLAM.context(), "step_slow", &slow1);
        if (churn > 0) {
            const int na = SLAM.m_nAddCalls - na0, nd = SLAM.m_nDeleteCalls - nd0;
            snprintf(churn_json, sizeof churn_json, "\"churn\": {\"every_frames\": %d, \"additions\": %d, \"deletions\": %d, \"ms_per_addition\": %.3f, \"ms_per_deletion\": %.3f, "
                     "\"map_change_share_of_wall\": %.3f, \"landmarks_at_end\": %d, \"step_fast_since_last_rebuild\": %lld, \"step_slow_since_last_rebuild\": %lld, "
                     "\"us_per_frame_by_distance_from_the_change\": {\"0_changes_the_map\": %.1f, \"1_need_reorder\": %.1f, \"2\": %.1f, \"3_and_later\": %.1f}}, ",
                     churn, na, nd, na ? (SLAM.m_addTime - add0) / na * 1e3 : 0.0, nd ? (SLAM.m_deleteTime - del0) / nd * 1e3 : 0.0,
                     (SLAM.m_addTime - add0 + SLAM.m_deleteTime - del0) / t_timed, SLAM.m_nMapFeatures, fast1, slow1,
                     n_phase[0] ? by_phase[0] / n_phase[0] * 1e6 : 0.0, n_phase[1] ? by_phase[1] / n_phase[1] * 1e6 : 0.0, n_phase[2] ? by_phase[2] / n_phase[2] * 1e6 : 0.0,
                     n_phase[3] ? by_phase[3] / n_phase[3] * 1e6 : 0.0);
        }
        if (!SLAM.lastError.empty()) { fprintf(stderr, "%s\n", SLAM.lastError.c_str()); return 1; }
        const int nn = SLAM.m_X_k.rows;
        for (int e = 0; e < 4; e++) pose[e] = SLAM.m_X_k.at(nn - 4 + e, 0);
        P4[0] = SLAM.m_P_k.at(nn - 4, nn - 4); P4[1] = SLAM.m_P_k.at(nn - 4, nn - 3); P4[4] = SLAM.m_P_k.at(nn - 3, nn - 4); P4[5] = SLAM.m_P_k.at(nn - 3, nn - 3);
        if (!churn && SLAM.m_nMapFeatures != N) { fprintf(stderr, "the map changed size (%d landmarks left)\n", SLAM.m_nMapFeatures); return 1; }
    } else {
        srukf_params p;
        srukf_default_params(&p);
        p.a1 = a4[0]; p.a2 = a4[1]; p.a3 = a4[2]; p.a4 = a4[3];
        srukf_ctx* c = nullptr;
        int rc = srukf_create(&c, N, &p, 0, nullptr);
        if (rc) { fprintf(stderr, "srukf_create: %d %s\n", rc, srukf_last_error(nullptr)); return 1; }
#define CK(call) do { rc = (call); if (rc) { fprintf(stderr, "%s: %d %s\n", #call, rc, srukf_last_error(c)); return 1; } } while (0)
        for (auto& kv : sets) CK(srukf_debug_set(c, kv.first.c_str(), kv.second));
        CK(srukf_set_state(c, X0.data(), S0.data()));
        std::vector<double> h(2 * N), Si(4 * N), zc(2 * N), corr(N);
        std::vector<int> vis(N), m(N), md(N);
        std::vector<unsigned char> gray;
        const bool assoc = mode == "assoc";
        if (assoc) {
            // texture: box-filtered noise (11 x 11), 640 x 480; every landmark's init patch is cut from it around its first predicted pixel, "created" at the start pose
            const int Wd = 640, Hd = 480;
            std::vector<double> t((size_t)Wd * Hd);
            unsigned long long s = 88172645463325252ull;
            for (auto& v : t) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s & 0xffff); }
            std::vector<double> b((size_t)Wd * Hd, 0.0);
            for (int y = 0; y < Hd; y++) for (int x = 0; x < Wd; x++) {
                double acc = 0; for (int dy = -5; dy <= 5; dy++) for (int dx = -5; dx <= 5; dx++) acc += t[(size_t)((y + dy + Hd) % Hd) * Wd + (x + dx + Wd) % Wd];
                b[(size_t)y * Wd + x] = acc;
            }
            double lo = b[0], hi = b[0]; for (double v : b) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
            gray.resize((size_t)Wd * Hd);
            for (size_t q = 0; q < gray.size(); q++) gray[q] = (unsigned char)((b[q] - lo) / (hi - lo) * 255.0);
            CK(srukf_predict_motion(c, &odo[0], &odo[3]));
            CK(srukf_predict_measurement(c, h.data(), Si.data(), vis.data()));
            for (int k = 0; k < N; k++) zc[2 * k] = z[2 * k], zc[2 * k + 1] = z[2 * k + 1];
            for (int k = 0; k < N; k++) m[k] = vis[k];
            const double th = X0[n - 1], R[9] = { cos(th), -sin(th), 0, sin(th), cos(th), 0, 0, 0, 1 }, tr[3] = { X0[n - 4], X0[n - 3], X0[n - 2] };
            for (int k = 0; k < N; k++) {
                double px[2] = { h[2 * k], h[2 * k + 1] };
                int cu = (int)lround(px[0]), cv = (int)lround(px[1]);
                if (!(cu >= 20 && cu < Wd - 20 && cv >= 20 && cv < Hd - 20)) { cu = 320; cv = 240; px[0] = 320; px[1] = 240; }
                unsigned char patch[441];
                for (int i = 0; i < 21; i++) for (int j = 0; j < 21; j++) patch[21 * i + j] = gray[(size_t)(cv - 10 + i) * Wd + cu - 10 + j];
                CK(srukf_set_landmark_appearance(c, k, patch, R, tr, px));
            }
            CK(srukf_update(c, zc.data(), m.data(), SRUKF_NEEDNOT_REORDER, SRUKF_UPDATE_BATCHED));
        }
        const int f0 = assoc ? 1 : 0;
        double t0 = 0.0;
        for (int fr = f0; fr < W + K; fr++) {
            if (fr == W) t0 = now_s();
            const double s0 = now_s();
            if (hint && fr + 2 <= F) CK(srukf_predict_motion_next(c, &odo[3 * fr + 3], &odo[3 * fr + 6]));      // (before the predict: its first launch then carries the third pose too)
            CK(srukf_predict_motion(c, &odo[3 * fr], &odo[3 * fr + 3]));
            const double s1 = now_s();
            CK(srukf_predict_measurement(c, h.data(), Si.data(), vis.data()));
            const double s2 = now_s();
            if (assoc) { CK(srukf_associate(c, gray.data(), zc.data(), md.data(), corr.data())); for (int k = 0; k < N; k++) matches_dev += md[k]; }
            const double* zz = &z[(size_t)fr * 2 * N];
            for (int k = 0; k < N; k++) m[k] = vis[k];                 // the host's association: every visible landmark found where the scene put it
            const double s3 = now_s();
            CK(srukf_update(c, zz, m.data(), SRUKF_NEEDNOT_REORDER, SRUKF_UPDATE_BATCHED));
            const double s4 = now_s();
            CK(srukf_get_robot(c, pose, P4));
            if (fr >= W) { tcall[0] += s1 - s0; tcall[1] += s2 - s1; tcall[2] += s3 - s2; tcall[3] += s4 - s3; tcall[4] += now_s() - s4; }
        }
        t_timed = now_s() - t0;
        srukf_debug_get(c, "meas_flag_ticks", &flag_ticks);
        srukf_destroy(c);
    }
    printf("{\"mode\": \"%s\", %s\"hint\": %d, \"landmarks\": %d, \"frames\": %d, \"warmup\": %d, \"frames_per_s\": %.2f, \"us_per_frame\": %.2f, "
           "\"pose\": [%.17g, %.17g, %.17g, %.17g], \"P_robot\": [%.17g, %.17g, %.17g, %.17g], \"device_matches\": %lld, \"stats_flag_us_into_first_launch\": %.2f, \"host_us_per_call\": {\"predict_motion\": %.2f, \"predict_measurement\": %.2f, \"association\": %.2f, \"update\": %.2f, \"get_robot\": %.2f}, "
           "\"filter_driven_by\": \"scene z / matched (host association)\"}\n",
           mode.c_str(), churn_json, hint, N, K, W, K / t_timed, t_timed / K * 1e6, pose[0], pose[1], pose[2], pose[3], P4[0], P4[1], P4[4], P4[5], matches_dev, flag_ticks * 0.01,
           tcall[0] / K * 1e6, tcall[1] / K * 1e6, tcall[2] / K * 1e6, tcall[3] / K * 1e6, tcall[4] / K * 1e6);
    fflush(stdout);                                                // (the line must not depend on what the runtimes' exit handlers do)
    return 0;
}
