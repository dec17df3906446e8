// cslam_replay — minimal C++ host that drives the CSLAM facade exactly the way the MFC view drives
// the reference class (construct, initializeParameters, loop SLAM(), read fields), with the image
// pipeline replaced by pre-computed "matched pixels" (synthetic data association).
//   cslam_replay scene.bin odometry.txt RobotPath.txt traj.bin [sequential|batched] [redirect=<counter>]
// redirect=<counter>: flags that odometry sample as a heading jump (what loadOdometryData does for |dtheta| > 45 deg,
//   SLAM.cpp:438-445), so that predictMotion takes the redirection restart (1354-1428); the host's addFeatures callback
//   "detects" the landmarks where the flagged frame's image shows them.
// extra=<file>: a map that changes mid-sequence.  File: int32 K, int32 f_starve, int32 keep, double uv_new[K][2], double z_new[F][2K].
//   In frame f_starve (0-based) only the first `keep` landmarks of the map are matched, so m_nMatches < m_minNUM and SLAM() calls
//   the addFeatures slot (SLAM.cpp:552-562), which "detects" the K key points uv_new; from then on landmark ID N + 1 + j is measured at
//   z_new[frame][2j..].  Landmarks leave through the deletion policy of updateFeaturesInformation (2443-2460).  Every map change is
//   printed as an `event` line.
// scene.bin: int32 N, int32 F, double a1..a4, double X0[n], double S0[n*n], double z[F][2N]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "cslam.hpp"

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s scene.bin odometry.txt RobotPath.txt traj.bin [sequential]\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int N = 0, F = 0; double a[4];
    if (fread(&N, 4, 1, f) != 1 || fread(&F, 4, 1, f) != 1 || fread(a, 8, 4, f) != 4) return 2;
    const int n = 6 * N + 4;
    std::vector<double> X0(n), S0((size_t)n * n), z((size_t)F * 2 * N);
    if (fread(X0.data(), 8, n, f) != (size_t)n || fread(S0.data(), 8, (size_t)n * n, f) != (size_t)n * n ||
        fread(z.data(), 8, z.size(), f) != z.size()) { fprintf(stderr, "short scene file\n"); return 2; }
    fclose(f);

    monoslam::CSLAM SLAM;                                   // MonoSLAMView.h:44
    SLAM.m_params.a1 = a[0]; SLAM.m_params.a2 = a[1]; SLAM.m_params.a3 = a[2]; SLAM.m_params.a4 = a[3];   // CSetParameters dialog (MonoSLAMView.cpp:386-443)
    if (argc > 5 && !strcmp(argv[5], "sequential")) SLAM.m_updateMode = SRUKF_UPDATE_SEQUENTIAL;
    int redirect = 0;
    for (int a = 5; a < argc; a++) if (!strncmp(argv[a], "redirect=", 9)) redirect = atoi(argv[a] + 9);
    int K_new = 0, f_starve = -1, keep = 0;
    std::vector<double> uv_new, z_new;
    for (int a = 5; a < argc; a++) if (!strncmp(argv[a], "extra=", 6)) {
        FILE* e = fopen(argv[a] + 6, "rb");
        if (!e) { perror(argv[a] + 6); return 2; }
        if (fread(&K_new, 4, 1, e) != 1 || fread(&f_starve, 4, 1, e) != 1 || fread(&keep, 4, 1, e) != 1) return 2;
        uv_new.resize(2 * (size_t)K_new); z_new.resize((size_t)F * 2 * K_new);
        if (fread(uv_new.data(), 8, uv_new.size(), e) != uv_new.size() || fread(z_new.data(), 8, z_new.size(), e) != z_new.size()) { fprintf(stderr, "short extra file\n"); return 2; }
        fclose(e);
    }
    if (!SLAM.setMap(N, X0.data(), S0.data(), nullptr)) { fprintf(stderr, "%s\n", SLAM.lastError.c_str()); return 1; }
    SLAM.MIN_STEP_X = SLAM.MIN_STEP_Y = 0.0;             // the synthetic odometry is already one pose per frame: keep every sample
    if (!SLAM.loadOdometryData(argv[2])) { fprintf(stderr, "%s\n", SLAM.lastError.c_str()); return 1; }
    SLAM.isRecordRobotInfo = true;
    SLAM.m_recordRobotDir = argv[3];
    remove(argv[3]);
    SLAM.dataAssociation = [&](monoslam::CSLAM& s) {        // stands in for loadPictures + dataAssociation
        const int fr = s.m_frame.counter - 1;
        int pos = 0;
        for (monoslam::PointsMap* mp = s.map; NULL != mp; mp = mp->next, pos++) {      // by landmark ID: the map may have changed
            monoslam::PointsMap& p = *mp;
            // IDs run on (SLAM.cpp:912): 1..N = the scene's landmarks, N+1.. = the `extra` key points; after a redirection restart
            // the scene's landmarks come back under new IDs, in the same order
            const double* zz = (K_new > 0 && p.ID > N) ? &z_new[(size_t)fr * 2 * K_new + 2 * (p.ID - N - 1)] : &z[(size_t)fr * 2 * N + 2 * ((p.ID - 1) % N)];
            p.isMatching = p.isVisible && !(fr == f_starve && pos >= keep);
            p.matchLocation.x = zz[0];
            p.matchLocation.y = zz[1];
        }
    };
    if (K_new > 0)
        SLAM.addFeatures = [&](monoslam::CSLAM& s, std::vector<double>& kp) {         // detectAndfilteringFeatures stand-in
            kp = uv_new;
            printf("event frame %d add %d\n", s.m_frame.counter - 1, K_new);
            return K_new;
        };
    if (redirect > 0) {
        SLAM.m_odoTheta.at(2, redirect) = 1;
        SLAM.addFeatures = [&](monoslam::CSLAM& s, std::vector<double>& kp) {     // detectAndfilteringFeatures stand-in
            const int fr = s.m_frame.counter - 1;           // detections on the image of the flagged frame index (cvLoadImage(image_dir, m_frame.index), 1381-1392)
            kp.assign(z.begin() + (size_t)fr * 2 * N, z.begin() + (size_t)(fr + 1) * 2 * N);
            return N;
        };
    }
    std::vector<double> traj((size_t)F * 8);
    const int steps = redirect > 0 ? F - 1 : F;             // the restart consumes one odometry sample (1424-1425)
    for (int fr = 0; fr < steps; fr++) {                    // OnBnClickedAuto loop, MonoSLAMView.cpp:526-572
        SLAM.SLAM();
        for (int q = 0; q < SLAM.m_nDeletes; q++) printf("event frame %d delete %d\n", fr, SLAM.m_deleteID[q]);
        if (!SLAM.lastError.empty()) { fprintf(stderr, "frame %d: %s\n", fr, SLAM.lastError.c_str()); return 1; }
        const int nn = SLAM.m_X_k.rows;
        for (int e = 0; e < 4; e++) traj[8 * fr + e] = SLAM.m_X_k.at(nn - 4 + e, 0);
        traj[8 * fr + 4] = SLAM.m_P_k.at(nn - 4, nn - 4); traj[8 * fr + 5] = SLAM.m_P_k.at(nn - 4, nn - 3);
        traj[8 * fr + 6] = SLAM.m_P_k.at(nn - 3, nn - 4); traj[8 * fr + 7] = SLAM.m_P_k.at(nn - 3, nn - 3);
    }
    FILE* o = fopen(argv[4], "wb");
    fwrite(traj.data(), 8, traj.size(), o);
    fclose(o);
    // what the OpenGL view reads per paint (OpenGlDisplay.cpp:449-583): xyz, cov, ellipsoid axes of every landmark
    if (!SLAM.refreshFeaturesDisplay()) { fprintf(stderr, "%s\n", SLAM.lastError.c_str()); return 1; }
    {
        std::string fn = std::string(argv[4]) + ".features";
        FILE* ff = fopen(fn.c_str(), "wb");
        // walked the way the reference's OpenGL view walks it (OpenGlDisplay.cpp:403-424): head pointer + next
        for (const monoslam::PointsMap* map_p = SLAM.map; NULL != map_p; map_p = map_p->next) {
            const monoslam::PointsMap& pm = *map_p;
            const double rec[19] = { pm.xyz.x, pm.xyz.y, pm.xyz.z, pm.cov[0], pm.cov[1], pm.cov[2], pm.cov[3], pm.cov[4], pm.cov[5], pm.cov[6], pm.cov[7], pm.cov[8],
                                     pm.axis.r, pm.axis.x, pm.axis.y, pm.axis.z, pm.sigma.x, pm.sigma.y, pm.sigma.z };
            fwrite(rec, 8, 19, ff);
        }
        fclose(ff);
    }
    if (redirect > 0) printf("redirection: archived %d  stored map %d  show map %d\n", (int)SLAM.m_featuresAllInfo.size(), SLAM.m_nStoreMap, SLAM.m_nShowMap);
    printf("frames %d  landmarks %d  predicts %d  matches %d  total %.3f s\n", F, SLAM.m_nMapFeatures, SLAM.m_nPredicts, SLAM.m_nMatches, SLAM.m_totalTime);
    return 0;
}
