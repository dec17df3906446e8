// cslam.cpp — CSLAM facade implementation (see cslam.hpp).  Host-side bookkeeping only; every
// numeric step is a C-ABI call into the gfx950 kernels.  No CPU fallback.
#include "cslam.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace monoslam {

static const double kPi = 3.14159265358979323846;

CSLAM::CSLAM(int device) : device_(device)
{
    initializeParameters();
}

CSLAM::~CSLAM()
{
    if (ctx_) srukf_destroy(ctx_);
    if (robotFile_) fclose(robotFile_);
}

bool CSLAM::check(int rc)
{
    if (rc == SRUKF_OK) return true;
    const char* m = srukf_last_error(ctx_);
    lastError = std::string("srukf error ") + std::to_string(rc) + ": " + (m ? m : "");
    return false;
}

// SLAM.cpp:158-353: numeric defaults; the image / video / dialog parts are the host's.
void CSLAM::initializeParameters()
{
    srukf_default_params(&m_params);
    m_X_k.create(4, 1);
    m_S_k.create(4, 4);
    m_S_k.at(0, 0) = m_params.sigma_x; m_S_k.at(1, 1) = m_params.sigma_y;        // 226-231
    m_S_k.at(2, 2) = m_params.sigma_z; m_S_k.at(3, 3) = m_params.sigma_theta;
    m_P_k.create(4, 4);
    m_odoXY.assign(2 * (CAPACITY + 1), 0.0);
    m_path.assign(2 * (CAPACITY + 1), 0.0);
    m_odoTheta.create(3, CAPACITY + 1);                                            // 235
    m_frame = FrameInfo();
    m_frame.stop = m_frame.start + CAPACITY;                                       // 244-246
    m_frame.index = m_frame.start;
    m_frame.counter = 1;
    m_odoCounter = 0; m_showCounter = 1; ID = 1;                                   // 248
    m_nMapFeatures = m_nPredicts = m_nMatches = m_nAddings = 0;
    m_nDeletes = m_nStores = 0; m_deleteID.clear(); isAdding = false;
    m_frameTime = m_totalTime = 0;
    mapStore.clear(); relinkMap();
}

// `map` = head of the singly linked list over mapStore (state order), NULL when the map is empty (SLAM.h:69, 154)
void CSLAM::relinkMap()
{
    for (size_t k = 0; k < mapStore.size(); k++) mapStore[k].next = (k + 1 < mapStore.size()) ? &mapStore[k + 1] : nullptr;
    map = mapStore.empty() ? nullptr : mapStore.data();
}

void CSLAM::resetAllParameters()
{
    if (ctx_) { srukf_destroy(ctx_); ctx_ = nullptr; }
    if (robotFile_) { fclose(robotFile_); robotFile_ = nullptr; }
    initializeParameters();
}

bool CSLAM::setMap(int N, const double* X, const double* S, const double* px, int n_added)
{
    if (ctx_) { srukf_destroy(ctx_); ctx_ = nullptr; }
    if (!check(srukf_create(&ctx_, N, &m_params, device_, nullptr))) return false;
    if (!check(srukf_set_state(ctx_, X, S))) return false;
    const int n = 6 * N + 4;
    m_X_k.create(n, 1); m_S_k.create(n, n); m_P_k.create(n, n);
    mapStore.assign(N, PointsMap()); relinkMap();
    ID = 1;
    for (int k = 0; k < N; k++) { map[k].ID = ID++; if (px) { map[k].initPixel.x = px[2 * k]; map[k].initPixel.y = px[2 * k + 1]; } }
    m_nMapFeatures = N;
    m_nAddings = n_added;    // 0 = steady state: FLAG_4_NEEDNOT_REORDER (SLAM.cpp:2083-2090)
    if (n_added > 0 && !check(srukf_set_new_landmarks(ctx_, n_added))) return false;                           // m_nFilters, 758-766
    refreshMirrors();
    return true;
}

namespace { struct Stopwatch { double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                              explicit Stopwatch(double& a) : acc(a) {} ~Stopwatch() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } }; }

bool CSLAM::integrateFeaturesInformation(int K, const double* kp)
{
    if (K <= 0) { m_nAddings = 0; return true; }                                                                // SLAM.cpp:820-821
    Stopwatch sw(m_addTime); m_nAddCalls++;
    if (!ctx_) {                                                                                                // frame 1: robot block only (221-231)
        if (!check(srukf_create(&ctx_, 0, &m_params, device_, nullptr))) return false;
        m_nMapFeatures = 0; mapStore.clear(); relinkMap();
    }
    if (!check(srukf_add_landmarks(ctx_, K, kp))) return false;
    const int N = m_nMapFeatures + K, n = 6 * N + 4;
    m_X_k.create(n, 1); m_S_k.create(n, n); m_P_k.create(n, n);
    for (int k = 0; k < K; k++) { PointsMap pm; pm.ID = ID++; pm.initPixel.x = kp[2 * k]; pm.initPixel.y = kp[2 * k + 1]; mapStore.push_back(pm); }
    relinkMap();
    m_nMapFeatures = N;                                                                                         // 766
    m_nAddings = K;                                                                                             // 758-765 (m_nFilters = m_nAddings = counter)
    refreshMirrors();
    return true;
}

bool CSLAM::deleteOneFeature(int id)
{
    if (!ctx_ || id < 0 || id >= m_nMapFeatures) { lastError = "deleteOneFeature: no such landmark"; return false; }
    Stopwatch sw(m_deleteTime); m_nDeleteCalls++;
    mirrorsFresh_ = false;
    if (!check(srukf_delete_landmark(ctx_, id))) return false;                                                  // 2643-2668
    mapStore.erase(mapStore.begin() + id); relinkMap();                                                                                // 2670-2705
    m_nMapFeatures--;                                                                                           // 2664
    if (m_nAddings > 0 && id >= m_nMapFeatures + 1 - m_nAddings) m_nAddings--;                                  // 2468-2492
    const int n = 6 * m_nMapFeatures + 4;
    m_X_k.create(n, 1); m_S_k.create(n, n); m_P_k.create(n, n);
    refreshMirrors();
    return true;
}

// ---- data association on the device ------------------------------------------------------------------------------
bool CSLAM::setFeatureAppearance(int id, const unsigned char* patch, const double R[9], const double t[3], const double px[2])
{
    if (!ctx_ || id < 0 || id >= m_nMapFeatures) { lastError = "setFeatureAppearance: no such landmark"; return false; }
    map[id].initPixel.x = px[0]; map[id].initPixel.y = px[1];
    return check(srukf_set_landmark_appearance(ctx_, id, patch, R, t, px));
}

bool CSLAM::dataAssociationOnDevice(const unsigned char* gray)
{
    const int N = m_nMapFeatures;
    if (!ctx_ || N == 0) return true;
    std::vector<double> z(2 * (size_t)N); std::vector<int> m(N);
    if (!check(srukf_associate(ctx_, gray, z.data(), m.data(), nullptr))) return false;
    m_nMatches = 0;
    for (int k = 0; k < N; k++) {                                                                               // 1989-2000
        map[k].isMatching = m[k] != 0;
        if (m[k]) { map[k].matchLocation.x = z[2 * k]; map[k].matchLocation.y = z[2 * k + 1]; m_nMatches++; }
    }
    return true;
}

// ---- display accessors -----------------------------------------------------------------------------------------
bool CSLAM::refreshFeaturesDisplay(bool withMirrors)
{
    const int N = m_nMapFeatures;
    if (!ctx_ || N == 0) return true;
    std::vector<double> xyz(3 * (size_t)N), cov(9 * (size_t)N);
    if (withMirrors && !fullCovariance) {
        double pose[4], P4[16];
        if (!check(srukf_get_frame_view(ctx_, m_X_k.data.data(), xyz.data(), cov.data(), pose, P4))) return false;
        const int n = m_X_k.rows;
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) m_P_k.at(n - 4 + a, n - 4 + b) = P4[4 * a + b];
        mirrorsFresh_ = true;
    } else {
        if (withMirrors && !check(srukf_get_state(ctx_, m_X_k.data.data(), nullptr))) return false;
        if (!check(srukf_get_landmarks_cartesian(ctx_, xyz.data(), cov.data()))) return false;
    }
    Mat c; c.create(3, 3);
    for (int k = 0; k < N; k++) {
        map[k].xyz.x = xyz[3 * k]; map[k].xyz.y = xyz[3 * k + 1]; map[k].xyz.z = xyz[3 * k + 2];
        for (int e = 0; e < 9; e++) { map[k].cov[e] = cov[9 * (size_t)k + e]; c.data[e] = map[k].cov[e]; }
        get3DdisplayInformation(map[k].axis, map[k].sigma, c);                                                 // 2575
    }
    return true;
}

// SLAM.cpp:2397-2621.  m_P_k = S^T S (2404) is what refreshMirrors / the device accessors stand for.
bool CSLAM::updateFeaturesInformation()
{
    if (!m_nMapFeatures || !ctx_) return true;                                                                  // 2399-2402
    // m_X_k, xyz / cov / axis / sigma from the posterior (2566-2567; also what an archived landmark takes along, 2528-2529) and the robot block the
    // mirrors show (2404): one device round trip (srukf_get_frame_view)
    if (!refreshFeaturesDisplay(true)) return false;
    const double imageWidth = m_params.image_w, imageHeight = m_params.image_h;
    m_nDeletes = 0; m_nStores = 0; m_deleteID.clear();                                                           // 2419-2422
    int id = 0;
    PointsMap* map_p = map;
    while (NULL != map_p) {                                                                                      // 2425
        const int dim = m_X_k.rows;
        const double zi = m_X_k.at(6 * id + 2, 0), theta = m_X_k.at(6 * id + 3, 0), phi = m_X_k.at(6 * id + 4, 0), rho = m_X_k.at(6 * id + 5, 0);
        const double Hlr_z = rho * (zi - m_X_k.at(dim - 2, 0)) + cos(phi) * cos(theta);                          // 2435
        const double px = map_p->predictLocation.x, py = map_p->predictLocation.y, dpx = imageWidth - px, dpy = imageHeight - py;
        const bool unmatched = map_p->nPredictTimes > 2 * map_p->nMatchTimes && map_p->nPredictTimes >= 10;
        const bool predBorder = px < DIST_2_BORDER || py < DIST_2_BORDER || dpx < DIST_2_BORDER || dpy < DIST_2_BORDER;
        bool isDelete = unmatched || rho < 0.01 || Hlr_z < 0.0 || predBorder;                                    // 2443-2446
        bool matchBorder = false;
        if (map_p->isMatching) {                                                                                 // 2448-2459
            const double mx = map_p->matchLocation.x, my = map_p->matchLocation.y;
            matchBorder = mx < DIST_2_BORDER || my < DIST_2_BORDER || imageWidth - mx < DIST_2_BORDER || imageHeight - my < DIST_2_BORDER;
            isDelete = isDelete || matchBorder;
        }
        if (isDelete) {
            bool isNeedStore = false;
            if (unmatched || rho < 0.01 || Hlr_z < 0.0) m_nPredicts--;                                           // 2465-2488
            else if (predBorder) { m_nPredicts--; if (map_p->isMatching) { isNeedStore = true; m_nStores++; m_nMatches--; } }      // 2489-2502
            else if (matchBorder) { isNeedStore = true; m_nStores++; m_nPredicts--; m_nMatches--; }              // 2503-2512
            m_deleteID.push_back(map_p->ID);                                                                     // 2514
            if (isNeedStore) {                                                                                   // 2516-2532
                FeatureInfo fi;
                fi.ID = map_p->ID; fi.isLoop = map_p->isLoop; fi.nPredictTimes = map_p->nPredictTimes; fi.nMatchTimes = map_p->nMatchTimes;
                fi.initXYZ = map_p->xyz; fi.initPixel = map_p->initPixel;
                for (int e = 0; e < 6; e++) fi.state[e] = m_X_k.at(6 * id + e, 0);
                fi.position = map_p->xyz; memcpy(fi.cov, map_p->cov, sizeof fi.cov); fi.axis = map_p->axis; fi.sigma = map_p->sigma;
                m_featuresAllInfo.push_back(fi);
            }
            if (!deleteOneFeature(id)) return false;                                                             // 2554 (m_nDeletes++, m_nMapFeatures--: 2662-2664)
            m_nDeletes++;
            map_p = (id < m_nMapFeatures) ? &mapStore[id] : nullptr;                                             // 2555-2570: the node now at position id
        } else {
            map_p->isVisible = false;                                                                            // 2598
        }
        if (id == m_nMapFeatures || !map_p) break;                                                               // 2607-2615 (a NULL node is dereferenced there)
        id++;
        map_p = map_p->next;
    }
    return true;
}

void CSLAM::getFeatureCartesianInformation(Point3d& xyz, Mat& sr, Mat& cov, const int& id) const
{
    xyz = map[id].xyz;
    cov.create(3, 3);
    for (int e = 0; e < 9; e++) cov.data[e] = map[id].cov[e];
    sr.create(0, 0);
    if (fullCovariance && m_S_k.rows >= 6 * id + 6) {
        sr.create(6, 6);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) sr.at(a, b) = m_S_k.at(6 * id + a, 6 * id + b);
    }
}

void CSLAM::get3DdisplayInformation(Quaternion& axis, Point3d& sigma, const Mat& matrix) const
{
    static thread_local Mat values, vectors;                   // (called once per landmark and frame: the buffers are kept)
    calculateEigenvaluesAndEigenvectors(matrix, values, vectors);
    matrix2Quaternion(axis, vectors);
    sigma.x = sqrt(values.at(0, 0)); sigma.y = sqrt(values.at(1, 1)); sigma.z = sqrt(values.at(2, 2));       // 1-sigma semi-axes
}

// Classical Jacobi method for a real symmetric matrix: repeatedly annihilate the off-diagonal element of largest
// magnitude with a plane rotation A <- R^T A R, accumulating V <- V R.  On return the diagonal of `eigenvalues` holds
// the eigenvalues (in the positions the rotations left them, not sorted) and column j of `eigenvectors` is the
// eigenvector of eigenvalues(j, j).  Stops when every off-diagonal magnitude is below EPSILON (false after 30 n^2 sweeps).
bool CSLAM::calculateEigenvaluesAndEigenvectors(const Mat& src, Mat& eigenvalues, Mat& eigenvectors) const
{
    const int n = src.rows;
    eigenvalues = src;
    eigenvectors.create(n, n);
    for (int i = 0; i < n; i++) eigenvectors.at(i, i) = 1.0;
    Mat& A = eigenvalues; Mat& V = eigenvectors;
    for (int it = 0; it <= 30 * n * n; it++) {
        int p = -1, q = -1; double big = 0.0;
        for (int i = 1; i < n; i++) for (int j = 0; j < i; j++) if (fabs(A.at(i, j)) > big) { big = fabs(A.at(i, j)); p = i; q = j; }
        if (big < m_params.epsilon) return true;
        // rotation angle phi with tan(2 phi) = 2 a_pq / (a_qq - a_pp), written through omega = sin(2 phi) so that the
        // half-angle formulas need no atan
        const double x = -A.at(p, q), y = 0.5 * (A.at(q, q) - A.at(p, p));
        double omega = x / sqrt(x * x + y * y);
        if (y < 0.0) omega = -omega;
        const double sn = omega / sqrt(2.0 * (1.0 + sqrt(1.0 - omega * omega)));
        const double cn = sqrt(1.0 - sn * sn);
        const double app = A.at(p, p), aqq = A.at(q, q), apq = A.at(p, q);
        A.at(p, p) = app * cn * cn + aqq * sn * sn + apq * omega;
        A.at(q, q) = app * sn * sn + aqq * cn * cn - apq * omega;
        A.at(p, q) = 0.0; A.at(q, p) = 0.0;
        for (int j = 0; j < n; j++) if (j != p && j != q) {
            const double ap = A.at(p, j), aq = A.at(q, j);
            A.at(p, j) = ap * cn + aq * sn; A.at(q, j) = -ap * sn + aq * cn;
        }
        for (int i = 0; i < n; i++) if (i != p && i != q) {
            const double ap = A.at(i, p), aq = A.at(i, q);
            A.at(i, p) = ap * cn + aq * sn; A.at(i, q) = -ap * sn + aq * cn;
        }
        for (int i = 0; i < n; i++) {
            const double vp = V.at(i, p), vq = V.at(i, q);
            V.at(i, p) = vp * cn + vq * sn; V.at(i, q) = -vp * sn + vq * cn;
        }
    }
    return false;
}

// rotation matrix -> unit quaternion (r, x, y, z), branching on the largest of trace / diagonal entries so that the
// square root argument stays away from zero; element pairing as in the reference (SLAM.cpp:2902-2948)
void CSLAM::matrix2Quaternion(Quaternion& qn, const Mat& m) const
{
    const double m11 = m.at(0, 0), m12 = m.at(0, 1), m13 = m.at(0, 2);
    const double m21 = m.at(1, 0), m22 = m.at(1, 1), m23 = m.at(1, 2);
    const double m31 = m.at(2, 0), m32 = m.at(2, 1), m33 = m.at(2, 2);
    const double tr = m11 + m22 + m33;
    if (tr > 0.0) {
        const double t = 0.5 / sqrt(tr + 1);
        qn.r = 0.25 / t; qn.x = (m23 - m32) * t; qn.y = (m31 - m13) * t; qn.z = (m12 - m21) * t;
    } else if (m11 > m22 && m11 > m33) {
        const double t = 2.0 * sqrt(1.0 + m11 - m22 - m33);
        qn.r = (m32 - m23) / t; qn.x = 0.25 * t; qn.y = (m12 + m21) / t; qn.z = (m13 + m31) / t;
    } else if (m22 > m33) {
        const double t = 2.0 * sqrt(1.0 + m22 - m11 - m33);
        qn.r = (m13 - m31) / t; qn.x = (m12 + m21) / t; qn.y = 0.25 * t; qn.z = (m23 + m32) / t;
    } else {
        const double t = 2.0 * sqrt(1.0 + m33 - m11 - m22);
        qn.r = (m21 - m12) / t; qn.x = (m13 + m31) / t; qn.y = (m23 + m32) / t; qn.z = 0.25 * t;
    }
}

// SLAM.cpp:462-496 + 363-450: "%d : %*lf %lf %lf %lf" lines, first sample is the origin, samples
// closer than MIN_STEP in both x and y are skipped, turns above MIN_STEP_THETA flag a redirection.
bool CSLAM::loadOdometryData(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) { lastError = "cannot open " + path; return false; }
    char line[500];
    m_odoCounter = 0;
    auto one = [&](int counter) -> bool {
        if (!fgets(line, sizeof line, f)) return false;
        int id = 0; double x = 0, y = 0, th = 0;
        if (sscanf(line, "%d : %*lf %lf %lf %lf", &id, &x, &y, &th) < 4) return false;
        m_odoTheta.at(0, counter) = id; m_odoTheta.at(1, counter) = th;
        if (counter == 0) {
            initOdo_[0] = x; initOdo_[1] = y;
            initPos_[0] = m_X_k.rows >= 4 ? m_X_k.at(m_X_k.rows - 4, 0) : 0; initPos_[1] = m_X_k.rows >= 4 ? m_X_k.at(m_X_k.rows - 3, 0) : 0;
            m_odoXY[0] = initPos_[0]; m_odoXY[1] = initPos_[1];
        } else {
            m_odoXY[2 * counter] = initPos_[0] + (x - initOdo_[0]);
            m_odoXY[2 * counter + 1] = initPos_[1] + (y - initOdo_[1]);
        }
        return true;
    };
    for (int skip = 1; skip < m_frame.start; skip++) if (!one(0)) break;                                       // 372-395: the origin is line m_frame.start
    if (!one(0)) { fclose(f); lastError = "empty odometry file"; return false; }
    m_odoCounter = 1;
    // 397: the robot heading starts at the first sample's heading
    if (m_X_k.rows >= 4) {
        m_X_k.at(m_X_k.rows - 1, 0) = m_odoTheta.at(1, 0);
        if (ctx_) {
            const int dim = m_X_k.rows;
            std::vector<double> X(dim), S((size_t)dim * dim);
            if (!check(srukf_get_state(ctx_, X.data(), S.data()))) { fclose(f); return false; }
            X[dim - 1] = m_odoTheta.at(1, 0);
            if (!check(srukf_set_state(ctx_, X.data(), S.data()))) { fclose(f); return false; }
            if (m_nAddings > 0 && !check(srukf_set_new_landmarks(ctx_, m_nAddings))) { fclose(f); return false; }
        }
    }
    m_odoTheta.at(2, 0) = 0;
    for (int i = 0; i < m_frame.stop - m_frame.start && m_odoCounter <= CAPACITY; i++) {                       // 400
        if (!one(m_odoCounter)) break;
        bool ok = true;
        while (std::fabs(m_odoXY[2 * m_odoCounter] - m_odoXY[2 * m_odoCounter - 2]) < MIN_STEP_X &&
               std::fabs(m_odoXY[2 * m_odoCounter + 1] - m_odoXY[2 * m_odoCounter - 1]) < MIN_STEP_Y) {          // 419-432
            if (!one(m_odoCounter)) { ok = false; break; }
        }
        if (!ok) break;
        double d = m_odoTheta.at(1, m_odoCounter) - m_odoTheta.at(1, m_odoCounter - 1);
        if (d > kPi) d -= 2 * kPi; else if (d < -kPi) d += 2 * kPi;                                              // wrapAngle 507-519
        m_odoTheta.at(2, m_odoCounter) = (std::fabs(d) > MIN_STEP_THETA * kPi / 180) ? 1 : 0;                    // 438-445
        m_odoCounter++;
    }
    fclose(f);
    return true;
}

// The redirection branch of predictMotion (SLAM.cpp:1354-1428): the odometry heading jumped by more than MIN_STEP_THETA
// (flag in m_odoTheta row 2, loadOdometryData 438-445).  The current map is archived landmark by landmark in
// m_featuresAllInfo (1357-1378), a fresh 4-state filter starts at the current x / y with the odometry heading and the
// initial robot sqrt covariance (1396-1406), the host's key points are joint-initialised (addFeatures, 1414-1416), and
// the frame counter moves on to the next odometry sample whose heading replaces the robot heading (1420-1427).
bool CSLAM::redirection()
{
    const int c = m_frame.counter, n = m_X_k.rows;
    if (m_nMapFeatures > 0 && !refreshFeaturesDisplay()) return false;                                         // xyz / cov / axis / sigma of every landmark
    if (!check(srukf_get_state(ctx_, m_X_k.data.data(), nullptr))) return false;
    int id = 0;
    for (const PointsMap* map_p = map; NULL != map_p; map_p = map_p->next, id++) {                             // 1357-1378
        FeatureInfo fi;
        fi.ID = map_p->ID; fi.isLoop = map_p->isLoop; fi.nPredictTimes = map_p->nPredictTimes; fi.nMatchTimes = map_p->nMatchTimes;
        fi.initXYZ = map_p->xyz; fi.initPixel = map_p->initPixel;
        for (int e = 0; e < 6; e++) fi.state[e] = m_X_k.at(6 * id + e, 0);
        fi.position = map_p->xyz;
        memcpy(fi.cov, map_p->cov, sizeof fi.cov);
        fi.axis = map_p->axis; fi.sigma = map_p->sigma;
        m_featuresAllInfo.push_back(fi);
    }
    const double X4[4] = { m_X_k.at(n - 4, 0), m_X_k.at(n - 3, 0), 0.0, m_odoTheta.at(1, c) };                 // 1396-1400
    const double S4[16] = { m_params.sigma_x, 0, 0, 0,  0, m_params.sigma_y, 0, 0,  0, 0, m_params.sigma_z, 0,  0, 0, 0, m_params.sigma_theta };   // 1402-1406
    srukf_destroy(ctx_); ctx_ = nullptr;
    if (!check(srukf_create(&ctx_, 0, &m_params, device_, nullptr))) return false;
    if (!check(srukf_set_state(ctx_, X4, S4))) return false;
    m_nStoreMap = m_nMapFeatures; m_nStorePredicts = m_nPredicts; m_nStoreMatches = m_nMatches;                // 1408-1410
    m_nMapFeatures = 0; m_nPredicts = 0; m_nMatches = 0; m_nAddings = 0;                                       // 1412-1416
    mapStore.clear(); relinkMap();
    m_X_k.create(4, 1); m_S_k.create(4, 4); m_P_k.create(4, 4);
    std::vector<double> keyPoints;
    const int K = addFeatures ? addFeatures(*this, keyPoints) : 0;                                             // 1418-1420 (isAdding; addFeatures 552-562)
    if (K > 0 && !integrateFeaturesInformation(K, keyPoints.data())) return false;
    m_nShowMap = m_nMapFeatures + m_nStoreMap;                                                                 // 1422
    m_frame.counter++;                                                                                         // 1424-1425
    m_frame.index = (int)m_odoTheta.at(0, m_frame.counter);
    const int dim = 6 * m_nMapFeatures + 4;                                                                    // 1427-1428: heading of the next odometry sample
    std::vector<double> X(dim), S((size_t)dim * dim);
    if (!check(srukf_get_state(ctx_, X.data(), S.data()))) return false;
    X[dim - 1] = m_odoTheta.at(1, m_frame.counter);
    if (!check(srukf_set_state(ctx_, X.data(), S.data()))) return false;
    if (m_nAddings > 0 && !check(srukf_set_new_landmarks(ctx_, m_nAddings))) return false;                     // set_state keeps K_new; explicit for clarity
    refreshMirrors();
    return true;
}

// SLAM.cpp:1343-1465: the redirection restart, then the numeric tail 1430-1465 on the device.
void CSLAM::predictMotion()
{
    if (!ctx_) { lastError = "predictMotion before setMap"; return; }
    m_frame.index = (int)m_odoTheta.at(0, m_frame.counter);                                                    // 1351
    if (m_odoTheta.at(2, m_frame.counter) == 1 && !redirection()) return;                                      // 1354-1428
    const int c = m_frame.counter;
    const double prev[3] = { m_odoXY[2 * c - 2], m_odoXY[2 * c - 1], m_odoTheta.at(1, c - 1) };               // 1444-1450
    const double cur[3]  = { m_odoXY[2 * c], m_odoXY[2 * c + 1], m_odoTheta.at(1, c) };
    // the odometry file is loaded whole before the first frame (loadOdometryData): the next frame's pair is known, and announcing it lets this frame's
    // update leave the next frame's sigma points projected (include/srukf.h: srukf_predict_motion_next).  Not across a redirection restart.
    if (c + 1 < m_odoCounter && c + 1 <= CAPACITY && m_odoTheta.at(2, c + 1) != 1) {
        const double next[3] = { m_odoXY[2 * c + 2], m_odoXY[2 * c + 3], m_odoTheta.at(1, c + 1) };
        if (!check(srukf_predict_motion_next(ctx_, cur, next))) return;
    }
    check(srukf_predict_motion(ctx_, prev, cur));
}

// SLAM.cpp:1604-1608 + the bookkeeping of QrAndCholeskyForMeasurement (1724-1745)
void CSLAM::predictMeasurement()
{
    if (!ctx_) return;
    const int N = m_nMapFeatures;
    std::vector<double> h(2 * N), Si(4 * N);
    std::vector<int> vis(N);
    if (!check(srukf_predict_measurement(ctx_, h.data(), Si.data(), vis.data()))) return;
    m_nPredicts = 0;
    for (int k = 0; k < N; k++) {
        PointsMap& p = map[k];
        p.isVisible = vis[k] != 0;
        if (p.isVisible) {                                                                                     // 1727-1738
            m_nPredicts++;
            p.isMatching = false;
            p.nPredictTimes++;
            p.predictLocation.x = h[2 * k]; p.predictLocation.y = h[2 * k + 1];
            memcpy(p.Si, &Si[4 * k], sizeof p.Si);
        }
    }
}

// SLAM.cpp:2048-2104
void CSLAM::KalmanUpdate()
{
    if (!ctx_) return;
    const int N = m_nMapFeatures;
    std::vector<double> z(2 * N, 0.0);
    std::vector<int> m(N, 0);
    m_nMatches = 0;
    for (int k = 0; k < N; k++)
        if (map[k].isMatching) { m[k] = 1; z[2 * k] = map[k].matchLocation.x; z[2 * k + 1] = map[k].matchLocation.y; m_nMatches++; map[k].nMatchTimes++; }
    if (m_nMatches == 0) return;                                                                               // 2050-2051
    const int reorder = (m_nAddings != 0) ? FLAG_4_NEED_REORDER : FLAG_4_NEEDNOT_REORDER;                      // 2083-2090
    check(srukf_update(ctx_, z.data(), m.data(), reorder, m_updateMode));
}

void CSLAM::refreshMirrors()
{
    if (!ctx_) return;
    const int n = m_X_k.rows;
    if (mirrorsFresh_ && !fullCovariance) { mirrorsFresh_ = false; return; }      // updateFeaturesInformation fetched them with the display refresh, and the map has not changed since
    if (fullCovariance) {
        check(srukf_get_state(ctx_, m_X_k.data.data(), m_S_k.data.data()));
        check(srukf_get_covariance(ctx_, m_P_k.data.data()));                                                  // 2404
    } else {
        check(srukf_get_state(ctx_, m_X_k.data.data(), nullptr));
        double pose[4], P4[16];
        check(srukf_get_robot(ctx_, pose, P4));                                                                // the 2x2 / 4x4 block the host reads (3539-3556)
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) m_P_k.at(n - 4 + a, n - 4 + b) = P4[4 * a + b];
    }
}

// SLAM.cpp:2957-3000: trajectory buffer the OpenGL view draws (m_path)
void CSLAM::updateRobotInformation()
{
    const int n = m_X_k.rows, c = m_frame.counter;
    if (c <= CAPACITY) { m_path[2 * c] = m_X_k.at(n - 4, 0); m_path[2 * c + 1] = m_X_k.at(n - 3, 0); }
}

// SLAM.cpp:3512-3562: idx \t odoX \t odoY \t x \t y \t P00 \t P01 \t P10 \t P11
void CSLAM::recordRobotInformation()
{
    if (!isRecordRobotInfo) return;
    const int n = m_X_k.rows, c = m_frame.counter;
    if (!robotFile_) {
        robotFile_ = fopen(m_recordRobotDir.c_str(), "a");
        if (!robotFile_) { lastError = "cannot open " + m_recordRobotDir; return; }
    }
    fprintf(robotFile_, "%d\t%f\t%f\t%f\t%f\t%f\t%f\t%f\t%f\t\n", m_showCounter, m_odoXY[2 * c], m_odoXY[2 * c + 1],
            m_X_k.at(n - 4, 0), m_X_k.at(n - 3, 0), m_P_k.at(n - 4, n - 4), m_P_k.at(n - 4, n - 3), m_P_k.at(n - 3, n - 4), m_P_k.at(n - 3, n - 3));
    fflush(robotFile_);
}

// SLAM.cpp:87-112
void CSLAM::SLAM()
{
    const auto t0 = std::chrono::steady_clock::now();                                                          // startTimer 122-132
    m_showCounter++;
    mirrorsFresh_ = false;
    predictMotion();
    predictMeasurement();
    if (dataAssociation) dataAssociation(*this);                                                               // loadPictures + dataAssociation (95-97)
    KalmanUpdate();
    updateFeaturesInformation();                                                                               // deletion policy + display refresh (2397-2621)
    refreshMirrors();                                                                                          // m_P_k (2404), m_X_k, m_S_k
    updateRobotInformation();
    recordRobotInformation();
    m_nAddings = 0;                                                                                            // addFeatures (552-562)
    if ((m_nMatches < m_minNUM || isAdding) && addFeatures) {                                                  // 556: the host detects, the facade joint-initialises
        std::vector<double> keyPoints;
        const int K = addFeatures(*this, keyPoints);
        if (K > 0) integrateFeaturesInformation(K, keyPoints.data());
    }
    m_frame.counter++;                                                                                         // stopTimer 142-151
    m_frameTime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    m_totalTime += m_frameTime;
}

}  // namespace monoslam
