// cslam.hpp — CSLAM-shaped C++ facade over the C-ABI (include/srukf.h).
//
// The reference host (MFC view, MonoSLAMView.h:44) embeds `CSLAM SLAM;` by value, calls
// SLAM.SLAM() per frame (MonoSLAMView.cpp:513,554), initializeParameters() / resetAllParameters()
// (:377,:577) and reads public fields (m_X_k, m_S_k, m_P_k, m_frame, m_nMapFeatures, m_nPredicts,
// m_nMatches, m_frameTime, m_totalTime, m_path, m_odoXY, map ...; MonoSLAMView.cpp:76-93,
// OpenGlDisplay.cpp:386-571).  This class keeps those names and meanings for the SRUKF path and
// routes the numerics to the MI355X kernels.  What is NOT here (out of scope): image
// I/O, feature detection, patch matching, drawing, MFC controls.  The step between
// predictMeasurement() and KalmanUpdate() — loadPictures() + dataAssociation() in the reference
// (SLAM.cpp:95-97) — is a host callback: it receives the predicted pixels / Si / visibility the
// reference's dataAssociation consumes and fills matchLocation / isMatching.
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "../../include/srukf.h"

namespace monoslam {

// minimal row-major fp64 stand-in for the cv::Mat members the host reads (rows/cols/ptr(i))
struct Mat {
    int rows = 0, cols = 0;
    std::vector<double> data;
    void create(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, 0.0); }
    double* ptr(int i) { return data.data() + (size_t)i * cols; }
    const double* ptr(int i) const { return data.data() + (size_t)i * cols; }
    double& at(int i, int j) { return data[(size_t)i * cols + j]; }
    double at(int i, int j) const { return data[(size_t)i * cols + j]; }
};

struct Point2d { double x = 0, y = 0; };
struct Point3d { double x = 0, y = 0, z = 0; };
struct Quaternion { double r = 1, x = 0, y = 0, z = 0; };      // SLAM.h:39-45

// SLAM.h:47-70 (numeric fields only)
struct PointsMap {
    int     ID = 0;
    bool    isVisible = false;
    bool    isMatching = false;
    bool    isLoop = false;           // delayed deletion flag the OpenGL view colours by (OpenGlDisplay.cpp:497)      (SLAM.h:52)
    int     nPredictTimes = 0;
    int     nMatchTimes = 0;
    Point2d predictLocation;
    Point2d matchLocation;
    double  Si[4] = {0, 0, 0, 0};     // 2x2 upper-triangular sqrt innovation covariance
    Point2d initPixel;
    Point3d xyz;                      // Cartesian mean                                  (SLAM.h:64)
    Quaternion axis;                  // orientation of the 1-sigma ellipsoid            (SLAM.h:66)
    Point3d sigma;                    // its semi-axes = sqrt of the eigenvalues of cov  (SLAM.h:67)
    double  cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // Cartesian 3x3 covariance
    PointsMap* next = nullptr;        // singly linked list in state order, NULL-terminated                           (SLAM.h:69)
};

// SLAM.h:94-112 (numeric fields): what the redirection restart archives per landmark of the map it leaves behind
struct FeatureInfo {
    bool    isLoop = false;
    int     ID = 0, nPredictTimes = 0, nMatchTimes = 0;
    Point3d initXYZ;
    Point2d initPixel;
    double  state[6] = {0, 0, 0, 0, 0, 0};   // the landmark's six rows of m_X_k
    Point3d position;                        // Cartesian mean
    double  cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Quaternion axis;
    Point3d sigma;
};

// SLAM.h:85-92
struct FrameInfo { int rate = 0, start = 1, stop = 0, index = 0, counter = 1; };

class CSLAM {
public:
    static const int CAPACITY = 3000;                          // SLAM.h:127
    const int FLAG_4_NEED_REORDER = 0, FLAG_4_NEEDNOT_REORDER = 1;   // SLAM.cpp:36-37

    explicit CSLAM(int device = 0);
    ~CSLAM();
    CSLAM(const CSLAM&) = delete;
    CSLAM& operator=(const CSLAM&) = delete;

    // ---- members the reference host calls (same names) ------------------------------------
    void initializeParameters();                               // SLAM.cpp:158-353 (numeric defaults only)
    void resetAllParameters();                                 // SLAM.cpp:3090-3128
    void SLAM();                                               // SLAM.cpp:87-112
    void predictMotion();                                      // SLAM.cpp:1343-1466 (numeric tail 1430-1465)
    void predictMeasurement();                                 // SLAM.cpp:1604-1608
    void KalmanUpdate();                                       // SLAM.cpp:2048-2104
    void updateRobotInformation();                             // SLAM.cpp:2957-3000 (m_path only)
    void recordRobotInformation();                             // SLAM.cpp:3512-3562 (RobotPath.txt rows)
    bool loadOdometryData(const std::string& path);            // SLAM.cpp:363-496 ("%d : %*lf %lf %lf %lf")

    // ---- map set-up (the reference builds it inside addFeatures -> integrateFeaturesInformation,
    //      SLAM.cpp:552-562, 818-1018; landmark augmentation on the device is a "next" row, so the
    //      host supplies the augmented state) ------------------------------------------------
    //      n_added = m_nFilters: the last n_added landmarks are new; the next KalmanUpdate then takes the
    //      FLAG_4_NEED_REORDER path (SLAM.cpp:2083-2090), as the reference does after integrateFeaturesInformation
    bool setMap(int n_landmarks, const double* X, const double* S, const double* init_pixels /*2N or null*/, int n_added = 0);
    // integrateFeaturesInformation (SLAM.cpp:818-871) for K key points the host detected (m_keyPoints[i].pt, distorted
    // pixels): joint initialisation on the device, the map and the mirrors grow by K, m_nFilters = m_nAddings = K so that
    // the next KalmanUpdate runs FLAG_4_NEED_REORDER.  Works from the empty map of initializeParameters (frame 1).
    bool integrateFeaturesInformation(int K, const double* keyPoints /*2K*/);
    // deleteOneFeature (SLAM.cpp:2637-2706): the id-th landmark of the state (0-based) leaves the filter and the map
    bool deleteOneFeature(int id);

    // ---- data association on the device ----------------------------------------------------------------
    // the appearance fields of PointsMap the reference fills at creation (SLAM.cpp:920-925): initPatch = the 21 x 21 gray
    // window around cvRound(initPixel) (row-major), initRotation = Rwc, initTrans = camera position
    bool setFeatureAppearance(int id, const unsigned char* initPatch, const double initRotation[9], const double initTrans[3], const double initPixel[2]);
    // wrapPatch() + dataAssociation() (SLAM.cpp:1803-2009) for the current gray frame (image_h x image_w uchar): fills
    // isMatching / matchLocation / nMatchTimes of every map entry and m_nMatches.  Install it as the association step:
    //   SLAM.dataAssociation = [&](monoslam::CSLAM& s) { s.dataAssociationOnDevice(grabGrayFrame()); };
    bool dataAssociationOnDevice(const unsigned char* gray);

    // ---- display accessors (what OpenGlDisplay.cpp:449-583 reads per paint) ------------------------------
    // updateFeaturesInformation (SLAM.cpp:2397-2621), as SLAM() calls it after KalmanUpdate: the deletion policy (2443-2460:
    // nPredictTimes > 2 nMatchTimes with >= 10 predictions, rho < 0.01, Hlr_z < 0, predicted or matched pixel within
    // DIST_2_BORDER of the image border) -> deleteOneFeature, landmarks that leave while matched are archived in
    // m_featuresAllInfo (2516-2532); the landmarks that stay get xyz / axis / sigma refreshed (2566-2580) and isVisible
    // cleared (2598).  The loop keeps the reference's traversal: the node that moves into a deleted node's place is not
    // examined in the same call (2554-2570, 2607-2615).
    bool updateFeaturesInformation();
    // the display part alone (2566-2580): xyz, cov, axis, sigma of every landmark of `map`, from ONE device call
    // (srukf_get_landmarks_cartesian) instead of a pass over the n x n m_P_k per landmark
    bool refreshFeaturesDisplay(bool withMirrors = false);   // withMirrors: m_X_k and the robot block of m_P_k in the same device round trip
    // getFeatureCartesianInformation (2721-2751): xyz and cov (3x3) of landmark id from the last refresh; sr (the 6x6
    // diagonal block of m_S_k) only when fullCovariance mirrors are on, otherwise left empty
    void getFeatureCartesianInformation(Point3d& xyz, Mat& sr, Mat& cov, const int& id) const;
    // get3DdisplayInformation (2791-2802), calculateEigenvaluesAndEigenvectors (2816-2891: classical Jacobi rotations,
    // largest off-diagonal pivot, eigenvalues left on the diagonal in place), matrix2Quaternion (2902-2948)
    void get3DdisplayInformation(Quaternion& axis, Point3d& sigma, const Mat& matrix) const;
    bool calculateEigenvaluesAndEigenvectors(const Mat& src, Mat& eigenvalues, Mat& eigenvectors) const;
    void matrix2Quaternion(Quaternion& quaternion, const Mat& matrix) const;

    // the reference's loadPictures()+dataAssociation() slot (SLAM.cpp:95-97)
    std::function<void(CSLAM&)> dataAssociation;
    // the reference's addFeatures() -> detectAndfilteringFeatures / insureEnoughFeatures slot (SLAM.cpp:552-562, 574-808):
    // the host detects key points on its current image and returns them as (u, v) pairs (distorted pixels); the facade
    // joint-initialises them (integrateFeaturesInformation).  Used by the redirection restart of predictMotion.
    std::function<int(CSLAM&, std::vector<double>& keyPoints)> addFeatures;
    // ---- redirection (SLAM.cpp:1354-1428): when the odometry heading jumps by more than MIN_STEP_THETA the reference
    //      archives the current map in m_featuresAllInfo and restarts a fresh 4-state filter at the current position ----
    std::vector<FeatureInfo> m_featuresAllInfo;          // SLAM.h:170
    int m_nStoreMap = 0, m_nStorePredicts = 0, m_nStoreMatches = 0, m_nShowMap = 0;   // 1408-1410, 1418

    // ---- public state, reference names (SLAM.h:154-290) -----------------------------------------
    // The reference's map is a singly linked list the host walks (`PointsMap* map_p = SLAM->map; while (NULL != map_p)
    // { ... map_p = map_p->next; }`, OpenGlDisplay.cpp:403-424, 460-509; SLAM.h:154).  Same here: `map` is the head (NULL
    // for an empty map), `next` links the nodes in state order.  The nodes live contiguously in mapStore, so map[k] is
    // also the k-th landmark; the links are rebuilt whenever landmarks are added or deleted.
    PointsMap* map = nullptr;
    std::vector<PointsMap> mapStore;
    FrameInfo m_frame;
    srukf_params m_params;               // the tunables of SLAM.cpp:172-198, 221-224, 329-337
    int    m_updateMode = SRUKF_UPDATE_BATCHED;
    int    m_nMapFeatures = 0, m_nPredicts = 0, m_nMatches = 0, m_nAddings = 0, m_odoCounter = 0, m_showCounter = 1;
    int    m_nDeletes = 0, m_nStores = 0;    // landmarks deleted / archived by the last updateFeaturesInformation (2419-2420)
    std::vector<int> m_deleteID;             // their IDs (m_deleteID, SLAM.h:278)
    int    ID = 1;                           // running landmark ID (SLAM.h:202, SLAM.cpp:248, 912)
    int    m_minNUM = 5;                     // addFeatures when fewer landmarks matched (SLAM.cpp:179, 556)
    bool   isAdding = false;                 // forces addFeatures (SLAM.cpp:268, 556)
    int    DIST_2_BORDER = 20;               // SLAM.cpp:48
    double m_frameTime = 0, m_totalTime = 0;
    std::vector<double> m_odoXY, m_path; // 2*(CAPACITY+1) each (SLAM.h:188-189)
    Mat    m_odoTheta;                   // 3 x (CAPACITY+1): index, theta, redirection flag (SLAM.cpp:235)
    Mat    m_X_k, m_S_k, m_P_k;          // host mirrors, refreshed after every frame (m_P_k: robot block only unless fullCovariance)
    bool   fullCovariance = false;       // true: m_P_k = S^T S in full (SLAM.cpp:2404), false: only the blocks the host reads
    bool   isRecordRobotInfo = false;
    std::string m_recordRobotDir = "RobotPath.txt";
    double MIN_STEP_X = 0.01, MIN_STEP_Y = 0.01, MIN_STEP_THETA = 45;   // SLAM.cpp:45-47
    std::string lastError;
    // wall time spent inside the map changes (integrateFeaturesInformation / deleteOneFeature: the device call and the host bookkeeping around it) and how many
    // there were: what a host that watches its frame rate under map churn wants to see (cslam_step_bench churn=P)
    double m_addTime = 0, m_deleteTime = 0; int m_nAddCalls = 0, m_nDeleteCalls = 0;
    srukf_ctx* context() const { return ctx_; }          // for srukf_debug_get / srukf_last_error next to the facade (diagnostics: the facade owns the handle)

private:
    bool redirection();
    void relinkMap();
    void refreshMirrors();
    bool check(int rc);
    srukf_ctx* ctx_ = nullptr;
    bool mirrorsFresh_ = false;          // m_X_k / the robot block of m_P_k were fetched by this frame's display refresh
    int device_ = 0;
    FILE* robotFile_ = nullptr;
    double initOdo_[2] = {0, 0}, initPos_[2] = {0, 0};
};

}  // namespace monoslam
