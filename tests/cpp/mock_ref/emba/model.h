// mock of reference include/emba/model.h (and of what it pulls in): ONLY the declarations the adapter defines or touches.  The LEGM
// member signatures below are those of reference include/emba/model.h:76-108 (tests/test_adapter_syntax.py also compares them with the
// reference's header text when /root/reference is present); everything else is the smallest type that makes them well-formed.
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include <opencv2/core.hpp>

namespace ros { struct Time { uint32_t sec = 0, nsec = 0; uint64_t toNSec() const { return (uint64_t)sec * 1000000000ull + nsec; } }; }
namespace dvs_msgs { struct Event { uint16_t x, y; ros::Time ts; uint8_t polarity; }; }          // dvs_msgs/Event.msg
namespace sensor_msgs { struct CameraInfo { uint32_t height = 0, width = 0; }; }

namespace dvs {
class EventWarper {                                            // reference include/utils/event_pano_warper.h
public:
    // (the reference builds the bearing vectors from camera_info through image_geometry, event_pano_warper.cpp:27-41 — ROS, absent here:
    // the mock takes them from a table the test program sets)
    static std::vector<cv::Point3d>& mockBearingTable() { static std::vector<cv::Point3d> t; return t; }
    void initialize(const sensor_msgs::CameraInfo&, int, int) { precomputed_bearing_vectors_ = mockBearingTable(); }
    const std::vector<cv::Point3d>& bearingVectors() const { return precomputed_bearing_vectors_; }   // the one-line getter INTEGRATION.md adds
private:
    std::vector<cv::Point3d> precomputed_bearing_vectors_;
};
}  // namespace dvs

struct MockSO3d { Eigen::Quaterniond q; const Eigen::Quaterniond& unit_quaternion() const { return q; } ~MockSO3d() { q.coeffs().setConstant(std::nan("")); } };   // Sophus::SO3d (the destructor poisons the value: a reference kept into a temporary shows up as NaN)
class Trajectory {                                             // reference include/utils/trajectory.h:23-100 (the abstract base)
public:
    virtual ~Trajectory() {}
    virtual size_t size() = 0;                                 // :35
    virtual MockSO3d getControlPose(const int idx) = 0;        // :47 (Sophus::SO3d)
    int64_t startTimeNs() const { return t_beg_ns_; }          // the two getters INTEGRATION.md adds for the members at :92
    int64_t knotIntervalNs() const { return dt_knots_ns_; }
    void mockSetTiming(int64_t t0, int64_t dt) { t_beg_ns_ = t0; dt_knots_ns_ = dt; }
protected:
    int64_t t_beg_ns_ = 0, dt_knots_ns_ = 0;                   // :92
};

namespace EMBA {

typedef std::vector<dvs_msgs::Event> EventPacket;
typedef Eigen::VectorXd VecXd;
typedef Eigen::MatrixXd MatXd;
typedef Eigen::Matrix2d Mat2d;

class Model {
protected:
    double C_th_;
    dvs::EventWarper* event_warper_ptr_;
};

class LEGM : public Model {
public:
    LEGM(const sensor_msgs::CameraInfo& camera_info_msg, double C_th,
         int pano_width, int pano_height);
    ~LEGM() { delete event_warper_ptr_; }
    VecXd evaluateDataError(Trajectory* traj_ptr, const cv::Mat& Gx, const cv::Mat& Gy,
                            const EventPacket& events, bool eval_deriv, cv::Mat& num_ev_map);
    void formNormalEq(MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks,
                      VecXd& b1, VecXd& b2, const VecXd& ep, const int num_ctrl_poses,
                      const cv::Mat& num_ev_map, const int thres_valid_pixel,
                      std::set<size_t>& active_pix_idxes, std::set<size_t>& inactive_pix_idxes);
    void formNormalEqIRLS(MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks,
                          VecXd& b1, VecXd& b2, const VecXd& ep, const int num_ctrl_poses,
                          const cv::Mat& num_ev_map, const int thres_valid_pixel,
                          std::set<size_t>& active_pix_idxes, std::set<size_t>& inactive_pix_idxes,
                          const std::string cost_type, const double a);
    void applyL2Reg(std::vector<Mat2d>& A22_blocks, VecXd& b2,
                    const std::set<size_t>& active_pix_idxes, const double alpha,
                    const cv::Mat& Gx, const cv::Mat& Gy);
    void solveNormalEq(const MatXd& A11, const MatXd& A12, const std::vector<Mat2d>& A22_blocks,
                       const VecXd& b1, const VecXd& b2, const double lambda,
                       VecXd& x1, VecXd& x2);
    std::pair<int, double> solveNormalEqCG(const MatXd& A11, const MatXd& A12,
                                           const std::vector<Mat2d>& A22_blocks,
                                           const VecXd& b1, const VecXd& b2,
                                           const double lambda, VecXd& x1, VecXd& x2);
    void updateMap(cv::Mat& Gx_new, cv::Mat& Gy_new,
                   const VecXd& x2, const double damping_factor,
                   const std::set<size_t>& active_pix_idxes,
                   const std::set<size_t>& inactive_pix_idxes);
};

}  // namespace EMBA
