// emba_amd/host/legm_adapter.hpp — the drop-in `EMBA::LEGM` for the reference tree (tub-rip/emba).
//
// NOT compiled in this repository: it needs the reference's own dependencies (ROS messages, OpenCV, Eigen, glog,
// the reference's utils/trajectory.h and emba/model.h typedefs).  A maintainer adds it to the reference as
// `src/emba/model_hip.cpp` in place of the LEGM part of `src/emba/model.cpp` (INTEGRATION.md has the CMake lines).
// It keeps the public signatures of reference include/emba/model.h:76-108 verbatim, so src/emba/solver.cpp and
// src/emba/emba.cpp compile unchanged, and forwards to emba_host::LEGM (legm_host.hpp) -> C ABI -> HIP kernels.
//
// What stays on the CPU in the reference: Model::updateTraj, LEGM::solveNormalEq[CG], updateMap, recoverA22FromBlocks
// (model.cpp:22-53, 721-903) — unchanged reference code ("next" rows of SURVEY §8f).
#pragma once
#ifndef EMBA_LEGM_ADAPTER_SKETCH
#error "This header documents the reference-side binding; build it inside the reference tree with -DEMBA_LEGM_ADAPTER_SKETCH"
#endif

#include <glog/logging.h>
#include <opencv2/core.hpp>

#include "emba/model.h"                 // reference header: class EMBA::LEGM, typedefs VecXd/MatXd/Mat2d/EventPacket
#include "emba_amd/host/legm_host.hpp"  // this repository

namespace EMBA {

// One emba_host::LEGM per EMBA::LEGM object, created in the constructor from the bearing LUT that the reference's own
// EventWarper::precomputeBearingVectors builds (event_pano_warper.cpp:27-41) — expose it with a one-line getter
// `const std::vector<cv::Point3d>& EventWarper::bearingVectors() const { return precomputed_bearing_vectors_; }`.
struct LegmHipState {
    std::unique_ptr<emba_host::LEGM> impl;
    emba_host::EventPacket packet;       // dvs_msgs::Event -> {x, y, t_ns, polarity}; rebuilt only when `events` changes
    const dvs_msgs::Event* src = nullptr; size_t n = 0;
    emba_host::NormalEquations ne;
};
static std::map<const LEGM*, LegmHipState> g_state;   // or a member `LegmHipState hip_;` added to class LEGM

LEGM::LEGM(const sensor_msgs::CameraInfo& camera_info_msg, double C_th, int pano_width, int pano_height)
{
    C_th_ = C_th;
    event_warper_ptr_ = new dvs::EventWarper();
    event_warper_ptr_->initialize(camera_info_msg, pano_width, pano_height);      // unchanged: builds the LUT
    const auto& bv = event_warper_ptr_->bearingVectors();
    std::vector<double> lut(3 * bv.size());
    for (size_t i = 0; i < bv.size(); ++i) { lut[3 * i] = bv[i].x; lut[3 * i + 1] = bv[i].y; lut[3 * i + 2] = bv[i].z; }
    try {
        g_state[this].impl.reset(new emba_host::LEGM(camera_info_msg.width, camera_info_msg.height, lut.data(), C_th,
                                                     pano_width, pano_height));
    } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
}

VecXd LEGM::evaluateDataError(Trajectory* traj_ptr, const cv::Mat& Gx, const cv::Mat& Gy, const EventPacket& events,
                              bool eval_deriv, cv::Mat& num_ev_map)
{
    auto& st = g_state[this];
    CHECK(Gx.isContinuous() && Gy.isContinuous() && Gx.type() == CV_64FC1 && num_ev_map.type() == CV_32SC1);
    if (st.src != events.data() || st.n != events.size()) {        // the sliding window hands the same packet to every LM trial
        st.packet.resize(events.size());
        for (size_t k = 0; k < events.size(); ++k)
            st.packet[k] = {events[k].x, events[k].y, (int64_t)events[k].ts.toNSec(), (bool)events[k].polarity};
        st.src = events.data(); st.n = events.size();
    }
    // control poses as quaternions (x,y,z,w) + spline timing exactly as LinearTrajectory stores them (trajectory.cpp:59-64)
    const int K = (int)traj_ptr->size();
    std::vector<double> knots(4 * K);
    for (int i = 0; i < K; ++i) {
        const Eigen::Quaterniond& q = traj_ptr->getControlPose(i).unit_quaternion();
        knots[4 * i] = q.x(); knots[4 * i + 1] = q.y(); knots[4 * i + 2] = q.z(); knots[4 * i + 3] = q.w();
    }
    emba_host::TrajectoryView tv{knots.data(), K, traj_ptr->startTimeNs(), traj_ptr->knotIntervalNs()};  // t_beg_ns_, dt_knots_ns_
    try {
        std::vector<double> ep = st.impl->evaluateDataError(tv, Gx.ptr<double>(), Gy.ptr<double>(), st.packet, eval_deriv,
                                                            num_ev_map.ptr<int32_t>());
        return Eigen::Map<VecXd>(ep.data(), (Eigen::Index)ep.size());
    } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    return VecXd();
}

static void export_blocks(const emba_host::NormalEquations& ne, MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks, VecXd& b1,
                          VecXd& b2, size_t num_pix, std::set<size_t>& active, std::set<size_t>& inactive, bool with_A12)
{
    const int dim = ne.dim_ctrl_poses; const size_t P = ne.num_active_pixels;
    A11 = Eigen::Map<const MatXd>(ne.A11.data(), dim, dim);
    b1 = Eigen::Map<const VecXd>(ne.b1.data(), dim);
    b2 = Eigen::Map<const VecXd>(ne.b2.data(), 2 * P);
    A22_blocks.resize(P);
    for (size_t i = 0; i < P; ++i) A22_blocks[i] << ne.A22_blocks[4 * i], ne.A22_blocks[4 * i + 1], ne.A22_blocks[4 * i + 2], ne.A22_blocks[4 * i + 3];
    if (with_A12) A12 = Eigen::Map<const MatXd>(ne.A12.data(), dim, 2 * P);
    active.clear(); inactive.clear();
    auto hint = active.end();
    for (uint32_t p : ne.active_pix_idxes) hint = active.insert(hint, p);          // already ascending: O(P)
    size_t a = 0; auto ih = inactive.end();
    for (size_t p = 0; p < num_pix; ++p) { if (a < P && ne.active_pix_idxes[a] == p) { ++a; continue; } ih = inactive.insert(ih, p); }
}

void LEGM::formNormalEq(MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks, VecXd& b1, VecXd& b2, const VecXd& ep,
                        const int num_ctrl_poses, const cv::Mat& num_ev_map, const int thres_valid_pixel,
                        std::set<size_t>& active_pix_idxes, std::set<size_t>& inactive_pix_idxes)
{
    auto& st = g_state[this];
    std::vector<double> epv(ep.data(), ep.data() + ep.size());
    try {
        // dense A12 (3K x 2P doubles, model.cpp:358) is what the reference's solveNormalEq consumes; ask for it only while that
        // solver is still the CPU one — the sparse factors (emba_get_A12_sparse) are the scalable form.
        st.impl->formNormalEq(st.ne, epv, num_ctrl_poses, thres_valid_pixel, /*want_dense_A12=*/true);
    } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    export_blocks(st.ne, A11, A12, A22_blocks, b1, b2, (size_t)num_ev_map.rows * num_ev_map.cols, active_pix_idxes, inactive_pix_idxes, true);
}

void LEGM::formNormalEqIRLS(MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks, VecXd& b1, VecXd& b2, const VecXd& ep,
                            const int num_ctrl_poses, const cv::Mat& num_ev_map, const int thres_valid_pixel,
                            std::set<size_t>& active_pix_idxes, std::set<size_t>& inactive_pix_idxes, const std::string cost_type,
                            const double a)
{
    auto& st = g_state[this];
    std::vector<double> epv(ep.data(), ep.data() + ep.size());
    try { st.impl->formNormalEqIRLS(st.ne, epv, num_ctrl_poses, thres_valid_pixel, cost_type, a, true); }
    catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    export_blocks(st.ne, A11, A12, A22_blocks, b1, b2, (size_t)num_ev_map.rows * num_ev_map.cols, active_pix_idxes, inactive_pix_idxes, true);
}

void LEGM::applyL2Reg(std::vector<Mat2d>& A22_blocks, VecXd& b2, const std::set<size_t>& active_pix_idxes, const double alpha,
                      const cv::Mat& Gx, const cv::Mat& Gy)
{
    auto& st = g_state[this];
    (void)Gx; (void)Gy; (void)active_pix_idxes;     // the device holds the same map and active set (evaluateDataError uploaded them)
    try { st.impl->applyL2Reg(st.ne, alpha); } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    const size_t P = st.ne.num_active_pixels;
    for (size_t i = 0; i < P; ++i) A22_blocks[i] << st.ne.A22_blocks[4 * i], st.ne.A22_blocks[4 * i + 1], st.ne.A22_blocks[4 * i + 2], st.ne.A22_blocks[4 * i + 3];
    b2 = Eigen::Map<const VecXd>(st.ne.b2.data(), 2 * P);
}

}  // namespace EMBA
