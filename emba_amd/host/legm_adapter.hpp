// emba_amd/host/legm_adapter.hpp — the drop-in `EMBA::LEGM` for the reference tree (tub-rip/emba).
//
// A maintainer adds it to the reference as `src/emba/model_hip.cpp` in place of the LEGM methods of `src/emba/model.cpp`
// (INTEGRATION.md has the CMake lines).  It keeps the public signatures of reference include/emba/model.h:76-128 verbatim — the eight
// methods solveTimeWindow calls (src/emba/solver.cpp:63-353): the constructor, evaluateDataError, formNormalEq, formNormalEqIRLS,
// applyL2Reg, solveNormalEq, solveNormalEqCG, updateMap — so src/emba/solver.cpp and src/emba/emba.cpp compile UNCHANGED, and forwards to
// emba_host::ShardedLEGM (legm_sharded.hpp) -> emba_group_* of the C ABI -> HIP kernels, on ONE GPU or on several
// (environment EMBA_HIP_DEVICES="0,1,...,7": events time-sharded, two RCCL exchanges per iteration; default "0").
//
// What stays reference code: Model::updateTraj, evaluateRegError, evaluateRobustDataCost, recoverA22FromBlocks (model.cpp:22-53, 260-314,
// 842-861) — O(K) / O(HW) host work on objects the caller owns.
//
// It needs the reference's own dependencies (ROS messages, OpenCV, Eigen, glog) to be built there.  In THIS repository it is compiled
// and run against tests/cpp/mock_ref (the few declarations it touches, with the LEGM signatures string-compared with the reference's
// header) and the reference's vendored Eigen: tests/cpp/adapter_test.cpp drives it through solveTimeWindow's call order on the GPU.
//
// The LM loop's hidden protocol, inferred from the calls (solver.cpp never tells the model whether a step was accepted):
//   updateMap(Gx_new, ..)            builds the TRIAL map on the device and fills the caller's clones
//   evaluateDataError(.., Gx_new, ..) evaluates it without an upload (the Mats are the ones updateMap just filled)
//   formNormalEq* next                => the trial was ACCEPTED  (solver.cpp:93-131 runs only when the cost has decreased)
//   solveNormalEq* next               => the trial was REJECTED  (solver.cpp:340-352: lambda *= 10, A and b reused): the device goes back to
//                                        the equations it formed before the trial (second record set, emba_map_reject) — no re-evaluation
#pragma once
#ifndef EMBA_LEGM_ADAPTER_SKETCH
#error "This header is the reference-side binding; build it inside the reference tree (or against tests/cpp/mock_ref) with -DEMBA_LEGM_ADAPTER_SKETCH"
#endif

#include <glog/logging.h>
#include <opencv2/core.hpp>

#include <malloc.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>

#include "emba/model.h"                    // reference header: class EMBA::LEGM, typedefs VecXd/MatXd/Mat2d/EventPacket
#include "emba_amd/host/legm_sharded.hpp"  // this repository

namespace EMBA {

// One emba_host::ShardedLEGM per EMBA::LEGM object, created in the constructor from the bearing LUT that the reference's own
// EventWarper::precomputeBearingVectors builds (event_pano_warper.cpp:27-41) — expose it with a one-line getter
// `const std::vector<cv::Point3d>& EventWarper::bearingVectors() const { return precomputed_bearing_vectors_; }`.
struct LegmHipState {
    std::unique_ptr<emba_host::ShardedLEGM> impl;
    emba_host::EventPacket packet;       // dvs_msgs::Event -> {x, y, t_ns, polarity}; rebuilt only when the CONTENT of `events` changes
    const dvs_msgs::Event* src = nullptr; size_t n = 0; uint64_t src_key = 0;     // allocation, size and a sampled content hash of the caller's packet
    emba_host::NormalEquations ne;
    int K = 0;                           // control poses of the last evaluation
    size_t n_ep = 0;                     // size of the ep vector the last evaluateDataError returned
    // the trial map: the Mats updateMap filled last (data pointers + what it wrote into them)
    const unsigned char* trial_gx = nullptr; const unsigned char* trial_gy = nullptr;
    std::vector<uint32_t> trial_active;  // the pixels updateMap wrote (the trial map is zero everywhere else, model.cpp:892-901) ...
    std::vector<double> trial_gxy;       // ... and the values: gxy[2i], gxy[2i+1] at trial_active[i]
    std::vector<double> trial_sample;    // a strided sample of the planes outside them
    bool trial_pending = false;          // the last evaluation was on the trial map and its fate is not known yet
    bool map_is_trial = false;           // the device holds a trial map nobody has decided about
    // the caller's CURRENT map is zero outside these pixels (the active set of the accepted trial it came from): the next updateMap zeroes
    // just them instead of two whole planes.  Invalid until a trial of this object has been accepted, and after any map it did not produce.
    // the x2 of the last Schur solve stays on the ranks' devices: updateMap recognises it (size + a strided sample) and skips the upload
    std::vector<double> x2_sample; size_t x2_size = 0; bool x2_on_device = false;
    bool numeric_failure = false;        // the last solve met a 2x2 block that is not positive definite: x1 / x2 are NaN like the reference's (model.cpp:750)
};
static std::map<const LEGM*, LegmHipState> g_state;   // or a member `LegmHipState hip_;` added to class LEGM

namespace legm_hip_detail {

inline std::vector<int> devices_from_env()
{
    std::vector<int> dev;
    if (const char* e = std::getenv("EMBA_HIP_DEVICES")) {
        std::string s(e), tok;
        for (size_t i = 0; i <= s.size(); ++i) {
            if (i == s.size() || s[i] == ',') { if (!tok.empty()) dev.push_back(std::atoi(tok.c_str())); tok.clear(); }
            else tok.push_back(s[i]);
        }
    }
    if (dev.empty()) dev.push_back(0);
    return dev;
}

// EMBA_HIP_OPTIONS="name=value,name=value": tuning switches of the device library (emba_group_set_option) for a node the launch file configures —
// like EMBA_HIP_DEVICES, the adapter's own deployment settings: the reference's constructor signature has no room for them
inline void options_from_env(emba_host::ShardedLEGM& impl)
{
    const char* e = std::getenv("EMBA_HIP_OPTIONS");
    if (!e) return;
    std::string s(e), tok;
    for (size_t i = 0; i <= s.size(); ++i) {
        if (i == s.size() || s[i] == ',') {
            const size_t eq = tok.find('=');
            if (eq != std::string::npos) impl.setOption(tok.substr(0, eq), std::atoi(tok.c_str() + eq + 1));
            tok.clear();
        } else tok.push_back(s[i]);
    }
}

// allocation + size + a hash over 4096 events spread through the packet (all four fields) + the last one: see ShardedLEGM::packetKey
inline uint64_t packet_key(const EventPacket& ev)
{
    uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t)ev.size();
    auto mix = [&](const dvs_msgs::Event& e) {
        h ^= (uint64_t)e.ts.toNSec(); h *= 0x100000001B3ull; h ^= ((uint64_t)e.x << 32) | ((uint64_t)e.y << 8) | (e.polarity ? 1u : 0u); h *= 0x100000001B3ull;
    };
    if (ev.empty()) return h;
    const size_t step = ev.size() / 4096 ? ev.size() / 4096 : 1;
    for (size_t i = 0; i < ev.size(); i += step) mix(ev[i]);
    mix(ev.back());
    return h;
}

constexpr size_t kSample = 4099;         // values compared to recognise the trial Mats outside the pixels updateMap wrote (a prime stride count over the plane)
inline void sample_plane(const double* p, size_t n, std::vector<double>& out)
{
    const size_t step = n / kSample ? n / kSample : 1;
    for (size_t i = 0; i < n; i += step) out.push_back(p[i]);
}

// the LM decision the reference never announces: see the header comment
inline void settle_trial(LegmHipState& st, bool accepted)
{
    if (!st.map_is_trial) return;
    try { if (accepted) st.impl->acceptMap(); else st.impl->rejectMap(); }
    catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    st.map_is_trial = false; st.trial_pending = false;
    if (!accepted) { st.trial_gx = st.trial_gy = nullptr; }
}

inline void export_blocks(const emba_host::NormalEquations& ne, MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks, VecXd& b1, VecXd& b2,
                          size_t num_pix, std::set<size_t>& active, std::set<size_t>& inactive)
{
    const int dim = ne.dim_ctrl_poses; const size_t P = ne.num_active_pixels;
    A11 = Eigen::Map<const MatXd>(ne.A11.data(), dim, dim);
    b1 = Eigen::Map<const VecXd>(ne.b1.data(), dim);
    (void)A22_blocks; (void)b2;      // (already filled: form_and_download wrote the caller's containers directly)
    // The dense 3K x 2P matrix of model.cpp:358 (0.7 GB at K = 201, P = 68 k) is NOT produced: its only consumers, solveNormalEq[CG], are
    // bound to the device solvers below, which work from the sparse factors.  An empty matrix with the right row count keeps the
    // first-window trim of solver.cpp:156-165 (A12.block(3, 0, dim - 3, A12.cols())) well-formed.
    A12 = MatXd(dim, 0);
    // The caller's set becomes exactly the (ascending) active list — by an in-order MERGE with what it holds, not clear() + P insertions: between two accepted steps
    // of an LM loop the active set hardly changes, and freeing and allocating 488 k tree nodes (config 2's shape) was ~20 ms of every formNormalEq.  Correct whatever
    // the set held before (an empty set takes the P insertions).
    inactive.clear();
    {
        auto it = active.begin();
        size_t a = 0;
        const std::vector<uint32_t>& idx = ne.active_pix_idxes;
        while (it != active.end() || a < P) {
            if (it == active.end()) { active.insert(active.end(), (size_t)idx[a]); ++a; }
            else if (a == P || *it < (size_t)idx[a]) it = active.erase(it);
            else if (*it == (size_t)idx[a]) { ++it; ++a; }
            else { active.insert(it, (size_t)idx[a]); ++a; }                       // (hint: right in front of `it`)
        }
    }
#ifdef EMBA_HIP_FILL_INACTIVE
    // (the reference's updateMap walks this set, model.cpp:889-901; the device updateMap below does not — 2 M tree nodes per call saved)
    size_t a = 0; auto ih = inactive.end();
    for (size_t p = 0; p < num_pix; ++p) { if (a < P && ne.active_pix_idxes[a] == p) { ++a; continue; } ih = inactive.insert(ih, p); }
#else
    (void)num_pix;
#endif
}

// formNormalEq[IRLS] on the device, then the blocks straight into the caller's containers: a Mat2d is four doubles {m00, m10, m01, m11} and the device's block is
// {xx, xy, xy, yy} — symmetric, the same bytes — so std::vector<Mat2d> and VecXd are download targets as they stand
static_assert(sizeof(Mat2d) == 4 * sizeof(double), "Mat2d is expected to be four packed doubles");
inline void form_and_download(LegmHipState& st, int num_ctrl_poses, int thres, const std::string& cost_type, double a, std::vector<Mat2d>& A22_blocks, VecXd& b2)
{
    const size_t P = st.impl->formNormalEqOnDevice(num_ctrl_poses, thres, cost_type, a);
    A22_blocks.resize(P); b2.resize((Eigen::Index)(2 * P));
    st.impl->downloadInto(st.ne, P ? reinterpret_cast<double*>(A22_blocks.data()) : nullptr, P ? b2.data() : nullptr);
}

}  // namespace legm_hip_detail

LEGM::LEGM(const sensor_msgs::CameraInfo& camera_info_msg, double C_th, int pano_width, int pano_height)
{
    C_th_ = C_th;
    event_warper_ptr_ = new dvs::EventWarper();
    event_warper_ptr_->initialize(camera_info_msg, pano_width, pano_height);      // unchanged: builds the LUT
    const auto& bv = event_warper_ptr_->bearingVectors();
    std::vector<double> lut(3 * bv.size());
    for (size_t i = 0; i < bv.size(); ++i) { lut[3 * i] = bv[i].x; lut[3 * i + 1] = bv[i].y; lut[3 * i + 2] = bv[i].z; }
    // The LM loop allocates and frees the same few large objects every iteration — the residual vector evaluateDataError returns (60 MB at 10 M events), two map
    // clones, the 2x2 blocks — and glibc serves anything above 32 MB by mmap / munmap: fresh zeroed pages, a trap per page on first write and a teardown per free,
    // a third of the drop-in's iteration at config 2's shape (measured: 32 -> 22 ms per iteration, first iteration 176 -> 99).  With mmap off and no trimming the
    // heap keeps those pages and hands them out again.  This is a policy of the PROCESS's allocator, set by the node's model object because the node exists to run
    // this loop; EMBA_HIP_MALLOC_KEEP=0 leaves the allocator alone.
    {
        const char* keep = std::getenv("EMBA_HIP_MALLOC_KEEP");
        if (!keep || std::atoi(keep) != 0) { mallopt(M_MMAP_MAX, 0); mallopt(M_TRIM_THRESHOLD, 0x7FFFFFFF); mallopt(M_TOP_PAD, 64 << 20); }
    }
    try {
        // ~LEGM is inline in the reference header (model.h:80): the adapter cannot hook it, so an object at a reused address finds the
        // previous one's entry — every field of it is reset here, not only the engine
        g_state[this] = LegmHipState();
        g_state[this].impl.reset(new emba_host::ShardedLEGM(camera_info_msg.width, camera_info_msg.height, lut.data(), C_th, pano_width, pano_height,
                                                            legm_hip_detail::devices_from_env()));
        legm_hip_detail::options_from_env(*g_state[this].impl);
    } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
}

VecXd LEGM::evaluateDataError(Trajectory* traj_ptr, const cv::Mat& Gx, const cv::Mat& Gy, const EventPacket& events,
                              bool eval_deriv, cv::Mat& num_ev_map)
{
    auto& st = g_state[this];
    CHECK(Gx.isContinuous() && Gy.isContinuous() && Gx.type() == CV_64FC1 && num_ev_map.type() == CV_32SC1);
    if (st.numeric_failure) {
        // The solve this trial point comes from met a 2x2 block A22m_i that is not positive definite: the reference's inverse() gives inf / nan
        // there (model.cpp:750), x1 and x2 are NaN, and so are the trajectory and the map handed in here.  Its trial cost is NaN and the step is
        // rejected (solver.cpp:299, 340-352: lambda *= 10); nothing is evaluated — one NaN residual says the same.
        st.numeric_failure = false; st.trial_pending = false; st.n_ep = 1;
        num_ev_map.setTo(0);
        return VecXd::Constant(1, std::numeric_limits<double>::quiet_NaN());
    }
    // EMBA_ADAPTER_TRACE=1: where this call's time goes, on stderr (scripts/adapter_timing.py)
    static const bool trace = std::getenv("EMBA_ADAPTER_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(); };
    double tr_key = 0, tr_res = 0, tr_eval = 0;
    // the sliding window hands the same packet to every LM trial; a new window may reuse the allocation: keyed on content
    const uint64_t key = legm_hip_detail::packet_key(events);
    if (st.src != events.data() || st.n != events.size() || st.src_key != key) {
        st.packet.resize(events.size());
        emba_host::parallel_chunks(events.size(), [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) st.packet[k] = {events[k].x, events[k].y, (int64_t)events[k].ts.toNSec(), (bool)events[k].polarity};
        });
        st.src = events.data(); st.n = events.size(); st.src_key = key;
    }
    // control poses as quaternions (x,y,z,w) + spline timing exactly as LinearTrajectory stores them (trajectory.cpp:59-64)
    const int K = (int)traj_ptr->size();
    std::vector<double> knots(4 * K);
    for (int i = 0; i < K; ++i) {
        const Eigen::Quaterniond q = traj_ptr->getControlPose(i).unit_quaternion();   // a COPY: getControlPose returns an SO3d by value (trajectory.h:47,135)
        knots[4 * i] = q.x(); knots[4 * i + 1] = q.y(); knots[4 * i + 2] = q.z(); knots[4 * i + 3] = q.w();
    }
    emba_host::TrajectoryView tv{knots.data(), K, traj_ptr->startTimeNs(), traj_ptr->knotIntervalNs()};  // t_beg_ns_, dt_knots_ns_
    tr_key = since();
    // Are these the Mats updateMap has just filled?  Then the device already holds this map (the trial): no upload.
    const size_t npix = (size_t)Gx.rows * Gx.cols;
    bool resident = false;
    if (st.map_is_trial && Gx.ptr<unsigned char>() == st.trial_gx && Gy.ptr<unsigned char>() == st.trial_gy) {
        // every value updateMap wrote is compared, and a strided sample of the (zero) rest of the planes
        const double* gx = Gx.ptr<double>(); const double* gy = Gy.ptr<double>();
        resident = true;
        for (size_t i = 0; i < st.trial_active.size() && resident; ++i) {
            const uint32_t p = st.trial_active[i];
            resident = std::memcmp(&gx[p], &st.trial_gxy[2 * i], sizeof(double)) == 0 && std::memcmp(&gy[p], &st.trial_gxy[2 * i + 1], sizeof(double)) == 0;
        }
        if (resident) {
            std::vector<double> smp;
            legm_hip_detail::sample_plane(gx, npix, smp); legm_hip_detail::sample_plane(gy, npix, smp);
            resident = (smp.size() == st.trial_sample.size()) && std::memcmp(smp.data(), st.trial_sample.data(), smp.size() * sizeof(double)) == 0;
        }
    }
    if (!resident && st.map_is_trial) legm_hip_detail::settle_trial(st, false);   // some other map: the pending trial is void
    tr_res = since();
    try {
        // (round 6) the evaluation with the inlier count first, then the residuals straight into the vector this call returns: one pass over its (fresh) pages,
        // fed from pinned staging buffers — before: an 8 B x events vector zero-filled per call, a pageable copy into it and a second copy into the VecXd
        const size_t n_ep = st.impl->evaluateDataErrorCount(tv, resident ? nullptr : Gx.ptr<double>(), resident ? nullptr : Gy.ptr<double>(), st.packet,
                                                            eval_deriv, num_ev_map.ptr<int32_t>());
        tr_eval = since();
        VecXd ep((Eigen::Index)n_ep);
        if (n_ep) st.impl->fetchEp(ep.data(), n_ep);
        st.K = K; st.n_ep = n_ep; st.trial_pending = resident;
        if (trace) fprintf(stderr, "[adapter] evaluateDataError: packet key + knots %.2f ms, is-it-the-trial-map check %.2f (resident %d), evaluation + count map %.2f, ep (%zu) into a new vector %.2f\n",
                           tr_key, tr_res - tr_key, (int)resident, tr_eval - tr_res, n_ep, since() - tr_eval);
        return ep;
    } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    return VecXd();
}

void LEGM::formNormalEq(MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks, VecXd& b1, VecXd& b2, const VecXd& ep,
                        const int num_ctrl_poses, const cv::Mat& num_ev_map, const int thres_valid_pixel,
                        std::set<size_t>& active_pix_idxes, std::set<size_t>& inactive_pix_idxes)
{
    auto& st = g_state[this];
    CHECK((size_t)ep.size() == st.n_ep) << "formNormalEq expects the residual vector evaluateDataError returned (solver.cpp:99-102: the model state is that call's)";
    static const bool trace = std::getenv("EMBA_ADAPTER_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(); };
    legm_hip_detail::settle_trial(st, true); st.x2_on_device = false;        // formNormalEq after a trial evaluation: the step was accepted (solver.cpp:93-131)
    const double t_settle = since();
    try { legm_hip_detail::form_and_download(st, num_ctrl_poses, thres_valid_pixel, "quadratic", 0.0, A22_blocks, b2); }
    catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    const double t_form = since();
    legm_hip_detail::export_blocks(st.ne, A11, A12, A22_blocks, b1, b2, (size_t)num_ev_map.rows * num_ev_map.cols, active_pix_idxes, inactive_pix_idxes);
    if (trace) fprintf(stderr, "[adapter] formNormalEq: accept the trial %.2f ms, form + download %.2f, blocks and sets to the caller %.2f\n", t_settle, t_form - t_settle, since() - t_form);
}

void LEGM::formNormalEqIRLS(MatXd& A11, MatXd& A12, std::vector<Mat2d>& A22_blocks, VecXd& b1, VecXd& b2, const VecXd& ep,
                            const int num_ctrl_poses, const cv::Mat& num_ev_map, const int thres_valid_pixel,
                            std::set<size_t>& active_pix_idxes, std::set<size_t>& inactive_pix_idxes, const std::string cost_type,
                            const double a)
{
    auto& st = g_state[this];
    CHECK((size_t)ep.size() == st.n_ep) << "formNormalEqIRLS expects the residual vector evaluateDataError returned";
    legm_hip_detail::settle_trial(st, true); st.x2_on_device = false;
    try {
        st.impl->setCost(cost_type, a);             // later evaluations accumulate the weighted per-pixel sums directly (speed only)
        legm_hip_detail::form_and_download(st, num_ctrl_poses, thres_valid_pixel, cost_type, a, A22_blocks, b2);
    } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    legm_hip_detail::export_blocks(st.ne, A11, A12, A22_blocks, b1, b2, (size_t)num_ev_map.rows * num_ev_map.cols, active_pix_idxes, inactive_pix_idxes);
}

void LEGM::applyL2Reg(std::vector<Mat2d>& A22_blocks, VecXd& b2, const std::set<size_t>& active_pix_idxes, const double alpha,
                      const cv::Mat& Gx, const cv::Mat& Gy)
{
    auto& st = g_state[this];
    (void)Gx; (void)Gy; (void)active_pix_idxes;     // the device holds the same map and active set (evaluateDataError uploaded them)
    static const bool trace = std::getenv("EMBA_ADAPTER_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(); };
    const size_t P = st.ne.num_active_pixels;
    CHECK(A22_blocks.size() == P && (size_t)b2.size() == 2 * P) << "applyL2Reg expects the blocks formNormalEq exported (solver.cpp:114-130)";
    try { st.impl->applyL2RegInto(alpha, P ? reinterpret_cast<double*>(A22_blocks.data()) : nullptr, P ? b2.data() : nullptr); }
    catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    if (trace) fprintf(stderr, "[adapter] applyL2Reg: device + blocks into the caller's containers %.2f ms\n", since());
}

// The blocks the caller passes are the ones formNormalEq + applyL2Reg exported (minus the first-window trim, detected from A11.rows()):
// the device solves on its resident copy and the sparse A12 factors.
void LEGM::solveNormalEq(const MatXd& A11, const MatXd& A12, const std::vector<Mat2d>& A22_blocks, const VecXd& b1, const VecXd& b2,
                         const double lambda, VecXd& x1, VecXd& x2)
{
    auto& st = g_state[this];
    (void)A12; (void)b1;
    CHECK(A11.rows() == 3 * st.K || A11.rows() == 3 * st.K - 3) << "A11 is neither the full nor the first-window-trimmed system";
    CHECK((size_t)b2.size() == 2 * A22_blocks.size() && A22_blocks.size() == st.ne.num_active_pixels);
    const bool fix_first_pose = A11.rows() == 3 * st.K - 3;          // solver.cpp:156-165 dropped the first control pose
    legm_hip_detail::settle_trial(st, false);       // solveNormalEq again without a formNormalEq in between: the trial was rejected
    std::vector<double> v1, v2;
    st.numeric_failure = false;
    try { st.impl->solveNormalEq(lambda, fix_first_pose, v1, v2); }
    catch (const emba_host::StatusError& e) {
        // EMBA_ERR_NUMERIC: a 2x2 block A22m_i is not positive definite.  The reference does not stop there: A22m_i.inverse() yields inf / nan
        // (model.cpp:750), S and with it x1 and x2 become NaN, the trial cost is NaN, the step is rejected and lambda grows (solver.cpp:340-352).
        // Same here: NaN updates, the trial evaluation short-circuited (evaluateDataError above).  Every other status stays fatal.
        if (e.status != EMBA_ERR_NUMERIC) LOG(FATAL) << e.what();
        v1.assign(3 * (size_t)st.K, std::numeric_limits<double>::quiet_NaN());
        v2.assign(2 * st.ne.num_active_pixels, std::numeric_limits<double>::quiet_NaN());
        st.numeric_failure = true;
    }
    catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    const int skip = fix_first_pose ? 3 : 0;
    x1 = Eigen::Map<const VecXd>(v1.data() + skip, (Eigen::Index)v1.size() - skip);
    x2 = Eigen::Map<const VecXd>(v2.data(), (Eigen::Index)v2.size());
    st.x2_sample.clear(); legm_hip_detail::sample_plane(v2.data(), v2.size(), st.x2_sample);
    st.x2_size = v2.size(); st.x2_on_device = !st.numeric_failure;
}

std::pair<int, double> LEGM::solveNormalEqCG(const MatXd& A11, const MatXd& A12, const std::vector<Mat2d>& A22_blocks, const VecXd& b1,
                                             const VecXd& b2, const double lambda, VecXd& x1, VecXd& x2)
{
    auto& st = g_state[this];
    (void)A12; (void)b1; (void)b2; (void)A22_blocks;
    CHECK(A11.rows() == 3 * st.K || A11.rows() == 3 * st.K - 3);
    const bool fix_first_pose = A11.rows() == 3 * st.K - 3;
    legm_hip_detail::settle_trial(st, false);
    std::vector<double> v1, v2;
    std::pair<int, double> res(0, 0.0);
    try { res = st.impl->solveNormalEqCG(lambda, fix_first_pose, v1, v2); } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    st.x2_on_device = false;     // (updateMap uploads the x2 the caller hands back: one rank keeps its CG solution in the solver's workspace, not where updateMap looks)
    const int skip = fix_first_pose ? 3 : 0;
    x1 = Eigen::Map<const VecXd>(v1.data() + skip, (Eigen::Index)v1.size() - skip);
    x2 = Eigen::Map<const VecXd>(v2.data(), (Eigen::Index)v2.size());
    return res;
}

// Gx_new / Gy_new are clones of the current map (solver.cpp:237-238); the device builds the same trial map from ITS current map and the
// caller's clones are overwritten with it, so that everything solver.cpp later does with them (Gx = Gx_new on acceptance, saveEvoData)
// sees the reference's values.
void LEGM::updateMap(cv::Mat& Gx_new, cv::Mat& Gy_new, const VecXd& x2, const double damping_factor,
                     const std::set<size_t>& active_pix_idxes, const std::set<size_t>& inactive_pix_idxes)
{
    auto& st = g_state[this];
    (void)inactive_pix_idxes;
    CHECK(active_pix_idxes.size() == st.ne.num_active_pixels && (size_t)x2.size() == 2 * st.ne.num_active_pixels);
    CHECK(Gx_new.isContinuous() && Gy_new.isContinuous() && Gx_new.type() == CV_64FC1);
    legm_hip_detail::settle_trial(st, false);       // (a trial nobody evaluated or decided about is dropped)
    // x2 is what solveNormalEq has just returned (solver.cpp:193-239)?  Then every rank still holds it: no upload.
    bool resident = false;
    if (st.x2_on_device && (size_t)x2.size() == st.x2_size && x2.size() > 0) {
        std::vector<double> s2;
        legm_hip_detail::sample_plane(x2.data(), (size_t)x2.size(), s2);
        resident = s2.size() == st.x2_sample.size() && std::memcmp(s2.data(), st.x2_sample.data(), s2.size() * sizeof(double)) == 0;
    }
    double* gx = Gx_new.ptr<double>(); double* gy = Gy_new.ptr<double>();
    const size_t npix = (size_t)Gx_new.rows * Gx_new.cols;
    // zero the planes, all of them, as the reference does (model.cpp:892-901).  (Rounds 4-5 cleared only the previous trial's active pixels when the clones looked
    // like the map an accepted trial of this object had produced, and verified that with a strided sample: a caller's other Mats with non-zeros between the sample
    // points would have kept stale values — ADVICE r5; a full scan costs what the two memsets cost.)
    std::memset(gx, 0, npix * sizeof(double)); std::memset(gy, 0, npix * sizeof(double));
    if (st.numeric_failure) {
        // x2 is NaN (see solveNormalEq): the reference's loop writes Gx + damping * NaN into the active pixels and zero elsewhere
        // (model.cpp:863-903); the device is not touched, the trial evaluation that follows is short-circuited
        for (size_t p : active_pix_idxes) { gx[p] = std::numeric_limits<double>::quiet_NaN(); gy[p] = std::numeric_limits<double>::quiet_NaN(); }
        return;       // (these clones are dropped with the rejected step; the caller's current map is what it was)
    }
    try {
        if (resident) st.impl->updateMapResident(damping_factor);
        else { std::vector<double> v2(x2.data(), x2.data() + x2.size()); st.impl->updateMap(v2, damping_factor); }
        // the trial map is zero outside the active pixels (model.cpp:892-901): 16 B x P cross PCIe instead of 16 B x H x W
        st.impl->mapAtActive(st.trial_gxy);
    } catch (const std::exception& e) { LOG(FATAL) << e.what(); }
    st.trial_active = st.ne.active_pix_idxes;
    for (size_t i = 0; i < st.trial_active.size(); ++i) { gx[st.trial_active[i]] = st.trial_gxy[2 * i]; gy[st.trial_active[i]] = st.trial_gxy[2 * i + 1]; }
    st.map_is_trial = true; st.trial_pending = false;
    st.trial_gx = Gx_new.ptr<unsigned char>(); st.trial_gy = Gy_new.ptr<unsigned char>();
    st.trial_sample.clear();
    legm_hip_detail::sample_plane(gx, npix, st.trial_sample); legm_hip_detail::sample_plane(gy, npix, st.trial_sample);
}

}  // namespace EMBA
