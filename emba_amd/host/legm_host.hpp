// emba_amd/host/legm_host.hpp — C++17 host side above the C ABI (include/emba_hip.h), free of ROS/OpenCV/Eigen types.
//
// Mirrors the reference's measurement-model object `EMBA::LEGM` (reference include/emba/model.h:72-133): same method
// names, argument meaning and fail-fast error behaviour, on plain pointers / std::vector.  The adapter a maintainer drops
// into the reference tree (emba_amd/host/legm_adapter.hpp, shown in INTEGRATION.md) is a thin type conversion over this
// class: cv::Mat -> double*, Eigen -> double*, std::vector<dvs_msgs::Event> -> struct of arrays.
//
// All computation happens on the GPU behind the C ABI; this header only converts containers and caches the event packet
// (the per-pixel event lists are pose-independent, so they are uploaded once per packet, not once per call).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/emba_hip.h"

namespace emba_host {

// what check() throws: the status of the C ABI call rides along, so that a caller can tell EMBA_ERR_NUMERIC (a 2x2 block that is not positive
// definite: the reference carries on with inf / nan and loses the step) from a real failure
struct StatusError : std::runtime_error {
    emba_status status;
    StatusError(emba_status st, const std::string& msg) : std::runtime_error("emba_hip status " + std::to_string((int)st) + ": " + msg), status(st) {}
};

// fn(lo, hi) over [0, n) in up to four pieces on as many threads (the calling one included): the once-per-window container conversions of a 10 M-event packet
// (array of structs -> the C ABI's struct of arrays) are memory passes of 150-300 MB, 25-50 ms on one thread
template <class F> inline void parallel_chunks(size_t n, F&& fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = (n < ((size_t)1 << 20)) ? 1 : (hw >= 4 ? 4 : (hw >= 2 ? 2 : 1));
    if (T == 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + T - 1) / T;
    for (size_t t = 1; t < T; ++t) th.emplace_back([&fn, per, n, t] { const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1)); if (hi > lo) fn(lo, hi); });
    fn((size_t)0, std::min(n, per));
    for (auto& t : th) t.join();
}

struct Event {              // dvs_msgs::Event: uint16 x, uint16 y, time ts, bool polarity
    uint16_t x, y;
    int64_t t_ns;           // ts.toNSec()
    bool polarity;
};
using EventPacket = std::vector<Event>;   // reference include/emba/model.h:17

struct TrajectoryView {     // what LinearTrajectory hands to the hot path (reference src/utils/trajectory.cpp:59-73,122-147)
    const double* knots_xyzw;   // size() unit quaternions (x,y,z,w): getControlPose(i).unit_quaternion().coeffs()
    int num_ctrl_poses;         // traj->size()
    int64_t t0_ns, dt_ns;       // int64_t(1e9*t_beg), int64_t(1e9*dt_knots)
};

struct NormalEquations {    // the in/out arguments of formNormalEq (model.h:93-96), column-major like Eigen::MatrixXd
    std::vector<double> A11, b1, A22_blocks /* P x [xx xy; xy yy] */, b2, A12 /* 3K x 2P, only if requested */;
    std::vector<uint32_t> active_pix_idxes;   // ascending panorama index (std::set order); inactive = complement
    size_t num_active_pixels = 0;
    int dim_ctrl_poses = 0;
};

class LEGM {
public:
    // LEGM(camera_info, C_th, pano_width, pano_height), model.cpp:56-70.  bearing_lut replaces camera_info
    // (sensor_w*sensor_h*3 doubles, the precomputed_bearing_vectors_ of event_pano_warper.cpp:27-41).
    LEGM(int sensor_w, int sensor_h, const double* bearing_lut, double C_th, int pano_width, int pano_height, int device = 0)
        : W_(pano_width), H_(pano_height)
    {
        emba_cfg cfg{};
        cfg.sensor_w = sensor_w; cfg.sensor_h = sensor_h; cfg.pano_w = pano_width; cfg.pano_h = pano_height;
        cfg.bearing_lut = bearing_lut; cfg.C_th = C_th; cfg.event_batch = 100; cfg.outlier_px = 10.0; cfg.device = device;
        const emba_status st = emba_create(&cfg, &ctx_);
        if (st != EMBA_OK) fatal(st, emba_last_error(nullptr));
    }
    ~LEGM() { emba_destroy(ctx_); }
    LEGM(const LEGM&) = delete;
    LEGM& operator=(const LEGM&) = delete;

    // VecXd evaluateDataError(traj, Gx, Gy, events, eval_deriv, num_ev_map)    model.cpp:72-258
    // Gx, Gy: pano_height*pano_width row-major doubles; num_ev_map: int32, same shape, overwritten.
    std::vector<double> evaluateDataError(const TrajectoryView& traj, const double* Gx, const double* Gy, const EventPacket& events,
                                          bool eval_deriv, int32_t* num_ev_map)
    {
        ensure_events(events);
        K_ = traj.num_ctrl_poses;
        std::vector<double> ep(events.size() ? events.size() : 1);
        size_t n_inl = 0;
        check(emba_eval_data_error(ctx_, traj.knots_xyzw, traj.num_ctrl_poses, traj.t0_ns, traj.dt_ns, Gx, Gy, eval_deriv ? 1 : 0,
                                   ep.data(), &n_inl, num_ev_map));
        ep.resize(n_inl);   // ep0.head(inlier_count), model.cpp:256
        return ep;
    }

    // 0.5*ep.dot(ep) (solver.cpp:88) / evaluateRobustDataCost(ep, cost_type, a) (model.cpp:279-314), on the resident residuals
    double evaluateRobustDataCost(const std::string& cost_type, double a)
    {
        double v = 0;
        check(emba_data_cost(ctx_, irls_code(cost_type), a, &v));
        return v;
    }
    // alpha*0.5*|evaluateRegError|^2 (model.cpp:260-277, solver.cpp:90) on the resident map
    double regCost(double alpha)
    {
        double v = 0;
        check(emba_reg_cost(ctx_, alpha, &v));
        return v;
    }

    // formNormalEq(A11, A12, A22_blocks, b1, b2, ep, num_ctrl_poses, num_ev_map, thres, active, inactive)   model.cpp:316-491
    void formNormalEq(NormalEquations& ne, const std::vector<double>& ep, int num_ctrl_poses, int thres_valid_pixel,
                      bool want_dense_A12 = false)
    {
        form(ne, ep, num_ctrl_poses, thres_valid_pixel, 0, 0.0, want_dense_A12);
    }
    // formNormalEqIRLS(..., cost_type, a)   model.cpp:493-687
    void formNormalEqIRLS(NormalEquations& ne, const std::vector<double>& ep, int num_ctrl_poses, int thres_valid_pixel,
                          const std::string& cost_type, double a, bool want_dense_A12 = false)
    {
        form(ne, ep, num_ctrl_poses, thres_valid_pixel, irls_code(cost_type), a, want_dense_A12);
    }
    // applyL2Reg(A22_blocks, b2, active, alpha, Gx, Gy)   model.cpp:689-719 (on the device-resident blocks; call once)
    void applyL2Reg(NormalEquations& ne, double alpha) { finish(ne, alpha, false); }

    emba_ctx* ctx() { return ctx_; }

private:
    static int irls_code(const std::string& t) { return t == "cauchy" ? 2 : t == "huber" ? 1 : 0; }

    [[noreturn]] static void fatal(emba_status st, const char* msg)
    {   // the reference aborts through glog CHECK / LOG(FATAL); callers that link glog can catch and LOG(FATAL)
        throw StatusError(st, msg ? msg : "");
    }
    void check(emba_status st) { if (st != EMBA_OK) fatal(st, emba_last_error(ctx_)); }

    void ensure_events(const EventPacket& ev)
    {
        const Event* d = ev.data();
        const int64_t tf = ev.empty() ? 0 : ev.front().t_ns, tl = ev.empty() ? 0 : ev.back().t_ns;
        if (have_ev_ && d == ev_ptr_ && ev.size() == ev_n_ && tf == ev_t0_ && tl == ev_t1_) return;
        std::vector<uint16_t> x(ev.size()), y(ev.size());
        std::vector<uint8_t> pol(ev.size());
        std::vector<int64_t> t(ev.size());
        parallel_chunks(ev.size(), [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) { x[k] = ev[k].x; y[k] = ev[k].y; pol[k] = ev[k].polarity ? 1 : 0; t[k] = ev[k].t_ns; } });
        check(emba_set_events(ctx_, x.data(), y.data(), pol.data(), t.data(), ev.size(), nullptr, nullptr, nullptr, 0));
        have_ev_ = true; ev_ptr_ = d; ev_n_ = ev.size(); ev_t0_ = tf; ev_t1_ = tl;
    }

    void form(NormalEquations& ne, const std::vector<double>& ep, int K, int thres, int irls, double a, bool dense)
    {
        if (K != K_) fatal(EMBA_ERR_INVALID_ARG, "num_ctrl_poses differs from the trajectory used in evaluateDataError");
        size_t P = 0, pl = 0;
        check(emba_form_active(ctx_, thres, &P, &pl));
        check(emba_form_accumulate(ctx_, ep.empty() ? nullptr : ep.data(), irls, a));
        P_ = P;
        finish(ne, 0.0, dense);
    }

    void finish(NormalEquations& ne, double alpha, bool dense)
    {
        const size_t P = P_, dim = 3 * (size_t)K_;
        ne.dim_ctrl_poses = (int)dim; ne.num_active_pixels = P;
        ne.A11.assign(dim * dim, 0.0); ne.b1.assign(dim, 0.0);
        ne.A22_blocks.assign(4 * P, 0.0); ne.b2.assign(2 * P, 0.0); ne.active_pix_idxes.assign(P, 0);
        if (dense) ne.A12.assign(dim * 2 * P, 0.0);
        check(emba_form_finish(ctx_, alpha, ne.A11.data(), ne.b1.data(), P ? ne.active_pix_idxes.data() : nullptr, P,
                               P ? ne.A22_blocks.data() : nullptr, P ? ne.b2.data() : nullptr, (dense && P) ? ne.A12.data() : nullptr));
    }

    emba_ctx* ctx_ = nullptr;
    int W_, H_, K_ = 0;
    size_t P_ = 0;
    bool have_ev_ = false; const Event* ev_ptr_ = nullptr; size_t ev_n_ = 0; int64_t ev_t0_ = 0, ev_t1_ = 0;
};

}  // namespace emba_host
