// emba_amd/host/legm_sharded.hpp — the reference's measurement model for a node with several GPUs, driven by ONE host thread.
//
// The reference front-end is a single process that owns a single `EMBA::LEGM` (reference src/emba/emba.cpp:378) and calls it from
// `EMBA::solveTimeWindow` (src/emba/solver.cpp:63-353).  `emba_host::ShardedLEGM` keeps that shape: one object, the same call order
// (evaluate / form / applyL2Reg fused into iterate(), solveNormalEq, updateMap, accept / reject), with the window's events sharded by
// time over the devices behind it.  All of it is the C ABI's emba_group_* (include/emba_hip.h): N contexts on N
// devices, the two per-iteration exchanges as grouped RCCL all-reduces on the contexts' streams, the Schur solve on records
// re-distributed by pixel owner.  No Python, no extra processes or threads.  Free of ROS / OpenCV / Eigen types like legm_host.hpp.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "legm_host.hpp"

namespace emba_host {

class ShardedLEGM {
public:
    // devices: one entry per rank.  Distinct devices: RCCL over xGMI.  All equal (e.g. {0, 0}): ranks share a GPU (tests).
    // flags: EMBA_GROUP_FORCE_RCCL / EMBA_GROUP_NO_THREADS (include/emba_hip.h)
    ShardedLEGM(int sensor_w, int sensor_h, const double* bearing_lut, double C_th, int pano_width, int pano_height, const std::vector<int>& devices,
                uint32_t flags = 0)
        : W_(pano_width), H_(pano_height)
    {
        emba_cfg cfg{};
        cfg.sensor_w = sensor_w; cfg.sensor_h = sensor_h; cfg.pano_w = pano_width; cfg.pano_h = pano_height;
        cfg.bearing_lut = bearing_lut; cfg.C_th = C_th; cfg.event_batch = 100; cfg.outlier_px = 10.0;
        std::vector<int32_t> dev(devices.begin(), devices.end());
        const emba_status st = emba_group_create_flags(&cfg, dev.data(), (int32_t)dev.size(), flags, &g_);
        if (st != EMBA_OK) throw StatusError(st, std::string("emba_group_create: ") + emba_last_error(nullptr));
    }
    // tuning / A-B switches by name (emba_group_set_option: every rank's emba_set_option + the group's own x2_split); results do not depend on them
    void setOption(const std::string& name, int value) { check(emba_group_set_option(g_, name.c_str(), (int32_t)value)); }
    ~ShardedLEGM() { emba_group_destroy(g_); }
    ShardedLEGM(const ShardedLEGM&) = delete;
    ShardedLEGM& operator=(const ShardedLEGM&) = delete;

    int world() const { return emba_group_size(g_); }
    bool usesRccl() const { return emba_group_uses_rccl(g_) != 0; }

    // the EventPacket argument of evaluateDataError (model.h:83-84), once per window
    void setEvents(const EventPacket& ev)
    {
        // struct of arrays for the C ABI; the staging vectors are kept (a sliding-window run registers a window of about the same size again and
        // again: 13 B per event of freshly mapped pages each time cost more than the conversion itself — 10 M events: ~25 ms)
        sx_.resize(ev.size()); sy_.resize(ev.size()); sp_.resize(ev.size()); st_.resize(ev.size());
        parallel_chunks(ev.size(), [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) { sx_[k] = ev[k].x; sy_[k] = ev[k].y; sp_[k] = ev[k].polarity ? 1 : 0; st_[k] = ev[k].t_ns; } });
        check(emba_group_set_events(g_, sx_.data(), sy_.data(), sp_.data(), st_.data(), ev.size()));
    }
    void uploadMap(const double* Gx, const double* Gy) { check(emba_group_upload_map(g_, Gx, Gy)); }

    // ---- the reference's call shape (what emba_amd/host/legm_adapter.hpp forwards to; solver.cpp:63-353 calls these in this order) ----
    // VecXd evaluateDataError(traj, Gx, Gy, events, eval_deriv, num_ev_map)   model.cpp:72-258.  Gx == Gy == nullptr: evaluate on the map
    // resident on the device (the trial map updateMap built).  Returns ep of ALL ranks merged into the reference's order; num_ev_map (may
    // be nullptr) receives the global count map.
    std::vector<double> evaluateDataError(const TrajectoryView& traj, const double* Gx, const double* Gy, const EventPacket& events, bool eval_deriv,
                                          int32_t* num_ev_map)
    {
        if (!eval_deriv) throw std::runtime_error("eval_deriv=false is never used by the reference (solver.cpp:75,251) and is not provided");
        ensureEvents(events);
        K_ = traj.num_ctrl_poses;
        std::vector<double> ep(events.size() ? events.size() : 1);
        size_t n_inl = 0;
        check(emba_group_eval(g_, traj.knots_xyzw, traj.num_ctrl_poses, traj.t0_ns, traj.dt_ns, Gx, Gy, ep.data(), &n_inl, num_ev_map));
        ep.resize(n_inl);
        n_inliers_ = n_inl;
        return ep;
    }
    // The same in two steps (round 6), for a caller that must RETURN ep by value: the evaluation with the inlier count (and num_ev_map) only, then the residuals
    // straight into the caller's vector — no 8 B x events staging vector, no second copy (the adapter: 35 -> see profiles/r06_adapter_timing.txt).
    size_t evaluateDataErrorCount(const TrajectoryView& traj, const double* Gx, const double* Gy, const EventPacket& events, bool eval_deriv, int32_t* num_ev_map)
    {
        if (!eval_deriv) throw std::runtime_error("eval_deriv=false is never used by the reference (solver.cpp:75,251) and is not provided");
        ensureEvents(events);
        K_ = traj.num_ctrl_poses;
        size_t n_inl = 0;
        check(emba_group_eval(g_, traj.knots_xyzw, traj.num_ctrl_poses, traj.t0_ns, traj.dt_ns, Gx, Gy, nullptr, &n_inl, num_ev_map));
        n_inliers_ = n_inl;
        return n_inl;
    }
    void fetchEp(double* dst, size_t n) { size_t m = 0; check(emba_group_get_ep(g_, dst, n, &m)); if (m != n) throw std::runtime_error("fetchEp: the inlier count changed"); }
    // formNormalEq / formNormalEqIRLS (model.cpp:316-687) on the device-resident residuals of the last evaluateDataError; applyL2Reg apart
    void formNormalEq(NormalEquations& ne, int num_ctrl_poses, int thres_valid_pixel, const std::string& cost_type = "quadratic", double a = 0.0)
    {
        if (num_ctrl_poses != K_) throw std::runtime_error("num_ctrl_poses differs from the trajectory used in evaluateDataError");
        size_t n_inl = 0;
        check(emba_group_form(g_, thres_valid_pixel, irls_code(cost_type), a, 0.0, &n_inl, &P_));
        download(ne);
    }
    void applyL2Reg(NormalEquations& ne, double alpha) { check(emba_group_apply_l2(g_, alpha)); download(ne); }
    // The same two calls for a host that owns the block storage (the EMBA::LEGM adapter: std::vector<Mat2d> and VecXd of the caller): the equations are formed,
    // P is known, the caller sizes its containers, and the 2x2 blocks {xx, xy, xy, yy} / b2 land in them directly — no second copy of 23 MB per call at
    // config 2's shape.  ne receives A11, b1 and the active list; its block vectors stay empty.
    size_t formNormalEqOnDevice(int num_ctrl_poses, int thres_valid_pixel, const std::string& cost_type = "quadratic", double a = 0.0)
    {
        if (num_ctrl_poses != K_) throw std::runtime_error("num_ctrl_poses differs from the trajectory used in evaluateDataError");
        size_t n_inl = 0;
        check(emba_group_form(g_, thres_valid_pixel, irls_code(cost_type), a, 0.0, &n_inl, &P_));
        return P_;
    }
    void downloadInto(NormalEquations& ne, double* A22 /* 4 P */, double* b2 /* 2 P */)
    {
        const size_t P = P_, dim = 3 * (size_t)K_;
        ne.dim_ctrl_poses = (int)dim; ne.num_active_pixels = P;
        ne.A11.resize(dim * dim); ne.b1.resize(dim); ne.active_pix_idxes.resize(P); ne.A22_blocks.clear(); ne.b2.clear();
        check(emba_group_download(g_, ne.A11.data(), ne.b1.data(), P ? ne.active_pix_idxes.data() : nullptr, P, P ? A22 : nullptr, P ? b2 : nullptr));
    }
    void applyL2RegInto(double alpha, double* A22, double* b2)
    {
        check(emba_group_apply_l2(g_, alpha));
        if (P_) check(emba_group_download(g_, nullptr, nullptr, nullptr, P_, A22, b2));
    }
    // declare the robust cost of the formNormalEq calls to come (speed only: emba_group_set_cost)
    void setCost(const std::string& cost_type, double a) { check(emba_group_set_cost(g_, irls_code(cost_type), a)); }
    // solveNormalEqCG (model.cpp:794-840): returns (cg.iterations(), cg.error())
    std::pair<int, double> solveNormalEqCG(double lambda, bool fix_first_pose, std::vector<double>& x1, std::vector<double>& x2)
    {
        x1.assign(3 * (size_t)K_, 0.0); x2.assign(2 * P_, 0.0);
        int32_t it = 0; double err = 0;
        check(emba_group_solve_cg(g_, lambda, fix_first_pose ? 1 : 0, 100, 1e-6, x1.data(), P_ ? x2.data() : nullptr, &it, &err));
        return {it, err};
    }
    void rejectTrial() { check(emba_group_trial_reject(g_)); }

    // ---- the same phases with NOTHING returned to the host (the resident loop of solve_time_window.hpp) ----
    // evaluateDataError on the resident (current or trial) map: residuals, count map and per-event state stay on the device(s); enqueue only
    void evaluateResident(const TrajectoryView& traj)
    {
        K_ = traj.num_ctrl_poses;
        check(emba_group_eval(g_, traj.knots_xyzw, traj.num_ctrl_poses, traj.t0_ns, traj.dt_ns, nullptr, nullptr, nullptr, nullptr, nullptr));
    }
    // formNormalEq[IRLS] + applyL2Reg on that state; the blocks stay in the ranks' packs.  Returns the number of inlier measurements.
    size_t formResident(int thres_valid_pixel, const std::string& cost_type, double a, double alpha)
    {
        size_t n_inl = 0;
        check(emba_group_form(g_, thres_valid_pixel, irls_code(cost_type), a, alpha, &n_inl, &P_));
        n_inliers_ = n_inl;
        return n_inl;
    }
    // solveNormalEq with x1 to the host and x2 left on every rank's device (for updateMapResident)
    void solveNormalEqResident(double lambda, bool fix_first_pose, std::vector<double>& x1)
    {
        x1.assign(3 * (size_t)K_, 0.0);
        check(emba_group_solve(g_, lambda, fix_first_pose ? 1 : 0, x1.data(), nullptr));
    }
    // both cost terms of the last evaluation, one host synchronisation (solver.cpp:88-91, 257-268)
    void costs(const std::string& cost_type, double a, double alpha, double& data_cost, double& reg_cost)
    {
        check(emba_group_costs(g_, irls_code(cost_type), a, alpha, &data_cost, &reg_cost));
    }
    void sync() { for (int r = 0; r < world(); ++r) if (emba_sync(emba_group_ctx(g_, r)) != EMBA_OK) throw StatusError(EMBA_ERR_HIP, emba_last_error(emba_group_ctx(g_, r))); }
    size_t numInliers() const { return n_inliers_; }

    // evaluateDataError (model.cpp:72-258) + formNormalEq[IRLS] (:316-687) + applyL2Reg (:689-719) on the resident map, over all ranks.
    // Returns the number of inlier measurements; numActivePixels() afterwards.
    size_t iterate(const TrajectoryView& traj, int thres_valid_pixel, const std::string& cost_type, double a, double alpha)
    {
        K_ = traj.num_ctrl_poses;
        size_t n_inl = 0;
        check(emba_group_step(g_, traj.knots_xyzw, traj.num_ctrl_poses, traj.t0_ns, traj.dt_ns, thres_valid_pixel, irls_code(cost_type), a, alpha, &n_inl, &P_));
        return n_inl;
    }
    size_t numActivePixels() const { return P_; }

    // the out-arguments of formNormalEq + applyL2Reg (reduced over the ranks)
    void download(NormalEquations& ne)
    {
        const size_t P = P_, dim = 3 * (size_t)K_;
        ne.dim_ctrl_poses = (int)dim; ne.num_active_pixels = P;
        ne.A11.resize(dim * dim); ne.b1.resize(dim); ne.A22_blocks.resize(4 * P); ne.b2.resize(2 * P); ne.active_pix_idxes.resize(P);      // (all of it overwritten: no zero fill)
        check(emba_group_download(g_, ne.A11.data(), ne.b1.data(), P ? ne.active_pix_idxes.data() : nullptr, P, P ? ne.A22_blocks.data() : nullptr,
                                  P ? ne.b2.data() : nullptr));
    }

    // 0.5*ep.dot(ep) / evaluateRobustDataCost (solver.cpp:88, model.cpp:279-314) + alpha*0.5*|evaluateRegError|^2 (model.cpp:260-277)
    double totalCost(const std::string& cost_type, double a, double alpha)
    {
        double d = 0, r = 0;
        check(emba_group_costs(g_, irls_code(cost_type), a, alpha, &d, &r));
        return d + r;
    }

    // solveNormalEq(A11, A12, A22_blocks, b1, b2, lambda, x1, x2)   model.cpp:721-792
    void solveNormalEq(double lambda, bool fix_first_pose, std::vector<double>& x1, std::vector<double>& x2)
    {
        x1.assign(3 * (size_t)K_, 0.0); x2.assign(2 * P_, 0.0);
        check(emba_group_solve(g_, lambda, fix_first_pose ? 1 : 0, x1.data(), P_ ? x2.data() : nullptr));
    }

    // updateMap (model.cpp:863-903) into the trial map; the LM decision (solver.cpp:299-352)
    void updateMap(const std::vector<double>& x2, double damping_factor) { check(emba_group_update_map(g_, x2.data(), damping_factor)); }
    // ... with the x2 the last solveNormalEq left on every rank's device (no upload)
    void updateMapResident(double damping_factor) { check(emba_group_update_map(g_, nullptr, damping_factor)); }
    void acceptMap() { check(emba_group_map_accept(g_)); }
    void rejectMap() { check(emba_group_map_reject(g_)); }
    void downloadMap(double* Gx, double* Gy) { check(emba_group_download_map(g_, Gx, Gy)); }
    // the trial map at the active pixels only — it is zero everywhere else (model.cpp:892-901): gxy[2i] = Gx[active_i], gxy[2i+1] = Gy[active_i]
    void mapAtActive(std::vector<double>& gxy) { gxy.assign(P_ ? 2 * P_ : 1, 0.0); check(emba_group_get_map_active(g_, gxy.data(), P_)); gxy.resize(2 * P_); }

    emba_group* group() { return g_; }

private:
    static int irls_code(const std::string& t) { return t == "cauchy" ? 2 : t == "huber" ? 1 : 0; }
    void check(emba_status st)
    {   // the reference aborts through glog CHECK / LOG(FATAL); callers that link glog can catch and LOG(FATAL)
        if (st != EMBA_OK) throw StatusError(st, emba_group_last_error(g_));
    }
    // the sliding window hands the same packet to every LM trial: registered once per CONTENT.  The key is the allocation, the size and a hash
    // over 4096 events spread evenly through the packet (all four fields) plus the first and the last one: a window that reuses the allocation
    // with other events of equal count — a re-subsampled window, say — differs at nearly every position, so it cannot hide from the sample;
    // hashing every event would cost more per call (16 MB at 1 M events) than the evaluation it guards.
    static uint64_t packetKey(const EventPacket& ev)
    {
        uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t)ev.size();
        auto mix = [&](const Event& e) {
            h ^= (uint64_t)e.t_ns; h *= 0x100000001B3ull; h ^= ((uint64_t)e.x << 32) | ((uint64_t)e.y << 8) | (e.polarity ? 1u : 0u); h *= 0x100000001B3ull;
        };
        if (ev.empty()) return h;
        const size_t step = ev.size() / 4096 ? ev.size() / 4096 : 1;
        for (size_t i = 0; i < ev.size(); i += step) mix(ev[i]);
        mix(ev.back());
        return h;
    }
    void ensureEvents(const EventPacket& ev)
    {
        const uint64_t key = packetKey(ev);
        if (have_ev_ && ev.data() == ev_ptr_ && ev.size() == ev_n_ && key == ev_key_) return;
        setEvents(ev);
        have_ev_ = true; ev_ptr_ = ev.data(); ev_n_ = ev.size(); ev_key_ = key;
    }
    emba_group* g_ = nullptr;
    int W_, H_, K_ = 0;
    size_t P_ = 0, n_inliers_ = 0;
    bool have_ev_ = false; const Event* ev_ptr_ = nullptr; size_t ev_n_ = 0; uint64_t ev_key_ = 0;
    std::vector<uint16_t> sx_, sy_; std::vector<uint8_t> sp_; std::vector<int64_t> st_;      // setEvents' staging (grow-only)
};

}  // namespace emba_host
