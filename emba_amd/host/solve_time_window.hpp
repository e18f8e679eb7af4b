// emba_amd/host/solve_time_window.hpp — EMBA::solveTimeWindow (reference src/emba/solver.cpp:11-368), the Levenberg-Marquardt loop that CALLS the hot
// path, as a C++17 host on emba_host::ShardedLEGM with everything of an iteration resident in HBM.
//
// The drop-in adapter (legm_adapter.hpp) keeps solver.cpp unchanged and therefore moves what its signatures name across PCIe every iteration — the
// residual vector, the count map, the blocks, two std::set of pixel indices, two cloned map planes: 60 ms per LM iteration at config 2's shape where the
// device needs 4.7 (profiles/r04_adapter_timing.txt).  This header is what a maintainer puts in the place of solveTimeWindow's BODY to get the other
// 13x: the same loop — same constants, same control flow, same accept / reject and stopping rules, same log files — in which
//   evaluateDataError      -> ShardedLEGM::evaluateResident   (residuals, count map and per-event state stay on the device; solver.cpp:75,251)
//   0.5 ep.ep + reg cost   -> ShardedLEGM::costs              (two reductions, one 16-byte read-back;                    solver.cpp:88-91, 257-268)
//   formNormalEq[IRLS] + applyL2Reg -> ShardedLEGM::formResident                                                         (solver.cpp:114-130)
//   solveNormalEq[CG]      -> x1 (3K doubles) to the host, x2 stays on the device                                        (solver.cpp:190-202)
//   updateTraj             -> on the host, K poses: knot_i <- exp(x1_i) * knot_i                                         (model.cpp:22-53, trajectory.cpp:296-304)
//   Gx.clone() + updateMap -> ShardedLEGM::updateMapResident: the trial map is built on the device from ITS current map  (solver.cpp:237-240)
//   accept / reject        -> acceptMap / rejectMap: pointer swaps; a rejection goes back to untouched equations          (solver.cpp:299-352)
// Per iteration 3K doubles and two scalars cross PCIe.  Works on any number of ranks (one GPU: devices = {0}).  The reference's run-time records
// (runtime_formEqs.txt, runtime_solveEqs.txt, runtime_objFuncs.txt, iterations.txt, CG_iterations.txt under <result_dir>/final_results;
// solver.cpp:105-151, 170-178, 196-223, 271-291) are written by RuntimeLog in the reference's line formats, so analysis scripts keep working.
// Free of ROS / OpenCV / Eigen types like the classes it sits on; tests/cpp/resident_test.cpp drives it against emba_amd/solver.py's log and the
// CPU oracle's loop, decision for decision.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <system_error>
#include <vector>

#include "legm_sharded.hpp"

namespace emba_host {

struct LMSettings {               // include/emba/params.h:4-12, values of launch/shapes.launch:31-33
    int max_num_iter = 50;
    double tol_fun = 1e-3;
    int num_times_tol_fun_sat = 2;
};

struct BASettings {               // include/emba/params.h:14-61, values of launch/shapes.launch:28-60
    bool use_IRLS = false;
    std::string cost_type = "quadratic";
    double eta = 0.1;
    int thres_valid_pixel = 5;
    double alpha = 5.0;
    double damping_factor = 1.0;
    bool first_time_window = true;      // the first control pose is held fixed (solver.cpp:156-165, 227-230)
    bool use_CG = false;                // solveNormalEqCG instead of the Schur solve (solver.cpp:190-202)
};

struct LMLogEntry { int iter; double log10_lambda, cost_min, cost_new; bool accepted; size_t num_active; int cg_iter; };

struct LMResult {
    std::vector<double> knots_xyzw;     // the refined control poses (K unit quaternions x, y, z, w); the refined map stays on the device (ShardedLEGM::downloadMap)
    double cost_min = 0.0;
    int iterations = 0;
    bool converged = false;
    std::vector<LMLogEntry> log;
    double eval_ms = 0.0, form_ms = 0.0, solve_ms = 0.0, update_ms = 0.0;   // host wall time inside the loop by phase (evaluation + costs: ends in the read-back of the two sums;
                                                                            // form / solve / update: their calls — what they enqueue may still be running when they return)
    double setup_ms = 0.0, loop_ms = 0.0;   // wall time of the window's registration (setEvents: AoS -> SoA + upload + device-side ordering; map upload) and of the LM loop proper
};

// The run-time records of EMBA::solveTimeWindow in the reference's line formats.  Like its function-static counters the totals run over all the
// windows of a process: keep ONE object per run.  The reference casts each duration to whole milliseconds before summing; the device's phases are
// shorter than that, so the sums here are of the unrounded seconds (same fields, finer values).
class RuntimeLog {
public:
    explicit RuntimeLog(const std::string& result_dir) : dir_(result_dir + "/final_results")
    {
        std::error_code ec;      // (no shell: a result_dir with a quote in it is a directory name, not a command — ADVICE r5; and no fork of a process that holds the GPU)
        std::filesystem::create_directories(dir_, ec);
        if (ec) throw std::runtime_error("cannot create " + dir_ + ": " + ec.message());
        FILE* f = std::fopen((dir_ + "/iterations.txt").c_str(), "w");       // emba.cpp:223 opens it afresh
        if (f) std::fclose(f);
    }
    void newWindow()                                                            // solver.cpp:55-59
    {
        ++window_;
        app("iterations.txt", "window #" + std::to_string(window_));
        app("iterations.txt", "---------------------------------------------------------");
    }
    enum Key { FormEqs = 0, SolveEqs = 1, ObjFunc = 2 };
    void add(Key key, int it, double seconds, long Np = -1)
    {
        n_[key] += 1; t_[key] += seconds;
        char buf[256];
        if (key == FormEqs)
            std::snprintf(buf, sizeof buf, "iter #%d count_formEqs = %ld sec_total_formEqs = %.6g sec_average_formEqs = %.6g", it, n_[key], t_[key], t_[key] / n_[key]);
        else if (key == SolveEqs)
            std::snprintf(buf, sizeof buf, "iter #%d count_solveEqs = %ld sec_total_solveEqs = %.9g sec_average_solveEqs = %.9g", it, n_[key], t_[key], t_[key] / n_[key]);
        else
            std::snprintf(buf, sizeof buf, "iter #%d count_obj_func = %ld sec_total_obj_func = %.9g sec_average_obj_func = %.9g Np = %ld", it, n_[key], t_[key],
                          t_[key] / n_[key], Np);
        app(key == FormEqs ? "runtime_formEqs.txt" : key == SolveEqs ? "runtime_solveEqs.txt" : "runtime_objFuncs.txt", buf);
    }
    void iteration(int it, double lambda, double cost_min, double cost_new, double cost_data, double cost_reg)      // solver.cpp:170-178
    {
        char buf[320];
        std::snprintf(buf, sizeof buf, "iter #%d:  log10(lambda) = %g  cost_min^2 = %g  cost_new^2 = %g  cost_data = %g  cost_reg = %g", it, std::log10(lambda), cost_min,
                      cost_new, cost_data, cost_reg);
        app("iterations.txt", buf);
    }
    void cg(int it, int iters, double err)                                     // solver.cpp:196-202
    {
        char buf[128];
        std::snprintf(buf, sizeof buf, "iter #%d iter_times = %d error = %g", it, iters, err);
        app("CG_iterations.txt", buf);
    }

private:
    void app(const std::string& name, const std::string& line)
    {
        FILE* f = std::fopen((dir_ + "/" + name).c_str(), "a");
        if (!f) return;
        std::fprintf(f, "%s\n", line.c_str());
        std::fclose(f);
    }
    std::string dir_;
    long n_[3] = {0, 0, 0};
    double t_[3] = {0.0, 0.0, 0.0};
    int window_ = 0;
};

// Model::updateTraj + LinearTrajectory::incrementalUpdate (model.cpp:22-53, trajectory.cpp:296-304): knot_i <- exp(x1_i) * knot_i, Sophus' exp
// (so3.hpp:583-619) and quaternion product (:324-339).  x1 has 3K entries (zeros for a fixed first pose, as the device solve returns it).
inline void incrementalUpdate(std::vector<double>& knots_xyzw, const std::vector<double>& x1, bool fix_first_pose)
{
    const size_t K = knots_xyzw.size() / 4;
    for (size_t i = fix_first_pose ? 1 : 0; i < K; ++i) {
        const double wx = x1[3 * i], wy = x1[3 * i + 1], wz = x1[3 * i + 2];
        const double th2 = wx * wx + wy * wy + wz * wz;
        double imag, real;
        if (th2 < 1e-20) { imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0; real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0; }
        else { const double th = std::sqrt(th2); imag = std::sin(0.5 * th) / th; real = std::cos(0.5 * th); }
        const double ax = imag * wx, ay = imag * wy, az = imag * wz, aw = real;
        double* b = &knots_xyzw[4 * i];
        const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
        double q[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                       aw * bw - ax * bx - ay * by - az * bz};
        const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; ++k) b[k] = q[k] / nrm;
    }
}

// EMBA::solveTimeWindow(traj, Gx, Gy, events) — solver.cpp:11-368 — resident on the device(s) behind `model`.
// traj: the window's initial control poses + spline timing; Gx / Gy: the initial map planes (H x W, row-major, uploaded once);
// rl: optional run-time records.  Returns the refined poses, the final cost and the iteration log; model.downloadMap() gives the refined map.
inline LMResult solveTimeWindow(ShardedLEGM& model, const TrajectoryView& traj0, const EventPacket& events, const double* Gx, const double* Gy,
                                const BASettings& ba = BASettings(), const LMSettings& lm = LMSettings(), RuntimeLog* rl = nullptr)
{
    using clock = std::chrono::steady_clock;
    auto secs = [](clock::time_point a) { return std::chrono::duration<double>(clock::now() - a).count(); };
    const int K = traj0.num_ctrl_poses;
    std::vector<double> knots(traj0.knots_xyzw, traj0.knots_xyzw + 4 * (size_t)K), knots_new;
    auto view = [&](const std::vector<double>& k) { return TrajectoryView{k.data(), K, traj0.t0_ns, traj0.dt_ns}; };
    const std::string cost_type = ba.use_IRLS ? ba.cost_type : std::string("quadratic");
    if (rl) rl->newWindow();

    double lambda = 1e-3; const double lambda_max = 1e3, lambda_min = 1e-300;                                  // solver.cpp:15-17
    double cost_min = 1e99, cost_min_old = 1e99, cost_new = 1e99, cost_data = 0.0, cost_reg = 0.0, cost_data_new = 0.0, cost_reg_new = 0.0;
    int iter = 0, count_tol = 0;
    bool cost_has_decreased = true;
    LMResult res;
    const auto t_setup = clock::now();
    model.setEvents(events);
    model.setCost(cost_type, ba.use_IRLS ? ba.eta : 0.0);        // evaluations accumulate the (IRLS-weighted) per-pixel sums directly: speed only
    model.uploadMap(Gx, Gy);
    res.setup_ms = secs(t_setup) * 1e3;
    const auto t_loop = clock::now();

    auto evaluate = [&](const std::vector<double>& k, double& d, double& r) {      // evaluateDataError + the two cost terms: ONE host synchronisation
        const auto t0 = clock::now();
        model.evaluateResident(view(k));
        model.costs(cost_type, ba.eta, ba.alpha, d, r);
        res.eval_ms += secs(t0) * 1e3;
        return d + r;
    };

    while (iter <= lm.max_num_iter && cost_min > 1e-16 && lambda <= lambda_max && lambda >= lambda_min) {      // solver.cpp:63-64
        if (cost_has_decreased) {
            if (iter == 0) cost_min = evaluate(knots, cost_data, cost_reg);                                     // :69-91
            const auto t0 = clock::now();
            model.formResident(ba.thres_valid_pixel, cost_type, ba.eta, ba.alpha);                              // :93-131
            res.form_ms += secs(t0) * 1e3;
            if (rl) { model.sync(); rl->add(RuntimeLog::FormEqs, iter, secs(t0)); }                             // :105-151
        }
        if (rl) rl->iteration(iter, lambda, cost_min, cost_new, cost_data, cost_reg);                           // :170-178
        std::vector<double> x1;
        int cg_it = -1;
        bool numeric_failure = false;
        {
            const auto t0 = clock::now();
            try {
                if (ba.use_CG) {                                                                                // :196-202
                    std::vector<double> x2;
                    const std::pair<int, double> r = model.solveNormalEqCG(lambda, ba.first_time_window, x1, x2);
                    cg_it = r.first;
                    if (rl) rl->cg(iter, r.first, r.second);
                    model.updateMap(x2, ba.damping_factor);                                                     // (x2 of the CG solve goes through the host, as in the reference)
                } else {
                    model.solveNormalEqResident(lambda, ba.first_time_window, x1);                              // :190-194; x2 stays on the device
                }
            } catch (const StatusError& e) {
                // EMBA_ERR_NUMERIC: a 2x2 block A22_i + lambda diag(A22_i) is not positive definite.  The reference's A22m_i.inverse() (model.cpp:750)
                // yields inf / nan there, x1 / x2 and the trial cost become NaN, `cost_new < cost_min` is false and the step is rejected
                // (solver.cpp:340-352): the same decision, without evaluating the NaN trial point
                if (e.status != EMBA_ERR_NUMERIC) throw;
                numeric_failure = true;
            }
            res.solve_ms += secs(t0) * 1e3;
            if (rl) { model.sync(); rl->add(RuntimeLog::SolveEqs, iter, secs(t0)); }                            // :205-223
        }
        if (numeric_failure) {
            iter += 1;
            // (the reference evaluates the NaN trial point and records that evaluation too, solver.cpp:271-291: a record of zero duration keeps the file's line count)
            if (rl) rl->add(RuntimeLog::ObjFunc, iter, 0.0, (long)model.numActivePixels());
            res.log.push_back({iter, std::log10(lambda), cost_min, INFINITY, false, model.numActivePixels(), cg_it});
            cost_has_decreased = false; lambda *= 10; count_tol = 0;
            continue;
        }
        knots_new = knots;                                                                                      // :226-234
        incrementalUpdate(knots_new, x1, ba.first_time_window);
        { const auto t0 = clock::now(); if (!ba.use_CG) model.updateMapResident(ba.damping_factor); res.update_ms += secs(t0) * 1e3; }   // :237-240
        const size_t n_active = model.numActivePixels();
        {
            const auto t0 = clock::now();
            cost_new = evaluate(knots_new, cost_data_new, cost_reg_new);                                        // :251-268
            if (rl) rl->add(RuntimeLog::ObjFunc, iter + 1, secs(t0), (long)n_active);                           // :271-291 (after iter += 1 in the reference)
        }
        iter += 1;
        const bool accepted = cost_new < cost_min;
        res.log.push_back({iter, std::log10(lambda), cost_min, cost_new, accepted, n_active, cg_it});
        if (accepted) {                                                                                         // :299-339
            cost_has_decreased = true;
            knots.swap(knots_new);
            model.acceptMap();
            lambda /= 10;
            cost_min_old = cost_min; cost_min = cost_new; cost_data = cost_data_new; cost_reg = cost_reg_new;
            if (std::fabs(1 - cost_min / (cost_min_old + 1e-10)) < lm.tol_fun) {
                if (++count_tol >= lm.num_times_tol_fun_sat) { res.converged = true; break; }
            }
        } else {                                                                                                // :340-352
            cost_has_decreased = false;
            model.rejectMap();           // the equations the trial evaluation set aside are current again: the solver is called again on untouched A, b
            lambda *= 10;
            count_tol = 0;
        }
    }
    res.knots_xyzw = knots; res.cost_min = cost_min; res.iterations = iter;
    res.loop_ms = secs(t_loop) * 1e3;
    return res;
}

}  // namespace emba_host
