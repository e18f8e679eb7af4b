// tests/cpp/adapter_test.cpp — the drop-in `EMBA::LEGM` (emba_amd/host/legm_adapter.hpp) COMPILED against tests/cpp/mock_ref + the
// reference's vendored Eigen and driven on the GPU through the call order of EMBA::solveTimeWindow (reference src/emba/solver.cpp:63-353):
//   iter 0: evaluateDataError :75 ; when the cost has decreased: formNormalEq[IRLS] :114-126, applyL2Reg :130, first-window trim :156-165 ;
//   solveNormalEq :190-194 ; updateTraj :226-234 ; Gx.clone() + updateMap :237-240 ; evaluateDataError at the new point :251 ;
//   accept (copyTo, lambda /= 10, tolerance test) :299-339 or reject (lambda *= 10) :340-352.
// The loop below restates that ORDER with the reference's method signatures (it is the test's stand-in for solver.cpp, which needs ROS);
// every model call goes through the adapter exactly as solver.cpp's would.  Prints one line per LM iteration; the Python test compares
// them with the same loop on the CPU oracle.
// Usage: adapter_test <in.bin> <max_iter> <use_irls 0|1> <use_cg 0|1>      (file layout: tests/cpp/host_test.cpp's; devices from EMBA_HIP_DEVICES)
#define EMBA_LEGM_ADAPTER_SKETCH
#include "emba_amd/host/legm_adapter.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

template <class T> static std::vector<T> rd(FILE* f, size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }
template <class T> static T rd1(FILE* f) { return rd<T>(f, 1)[0]; }

// LinearTrajectory as far as the path sees it (include/utils/trajectory.h:109-190): control rotations + spline timing
class LinTraj : public Trajectory {
public:
    std::vector<Eigen::Quaterniond> q;
    size_t size() override { return q.size(); }
    MockSO3d getControlPose(const int idx) override { return MockSO3d{q[idx]}; }      // BY VALUE, like trajectory.h:47,135
    LinTraj* clone() const { return new LinTraj(*this); }
};

// Sophus::SO3d::exp (so3.hpp:583-619) as a quaternion
static Eigen::Quaterniond so3_exp(const Eigen::Vector3d& w)
{
    const double th2 = w.squaredNorm();
    double imag, real;
    if (th2 < 1e-20) { imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0; real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0; }
    else { const double th = std::sqrt(th2); imag = std::sin(0.5 * th) / th; real = std::cos(0.5 * th); }
    return Eigen::Quaterniond(real, imag * w.x(), imag * w.y(), imag * w.z());
}
// Model::updateTraj(traj, x1, idx_beg) + LinearTrajectory::incrementalUpdate (model.cpp:22-53, trajectory.cpp:296-304): knot <- exp(dx) * knot
static void update_traj(LinTraj* t, const EMBA::VecXd& x1, int idx_beg)
{
    for (size_t i = idx_beg; i < t->q.size(); ++i) {
        const Eigen::Vector3d d = x1.segment<3>(3 * (i - idx_beg));
        Eigen::Quaterniond r = so3_exp(d) * t->q[i];
        r.normalize();
        t->q[i] = r;
    }
}

static double robust_cost(const EMBA::VecXd& ep, int irls, double a)      // 0.5 ep.ep (solver.cpp:88) / evaluateRobustDataCost (model.cpp:279-314)
{
    double c = 0;
    for (Eigen::Index i = 0; i < ep.size(); ++i) {
        const double e = ep(i);
        if (irls == 0) c += 0.5 * e * e;
        else if (irls == 2) c += 0.5 / a * std::log1p(a * e * e);
        else { const double m = std::fabs(e); c += (m < a) ? 0.5 * m * m : a * m - 0.5 * a * a; }
    }
    return c;
}
static double reg_cost(const cv::Mat& Gx, const cv::Mat& Gy, double alpha)      // alpha*0.5*|evaluateRegError|^2 (model.cpp:260-277, solver.cpp:90)
{
    double s = 0; const size_t n = (size_t)Gx.rows * Gx.cols; const double* a = Gx.ptr<double>(); const double* b = Gy.ptr<double>();
    for (size_t i = 0; i < n; ++i) s += a[i] * a[i] + b[i] * b[i];
    return 0.5 * alpha * s;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const int max_iter = atoi(argv[2]); const bool use_irls = atoi(argv[3]) != 0, use_cg = atoi(argv[4]) != 0;
    const int sw = rd1<int32_t>(f), sh = rd1<int32_t>(f), W = rd1<int32_t>(f), H = rd1<int32_t>(f), K = rd1<int32_t>(f), thres = rd1<int32_t>(f);
    const int64_t t0 = rd1<int64_t>(f), dt = rd1<int64_t>(f), n = rd1<int64_t>(f);
    const double C_th = rd1<double>(f), alpha = rd1<double>(f);
    auto lut = rd<double>(f, (size_t)sw * sh * 3); auto knots = rd<double>(f, (size_t)K * 4);
    auto gx = rd<double>(f, (size_t)W * H); auto gy = rd<double>(f, (size_t)W * H);
    auto x = rd<uint16_t>(f, n); auto y = rd<uint16_t>(f, n); auto pol = rd<uint8_t>(f, n); auto t = rd<int64_t>(f, n);
    fclose(f);
    const std::string cost_type = "huber"; const double eta = 0.1; const int irls = use_irls ? 1 : 0;
    const double damping = 1.0, tol_fun = 1e-3; const int num_times_tol = 2;

    auto& table = dvs::EventWarper::mockBearingTable();
    table.resize((size_t)sw * sh);
    for (size_t i = 0; i < table.size(); ++i) table[i] = {lut[3 * i], lut[3 * i + 1], lut[3 * i + 2]};
    sensor_msgs::CameraInfo info; info.width = sw; info.height = sh;

    EMBA::EventPacket events(n);
    auto fill_events = [&](bool scrambled) {
        for (int64_t k = 0; k < n; ++k) {
            events[k].x = x[k]; events[k].y = y[k]; events[k].ts.sec = (uint32_t)(t[k] / 1000000000LL); events[k].ts.nsec = (uint32_t)(t[k] % 1000000000LL); events[k].polarity = pol[k];
            // "another window in the same allocation": same count, same first / middle / last event, other pixels in between
            if (scrambled && k > 0 && k != n / 2 && k != n - 1) { events[k].x = (uint16_t)((x[k] + 7) % sw); events[k].y = (uint16_t)((y[k] + 3) % sh); }
        }
    };
    LinTraj* traj = new LinTraj();
    for (int i = 0; i < K; ++i) traj->q.emplace_back(knots[4 * i + 3], knots[4 * i], knots[4 * i + 1], knots[4 * i + 2]);
    traj->mockSetTiming(t0, dt);
    cv::Mat Gx = cv::Mat::zeros(H, W, CV_64FC1), Gy = cv::Mat::zeros(H, W, CV_64FC1);
    std::memcpy(Gx.ptr<double>(), gx.data(), gx.size() * 8); std::memcpy(Gy.ptr<double>(), gy.data(), gy.size() * 8);

    // ---- a PREDECESSOR at the address the model under test will live at (a sliding-window host that re-creates its model): it evaluates another
    // packet in the same allocation, forms, solves and leaves a trial map + a resident x2 behind, then goes away.  ~LEGM is inline in the
    // reference header, so the adapter never sees the destruction: whatever it keyed by `this` must not reach the successor.
    alignas(EMBA::LEGM) static unsigned char slot[sizeof(EMBA::LEGM)];
    {
        fill_events(true);
        EMBA::LEGM* pre = new (slot) EMBA::LEGM(info, C_th, W, H);
        cv::Mat nm = cv::Mat::zeros(H, W, CV_32SC1);
        EMBA::MatXd a11, a12; std::vector<EMBA::Mat2d> a22; EMBA::VecXd c1, c2, y1, y2;
        std::set<size_t> act, inact;
        EMBA::VecXd e0 = pre->evaluateDataError(traj, Gx, Gy, events, true, nm);
        pre->formNormalEq(a11, a12, a22, c1, c2, e0, K, nm, thres, act, inact);
        pre->applyL2Reg(a22, c2, act, alpha, Gx, Gy);
        pre->solveNormalEq(a11, a12, a22, c1, c2, 1e-2, y1, y2);
        cv::Mat gxn = Gx.clone(), gyn = Gy.clone();
        pre->updateMap(gxn, gyn, y2, 1.0, act, inact);
        pre->~LEGM();
    }
    fill_events(false);                              // the real packet, in the SAME allocation (same size, same first / middle / last timestamp)
    EMBA::LEGM& model = *new (slot) EMBA::LEGM(info, C_th, W, H);
    const bool timing = getenv("ADAPTER_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_iter = now();
    // ADAPTER_TIMING: wall time inside each of the model's calls and in the loop's own host work, per LM iteration (scripts/adapter_timing.py)
    double tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // eval, form + L2, trim, solve, clones, updateMap, costs, accept copies
    auto lap = [&](int k, const std::chrono::steady_clock::time_point& t0) { if (timing) tp[k] += std::chrono::duration<double, std::milli>(now() - t0).count(); };

    // ---- solveTimeWindow's state (solver.cpp:15-61)
    double lambda = 1e-3; const double lambda_max = 1e3, lambda_min = 1e-300;
    double cost_min = 1e99, cost_min_old = 1e99, cost_new = 1e99;
    int iter = 0, count_tol = 0; bool cost_has_decreased = true; const bool first_time_window = true;
    cv::Mat num_ev_map = cv::Mat::zeros(H, W, CV_32SC1), num_ev_map_new = cv::Mat::zeros(H, W, CV_32SC1);
    EMBA::MatXd A11, A12; std::vector<EMBA::Mat2d> A22_blocks; EMBA::VecXd b1, b2, x1, x2, ep, ep_new;
    std::set<size_t> active, inactive;
    bool converged = false;
    while (iter <= max_iter && cost_min > 1e-16 && lambda <= lambda_max && lambda >= lambda_min) {          // :63-64
        if (cost_has_decreased) {
            if (iter == 0) {                                                                                 // :69-91
                ep = model.evaluateDataError(traj, Gx, Gy, events, true, num_ev_map);
                cost_min = robust_cost(ep, irls, eta) + reg_cost(Gx, Gy, alpha);
            } else {
                ep = ep_new; num_ev_map_new.copyTo(num_ev_map);                                             // :99-102
            }
            auto tf = now();
            if (use_irls) model.formNormalEqIRLS(A11, A12, A22_blocks, b1, b2, ep, K, num_ev_map, thres, active, inactive, cost_type, eta);   // :114-119
            else model.formNormalEq(A11, A12, A22_blocks, b1, b2, ep, K, num_ev_map, thres, active, inactive);                                  // :122-126
            model.applyL2Reg(A22_blocks, b2, active, alpha, Gx, Gy);                                        // :130
            lap(1, tf);
            auto tt = now();
            if (first_time_window) {                                                                         // :156-165
                const size_t left = 3 * (K - 1);
                EMBA::MatXd A11_1st = A11.block(3, 3, left, left);
                EMBA::MatXd A12_1st = A12.block(3, 0, left, A12.cols());
                EMBA::VecXd b1_1st = b1.tail(left);
                A11 = A11_1st; A12 = A12_1st; b1 = b1_1st;
            }
            lap(2, tt);
        }
        int cg_it = -1;
        auto ts = now();
        if (!use_cg) model.solveNormalEq(A11, A12, A22_blocks, b1, b2, lambda, x1, x2);                      // :190-194
        else cg_it = model.solveNormalEqCG(A11, A12, A22_blocks, b1, b2, lambda, x1, x2).first;             // :196-202
        lap(3, ts);
        LinTraj* traj_new = traj->clone();                                                                   // :226-234
        update_traj(traj_new, x1, first_time_window ? 1 : 0);
        auto tc = now();
        cv::Mat Gx_new = Gx.clone(), Gy_new = Gy.clone();                                                    // :237-240
        lap(4, tc);
        auto tu = now();
        model.updateMap(Gx_new, Gy_new, x2, damping, active, inactive);
        lap(5, tu);
        auto te = now();
        ep_new = model.evaluateDataError(traj_new, Gx_new, Gy_new, events, true, num_ev_map_new);            // :251
        lap(0, te);
        auto tk = now();
        cost_new = robust_cost(ep_new, irls, eta) + reg_cost(Gx_new, Gy_new, alpha);                         // :257-268
        lap(6, tk);
        iter += 1;
        const bool accepted = cost_new < cost_min;
        printf("LM %d %.1f %.17g %.17g %d %zu %d\n", iter, std::log10(lambda), cost_min, cost_new, accepted ? 1 : 0, active.size(), cg_it);
        if (timing) {
            const auto t1 = now();
            printf("TIME %d %.3f ms (whole LM iteration through the adapter, incl. the loop's own Mat clones / copies)\n", iter, std::chrono::duration<double, std::milli>(t1 - t_iter).count());
            printf("PHASES %d evaluateDataError %.2f  formNormalEq+applyL2Reg %.2f  first-window trim %.2f  solve %.2f  Gx.clone x2 %.2f  updateMap %.2f  host cost sums %.2f  (ms)\n",
                   iter, tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6]);
            for (double& v : tp) v = 0;
        }
        if (accepted) {                                                                                      // :299-339
            cost_has_decreased = true;
            delete traj; traj = traj_new;
            Gx_new.copyTo(Gx); Gy_new.copyTo(Gy);
            lambda /= 10; cost_min_old = cost_min; cost_min = cost_new;
            if (std::fabs(1 - cost_min / (cost_min_old + 1e-10)) < tol_fun) { if (++count_tol >= num_times_tol) { converged = true; break; } }
        } else {                                                                                             // :340-352
            cost_has_decreased = false; delete traj_new;
            lambda *= 10; count_tol = 0;
        }
        t_iter = now();
    }
    printf("END %d %d %.17g\n", iter, converged ? 1 : 0, cost_min);
    for (int i = 0; i < K; ++i) printf("KNOT %.17g %.17g %.17g %.17g\n", traj->q[i].x(), traj->q[i].y(), traj->q[i].z(), traj->q[i].w());
    double sx = 0, sy = 0; const size_t np = (size_t)W * H;
    for (size_t i = 0; i < np; ++i) { sx += Gx.ptr<double>()[i] * (double)((i % 7) + 1); sy += Gy.ptr<double>()[i] * (double)((i % 5) + 1); }
    printf("MAP %.17g %.17g\n", sx, sy);
    delete traj;
    model.~LEGM();
    return 0;
}
