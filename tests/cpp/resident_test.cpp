// tests/cpp/resident_test.cpp — emba_host::solveTimeWindow (emba_amd/host/solve_time_window.hpp): the C++ resident host of the reference's
// Levenberg-Marquardt loop (src/emba/solver.cpp:11-368) on emba_host::ShardedLEGM, driven on the GPU.  Prints one line per LM iteration in
// adapter_test's format; tests/test_cpp_host.py compares them with emba_amd/solver.py's loop on the same device path and with the loop on the CPU
// oracle, decision for decision, and checks the run-time records written in the reference's formats.
// Usage: resident_test <in.bin> <devices> <max_iter> <use_irls 0|1> <use_cg 0|1> [result_dir] [repeat]      (file layout: tests/cpp/host_test.cpp's)
#include "../../emba_amd/host/solve_time_window.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <class T> static std::vector<T> rd(FILE* f, size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }
template <class T> static T rd1(FILE* f) { return rd<T>(f, 1)[0]; }

int main(int argc, char** argv)
{
    if (argc < 6) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<int> devices;
    for (char* tok = strtok(argv[2], ","); tok; tok = strtok(nullptr, ",")) devices.push_back(atoi(tok));
    const int max_iter = atoi(argv[3]); const bool use_irls = atoi(argv[4]) != 0, use_cg = atoi(argv[5]) != 0;
    const char* result_dir = argc > 6 && argv[6][0] ? argv[6] : nullptr;
    const int repeat = argc > 7 ? atoi(argv[7]) : 1;
    const int sw = rd1<int32_t>(f), sh = rd1<int32_t>(f), W = rd1<int32_t>(f), H = rd1<int32_t>(f), K = rd1<int32_t>(f), thres = rd1<int32_t>(f);
    const int64_t t0 = rd1<int64_t>(f), dt = rd1<int64_t>(f), n = rd1<int64_t>(f);
    const double C_th = rd1<double>(f), alpha = rd1<double>(f);
    auto lut = rd<double>(f, (size_t)sw * sh * 3); auto knots = rd<double>(f, (size_t)K * 4);
    auto gx = rd<double>(f, (size_t)W * H); auto gy = rd<double>(f, (size_t)W * H);
    auto x = rd<uint16_t>(f, n); auto y = rd<uint16_t>(f, n); auto pol = rd<uint8_t>(f, n); auto t = rd<int64_t>(f, n);
    fclose(f);
    emba_host::EventPacket ev(n);
    for (int64_t k = 0; k < n; ++k) ev[k] = {x[k], y[k], t[k], pol[k] != 0};
    try {
        emba_host::ShardedLEGM model(sw, sh, lut.data(), C_th, W, H, devices);
        emba_host::BASettings ba;
        ba.use_IRLS = use_irls; ba.cost_type = "huber"; ba.eta = 0.1; ba.thres_valid_pixel = thres; ba.alpha = alpha; ba.use_CG = use_cg;
        emba_host::LMSettings lm; lm.max_num_iter = max_iter;
        emba_host::RuntimeLog* rl = result_dir ? new emba_host::RuntimeLog(result_dir) : nullptr;
        emba_host::TrajectoryView traj{knots.data(), K, t0, dt};
        emba_host::LMResult r;
        for (int rep = 0; rep < repeat; ++rep) {         // repeat > 1: the same window again on the same context (steady state: buffers allocated) — timing
            const auto t_begin = std::chrono::steady_clock::now();
            r = emba_host::solveTimeWindow(model, traj, ev, gx.data(), gy.data(), ba, lm, rep == 0 ? rl : nullptr);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            const int it = r.iterations > 0 ? r.iterations : 1;
            printf("WINDOW %d %.3f ms, %d LM iterations: %.3f ms per iteration all in; registration (setEvents: AoS -> SoA, upload, device-side ordering; map upload) %.3f ms, "
                   "the LM loop proper %.3f ms = %.3f ms per iteration (host time per iteration: evaluation + costs %.3f, equations %.3f, solve %.3f, updateMap %.3f)\n", rep, ms,
                   r.iterations, ms / it, r.setup_ms, r.loop_ms, r.loop_ms / it, r.eval_ms / it, r.form_ms / it, r.solve_ms / it, r.update_ms / it);
        }
        for (const auto& e : r.log) printf("LM %d %.1f %.17g %.17g %d %zu %d\n", e.iter, e.log10_lambda, e.cost_min, e.cost_new, e.accepted ? 1 : 0, e.num_active, e.cg_iter);
        printf("END %d %d %.17g\n", r.iterations, r.converged ? 1 : 0, r.cost_min);
        for (int i = 0; i < K; ++i) printf("KNOT %.17g %.17g %.17g %.17g\n", r.knots_xyzw[4 * i], r.knots_xyzw[4 * i + 1], r.knots_xyzw[4 * i + 2], r.knots_xyzw[4 * i + 3]);
        std::vector<double> Gx((size_t)W * H), Gy((size_t)W * H);
        model.downloadMap(Gx.data(), Gy.data());
        double sx = 0, sy = 0;
        for (size_t i = 0; i < Gx.size(); ++i) { sx += Gx[i] * (double)((i % 7) + 1); sy += Gy[i] * (double)((i % 5) + 1); }
        printf("MAP %.17g %.17g\n", sx, sy);
        delete rl;
    } catch (const std::exception& e) { printf("FAIL exception %s\n", e.what()); return 1; }
    return 0;
}
