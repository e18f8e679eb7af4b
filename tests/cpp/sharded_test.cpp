// tests/cpp/sharded_test.cpp — the single-process multi-GPU host (emba_amd/host/legm_sharded.hpp over emba_group_*) on the GPU, checked
// against oracle values the Python test passes in through a binary file.  Usage: sharded_test <in.bin> <device,device,...>
// ("0,0" = two ranks on one GPU through the in-library exchange; "0" = one rank; distinct devices use RCCL).  File layout: host_test's,
// followed by double lambda ; int32 fix_first ; x1[3K] ; x2[2P] (the oracle's solveNormalEq on the same blocks).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../emba_amd/host/legm_sharded.hpp"

template <class T> static std::vector<T> rd(FILE* f, size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }
template <class T> static T rd1(FILE* f) { return rd<T>(f, 1)[0]; }
static double rel(const std::vector<double>& a, const std::vector<double>& b)
{
    if (a.size() != b.size()) return 1e300;
    double d = 0, s = 0;
    for (size_t i = 0; i < a.size(); ++i) { d = std::fmax(d, std::fabs(a[i] - b[i])); s = std::fmax(s, std::fabs(b[i])); }
    return d == 0 ? 0 : d / (s > 0 ? s : 1);
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<int> devices;
    for (char* tok = strtok(argv[2], ","); tok; tok = strtok(nullptr, ",")) devices.push_back(atoi(tok));
    const int x2_split = argc > 3 ? atoi(argv[3]) : -1;                       // option x2_split: -1 auto, 0 one piece, 1 split
    const uint32_t flags = argc > 4 ? (uint32_t)atoi(argv[4]) : 0u;           // EMBA_GROUP_FORCE_RCCL = 1
    const int sw = rd1<int32_t>(f), sh = rd1<int32_t>(f), W = rd1<int32_t>(f), H = rd1<int32_t>(f), K = rd1<int32_t>(f), thres = rd1<int32_t>(f);
    const int64_t t0 = rd1<int64_t>(f), dt = rd1<int64_t>(f), n = rd1<int64_t>(f);
    const double C_th = rd1<double>(f), alpha = rd1<double>(f);
    auto lut = rd<double>(f, (size_t)sw * sh * 3); auto knots = rd<double>(f, (size_t)K * 4);
    auto Gx = rd<double>(f, (size_t)W * H); auto Gy = rd<double>(f, (size_t)W * H);
    auto x = rd<uint16_t>(f, n); auto y = rd<uint16_t>(f, n); auto pol = rd<uint8_t>(f, n); auto t = rd<int64_t>(f, n);
    const int64_t m = rd1<int64_t>(f);
    auto ep_ref = rd<double>(f, m); auto nem_ref = rd<int32_t>(f, (size_t)W * H);
    const int64_t P = rd1<int64_t>(f);
    auto A11 = rd<double>(f, (size_t)9 * K * K); auto b1 = rd<double>(f, (size_t)3 * K);
    auto A22 = rd<double>(f, 4 * P); auto b2 = rd<double>(f, 2 * P); auto act = rd<uint32_t>(f, P);
    const double lambda = rd1<double>(f); const int fix = rd1<int32_t>(f);
    auto x1_ref = rd<double>(f, (size_t)3 * K); auto x2_ref = rd<double>(f, 2 * P);
    fclose(f);
    double cost_ref = 0; for (double e : ep_ref) cost_ref += 0.5 * e * e;

    emba_host::EventPacket ev(n);
    for (int64_t k = 0; k < n; ++k) ev[k] = {x[k], y[k], t[k], pol[k] != 0};
    try {
        emba_host::ShardedLEGM model(sw, sh, lut.data(), C_th, W, H, devices, flags);
        model.setOption("x2_split", x2_split);
        emba_host::TrajectoryView traj{knots.data(), K, t0, dt};
        model.setEvents(ev);
        model.uploadMap(Gx.data(), Gy.data());
        emba_host::NormalEquations ne;
        for (int it = 0; it < 2; ++it) {   // twice: the LM loop repeats these on one packet (solver.cpp:63-353)
            const size_t n_inl = model.iterate(traj, thres, "quadratic", 0.0, alpha);
            if ((int64_t)n_inl != m) { printf("FAIL inlier count %zu vs %lld\n", n_inl, (long long)m); return 1; }
            model.download(ne);
            if ((int64_t)ne.num_active_pixels != P || ne.active_pix_idxes != act) { printf("FAIL active set differs\n"); return 1; }
            const double e[4] = {rel(ne.A11, A11), rel(ne.b1, b1), rel(ne.A22_blocks, A22), rel(ne.b2, b2)};
            for (double v : e) if (!(v < 1e-9)) { printf("FAIL rel err %g %g %g %g\n", e[0], e[1], e[2], e[3]); return 1; }
            double reg = 0; for (size_t i = 0; i < Gx.size(); ++i) reg += Gx[i] * Gx[i] + Gy[i] * Gy[i];
            const double c = model.totalCost("quadratic", 0.0, alpha), c_ref = cost_ref + 0.5 * alpha * reg;
            if (!(std::fabs(c - c_ref) <= 1e-9 * std::fabs(c_ref))) { printf("FAIL cost %.12e vs %.12e\n", c, c_ref); return 1; }
        }
        std::vector<double> x1, x2;
        model.solveNormalEq(lambda, fix != 0, x1, x2);
        const double s1 = rel(x1, x1_ref), s2 = rel(x2, x2_ref);
        if (!(s1 < 1e-7) || !(s2 < 1e-7)) { printf("FAIL solve rel err x1 %g x2 %g\n", s1, s2); return 1; }
        // one LM trial on the device-resident map: update, re-evaluate, reject, and the original blocks must come back
        model.updateMap(x2, 1.0);
        model.iterate(traj, thres, "quadratic", 0.0, alpha);
        model.rejectMap();
        model.iterate(traj, thres, "quadratic", 0.0, alpha);
        model.download(ne);
        if (!(rel(ne.A11, A11) < 1e-9) || !(rel(ne.b2, b2) < 1e-9)) { printf("FAIL blocks after reject\n"); return 1; }
        printf("OK world=%d rccl=%d inliers=%lld P=%lld solve relerr x1=%.2e x2=%.2e\n", model.world(), (int)model.usesRccl(), (long long)m, (long long)P, s1, s2);
    } catch (const std::exception& e) { printf("FAIL exception %s\n", e.what()); return 1; }
    return 0;
}
