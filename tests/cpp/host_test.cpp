// tests/cpp/host_test.cpp — exercises the C++ host class (emba_amd/host/legm_host.hpp) over the C ABI on the GPU and
// checks it against values the Python parity test passes in through a binary file (produced from the oracle).
// Usage: host_test <in.bin> ; prints "OK" and max relative errors.  File layout (little endian):
//   int32 sw,sh,W,H,K,thres ; int64 t0,dt,n ; double C_th,alpha ; lut[sw*sh*3] ; knots[K*4] ; Gx[H*W] ; Gy[H*W] ;
//   x[n] u16 ; y[n] u16 ; pol[n] u8 ; t[n] i64 ; int64 m ; ep[m] ; num_ev_map[H*W] i32 ; int64 P ; A11[(3K)^2] ; b1[3K] ;
//   A22[4P] ; b2[2P] ; active[P] u32
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../emba_amd/host/legm_host.hpp"

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
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
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
    fclose(f);

    emba_host::EventPacket ev(n);
    for (int64_t k = 0; k < n; ++k) ev[k] = {x[k], y[k], t[k], pol[k] != 0};
    try {
        emba_host::LEGM model(sw, sh, lut.data(), C_th, W, H);
        emba_host::TrajectoryView traj{knots.data(), K, t0, dt};
        std::vector<int32_t> nem((size_t)W * H, -1);
        emba_host::NormalEquations ne;
        for (int it = 0; it < 2; ++it) {   // twice: the LM loop calls these repeatedly on one packet (solver.cpp:63-353)
            auto ep = model.evaluateDataError(traj, Gx.data(), Gy.data(), ev, true, nem.data());
            if ((int64_t)ep.size() != m) { printf("FAIL inlier count %zu vs %lld\n", ep.size(), (long long)m); return 1; }
            if (nem != nem_ref) { printf("FAIL num_ev_map differs\n"); return 1; }
            model.formNormalEq(ne, ep, K, thres);
            model.applyL2Reg(ne, alpha);
            if ((int64_t)ne.num_active_pixels != P || ne.active_pix_idxes != act) { printf("FAIL active set differs\n"); return 1; }
            const double e[5] = {rel(ep, ep_ref), rel(ne.A11, A11), rel(ne.b1, b1), rel(ne.A22_blocks, A22), rel(ne.b2, b2)};
            for (double v : e) if (!(v < 1e-9)) { printf("FAIL rel err %g %g %g %g %g\n", e[0], e[1], e[2], e[3], e[4]); return 1; }
            if (it == 1) printf("OK inliers=%lld P=%lld relerr ep=%.2e A11=%.2e b1=%.2e A22=%.2e b2=%.2e cost=%.6e\n", (long long)m, (long long)P,
                                e[0], e[1], e[2], e[3], e[4], model.evaluateRobustDataCost("quadratic", 0));
        }
        // fail-fast behaviour: a trajectory that does not cover the events must raise (BASALT_ASSERT in the reference)
        bool threw = false;
        try { emba_host::TrajectoryView bad{knots.data(), 2, t0, dt}; model.evaluateDataError(bad, Gx.data(), Gy.data(), ev, true, nem.data()); }
        catch (const std::runtime_error&) { threw = true; }
        if (!threw) { printf("FAIL no error for short trajectory\n"); return 1; }
    } catch (const std::exception& e) { printf("FAIL exception %s\n", e.what()); return 1; }
    return 0;
}
