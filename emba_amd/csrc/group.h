// emba_amd/csrc/group.h — single-process multi-GPU host of the hot path, behind the C ABI (emba_group_*).
//
// The reference front-end is ONE process that owns ONE LEGM (src/emba/emba.cpp:378, called from solver.cpp:63-353).  An emba_group
// is its drop-in for a node with several GPUs: N contexts on N devices driven by the caller's single host thread, events sharded by
// time on the global 100-event batch grid with a per-sensor-pixel halo, and per Gauss-Newton iteration the two exchanges of §8e —
// X1 (count map, as saturated bytes when the activity threshold allows) and X2 (the fp64 pack [A11 | b1 | A22b2]) — as grouped RCCL
// all-reduces on the contexts' own HIP streams (ncclCommInitAll: one communicator per device, no extra threads or processes).  The
// Schur solve re-distributes the sparse A12 factors by pixel owner first (grouped ncclSend / ncclRecv), see emba_solve_shard_*.
//
// RCCL is bound at run time (dlopen of librccl.so.1) the first time a group spans DISTINCT devices, so the library carries no link
// dependency on it and never meets a second copy of it in a process that already has one (PyTorch ships its own).  Ranks that share a
// device (what a one-GPU test box can run: two contexts, two streams, devices = {0, 0}) and single-rank groups exchange through
// in-library copies and add kernels ordered by events instead — same protocol, same call sequence, no RCCL.
#pragma once
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <thread>

namespace emba {

__global__ void emba_add_f64_kernel(double* __restrict__ dst, const double* __restrict__ src, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
__global__ void emba_add_i32_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
__global__ void emba_add_u8_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint8_t)(dst[i] + src[i]);
}

// ranks that share a device: every buffer <- the sum of all of them, in one launch (n <= kLocalMax)
constexpr int kLocalMax = 16;
template <typename T> struct LocalBufs { T* p[kLocalMax]; int n; };
template <typename T>
__global__ void emba_sum_all_kernel(LocalBufs<T> b, long count)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T acc = b.p[0][i];
    for (int r = 1; r < b.n; ++r) acc = (T)(acc + b.p[r][i]);
    for (int r = 0; r < b.n; ++r) b.p[r][i] = acc;
}

}  // namespace emba

namespace {

// ---- RCCL, bound at run time --------------------------------------------------------------------------------------------------
struct Rccl {
    void* h = nullptr;
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    enum { kUint8 = 1, kInt32 = 2, kFloat64 = 8, kSum = 0 };   // rccl.h: ncclDataType_t / ncclRedOp_t
    bool load(std::string* err)
    {
        if (h) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) { *err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define EMBA_SYM(field, sym) do { *(void**)(&field) = dlsym(h, sym); if (!field) { *err = std::string("librccl lacks ") + sym; return false; } } while (0)
        EMBA_SYM(CommInitAll, "ncclCommInitAll"); EMBA_SYM(CommDestroy, "ncclCommDestroy"); EMBA_SYM(AllReduce, "ncclAllReduce");
        EMBA_SYM(Send, "ncclSend"); EMBA_SYM(Recv, "ncclRecv"); EMBA_SYM(GroupStart, "ncclGroupStart"); EMBA_SYM(GroupEnd, "ncclGroupEnd");
        EMBA_SYM(GetErrorString, "ncclGetErrorString");
#undef EMBA_SYM
        return true;
    }
};
Rccl g_rccl;

}  // namespace

// One host thread per rank, so that a step's launches on N devices are issued side by side instead of one rank after the other
// (8 ranks x ~6 launches x ~6 us of launch time is more than the 0.12 ms step they start at 1 M events per GPU).  Fork-join: run(fn) calls
// fn(rank) on every rank's thread and returns the first non-OK status.  The collectives themselves are issued by the caller's thread
// (ncclGroupStart / End around all ranks' calls).
struct RankPool {
    int n = 0;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go;
    std::function<emba_status(int)> job;
    std::vector<emba_status> st;
    std::atomic<long> gen{0};
    std::atomic<int> remaining{0};
    std::atomic<bool> stop{false};
    // A step is a handful of fork-joins a few tens of microseconds apart: the workers spin on the generation counter for a while before they
    // go to sleep on the condition variable (a futex wake-up per rank and phase would cost more than the phase's launches), the caller
    // spins on the completion count.
    static constexpr int kSpin = 20000;
    void start(int n_, const std::vector<int>& dev)
    {
        n = n_; st.assign(n, EMBA_OK);
        for (int r = 0; r < n; ++r)
            th.emplace_back([this, r, d = dev[r]]() {
                (void)hipSetDevice(d);
                long seen = 0;
                for (;;) {
                    int spins = 0;
                    while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed)) {
                        if (++spins < kSpin) { __builtin_ia32_pause(); continue; }
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait_for(lk, std::chrono::milliseconds(50), [&] { return stop.load() || gen.load() != seen; });
                    }
                    if (stop.load()) return;
                    seen = gen.load(std::memory_order_acquire);
                    st[r] = job(r);
                    remaining.fetch_sub(1, std::memory_order_acq_rel);
                }
            });
    }
    emba_status run(const std::function<emba_status(int)>& f)
    {
        if (th.empty()) { for (int r = 0; r < n; ++r) { const emba_status s = f(r); if (s) return s; } return EMBA_OK; }
        job = f;
        remaining.store(n, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(mu); gen.fetch_add(1, std::memory_order_acq_rel); }
        cv_go.notify_all();
        while (remaining.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
        for (int r = 0; r < n; ++r) if (st[r]) return st[r];
        return EMBA_OK;
    }
    void shutdown()
    {
        stop.store(true);
        { std::lock_guard<std::mutex> lk(mu); }
        cv_go.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
        th.clear();
    }
};

struct emba_group {
    int n = 0;
    int opt_x2_split = -1;      // -1 auto (from 3 M events per rank), 0 one piece, 1 split
    int opt_step_fast = 1;      // the ranks' forms run as resident steps where they can (emba_step_form_active); option group_step_fast = 0: the sweeping forms (A/B)
    std::vector<emba_ctx*> ctx;
    std::vector<int> dev;
    bool use_rccl = false;
    std::vector<Rccl::comm_t> comm;
    std::vector<hipEvent_t> ev;          // one per rank (local exchange ordering)
    hipEvent_t ev0 = nullptr;
    std::vector<hipStream_t> side;       // one per rank: exchange 2b (the A22 | b2 rows) runs here while the rank's own stream forms A11 | b1
    std::vector<hipEvent_t> ev_side, ev_side0;   // ordering between a rank's stream and its side stream / among the side streams
    std::string err;
    size_t npix = 0; int sw = 0;
    // exchange buffers, per rank, on the rank's device
    std::vector<int32_t*> count; std::vector<uint8_t*> count_u8; std::vector<double*> pack; size_t pack_cap = 0; int pack_K = 0;
    // per-iteration results
    size_t P = 0, n_inliers = 0; int K = 0;
    std::vector<size_t> n_local;
    std::vector<size_t> lo;              // first global event of every rank's range
    bool x1_done = false;                // the count maps of the last evaluation have been all-reduced already (emba_group_eval returned num_ev_map)
    int decl_irls = 0; double decl_eta = 0.0;   // robust cost declared for the evaluations (emba_group_set_cost)
    // grow-only scratch of the sharded solve, per rank (an LM loop calls it every iteration)
    std::vector<double*> sv_send, sv_recv, sv_S, sv_x2; std::vector<size_t> cap_send, cap_recv, cap_S, cap_x2;
    bool last_solve_exchanged = true;   // the last sharded solve ran the record exchange (false: every rank had its received records cached)
    bool x2_on_ranks = false;   // sv_x2[r] holds the all-reduced x2 of the last emba_group_solve (emba_group_update_map with x2_host == NULL)
    RankPool pool;
    int sw_ = 0;
    std::vector<uint16_t> ev_x, ev_y;    // sensor coordinates of the window's events (host copy): the merged residual vector of emba_group_eval is ordered by sensor pixel
};

namespace {

emba_status gfail(emba_group* g, emba_status st, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g->err = buf;
    return st;
}
#define G_TRY(g, r, call) do { emba_status st_ = (call); if (st_) return gfail((g), st_, "rank %d: %s", (r), emba_last_error((g)->ctx[(r)])); } while (0)
#define G_HIP(g, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return gfail((g), EMBA_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define G_NCCL(g, call) do { int e_ = (call); if (e_ != 0) return gfail((g), EMBA_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(e_)); } while (0)

// fork-join over the ranks' threads; a failure is reported with the first failing rank's message
emba_status gpool(emba_group* g, const std::function<emba_status(int)>& f)
{
    const emba_status st = g->pool.run(f);
    if (!st) return EMBA_OK;
    int bad = 0;
    if (!g->pool.th.empty()) { for (int r = 0; r < g->n; ++r) if (g->pool.st[r]) { bad = r; break; } }
    else { for (int r = 0; r < g->n; ++r) if (emba_last_error(g->ctx[r])[0]) { bad = r; break; } }
    return gfail(g, st, "rank %d: %s", bad, emba_last_error(g->ctx[bad]));
}

template <typename T>
emba_status grow(emba_group* g, int r, T** p, size_t* cap, size_t count)
{
    if (*p && *cap >= count) return EMBA_OK;
    G_HIP(g, hipSetDevice(g->dev[r]));
    if (*p) { G_HIP(g, hipStreamSynchronize(g->ctx[r]->stream)); (void)hipFree(*p); *p = nullptr; }
    const size_t want = std::max<size_t>(count + count / 8, 16);
    G_HIP(g, hipMalloc((void**)p, want * sizeof(T)));
    *cap = want;
    return EMBA_OK;
}

enum class XType { U8, I32, F64 };
inline size_t xsize(XType t) { return t == XType::U8 ? 1 : t == XType::I32 ? 4 : 8; }

// all-reduce(SUM) of bufs[r] (count elements each) over the ranks, in place, on the ranks' streams
emba_status group_allreduce(emba_group* g, void* const* bufs, size_t count, XType t, bool on_side = false)
{
    if ((g->n == 1 && !g->use_rccl) || count == 0) return EMBA_OK;
    auto S = [&](int r) { return on_side ? g->side[r] : g->ctx[r]->stream; };
    std::vector<hipEvent_t>& ev = on_side ? g->ev_side0 : g->ev;     // (the two exchanges may be in flight at once: separate events)
    hipEvent_t ev0 = on_side ? g->ev_side0[g->n] : g->ev0;
    if (g->use_rccl) {
        const int dt = t == XType::U8 ? Rccl::kUint8 : t == XType::I32 ? Rccl::kInt32 : Rccl::kFloat64;
        G_NCCL(g, g_rccl.GroupStart());
        for (int r = 0; r < g->n; ++r) {
            G_HIP(g, hipSetDevice(g->dev[r]));
            G_NCCL(g, g_rccl.AllReduce(bufs[r], bufs[r], count, dt, Rccl::kSum, g->comm[r], S(r)));
        }
        G_NCCL(g, g_rccl.GroupEnd());
        return EMBA_OK;
    }
    // ranks on one device: rank 0's stream waits for the producers, ONE kernel leaves the sum in every rank's buffer, the others wait for it
    G_HIP(g, hipSetDevice(g->dev[0]));
    hipStream_t s0 = S(0);
    for (int r = 1; r < g->n; ++r) { G_HIP(g, hipEventRecord(ev[r], S(r))); G_HIP(g, hipStreamWaitEvent(s0, ev[r], 0)); }
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (g->n <= emba::kLocalMax) {
        if (t == XType::F64) { emba::LocalBufs<double> b{}; b.n = g->n; for (int r = 0; r < g->n; ++r) b.p[r] = (double*)bufs[r]; hipLaunchKernelGGL(emba::emba_sum_all_kernel<double>, dim3(grid), dim3(256), 0, s0, b, (long)count); }
        else if (t == XType::I32) { emba::LocalBufs<int32_t> b{}; b.n = g->n; for (int r = 0; r < g->n; ++r) b.p[r] = (int32_t*)bufs[r]; hipLaunchKernelGGL(emba::emba_sum_all_kernel<int32_t>, dim3(grid), dim3(256), 0, s0, b, (long)count); }
        else { emba::LocalBufs<uint8_t> b{}; b.n = g->n; for (int r = 0; r < g->n; ++r) b.p[r] = (uint8_t*)bufs[r]; hipLaunchKernelGGL(emba::emba_sum_all_kernel<uint8_t>, dim3(grid), dim3(256), 0, s0, b, (long)count); }
        G_HIP(g, hipGetLastError());
        G_HIP(g, hipEventRecord(ev0, s0));
        for (int r = 1; r < g->n; ++r) G_HIP(g, hipStreamWaitEvent(S(r), ev0, 0));
        return EMBA_OK;
    }
    for (int r = 1; r < g->n; ++r) {
        if (t == XType::F64) hipLaunchKernelGGL(emba::emba_add_f64_kernel, dim3(grid), dim3(256), 0, s0, (double*)bufs[0], (const double*)bufs[r], (long)count);
        else if (t == XType::I32) hipLaunchKernelGGL(emba::emba_add_i32_kernel, dim3(grid), dim3(256), 0, s0, (int32_t*)bufs[0], (const int32_t*)bufs[r], (long)count);
        else hipLaunchKernelGGL(emba::emba_add_u8_kernel, dim3(grid), dim3(256), 0, s0, (uint8_t*)bufs[0], (const uint8_t*)bufs[r], (long)count);
    }
    G_HIP(g, hipGetLastError());
    G_HIP(g, hipEventRecord(ev0, s0));
    for (int r = 1; r < g->n; ++r) {
        G_HIP(g, hipStreamWaitEvent(S(r), ev0, 0));
        G_HIP(g, hipMemcpyAsync(bufs[r], bufs[0], count * xsize(t), hipMemcpyDeviceToDevice, S(r)));
    }
    // rank 0 must not overwrite its buffer before the copies have read it
    for (int r = 1; r < g->n; ++r) { G_HIP(g, hipEventRecord(ev[r], S(r))); G_HIP(g, hipStreamWaitEvent(s0, ev[r], 0)); }
    return EMBA_OK;
}

// all-to-all of doubles: rank src sends cnt[src][dst] elements (from send[src], destination-major) to rank dst, which stores them
// source-major in recv[dst]
emba_status group_alltoall(emba_group* g, double* const* send, double* const* recv, const std::vector<std::vector<size_t>>& cnt)
{
    const int n = g->n;
    std::vector<std::vector<size_t>> soff(n, std::vector<size_t>(n, 0)), roff(n, std::vector<size_t>(n, 0));
    for (int s = 0; s < n; ++s) { size_t run = 0; for (int d = 0; d < n; ++d) { soff[s][d] = run; run += cnt[s][d]; } }
    for (int d = 0; d < n; ++d) { size_t run = 0; for (int s = 0; s < n; ++s) { roff[d][s] = run; run += cnt[s][d]; } }
    if (g->use_rccl) {
        G_NCCL(g, g_rccl.GroupStart());
        for (int r = 0; r < n; ++r) {
            G_HIP(g, hipSetDevice(g->dev[r]));
            for (int q = 0; q < n; ++q) {
                if (cnt[r][q]) G_NCCL(g, g_rccl.Send(send[r] + soff[r][q], cnt[r][q], Rccl::kFloat64, q, g->comm[r], g->ctx[r]->stream));
                if (cnt[q][r]) G_NCCL(g, g_rccl.Recv(recv[r] + roff[r][q], cnt[q][r], Rccl::kFloat64, q, g->comm[r], g->ctx[r]->stream));
            }
        }
        G_NCCL(g, g_rccl.GroupEnd());
        return EMBA_OK;
    }
    for (int s = 0; s < n; ++s) G_HIP(g, hipEventRecord(g->ev[s], g->ctx[s]->stream));
    for (int d = 0; d < n; ++d)
        for (int s = 0; s < n; ++s) {
            if (!cnt[s][d]) continue;
            if (s != d) G_HIP(g, hipStreamWaitEvent(g->ctx[d]->stream, g->ev[s], 0));
            G_HIP(g, hipMemcpyAsync(recv[d] + roff[d][s], send[s] + soff[s][d], cnt[s][d] * 8, hipMemcpyDeviceToDevice, g->ctx[d]->stream));
        }
    // a sender may reuse its buffer only after every receiver has copied from it
    for (int d = 0; d < n; ++d) G_HIP(g, hipEventRecord(g->ev[d], g->ctx[d]->stream));
    for (int s = 0; s < n; ++s) for (int d = 0; d < n; ++d) if (s != d) G_HIP(g, hipStreamWaitEvent(g->ctx[s]->stream, g->ev[d], 0));
    return EMBA_OK;
}

emba_status group_ensure_buffers(emba_group* g, int K)
{
    const size_t need = (size_t)9 * K * K + (size_t)3 * K + 5 * g->npix;
    if (g->pack_cap >= need && !g->pack.empty()) return EMBA_OK;
    for (int r = 0; r < g->n; ++r) {
        G_HIP(g, hipSetDevice(g->dev[r]));
        G_HIP(g, hipStreamSynchronize(g->ctx[r]->stream));
        if (g->pack[r]) (void)hipFree(g->pack[r]);
        g->pack[r] = nullptr;
        G_HIP(g, hipMalloc((void**)&g->pack[r], need * sizeof(double)));
        if (!g->count[r]) {
            G_HIP(g, hipMalloc((void**)&g->count[r], g->npix * sizeof(int32_t)));
            G_HIP(g, hipMalloc((void**)&g->count_u8[r], g->npix));
            G_HIP(g, hipMemset(g->count[r], 0, g->npix * sizeof(int32_t)));
            G_HIP(g, hipDeviceSynchronize());      // (default stream: not ordered in front of the rank's non-blocking stream)
        }
        G_TRY(g, r, emba_bind_exchange_buffers(g->ctx[r], g->count[r], g->pack[r], need));
    }
    g->pack_cap = need; g->pack_K = K;
    return EMBA_OK;
}

}  // namespace

extern "C" {

emba_status emba_group_create(const emba_cfg* cfg, const int32_t* devices, int32_t n_ranks, emba_group** out)
{
    return emba_group_create_flags(cfg, devices, n_ranks, 0, out);
}

emba_status emba_group_create_flags(const emba_cfg* cfg, const int32_t* devices, int32_t n_ranks, uint32_t flags, emba_group** out)
{
    if (!out) return EMBA_ERR_INVALID_ARG;
    *out = nullptr;
    if (!cfg || !devices || n_ranks < 1 || n_ranks > 64) return fail(nullptr, EMBA_ERR_INVALID_ARG, "emba_group_create: bad arguments");
    emba_group* g = new emba_group();
    g->n = n_ranks;
    g->npix = (size_t)cfg->pano_w * cfg->pano_h; g->sw = cfg->sensor_w;
    g->ctx.assign(n_ranks, nullptr); g->dev.assign(devices, devices + n_ranks); g->comm.assign(n_ranks, nullptr); g->ev.assign(n_ranks, nullptr);
    g->count.assign(n_ranks, nullptr); g->count_u8.assign(n_ranks, nullptr); g->pack.assign(n_ranks, nullptr); g->n_local.assign(n_ranks, 0);
    g->side.assign(n_ranks, nullptr); g->ev_side.assign(n_ranks, nullptr); g->ev_side0.assign(n_ranks + 1, nullptr);
    g->lo.assign(n_ranks, 0); g->sw_ = cfg->sensor_w;
    g->sv_send.assign(n_ranks, nullptr); g->sv_recv.assign(n_ranks, nullptr); g->sv_S.assign(n_ranks, nullptr); g->sv_x2.assign(n_ranks, nullptr);
    g->cap_send.assign(n_ranks, 0); g->cap_recv.assign(n_ranks, 0); g->cap_S.assign(n_ranks, 0); g->cap_x2.assign(n_ranks, 0);
    auto bail = [&](emba_status st, const std::string& msg) { fail(nullptr, st, "%s", msg.c_str()); emba_group_destroy(g); return st; };
    for (int r = 0; r < n_ranks; ++r) {
        emba_cfg c2 = *cfg;
        c2.device = devices[r]; c2.stream = nullptr;      // every rank gets a stream of its own
        const emba_status st = emba_create(&c2, &g->ctx[r]);
        if (st) return bail(st, std::string("rank ") + std::to_string(r) + ": " + emba_last_error(nullptr));
        if (hipSetDevice(devices[r]) != hipSuccess || hipEventCreateWithFlags(&g->ev[r], hipEventDisableTiming) != hipSuccess) return bail(EMBA_ERR_HIP, "hipEventCreate failed");
        if (hipStreamCreateWithFlags(&g->side[r], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&g->ev_side[r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_side0[r], hipEventDisableTiming) != hipSuccess)
            return bail(EMBA_ERR_HIP, "side stream / event creation failed");
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&g->ev0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_side0[n_ranks], hipEventDisableTiming) != hipSuccess)
        return bail(EMBA_ERR_HIP, "hipEventCreate failed");
    const std::set<int> distinct(g->dev.begin(), g->dev.end());
    // EMBA_GROUP_FORCE_RCCL: a one-rank group goes through RCCL as well (what a one-GPU box can rehearse of the RCCL path: the run-time
    // binding, communicator set-up and every collective call, with world size 1)
    const bool force = n_ranks == 1 && (flags & EMBA_GROUP_FORCE_RCCL);
    if ((n_ranks > 1 && (int)distinct.size() == n_ranks) || force) {
        std::string e;
        if (!g_rccl.load(&e)) return bail(EMBA_ERR_HIP, e);
        const int rc = g_rccl.CommInitAll(g->comm.data(), n_ranks, g->dev.data());
        if (rc != 0) return bail(EMBA_ERR_HIP, std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(rc));
        g->use_rccl = true;
    } else if (n_ranks > 1 && distinct.size() != 1) {
        return bail(EMBA_ERR_INVALID_ARG, "a group's ranks must sit on distinct devices (RCCL) or all on one device (in-library exchange)");
    }
    g->pool.n = n_ranks;
    // EMBA_GROUP_NO_THREADS: the caller's thread drives every rank itself (debugging)
    if (n_ranks > 1 && !(flags & EMBA_GROUP_NO_THREADS)) g->pool.start(n_ranks, g->dev);
    *out = g;
    return EMBA_OK;
}

void emba_group_destroy(emba_group* g)
{
    if (!g) return;
    g->pool.shutdown();
    for (int r = 0; r < g->n; ++r) {
        if (g->ctx[r]) { (void)hipSetDevice(g->dev[r]); (void)hipStreamSynchronize(g->ctx[r]->stream); }
        for (double* p : {g->sv_send[r], g->sv_recv[r], g->sv_S[r], g->sv_x2[r]}) if (p) (void)hipFree(p);
    }
    for (int r = 0; r < g->n; ++r) {
        if (g->ctx[r]) { (void)hipSetDevice(g->dev[r]); (void)hipStreamSynchronize(g->ctx[r]->stream); }
        if (g->use_rccl && g->comm[r]) (void)g_rccl.CommDestroy(g->comm[r]);
        if (g->count[r]) (void)hipFree(g->count[r]);
        if (g->count_u8[r]) (void)hipFree(g->count_u8[r]);
        if (g->pack[r]) (void)hipFree(g->pack[r]);
        if (g->ev[r]) (void)hipEventDestroy(g->ev[r]);
        if (g->side[r]) { (void)hipStreamSynchronize(g->side[r]); (void)hipStreamDestroy(g->side[r]); }
        if (g->ev_side[r]) (void)hipEventDestroy(g->ev_side[r]);
        if (g->ev_side0[r]) (void)hipEventDestroy(g->ev_side0[r]);
        emba_destroy(g->ctx[r]);
    }
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (!g->ev_side0.empty() && g->ev_side0[g->n]) (void)hipEventDestroy(g->ev_side0[g->n]);
    delete g;
}

const char* emba_group_last_error(const emba_group* g) { return g ? g->err.c_str() : emba_last_error(nullptr); }
int32_t emba_group_size(const emba_group* g) { return g ? g->n : 0; }
int32_t emba_group_uses_rccl(const emba_group* g) { return (g && g->use_rccl) ? 1 : 0; }
emba_ctx* emba_group_ctx(emba_group* g, int32_t rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }

// Events sorted by time, the reference's EventPacket.  Rank r gets the whole global batches [nb r / N, nb (r+1) / N) and, per sensor
// pixel, the last event before its range with the midpoint time of the (global) batch that event belongs to (emba_set_events).
emba_status emba_group_set_events(emba_group* g, const uint16_t* x, const uint16_t* y, const uint8_t* pol, const int64_t* t_ns, size_t n)
{
    if (!g || (n && (!x || !y || !pol || !t_ns))) return g ? gfail(g, EMBA_ERR_INVALID_ARG, "event arrays are NULL") : EMBA_ERR_INVALID_ARG;
    const size_t nb = n / 100;
    const size_t S = (size_t)g->ctx[0]->sw * g->ctx[0]->sh;
    for (size_t k = 0; k < nb * 100; ++k)
        if (x[k] >= g->ctx[0]->sw || y[k] >= g->ctx[0]->sh) return gfail(g, EMBA_ERR_INVALID_ARG, "event %zu lies outside the sensor", k);
    std::vector<int64_t> last(S, -1);
    size_t k = 0, b = 0;
    for (int r = 0; r < g->n; ++r) {
        const size_t cnt = nb / g->n + ((size_t)r < nb % g->n ? 1 : 0);
        const size_t lo = b * 100, hi = (b + cnt) * 100;
        for (; k < lo; ++k) last[(size_t)y[k] * g->sw + x[k]] = (int64_t)k;      // events before this rank's range
        std::vector<uint16_t> hx, hy; std::vector<int64_t> hbt;
        if (lo) {
            std::vector<int64_t> idx;
            for (size_t p = 0; p < S; ++p) if (last[p] >= 0) idx.push_back(last[p]);
            std::sort(idx.begin(), idx.end());                                      // time order
            for (int64_t i : idx) {
                const size_t bb = (size_t)i / 100;
                hx.push_back(x[i]); hy.push_back(y[i]); hbt.push_back(batch_mid_ns(t_ns[100 * bb], t_ns[100 * bb + 99]));
            }
        }
        // the last rank also receives the n % 100 tail the reference drops (quirk Q1): emba_set_events ignores it the same way
        const size_t n_r = (r == g->n - 1) ? n - lo : hi - lo;
        G_TRY(g, r, emba_set_events(g->ctx[r], x + lo, y + lo, pol + lo, t_ns + lo, n_r, hx.data(), hy.data(), hbt.data(), hx.size()));
        g->n_local[r] = hi - lo; g->lo[r] = lo;
        b += cnt;
    }
    g->x1_done = false;
    return EMBA_OK;
}

emba_status emba_group_upload_map(emba_group* g, const double* Gx, const double* Gy)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_upload_map(g->ctx[r], Gx, Gy));
    return EMBA_OK;
}

// Declare the robust cost of the formNormalEq[IRLS] calls that follow the next evaluations (emba_set_cost on every rank): the per-pixel
// sums then carry its weights and the A22 | b2 rows of exchange 2 are final as soon as the active set has been written.  Speed only.
emba_status emba_group_set_cost(emba_group* g, int32_t irls, double eta)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_set_cost(g->ctx[r], irls, eta));
    g->decl_irls = irls; g->decl_eta = irls ? eta : 0.0;
    return EMBA_OK;
}

// exchange 1 on the count maps of the last evaluation: exact int32 sums, or saturated bytes when the activity threshold allows
emba_status group_exchange_counts(emba_group* g, int thres, bool exact, const std::function<emba_status(int)>* after = nullptr)
{   // after: per-rank work that follows the exchange directly (rides in the same fork-join as the expansion)
    if (g->x1_done) { if (after) return gpool(g, *after); return EMBA_OK; }
    const int cap = 255 / g->n;
    if (!exact && thres >= 1 && thres <= cap) {   // sum_i min(c_i, cap) >= thres <=> sum_i c_i >= thres, and world * cap <= 255 cannot wrap
        { emba_status st = gpool(g, [&](int r) { return emba_count_compress(g->ctx[r], g->count_u8[r], cap); }); if (st) return st; }
        { emba_status st = group_allreduce(g, (void* const*)g->count_u8.data(), g->npix, XType::U8); if (st) return st; }
        { emba_status st = gpool(g, [&](int r) { emba_status s1 = emba_count_expand(g->ctx[r], g->count_u8[r]); return (s1 || !after) ? s1 : (*after)(r); }); if (st) return st; }
        g->x1_done = true;
        return EMBA_OK;
    } else {
        { emba_status st = gpool(g, [&](int r) { return emba_count_map_ready(g->ctx[r]); }); if (st) return st; }
        { emba_status st = group_allreduce(g, (void* const*)g->count.data(), g->npix, XType::I32); if (st) return st; }
    }
    g->x1_done = true;
    if (after) return gpool(g, *after);
    return EMBA_OK;
}

// LEGM::evaluateDataError (model.cpp:72-258) over all ranks: E1 on every rank's shard (+ X1 when the caller wants num_ev_map).
// Gx / Gy: host planes to upload first, or both NULL for the resident (current or trial) map.  Outputs (any may be NULL):
// ep_out (capacity >= events used) = the residuals of ALL ranks merged into the reference's order (sensor pixel major, then time:
// ranks are time-ordered, so inside a pixel rank r's measurements precede rank r+1's); *n_inliers their number; num_ev_map_out the
// GLOBAL count map (exact int32 exchange).
emba_status emba_group_get_ep(emba_group* g, double* ep_out, size_t cap, size_t* n_inliers);
emba_status emba_group_eval(emba_group* g, const double* knots, int32_t K, int64_t t0_ns, int64_t dt_ns, const double* Gx, const double* Gy,
                            double* ep_out, size_t* n_inliers, int32_t* num_ev_map_out)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    if ((Gx == nullptr) != (Gy == nullptr)) return gfail(g, EMBA_ERR_INVALID_ARG, "pass both Gx and Gy, or neither");
    g->K = K;
    const bool single = (g->n == 1 && !g->use_rccl);
    if (Gx) { emba_status st = emba_group_upload_map(g, Gx, Gy); if (st) return st; }
    if (single) {
        G_TRY(g, 0, emba_eval_launch(g->ctx[0], knots, K, t0_ns, dt_ns));
        G_TRY(g, 0, emba_eval_finish(g->ctx[0], ep_out, (ep_out || n_inliers || num_ev_map_out) ? &g->n_inliers : nullptr, num_ev_map_out));
        if (n_inliers) *n_inliers = g->n_inliers;
        return EMBA_OK;
    }
    { emba_status st = group_ensure_buffers(g, K); if (st) return st; }
    { emba_status st = gpool(g, [&](int r) { return emba_eval_launch(g->ctx[r], knots, K, t0_ns, dt_ns); }); if (st) return st; }    // E1
    g->x1_done = false;
    if (num_ev_map_out) {
        { emba_status st = group_exchange_counts(g, 0, /*exact=*/true); if (st) return st; }                                              // X1
        G_HIP(g, hipSetDevice(g->dev[0]));
        G_HIP(g, hipStreamSynchronize(g->ctx[0]->stream));
        G_HIP(g, hipMemcpy(num_ev_map_out, g->count[0], g->npix * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    if (!ep_out && !n_inliers)      // nothing asked for: E2 is enqueued only (the costs / formNormalEq that follow find an evaluation to work on)
        for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_eval_finish(g->ctx[r], nullptr, nullptr, nullptr));
    if (ep_out || n_inliers) {
        size_t total = 0;
        for (int r = 0; r < g->n; ++r) { size_t m = 0; G_TRY(g, r, emba_eval_finish(g->ctx[r], nullptr, &m, nullptr)); total += m; }
        g->n_inliers = total;
        if (n_inliers) *n_inliers = total;
        if (ep_out) return emba_group_get_ep(g, ep_out, total, nullptr);
    }
    return EMBA_OK;
}

// The residual vector of the last emba_group_eval in the reference's order (model.cpp:221,256: sensor pixel major, then time), into memory the caller owns — apart
// from the evaluation (round 6), so that a host that must RETURN it by value (the EMBA::LEGM adapter: VecXd evaluateDataError) can size its vector from the inlier
// count first and have the residuals land in it directly.  Several ranks: each rank's vector + the sensor pixel of each entry, merged by (pixel, rank).
emba_status emba_group_get_ep(emba_group* g, double* ep_out, size_t cap, size_t* n_inliers)
{
    if (!g || (!ep_out && cap)) return EMBA_ERR_INVALID_ARG;
    if (g->n == 1 && !g->use_rccl) {
        size_t m = 0;
        G_TRY(g, 0, emba_get_ep(g->ctx[0], ep_out, cap, &m));
        g->n_inliers = m;
        if (n_inliers) *n_inliers = m;
        return EMBA_OK;
    }
    // (round 6) Every rank's residuals are in (sensor pixel, time) order, so a pixel's are one contiguous run per rank: S + 1 start offsets per rank say where, and
    // every rank streams its vector through its pinned buffers and copies its runs to their places in ep_out from its own thread.  Until then: a copy of every
    // rank's vector AND of one pixel word per residual to the host, then an element-by-element scatter — 47-56 ms per evaluateDataError on two ranks at config 2's
    // shape (7.5 M residuals), the largest item of the multi-GPU drop-in's iteration.
    const size_t S = (size_t)g->ctx[0]->sw * g->ctx[0]->sh;
    static const bool trace = std::getenv("EMBA_GROUP_TRACE") != nullptr;      // stage times on stderr
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<std::vector<uint32_t>> starts(g->n, std::vector<uint32_t>(S + 1));
    { emba_status st = gpool(g, [&](int r) { return emba_get_inlier_pixel_starts(g->ctx[r], starts[r].data()); }); if (st) return st; }
    const double t_starts = ms();
    size_t total = 0;
    for (int r = 0; r < g->n; ++r) total += starts[r][S];
    g->n_inliers = total;
    if (n_inliers) *n_inliers = total;
    if (cap < total) return gfail(g, EMBA_ERR_CAPACITY, "cap=%zu < inliers=%zu", cap, total);
    std::vector<std::vector<uint64_t>> dst(g->n, std::vector<uint64_t>(S));
    for (size_t p = 0; p < S; ++p) {
        uint64_t at = 0;
        for (int r = 0; r < g->n; ++r) at += starts[r][p];              // everything of the pixels in front
        for (int r = 0; r < g->n; ++r) { dst[r][p] = at; at += (uint64_t)starts[r][p + 1] - starts[r][p]; }   // rank-major inside a pixel: earlier ranks first
    }
    // The ranks' runs alternate inside every page of ep_out: left to the placement, two threads trap on the same fresh pages at the same time (measured: 16-17 ms for
    // 60 MB on two ranks, slower than ONE thread's 8.8).  Every rank first has its own n-th of the vector's pages mapped, then places.
    { emba_status st = gpool(g, [&](int r) { const size_t per = (total * sizeof(double) / g->n + 4095) & ~(size_t)4095, lo = std::min(total * sizeof(double), per * r),
                                                  hi = (r == g->n - 1) ? total * sizeof(double) : std::min(total * sizeof(double), per * (r + 1));
                                             if (hi > lo) populate_pages((char*)ep_out + lo, hi - lo);
                                             return EMBA_OK; });
      if (st) return st; }
    const double t_dst = ms();
    const emba_status st = gpool(g, [&](int r) { return starts[r][S] ? emba_get_ep_by_pixel(g->ctx[r], ep_out, dst[r].data()) : EMBA_OK; });
    if (trace) fprintf(stderr, "[group ep] %d ranks, %zu residuals: pixel starts %.2f ms, offsets + pages %.2f, placement %.2f\n", g->n, total, t_starts, t_dst - t_starts, ms() - t_dst);
    return st;
}

// LEGM::formNormalEq[IRLS] + applyL2Reg (model.cpp:316-719) over all ranks on the state of the last emba_group_eval:
// X1 (unless the evaluation already exchanged the counts) | E2, F1 | F2 | X2 | F3.
emba_status emba_group_form(emba_group* g, int32_t thres, int32_t irls, double eta, double alpha, size_t* n_inliers, size_t* P)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    g->x2_on_ranks = false;     // (what a solve of the previous equations left on the ranks)
    if (g->n == 1 && !g->use_rccl) {
        emba_ctx* c = g->ctx[0];
        G_TRY(g, 0, emba_eval_finish(c, nullptr, nullptr, nullptr));
        const bool fuse = (irls == c->acc_irls) && (irls == 0 || eta == c->acc_eta);
        if (fuse) c->fused_alpha = alpha;   // A22 / b2 come from the accumulator lines: applyL2Reg rides along with the gather (as in emba_step)
        G_TRY(g, 0, emba_form_active(c, thres, nullptr, nullptr));
        G_TRY(g, 0, emba_form_accumulate(c, nullptr, irls, eta));
        G_TRY(g, 0, emba_form_finish(c, alpha, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr));
        G_TRY(g, 0, emba_last_counts(c, &g->n_inliers, &g->P));
        if (n_inliers) *n_inliers = g->n_inliers;
        if (P) *P = g->P;
        return EMBA_OK;
    }
    // Round 5: a rank's step = the one-GPU step.  Where exchange 1 travels as saturated bytes, the per-pixel sums carry this cost's weights and the
    // shards are small enough for the list-driven gather (below the size from which exchange 2 is split: the two do not combine — the rows are written inside the
    // Gram launch), every rank runs emba_step_form_active on the exchanged bytes: launch A with lists + zeroing, gather inside the Gram kernel, no clearing pass
    // in the next evaluation, no expansion of the bytes into the int32 map.  The sequence of collectives is the same as below: all-reduce(u8), all-reduce(pack).
    {
        size_t n_max = 0;
        for (int r = 0; r < g->n; ++r) n_max = std::max(n_max, g->n_local[r]);
        bool fast = !g->x1_done && thres >= 1 && thres <= 255 / g->n && n_max < 3000000 && g->opt_x2_split <= 0 && g->opt_step_fast != 0;
        for (int r = 0; r < g->n; ++r) fast = fast && (irls == g->ctx[r]->acc_irls) && (irls == 0 || eta == g->ctx[r]->acc_eta) && g->ctx[r]->eval_launched;
        if (fast) {
            const int cap = 255 / g->n;
            { emba_status st = gpool(g, [&](int r) { return emba_count_compress(g->ctx[r], g->count_u8[r], cap); }); if (st) return st; }
            { emba_status st = group_allreduce(g, (void* const*)g->count_u8.data(), g->npix, XType::U8); if (st) return st; }                      // X1
            { emba_status st = gpool(g, [&](int r) {                                                                                              // E2, F1, F2
                  emba_status s1 = emba_step_form_active(g->ctx[r], thres, g->count_u8[r]);
                  return s1 ? s1 : emba_form_accumulate(g->ctx[r], nullptr, irls, eta); });
              if (st) return st; }
            size_t ni0 = 0;
            G_TRY(g, 0, emba_last_counts(g->ctx[0], &ni0, &g->P));          // the gather's first block publishes P: polled, the Gram kernels still run
            const size_t pl = g->ctx[0]->pack_len;
            { emba_status st = group_allreduce(g, (void* const*)g->pack.data(), pl, XType::F64); if (st) return st; }                             // X2
            std::vector<size_t> ni(g->n, 0), pp(g->n, 0);
            { emba_status st = gpool(g, [&](int r) {                                                                                              // F3
                  emba_status s1 = emba_form_finish(g->ctx[r], alpha, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
                  return s1 ? s1 : emba_last_counts(g->ctx[r], &ni[r], &pp[r]); });
              if (st) return st; }
            g->n_inliers = 0;
            for (int r = 0; r < g->n; ++r) {
                g->n_inliers += ni[r];
                if (pp[r] != g->P) return gfail(g, EMBA_ERR_STATE, "ranks disagree on the active set (%zu vs %zu pixels)", pp[r], g->P);
            }
            if (n_inliers) *n_inliers = g->n_inliers;
            if (P) *P = g->P;
            return EMBA_OK;
        }
    }
    {   // X1, then E2 and F1 (enqueue only) in the fork-join that expands the exchanged counts
        const std::function<emba_status(int)> e2f1 = [&](int r) {
            emba_status s1 = emba_eval_finish(g->ctx[r], nullptr, nullptr, nullptr);
            return s1 ? s1 : emba_form_active(g->ctx[r], thres, nullptr, nullptr); };
        emba_status st = group_exchange_counts(g, thres, /*exact=*/false, &e2f1);
        if (st) return st;
    }
    g->x1_done = false;   // (consumed: the next evaluation starts a new count map)
    // the active set comes from the GLOBAL counts: identical on every rank, so rank 0's P sizes exchange 2 (one host wait instead of N)
    size_t pl = 0;
    { size_t ni = 0; G_TRY(g, 0, emba_last_counts(g->ctx[0], &ni, &g->P)); pl = g->ctx[0]->pack_len; }
    // X2 in two parts.  The A22 | b2 rows (5 doubles per active pixel: the bulk of the exchange) are final once the active set has been
    // written — when the cost was declared before the evaluation, form_accumulate only adds the A11 | b1 head — and their all-reduce runs
    // on the ranks' SIDE streams while the Gram kernels form the head on the ranks' own streams; the small head follows.
    // (worth it once the Gram kernel is long enough to hide a collective behind — the head then costs one more collective's latency: from a
    // few million events per rank; option x2_split overrides)
    size_t n_max = 0;
    for (int r = 0; r < g->n; ++r) n_max = std::max(n_max, g->n_local[r]);
    // "final" is decided from the cost the LAST EVALUATION weighted its per-pixel sums with (acc_irls / acc_eta, recorded by emba_eval_launch),
    // not from the cost last declared: a cost declared after the evaluation (the IRLS form's first iteration) leaves gathered rows that
    // emba_form_accumulate is about to rebuild from the records — they must not be on their way through a collective meanwhile.
    bool rows_final = true;
    for (int r = 0; r < g->n; ++r) rows_final = rows_final && (irls == g->ctx[r]->acc_irls) && (irls == 0 || eta == g->ctx[r]->acc_eta);
    bool split = n_max >= 3000000;
    if (g->opt_x2_split >= 0) split = g->opt_x2_split != 0;      // emba_group_set_option("x2_split")
    split = split && rows_final;
    if (!split) {
        { emba_status st = gpool(g, [&](int r) { return emba_form_accumulate(g->ctx[r], nullptr, irls, eta); }); if (st) return st; }      // F2
        { emba_status st = group_allreduce(g, (void* const*)g->pack.data(), pl, XType::F64); if (st) return st; }                          // X2
    } else {
        const size_t head = pl - 5 * g->P;
        std::vector<void*> rows(g->n, nullptr);
        for (int r = 0; r < g->n; ++r) {
            rows[r] = g->pack[r] + head;
            G_HIP(g, hipSetDevice(g->dev[r]));
            G_HIP(g, hipEventRecord(g->ev_side[r], g->ctx[r]->stream));          // the active-set write of this rank
            G_HIP(g, hipStreamWaitEvent(g->side[r], g->ev_side[r], 0));
        }
        { emba_status st = group_allreduce(g, rows.data(), 5 * g->P, XType::F64, /*on_side=*/true); if (st) return st; }                    // X2b
        { emba_status st = gpool(g, [&](int r) { return emba_form_accumulate(g->ctx[r], nullptr, irls, eta); }); if (st) return st; }      // F2
        { emba_status st = group_allreduce(g, (void* const*)g->pack.data(), head, XType::F64); if (st) return st; }                        // X2a
        for (int r = 0; r < g->n; ++r) {
            G_HIP(g, hipSetDevice(g->dev[r]));
            G_HIP(g, hipEventRecord(g->ev_side[r], g->side[r]));
            G_HIP(g, hipStreamWaitEvent(g->ctx[r]->stream, g->ev_side[r], 0));  // F3 reads the reduced rows
        }
    }
    std::vector<size_t> ni(g->n, 0), pp(g->n, 0);
    { emba_status st = gpool(g, [&](int r) {                                                                                              // F3: applyL2Reg once, after the reduce
          emba_status s1 = emba_form_finish(g->ctx[r], alpha, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
          return s1 ? s1 : emba_last_counts(g->ctx[r], &ni[r], &pp[r]); });
      if (st) return st; }
    g->n_inliers = 0;
    for (int r = 0; r < g->n; ++r) {
        g->n_inliers += ni[r];
        if (pp[r] != g->P) return gfail(g, EMBA_ERR_STATE, "ranks disagree on the active set (%zu vs %zu pixels)", pp[r], g->P);
    }
    if (n_inliers) *n_inliers = g->n_inliers;
    if (P) *P = g->P;
    return EMBA_OK;
}

// One evaluateDataError + formNormalEq[IRLS] + applyL2Reg over all ranks (the map must be resident): E1 | X1 | E2, F1 | F2 | X2 | F3.
emba_status emba_group_step(emba_group* g, const double* knots, int32_t K, int64_t t0_ns, int64_t dt_ns, int32_t thres, int32_t irls, double eta,
                            double alpha, size_t* n_inliers, size_t* P)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    g->K = K;
    if (g->n == 1 && !g->use_rccl) {
        G_TRY(g, 0, emba_step(g->ctx[0], knots, K, t0_ns, dt_ns, thres, irls, eta, alpha, &g->n_inliers, &g->P));
        if (n_inliers) *n_inliers = g->n_inliers;
        if (P) *P = g->P;
        return EMBA_OK;
    }
    { emba_status st = emba_group_set_cost(g, irls, eta); if (st) return st; }      // the step's own cost: A22 | b2 final after F1
    { emba_status st = emba_group_eval(g, knots, K, t0_ns, dt_ns, nullptr, nullptr, nullptr, nullptr, nullptr); if (st) return st; }
    return emba_group_form(g, thres, irls, eta, alpha, n_inliers, P);
}

// LEGM::applyL2Reg (model.cpp:689-719) as a call of its own, for hosts that keep the reference's formNormalEq / applyL2Reg split
// (emba_group_form with alpha = 0 first): applied once per set of equations on every rank's replica of the reduced pack.
emba_status emba_group_apply_l2(emba_group* g, double alpha)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    return gpool(g, [&](int r) { return emba_form_finish(g->ctx[r], alpha, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr); });
}

// The reduced blocks (identical on every rank; read from rank 0): the out-arguments of formNormalEq + applyL2Reg.
emba_status emba_group_download(emba_group* g, double* A11, double* b1, uint32_t* active_idx, size_t cap_P, double* A22, double* b2)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    // (alpha = 0 here: the L2 term was applied by the step; form_finish applies it once per set of blocks anyway)
    G_TRY(g, 0, emba_form_finish(g->ctx[0], 0.0, A11, b1, active_idx, cap_P, A22, b2, nullptr));
    return EMBA_OK;
}

// 0.5 * ep.ep (or the robust cost) summed over the ranks' measurements + alpha/2 |G|^2 from the replicated map.
emba_status emba_group_costs(emba_group* g, int32_t irls, double eta, double alpha, double* data_cost, double* reg_cost)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    // every rank's reductions are enqueued before any is waited for (rank 0 also reduces the replicated map): N overlapping waits, not 2N serial ones
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_costs_launch(g->ctx[r], irls, eta, (r == 0 && reg_cost) ? 1 : 0));
    double d = 0;
    for (int r = 0; r < g->n; ++r) { double v = 0, rg = 0; G_TRY(g, r, emba_costs_finish(g->ctx[r], irls, eta, alpha, &v, &rg)); d += v; if (r == 0 && reg_cost) *reg_cost = rg; }
    if (data_cost) *data_cost = d;
    return EMBA_OK;
}

// Records to the ranks that own their pixels (count, pack, all-to-all) — unless every rank still holds, in pixel order, what it received for the CURRENT equations
// (round 6, VERDICT r5 #4: a re-solve with another lambda after a rejected trial, solver.cpp:340-352; two ranks on one device at config 2's shape: 20.5 ms per solve,
// most of it this exchange).  n_recv[r]: records rank r owns; g->last_solve_exchanged says which way it went.
}  // extern "C"
static emba_status group_exchange_records(emba_group* g, std::vector<size_t>* n_recv_out, const std::function<void(const char*)>& stage)
{
    const int n = g->n;
    std::vector<int32_t> have(n, 0);
    std::vector<size_t> n_recv(n, 0);
    for (int r = 0; r < n; ++r) G_TRY(g, r, emba_solve_shard_cached(g->ctx[r], r, n, &have[r], &n_recv[r]));
    bool all = true;
    for (int r = 0; r < n; ++r) all = all && have[r];
    g->last_solve_exchanged = !all;
    if (all) { *n_recv_out = n_recv; stage("records cached on their owners"); return EMBA_OK; }
    std::vector<std::vector<size_t>> cnt(n, std::vector<size_t>(n, 0));
    { emba_status st = gpool(g, [&](int r) { return emba_solve_shard_count(g->ctx[r], n, cnt[r].data()); }); if (st) return st; }
    stage("shard_count");
    std::vector<size_t> n_send(n, 0);
    std::fill(n_recv.begin(), n_recv.end(), 0);
    std::vector<std::vector<size_t>> cnt16(n, std::vector<size_t>(n, 0));
    for (int r = 0; r < n; ++r) for (int d = 0; d < n; ++d) { n_send[r] += cnt[r][d]; n_recv[d] += cnt[r][d]; cnt16[r][d] = 16 * cnt[r][d]; }
    // grow-only scratch (an LM loop solves every iteration: no hipMalloc / hipFree per call)
    for (int r = 0; r < n; ++r) {
        emba_status st;
        if ((st = grow(g, r, &g->sv_send[r], &g->cap_send[r], std::max<size_t>(n_send[r], 1) * 16)) || (st = grow(g, r, &g->sv_recv[r], &g->cap_recv[r], std::max<size_t>(n_recv[r], 1) * 16)))
            return st;
    }
    stage("grow");
    { emba_status st = gpool(g, [&](int r) { return emba_solve_shard_pack(g->ctx[r], n, g->sv_send[r]); }); if (st) return st; }
    stage("shard_pack");
    { emba_status st = group_alltoall(g, g->sv_send.data(), g->sv_recv.data(), cnt16); if (st) return st; }
    stage("alltoall (enqueue)");
    *n_recv_out = n_recv;
    return EMBA_OK;
}
extern "C" {

// LEGM::solveNormalEq (model.cpp:721-792) over the group: records to their pixel owners, partial Schur sums, all-reduce, replicated
// Cholesky, x2 exchanged.  x1_host: 3K, x2_host: 2P (either may be NULL).
emba_status emba_group_solve(emba_group* g, double lambda, int32_t fix_first_pose, double* x1_host, double* x2_host)
{
    if (g) g->x2_on_ranks = false;
    if (!g) return EMBA_ERR_INVALID_ARG;
    if (g->n == 1 && !g->use_rccl) { G_TRY(g, 0, emba_solve_normal_eq(g->ctx[0], lambda, fix_first_pose, x1_host, x2_host)); return EMBA_OK; }
    const int n = g->n;
    const auto t_dbg0 = std::chrono::steady_clock::now(); auto t_dbg = t_dbg0;
    auto stage = [&](const char* what) {      // option solve_debug on rank 0: host time per stage of the sharded solve (stderr)
        if (!g->ctx[0]->opt_solve_debug) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[group solve] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_dbg).count());
        t_dbg = now;
    };
    std::vector<size_t> n_recv(n, 0);
    { emba_status st = group_exchange_records(g, &n_recv, stage); if (st) return st; }
    const bool cached = !g->last_solve_exchanged;
    size_t s_doubles = 0;
    G_TRY(g, 0, emba_solve_shard_size(g->ctx[0], &s_doubles));
    for (int r = 0; r < n; ++r) {
        emba_status st;
        if ((st = grow(g, r, &g->sv_S[r], &g->cap_S[r], s_doubles)) || (st = grow(g, r, &g->sv_x2[r], &g->cap_x2[r], std::max<size_t>(2 * g->P, 2)))) return st;
    }
    stage("grow S, x2");
    { emba_status st = gpool(g, [&](int r) { return emba_solve_shard_partial(g->ctx[r], r, n, cached ? nullptr : g->sv_recv[r], n_recv[r], lambda, g->sv_S[r]); }); if (st) return st; }
    stage("shard_partial");
    { emba_status st = group_allreduce(g, (void* const*)g->sv_S.data(), s_doubles, XType::F64); if (st) return st; }
    stage("allreduce S (enqueue)");
    // a 2x2 block that is not positive definite shows up on its pixel's owner only: every rank finishes (x2 exchange included) and the
    // failure is reported once, for the group
    std::vector<emba_status> fin(n, EMBA_OK);
    (void)g->pool.run([&](int r) {
        fin[r] = emba_solve_shard_finish(g->ctx[r], r, n, cached ? nullptr : g->sv_recv[r], n_recv[r], lambda, fix_first_pose, g->sv_S[r], r == 0 ? x1_host : nullptr, g->sv_x2[r]);
        return EMBA_OK; });
    stage("shard_finish");
    for (int r = 0; r < n; ++r) if (fin[r] && fin[r] != EMBA_ERR_NUMERIC) return gfail(g, fin[r], "rank %d: %s", r, emba_last_error(g->ctx[r]));
    { emba_status st = group_allreduce(g, (void* const*)g->sv_x2.data(), 2 * g->P, XType::F64); if (st) return st; }
    G_HIP(g, hipSetDevice(g->dev[0]));
    if (x2_host && g->P) G_HIP(g, hipMemcpyAsync(x2_host, g->sv_x2[0], 2 * g->P * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    for (int r = 0; r < n; ++r) { G_HIP(g, hipSetDevice(g->dev[r])); G_HIP(g, hipStreamSynchronize(g->ctx[r]->stream)); }
    stage("x2 allreduce + sync");
    for (int r = 0; r < n; ++r) if (fin[r]) return gfail(g, fin[r], "rank %d: %s", r, emba_last_error(g->ctx[r]));
    g->x2_on_ranks = true;
    return EMBA_OK;
}

// LEGM::solveNormalEqCG (model.cpp:794-840) over the group.  One rank: the single-context solver.  Several (round 6): the pixels are sharded as in the Schur
// solve (same record exchange, cached across re-solves) and Eigen's loop (ConjugateGradient.h:28-88) runs here on scalars that every rank reads from the same
// all-reduced sums — emba_cg_shard_* (emba_hip.hip) are the per-rank steps, the collectives are one all-reduce of 3K + 2 doubles per matrix application and one
// of 2 doubles per iteration.
emba_status emba_group_solve_cg(emba_group* g, double lambda, int32_t fix_first_pose, int32_t max_iter, double tol, double* x1_host, double* x2_host,
                                int32_t* iterations, double* error)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    g->x2_on_ranks = false;
    if (g->n == 1 && !g->use_rccl) {
        G_TRY(g, 0, emba_solve_normal_eq_cg(g->ctx[0], lambda, fix_first_pose, max_iter, tol, x1_host, x2_host, iterations, error));
        return EMBA_OK;
    }
    const int n = g->n;
    if (max_iter <= 0) max_iter = 100;     // model.cpp:823-824
    if (!(tol > 0)) tol = 1e-6;
    std::vector<size_t> n_recv(n, 0);
    { emba_status st = group_exchange_records(g, &n_recv, [](const char*) {}); if (st) return st; }
    const bool cached = !g->last_solve_exchanged;
    size_t red_len = 0;
    G_TRY(g, 0, emba_cg_shard_size(g->ctx[0], &red_len));
    const size_t nn = red_len - 2;
    for (int r = 0; r < n; ++r) {
        emba_status st;      // (sv_S doubles as the reduce buffer: (3K+1)^2 doubles when the Schur solve has grown it, at least 3K + 2 here)
        if ((st = grow(g, r, &g->sv_S[r], &g->cap_S[r], red_len)) || (st = grow(g, r, &g->sv_x2[r], &g->cap_x2[r], std::max<size_t>(2 * g->P, 2)))) return st;
    }
    // the two scalars behind the pose part of every rank's reduce buffer, read from rank 0's copy (identical everywhere)
    auto read2 = [&](double* a, double* b) -> emba_status {
        double h[2] = {0, 0};
        G_HIP(g, hipSetDevice(g->dev[0]));
        G_HIP(g, hipMemcpyAsync(h, g->sv_S[0] + nn, 16, hipMemcpyDeviceToHost, g->ctx[0]->stream));
        G_HIP(g, hipStreamSynchronize(g->ctx[0]->stream));
        if (a) *a = h[0];
        if (b) *b = h[1];
        return EMBA_OK;
    };
    { emba_status st = gpool(g, [&](int r) { return emba_cg_shard_begin(g->ctx[r], r, n, cached ? nullptr : g->sv_recv[r], n_recv[r], lambda, fix_first_pose, g->sv_S[r]); }); if (st) return st; }
    { emba_status st = group_allreduce(g, (void* const*)g->sv_S.data(), red_len, XType::F64); if (st) return st; }
    double rhs2 = 0, absNew = 0;
    { emba_status st = read2(&rhs2, &absNew); if (st) return st; }
    int it = 0;
    double err = 0;
    if (rhs2 != 0) {
        const double thr = std::max(tol * tol * rhs2, std::numeric_limits<double>::min());
        double rn2 = rhs2;
        if (rn2 >= thr) {
            while (it < max_iter) {
                { emba_status st = gpool(g, [&](int r) { return emba_cg_shard_apply(g->ctx[r], g->sv_S[r]); }); if (st) return st; }
                { emba_status st = group_allreduce(g, (void* const*)g->sv_S.data(), red_len, XType::F64); if (st) return st; }
                std::vector<double> pt(n, 0.0);
                { emba_status st = gpool(g, [&](int r) { return emba_cg_shard_pt(g->ctx[r], g->sv_S[r], &pt[r]); }); if (st) return st; }
                const double alpha = absNew / pt[0];      // (pt[r] are equal: the same reduced values, the same one-block dot)
                { emba_status st = gpool(g, [&](int r) { return emba_cg_shard_update(g->ctx[r], alpha, g->sv_S[r]); }); if (st) return st; }
                std::vector<void*> tail(n);
                for (int r = 0; r < n; ++r) tail[r] = g->sv_S[r] + nn;
                { emba_status st = group_allreduce(g, tail.data(), 2, XType::F64); if (st) return st; }
                double absNext = 0;
                { emba_status st = read2(&rn2, &absNext); if (st) return st; }
                if (rn2 < thr) break;
                const double beta = absNext / absNew;
                absNew = absNext;
                { emba_status st = gpool(g, [&](int r) { return emba_cg_shard_direction(g->ctx[r], beta); }); if (st) return st; }
                ++it;
            }
        }
        err = std::sqrt(rn2 / rhs2);
    }
    { emba_status st = gpool(g, [&](int r) { return emba_cg_shard_end(g->ctx[r], r == 0 ? x1_host : nullptr, g->sv_x2[r]); }); if (st) return st; }
    { emba_status st = group_allreduce(g, (void* const*)g->sv_x2.data(), 2 * g->P, XType::F64); if (st) return st; }
    G_HIP(g, hipSetDevice(g->dev[0]));
    if (x2_host && g->P) G_HIP(g, hipMemcpyAsync(x2_host, g->sv_x2[0], 2 * g->P * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    for (int r = 0; r < n; ++r) { G_HIP(g, hipSetDevice(g->dev[r])); G_HIP(g, hipStreamSynchronize(g->ctx[r]->stream)); }
    g->x2_on_ranks = true;
    if (iterations) *iterations = it;
    if (error) *error = err;
    return EMBA_OK;
}

emba_status emba_group_last_solve_exchanged(const emba_group* g, int32_t* exchanged)
{
    if (!g || !exchanged) return EMBA_ERR_INVALID_ARG;
    *exchanged = g->last_solve_exchanged ? 1 : 0;
    return EMBA_OK;
}

// LEGM::updateMap (model.cpp:863-903) and the LM decision on every rank's replica of the map
emba_status emba_group_update_map(emba_group* g, const double* x2_host, double damping)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    // x2_host == NULL: every rank applies the x2 the last emba_group_solve left in ITS device memory (the all-reduced vector) — nothing crosses
    // to the host and back, and not once per rank
    if (!x2_host && (g->n > 1 || g->use_rccl)) {
        if (!g->x2_on_ranks) return gfail(g, EMBA_ERR_STATE, "x2 NULL: no emba_group_solve / emba_group_solve_cg has left an x2 on the ranks");
        for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_update_map_dev(g->ctx[r], g->sv_x2[r], damping));
        return EMBA_OK;
    }
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_update_map(g->ctx[r], x2_host, damping));
    return EMBA_OK;
}
emba_status emba_group_map_accept(emba_group* g)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_map_accept(g->ctx[r]));
    return EMBA_OK;
}
emba_status emba_group_map_reject(emba_group* g)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_map_reject(g->ctx[r]));
    return EMBA_OK;
}
emba_status emba_group_download_map(emba_group* g, double* Gx, double* Gy)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    G_TRY(g, 0, emba_download_map(g->ctx[0], Gx, Gy));
    return EMBA_OK;
}
emba_status emba_group_get_map_active(emba_group* g, double* gxy_host, size_t cap_P)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    G_TRY(g, 0, emba_get_map_active(g->ctx[0], gxy_host, cap_P));
    return EMBA_OK;
}
// the last evaluation was a rejected trial that involved no map update: the equations formed before it are current again on every rank
emba_status emba_group_trial_reject(emba_group* g)
{
    if (!g) return EMBA_ERR_INVALID_ARG;
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_trial_reject(g->ctx[r]));
    return EMBA_OK;
}

emba_status emba_group_set_option(emba_group* g, const char* name, int32_t value)
{
    if (!g || !name) return EMBA_ERR_INVALID_ARG;
    if (!strcmp(name, "x2_split")) {
        if (value < -1 || value > 1) return EMBA_ERR_INVALID_ARG;
        g->opt_x2_split = value;
        return EMBA_OK;
    }
    if (!strcmp(name, "group_step_fast")) {
        if (value < 0 || value > 1) return EMBA_ERR_INVALID_ARG;
        g->opt_step_fast = value;
        return EMBA_OK;
    }
    for (int r = 0; r < g->n; ++r) G_TRY(g, r, emba_set_option(g->ctx[r], name, value));
    return EMBA_OK;
}

}  // extern "C"
