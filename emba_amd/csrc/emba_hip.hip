// emba_amd/csrc/emba_hip.hip — context, HBM residency and the C ABI of include/emba_hip.h.
// Host code is C++17; every per-event / per-pixel computation runs in the HIP kernels of kernels.h.
// There is no CPU compute path in this file: the host only sorts indices once per window
// (emba_set_events: pose-independent structure), launches kernels and moves bytes.
#include "../../include/emba_hip.h"

#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "order_kernels.h"
#include "solve_kernels.h"
#include "poisson_kernels.h"

using namespace emba;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct SolveLists { uint32_t* off = nullptr; emba::RecView sorted{}; };   // per-pixel record lists of a solve — sorted: the participating records in pixel order; off[i]: first record of pixel i

struct emba_ctx {
    emba_cfg cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    int sw = 0, sh = 0, W = 0, H = 0;
    size_t S = 0, npix = 0;
    double fx = 0, fy = 0, cx = 0, cy = 0, C_th = 0, outlier_px = 10.0;
    double fov_x = M_PI, fov_y = M_PI;    // the sensor's field of view (rad), from the bearing LUT

    // persistent device buffers
    double* d_lut = nullptr;
    double* d_texel = nullptr;
    double* d_Gx_own = nullptr; double* d_Gy_own = nullptr;
    const double* d_Gx = nullptr; const double* d_Gy = nullptr;   // the map planes the next evaluation reads (current or trial)
    const double* d_Gx_cur = nullptr; const double* d_Gy_cur = nullptr;   // current (accepted) map: own upload, bound, or accepted trial
    double* d_Gx_trial = nullptr; double* d_Gy_trial = nullptr; bool map_is_trial = false;
    // the per-pixel record lists + pixel-ordered records of the last LOCAL solve (workspaces 0, 33): a re-solve of the same equations with another
    // lambda (a rejected LM trial, solver.cpp:340-352) reuses them
    bool lists_valid = false; uint32_t lists_stamp = 0; size_t lists_P = 0, lists_nrec = 0;
    // round 6: ... and so does the SHARDED solve — lists_packed: the workspaces hold the records this rank RECEIVED as the owner of the pixels
    // [lists_lo, lists_lo + lists_P), keyed by its own working set's stamp (the ranks evaluate and form in lockstep); a re-solve skips count, pack and the all-to-all
    bool lists_packed = false; long lists_lo = 0;
    struct CgShard { bool active = false; int rank = 0, n_ranks = 1, n = 0, skip = 0; size_t lo = 0, npix = 0, Nl = 0; double lambda = 0;
                     double *x = nullptr, *r = nullptr, *p = nullptr, *z = nullptr, *t = nullptr, *invd = nullptr, *sc = nullptr; SolveLists L; } cg;   // emba_cg_shard_*
    uint32_t count_stamp = 0;   // record stamp (set_stamp) of the evaluation whose materialised, LOCAL counts d_count_own holds; 0: none (build_lists)
    size_t perm_lo = 0, perm_n = 0;   // ... of the pixels [perm_lo, perm_lo + perm_n) of the active set (a rank's owned range in the sharded solve)
    bool perm_valid = false; uint32_t* d_perm = nullptr;   // column order of U for the local Schur solve (solve_perm), valid with the lists
    int solve_perm_mode = -1;                               // option solve_perm (A/B): -1 auto, 0 off, 1 on
    double* h_cost = nullptr;           // pinned: {data cost sum, reg cost sum, error word, sequence number} of emba_costs, written by the kernel itself
    double* h_cost_dev = nullptr; double* d_cost_acc = nullptr; int cost_seq = 0;      // its device pointer; device: {data, reg} partial sums + the blocks' ticket counter (zero between calls)
    double* d_x2 = nullptr; size_t x2_cap = 0; size_t x2_resident_P = (size_t)-1;   // x2_resident_P: d_x2 holds the x2 of the last solve on this context (for that many pixels)
    int32_t* d_count_own = nullptr; int32_t* d_count = nullptr;
    bool counts_raw = false;   // the count map holds the warp kernel's markers, not yet the counts (see ensure_counts)
    double* d_pixacc = nullptr; bool pix_dirty_all = true;   // per-pixel A22/b2 accumulator lines (64 B each)
    bool pixacc_clean = false;     // every accumulator line is zero: the last evaluation's sums were gathered AND cleared by the resident step (no prep blocks needed)
    bool pixacc_consumed = false;  // ... so a second formNormalEq on the same evaluation rebuilds A22 | b2 from the records instead
    bool step_consume = false;     // set by emba_step around its emba_form_active: this gather is the sums' only reader
    bool force_generic_a22 = false;
    int32_t count_mark = 0;        // count_marker(stamp) of the last evaluation: what its touched pixels hold in a raw count map
    int texel_mode = 0;   // 0 auto, 1 pack every texel, 2 always on-the-fly stencil, 3 texel rectangle (option texel)
    int use_texel = 0;    // what the current evaluation uses: 0 fly, 1 full pack, 3 rectangle
    int* d_rect = nullptr;       // {xmin,ymin,xmax,ymax} of the pixels the previous evaluation touched
    int* d_blk_rect = nullptr;   // per prep-block boxes
    int32_t* d_compact = nullptr; uint8_t* d_active_bits = nullptr;   // 1 bit per pixel: count >= thres of the current active set
    uint32_t* d_active = nullptr;
    uint32_t* d_ablk_cnt = nullptr; uint32_t* d_ablk_off = nullptr; size_t n_ablk = 0;
    double* d_pack_own = nullptr; size_t pack_own_cap = 0;
    double* d_pack = nullptr; size_t pack_cap = 0; bool pack_bound = false;
    double* d_knots = nullptr; int knots_cap = 0;
    int* d_err = nullptr;           // status word of the CURRENT evaluation: one of the two words of d_err2 (they alternate: an evaluation's first launch
    int* d_err2 = nullptr;          // clears the NEXT evaluation's word while its own pose threads may already be setting bits of the current one)
    uint32_t eval_seq = 0;
    uint32_t* d_total = nullptr;    // [0] inliers, [1] active pixels
    double* d_scalar = nullptr;     // cost reductions
    std::vector<uint32_t> h_pix_starts; uint32_t pix_starts_seq = 0;   // emba_get_inlier_pixel_starts of evaluation number pix_starts_seq (eval_seq), for emba_get_ep_by_pixel
    void* h_stage[2] = {nullptr, nullptr}; hipEvent_t stage_ev[2]{};   // two pinned 8-MB buffers: device -> PAGEABLE host memory in pipelined chunks (d2h_chunks)
    int* h_pinned = nullptr;        // pinned, device-visible status words the kernels write: [0] inliers [1] err [2] P [3] step sequence number
    int seq = 0;                    // sequence number of the last step whose post-warp kernels publish [3]
    bool seq_armed = false;         // the pending counts come from kernels that publish the sequence number
    bool spun = false;              // counts were taken by polling: later kernels of the stream may still be running
    int* h_pinned_dev = nullptr;    // the same memory through its device pointer
    double* h_knots_dev = nullptr;  // device pointer of the pinned control-pose staging buffer

    // per-window (set_events) state — all of it lives on the device (order_kernels.h)
    bool have_events = false, have_map = false;
    size_t n_in = 0, n_used = 0, n_halo = 0, n_pm = 0, n_sorted = 0, n_batch = 0, n_cand = 0;   // n_pm: pm-order entries (events + halo); n_sorted: entries of the device order
    long nblk = 0;
    uint32_t* d_pm_pix = nullptr; uint32_t* d_pm_batch = nullptr; uint32_t* d_pm_orig = nullptr;   // pm-order = (sensor pixel, time): the reference's per-pixel vectors laid end to end
    uint32_t* d_ev_pix = nullptr; uint32_t* d_ev_batch = nullptr; uint32_t* d_ev_slot = nullptr;    // device order (== pm-order arrays in pixel order; own arrays in tile order)
    uint32_t* d_ev_pix_own = nullptr; uint32_t* d_ev_batch_own = nullptr;                          // tile order: the arrays d_ev_pix / d_ev_batch point to
    uint32_t* d_ev_pm = nullptr;                                                                    // tile order only: entry -> pm index
    bool have_ev_pm = false;
    std::unordered_map<void**, size_t> caps;                                                        // capacities of the grow-only device buffers (dev_alloc)
    uint16_t* d_cp = nullptr; double* d_batch_u = nullptr;                                          // control-pose index (= spline segment) and spline parameter per batch
    double* d_ev_u = nullptr; uint16_t* d_ev_seg = nullptr;                                         // tile order: the same per entry of the device order
    ChunkDesc* d_chunks = nullptr; long n_chunks = 0;                                               // tile order: one per workgroup of the tiled warp kernel
    int segpose_mode = 0;      // option segpose (A/B; 0 auto = yes, 1 no, 2 yes): pixel order evaluates the pose per event from segment records (default: yes)
    bool segpose = false;      // ... in the current evaluation
    bool chunks_lpt = false;   // the chunk list is sorted longest first and walked in grid order (no XCD-contiguous remapping)
    double order_inl_pred = 0.0;   // ... and the inlier fraction the predicted pixels give (the rule's estimate of what the pixel order pays per event)
    double order_per_px = 0.0, order_lead_frac = 0.0;   // what the last order decision saw (diagnostics: emba_last_order_stats)
    bool tile_order = false; int order_mode = 0;   // option order: 0 auto, 1 pixel, 2 tile
    size_t n_lead = 0;                             // lead-in copies the tile order added
    int64_t* d_batch_t = nullptr; double* d_pose = nullptr;   // pose table: 112 B per batch (pixel order; the tile order only uses it to predict the bins)
    double* d_seg = nullptr; int seg_cap = 0;                 // tile order: per-segment constants the tiled kernel evaluates each event's pose from (12 doubles per segment)
    int step_gather = 2;    // option step_gather: how emba_step writes its active set + A22 | b2 rows — 0 the sweeping kernel (emba_active_write_kernel), 1 the list-driven
                            // gather as a kernel of its own, 2 (default) the list-driven gather as the head of the compact Gram kernel
    uint16_t* d_seg_act = nullptr;   // launch A's per-unit active lists (offsets inside the unit), n_ablk * kActivePix entries
    bool aw_in_gram = false; ActiveWriteParams aw_saved{};   // the gather of the running step, to be issued with its Gram launch (emba_form_accumulate)
    int step_one_set = 1;      // option step_one_set = 0: emba_step alternates between two record sets like an LM loop's evaluations (A/B)
    bool no_alt_set = false;   // set by emba_step around its evaluation: the step re-forms the equations itself, nothing of the previous ones can be gone back to —
                               // no second record set (+12.8 GB at 100 M events for callers that can never reject)
    int step_fast = 1;   // option step_fast = 0: emba_step keeps the clearing pass in front of every evaluation (A/B)
    double* d_tag = nullptr; int use_tags = 1;   // per-slot {pano pixel, stamp}: lets the Gram kernel skip dead slots without fetching them (option gram_tags = 0 disables)
    double* d_rec = nullptr; uint32_t* d_slot_key = nullptr; uint32_t rec_stamp = 0;   // evaluation number stamped into the records (record_valid)
    // Two record sets (a rejected LM trial must not cost a re-evaluation).  d_rec / d_tag / set_stamp are the WORKING set: what
    // the last evaluation wrote and what formNormalEq and the solvers read.  An evaluation that would overwrite records the current normal
    // equations were formed from (accum_done) first swaps in the other set; emba_map_reject / emba_trial_reject swaps back, so the solver can be
    // called again with a larger lambda on untouched equations (solver.cpp:340-352 reuses A, b).  Nothing else needs a second copy: the
    // pack, the active set and the compact index are written by formNormalEq only, which never runs on a rejected trial.
    double* d_rec2 = nullptr; double* d_tag2 = nullptr;
    uint32_t set_stamp = 0, set_stamp2 = 0;      // stamp of the evaluation that wrote each set
    bool eq_in_alt = false;                      // the equations in the pack belong to the OTHER set (a trial evaluation has been written since)
    struct EqState { bool active_done = false, accum_done = false, finish_done = false, compact_valid = false, l2_fused = false; size_t P = 0, pack_len = 0; int K = 0, thres = 0, irls = 0; double eta = 0; } eq_saved;
    double* d_e_sorted = nullptr; uint8_t* d_flag = nullptr; int32_t* d_inl_idx = nullptr;
    size_t n_outside_tile = 0; int n_rebin = 0; uint32_t last_rebin_stamp = 0;   // tile order: inliers found outside their tile in the last evaluation; how often the window was re-binned
    bool inl_idx_valid = false;   // the per-event inlier numbers are produced on demand (dumps, caller-supplied ep): 4 B/event the step does not write
    bool ep_valid = false;        // d_ep holds the current evaluation's residuals in the reference's order (the resident step's Gram launch compacts them in its tail blocks)
    bool ep_in_gram = false;      // ... the Gram launch of the equations being formed will do that (set by emba_form_active's fused branch)
    bool ep_after_gram = false;   // ... or, for windows too long for the tail form, the scan + compaction launches behind it
    int opt_poison = 0;          // option poison = 1 (tests): every NEW device allocation of the context is filled with 0xFF bytes
    int opt_tile_reserve = 2, opt_tile_shape = -1, opt_tile_fine = -1, opt_tile_min_events = 1650000, opt_tile_chunk = 0;   // emba_set_option: the tile order's window rule (prepare_order)
    int tile_shape = 0; bool tile_fine = false;   // ... and what the current order uses: index into kTileShapes, its finer pitch grid
    int opt_gather_waves = 0, opt_chunk_order_bin = 0, opt_solve_counts = -1, opt_syrk_dense = 0, opt_syrk_lists = 0, opt_gram_sparse = -1, opt_gram_sparse_chunk = 4, opt_syrk_min_cols = 512, opt_syrk_item_cap = 4096, opt_solve_debug = 0, opt_poisson = 0, opt_gemm64 = 0;   // emba_set_option
    int step_ep = 1;              // emba_step produces ep (what evaluateDataError returns, model.cpp:256) in every step; 0: on demand only (A/B, bench.py's no_ep block)
    bool step_wants_ep = false;   // set by emba_step around its emba_form_active
    const uint8_t* global_u8 = nullptr;   // set by emba_step_form_active around its emba_form_active: the all-reduced saturated byte counts activity is decided from
    uint32_t* d_fblk_cnt = nullptr; uint32_t* d_fblk_off = nullptr; long n_fblk = 0;   // inlier-flag counts per kFlagBlk pm-order entries
    uint32_t* d_fsup = nullptr; long n_fsup = 0; int fsup_half = 0;                    // ... summed per kFlagSup of those by launch A (two arrays used alternately; fsup_half: the one the last launch A filled)
    double* d_ep = nullptr;
    // order / key cache
    bool keys_ready = false; int64_t key_t0 = 0, key_dt = 0; int key_K = 0;
    double set_events_ms = 0, prepare_ms = 0;   // wall time of the last emba_set_events[_dev] / order preparation (diagnostics)

    // per-iteration state
    int K = 0;
    bool eval_launched = false, eval_done = false, active_done = false, accum_done = false;
    size_t n_inliers = 0, P = 0, pack_len = 0;
    size_t P_prev = 0;                                      // active pixels of the last equations whose count the host has seen (this window): the Gram kernel's form is chosen by it
    bool compact_valid = false;   // d_compact matches the current active set (built on demand)
    int cost_irls = 0; double cost_eta = 0.0;   // robust cost declared with emba_set_cost: what the NEXT evaluation weights its per-pixel sums with
    int acc_irls = 0; double acc_eta = 0.0;     // ... and what the per-pixel sums of the LAST evaluation were weighted with
    double fused_alpha = 0.0; bool l2_fused = false;   // emba_step on one GPU folds applyL2Reg into the active-set gather
    bool ep_deferred = false;   // residual compaction not launched yet (it rides along with the active-set kernels)
    bool inl_pending = false, P_pending = false;   // counters enqueued for readback but not yet resolved (no host sync yet)
    double* h_knots = nullptr; int h_knots_cap = 0; hipEvent_t knots_copied = nullptr; bool knots_in_flight = false;   // pinned staging for the control poses
    int thres = 0, irls = 0; double eta = 0;

    // timing
    hipEvent_t ev_start[8]{}, ev_stop[8]{};
    bool kernel_timing = false;
    hipEvent_t kt_sets[16][5]{}; hipEvent_t* kt = kt_sets[0]; int kt_slot = 0;  // per slot: warp start/stop, accum start/stop (emba_enable_kernel_timing), [4]: in front of the step's first launch
    bool kt_valid[16][3]{};
    bool kt_all = false;        // emba_kernel_timing_all: also record [4], so that the four intervals [4]->[0]->[1]->[2]->[3] tile the whole step (prep | warp | post-warp | Gram)
    hipEvent_t cal_ev[2]{};     // emba_bracket_overhead_us / emba_clock_probe
    unsigned long long* d_probe = nullptr;
    bool kt_warp_valid = false, kt_accum_valid = false;
    int n_cu = 256;  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    int ablate = 0;  // diagnostics builds only (-DEMBA_DIAG): EMBA_ABLATE bit mask; always 0 in the shipped library
    bool finish_done = false;   // emba_form_finish ran (L2 applied): the state emba_solve_normal_eq works on
    int solve_info = 0;         // last Schur solve: bit 0 a 2x2 block not positive definite (error), bit 1 a pivot of S vanished (zero update, like Eigen's ldlt)
    // grow-only workspaces of the Schur solve (an LM loop calls it every iteration)
    struct { void* p = nullptr; size_t bytes = 0; } ws[44];   // 0-15 and 32-39 Schur solve, 16-31 sort / order preparation, 42 pixel starts of ep, 43 unpacked A22 | b2 of a download
    // f3 (Poisson reconstruction): sine matrices and eigenvalues of the two transform lengths, scratch planes
    double *d_SH = nullptr, *d_SW = nullptr, *d_lamH = nullptr, *d_lamW = nullptr, *d_pF = nullptr, *d_pT = nullptr, *d_pGx = nullptr, *d_pGy = nullptr;
    double* d_thomas = nullptr;   // Thomas factors of T_W + lambda1[i] I (W x H)
    double* d_Sfold = nullptr;    // the sine matrix folded by its symmetry: two (H/2 x H/2) blocks (even H)
    double* d_pE = nullptr;       // [E | O] of the folded transform (W x H/2 each)
};

namespace {

emba_status fail(emba_ctx* c, emba_status st, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return st;
}

#define HIP_TRY(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), EMBA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

// Grow-only device buffers: a buffer is re-allocated only when it has to grow, so that registering the next window of a sliding-window
// run (same sizes) costs kernels, not hipMalloc / hipFree of gigabytes (measured at 100 M events: 270 ms of allocator time around
// 10 ms of kernels).  `fresh` tells the caller that the memory is new (uninitialised).
template <typename T>
emba_status dev_alloc(emba_ctx* c, T** p, size_t count, bool* fresh = nullptr)
{
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    void** key = reinterpret_cast<void**>(p);
    auto it = c->caps.find(key);
    if (*p && it != c->caps.end() && it->second >= bytes) { if (fresh) *fresh = false; return EMBA_OK; }
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    c->caps.erase(key);
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(p), bytes));
    if (c->opt_poison) HIP_TRY(c, hipMemsetAsync(*p, 0xFF, bytes, c->stream));
    c->caps[key] = bytes;
    if (fresh) *fresh = true;
    return EMBA_OK;
}

template <typename T>
void dev_free(emba_ctx* c, T*& p)
{
    if (p) (void)hipFree(p);
    if (c) c->caps.erase(reinterpret_cast<void**>(&p));
    p = nullptr;
}

void free_window(emba_ctx* c)
{   // (the buffers stay: the next window reuses them, see dev_alloc)
    c->d_ev_pix = nullptr; c->d_ev_batch = nullptr;
    c->have_events = false; c->keys_ready = false; c->tile_order = false; c->have_ev_pm = false; c->n_chunks = 0; c->n_lead = 0; c->P_prev = 0;
    c->eval_launched = c->eval_done = c->active_done = c->accum_done = false;
    // (counters of the previous window that were never resolved must not be read as this window's)
    c->inl_pending = c->P_pending = false; c->ep_deferred = false; c->ep_valid = false; c->inl_idx_valid = false; c->seq_armed = false;
}

void free_all_buffers(emba_ctx* c)
{
    std::vector<void**> keys;
    for (auto& kv : c->caps) keys.push_back(kv.first);
    for (void** k : keys) { if (*k) (void)hipFree(*k); *k = nullptr; }
    c->caps.clear();
}

// ros::Time/Duration midpoint of a batch (model.cpp:116-119; rostime semantics per SURVEY Appendix A):
// integer nanoseconds plus one double scale-and-round.  Pose-independent, so it is computed once per window.
int64_t batch_mid_ns(int64_t t_first, int64_t t_last)
{
    const int64_t d = t_last - t_first;
    int64_t dsec = d / 1000000000LL, dnsec = d % 1000000000LL;
    if (dnsec < 0) { dnsec += 1000000000LL; dsec -= 1; }
    const double half = ((double)dsec + 1e-9 * (double)dnsec) * 0.5;   // Duration::toSec() * 0.5
    int64_t hsec = (int64_t)std::floor(half);
    int64_t hnsec = (int64_t)std::round((half - (double)hsec) * 1e9);  // Duration::fromSec
    hsec += hnsec / 1000000000LL;
    hnsec = hnsec % 1000000000LL;
    return t_first + hsec * 1000000000LL + hnsec;
}

// ---- device-side helpers of the once-per-window structure (order_kernels.h) ----------------------------------------------------
emba_status ws_get(emba_ctx* c, int slot, size_t bytes, void** out)
{
    auto& w = c->ws[slot];
    if (w.bytes < bytes || !w.p) {
        if (w.p) (void)hipFree(w.p);
        w.p = nullptr; w.bytes = 0;
        if (hipMalloc(&w.p, std::max<size_t>(bytes, 8)) != hipSuccess) return fail(c, EMBA_ERR_HIP, "hipMalloc of %zu bytes failed (workspace %d)", bytes, slot);
        w.bytes = std::max<size_t>(bytes, 8);
        if (c->opt_poison) { (void)hipMemsetAsync(w.p, 0xFF, w.bytes, c->stream); }      // option poison (tests): new memory reads as NaN / 0xFFFFFFFF, so that a read of never-written workspace shows
    }
    *out = w.p;
    return EMBA_OK;
}

void ws_release(emba_ctx* c, int first, int last)
{
    for (int i = first; i <= last; ++i) { if (c->ws[i].p) (void)hipFree(c->ws[i].p); c->ws[i].p = nullptr; c->ws[i].bytes = 0; }
}

inline unsigned nblocks(size_t n, unsigned per = 256) { return (unsigned)std::max<size_t>((n + per - 1) / per, 1); }

// out[i] = sum_{j<i} in[j]; total_dev[0] = sum of all (may be nullptr).  Workspaces 16, 17.
emba_status dev_scan(emba_ctx* c, const uint32_t* in, uint32_t* out, size_t n, uint32_t* total_dev, int* total_host = nullptr, const int* err_dev = nullptr, int* err_host = nullptr)
{   // (total_host / err_host: pinned, device-visible words the middle launch writes the total and the evaluation's status word to — no copy node)
    hipStream_t s = c->stream;
    const size_t ntiles = (n + kScanTile - 1) / kScanTile;
    uint32_t *sums = nullptr, *offs = nullptr, *tot = nullptr;
    emba_status st;
    if ((st = ws_get(c, 16, (ntiles + 1) * 4, (void**)&sums)) || (st = ws_get(c, 17, (ntiles + 2) * 4, (void**)&offs))) return st;
    tot = total_dev ? total_dev : offs + ntiles + 1;
    if (!n) { HIP_TRY(c, hipMemsetAsync(tot, 0, 4, s)); return EMBA_OK; }
    hipLaunchKernelGGL(emba_scan_tile_sums_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, in, (long)n, sums);
    hipLaunchKernelGGL(emba_scan_kernel, dim3(1), dim3(256), 0, s, sums, offs, (long)ntiles, tot, total_host, err_dev, err_host);
    hipLaunchKernelGGL(emba_scan_apply_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, in, (long)n, offs, out);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

// Stable LSD radix sort of (keys, vals) on the low `bits` bits of the keys; the sorted arrays end up in (*keys, *vals), the other
// pair of buffers is scratch (pointers are swapped per pass).  Workspace 18 (+ 16, 17 through dev_scan).
emba_status dev_sort(emba_ctx* c, uint32_t** keys, uint32_t** vals, uint32_t** keys_alt, uint32_t** vals_alt, size_t n, int bits)
{
    if (n < 2 || bits <= 0) return EMBA_OK;
    hipStream_t s = c->stream;
    const size_t ntiles = (n + kSortTile - 1) / kSortTile;
    uint32_t* hist = nullptr;
    emba_status st;
    if ((st = ws_get(c, 18, 256 * ntiles * 4, (void**)&hist))) return st;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(emba_sort_hist_kernel, dim3(nblocks(ntiles, 4)), dim3(256), 0, s, *keys, (long)n, shift, (long)ntiles, hist);
        if ((st = dev_scan(c, hist, hist, 256 * ntiles, nullptr))) return st;
        hipLaunchKernelGGL(emba_sort_scatter_kernel, dim3(nblocks(ntiles, 4)), dim3(256), 0, s, *keys, *vals, (long)n, shift, (long)ntiles, hist, *keys_alt, *vals_alt);
        std::swap(*keys, *keys_alt); std::swap(*vals, *vals_alt);
    }
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

int bits_for(size_t n_values) { int b = 1; while (((size_t)1 << b) < n_values) ++b; return b; }

// At the first evaluation of a window (and again whenever the spline timing changes): the control-pose pair of every measurement,
// the device order (pixel order, or tile order from the panorama positions the given control poses predict) and the record slots
// sorted by pair.  Pose-independent except for the tile binning, which only affects speed.
emba_status prepare_order(emba_ctx* c, const double* knots_host, int64_t t0, int64_t dt, int K)
{
    if (c->keys_ready && c->key_t0 == t0 && c->key_dt == dt && c->key_K == K) return EMBA_OK;
    if (dt <= 0 || K < 2 || K > 65535) return fail(c, EMBA_ERR_INVALID_ARG, "bad spline: dt_ns=%lld K=%d", (long long)dt, K);
    const auto t_begin = std::chrono::steady_clock::now();
    hipStream_t s = c->stream;
    emba_status st;
    const size_t ns = c->n_pm, nbatch = c->n_batch;
    // (a previous order of this window goes away; its buffers are reused)
    c->d_ev_pix = nullptr; c->d_ev_batch = nullptr;
    c->tile_order = false; c->have_ev_pm = false; c->n_chunks = 0; c->n_lead = 0;
    c->order_per_px = 0.0; c->order_lead_frac = 0.0; c->order_inl_pred = 0.0;

    // control-pose index per batch; a batch outside the knots is an error (BASALT_ASSERT_STREAM at so3_spline.h:221-229)
    uint32_t* d_err = nullptr;
    if ((st = ws_get(c, 19, 64, (void**)&d_err))) return st;
    HIP_TRY(c, hipMemsetAsync(d_err, 0xFF, 64, s));
    if ((st = dev_alloc(c, &c->d_cp, nbatch)) || (st = dev_alloc(c, &c->d_batch_u, nbatch))) return st;
    if (nbatch) hipLaunchKernelGGL(emba_batch_cp_kernel, dim3(nblocks(nbatch)), dim3(256), 0, s, c->d_batch_t, (long)nbatch, t0, dt, K, c->d_cp, c->d_batch_u, d_err);
    uint32_t h_err[16];
    HIP_TRY(c, hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (h_err[0] != 0xFFFFFFFFu) {
        int64_t tb = 0;
        (void)hipMemcpy(&tb, c->d_batch_t + h_err[0], 8, hipMemcpyDeviceToHost);
        return fail(c, EMBA_ERR_TIME_RANGE, "batch %u midpoint %lld ns outside spline [%lld, %lld) (K=%d)", h_err[0], (long long)tb, (long long)t0,
                    (long long)(t0 + dt * (K - 1)), K);
    }

    // ---- which order?  Tile order pays when many events share a panorama pixel (the per-pixel sums are then combined in LDS) and the
    // chains of a sensor pixel stay in a tile for a while (every tile entry costs one extra warp of the predecessor).
    // Round 6: WHICH tile (order_kernels.h: the window rule).  The 1152-pixel LDS tile comes in four shapes (kernels.h: kTileShapes); every chain is cut into the
    // longest segments that fit a tile of the shape on its pitch grid, and the shape with the fewest entries + chunks wins — a fast pan wants wide tiles, a
    // trajectory that pitches wants tall ones (scripts/lead_in_sim.py: 5 M events over 4.8 s at 0.5 rad/s: 19.5 % lead-ins at 48 x 24, 11.5 % at 72 x 16).
    BinGeom g{};
    uint32_t *d_bin = nullptr, *d_emit = nullptr, *d_pos = nullptr, *d_pred = nullptr; uint8_t* d_used = nullptr; unsigned long long* d_breaks = nullptr;
    bool tile = false;
    size_t n_break = 0, n_used_bins = 0, nbins = 1;
    // (auto on a window below tile_min_events never takes the tile order: nothing to analyse — 1.2 -> 0.3 ms of a 1 M-event window's first evaluation)
    if (c->order_mode != 1 && ns && knots_host && (c->order_mode == 2 || c->n_used >= (size_t)c->opt_tile_min_events)) {
        const size_t max_bins = (size_t)((c->W + 7) / 8) * ((c->H + 1) / 2) + 1;      // (the finest pitch any candidate uses is 8 x 2)
        if ((st = ws_get(c, 20, ns * 4, (void**)&d_bin)) || (st = ws_get(c, 21, ns * 4, (void**)&d_emit)) || (st = ws_get(c, 22, ns * 4, (void**)&d_pos)) ||
            (st = ws_get(c, 23, max_bins + 8, (void**)&d_used)) || (st = ws_get(c, 29, ns * 4, (void**)&d_pred)))
            return st;
        d_breaks = reinterpret_cast<unsigned long long*>(d_err + 8);
        // the poses the caller starts from (staged through the pinned buffer like every evaluation's)
        HIP_TRY(c, hipMemcpyAsync(c->d_knots, knots_host, (size_t)4 * K * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(emba_pose_kernel, dim3(nblocks(nbatch, 64)), dim3(64), 0, s, c->d_batch_t, (int)nbatch, c->d_knots, K, t0, dt, c->d_pose, c->d_err);
        hipLaunchKernelGGL(emba_predict_pixel_kernel, dim3(nblocks(ns)), dim3(256), 0, s, c->d_pm_pix, c->d_pm_batch, (long)ns, c->d_pose, kPoseStride, c->d_lut,
                           c->fx, c->fy, c->cx, c->cy, c->W, c->H, d_pred);
        // the chains of the pm-order: heads -> list (one wave of the assignment kernel per chain)
        uint32_t *d_hflag = d_emit, *d_hpos = d_pos, *d_heads = nullptr, *d_nheads = d_err + 6;
        if ((st = ws_get(c, 30, ns * 4, (void**)&d_heads))) return st;
        hipLaunchKernelGGL(emba_head_flag_kernel, dim3(nblocks(ns)), dim3(256), 0, s, c->d_pm_pix, (long)ns, d_hflag);
        if ((st = dev_scan(c, d_hflag, d_hpos, ns, d_nheads))) return st;
        hipLaunchKernelGGL(emba_head_list_kernel, dim3(nblocks(ns)), dim3(256), 0, s, d_hflag, d_hpos, (long)ns, d_heads);
        uint32_t h_nheads = 0;
        HIP_TRY(c, hipMemcpyAsync(&h_nheads, d_nheads, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        const long n_heads = (long)h_nheads;
        double f_pred = 1.0;
        {
            HIP_TRY(c, hipMemsetAsync(d_breaks, 0, 8, s));
            hipLaunchKernelGGL(emba_count_pred_inliers_kernel, dim3((unsigned)std::min<size_t>(nblocks(ns), 1024)), dim3(256), 0, s, c->d_pm_pix, d_pred, (long)ns, c->outlier_px, d_breaks);
            unsigned long long hi = 0;
            HIP_TRY(c, hipMemcpyAsync(&hi, d_breaks, 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            f_pred = c->n_used ? (double)hi / (double)c->n_used : 0.0;
        }
        c->order_inl_pred = f_pred;
        const bool hopeless = c->order_mode == 0 && 83.0 * f_pred <= 41.0;      // the rule below cannot hold even without a single lead-in: skip the search for a tile
        const int r = std::max(0, std::min(c->opt_tile_reserve, 5));
        std::vector<uint8_t> h_used(max_bins);
        auto geom = [&](int shape, bool fine) {
            BinGeom q{};
            const TileShape& ts = kTileShapes[shape];
            q.W = c->W; q.H = c->H; q.tw = ts.tw; q.th = ts.th; q.r = r;
            q.bw = fine ? ts.fine_pw : ts.pw; q.bh = fine ? ts.fine_ph : ts.ph;
            q.bw = std::min(q.bw, ts.tw - 2 * r); q.bh = std::min(q.bh, ts.th - 2 * r);     // (a single pixel must fit wherever it lies in its pitch cell)
            q.nbx = (c->W + q.bw - 1) / q.bw; q.nby = (c->H + q.bh - 1) / q.bh;
            return q;
        };
        // lead-ins and occupied tiles of one candidate; leaves its assignment in d_bin / d_emit
        auto evaluate = [&](const BinGeom& q, size_t* breaks, size_t* used) -> emba_status {
            const size_t nb_ = (size_t)q.nbx * q.nby + 1;
            HIP_TRY(c, hipMemsetAsync(d_used, 0, nb_, s));
            HIP_TRY(c, hipMemsetAsync(d_breaks, 0, 8, s));
            hipLaunchKernelGGL(emba_assign_tiles_kernel, dim3((unsigned)((n_heads + 3) / 4)), dim3(256), 0, s, c->d_pm_pix, d_pred, d_heads, n_heads, (long)ns, q, d_bin, d_used);
            hipLaunchKernelGGL(emba_expand_count_kernel, dim3(nblocks(ns)), dim3(256), 0, s, c->d_pm_pix, d_bin, (long)ns, d_emit);
            hipLaunchKernelGGL(emba_count_breaks_kernel, dim3((unsigned)std::min<size_t>(nblocks(ns), 1024)), dim3(256), 0, s, d_emit, (long)ns, d_breaks);
            unsigned long long hb = 0;
            HIP_TRY(c, hipMemcpyAsync(h_used.data(), d_used, nb_, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(&hb, d_breaks, 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            *breaks = (size_t)hb; *used = 0;
            for (size_t b = 0; b + 1 < nb_; ++b) *used += h_used[b];
            return EMBA_OK;
        };
        // what a candidate costs the warp kernel, in entries: every lead-in is a full warp, every chunk zeroes and flushes an LDS tile (~ 256 entries' worth:
        // 3 M events, round 3: 902 -> 2520 entries per chunk took the kernel from 191 to 174 us)
        auto cost = [&](size_t breaks, size_t used) {
            const double entries = (double)c->n_used + (double)breaks;
            const double chunks = std::max((double)used, entries / (double)(kWarpNew * kTileWaves * 8));
            return entries + 256.0 * chunks;
        };
        if (!hopeless) {
        int best = -1; bool best_fine = false; double best_cost = 0; size_t best_breaks = 0, best_used = 0; int last = -1; bool last_fine = false;
        // (the tall shape first, the others have to beat it by 2 %: at equal lead-ins 36 x 32 measured 3-5 % faster than 48 x 24 — 3 M events 155 vs 161 us, config 4's
        // shard 245 vs 260, 2 M 111 vs 115, city 463 vs 465; profiles/r06_regime_sweep.txt)
        static const int kShapeOrder[kNumTileShapes] = {3, 0, 1, 2};
        for (int si = 0; si < kNumTileShapes; ++si) {
            const int sh = kShapeOrder[si];
            if (c->opt_tile_shape >= 0 && sh != c->opt_tile_shape) continue;
            size_t br = 0, us = 0;
            if ((st = evaluate(geom(sh, false), &br, &us))) return st;
            last = sh; last_fine = false;
            const double cc = cost(br, us);
            if (best < 0 || cc < 0.98 * best_cost) { best = sh; best_fine = false; best_cost = cc; best_breaks = br; best_used = us; }
        }
        // dense tiles can afford a finer grid of tile origins (segments end closer to the tile's far edge, and a tile's entries spread over more chunks): config 4's
        // shard (11.7 k entries per tile) 258 -> 229 us; at 3-5 k entries per tile it loses (3 M: 162 -> 173, 5 M: 236 -> 271)
        // (... and only below 16 M events: the finer grid doubles the chunks — 40 M events 1749 -> 1879 us, config 5's shard 566 -> 655, 100 M + 2 % per step;
        // profiles/r06_large_window_ab.txt)
        if (best >= 0 && c->opt_tile_fine != 0 &&
            (c->opt_tile_fine == 1 || (c->n_used < (size_t)16000000 && (double)c->n_used / std::max<size_t>(best_used, 1) >= 2.0 * kWarpNew * kTileWaves * 8))) {
            size_t br = 0, us = 0;
            if ((st = evaluate(geom(best, true), &br, &us))) return st;
            last_fine = true; last = best;
            const double cc = cost(br, us);
            if (cc < best_cost || c->opt_tile_fine == 1) { best_fine = true; best_cost = cc; best_breaks = br; best_used = us; }
        }
        if (best < 0) return fail(c, EMBA_ERR_INVALID_ARG, "option tile_shape %d: no such shape", c->opt_tile_shape);
        g = geom(best, best_fine);
        if (last != best || last_fine != best_fine) { size_t br = 0, us = 0; if ((st = evaluate(g, &br, &us))) return st; }   // d_bin / d_emit of the winner
        c->tile_shape = best; c->tile_fine = best_fine;
        nbins = (size_t)g.nbx * g.nby + 1;
        n_break = best_breaks; n_used_bins = best_used;
        const double per_px = n_used_bins ? (double)c->n_used / ((double)n_used_bins * g.bw * g.bh) : 0.0;     // events per panorama pixel of the occupied pitch cells
        const double lead_frac = c->n_used ? (double)n_break / (double)c->n_used : 1.0;
        // Measured (profiles/r02c_order_sweep.txt): the tile order wins once the working set has left the Infinity Cache (3 M events, 1024x2048:
        // 374 vs 394 us per step; 5 M / K=97: 558 vs 618; 100 M: 4.8 vs 8.7 ms warp) and loses below it (1 M events, 24 per pixel: 82 vs 52 us
        // — every entry of the tile order is a warp, lead-ins included, and a workgroup's LDS tile is zeroed and flushed for a handful of groups).
        // (round 3, with at least 5 groups per wave and chunk: 2 M events 220 vs 246 us per step, 1.5 M 188 vs 155 — the pixel order falls off a cliff
        // between 1.5 M and 2 M events: twice the events on the same footprint are twice as close along a chain, nearly all of them inliers — 3 x the atomic requests)
        // (round 5: a slow pan over a big sensor — the city shape at 0.1 rad/s: 10 M events on 640x480, 50 events per panorama pixel — is
        // atomic-request bound in pixel order (7.3 M requests on 155 k lines); the tile order's LDS sums win there in spite of the extra entries: step 867 vs
        // 946-995 us.  Hence the second clause: very dense tiles tolerate more lead-ins.)
        // Round 6: the rule prices both orders.  Per million events, fitted to profiles/r06_regime_sweep.txt and r06_sparse_order_ab.txt (warp + Gram kernels, us):
        // pixel order 20 + 93 f (it pays per INLIER: an atomic request, a record, a live slot of the Gram kernel's tag stream; f = inlier fraction), tile order
        // 31 (1 + lead) + 10 f + 30 (it pays per ENTRY, lead-in copies included, and its Gram kernel reads every candidate's slot).  The tile order wins where
        // 83 f > 41 + 31 lead.  f is estimated from the predicted pixels (emba_count_pred_inliers_kernel).  (Rounds 2-5 asked for lead <= 0.35 only: with the window
        // rule's fewer lead-ins that sent a 34 %-inlier stream — 10 M events on 640x480 at 0.5 rad/s — to the tile order: 850 us per step against 636 in pixel order.)
        tile = (c->order_mode == 2) || (c->n_used >= (size_t)c->opt_tile_min_events && per_px >= 8.0 && 83.0 * f_pred > 41.0 + 31.0 * lead_frac);
        c->order_per_px = per_px; c->order_lead_frac = lead_frac;
        }
    }

    if (!tile) {
        c->d_ev_pix = c->d_pm_pix; c->d_ev_batch = c->d_pm_batch;
        c->n_sorted = ns;
        // per entry the spline parameter and segment of its batch (the warp kernel's per-event pose: 10 B per entry, streamed with the event words)
        if ((st = dev_alloc(c, &c->d_ev_u, std::max<size_t>(ns, 1))) || (st = dev_alloc(c, &c->d_ev_seg, std::max<size_t>(ns, 1)))) return st;
        if (ns) hipLaunchKernelGGL(emba_entry_pose_args_kernel, dim3(nblocks(ns)), dim3(256), 0, s, c->d_pm_batch, (long)ns, c->d_cp, c->d_batch_u, c->d_ev_seg, c->d_ev_u);
    } else {
        // expanded list: every event, plus a lead-in copy of its predecessor where its chain enters a tile -> stable sort by tile
        uint32_t* d_tot = d_err + 4;
        if ((st = dev_scan(c, d_emit, d_pos, ns, d_tot))) return st;
        uint32_t nd32 = 0;
        HIP_TRY(c, hipMemcpyAsync(&nd32, d_tot, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        const size_t nd = nd32;
        uint32_t *k0 = nullptr, *v0 = nullptr, *k1 = nullptr, *v1 = nullptr, *d_bin_start = nullptr;
        if ((st = ws_get(c, 24, nd * 4, (void**)&k0)) || (st = ws_get(c, 25, nd * 4, (void**)&v0)) || (st = ws_get(c, 26, nd * 4, (void**)&k1)) ||
            (st = ws_get(c, 27, nd * 4, (void**)&v1)) || (st = ws_get(c, 28, (nbins + 1) * 4, (void**)&d_bin_start)))
            return st;
        hipLaunchKernelGGL(emba_expand_write_kernel, dim3(nblocks(ns)), dim3(256), 0, s, c->d_pm_pix, d_bin, d_emit, d_pos, (long)ns, k0, v0);
        if ((st = dev_sort(c, &k0, &v0, &k1, &v1, nd, bits_for(nbins)))) return st;
        if ((st = dev_alloc(c, &c->d_ev_pix_own, nd)) || (st = dev_alloc(c, &c->d_ev_batch_own, nd)) || (st = dev_alloc(c, &c->d_ev_pm, nd)) ||
            (st = dev_alloc(c, &c->d_ev_u, nd)) || (st = dev_alloc(c, &c->d_ev_seg, nd)))
            return st;
        c->d_ev_pix = c->d_ev_pix_own; c->d_ev_batch = c->d_ev_batch_own; c->have_ev_pm = true;
        HIP_TRY(c, hipMemsetAsync(d_bin_start, 0xFF, (nbins + 1) * 4, s));
        uint32_t* d_cf = d_emit;    // (emit is dead: its buffer now takes the candidate flags of the device order — nd <= 2 ns may exceed it)
        if (nd > ns && (st = ws_get(c, 21, nd * 4, (void**)&d_cf))) return st;
        hipLaunchKernelGGL(emba_dev_gather_kernel, dim3(nblocks(nd)), dim3(256), 0, s, k0, v0, (long)nd, c->d_pm_pix, c->d_pm_batch, c->d_ev_pix, c->d_ev_batch,
                           c->d_ev_pm, c->d_cp, c->d_batch_u, c->d_ev_seg, c->d_ev_u, d_cf, d_bin_start);
        // chunks: every occupied tile is cut into workgroup-sized pieces (host: <= 32 k tiles)
        std::vector<uint32_t> h_start(nbins + 1);
        HIP_TRY(c, hipMemcpyAsync(h_start.data(), d_bin_start, (nbins + 1) * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        std::vector<std::pair<uint32_t, uint32_t>> occ;   // (bin, start)
        for (size_t b = 0; b < nbins; ++b) if (h_start[b] != 0xFFFFFFFFu) occ.emplace_back((uint32_t)b, h_start[b]);
        std::vector<ChunkDesc> h_chunks;
        const size_t slots = (size_t)c->n_cu * 2;      // workgroups of the tiled kernel the chip holds at a time
        {
        // chunk size: enough workgroups for ~8 rounds of the chip, at most 16 groups of 63 entries per wave
        size_t chunk = (nd + slots * 8 - 1) / (slots * 8);
        // (at least 5 groups per wave: a workgroup zeroes and flushes its 55-KB LDS tile whatever it has to do — 3 M events: 902 -> 2520 entries per
        // chunk, warp kernel 191 -> 174 us; 5 M: 280 -> 271; from 10 M on the first rule gives more than that anyway)
        // round 4, with the chunks dispatched longest first: 8 groups per wave up to ~8 M entries (3 M events: 1339 chunks, warp kernel 160 -> 153 us;
        // 5 M: 257 -> 242), 5 beyond (10 M: 8 groups 493 us, 5 groups 464 — there the first rule decides most chunks anyway)
        const size_t min_groups = nd < (size_t)8000000 ? 8 : 5;
        chunk = std::min<size_t>(std::max<size_t>(chunk, (size_t)kWarpNew * kTileWaves * min_groups), (size_t)kWarpNew * kTileWaves * 16);
        if (c->opt_tile_chunk > 0) chunk = (size_t)c->opt_tile_chunk;
        for (size_t k = 0; k < occ.size(); ++k) {
            const uint32_t b = occ[k].first, b0 = occ[k].second, b1 = (k + 1 < occ.size()) ? occ[k + 1].second : (uint32_t)nd;
            // (pieces in whole ROUNDS of the workgroup's waves: a piece of G groups takes ceil(G / 8) rounds whatever it holds, so only a tile's last piece may be ragged)
            constexpr size_t kRound = (size_t)kWarpNew * kTileWaves;
            const size_t cnt = b1 - b0, nch = (cnt + chunk - 1) / chunk, per = ((cnt + nch - 1) / nch + kRound - 1) / kRound * kRound;
            const int bx = (int)(b % (uint32_t)g.nbx), by = (int)(b / (uint32_t)g.nbx);
            for (size_t q = 0; q < nch; ++q) {
                ChunkDesc d;
                d.begin = b0 + (uint32_t)(q * per); d.end = (uint32_t)std::min<size_t>(b0 + (q + 1) * per, b1);
                d.x0 = bx * g.bw - g.r; d.y0 = by * g.bh - g.r;
                if (d.begin < d.end) h_chunks.push_back(d);
            }
        }
        // Chunk sizes differ (every bin is cut on its own) and the grid is a few rounds of the chip's workgroup slots: with the longest chunks FIRST the
        // last round is made of the short ones (longest-processing-time order; workgroups are dispatched in grid order as slots free up).
        // Option chunk_order_bin keeps the bins' order (neighbouring chunks on one XCD).
        c->chunks_lpt = !c->opt_chunk_order_bin;
        if (c->chunks_lpt) std::stable_sort(h_chunks.begin(), h_chunks.end(), [](const ChunkDesc& a, const ChunkDesc& b) { return a.end - a.begin > b.end - b.begin; });
        }
        c->n_chunks = (long)h_chunks.size();
        if ((st = dev_alloc(c, &c->d_chunks, h_chunks.size()))) return st;
        HIP_TRY(c, hipMemcpyAsync(c->d_chunks, h_chunks.data(), h_chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        c->n_sorted = nd; c->n_lead = nd - c->n_used; c->tile_order = true;
        d_emit = d_cf;
    }
    const size_t nd = c->n_sorted;
    c->nblk = (long)((nd + kWarpNew - 1) / kWarpNew);

    // record slots sorted by control-pose pair: candidates of the device order -> (key, entry) -> stable sort -> slot
    {
        const size_t M = c->n_cand;
        uint32_t *d_cf = nullptr, *d_cpos = nullptr, *k0 = nullptr, *v0 = nullptr, *k1 = nullptr, *v1 = nullptr;
        if ((st = ws_get(c, 22, std::max<size_t>(nd, 1) * 4, (void**)&d_cpos))) return st;
        if (tile) d_cf = d_emit;
        else {
            if ((st = ws_get(c, 21, std::max<size_t>(nd, 1) * 4, (void**)&d_cf))) return st;
            if (nd) hipLaunchKernelGGL(emba_cand_flag_kernel, dim3(nblocks(nd)), dim3(256), 0, s, c->d_ev_pix, (long)nd, d_cf);
        }
        if ((st = dev_scan(c, d_cf, d_cpos, nd, nullptr))) return st;
        if ((st = ws_get(c, 24, std::max<size_t>(M, 1) * 4, (void**)&k0)) || (st = ws_get(c, 25, std::max<size_t>(M, 1) * 4, (void**)&v0)) ||
            (st = ws_get(c, 26, std::max<size_t>(M, 1) * 4, (void**)&k1)) || (st = ws_get(c, 27, std::max<size_t>(M, 1) * 4, (void**)&v1)))
            return st;
        if ((st = dev_alloc(c, &c->d_ev_slot, nd))) return st;
        HIP_TRY(c, hipMemsetAsync(c->d_ev_slot, 0xFF, std::max<size_t>(nd, 1) * 4, s));
        if (nd) hipLaunchKernelGGL(emba_cand_keys_kernel, dim3(nblocks(nd)), dim3(256), 0, s, c->d_ev_pix, c->d_ev_batch, c->d_cp, (long)nd, d_cpos, k0, v0);
        // the key's low half is cp_p, the high half cp_c, both < K: sort on the bits they really use
        const int kb = bits_for((size_t)K);
        if (kb <= 8) {   // fold the two halves into one 16-bit key for the sort (two passes instead of four)
            if (M) hipLaunchKernelGGL(emba_fold_keys_kernel, dim3(nblocks(M)), dim3(256), 0, s, k0, (long)M, 1);
            if ((st = dev_sort(c, &k0, &v0, &k1, &v1, M, 16))) return st;
            if (M) hipLaunchKernelGGL(emba_fold_keys_kernel, dim3(nblocks(M)), dim3(256), 0, s, k0, (long)M, 0);
        } else if ((st = dev_sort(c, &k0, &v0, &k1, &v1, M, 32))) return st;
        if (M) hipLaunchKernelGGL(emba_slot_assign_kernel, dim3(nblocks(M)), dim3(256), 0, s, k0, v0, (long)M, c->d_ev_slot, c->d_slot_key);
    }
    // per-entry outputs of the evaluations
    // per-event outputs of the evaluations, indexed in pm-order in both orders
    if ((st = dev_alloc(c, &c->d_e_sorted, ns)) || (st = dev_alloc(c, &c->d_flag, ns)) || (st = dev_alloc(c, &c->d_inl_idx, ns))) return st;
    HIP_TRY(c, hipMemsetAsync(c->d_flag, 0, std::max<size_t>(ns, 1), s));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(s));
    c->keys_ready = true; c->key_t0 = t0; c->key_dt = dt; c->key_K = K;
    c->prepare_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return EMBA_OK;
}

// the tag stream pays where slots are dead (pixel order: about half of them at the BASELINE workload); in the tile order (dense regime: nearly every
// slot is live) the warp kernel's scattered 8-B tag stores cost more than the Gram kernel saves (40 M events: +370 vs -180 us)
bool gram_uses_tags(const emba_ctx* c, bool ep_host) { return c->use_tags && !c->tile_order && !ep_host; }

emba_status ensure_pack(emba_ctx* c, int K)
{
    const size_t need = (size_t)9 * K * K + (size_t)3 * K + 5 * c->npix;
    if (c->pack_bound) {
        if (c->pack_cap < (size_t)9 * K * K + (size_t)3 * K)
            return fail(c, EMBA_ERR_CAPACITY, "bound pack buffer too small for K=%d", K);
        return EMBA_OK;
    }
    if (c->pack_own_cap < need) {
        dev_free(c, c->d_pack_own);
        emba_status st = dev_alloc(c, &c->d_pack_own, need);
        if (st) return st;
        c->pack_own_cap = need;
    }
    c->d_pack = c->d_pack_own;
    c->pack_cap = c->pack_own_cap;
    return EMBA_OK;
}

inline double* pack_A11(emba_ctx* c) { return c->d_pack; }
inline double* pack_b1(emba_ctx* c) { return c->d_pack + (size_t)9 * c->K * c->K; }
inline double* pack_A22b2(emba_ctx* c) { return c->d_pack + (size_t)9 * c->K * c->K + (size_t)3 * c->K; }

long grid8(long n) { return (n + 7) / 8 * 8; }

// The pano -> compact index map is only read by the generic (weighted / external-ep) A22 path, the A12 exports and the Schur solve,
// so it is produced when one of them asks (8 MB less traffic on every ordinary step).
emba_status ensure_compact(emba_ctx* c)
{
    if (c->compact_valid) return EMBA_OK;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->d_compact, 0xFF, c->npix * sizeof(int32_t), s));
    const size_t bound = c->P_pending ? c->npix : c->P;
    if (bound)
        hipLaunchKernelGGL(emba_compact_map_kernel, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, s, c->d_active, c->d_total + 1, c->d_compact);
    HIP_TRY(c, hipGetLastError());
    c->compact_valid = true;
    return EMBA_OK;
}

// The warp kernel only MARKS touched pixels in the int32 count map (the count itself is accumulated next to the A22/b2 sums, one
// atomic request per measurement).  The first post-warp launch of the resident step turns the markers into counts as a side
// effect of its dense pass; whoever needs num_ev_map before that (download, exchange 1, the non-fused active-set path) calls this.
emba_status ensure_counts(emba_ctx* c)
{
    if (!c->counts_raw) return EMBA_OK;
    c->counts_raw = false;
    c->count_stamp = (c->d_count == c->d_count_own) ? c->set_stamp : 0u;
    hipLaunchKernelGGL(emba_count_materialise_kernel, dim3((unsigned)((c->npix + 2047) / 2048)), dim3(256), 0, c->stream, c->d_count, c->d_pixacc, (long)c->npix, c->count_mark);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

// Standalone residual compaction (scan of the per-wave inlier counts, then the compaction): used when the host asks for
// ep / counts before the active-set kernels run; otherwise emba_form_active launches it fused with its own stages.
emba_status launch_ep_compaction(emba_ctx* c)
{
    if (!c->ep_deferred) return EMBA_OK;
    c->ep_deferred = false;
    hipStream_t s = c->stream;
    if (c->n_pm) {
        const uint32_t* perm = nullptr;    // (flags and residuals are stored in pm-order by both warp kernels)
        hipLaunchKernelGGL(emba_flag_count_kernel, dim3((unsigned)c->n_fblk), dim3(256), 0, s, c->d_flag, perm, (long)c->n_pm, c->d_fblk_cnt);
        // (round 6: the block counts by the three-launch scan — one 256-thread block walked all of them before: 101 us for 97 k counts at 100 M events)
        { emba_status st = dev_scan(c, c->d_fblk_cnt, c->d_fblk_off, (size_t)c->n_fblk, c->d_total, c->h_pinned_dev, c->d_err, c->h_pinned_dev + 1); if (st) return st; }
        hipLaunchKernelGGL(emba_compact_ep_kernel, dim3((unsigned)c->n_fblk), dim3(256), 0, s, c->d_e_sorted, c->d_flag, perm, c->d_fblk_off,
                           (long)c->n_pm, c->d_ep, c->d_inl_idx);
        c->inl_idx_valid = true; c->ep_valid = true;
        HIP_TRY(c, hipGetLastError());
    } else {
        HIP_TRY(c, hipMemsetAsync(c->d_total, 0, sizeof(uint32_t), s));
        HIP_TRY(c, hipMemcpyAsync(&c->h_pinned[0], c->d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(&c->h_pinned[1], c->d_err, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    c->inl_pending = true;
    return EMBA_OK;
}

// Per-event inlier numbers (index into ep), for the consumers that need them: the fused post-warp launches skip them.
emba_status ensure_inl_idx(emba_ctx* c)
{
    { emba_status st = launch_ep_compaction(c); if (st) return st; }
    if (c->inl_idx_valid || !c->n_pm) return EMBA_OK;
    // (the fused step leaves the per-block inlier counts of this evaluation in d_fblk_cnt, not their prefix)
    { emba_status st = dev_scan(c, c->d_fblk_cnt, c->d_fblk_off, (size_t)c->n_fblk, c->d_total); if (st) return st; }
    hipLaunchKernelGGL(emba_compact_ep_kernel, dim3((unsigned)c->n_fblk), dim3(256), 0, c->stream, c->d_e_sorted, c->d_flag, (const uint32_t*)nullptr, c->d_fblk_off,
                       (long)c->n_pm, c->d_ep, c->d_inl_idx);   // (ep is rewritten with the same values)
    HIP_TRY(c, hipGetLastError());
    c->inl_idx_valid = true; c->ep_valid = true;
    return EMBA_OK;
}

// Synchronize the stream and turn the counters that were read back asynchronously (inlier count, device error
// word, active-pixel count) into host state.  Called only where the host really needs a value.
// counts_only: the caller needs the inlier / active-pixel counts and nothing else from the device.  When they were produced by
// the fused post-warp kernels, the host polls the sequence word those kernels publish after the counts instead of waiting for
// the whole stream: it returns while the later kernels of the step (active-set gather, Gram) still run, so the next step's
// launches queue up behind them and the GPU never idles for a host round trip.  Everything that reads device data on the
// host goes through the full form (counts_only = false), which drains the stream.
emba_status resolve_pending(emba_ctx* c, bool counts_only = false)
{
    { emba_status st = launch_ep_compaction(c); if (st) return st; }
    if (!c->inl_pending && !c->P_pending) {
        if (c->spun && !counts_only) { HIP_TRY(c, hipStreamSynchronize(c->stream)); c->spun = false; c->knots_in_flight = false; }
        return EMBA_OK;
    }
    bool polled = false;
    if (counts_only && c->seq_armed && c->inl_pending && c->P_pending) {
        volatile int* w = c->h_pinned + 3;                   // [3] behind the active-pixel count, [4] behind the inlier count
        for (long spin = 0; spin < 50000000L; ++spin) {      // bounded: a faulted kernel never publishes; fall back to the stream
            if (w[0] == c->seq && w[1] == c->seq) { polled = true; break; }
            __builtin_ia32_pause();
        }
        if (polled) std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (polled) c->spun = true;
    else { HIP_TRY(c, hipStreamSynchronize(c->stream)); c->spun = false; }
    c->seq_armed = false;
    c->knots_in_flight = false;   // (the prep kernel that reads the pinned knot buffer precedes the post-warp kernels)
    if (c->inl_pending) {
        c->inl_pending = false;
        if (c->h_pinned[1] & 1) return fail(c, EMBA_ERR_TIME_RANGE, "a batch midpoint lies outside the spline's knots");
        c->n_inliers = (size_t)(uint32_t)c->h_pinned[0];
        c->n_outside_tile = (size_t)((uint32_t)c->h_pinned[1] >> 1);
        // Tile order: the bins were predicted with the trajectory of the window's first evaluation.  Events that have since left their
        // tile (+ margin) are still handled correctly, but one by one through HBM atomics; once that is a fifth of the inliers the order
        // is rebuilt from the CURRENT trajectory at the next evaluation (an LM loop that starts far from its solution).
        // (not again within three evaluations: a loop whose every trial moves the events further than the margin would re-bin each time)
        if (c->tile_order && c->n_inliers && c->n_outside_tile * 5 > c->n_inliers && c->rec_stamp - c->last_rebin_stamp >= 3) {
            c->keys_ready = false; ++c->n_rebin; c->last_rebin_stamp = c->rec_stamp;
        }
        c->eval_done = true;
    }
    if (c->P_pending) {
        c->P_pending = false;
        c->P = (size_t)(uint32_t)c->h_pinned[2];
        c->P_prev = c->P;
        c->pack_len = (size_t)9 * c->K * c->K + (size_t)3 * c->K + 5 * c->P;
        if (c->pack_len > c->pack_cap)
            return fail(c, EMBA_ERR_CAPACITY, "pack buffer too small: need %zu doubles, have %zu", c->pack_len, c->pack_cap);
        c->active_done = true;
    }
    return EMBA_OK;
}

// Device -> host into memory the CALLER owns (pageable: an Eigen vector, a cv::Mat, a numpy array).  hipMemcpy stages such a copy through the runtime's own bounce
// buffers one chunk after the other; here the DMA of chunk i + 1 into one pinned buffer runs while the CPU copies chunk i out of the other — the two halves of the
// drop-in's largest transfer (ep: 56 MB per evaluateDataError at 10 M events) overlap instead of adding up.  The stream must have been drained up to `src`'s producer.
// A few helper threads for the CPU half of a large device -> pageable copy (round 6): memcpy into FRESH pages is bound by the page faults of the one thread that touches
// them (ep into the vector evaluateDataError returns, 60 MB at config 2's shape: 8.8 ms = 6.8 GB/s).  The pool is process-wide, created at the first large copy and never
// torn down (its threads sleep on a condition variable; a caller that arrives while another copy runs copies alone).
struct CopyPool {
    static constexpr int kHelpers = 3;
    std::mutex m, use; std::condition_variable go, done;
    uint64_t gen = 0; int pending = 0; bool started = false;
    const std::function<void(int)>* job = nullptr;      // job(h), h = 0 (the caller) .. kHelpers
    static void piece(int h, size_t n, size_t& lo, size_t& hi)
    {
        const size_t per = ((n / (kHelpers + 1)) + 4095) & ~(size_t)4095;      // whole pages to every thread
        lo = std::min(n, per * (size_t)h); hi = (h == kHelpers) ? n : std::min(n, per * (size_t)(h + 1));
    }
    void worker(int h)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            { std::unique_lock<std::mutex> l(m); go.wait(l, [&] { return gen != seen; }); seen = gen; f = job; }
            (*f)(h + 1);
            { std::lock_guard<std::mutex> l(m); if (--pending == 0) done.notify_one(); }
        }
    }
    // f(0) on the caller, f(1 .. kHelpers) on the helpers; alone (f(0 .. kHelpers) in turn) when another caller holds the pool
    void run(const std::function<void(int)>& f)
    {
        std::unique_lock<std::mutex> u(use, std::try_to_lock);
        if (!u.owns_lock()) { for (int h = 0; h <= kHelpers; ++h) f(h); return; }
        if (!started) { for (int h = 0; h < kHelpers; ++h) std::thread([this, h] { worker(h); }).detach(); started = true; }
        { std::lock_guard<std::mutex> l(m); job = &f; pending = kHelpers; ++gen; }
        go.notify_all();
        f(0);
        { std::unique_lock<std::mutex> l(m); done.wait(l, [&] { return pending == 0; }); }
    }
    void copy(void* d, const void* sp, size_t nn)
    {
        if (nn < ((size_t)1 << 20)) { std::memcpy(d, sp, nn); return; }
        run([&](int h) { size_t lo, hi; piece(h, nn, lo, hi); if (hi > lo) std::memcpy((char*)d + lo, (const char*)sp + lo, hi - lo); });
    }
};
CopyPool* copy_pool()      // (never destroyed: its detached threads may outlive every context)
{
    static CopyPool* p = [] {
        CopyPool* q = new CopyPool;
        // a fork()ed child has none of the helper threads and possibly a mutex that a thread of the parent held: it starts from a fresh pool
        static CopyPool* self = q;
        (void)pthread_atfork(nullptr, nullptr, [] { new (self) CopyPool; });
        return q;
    }();
    return p;
}

// Have the pages of [p, p + bytes) mapped before they are written: memory a caller has just allocated (the vector evaluateDataError returns) has no pages yet, and
// a first write per page is a trap each (60 MB: 15 k of them on the copying thread).  One MADV_POPULATE_WRITE per piece does the same inside the kernel, without
// changing what the pages hold; where the kernel does not know it (< 5.14) the pages are simply faulted in by the copy that follows.
void populate_pages(void* p, size_t bytes)
{
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
    const uintptr_t lo = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE);
}
void populate_pages_parallel(void* p, size_t bytes)
{
    if (bytes < ((size_t)4 << 20)) return;
    copy_pool()->run([&](int h) { size_t lo, hi; CopyPool::piece(h, bytes, lo, hi); if (hi > lo) populate_pages((char*)p + lo, hi - lo); });
}

// device -> host in pipelined chunks through the context's two pinned buffers: the DMA of chunk i + 1 runs while `consume(chunk, byte offset, bytes)` works on chunk i.
// The stream must have been drained up to `src`'s producer.
emba_status d2h_chunks(emba_ctx* c, const void* src, size_t bytes, const std::function<void(const void*, size_t, size_t)>& consume,
                       const std::function<void()>& while_first_chunk_travels = nullptr)
{
    if (!bytes) return EMBA_OK;
    constexpr size_t kChunk = (size_t)8 << 20;
    for (int k = 0; k < 2; ++k)
        if (!c->h_stage[k]) { HIP_TRY(c, hipHostMalloc(&c->h_stage[k], kChunk, hipHostMallocDefault)); HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[k], hipEventDisableTiming)); }
    hipStream_t s = c->stream;
    const size_t n = (bytes + kChunk - 1) / kChunk;
    auto len = [&](size_t i) { return std::min(kChunk, bytes - i * kChunk); };
    HIP_TRY(c, hipMemcpyAsync(c->h_stage[0], src, len(0), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(c->stage_ev[0], s));
    if (while_first_chunk_travels) while_first_chunk_travels();
    for (size_t i = 0; i < n; ++i) {
        if (i + 1 < n) {
            HIP_TRY(c, hipMemcpyAsync(c->h_stage[(i + 1) & 1], (const char*)src + (i + 1) * kChunk, len(i + 1), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipEventRecord(c->stage_ev[(i + 1) & 1], s));
        }
        HIP_TRY(c, hipEventSynchronize(c->stage_ev[i & 1]));
        consume(c->h_stage[i & 1], i * kChunk, len(i));
    }
    return EMBA_OK;
}

emba_status d2h_pageable(emba_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return EMBA_OK;
    if (bytes <= ((size_t)1 << 20)) { HIP_TRY(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return EMBA_OK; }
    // (the first chunk's DMA is queued before the destination's pages are populated: the two run side by side)
    bool populated = false;
    return d2h_chunks(c, src, bytes, [&](const void* chunk, size_t off, size_t len) { copy_pool()->copy((char*)dst + off, chunk, len); },
                      [&]() { if (!populated) { populate_pages_parallel(dst, bytes); populated = true; } });
}

}  // namespace

extern "C" {

int emba_abi_version(void) { return EMBA_ABI_VERSION; }

const char* emba_build_info(void)
{
#ifdef EMBA_DIAG
    return "emba_hip: HIP/gfx950 (CDNA4, wave64) kernels, fp64, DIAGNOSTICS build (EMBA_ABLATE honoured: results may be wrong), built " __DATE__ " " __TIME__;
#else
    return "emba_hip: HIP/gfx950 (CDNA4, wave64) kernels, fp64, built " __DATE__ " " __TIME__;
#endif
}

const char* emba_last_error(const emba_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

emba_status emba_create(const emba_cfg* cfg, emba_ctx** out)
{
    if (!out) return fail(nullptr, EMBA_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!cfg || !cfg->bearing_lut) return fail(nullptr, EMBA_ERR_INVALID_ARG, "cfg or bearing_lut is NULL");
    if (cfg->sensor_w <= 0 || cfg->sensor_h <= 0 || cfg->pano_w <= 0 || cfg->pano_h <= 0 ||
        (size_t)cfg->sensor_w * cfg->sensor_h >= 0x7FFFFFFFull || (size_t)cfg->pano_w * cfg->pano_h >= 0x7FFFFFFFull)
        return fail(nullptr, EMBA_ERR_INVALID_ARG, "bad sensor/pano size");
    if (cfg->event_batch != 0 && cfg->event_batch != 100)
        return fail(nullptr, EMBA_ERR_INVALID_ARG, "event_batch must be 100 (reference hard-codes it, model.cpp:78)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, EMBA_ERR_NO_DEVICE, "no HIP device (%s); this library has no CPU path", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, EMBA_ERR_NO_DEVICE, "device ordinal %d not present (%d devices)", cfg->device, ndev);

    emba_ctx* c = new emba_ctx();
    c->cfg = *cfg;
    c->device = cfg->device;
    c->sw = cfg->sensor_w; c->sh = cfg->sensor_h; c->W = cfg->pano_w; c->H = cfg->pano_h;
    c->S = (size_t)c->sw * c->sh; c->npix = (size_t)c->W * c->H;
    c->C_th = cfg->C_th;
    c->outlier_px = cfg->outlier_px > 0 ? cfg->outlier_px : 10.0;
    // focalFromFOV(imageSize, 360, 180), equirectangular_camera.h:64-67
    c->fx = (double)((c->W / 360.0) * 180.0 / M_PI);
    c->fy = (double)((c->H / 180.0) * 180.0 / M_PI);
    c->cx = (double)c->W / 2.0; c->cy = (double)c->H / 2.0;
#ifdef EMBA_DIAG
    if (const char* ab = getenv("EMBA_ABLATE")) { c->ablate = atoi(ab); if (c->ablate) fprintf(stderr, "emba_hip: DIAGNOSTICS build, EMBA_ABLATE=%d: results are WRONG\n", c->ablate); }
#endif
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && ncu > 0) c->n_cu = ncu; }
    // (the A/B switches of rounds 1-4 — EMBA_ORDER, EMBA_STEP_FAST, EMBA_STEP_GATHER, EMBA_SEGPOSE, EMBA_GRAM_TAGS, ... — are options now: emba_set_option)

#define CREATE_TRY(call)                                                                                  \
    do {                                                                                                  \
        hipError_t e2_ = (call);                                                                          \
        if (e2_ != hipSuccess) {                                                                          \
            fail(nullptr, EMBA_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e2_));                  \
            emba_destroy(c);                                                                              \
            return EMBA_ERR_HIP;                                                                          \
        }                                                                                                 \
    } while (0)

    CREATE_TRY(hipSetDevice(c->device));
    if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
    else { CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    CREATE_TRY(hipMalloc((void**)&c->d_lut, c->S * 3 * sizeof(double))); c->caps[reinterpret_cast<void**>(&c->d_lut)] = c->S * 3 * sizeof(double);
    CREATE_TRY(hipMemcpy(c->d_lut, cfg->bearing_lut, c->S * 3 * sizeof(double), hipMemcpyHostToDevice));
    {   // the sensor's field of view from the bearing vectors (x / z, y / z extents): what schur_accumulate estimates the band of U with
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (size_t i = 0; i < c->S; ++i) {
            const double z = cfg->bearing_lut[3 * i + 2];
            if (!(z > 0.0)) continue;
            const double ax = atan(cfg->bearing_lut[3 * i] / z), ay = atan(cfg->bearing_lut[3 * i + 1] / z);
            x0 = std::min(x0, ax); x1 = std::max(x1, ax); y0 = std::min(y0, ay); y1 = std::max(y1, ay);
        }
        c->fov_x = x1 > x0 ? x1 - x0 : M_PI; c->fov_y = y1 > y0 ? y1 - y0 : M_PI;
    }
    c->cfg.bearing_lut = nullptr;  // not retained
    CREATE_TRY(hipMalloc((void**)&c->d_texel, c->npix * kTexelStride * sizeof(double))); c->caps[reinterpret_cast<void**>(&c->d_texel)] = c->npix * kTexelStride * sizeof(double);
    CREATE_TRY(hipMalloc((void**)&c->d_count_own, c->npix * sizeof(int32_t))); c->caps[reinterpret_cast<void**>(&c->d_count_own)] = c->npix * sizeof(int32_t);
    c->d_count = c->d_count_own;
    CREATE_TRY(hipMalloc((void**)&c->d_pixacc, c->npix * kPixAccStride * sizeof(double))); c->caps[reinterpret_cast<void**>(&c->d_pixacc)] = c->npix * kPixAccStride * sizeof(double);
    CREATE_TRY(hipMalloc((void**)&c->d_compact, c->npix * sizeof(int32_t))); c->caps[reinterpret_cast<void**>(&c->d_compact)] = c->npix * sizeof(int32_t);
    CREATE_TRY(hipMalloc((void**)&c->d_active_bits, (c->npix + 31) / 32 * 4 + 8)); c->caps[reinterpret_cast<void**>(&c->d_active_bits)] = (c->npix + 31) / 32 * 4 + 8;
    CREATE_TRY(hipMalloc((void**)&c->d_active, c->npix * sizeof(uint32_t))); c->caps[reinterpret_cast<void**>(&c->d_active)] = c->npix * sizeof(uint32_t);
    c->n_ablk = (c->npix + kActivePix - 1) / kActivePix;
    CREATE_TRY(hipMalloc((void**)&c->d_seg_act, c->n_ablk * kActivePix * sizeof(uint16_t))); c->caps[reinterpret_cast<void**>(&c->d_seg_act)] = c->n_ablk * kActivePix * sizeof(uint16_t);
    CREATE_TRY(hipMalloc((void**)&c->d_ablk_cnt, c->n_ablk * sizeof(uint32_t))); c->caps[reinterpret_cast<void**>(&c->d_ablk_cnt)] = c->n_ablk * sizeof(uint32_t);
    CREATE_TRY(hipMalloc((void**)&c->d_ablk_off, c->n_ablk * sizeof(uint32_t))); c->caps[reinterpret_cast<void**>(&c->d_ablk_off)] = c->n_ablk * sizeof(uint32_t);
    CREATE_TRY(hipMalloc((void**)&c->d_err2, 2 * sizeof(int))); c->caps[reinterpret_cast<void**>(&c->d_err2)] = 2 * sizeof(int);
    CREATE_TRY(hipMemset(c->d_err2, 0, 2 * sizeof(int)));
    c->d_err = c->d_err2;
    CREATE_TRY(hipMalloc((void**)&c->d_rect, 4 * sizeof(int))); c->caps[reinterpret_cast<void**>(&c->d_rect)] = 4 * sizeof(int);
    { const int init[4] = {0x7FFFFFFF, 0x7FFFFFFF, -1, -1}; CREATE_TRY(hipMemcpy(c->d_rect, init, sizeof init, hipMemcpyHostToDevice)); }
    CREATE_TRY(hipMalloc((void**)&c->d_blk_rect, c->n_ablk * 4 * sizeof(int))); c->caps[reinterpret_cast<void**>(&c->d_blk_rect)] = c->n_ablk * 4 * sizeof(int);   // per active-count block: box of the touched pixels
    CREATE_TRY(hipMalloc((void**)&c->d_total, 4 * sizeof(uint32_t))); c->caps[reinterpret_cast<void**>(&c->d_total)] = 4 * sizeof(uint32_t);   // [0] inliers [1] P [2] scratch total
    CREATE_TRY(hipMalloc((void**)&c->d_scalar, 2 * sizeof(double))); c->caps[reinterpret_cast<void**>(&c->d_scalar)] = 2 * sizeof(double);
    CREATE_TRY(hipHostMalloc((void**)&c->h_pinned, 64, hipHostMallocMapped));
    memset(c->h_pinned, 0, 64);
    CREATE_TRY(hipHostGetDevicePointer((void**)&c->h_pinned_dev, c->h_pinned, 0));
    for (int i = 0; i < 8; ++i) { CREATE_TRY(hipEventCreate(&c->ev_start[i])); CREATE_TRY(hipEventCreate(&c->ev_stop[i])); }
    // every slot of the kernel-timing events up front: creating one later can stall the calling thread for tens of milliseconds (round 4: a 38-51 ms
    // pause inside bench.py's timed loop, once per process, at the first use of a new slot — the runtime growing its signal pool)
    for (int k = 0; k < 16; ++k) for (int i = 0; i < 5; ++i) CREATE_TRY(hipEventCreate(&c->kt_sets[k][i]));
    for (int i = 0; i < 2; ++i) CREATE_TRY(hipEventCreate(&c->cal_ev[i]));
    CREATE_TRY(hipMalloc((void**)&c->d_probe, 8 * sizeof(unsigned long long))); c->caps[reinterpret_cast<void**>(&c->d_probe)] = 8 * sizeof(unsigned long long);
    CREATE_TRY(hipEventCreateWithFlags(&c->knots_copied, hipEventDisableTiming));
    CREATE_TRY(hipDeviceSynchronize());      // (the memsets above ran on the default stream, which does not order the context's non-blocking stream behind it)
#undef CREATE_TRY
    *out = c;
    return EMBA_OK;
}

void emba_destroy(emba_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_window(c);
    free_all_buffers(c);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_cost) (void)hipHostFree(c->h_cost);
    if (c->d_cost_acc) (void)hipFree(c->d_cost_acc);
    if (c->h_knots) (void)hipHostFree(c->h_knots);
    for (int k = 0; k < 2; ++k) { if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]); if (c->stage_ev[k]) (void)hipEventDestroy(c->stage_ev[k]); }
    if (c->knots_copied) (void)hipEventDestroy(c->knots_copied);
    for (int i = 0; i < 8; ++i) { if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]); if (c->ev_stop[i]) (void)hipEventDestroy(c->ev_stop[i]); }
    for (int k = 0; k < 16; ++k) for (int i = 0; i < 5; ++i) if (c->kt_sets[k][i]) (void)hipEventDestroy(c->kt_sets[k][i]);
    for (int i = 0; i < 2; ++i) if (c->cal_ev[i]) (void)hipEventDestroy(c->cal_ev[i]);
    for (auto& w : c->ws) if (w.p) (void)hipFree(w.p);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // extern "C"

namespace {

// Shared by emba_set_events (host arrays: uploaded first) and emba_set_events_dev (arrays already in HBM).  x, y, pol, hx, hy, hbt are
// DEVICE pointers; t_dev (device) or batch_t_host (the nb midpoints, computed by the caller) supplies the times.
emba_status set_events_core(emba_ctx* c, const uint16_t* x, const uint16_t* y, const uint8_t* pol, const int64_t* t_dev, const int64_t* batch_t_host,
                            size_t n, const uint16_t* hx, const uint16_t* hy, const int64_t* hbt, size_t n_halo)
{
    hipStream_t s = c->stream;
    c->last_rebin_stamp = c->rec_stamp - 3; c->n_outside_tile = 0;   // a new window may be re-binned at its first drifted evaluation
    const size_t n_used = (n / 100) * 100;   // quirk Q1: std::ceil of an integer division (model.cpp:79)
    const size_t nb = n_used / 100;
    if (n_used + n_halo >= 0x3FFFFFFFull) return fail(c, EMBA_ERR_INVALID_ARG, "too many events for 32-bit indices");
    c->n_in = n; c->n_used = n_used; c->n_halo = n_halo; c->n_batch = nb + n_halo;
    const size_t ns = n_used + n_halo;
    c->n_pm = ns; c->n_sorted = ns;
    emba_status st;
    uint32_t* d_err = nullptr;
    if ((st = ws_get(c, 19, 64, (void**)&d_err))) return st;
    HIP_TRY(c, hipMemsetAsync(d_err, 0xFF, 64, s));
    hipLaunchKernelGGL(emba_validate_events_kernel, dim3(nblocks(std::max(n_used, n_halo))), dim3(256), 0, s, x, y, t_dev, (long)n_used, c->sw, c->sh, hx, hy,
                       (long)n_halo, d_err);
    if ((st = dev_alloc(c, &c->d_batch_t, c->n_batch))) return st;
    if (t_dev) { if (nb) hipLaunchKernelGGL(emba_batch_mid_kernel, dim3(nblocks(nb)), dim3(256), 0, s, t_dev, (long)nb, c->d_batch_t); }
    else if (nb) HIP_TRY(c, hipMemcpyAsync(c->d_batch_t, batch_t_host, nb * 8, hipMemcpyHostToDevice, s));
    if (n_halo) HIP_TRY(c, hipMemcpyAsync(c->d_batch_t + nb, hbt, n_halo * 8, hipMemcpyDeviceToDevice, s));

    // pm-order: stable sort by sensor pixel == the per-pixel vectors of EventMap::addEvent (event_map.h:34-37), halo entries in front
    uint32_t *k0 = nullptr, *v0 = nullptr, *k1 = nullptr, *v1 = nullptr, *d_cf = nullptr, *d_cpos = nullptr;
    if ((st = ws_get(c, 24, std::max<size_t>(ns, 1) * 4, (void**)&k0)) || (st = ws_get(c, 25, std::max<size_t>(ns, 1) * 4, (void**)&v0)) ||
        (st = ws_get(c, 26, std::max<size_t>(ns, 1) * 4, (void**)&k1)) || (st = ws_get(c, 27, std::max<size_t>(ns, 1) * 4, (void**)&v1)) ||
        (st = ws_get(c, 21, std::max<size_t>(ns, 1) * 4, (void**)&d_cf)) || (st = ws_get(c, 22, std::max<size_t>(ns, 1) * 4, (void**)&d_cpos)))
        return st;
    if (ns) hipLaunchKernelGGL(emba_pixel_keys_kernel, dim3(nblocks(ns)), dim3(256), 0, s, x, y, (long)n_used, c->sw, hx, hy, (long)n_halo, k0, v0);
    if ((st = dev_sort(c, &k0, &v0, &k1, &v1, ns, bits_for(c->S)))) return st;
    if ((st = dev_alloc(c, &c->d_pm_pix, ns)) || (st = dev_alloc(c, &c->d_pm_batch, ns)) || (st = dev_alloc(c, &c->d_pm_orig, ns))) return st;
    if (ns) hipLaunchKernelGGL(emba_pm_gather_kernel, dim3(nblocks(ns)), dim3(256), 0, s, k0, v0, (long)ns, pol, (long)nb, c->d_pm_pix, c->d_pm_batch, c->d_pm_orig, d_cf);
    uint32_t* d_tot = d_err + 4;
    if ((st = dev_scan(c, d_cf, d_cpos, ns, d_tot))) return st;    // (only the total is used: the number of measurement candidates)
    uint32_t h_err[16];
    HIP_TRY(c, hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (h_err[0] != 0xFFFFFFFFu) return fail(c, EMBA_ERR_INVALID_ARG, "event %u lies outside the %dx%d sensor", h_err[0], c->sw, c->sh);
    if (h_err[1] != 0xFFFFFFFFu) return fail(c, EMBA_ERR_INVALID_ARG, "timestamps not sorted at event %u", h_err[1]);
    if (h_err[2] != 0xFFFFFFFFu) return fail(c, EMBA_ERR_INVALID_ARG, "halo event %u outside the sensor", h_err[2]);
    c->n_cand = ns ? h_err[4] : 0;
    const size_t n_cand = c->n_cand;

    if ((st = dev_alloc(c, &c->d_pose, c->n_batch * kPoseStride))) return st;
    bool rec_fresh = false;
    if ((st = dev_alloc(c, &c->d_rec, (std::max<size_t>(n_cand, 1) + kGramPad) * kRecStride, &rec_fresh))) return st;
    if ((st = dev_alloc(c, &c->d_slot_key, n_cand))) return st;
    {
        bool tag_fresh = false;
        if ((st = dev_alloc(c, &c->d_tag, n_cand + kGramPad, &tag_fresh))) return st;
        if (tag_fresh) HIP_TRY(c, hipMemsetAsync(c->d_tag, 0, c->caps[reinterpret_cast<void**>(&c->d_tag)], s));
    }
    c->n_fblk = (long)std::max<size_t>((ns + kFlagBlk - 1) / kFlagBlk, 1);
    if ((st = dev_alloc(c, &c->d_fblk_cnt, (size_t)c->n_fblk)) || (st = dev_alloc(c, &c->d_fblk_off, (size_t)c->n_fblk))) return st;
    c->n_fsup = (c->n_fblk + kFlagSup - 1) / kFlagSup;
    if ((st = dev_alloc(c, &c->d_fsup, (size_t)2 * c->n_fsup * kFlagSupStride))) return st;
    HIP_TRY(c, hipMemsetAsync(c->d_fsup, 0, (size_t)2 * c->n_fsup * kFlagSupStride * sizeof(uint32_t), c->stream));      // (both halves: launch A adds into one and zeroes the other for the next step)
    if ((st = dev_alloc(c, &c->d_ep, ns))) return st;
    // a record is valid iff it carries the current evaluation's stamp (record_valid): a reused buffer holds older stamps only, new memory is cleared
    if (rec_fresh) HIP_TRY(c, hipMemsetAsync(c->d_rec, 0, c->caps[reinterpret_cast<void**>(&c->d_rec)], s));
    c->eq_in_alt = false;
    if (c->d_rec2) {   // the second record set (if an LM loop has made one) follows the window's size
        bool f2 = false, t2 = false;
        if ((st = dev_alloc(c, &c->d_rec2, (std::max<size_t>(n_cand, 1) + kGramPad) * kRecStride, &f2)) || (st = dev_alloc(c, &c->d_tag2, n_cand + kGramPad, &t2))) return st;
        if (f2) HIP_TRY(c, hipMemsetAsync(c->d_rec2, 0, c->caps[reinterpret_cast<void**>(&c->d_rec2)], s));
        if (t2) HIP_TRY(c, hipMemsetAsync(c->d_tag2, 0, c->caps[reinterpret_cast<void**>(&c->d_tag2)], s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    c->nblk = (long)((ns + kWarpNew - 1) / kWarpNew);
    c->have_events = true;
    return EMBA_OK;
}

}  // namespace

extern "C" {

emba_status emba_set_events(emba_ctx* c, const uint16_t* x, const uint16_t* y, const uint8_t* pol, const int64_t* t_ns,
                            size_t n, const uint16_t* hx, const uint16_t* hy, const int64_t* hbt, size_t n_halo)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (n && (!x || !y || !pol || !t_ns)) return fail(c, EMBA_ERR_INVALID_ARG, "event arrays are NULL");
    if (n_halo && (!hx || !hy || !hbt)) return fail(c, EMBA_ERR_INVALID_ARG, "halo arrays are NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const auto t_begin = std::chrono::steady_clock::now();
    free_window(c);
    const size_t n_used = (n / 100) * 100, nb = n_used / 100;
    // The host only touches the timestamps: sortedness and the batch midpoints (model.cpp:116-119) need two reads per batch plus one
    // pass of comparisons; shipping 8 B per event over PCIe for that would cost more than the pass.
    for (size_t k = 1; k < n_used; ++k)
        if (t_ns[k] < t_ns[k - 1]) return fail(c, EMBA_ERR_INVALID_ARG, "timestamps not sorted at event %zu", k);
    std::vector<int64_t> bt(nb ? nb : 1);
    for (size_t b = 0; b < nb; ++b) bt[b] = batch_mid_ns(t_ns[100 * b], t_ns[100 * b + 99]);
    // x, y, polarity (5 B per event) and the halo go to the device as they are
    hipStream_t s = c->stream;
    uint16_t *dx = nullptr, *dy = nullptr, *dhx = nullptr, *dhy = nullptr; uint8_t* dp = nullptr; int64_t* dhb = nullptr;
    emba_status st;
    if ((st = ws_get(c, 28, std::max<size_t>(n_used, 1) * 2, (void**)&dx)) || (st = ws_get(c, 29, std::max<size_t>(n_used, 1) * 2, (void**)&dy)) ||
        (st = ws_get(c, 30, std::max<size_t>(n_used, 1), (void**)&dp)) || (st = ws_get(c, 31, std::max<size_t>(n_halo, 1) * 12, (void**)&dhb)))
        return st;
    dhx = reinterpret_cast<uint16_t*>(dhb + n_halo); dhy = dhx + n_halo;
    if (n_used) {
        HIP_TRY(c, hipMemcpyAsync(dx, x, n_used * 2, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(dy, y, n_used * 2, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(dp, pol, n_used, hipMemcpyHostToDevice, s));
    }
    if (n_halo) {
        HIP_TRY(c, hipMemcpyAsync(dhb, hbt, n_halo * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(dhx, hx, n_halo * 2, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(dhy, hy, n_halo * 2, hipMemcpyHostToDevice, s));
    }
    st = set_events_core(c, dx, dy, dp, nullptr, bt.data(), n, dhx, dhy, dhb, n_halo);
    c->set_events_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return st;
}

emba_status emba_set_events_dev(emba_ctx* c, const uint16_t* x_dev, const uint16_t* y_dev, const uint8_t* pol_dev, const int64_t* t_ns_dev, size_t n,
                                const uint16_t* hx_dev, const uint16_t* hy_dev, const int64_t* hbt_dev, size_t n_halo)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (n && (!x_dev || !y_dev || !pol_dev || !t_ns_dev)) return fail(c, EMBA_ERR_INVALID_ARG, "event arrays are NULL");
    if (n_halo && (!hx_dev || !hy_dev || !hbt_dev)) return fail(c, EMBA_ERR_INVALID_ARG, "halo arrays are NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const auto t_begin = std::chrono::steady_clock::now();
    free_window(c);
    emba_status st = set_events_core(c, x_dev, y_dev, pol_dev, t_ns_dev, nullptr, n, hx_dev, hy_dev, hbt_dev, n_halo);
    c->set_events_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return st;
}

emba_status emba_last_tile_drift(const emba_ctx* c, size_t* n_outside, int32_t* n_rebin)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (n_outside) *n_outside = c->n_outside_tile;
    if (n_rebin) *n_rebin = c->n_rebin;
    return EMBA_OK;
}

emba_status emba_last_setup_ms(const emba_ctx* c, double* set_events_ms, double* prepare_ms, int32_t* tile_order, size_t* n_entries, size_t* n_chunks)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (set_events_ms) *set_events_ms = c->set_events_ms;
    if (prepare_ms) *prepare_ms = c->prepare_ms;
    if (tile_order) *tile_order = c->tile_order ? 1 : 0;
    if (n_entries) *n_entries = c->n_sorted;
    if (n_chunks) *n_chunks = (size_t)c->n_chunks;
    return EMBA_OK;
}

emba_status emba_last_order_stats(const emba_ctx* c, double* events_per_pano_px, double* lead_in_frac)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (events_per_pano_px) *events_per_pano_px = c->order_per_px;
    if (lead_in_frac) *lead_in_frac = c->order_lead_frac;
    return EMBA_OK;
}

emba_status emba_last_order_inlier_estimate(const emba_ctx* c, double* inlier_frac)
{
    if (!c || !inlier_frac) return EMBA_ERR_INVALID_ARG;
    *inlier_frac = c->order_inl_pred;
    return EMBA_OK;
}

emba_status emba_last_tile_geometry(const emba_ctx* c, int32_t* tile_w, int32_t* tile_h, int32_t* pitch_x, int32_t* pitch_y, int32_t* reserve)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    const TileShape& ts = kTileShapes[c->tile_shape];
    const int r = std::max(0, std::min(c->opt_tile_reserve, 5));
    if (tile_w) *tile_w = ts.tw;
    if (tile_h) *tile_h = ts.th;
    if (pitch_x) *pitch_x = std::min(c->tile_fine ? ts.fine_pw : ts.pw, ts.tw - 2 * r);
    if (pitch_y) *pitch_y = std::min(c->tile_fine ? ts.fine_ph : ts.ph, ts.th - 2 * r);
    if (reserve) *reserve = r;
    return EMBA_OK;
}

emba_status emba_event_counts(const emba_ctx* c, size_t* n_used, size_t* n_cand)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (n_used) *n_used = c->n_used;
    if (n_cand) *n_cand = c->n_cand;
    return EMBA_OK;
}

emba_status emba_upload_map(emba_ctx* c, const double* Gx, const double* Gy)
{
    if (!c || !Gx || !Gy) return c ? fail(c, EMBA_ERR_INVALID_ARG, "Gx/Gy NULL") : EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->d_Gx_own) {
        emba_status st;
        if ((st = dev_alloc(c, &c->d_Gx_own, c->npix))) return st;
        if ((st = dev_alloc(c, &c->d_Gy_own, c->npix))) return st;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_Gx_own, Gx, c->npix * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_Gy_own, Gy, c->npix * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->d_Gx = c->d_Gx_cur = c->d_Gx_own; c->d_Gy = c->d_Gy_cur = c->d_Gy_own;
    c->map_is_trial = false;
    c->have_map = true;
    return EMBA_OK;
}

emba_status emba_bind_map_dev(emba_ctx* c, const double* Gx_dev, const double* Gy_dev)
{
    if (!c || !Gx_dev || !Gy_dev) return c ? fail(c, EMBA_ERR_INVALID_ARG, "Gx/Gy NULL") : EMBA_ERR_INVALID_ARG;
    c->d_Gx = c->d_Gx_cur = Gx_dev; c->d_Gy = c->d_Gy_cur = Gy_dev;
    c->map_is_trial = false;
    c->have_map = true;
    return EMBA_OK;
}

namespace {
emba_status ensure_x2(emba_ctx* c, size_t P)
{
    if (c->x2_cap < 2 * P || !c->d_x2) {
        dev_free(c, c->d_x2);
        emba_status st = dev_alloc(c, &c->d_x2, std::max<size_t>(2 * P, 2));
        if (st) return st;
        c->x2_cap = std::max<size_t>(2 * P, 2);
    }
    return EMBA_OK;
}
// src_kind: 0 host pointer, 1 device pointer, 2 the x2 the last solve left in d_x2
emba_status update_map_impl(emba_ctx* c, const double* x2, int src_kind, double damping)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->have_map) return fail(c, EMBA_ERR_STATE, "no map resident");
    if (!c->active_done && !c->P_pending) return fail(c, EMBA_ERR_STATE, "updateMap needs the active set of formNormalEq");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st = resolve_pending(c);
    if (st) return st;
    if (src_kind == 2 && c->P && c->x2_resident_P != c->P) return fail(c, EMBA_ERR_STATE, "x2 NULL: no solve of the current normal equations has left its x2 on the device");
    if (src_kind != 2 && c->P && !x2) return fail(c, EMBA_ERR_INVALID_ARG, "x2 NULL");
    if (!c->d_Gx_trial) {
        if ((st = dev_alloc(c, &c->d_Gx_trial, c->npix))) return st;
        if ((st = dev_alloc(c, &c->d_Gy_trial, c->npix))) return st;
    }
    if ((st = ensure_x2(c, c->P))) return st;
    hipStream_t s = c->stream;
    if ((st = ensure_compact(c))) return st;
    if (c->P && src_kind != 2) {
        HIP_TRY(c, hipMemcpyAsync(c->d_x2, x2, 2 * c->P * sizeof(double), src_kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
        c->x2_resident_P = (size_t)-1;      // (whatever a solve left there is overwritten)
    }
    hipLaunchKernelGGL(emba_update_map_kernel, dim3((unsigned)((c->npix + 255) / 256)), dim3(256), 0, s, c->d_Gx_cur, c->d_Gy_cur, c->d_compact,
                       c->d_x2, damping, (long)c->npix, c->d_Gx_trial, c->d_Gy_trial);
    HIP_TRY(c, hipGetLastError());
    if (src_kind == 0) HIP_TRY(c, hipStreamSynchronize(s));   // x2_host may be freed by the caller
    c->d_Gx = c->d_Gx_trial; c->d_Gy = c->d_Gy_trial;
    c->map_is_trial = true;
    return EMBA_OK;
}
// the solvers leave their x2 in d_x2 (device to device: 2P doubles), so that updateMap needs no trip through the host
emba_status keep_x2(emba_ctx* c, const double* d_src, size_t P)
{
    emba_status st = ensure_x2(c, P);
    if (st) return st;
    if (P) HIP_TRY(c, hipMemcpyAsync(c->d_x2, d_src, 2 * P * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->x2_resident_P = P;
    return EMBA_OK;
}
}  // namespace

emba_status emba_update_map(emba_ctx* c, const double* x2_host, double damping) { return update_map_impl(c, x2_host, x2_host ? 0 : 2, damping); }
emba_status emba_update_map_dev(emba_ctx* c, const double* x2_dev, double damping) { return update_map_impl(c, x2_dev, x2_dev ? 1 : 2, damping); }

emba_status emba_map_accept(emba_ctx* c)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->map_is_trial) return fail(c, EMBA_ERR_STATE, "no trial map (call emba_update_map first)");
    // the trial buffers become the current map; the previous current buffers (if ours) become the next trial buffers
    double* old_x = (c->d_Gx_cur == c->d_Gx_own) ? c->d_Gx_own : nullptr;
    double* old_y = (c->d_Gy_cur == c->d_Gy_own) ? c->d_Gy_own : nullptr;
    c->d_Gx_own = c->d_Gx_trial; c->d_Gy_own = c->d_Gy_trial;
    c->d_Gx_cur = c->d_Gx = c->d_Gx_own; c->d_Gy_cur = c->d_Gy = c->d_Gy_own;
    c->d_Gx_trial = old_x; c->d_Gy_trial = old_y;
    c->map_is_trial = false;
    return EMBA_OK;
}

emba_status emba_trial_reject(emba_ctx* c)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->eq_in_alt) return EMBA_OK;     // nothing was evaluated since the equations were formed (or they have been re-formed)
    HIP_TRY(c, hipSetDevice(c->device));
    // (a trial that was only launched: its compacted residual vector and inlier numbers will never be asked for — nothing to resolve)
    if (c->ep_deferred && !c->P_pending && !c->inl_pending) c->ep_deferred = false;
    else if (c->P_pending || c->inl_pending || c->ep_deferred) { emba_status st = resolve_pending(c); if (st) return st; }
    std::swap(c->d_rec, c->d_rec2); std::swap(c->d_tag, c->d_tag2); std::swap(c->set_stamp, c->set_stamp2);
    std::swap(c->caps[reinterpret_cast<void**>(&c->d_rec)], c->caps[reinterpret_cast<void**>(&c->d_rec2)]);
    std::swap(c->caps[reinterpret_cast<void**>(&c->d_tag)], c->caps[reinterpret_cast<void**>(&c->d_tag2)]);
    c->eq_in_alt = false;
    c->active_done = c->eq_saved.active_done; c->accum_done = c->eq_saved.accum_done; c->finish_done = c->eq_saved.finish_done;
    c->compact_valid = c->eq_saved.compact_valid; c->l2_fused = c->eq_saved.l2_fused; c->P = c->eq_saved.P; c->pack_len = c->eq_saved.pack_len;
    c->K = c->eq_saved.K; c->thres = c->eq_saved.thres; c->irls = c->eq_saved.irls; c->eta = c->eq_saved.eta;
    // the per-event residuals, the count map and the per-pixel sums are the rejected trial's: formNormalEq needs a new evaluation
    c->eval_launched = c->eval_done = false; c->inl_pending = c->P_pending = false; c->ep_deferred = false; c->inl_idx_valid = false;
    c->ep_valid = false; c->ep_in_gram = false;
    return EMBA_OK;
}

emba_status emba_map_reject(emba_ctx* c)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->map_is_trial) return fail(c, EMBA_ERR_STATE, "no trial map (call emba_update_map first)");
    c->d_Gx = c->d_Gx_cur; c->d_Gy = c->d_Gy_cur;
    c->map_is_trial = false;
    return emba_trial_reject(c);           // and the normal equations the trial evaluation set aside are current again
}

emba_status emba_download_map(emba_ctx* c, double* Gx_host, double* Gy_host)
{
    if (!c || !Gx_host || !Gy_host) return c ? fail(c, EMBA_ERR_INVALID_ARG, "Gx/Gy NULL") : EMBA_ERR_INVALID_ARG;
    if (!c->have_map) return fail(c, EMBA_ERR_STATE, "no map resident");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(Gx_host, c->d_Gx, c->npix * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(Gy_host, c->d_Gy, c->npix * sizeof(double), hipMemcpyDeviceToHost));
    return EMBA_OK;
}

emba_status emba_get_map_active(emba_ctx* c, double* gxy_host, size_t cap_P)
{
    if (!c || !gxy_host) return c ? fail(c, EMBA_ERR_INVALID_ARG, "gxy_host NULL") : EMBA_ERR_INVALID_ARG;
    if (!c->have_map) return fail(c, EMBA_ERR_STATE, "no map resident");
    if (!c->active_done && !c->P_pending) return fail(c, EMBA_ERR_STATE, "no active set (formNormalEq) yet");
    // (only a trial map — emba_update_map's output — is zero outside the active set it was built from; an uploaded or accepted map is not)
    if (!c->map_is_trial) return fail(c, EMBA_ERR_STATE, "emba_get_map_active returns the trial map emba_update_map built; no trial map is resident");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st = resolve_pending(c);
    if (st) return st;
    if (cap_P < c->P) return fail(c, EMBA_ERR_CAPACITY, "cap_P=%zu < P=%zu", cap_P, c->P);
    if (!c->P) return EMBA_OK;
    double* d_out = nullptr;
    if ((st = ws_get(c, 15, 2 * c->P * sizeof(double), (void**)&d_out))) return st;
    hipLaunchKernelGGL(emba_map_active_kernel, dim3(nblocks(c->P)), dim3(256), 0, c->stream, c->d_Gx, c->d_Gy, c->d_active, (long)c->P, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(gxy_host, d_out, 2 * c->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EMBA_OK;
}

emba_status emba_bind_exchange_buffers(emba_ctx* c, int32_t* count_map_dev, double* pack_dev, size_t pack_cap)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    c->d_count = count_map_dev ? count_map_dev : c->d_count_own;
    c->count_stamp = 0;
    c->pix_dirty_all = true;   // the new count buffer says nothing about which pixacc lines are dirty
    c->pixacc_clean = false;
    if (pack_dev) { c->d_pack = pack_dev; c->pack_cap = pack_cap; c->pack_bound = true; }
    else { c->pack_bound = false; c->d_pack = c->d_pack_own; c->pack_cap = c->pack_own_cap; }
    return EMBA_OK;
}

emba_status emba_count_compress(emba_ctx* c, uint8_t* u8_dev, int32_t cap)
{
    if (!c || !u8_dev || cap < 1 || cap > 255) return c ? fail(c, EMBA_ERR_INVALID_ARG, "count_compress: bad arguments") : EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched) return fail(c, EMBA_ERR_STATE, "emba_eval_launch has not been called");
    if (c->counts_raw) {      // markers -> this rank's counts AND their saturated bytes in one sweep (ensure_counts + the compression below)
        c->counts_raw = false;
        c->count_stamp = (c->d_count == c->d_count_own) ? c->set_stamp : 0u;
        hipLaunchKernelGGL(emba_count_materialise_compress_kernel, dim3((unsigned)((c->npix + 2047) / 2048)), dim3(256), 0, c->stream, c->d_count, c->d_pixacc, (long)c->npix,
                           c->count_mark, (int)cap, u8_dev);
    } else {
        hipLaunchKernelGGL(emba_count_compress_kernel, dim3((unsigned)((c->npix + 1023) / 1024)), dim3(256), 0, c->stream, c->d_count, (long)c->npix, (int)cap, u8_dev);
    }
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

emba_status emba_count_expand(emba_ctx* c, const uint8_t* u8_dev)
{
    if (!c || !u8_dev) return c ? fail(c, EMBA_ERR_INVALID_ARG, "count_expand: NULL") : EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched) return fail(c, EMBA_ERR_STATE, "emba_eval_launch has not been called");
    c->count_stamp = 0;      // (exchanged counts: no longer this context's own)
    hipLaunchKernelGGL(emba_count_expand_kernel, dim3((unsigned)((c->npix + 1023) / 1024)), dim3(256), 0, c->stream, u8_dev, (long)c->npix, c->d_count);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

emba_status emba_eval_launch(emba_ctx* c, const double* knots, int32_t K, int64_t t0_ns, int64_t dt_ns)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!knots) return fail(c, EMBA_ERR_INVALID_ARG, "knots NULL");
    if (!c->have_events) return fail(c, EMBA_ERR_STATE, "emba_set_events has not been called");
    if (!c->have_map) return fail(c, EMBA_ERR_STATE, "no map: call emba_upload_map or emba_bind_map_dev");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st;
    if (c->knots_cap < K) {
        dev_free(c, c->d_knots);
        if ((st = dev_alloc(c, &c->d_knots, (size_t)4 * K))) return st;
        c->knots_cap = K;
    }
    if (c->seg_cap < K) {
        dev_free(c, c->d_seg);
        if ((st = dev_alloc(c, &c->d_seg, (size_t)kSegStride * K))) return st;
        c->seg_cap = K;
    }
    if ((st = prepare_order(c, knots, t0_ns, dt_ns, K))) return st;
    if ((st = ensure_pack(c, K))) return st;
    if (c->h_knots_cap < K) {
        if (c->h_knots) (void)hipHostFree(c->h_knots);
        HIP_TRY(c, hipHostMalloc((void**)&c->h_knots, (size_t)4 * K * sizeof(double), hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->h_knots_dev, c->h_knots, 0));
        c->h_knots_cap = K;
    }
    hipStream_t s = c->stream;
    if (c->accum_done && !c->no_alt_set) {
        // The working record set is what the current normal equations were formed from: this evaluation (an LM trial, or simply the next
        // step) writes the OTHER set, so that a rejection can go back to untouched equations (emba_trial_reject).
        if (c->P_pending || c->inl_pending) { if ((st = resolve_pending(c, true))) return st; }
        if (!c->d_rec2) {
            const size_t nrec = (std::max<size_t>(c->n_cand, 1) + kGramPad) * kRecStride, ntag = c->n_cand + kGramPad;
            if ((st = dev_alloc(c, &c->d_rec2, nrec)) || (st = dev_alloc(c, &c->d_tag2, ntag))) return st;
            HIP_TRY(c, hipMemsetAsync(c->d_rec2, 0, nrec * sizeof(double), s));     // (new memory: no record may look valid)
            HIP_TRY(c, hipMemsetAsync(c->d_tag2, 0, ntag * sizeof(double), s));
        }
        c->eq_saved.active_done = c->active_done; c->eq_saved.accum_done = c->accum_done; c->eq_saved.finish_done = c->finish_done;
        c->eq_saved.compact_valid = c->compact_valid; c->eq_saved.l2_fused = c->l2_fused; c->eq_saved.P = c->P; c->eq_saved.pack_len = c->pack_len;
        c->eq_saved.K = c->K; c->eq_saved.thres = c->thres; c->eq_saved.irls = c->irls; c->eq_saved.eta = c->eta;
        std::swap(c->d_rec, c->d_rec2); std::swap(c->d_tag, c->d_tag2); std::swap(c->set_stamp, c->set_stamp2);
        std::swap(c->caps[reinterpret_cast<void**>(&c->d_rec)], c->caps[reinterpret_cast<void**>(&c->d_rec2)]);
        std::swap(c->caps[reinterpret_cast<void**>(&c->d_tag)], c->caps[reinterpret_cast<void**>(&c->d_tag2)]);
        c->eq_in_alt = true;
    }
    c->K = K;
    c->count_stamp = 0;      // (the warp kernels are about to mark the count map for a new evaluation)
    c->eval_launched = c->eval_done = c->active_done = c->accum_done = false;
    c->inl_pending = c->P_pending = false; c->ep_deferred = false; c->inl_idx_valid = false;
    c->ep_valid = false; c->ep_in_gram = false; c->ep_after_gram = false;
    if (c->pix_dirty_all) {   // first use of these buffers: num_ev_map.setTo(0), model.cpp:85 (+ every per-pixel accumulator line)
        HIP_TRY(c, hipMemsetAsync(c->d_count, 0, c->npix * sizeof(int32_t), s));
        HIP_TRY(c, hipMemsetAsync(c->d_pixacc, 0, c->npix * kPixAccStride * sizeof(double), s));
        c->pix_dirty_all = false;
        c->pixacc_clean = true;
    }
    int* rect_cur = c->d_rect;
    // the clearing pass over count map + accumulator lines is only needed when the previous evaluation's sums are still in their lines (it was
    // never formed by the resident step, whose gather clears them behind itself); the count map's entries are stamped and need no clearing
    // (... and only when a warp kernel follows to re-mark it — an EMPTY window launches none, so its count map and active set would be the
    // previous window's; the reference clears, model.cpp:85, and finds P = 0)
    const int n_prep_blk = (c->pixacc_clean && c->n_sorted) ? 0 : (int)((c->npix + 1023) / 1024);
    // Hessian source: with several events per panorama pixel (measured break-even: ~4) the full texel pack (one 48-B gather per
    // measurement instead of an 18-load stencil) pays for itself; otherwise texels are packed only inside the bounding box of the pixels an
    // earlier evaluation touched, and the warp kernel falls back to the stencil outside it.
    c->use_texel = c->texel_mode == 1 ? 1 : c->texel_mode == 2 ? 0 : c->texel_mode == 3 ? 3 : (c->n_sorted > 4 * c->npix ? 1 : 3);
    {   // ONE launch in front of the warp kernel: prep || pose table (or segment records) || texel rectangle — independent of each other
        PrepPoseTexelParams q{};
        InlineKnots kn;
        const int nb = (int)c->n_batch;
        ++c->eval_seq;
        c->d_err = c->d_err2 + (c->eval_seq & 1u);
        q.count = c->d_count; q.npix = (long)c->npix; q.pixacc = c->d_pixacc; q.W = c->W; q.H = c->H; q.n_prep = n_prep_blk;
        q.batch_t_ns = c->d_batch_t; q.nb = nb; q.K = (int)K; q.t0_ns = t0_ns; q.dt_ns = dt_ns; q.pose = c->d_pose; q.err = c->d_err;
        q.err_next = c->d_err2 + ((c->eval_seq + 1u) & 1u);
        // pixel order: per-event pose from the segment records too (warp_lane SEGPOSE; round 4, late) instead of one 112-B pose record per batch gathered by
        // every event — 7 x 16-B gathers over 64 different lines per wave, and past ~7 M events a table that no longer fits the L2s.  Measured, same box,
        // step time: 1 M events 101.2 -> 98.2 us, 1.5 M 150.8 -> 143.5, 10 M on 640x480 (city shape) 779 -> 714 (warp 0.60 -> 0.68 of the roofline),
        // 10 M on 2048x4096 / K = 256 940 -> 861.  (Round 1 had measured the opposite at 1 M events, + 6 us, on a kernel that was then VALU-heavier in
        // other places; option segpose = 1 keeps the per-batch table for comparison.)
        c->segpose = !c->tile_order && (c->segpose_mode ? c->segpose_mode == 2 : true);
        const bool seg_records = c->tile_order || c->segpose;
        q.n_pose = ((seg_records ? (int)K - 1 : nb) + 63) / 64;   // (K-1 segment records instead of nb batch poses)
        q.seg = seg_records ? c->d_seg : nullptr;
        q.knots_dev = c->d_knots; q.knots_out = c->d_knots;
        q.inline_knots = (K <= kInlineKnots) ? 1 : 0;
        if (q.inline_knots) memcpy(kn.q, knots, (size_t)4 * K * sizeof(double));       // by value in the kernel arguments: no staging copy at all
        else {
            if (c->knots_in_flight) HIP_TRY(c, hipStreamSynchronize(s));               // the previous copy must have consumed the pinned staging buffer
            memcpy(c->h_knots, knots, (size_t)4 * K * sizeof(double));
            HIP_TRY(c, hipMemcpyAsync(c->d_knots, c->h_knots, (size_t)4 * K * sizeof(double), hipMemcpyHostToDevice, s));
            c->knots_in_flight = true;   // cleared by the next host synchronisation
        }
        q.n_tex = (c->use_texel == 3) ? 1024 : 0;
        q.Gx = c->d_Gx; q.Gy = c->d_Gy; q.rect = rect_cur; q.texel = c->d_texel;
        if (q.n_pose + q.n_tex + q.n_prep == 0) q.n_prep = 1;   // (an empty window on clean lines: block 0 still clears the next status word)
        if (c->kernel_timing && c->kt_all) { HIP_TRY(c, hipEventRecord(c->kt[4], s)); c->kt_valid[c->kt_slot][2] = true; }
        hipLaunchKernelGGL(emba_prep_pose_texel_kernel, dim3((unsigned)(q.n_pose + q.n_tex + q.n_prep)), dim3(256), 0, s, q, kn);
    }
    if (c->use_texel == 1)
        hipLaunchKernelGGL(emba_texel_kernel, dim3((c->W + 255) / 256, c->H), dim3(256), 0, s, c->d_Gx, c->d_Gy, c->H, c->W,
                           c->d_texel);
    if (c->n_sorted) {
        WarpParams p{};
        p.ev_pix = c->d_ev_pix; p.ev_batch = c->d_ev_batch; p.ev_slot = c->d_ev_slot; p.ev_pm = c->tile_order ? c->d_ev_pm : nullptr; p.n_sorted = (long)c->n_sorted;
        p.ev_u = c->d_ev_u; p.ev_seg = c->d_ev_seg;      // per entry, in both orders
        p.nblk = c->nblk; p.pose = c->d_pose; p.seg = c->d_seg; p.lut = c->d_lut; p.texel = c->use_texel ? c->d_texel : nullptr; p.W = c->W; p.H = c->H;
        p.rect_acc = (c->use_texel == 3) ? rect_cur : nullptr;
        p.Gx = c->d_Gx; p.Gy = c->d_Gy; p.pixacc = c->d_pixacc;
        p.fx = c->fx; p.fy = c->fy; p.cx = c->cx; p.cy = c->cy; p.C_th = c->C_th; p.outlier_px = c->outlier_px;
        p.count = c->d_count; p.rec = c->d_rec; p.tag = (c->use_tags && !c->tile_order) ? c->d_tag : nullptr; p.e_sorted = c->d_e_sorted; p.flag = c->d_flag;
        p.err = c->d_err;
        p.ablate = c->ablate;
        p.rec_nt = (c->tile_order || c->n_cand * (size_t)(kRecStride * 8) > ((size_t)144 << 20)) ? 1 : 0;      // (1 M slots = 128 MB: kept in the Infinity Cache for the Gram kernel)
        p.irls = c->cost_irls; p.eta = c->cost_eta;
        p.stamp = ++c->rec_stamp; c->set_stamp = p.stamp;
        p.marker = c->count_mark = count_marker(p.stamp);
        c->pixacc_clean = false; c->pixacc_consumed = false;
        p.chunks = c->d_chunks; p.n_chunks = c->n_chunks; p.chunks_linear = c->chunks_lpt ? 1 : 0;
        if (c->kernel_timing) HIP_TRY(c, hipEventRecord(c->kt[0], s));
        if (c->tile_order) launch_warp_tiled(c->tile_shape, dim3((unsigned)grid8(c->n_chunks)), s, p);
        else if (c->segpose) hipLaunchKernelGGL((emba_warp_residual_kernel<false, false, true>), dim3((unsigned)grid8(c->nblk)), dim3(kWarpBlock), 0, s, p);
        else hipLaunchKernelGGL(emba_warp_residual_kernel<false>, dim3((unsigned)grid8(c->nblk)), dim3(kWarpBlock), 0, s, p);
        if (c->kernel_timing) { HIP_TRY(c, hipEventRecord(c->kt[1], s)); c->kt_warp_valid = true; c->kt_valid[c->kt_slot][0] = true; }
        c->counts_raw = true;
    } else {
        c->counts_raw = false;
    }
    HIP_TRY(c, hipGetLastError());
    c->acc_irls = c->cost_irls; c->acc_eta = c->cost_eta;
    c->eval_launched = true;
    return EMBA_OK;
}

emba_status emba_eval_finish(emba_ctx* c, double* ep_out, size_t* n_inliers, int32_t* num_ev_map_out)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched) return fail(c, EMBA_ERR_STATE, "emba_eval_launch has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    c->ep_deferred = true;
    if (!ep_out && !n_inliers && !num_ev_map_out) return EMBA_OK;   // fully asynchronous: the compaction rides with the next phase
    emba_status st = resolve_pending(c);
    if (st) return st;
    if (n_inliers) *n_inliers = c->n_inliers;
    if (ep_out && c->n_inliers) { HIP_TRY(c, hipStreamSynchronize(c->stream)); if ((st = d2h_pageable(c, ep_out, c->d_ep, c->n_inliers * sizeof(double)))) return st; }
    if (num_ev_map_out) {
        { emba_status st0 = ensure_counts(c); if (st0) return st0; }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if ((st = d2h_pageable(c, num_ev_map_out, c->d_count, c->npix * sizeof(int32_t)))) return st;
    }
    return EMBA_OK;
}

emba_status emba_eval_data_error(emba_ctx* c, const double* knots, int32_t K, int64_t t0_ns, int64_t dt_ns, const double* Gx,
                                 const double* Gy, int32_t eval_deriv, double* ep_out, size_t* n_inliers,
                                 int32_t* num_ev_map_out)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!eval_deriv) return fail(c, EMBA_ERR_INVALID_ARG, "eval_deriv=false is never used by the reference (solver.cpp:75,251) and is not provided");
    emba_status st;
    if (Gx || Gy) { if ((st = emba_upload_map(c, Gx, Gy))) return st; }   // both NULL: evaluate on the resident (current or trial) map
    if ((st = emba_eval_launch(c, knots, K, t0_ns, dt_ns))) return st;
    return emba_eval_finish(c, ep_out, n_inliers, num_ev_map_out);
}

emba_status emba_form_active(emba_ctx* c, int32_t thres, size_t* P, size_t* pack_len)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->eval_done && !c->inl_pending && !c->ep_deferred) return fail(c, EMBA_ERR_STATE, "formNormalEq needs the state of evaluateDataError (solver.cpp:99-102)");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const long npix = (long)c->npix;
    const long head = (long)9 * c->K * c->K + (long)3 * c->K;
    ActiveWriteParams aw{};
    aw.count = c->d_count; aw.npix = npix; aw.thres = thres; aw.blk_off = c->d_ablk_off; aw.compact = nullptr; aw.active_idx = c->d_active; aw.pixacc = c->d_pixacc;
    aw.A22b2 = pack_A22b2(c); aw.pack_head = c->d_pack; aw.head_len = head; aw.alpha = c->fused_alpha; aw.Gx = c->d_Gx; aw.Gy = c->d_Gy;
    aw.active_bits = c->d_active_bits; aw.max_P = (long)((c->pack_cap - (size_t)head) / 5); aw.n_ablk = (long)c->n_ablk; aw.ablate = c->ablate;
    // the resident step cleared this evaluation's per-pixel sums behind its gather: a second formNormalEq on the same evaluation takes A22 | b2
    // from the records (the generic path of emba_form_accumulate), and its L2 term from emba_form_finish
    c->force_generic_a22 = c->pixacc_consumed;
    if (c->force_generic_a22) { aw.A22b2 = nullptr; aw.alpha = 0.0; c->fused_alpha = 0.0; }
    // a sharded window's rank in resident-step mode (emba_step_form_active): activity from the exchanged byte counts, which launch A reads beside this rank's
    // own counts — where the list-driven gather cannot be used the bytes are expanded into the count map first and the sweeping forms below see global counts
    const bool lists_ok = c->step_consume && c->step_gather && !c->force_generic_a22 && (long)c->n_ablk <= kGatherMaxUnits && c->n_cand &&
                          (c->step_gather == 3 || !c->tile_order || c->n_cand <= 3500000);
    if (c->global_u8 && !(lists_ok && c->ep_deferred && c->n_sorted && !c->counts_raw)) {
        { emba_status st0 = ensure_counts(c); if (st0) return st0; }
        c->count_stamp = 0;
        hipLaunchKernelGGL(emba_count_expand_kernel, dim3((unsigned)((c->npix + 1023) / 1024)), dim3(256), 0, s, c->global_u8, npix, c->d_count);
        c->global_u8 = nullptr;
    }
    if (c->ep_deferred && c->n_sorted) {
        c->ep_deferred = false;
        PostWarpParams q{};
        q.count = c->d_count; q.npix = npix; q.thres = thres; q.ablk_cnt = c->d_ablk_cnt; q.ablk_off = c->d_ablk_off; q.n_ablk = (long)c->n_ablk;
        q.total_P = c->d_total + 1; q.total_P_host = c->h_pinned_dev + 2;
        q.fblk_cnt = c->d_fblk_cnt; q.fblk_off = c->d_fblk_off; q.n_fblk = c->n_fblk; q.perm = nullptr; q.n_pm = (long)c->n_pm;
        q.total_inl = c->d_total; q.total_inl_host = c->h_pinned_dev;
        q.err_dev = c->d_err; q.err_host = c->h_pinned_dev + 1;
        q.e_sorted = c->d_e_sorted; q.flag = c->d_flag; q.ep = c->d_ep; q.inl_idx = nullptr;   // (inlier numbers: on demand, ensure_inl_idx)
        q.seq = ++c->seq; q.seq_host = c->h_pinned_dev + 3; c->seq_armed = true;
        if (c->counts_raw) { q.raw_count = c->d_count; q.pixacc = c->d_pixacc; q.marker = c->count_mark; c->counts_raw = false;      // launch A turns the markers into counts
                             c->count_stamp = (c->d_count == c->d_count_own) ? c->set_stamp : 0u; }
        const bool consume = c->step_consume && c->step_fast && !c->force_generic_a22;    // this gather is the per-pixel sums' only reader: lines are zeroed behind it
        // list-driven gather: it rides in the head of the Gram kernel (as a kernel of its own it is no faster than the sweeping write: 109.6 vs
        // 108.5 us per step at 1 M events; option step_gather = 1 forces that form for comparison, 0 the sweeping kernel)
        // Where it paid as a HEAD in front of the stream (same box, step time with the head vs with the sweeping kernel): 1 M events 101.2 vs 105.7 us,
        // the 1 M-event shard of the 8 M-event stream 121.8 vs 127, scene-driven 1.17 M events 121 vs 135 — but 1.5 M events 161 vs 154, 3 M (tile order)
        // 289 vs 281, 10 M 833 vs 817: with more active pixels per block the head's dependent trips grew past what the launch saved.
        q.global_u8 = c->global_u8;
        const bool lists = lists_ok && (q.raw_count || q.global_u8);      // (step_gather = 3: everywhere, for comparison)
        // (round 4, late: the gather is now the work of 4 of a Gram block's 16 waves BESIDE the record stream, not a head in front of it: step time
        // with it / with the sweeping launch — 1 M 92.6 / 96.4 us, 1.5 M 134.7 / 143.8, 2 M (tile order) 193.4 / 208.5, 3 M 274.4 / 280.9, 10 M on
        // 640x480 (pixel order) 612.5 / 626.1, on 2048x4096 766.7 / 776.3; but 5 M (tile) 436.0 / 432.2, 40 M 3137 / 3014: a bandwidth-bound stream
        // misses the four waves more than it gains from the launch — pixel order everywhere, tile order up to 3.5 M candidates)
        if (consume) { aw.clear_pixacc = c->d_pixacc; c->pixacc_clean = true; c->pixacc_consumed = true; }
        if (lists) { q.seg = c->d_seg_act; aw.seg = c->d_seg_act; if (consume) q.clear_inactive = c->d_pixacc; }
        // launch A: {active counts (+ markers -> counts, activity bits, cleared A11 | b1) || inlier-flag counts}; launch B: the active-set write,
        // whose blocks take their own prefix over launch A's per-block counts and whose last block publishes P, the inlier total, the status
        // word and the sequence words the host polls.  (Nothing on the device reads the compacted residual vector `ep` — costs, Gram and solvers
        // work from the records and the per-event residuals — so it is produced when the host asks for it: resolve_pending / ensure_inl_idx
        // run the standalone compaction from the per-block flag counts left here.  100 M events: 0.65 -> 0.2 ms.)
        // Tried and dropped (round 3, 1 M events): both launches as ONE kernel with the per-block counts published through flags (look-back,
        // and "sum every predecessor"): 38-270 us against 6.3 + 11.2 — the eight XCDs' L2s are not coherent with each other, so every flag is a
        // round trip to the memory side (and a release / acquire pair writes back / invalidates a whole L2); a kernel boundary is cheaper.
        c->fsup_half ^= 1;
        q.fsup = c->d_fsup + (size_t)c->fsup_half * c->n_fsup * kFlagSupStride; q.fsup_next = c->d_fsup + (size_t)(c->fsup_half ^ 1) * c->n_fsup * kFlagSupStride; q.n_sup = c->n_fsup;
        q.active_bits = c->d_active_bits; q.pack_head = c->d_pack; q.head_len = head; aw.bits_head_done = 1;
        q.blk_rect = c->d_blk_rect; q.W = c->W; aw.blk_rect = c->d_blk_rect; aw.rect_out = c->d_rect;   // the texel rectangle of the NEXT evaluation
        hipLaunchKernelGGL(emba_post_warp_a_kernel, dim3((unsigned)(c->n_ablk + c->n_fblk)), dim3(256), 0, s, q);
        aw.blk_cnt = c->d_ablk_cnt; aw.fblk_cnt = c->d_fblk_cnt; aw.n_fblk = c->n_fblk; aw.total_P = q.total_P; aw.total_P_host = q.total_P_host;
        aw.total_inl = q.total_inl; aw.total_inl_host = q.total_inl_host; aw.err_dev = q.err_dev; aw.err_host = q.err_host; aw.seq = q.seq; aw.seq_host = q.seq_host;
        // (round 4, measured and dropped: this write on a side stream beside the Gram kernel — both only depend on launch A — costs more than it
        // hides: each cross-stream event edge opens a 7-12 us bubble on this stack, 114.5 vs 107.6 us per step)
        // The resident one-GPU step (lists): the write is list-driven and balanced (active_gather_block) and rides in the head of the compact Gram
        // kernel — emba_form_accumulate issues it — or runs as a kernel of its own (step_gather = 1, or where the Gram kernel is another form)
        c->aw_in_gram = false;
        // the residual vector ep of this evaluation: compacted by tail blocks of the Gram launch that follows (kernels.h: ep_tail_block) — launch A has just
        // left the per-block inlier-flag counts they need
        c->ep_in_gram = c->step_wants_ep && c->step_ep != 2 && c->n_cand;
        c->ep_after_gram = c->step_wants_ep && !c->ep_in_gram;
        if (lists && c->step_gather >= 2) { c->aw_saved = aw; c->aw_in_gram = true; }
        else if (lists) hipLaunchKernelGGL(emba_active_gather_kernel, dim3(1024), dim3(256), 0, s, aw);
        else hipLaunchKernelGGL(emba_active_write_kernel, dim3((unsigned)c->n_ablk), dim3(256), 0, s, aw);
        c->inl_pending = true;
    } else {
        { emba_status st0 = launch_ep_compaction(c); if (st0) return st0; }
        { emba_status st0 = ensure_counts(c); if (st0) return st0; }
        hipLaunchKernelGGL(emba_active_count_kernel, dim3((unsigned)c->n_ablk), dim3(256), 0, s, c->d_count, npix, (int)thres, c->d_ablk_cnt);
        hipLaunchKernelGGL(emba_scan_kernel, dim3(1), dim3(256), 0, s, c->d_ablk_cnt, c->d_ablk_off, (long)c->n_ablk, c->d_total + 1,
                           c->h_pinned_dev + 2, (const int*)nullptr, (int*)nullptr);
        hipLaunchKernelGGL(emba_active_write_kernel, dim3((unsigned)c->n_ablk), dim3(256), 0, s, aw);
    }
    HIP_TRY(c, hipGetLastError());
    c->compact_valid = false;
    c->l2_fused = (c->fused_alpha != 0.0);
    c->fused_alpha = 0.0;
    c->thres = thres;
    c->eq_in_alt = false;   // new equations are being formed from the working set: the other set's are obsolete
    c->P_pending = true; c->active_done = false; c->accum_done = false;
    c->finish_done = false;          // (the head of the pack has just been cleared: what a solve would read is no set of equations — found by the call-order pair test)
    c->x2_resident_P = (size_t)-1;   // (a solve of the PREVIOUS equations may have left its x2 on the device)
    c->lists_valid = false; c->perm_valid = false;          // (new active set)
    if (!P && !pack_len) return EMBA_OK;   // asynchronous: P is read from device memory by the kernels that need it
    emba_status st = resolve_pending(c, true);   // counts only: a sharded host sizes exchange 2 from P while the gather still runs
    if (st) return st;
    if (P) *P = c->P;
    if (pack_len) *pack_len = c->pack_len;
    return EMBA_OK;
}

emba_status emba_form_accumulate(emba_ctx* c, const double* ep_host, int32_t irls, double eta)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->active_done && !c->P_pending) return fail(c, EMBA_ERR_STATE, "emba_form_active has not been called");
    if (c->accum_done) return fail(c, EMBA_ERR_STATE, "these equations have been accumulated already: A11 | b1 are cleared by emba_form_active only (a second pass would add the sums again)");
    if (irls < 0 || irls > 2) return fail(c, EMBA_ERR_INVALID_ARG, "irls must be 0 (quadratic), 1 (huber) or 2 (cauchy)");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // the per-pixel sums of the evaluation already carry this cost's weights (emba_set_cost / emba_step)?  Then they ARE A22/b2.
    const bool acc_matches = (irls == c->acc_irls) && (irls == 0 || eta == c->acc_eta);
    const bool generic_a22 = !acc_matches || (ep_host != nullptr) || c->force_generic_a22;
    if (c->aw_in_gram && (generic_a22 || !c->n_cand)) {   // (not what emba_step does: the deferred gather as a launch of its own after all)
        hipLaunchKernelGGL(emba_active_gather_kernel, dim3(1024), dim3(256), 0, s, c->aw_saved);
        c->aw_in_gram = false;
    }
    if (generic_a22) { emba_status st = resolve_pending(c); if (st) return st; }   // needs n_inliers / P on the host (rare path)
    if (ep_host && c->n_inliers) {
        { emba_status st = ensure_inl_idx(c); if (st) return st; }
        HIP_TRY(c, hipMemcpyAsync(c->d_ep, ep_host, c->n_inliers * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(emba_override_ep_kernel, dim3((unsigned)((c->n_sorted + 255) / 256)), dim3(256), 0, s, c->d_ep, c->d_flag,
                           c->d_inl_idx, c->d_ev_slot, c->d_ev_pix, c->tile_order ? c->d_ev_pm : nullptr, (long)c->n_sorted, c->d_rec, c->d_e_sorted);
    }
    // A11 = Zero, b1 = Zero (model.cpp:357-361).  A22/b2 of the active pixels were gathered from the per-pixel
    // accumulator by emba_form_active (quadratic cost, device-resident residuals); with IRLS weights or a
    // caller-supplied ep they are rebuilt from the records instead.
    // (the head of the pack, A11 | b1, was zeroed by emba_form_active's write kernel)
    c->irls = irls; c->eta = eta;
    if (generic_a22 && c->P) {
        { emba_status st = ensure_compact(c); if (st) return st; }
        HIP_TRY(c, hipMemsetAsync(pack_A22b2(c), 0, 5 * c->P * sizeof(double), s));
        if (c->n_cand)
            hipLaunchKernelGGL(emba_a22_from_records_kernel, dim3((unsigned)((c->n_cand + 255) / 256)), dim3(256), 0, s, c->d_rec,
                               (long)c->n_cand, c->d_count, c->d_compact, c->thres, irls, eta, pack_A22b2(c), c->set_stamp);
    }
    if (c->n_cand) {
        GramParams p{};
        p.rec = c->d_rec; p.slot_key = c->d_slot_key; p.n_slots = (long)c->n_cand; p.active_bits = reinterpret_cast<const uint32_t*>(c->d_active_bits);
        p.irls = irls; p.eta = eta; p.stamp = c->set_stamp; p.A11 = pack_A11(c); p.b1 = pack_b1(c);
        // the tag stream pays where slots are dead (pixel order: about half of them at the BASELINE workload); in the tile order (dense regime:
        // nearly every slot is live) the warp kernel's scattered 8-B tag stores cost more than the Gram kernel saves (40 M events: +370 vs -180 us)
        p.tag = gram_uses_tags(c, ep_host != nullptr) ? c->d_tag : nullptr;
        p.dim = 3 * c->K;
        p.ablate = c->ablate;
        // slots per wave: whole rounds of one 16-wave block per CU with equal shares (1 M events: one round of 236 slots per wave),
        // between kGramChunkMin and kGramChunk slots
        // Sparse slot streams (pixel order on a large panorama: most inliers fall on pixels that stay inactive): stages of 128 tags, only the live records fetched (180 -> 78 us at 10 M events on 2048 x 4096, K = 256).
        // Which form: option gram_sparse (0 / 1), else by the LAST formed equations of this context — at least thres records per active pixel are live, and where
        // 4 thres P is still below a quarter of the slots the stream is sparse (BASELINE: 5 x 68.6 k of 1 M slots = 0.34 -> dense; 10 M events on 2048 x 4096: 0.055)
        const bool sparse = p.tag && (c->opt_gram_sparse == 1 || (c->opt_gram_sparse < 0 && c->P_prev > 0 && c->n_cand >= (2u << 20) && 16ull * (size_t)c->thres * c->P_prev < c->n_cand));
        // (sparse form: up to four times the slots per wave — a wave's pipeline takes three stages to fill; option gram_sparse_chunk: 1 ... 8, no difference from 2 up)
        const long chunk_cap = sparse ? (long)c->opt_gram_sparse_chunk * kGramChunk : kGramChunk;
        const long per_round = (long)c->n_cu * (kGramBlock / 64);
        const long rounds = std::max<long>(1, ((long)c->n_cand + per_round * chunk_cap - 1) / (per_round * chunk_cap));
        long chunk = ((long)c->n_cand + per_round * rounds - 1) / (per_round * rounds);
        chunk = (chunk + 7) & ~7L;
        chunk = std::min<long>(std::max<long>(chunk, kGramChunkMin), chunk_cap);
        p.chunk = (int)chunk;
        const long waves = ((long)c->n_cand + chunk - 1) / chunk;
        if (c->kernel_timing) HIP_TRY(c, hipEventRecord(c->kt[2], s));
        constexpr long wpb = kGramBlock / 64;
        unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
        p.n_gram_blocks = (int)grid;
        const bool ep_tail = c->ep_in_gram && !ep_host && c->n_pm;
        if (ep_tail) {
            p.ep_flag = c->d_flag; p.ep_e = c->d_e_sorted; p.ep_out = c->d_ep; p.ep_fblk_cnt = c->d_fblk_cnt; p.ep_n_pm = (long)c->n_pm; p.ep_n_fblk = c->n_fblk;
            p.ep_fsup = c->d_fsup + (size_t)c->fsup_half * c->n_fsup * kFlagSupStride;
            grid += (unsigned)((c->n_pm + kEpTailBlk - 1) / kEpTailBlk);
        }
        c->ep_in_gram = false;
        const ActiveWriteParams aw = c->aw_in_gram ? c->aw_saved : ActiveWriteParams{};
        {   // gather waves per Gram block (option gather_waves = 1|2|4 overrides)
            const int gw_env = c->opt_gather_waves;
            // (measured, scripts/r04_exp15.sh: with ONE gather wave the gather outlasts the stream at every size — Gram kernel 57 vs 36 us at 1 M events,
            // 1288 vs 1047 at 40 M —, two are within noise of four or of the sweeping launch from 5 M events up: four wherever the lists are used)
            p.gather_waves = (gw_env == 1 || gw_env == 2 || gw_env == 4) ? gw_env : 4;
        }
        if (p.tag && sparse) {
            if (c->aw_in_gram) hipLaunchKernelGGL((emba_gram_kernel<true, true, true>), dim3(grid), dim3(kGramBlock), 0, s, p, aw);
            else hipLaunchKernelGGL((emba_gram_kernel<true, false, true>), dim3(grid), dim3(kGramBlock), 0, s, p, aw);
        } else if (p.tag) {
            if (c->aw_in_gram) hipLaunchKernelGGL((emba_gram_kernel<true, true>), dim3(grid), dim3(kGramBlock), 0, s, p, aw);
            else hipLaunchKernelGGL((emba_gram_kernel<true, false>), dim3(grid), dim3(kGramBlock), 0, s, p, aw);
        } else {
            if (c->aw_in_gram) hipLaunchKernelGGL((emba_gram_kernel<false, true>), dim3(grid), dim3(kGramBlock), 0, s, p, aw);
            else hipLaunchKernelGGL((emba_gram_kernel<false, false>), dim3(grid), dim3(kGramBlock), 0, s, p, aw);
        }
        c->aw_in_gram = false;
        if (ep_tail) c->ep_valid = true;
        if (c->kernel_timing) { HIP_TRY(c, hipEventRecord(c->kt[3], s)); c->kt_accum_valid = true; c->kt_valid[c->kt_slot][1] = true; }
    }
    if (c->ep_after_gram && !ep_host && c->n_pm) {      // option step_ep = 2 (A/B): the step's ep by launches of its own behind the Gram kernel: launch A's per-block flag counts -> offsets -> compaction
        { emba_status st = dev_scan(c, c->d_fblk_cnt, c->d_fblk_off, (size_t)c->n_fblk, c->d_total + 2); if (st) return st; }
        hipLaunchKernelGGL(emba_compact_ep_kernel, dim3((unsigned)c->n_fblk), dim3(256), 0, s, c->d_e_sorted, c->d_flag, (const uint32_t*)nullptr, c->d_fblk_off, (long)c->n_pm,
                           c->d_ep, (int32_t*)nullptr);
        c->ep_valid = true;
    }
    c->ep_after_gram = false;
    HIP_TRY(c, hipGetLastError());
    c->accum_done = true; c->finish_done = false;
    return EMBA_OK;
}

emba_status emba_form_finish(emba_ctx* c, double alpha, double* A11, double* b1, uint32_t* active_idx, size_t cap_P, double* A22,
                             double* b2, double* A12_dense)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->accum_done) return fail(c, EMBA_ERR_STATE, "emba_form_accumulate has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (alpha != 0.0 && !c->l2_fused) {   // (l2_fused doubles as "already applied to this set of blocks": applyL2Reg acts once)
        c->l2_fused = true;
        // P may still be unresolved on the host: the kernel reads it from device memory, the grid covers the bound
        const size_t bound = c->P_pending ? c->npix : c->P;
        if (bound)
            hipLaunchKernelGGL(emba_l2reg_kernel, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, s, pack_A22b2(c), c->d_active,
                               c->d_total + 1, alpha, c->d_Gx, c->d_Gy);
    }
    HIP_TRY(c, hipGetLastError());
    const bool download = A11 || b1 || active_idx || A22 || b2 || A12_dense;
    emba_status st = resolve_pending(c, !download);   // the step's one host wait when nothing was resolved earlier
    if (st) return st;
    if (!download) { c->finish_done = true; return EMBA_OK; }
    const size_t P = c->P;
    const int dim = 3 * c->K;
    if ((A22 || b2 || A12_dense || active_idx) && cap_P < P) return fail(c, EMBA_ERR_CAPACITY, "cap_P=%zu < P=%zu", cap_P, P);
    if (A11) HIP_TRY(c, hipMemcpyAsync(A11, pack_A11(c), (size_t)dim * dim * sizeof(double), hipMemcpyDeviceToHost, s));
    if (b1) HIP_TRY(c, hipMemcpyAsync(b1, pack_b1(c), (size_t)dim * sizeof(double), hipMemcpyDeviceToHost, s));
    if (active_idx && P) HIP_TRY(c, hipMemcpyAsync(active_idx, c->d_active, P * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    double* d_A22 = nullptr; double* d_b2 = nullptr; double* d_A12 = nullptr;
    if ((A22 || b2) && P) {
        // (round 6) the unpacked blocks sit in a workspace (a hipMalloc / hipFree pair per call until then) and travel through the pinned pipeline + the copy helpers
        // like ep: the drop-in downloads them twice per accepted step (formNormalEq, applyL2Reg), 23 MB each at config 2's shape
        if ((st = ws_get(c, 43, 6 * P * sizeof(double), (void**)&d_A22))) return st;
        d_b2 = d_A22 + 4 * P;
        hipLaunchKernelGGL(emba_unpack_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, pack_A22b2(c), (long)P, d_A22, d_b2);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(s));
        if (A22 && (st = d2h_pageable(c, A22, d_A22, 4 * P * sizeof(double)))) return st;
        if (b2 && (st = d2h_pageable(c, b2, d_b2, 2 * P * sizeof(double)))) return st;
    }
    if (A12_dense && P) {
        if ((st = ensure_compact(c))) return st;
        const size_t n12 = (size_t)dim * 2 * P;
        if ((st = dev_alloc(c, &d_A12, n12))) return st;
        (void)hipMemsetAsync(d_A12, 0, n12 * sizeof(double), s);
        if (c->n_cand)
            hipLaunchKernelGGL(emba_dense_a12_kernel, dim3((unsigned)((c->n_cand + 255) / 256)), dim3(256), 0, s, c->d_rec, c->d_slot_key,
                               (long)c->n_cand, c->d_count, c->d_compact, c->thres, c->irls, c->eta, dim, d_A12, c->set_stamp);
        (void)hipMemcpyAsync(A12_dense, d_A12, n12 * sizeof(double), hipMemcpyDeviceToHost, s);
    }
    hipError_t e = hipStreamSynchronize(s);
    c->knots_in_flight = false;
    dev_free(c, d_A12);
    if (e != hipSuccess) return fail(c, EMBA_ERR_HIP, "form_finish: %s", hipGetErrorString(e));
    HIP_TRY(c, hipGetLastError());
    c->finish_done = true;
    return EMBA_OK;
}

emba_status emba_form_normal_eq(emba_ctx* c, const double* ep, int32_t thres, int32_t irls, double eta, double alpha, double* A11,
                                double* b1, size_t* P, uint32_t* active_idx, size_t cap_P, double* A22, double* b2, double* A12_dense)
{
    emba_status st;
    size_t Pl = 0, pl = 0;
    if ((st = emba_form_active(c, thres, &Pl, &pl))) return st;
    if (P) *P = Pl;
    if ((st = emba_form_accumulate(c, ep, irls, eta))) return st;
    return emba_form_finish(c, alpha, A11, b1, active_idx, cap_P, A22, b2, A12_dense);
}

emba_status emba_get_A12_sparse(emba_ctx* c, int32_t* cp_c, int32_t* cp_p, int32_t* pix, double* w, double* jc, double* jp, double* dp)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->accum_done) return fail(c, EMBA_ERR_STATE, "no normal equations formed yet");
    HIP_TRY(c, hipSetDevice(c->device));
    { emba_status st0 = resolve_pending(c); if (st0) return st0; }
    const size_t M = c->n_cand;
    if (!M) return EMBA_OK;
    int32_t *d_c = nullptr, *d_p = nullptr, *d_x = nullptr; double *d_w = nullptr, *d_jc = nullptr, *d_jp = nullptr, *d_dp = nullptr;
    emba_status st;
    if ((st = dev_alloc(c, &d_c, M)) || (st = dev_alloc(c, &d_p, M)) || (st = dev_alloc(c, &d_x, M)) || (st = dev_alloc(c, &d_w, M)) ||
        (st = dev_alloc(c, &d_jc, 6 * M)) || (st = dev_alloc(c, &d_jp, 6 * M)) || (st = dev_alloc(c, &d_dp, 2 * M))) {
        dev_free(c, d_c); dev_free(c, d_p); dev_free(c, d_x); dev_free(c, d_w); dev_free(c, d_jc); dev_free(c, d_jp); dev_free(c, d_dp);
        return st;
    }
    hipStream_t s = c->stream;
    if ((st = ensure_compact(c))) { dev_free(c, d_c); dev_free(c, d_p); dev_free(c, d_x); dev_free(c, d_w); dev_free(c, d_jc); dev_free(c, d_jp); dev_free(c, d_dp); return st; }
    hipLaunchKernelGGL(emba_export_a12_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, c->d_rec, c->d_slot_key, (long)M,
                       c->d_count, c->d_compact, c->thres, c->irls, c->eta, d_c, d_p, d_x, d_w, d_jc, d_jp, d_dp, c->set_stamp);
    if (cp_c) (void)hipMemcpyAsync(cp_c, d_c, M * 4, hipMemcpyDeviceToHost, s);
    if (cp_p) (void)hipMemcpyAsync(cp_p, d_p, M * 4, hipMemcpyDeviceToHost, s);
    if (pix) (void)hipMemcpyAsync(pix, d_x, M * 4, hipMemcpyDeviceToHost, s);
    if (w) (void)hipMemcpyAsync(w, d_w, M * 8, hipMemcpyDeviceToHost, s);
    if (jc) (void)hipMemcpyAsync(jc, d_jc, 6 * M * 8, hipMemcpyDeviceToHost, s);
    if (jp) (void)hipMemcpyAsync(jp, d_jp, 6 * M * 8, hipMemcpyDeviceToHost, s);
    if (dp) (void)hipMemcpyAsync(dp, d_dp, 2 * M * 8, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    dev_free(c, d_c); dev_free(c, d_p); dev_free(c, d_x); dev_free(c, d_w); dev_free(c, d_jc); dev_free(c, d_jp); dev_free(c, d_dp);
    if (e != hipSuccess) return fail(c, EMBA_ERR_HIP, "get_A12_sparse: %s", hipGetErrorString(e));
    return EMBA_OK;
}

emba_status emba_compact_ep(emba_ctx* c)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched) return fail(c, EMBA_ERR_STATE, "no evaluateDataError state");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->ep_valid) return EMBA_OK;      // (the resident step's Gram launch has produced it already)
    c->inl_idx_valid = false;      // (asked for explicitly: produce it for THIS call)
    return ensure_inl_idx(c);
}

emba_status emba_get_ep(emba_ctx* c, double* ep_host, size_t cap, size_t* n_inliers)
{
    if (!c || (!ep_host && cap)) return c ? fail(c, EMBA_ERR_INVALID_ARG, "ep_host NULL") : EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched || (!c->eval_done && !c->inl_pending && !c->ep_deferred)) return fail(c, EMBA_ERR_STATE, "no evaluateDataError state");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st;
    if ((st = resolve_pending(c))) return st;
    if (!c->ep_valid && (st = ensure_inl_idx(c))) return st;      // (not produced by a resident step: the stand-alone compaction)
    if (n_inliers) *n_inliers = c->n_inliers;
    if (cap < c->n_inliers) return fail(c, EMBA_ERR_CAPACITY, "cap=%zu < inliers=%zu", cap, c->n_inliers);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->n_inliers && (st = d2h_pageable(c, ep_host, c->d_ep, c->n_inliers * sizeof(double)))) return st;
    return EMBA_OK;
}

emba_status emba_get_inlier_pixels(emba_ctx* c, uint32_t* pix_host)
{
    if (!c || !pix_host) return c ? fail(c, EMBA_ERR_INVALID_ARG, "pix_host NULL") : EMBA_ERR_INVALID_ARG;
    if (!c->eval_done && !c->inl_pending && !c->ep_deferred) return fail(c, EMBA_ERR_STATE, "no evaluateDataError state");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st;
    if ((st = resolve_pending(c)) || (st = ensure_inl_idx(c))) return st;
    if (!c->n_inliers) return EMBA_OK;
    uint32_t* d_out = nullptr;
    if ((st = ws_get(c, 15, c->n_inliers * 4, (void**)&d_out))) return st;
    hipLaunchKernelGGL(emba_inlier_pix_kernel, dim3(nblocks(c->n_pm)), dim3(256), 0, c->stream, c->d_pm_pix, c->d_flag, c->d_inl_idx, (long)c->n_pm, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(pix_host, d_out, c->n_inliers * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EMBA_OK;
}

emba_status emba_get_inlier_pixel_starts(emba_ctx* c, uint32_t* starts_host)
{
    if (!c || !starts_host) return c ? fail(c, EMBA_ERR_INVALID_ARG, "starts_host NULL") : EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched || (!c->eval_done && !c->inl_pending && !c->ep_deferred)) return fail(c, EMBA_ERR_STATE, "no evaluateDataError state");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st;
    if ((st = resolve_pending(c)) || (st = ensure_inl_idx(c))) return st;
    const uint32_t S = (uint32_t)((size_t)c->sw * c->sh);
    uint32_t *d_px = nullptr, *d_starts = nullptr;
    if ((st = ws_get(c, 15, std::max<size_t>(c->n_inliers, 1) * 4, (void**)&d_px)) || (st = ws_get(c, 42, ((size_t)S + 1) * 4, (void**)&d_starts))) return st;
    if (c->n_inliers) hipLaunchKernelGGL(emba_inlier_pix_kernel, dim3(nblocks(c->n_pm)), dim3(256), 0, c->stream, c->d_pm_pix, c->d_flag, c->d_inl_idx, (long)c->n_pm, d_px);
    hipLaunchKernelGGL(emba_pix_starts_kernel, dim3(nblocks(c->n_inliers + 1)), dim3(256), 0, c->stream, d_px, (long)c->n_inliers, S, d_starts);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(starts_host, d_starts, ((size_t)S + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->h_pix_starts.assign(starts_host, starts_host + S + 1);
    c->pix_starts_seq = c->eval_seq;
    return EMBA_OK;
}

emba_status emba_get_ep_by_pixel(emba_ctx* c, double* ep_out, const uint64_t* dst)
{
    if (!c || !ep_out || !dst) return c ? fail(c, EMBA_ERR_INVALID_ARG, "ep_out / dst NULL") : EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched || (!c->eval_done && !c->inl_pending && !c->ep_deferred)) return fail(c, EMBA_ERR_STATE, "no evaluateDataError state");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st;
    if ((st = resolve_pending(c))) return st;
    const size_t S = (size_t)c->sw * c->sh;
    if (c->pix_starts_seq != c->eval_seq || c->h_pix_starts.size() != S + 1) {
        std::vector<uint32_t> tmp(S + 1);
        if ((st = emba_get_inlier_pixel_starts(c, tmp.data()))) return st;
    }
    if (!c->ep_valid && (st = ensure_inl_idx(c))) return st;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t* starts = c->h_pix_starts.data();
    if ((size_t)starts[S] != c->n_inliers) return fail(c, EMBA_ERR_STATE, "the pixel starts do not belong to this evaluation (%u residuals there, %zu here)", starts[S], c->n_inliers);
    size_t p = 0;      // the pixel cursor only moves forward: the chunks arrive in ep order
    return d2h_chunks(c, c->d_ep, c->n_inliers * sizeof(double), [&](const void* chunk, size_t off, size_t len) {
        const size_t a = off / sizeof(double), b = (off + len) / sizeof(double);
        while (p < S && (size_t)starts[p + 1] <= a) ++p;
        for (size_t q = p; q < S && (size_t)starts[q] < b; ++q) {
            const size_t lo = std::max<size_t>(starts[q], a), hi = std::min<size_t>(starts[q + 1], b);
            if (hi > lo) std::memcpy(ep_out + dst[q] + (lo - starts[q]), (const double*)chunk + (lo - a), (hi - lo) * sizeof(double));
        }
    });
}

emba_status emba_data_cost(emba_ctx* c, int32_t irls, double eta, double* cost)
{
    if (!c || !cost) return EMBA_ERR_INVALID_ARG;
    if (!c->eval_done && !c->inl_pending && !c->ep_deferred) return fail(c, EMBA_ERR_STATE, "no residuals yet");
    HIP_TRY(c, hipSetDevice(c->device));
    { emba_status st0 = resolve_pending(c); if (st0) return st0; }
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->d_scalar, 0, sizeof(double), s));
    if (c->n_sorted) {
        const unsigned grid = (unsigned)std::min<size_t>((c->n_pm + 255) / 256, 2048);
        hipLaunchKernelGGL(emba_data_cost_kernel, dim3(grid), dim3(256), 0, s, c->d_e_sorted, c->d_flag, (long)c->n_pm, (int)irls, eta,
                           c->d_scalar);
    }
    double v = 0;
    HIP_TRY(c, hipMemcpyAsync(&v, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (irls == 0) v *= 0.5;
    else if (irls == 2) v *= 0.5 / eta;
    *cost = v;
    return EMBA_OK;
}

// Both cost terms of a trial point (solver.cpp:88-91, 265-268) in ONE launch (emba_costs_kernel: its last block writes the two sums, the step's status word and a
// sequence number to pinned host memory; the host spins on the number — no memset, no copies, no stream synchronise: 159 -> 130 us per LM iteration at the BASELINE
// shape, most of it the evaluation it waits for).  emba_costs_launch / _finish split it so that a group can enqueue every rank before it waits for any.
emba_status emba_costs_launch(emba_ctx* c, int32_t irls, double eta, int32_t with_reg)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->eval_done && !c->inl_pending && !c->ep_deferred) return fail(c, EMBA_ERR_STATE, "no residuals yet");
    if (with_reg && !c->have_map) return fail(c, EMBA_ERR_STATE, "no map");
    HIP_TRY(c, hipSetDevice(c->device));
    // An evaluation that has only been launched stays that way: the reductions read the per-event residuals and flags the warp kernel wrote,
    // not the compacted vector — so the formNormalEq that follows an accepted trial still finds the post-warp work fused (emba_form_active), and
    // a rejected trial never pays for a residual vector nobody asks for.  The step's status word comes back with the sums (emba_costs_finish).
    if (!c->ep_deferred) { emba_status st0 = resolve_pending(c); if (st0) return st0; }
    if (!c->h_cost) {
        HIP_TRY(c, hipHostMalloc((void**)&c->h_cost, 4 * sizeof(double), hipHostMallocDefault));
        memset(c->h_cost, 0, 4 * sizeof(double));
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->h_cost_dev, c->h_cost, 0));
        HIP_TRY(c, hipMalloc((void**)&c->d_cost_acc, 4 * sizeof(double)));
        HIP_TRY(c, hipMemsetAsync(c->d_cost_acc, 0, 4 * sizeof(double), c->stream));      // (on the kernels' stream: a hipMemset on the default stream is not ordered in front of them)
    }
    hipStream_t s = c->stream;
    CostsParams p{};
    p.e_sorted = c->d_e_sorted; p.flag = c->d_flag; p.n_pm = c->n_sorted ? (long)c->n_pm : 0L; p.irls = (int)irls; p.eta = eta;
    p.Gx = c->d_Gx; p.Gy = c->d_Gy; p.npix = with_reg ? (long)c->npix : 0L;
    // (a few hundred blocks: every block ends with two same-address atomics — its sum and its ticket —, and 4096 of them on one word serialise for longer than the sums take)
    p.nb_data = (int)std::max<size_t>(1, std::min<size_t>(((size_t)p.n_pm + 255) / 256, (size_t)c->n_cu));
    p.nb_reg = with_reg ? (int)std::max<size_t>(1, std::min<size_t>((c->npix + 255) / 256, (size_t)2 * c->n_cu)) : 0;
    p.acc = c->d_cost_acc; p.ticket = reinterpret_cast<unsigned int*>(c->d_cost_acc + 2); p.err_dev = c->d_err;
    p.host_out = c->h_cost_dev; p.seq = ++c->cost_seq;
    hipLaunchKernelGGL(emba_costs_kernel, dim3((unsigned)(p.nb_data + p.nb_reg)), dim3(256), 0, s, p);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}
emba_status emba_costs_finish(emba_ctx* c, int32_t irls, double eta, double alpha, double* data_cost, double* reg_cost)
{
    if (!c || !c->h_cost) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    {   // the kernel's last block publishes the sequence number behind the sums: spin on it (a stream synchronise costs tens of microseconds); bounded — a faulted
        // kernel never publishes — with the stream as the fallback
        volatile int* w = reinterpret_cast<volatile int*>(c->h_cost + 3);
        bool polled = false;
        for (long spin = 0; spin < 50000000L; ++spin) { if (w[0] == c->cost_seq) { polled = true; break; } __builtin_ia32_pause(); }
        if (polled) { std::atomic_thread_fence(std::memory_order_acquire); c->spun = true; }
        else { HIP_TRY(c, hipStreamSynchronize(c->stream)); c->spun = false; }
    }
    { int w = 0; memcpy(&w, c->h_cost + 2, sizeof(int)); if (w & 1) return fail(c, EMBA_ERR_TIME_RANGE, "a batch midpoint lies outside the spline's knots"); }
    double v = c->h_cost[0];
    if (irls == 0) v *= 0.5;
    else if (irls == 2) v *= 0.5 / eta;
    if (data_cost) *data_cost = v;
    if (reg_cost) *reg_cost = 0.5 * alpha * c->h_cost[1];
    return EMBA_OK;
}
emba_status emba_costs(emba_ctx* c, int32_t irls, double eta, double alpha, double* data_cost, double* reg_cost)
{
    emba_status st = emba_costs_launch(c, irls, eta, reg_cost ? 1 : 0);
    if (st) return st;
    return emba_costs_finish(c, irls, eta, alpha, data_cost, reg_cost);
}

emba_status emba_reg_cost(emba_ctx* c, double alpha, double* cost)
{
    if (!c || !cost) return EMBA_ERR_INVALID_ARG;
    if (!c->have_map) return fail(c, EMBA_ERR_STATE, "no map");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->d_scalar + 1, 0, sizeof(double), s));
    const unsigned grid = (unsigned)std::min<size_t>((c->npix + 255) / 256, 2048);
    hipLaunchKernelGGL(emba_reg_cost_kernel, dim3(grid), dim3(256), 0, s, c->d_Gx, c->d_Gy, (long)c->npix, c->d_scalar + 1);
    double v = 0;
    HIP_TRY(c, hipMemcpyAsync(&v, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *cost = 0.5 * alpha * v;
    return EMBA_OK;
}

emba_status emba_dump_state(emba_ctx* c, double* pm, double* D, int32_t* cp_idx, int32_t* inlier_idx, int32_t* pm_int, double* dp,
                            double* Gpm, double* temp)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->eval_done && !c->inl_pending && !c->ep_deferred) return fail(c, EMBA_ERR_STATE, "no evaluateDataError state to dump");
    HIP_TRY(c, hipSetDevice(c->device));
    { emba_status st0 = resolve_pending(c); if (st0) return st0; }
    if (inlier_idx) { emba_status st0 = ensure_inl_idx(c); if (st0) return st0; }
    const size_t ns = c->n_sorted, n = c->n_in;
    if (!ns) return EMBA_OK;
    hipStream_t s = c->stream;
    // device and host staging only for what was asked for (a 100 M-event pm dump is 1.6 GB, the full state 17 GB)
    const bool w_pm = pm, w_D = D, w_dp = dp, w_G = Gpm, w_t = temp, w_pi = pm_int, w_inl = inlier_idx, w_flag = inlier_idx || pm_int || Gpm || temp;
    double *d_pm = nullptr, *d_D = nullptr, *d_dp = nullptr, *d_G = nullptr, *d_t = nullptr; int32_t* d_pi = nullptr;
    auto free_all = [&]() { dev_free(c, d_pm); dev_free(c, d_D); dev_free(c, d_dp); dev_free(c, d_G); dev_free(c, d_t); dev_free(c, d_pi); };
    emba_status st = EMBA_OK;
    if ((w_pm && (st = dev_alloc(c, &d_pm, 2 * ns))) || (w_D && (st = dev_alloc(c, &d_D, 12 * ns))) || (w_dp && (st = dev_alloc(c, &d_dp, 2 * ns))) ||
        (w_G && (st = dev_alloc(c, &d_G, 2 * ns))) || (w_t && (st = dev_alloc(c, &d_t, 2 * ns))) || (w_pi && (st = dev_alloc(c, &d_pi, 2 * ns)))) {
        free_all();
        return st;
    }
    if (w_dp) (void)hipMemsetAsync(d_dp, 0, 2 * ns * 8, s);
    if (w_G) (void)hipMemsetAsync(d_G, 0, 2 * ns * 8, s);
    if (w_t) (void)hipMemsetAsync(d_t, 0, 2 * ns * 8, s);
    if (w_pi) (void)hipMemsetAsync(d_pi, 0xFF, 2 * ns * 4, s);
    WarpParams p{};
    p.ev_pix = c->d_ev_pix; p.ev_batch = c->d_ev_batch; p.ev_slot = c->d_ev_slot; p.ev_pm = c->tile_order ? c->d_ev_pm : nullptr; p.n_sorted = (long)ns; p.nblk = c->nblk;
    p.ev_u = c->d_ev_u; p.ev_seg = c->d_ev_seg;
    p.pose = c->d_pose; p.seg = c->d_seg; p.lut = c->d_lut; p.texel = nullptr; p.rect_acc = nullptr; p.Gx = c->d_Gx; p.Gy = c->d_Gy; p.pixacc = c->d_pixacc;
    p.W = c->W; p.H = c->H; p.fx = c->fx; p.fy = c->fy; p.cx = c->cx;
    p.cy = c->cy; p.C_th = c->C_th; p.outlier_px = c->outlier_px; p.count = c->d_count; p.rec = c->d_rec; p.e_sorted = c->d_e_sorted;
    p.flag = c->d_flag; p.d_pm = d_pm; p.d_D = d_D; p.d_dp = d_dp; p.d_Gpm = d_G; p.d_temp = d_t; p.d_pm_int = d_pi;
    if (c->tile_order) hipLaunchKernelGGL((emba_warp_residual_kernel<true, true>), dim3((unsigned)grid8(c->nblk)), dim3(kWarpBlock), 0, s, p);
    else if (c->segpose) hipLaunchKernelGGL((emba_warp_residual_kernel<true, false, true>), dim3((unsigned)grid8(c->nblk)), dim3(kWarpBlock), 0, s, p);   // (the last evaluation left segment records, not a pose table)
    else hipLaunchKernelGGL((emba_warp_residual_kernel<true, false>), dim3((unsigned)grid8(c->nblk)), dim3(kWarpBlock), 0, s, p);
    std::vector<double> h_pm(w_pm ? 2 * ns : 0), h_D(w_D ? 12 * ns : 0), h_dp(w_dp ? 2 * ns : 0), h_G(w_G ? 2 * ns : 0), h_t(w_t ? 2 * ns : 0);
    std::vector<uint16_t> h_cp(cp_idx ? c->n_batch : 0);
    std::vector<int32_t> h_pi(w_pi ? 2 * ns : 0), h_inl(w_inl ? c->n_pm : 0);     // (inlier numbers and flags are indexed in pm-order)
    std::vector<uint8_t> h_flag(w_flag ? c->n_pm : 0);
    if (w_pm) (void)hipMemcpyAsync(h_pm.data(), d_pm, 2 * ns * 8, hipMemcpyDeviceToHost, s);
    if (w_D) (void)hipMemcpyAsync(h_D.data(), d_D, 12 * ns * 8, hipMemcpyDeviceToHost, s);
    if (w_dp) (void)hipMemcpyAsync(h_dp.data(), d_dp, 2 * ns * 8, hipMemcpyDeviceToHost, s);
    if (w_G) (void)hipMemcpyAsync(h_G.data(), d_G, 2 * ns * 8, hipMemcpyDeviceToHost, s);
    if (w_t) (void)hipMemcpyAsync(h_t.data(), d_t, 2 * ns * 8, hipMemcpyDeviceToHost, s);
    if (w_pi) (void)hipMemcpyAsync(h_pi.data(), d_pi, 2 * ns * 4, hipMemcpyDeviceToHost, s);
    if (w_inl) (void)hipMemcpyAsync(h_inl.data(), c->d_inl_idx, c->n_pm * 4, hipMemcpyDeviceToHost, s);
    if (w_flag) (void)hipMemcpyAsync(h_flag.data(), c->d_flag, c->n_pm, hipMemcpyDeviceToHost, s);
    if (cp_idx) (void)hipMemcpyAsync(h_cp.data(), c->d_cp, h_cp.size() * 2, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    free_all();
    if (e != hipSuccess) return fail(c, EMBA_ERR_HIP, "dump_state: %s", hipGetErrorString(e));
    // pure re-indexing (pixel-major -> original time order); no arithmetic on the host
    if (cp_idx) for (size_t k = 0; k < n; ++k) cp_idx[k] = -1;
    if (inlier_idx) for (size_t k = 0; k < n; ++k) inlier_idx[k] = -2;
    if (pm_int) for (size_t k = 0; k < 2 * n; ++k) pm_int[k] = -1;
    if (pm) memset(pm, 0, 2 * n * 8);
    if (D) memset(D, 0, 12 * n * 8);
    if (dp) memset(dp, 0, 2 * n * 8);
    if (Gpm) memset(Gpm, 0, 2 * n * 8);
    if (temp) memset(temp, 0, 2 * n * 8);
    // entry of the device order -> pm-order index -> original event (lead-in copies and halo entries map to nothing)
    std::vector<uint32_t> h_evpix(ns), h_evbatch(cp_idx ? ns : 0), h_evpm(c->have_ev_pm ? ns : 0), h_pmorig(c->n_pm);
    HIP_TRY(c, hipMemcpy(h_evpix.data(), c->d_ev_pix, ns * 4, hipMemcpyDeviceToHost));
    if (cp_idx) HIP_TRY(c, hipMemcpy(h_evbatch.data(), c->d_ev_batch, ns * 4, hipMemcpyDeviceToHost));
    if (c->have_ev_pm) HIP_TRY(c, hipMemcpy(h_evpm.data(), c->d_ev_pm, ns * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(h_pmorig.data(), c->d_pm_orig, c->n_pm * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ns; ++i) {
        if (h_evpix[i] & kEvLead) continue;                        // lead-in copy / halo
        const size_t f = c->have_ev_pm ? h_evpm[i] : i;           // pm-order index: where flag / inlier number of this entry live
        const uint32_t k = h_pmorig[f];
        if (k == 0xFFFFFFFFu) continue;
        const bool cand = (h_evpix[i] & kEvHasPred) != 0;
        if (pm) { pm[2 * k] = h_pm[2 * i]; pm[2 * k + 1] = h_pm[2 * i + 1]; }
        if (D) memcpy(D + 12 * (size_t)k, &h_D[12 * i], 12 * 8);
        if (cp_idx) cp_idx[k] = (int32_t)h_cp[h_evbatch[i]];
        if (inlier_idx) inlier_idx[k] = cand ? (h_flag[f] ? h_inl[f] : -1) : -2;
        if (pm_int && h_flag[f]) { pm_int[2 * k] = h_pi[2 * i]; pm_int[2 * k + 1] = h_pi[2 * i + 1]; }
        if (dp && cand) { dp[2 * k] = h_dp[2 * i]; dp[2 * k + 1] = h_dp[2 * i + 1]; }
        if (Gpm && h_flag[f]) { Gpm[2 * k] = h_G[2 * i]; Gpm[2 * k + 1] = h_G[2 * i + 1]; }
        if (temp && h_flag[f]) { temp[2 * k] = h_t[2 * i]; temp[2 * k + 1] = h_t[2 * i + 1]; }
    }
    return EMBA_OK;
}

emba_status emba_step(emba_ctx* c, const double* knots, int32_t K, int64_t t0_ns, int64_t dt_ns, int32_t thres, int32_t irls, double eta,
                      double alpha, size_t* n_inliers, size_t* P)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (irls < 0 || irls > 2) return fail(c, EMBA_ERR_INVALID_ARG, "irls must be 0 (quadratic), 1 (huber) or 2 (cauchy)");
    emba_status st;
    {   // the evaluation weights its per-pixel sums with THIS step's cost (whatever emba_set_cost declared for other callers)
        const int keep_irls = c->cost_irls; const double keep_eta = c->cost_eta;
        c->cost_irls = irls; c->cost_eta = irls ? eta : 0.0;
        c->no_alt_set = (c->step_one_set != 0);
        st = emba_eval_launch(c, knots, K, t0_ns, dt_ns);
        c->no_alt_set = false;
        c->cost_irls = keep_irls; c->cost_eta = keep_eta;
        if (st) return st;
    }
    if ((st = emba_eval_finish(c, nullptr, nullptr, nullptr))) return st;
    c->fused_alpha = alpha;   // A22/b2 come from the accumulator, so applyL2Reg rides along with the gather
    c->step_consume = (c->step_fast != 0);   // ... which is their only reader: it zeroes the lines behind itself and the next evaluation needs no clearing pass
    c->step_wants_ep = (c->step_ep != 0);    // the step returns what evaluateDataError returns: ep, compacted in the tail of its Gram launch
    st = emba_form_active(c, thres, nullptr, nullptr);
    c->step_consume = false; c->step_wants_ep = false;
    if (st) return st;
    if ((st = emba_form_accumulate(c, nullptr, irls, eta))) return st;
    if ((st = emba_form_finish(c, alpha, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr))) return st;
    if (n_inliers) *n_inliers = c->n_inliers;
    if (P) *P = c->P;
    return EMBA_OK;
}

emba_status emba_step_form_active(emba_ctx* c, int32_t thres, const uint8_t* global_counts_u8_dev)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->eval_launched) return fail(c, EMBA_ERR_STATE, "emba_eval_launch has not been called");
    emba_status st;
    if ((st = emba_eval_finish(c, nullptr, nullptr, nullptr))) return st;
    c->step_consume = (c->step_fast != 0);
    c->global_u8 = global_counts_u8_dev;
    st = emba_form_active(c, thres, nullptr, nullptr);
    c->step_consume = false; c->global_u8 = nullptr;
    return st;
}

emba_status emba_count_map_ready(emba_ctx* c)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return ensure_counts(c);
}

emba_status emba_set_cost(emba_ctx* c, int32_t irls, double eta)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (irls < 0 || irls > 2) return fail(c, EMBA_ERR_INVALID_ARG, "irls must be 0 (quadratic), 1 (huber) or 2 (cauchy)");
    c->cost_irls = irls; c->cost_eta = irls ? eta : 0.0;
    return EMBA_OK;
}

// ---- tuning and A/B switches (documented, versioned by EMBA_ABI_VERSION; not environment variables) ----------------------------
namespace {
struct OptionRef { const char* name; int emba_ctx::*field; int lo, hi; };
const OptionRef kOptions[] = {
    {"step_ep", &emba_ctx::step_ep, 0, 2},
    {"step_fast", &emba_ctx::step_fast, 0, 1},
    {"step_gather", &emba_ctx::step_gather, 0, 3},
    {"step_one_set", &emba_ctx::step_one_set, 0, 1},
    {"gram_tags", &emba_ctx::use_tags, 0, 1},
    {"segpose", &emba_ctx::segpose_mode, 0, 2},
    {"order", &emba_ctx::order_mode, 0, 2},
    {"texel", &emba_ctx::texel_mode, 0, 3},
    {"solve_perm", &emba_ctx::solve_perm_mode, -1, 1},
    {"gather_waves", &emba_ctx::opt_gather_waves, 0, 4},
    {"chunk_order_bin", &emba_ctx::opt_chunk_order_bin, 0, 1},
    {"poison", &emba_ctx::opt_poison, 0, 1},
    {"tile_reserve", &emba_ctx::opt_tile_reserve, 0, 5},
    {"tile_shape", &emba_ctx::opt_tile_shape, -1, kNumTileShapes - 1},
    {"tile_fine", &emba_ctx::opt_tile_fine, -1, 1},
    {"tile_min_events", &emba_ctx::opt_tile_min_events, 0, 2000000000},
    {"tile_chunk", &emba_ctx::opt_tile_chunk, 0, 1 << 20},
    {"solve_counts", &emba_ctx::opt_solve_counts, -1, 2},
    {"syrk_dense", &emba_ctx::opt_syrk_dense, 0, 1},
    {"syrk_lists", &emba_ctx::opt_syrk_lists, 0, 2},
    {"gram_sparse", &emba_ctx::opt_gram_sparse, -1, 1},
    {"gram_sparse_chunk", &emba_ctx::opt_gram_sparse_chunk, 1, 8},
    {"syrk_min_cols", &emba_ctx::opt_syrk_min_cols, 64, 4096},
    {"syrk_item_cap", &emba_ctx::opt_syrk_item_cap, 1, 65536},
    {"solve_debug", &emba_ctx::opt_solve_debug, 0, 1},
    {"poisson", &emba_ctx::opt_poisson, 0, 2},
    {"gemm64", &emba_ctx::opt_gemm64, 0, 1},
};
}  // namespace

emba_status emba_set_option(emba_ctx* c, const char* name, int32_t value)
{
    if (!c || !name) return EMBA_ERR_INVALID_ARG;
    for (const OptionRef& o : kOptions)
        if (!strcmp(o.name, name)) {
            if (value < o.lo || value > o.hi) return fail(c, EMBA_ERR_INVALID_ARG, "option %s: %d outside [%d, %d]", name, (int)value, o.lo, o.hi);
            c->*(o.field) = (int)value;
            if (!strcmp(name, "order") || !strcmp(name, "chunk_order_bin") || !strncmp(name, "tile_", 5)) c->keys_ready = false;      // the device order is rebuilt at the next evaluation
            return EMBA_OK;
        }
    return fail(c, EMBA_ERR_INVALID_ARG, "unknown option '%s'", name);
}

emba_status emba_get_option(emba_ctx* c, const char* name, int32_t* value)
{
    if (!c || !name || !value) return EMBA_ERR_INVALID_ARG;
    if (!strcmp(name, "ep_valid")) { *value = c->ep_valid ? 1 : 0; return EMBA_OK; }      // read-only: the device holds the last evaluation's ep (a step produced it, or a compaction)
    for (const OptionRef& o : kOptions)
        if (!strcmp(o.name, name)) { *value = c->*(o.field); return EMBA_OK; }
    return fail(c, EMBA_ERR_INVALID_ARG, "unknown option '%s'", name);
}

emba_status emba_last_counts(emba_ctx* c, size_t* n_inliers, size_t* P)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st = resolve_pending(c, true);
    if (st) return st;
    if (n_inliers) *n_inliers = c->n_inliers;
    if (P) *P = c->P;
    return EMBA_OK;
}

emba_status emba_sync(emba_ctx* c)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st = resolve_pending(c);
    if (st) return st;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->spun = false;
    return EMBA_OK;
}

emba_status emba_timer_start(emba_ctx* c, int32_t slot)
{
    if (!c || slot < 0 || slot >= 8) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipEventRecord(c->ev_start[slot], c->stream));
    return EMBA_OK;
}

emba_status emba_timer_stop(emba_ctx* c, int32_t slot)
{
    if (!c || slot < 0 || slot >= 8) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipEventRecord(c->ev_stop[slot], c->stream));
    return EMBA_OK;
}

emba_status emba_timer_elapsed_ms(emba_ctx* c, int32_t slot, float* ms)
{
    if (!c || !ms || slot < 0 || slot >= 8) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipEventSynchronize(c->ev_stop[slot]));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev_start[slot], c->ev_stop[slot]));
    return EMBA_OK;
}

emba_status emba_enable_kernel_timing(emba_ctx* c, int32_t on)
{
    if (!c || on < 0 || on > 16) return EMBA_ERR_INVALID_ARG;
    c->kernel_timing = on != 0;
    if (on) {       // on = 1 + slot: the next evaluation / form record their events in that slot, to be read later (emba_kernel_ms_slot)
        const int slot = on - 1;
        for (int i = 0; i < 5; ++i) if (!c->kt_sets[slot][i]) HIP_TRY(c, hipEventCreate(&c->kt_sets[slot][i]));
        c->kt = c->kt_sets[slot]; c->kt_slot = slot;
        c->kt_valid[slot][0] = c->kt_valid[slot][1] = c->kt_valid[slot][2] = false;
    }
    c->kt_warp_valid = c->kt_accum_valid = false;
    return EMBA_OK;
}

emba_status emba_kernel_ms_slot(emba_ctx* c, int32_t slot, float* warp_ms, float* accum_ms)
{
    if (!c || slot < 0 || slot >= 16) return EMBA_ERR_INVALID_ARG;
    hipEvent_t* k = c->kt_sets[slot];
    if (warp_ms) {
        *warp_ms = -1.f;
        if (c->kt_valid[slot][0]) { HIP_TRY(c, hipEventSynchronize(k[1])); HIP_TRY(c, hipEventElapsedTime(warp_ms, k[0], k[1])); }
    }
    if (accum_ms) {
        *accum_ms = -1.f;
        if (c->kt_valid[slot][1]) { HIP_TRY(c, hipEventSynchronize(k[3])); HIP_TRY(c, hipEventElapsedTime(accum_ms, k[2], k[3])); }
    }
    return EMBA_OK;
}

emba_status emba_last_kernel_ms(emba_ctx* c, float* warp_ms, float* accum_ms)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    return emba_kernel_ms_slot(c, c->kt_slot, warp_ms, accum_ms);
}

emba_status emba_kernel_timing_all(emba_ctx* c, int32_t on)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    c->kt_all = on != 0;
    return EMBA_OK;
}

emba_status emba_kernel_ms_all(emba_ctx* c, int32_t slot, float* ms4)
{
    if (!c || !ms4 || slot < 0 || slot >= 16) return EMBA_ERR_INVALID_ARG;
    hipEvent_t* k = c->kt_sets[slot];
    for (int i = 0; i < 4; ++i) ms4[i] = -1.f;
    if (!(c->kt_valid[slot][0] && c->kt_valid[slot][1] && c->kt_valid[slot][2])) return EMBA_OK;
    HIP_TRY(c, hipEventSynchronize(k[3]));
    HIP_TRY(c, hipEventElapsedTime(ms4 + 0, k[4], k[0]));     // prep || pose || texel (+ the full texel pack where it is used)
    HIP_TRY(c, hipEventElapsedTime(ms4 + 1, k[0], k[1]));     // warp kernel
    HIP_TRY(c, hipEventElapsedTime(ms4 + 2, k[1], k[2]));     // launch A (+ the sweeping active write where it is a launch of its own)
    HIP_TRY(c, hipEventElapsedTime(ms4 + 3, k[2], k[3]));     // Gram kernel (with the gather inside)
    return EMBA_OK;
}

emba_status emba_bracket_overhead_us(emba_ctx* c, int32_t reps, float* us)
{
    if (!c || !us || reps < 1 || reps > 1000) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    double sum = 0.0;
    for (int r = 0; r < reps; ++r) {
        // the bracket sits BEHIND queued work, like the brackets of a step: a busy kernel first, then event | empty kernel | event
        hipLaunchKernelGGL(emba_clock_probe_kernel, dim3(1), dim3(64), 0, s, c->d_probe + 4, 256);
        HIP_TRY(c, hipEventRecord(c->cal_ev[0], s));
        hipLaunchKernelGGL(emba_empty_kernel, dim3(1), dim3(64), 0, s);
        HIP_TRY(c, hipEventRecord(c->cal_ev[1], s));
        HIP_TRY(c, hipEventSynchronize(c->cal_ev[1]));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->cal_ev[0], c->cal_ev[1]));
        sum += ms;
    }
    HIP_TRY(c, hipGetLastError());
    *us = (float)(sum / reps * 1e3);
    return EMBA_OK;
}

emba_status emba_clock_probe(emba_ctx* c, double* sclk_mhz_est, double* probe_us, double* memtime_per_realtime, int32_t* clock_rate_khz,
                             int32_t* mem_clock_rate_khz, int32_t* wall_clock_rate_khz)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int v = 0;
    if (clock_rate_khz) { *clock_rate_khz = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, c->device) == hipSuccess) *clock_rate_khz = v; }
    if (mem_clock_rate_khz) { *mem_clock_rate_khz = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, c->device) == hipSuccess) *mem_clock_rate_khz = v; }
    int wall_khz = 100000;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && v > 0) wall_khz = v;
    if (wall_clock_rate_khz) *wall_clock_rate_khz = wall_khz;
    // every SIMD of the chip busy with one wave's dependent chain of kClockProbeOps fp32 adds (4 cycles each on a 16-lane SIMD, issued back to back)
    const int loops = 1024;
    HIP_TRY(c, hipEventRecord(c->cal_ev[0], s));
    hipLaunchKernelGGL(emba_clock_probe_kernel, dim3((unsigned)c->n_cu), dim3(256), 0, s, c->d_probe, loops);
    HIP_TRY(c, hipEventRecord(c->cal_ev[1], s));
    HIP_TRY(c, hipGetLastError());
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(h, c->d_probe, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->spun = false;
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->cal_ev[0], c->cal_ev[1]));
    const double wall_s = (double)h[1] / ((double)wall_khz * 1e3);      // the wave's own constant-rate clock around its chain
    if (probe_us) *probe_us = wall_s * 1e6;
    // s_memtime counts shader cycles (measured, round 5: 8.08 ticks per dependent v_add_f32 of the chain, the pipeline's 8-cycle dependent-issue
    // cadence; its ratio to the 100-MHz counter moves with the power state), so ticks(s_memtime) / seconds(s_memrealtime) IS the shader clock
    if (sclk_mhz_est) *sclk_mhz_est = wall_s > 0 ? (double)h[0] / wall_s * 1e-6 : 0.0;
    if (memtime_per_realtime) *memtime_per_realtime = h[1] ? (double)h[0] / ((double)kClockProbeUnroll * loops) : 0.0;
    (void)ms;
    return EMBA_OK;
}

emba_status emba_device_pci_bus_id(emba_ctx* c, char* buf, size_t len)
{
    if (!c || !buf || len < 16) return EMBA_ERR_INVALID_ARG;
    HIP_TRY(c, hipDeviceGetPCIBusId(buf, (int)len, c->device));
    return EMBA_OK;
}

}  // extern "C"

// ---- f1: the solvers (solve_kernels.h) ------------------------------------------------------------------------------------------
namespace {


// A kernel that keeps the 3K-vector x1 (or the pose part of a CG vector) in dynamic LDS: above 64 KB (K > 2730) the launch needs the attribute raised, above the
// CU's 160 KB there is no such launch — say so instead of a generic launch failure
emba_status pose_vector_lds(emba_ctx* c, const void* kernel, size_t bytes, const char* what, size_t per_pose = 24)
{
    if (bytes > (size_t)160 * 1024)
        return fail(c, EMBA_ERR_CAPACITY, "%s: K = %zu control poses need %zu bytes of LDS for the pose vector (limit 160 KB: K <= %zu)", what, bytes / per_pose, bytes, (size_t)160 * 1024 / per_pose);
    if (bytes > (size_t)64 * 1024) HIP_TRY(c, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return EMBA_OK;
}

// per-pixel record lists over `n_pix` pixels of `view` (n_rec records): counts from the records themselves, exclusive scan, fill.
// Workspaces 0 (off), 1 (cursor), 33 (the records in pixel order).
emba_status build_lists(emba_ctx* c, const RecView& view, size_t n_rec, size_t n_pix, SolveLists* out)
{
    hipStream_t s = c->stream;
    uint32_t *d_off = nullptr, *d_cursor = nullptr; double* d_sorted = nullptr;
    emba_status st;
    const uint32_t key_stamp = view.packed ? c->set_stamp : view.stamp;
    if (c->lists_valid && c->lists_packed == (view.packed != 0) && c->lists_stamp == key_stamp && c->lists_P == n_pix && c->lists_nrec == n_rec && c->ws[0].p && c->ws[33].p &&
        (!view.packed || c->lists_lo == view.pix_base)) {
        out->off = (uint32_t*)c->ws[0].p;
        out->sorted = RecView{};
        out->sorted.rec = (const double*)c->ws[33].p; out->sorted.packed = 1; out->sorted.pix_base = 0;
        return EMBA_OK;
    }
    if (view.packed && !view.rec) return fail(c, EMBA_ERR_STATE, "the received records of these equations are not cached on this rank: run the record exchange (emba_solve_shard_cached says when it can be skipped)");
    c->lists_valid = false; c->perm_valid = false;
    if ((st = ws_get(c, 0, (n_pix + 2) * 4, (void**)&d_off)) || (st = ws_get(c, 1, (n_pix + 1) * 4, (void**)&d_cursor)) ||
        (st = ws_get(c, 33, (n_rec + 1) * kRecStride * sizeof(double), (void**)&d_sorted)))
        return st;
    // list lengths: from the context's own count map while it still belongs to the evaluation that wrote these records (count_stamp), else counted
    // from the records (all-reduced / saturated / externally bound counts, a trial evaluation since, the sharded solve's received records)
    const int cmap_env = c->opt_solve_counts;      // option solve_counts — 0: always from the records; 2: both, compared (diagnostic)
    const bool from_map = cmap_env != 0 && !view.packed && c->count_stamp != 0 && c->count_stamp == view.stamp && c->d_count == c->d_count_own && n_pix == c->P && n_pix;
    if (from_map && cmap_env != 2) {
        hipLaunchKernelGGL(emba_csr_count_from_map_kernel, dim3(nblocks(n_pix)), dim3(256), 0, s, c->d_active, c->d_count, (long)n_pix, d_cursor);
    } else {
        HIP_TRY(c, hipMemsetAsync(d_cursor, 0, (n_pix + 1) * 4, s));
        if (n_rec) hipLaunchKernelGGL(emba_csr_count_kernel, dim3(nblocks(n_rec)), dim3(256), 0, s, view, (long)n_rec, d_cursor);
        if (from_map) {      // diagnostic: the two must agree
            std::vector<uint32_t> a(n_pix), b(n_pix);
            uint32_t* d_tmp = nullptr;
            if ((st = ws_get(c, 38, n_pix * 4, (void**)&d_tmp))) return st;
            hipLaunchKernelGGL(emba_csr_count_from_map_kernel, dim3(nblocks(n_pix)), dim3(256), 0, s, c->d_active, c->d_count, (long)n_pix, d_tmp);
            HIP_TRY(c, hipMemcpyAsync(a.data(), d_cursor, n_pix * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(b.data(), d_tmp, n_pix * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            size_t bad = 0; for (size_t i = 0; i < n_pix; ++i) bad += a[i] != b[i];
            fprintf(stderr, "[solve counts] P %zu: %zu pixels where the count map and the records disagree\n", n_pix, bad);
            if (bad) return fail(c, EMBA_ERR_STATE, "count map and record counts disagree on %zu pixels", bad);
        }
    }
    if ((st = dev_scan(c, d_cursor, d_off, n_pix, d_off + n_pix))) return st;
    HIP_TRY(c, hipMemsetAsync(d_cursor, 0, (n_pix + 1) * 4, s));
    // the participating records, copied into pixel order (the passes over them — U build, x2, every CG iteration — then stream)
    if (n_rec) hipLaunchKernelGGL(emba_csr_fill_sorted_kernel, dim3(nblocks(n_rec, 32)), dim3(256), 0, s, view, (long)n_rec, d_off, d_cursor, d_sorted);
    HIP_TRY(c, hipGetLastError());
    out->off = d_off;
    out->sorted = RecView{};
    out->sorted.rec = d_sorted; out->sorted.packed = 1; out->sorted.pix_base = 0;
    c->lists_valid = true; c->lists_packed = view.packed != 0; c->lists_stamp = key_stamp; c->lists_P = n_pix; c->lists_nrec = n_rec; c->lists_lo = view.pix_base;
    return EMBA_OK;
}

// S_aug (lds x (n+1), zero or pre-initialised) -= U_aug U_aug^T over the pixels [0, n_pix) of the lists; A22b2 points at the first of
// those pixels' {xx xy yy bx by}.  Also leaves y = C^-1 b2 (d_y) and the 2x2 Cholesky factors (d_cf).  Workspaces 8 (U), 12 (slabs).
bool axis_path(const emba_ctx* c, double* path_az_out, double* path_el_out);

emba_status schur_accumulate(emba_ctx* c, const RecView& view, const SolveLists& L, size_t n_pix, const double* A22b2, double lambda, int n,
                             double* d_S, long lds_, double* d_y, double* d_cf, int* d_info, const uint32_t* perm = nullptr)
{
    hipStream_t s = c->stream;
    // (the SYRK covers the n rows of S; row n of the augmented matrix, the right-hand side b1 - U y, is accumulated by the build kernel itself)
    // (round 6, measured and dropped: chunk i + 1's U build on this stream beside chunk i's product on a second one — every chunk count was slower than one after
    // the other, 3.87 -> 4.2 / 4.6 / 5.3 ms for 2 / 4 / 8 chunks at config 2's shape: profiles/r06_schur_pipe_ab.txt)
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(std::max<size_t>(n_pix, 1), (size_t)(6ull << 30) / (16ull * (size_t)lds_)));   // <= 6 GB of U
    const int nb64 = (n + 63) / 64, nbp = nb64 * (nb64 + 1) / 2;
    const int nks_max = std::max(1, (4 * c->n_cu + nbp - 1) / nbp);   // enough (tile pair, K slab) blocks to fill the chip ...
    double *d_U = nullptr, *d_slab = nullptr;
    emba_status st;
    if ((st = ws_get(c, 8, (size_t)lds_ * 2 * chunk * 8, (void**)&d_U)) || (st = ws_get(c, 12, (size_t)nks_max * nbp * 4096 * 8, (void**)&d_slab))) return st;
    SchurBuildParams bp{};
    bp.view = view; bp.off = L.off; bp.A22b2 = A22b2; bp.lambda = lambda;
    bp.irls = c->irls; bp.eta = c->eta; bp.n = n; bp.U = d_U; bp.ldu = lds_; bp.yv = d_y; bp.cfac = d_cf; bp.info = d_info;
    bp.rhs_row = d_S + n; bp.lds = lds_; bp.perm = perm;
    const size_t lds_bytes = (size_t)(2 * kBuildWaves + 1) * n * sizeof(double);
    if (lds_bytes > 160 * 1024) return fail(c, EMBA_ERR_CAPACITY, "K=%d too large for the per-wave column staging in LDS", n / 3);
    if (lds_bytes > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)emba_schur_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    // Block-sparse SYRK (>= 4 row blocks, i.e. K >= 64): the columns of a slice of kSyrkSlicePix consecutive active pixels — a piece of a
    // panorama row — are non-zero only in the rows of the control poses in view while the camera looked there; over a long window
    // (config 2: 10 s, K = 201) that is a band, and only the (row-block pair, slice) products with both blocks populated are formed.
    // Option syrk_dense switches it off (comparison).
    const bool sparse = nb64 >= 4 && nb64 <= 64 && !c->opt_syrk_dense;
    for (size_t p0 = 0; p0 < n_pix; p0 += chunk) {
        const size_t p1 = std::min(n_pix, p0 + chunk);
        bp.p0 = (long)p0; bp.p1 = (long)p1;
        const long kc = (long)(2 * (p1 - p0));
        const int n_slices = (int)((p1 - p0 + kSyrkSlicePix - 1) / kSyrkSlicePix);
        unsigned long long* d_mask = nullptr; uint32_t *d_list = nullptr, *d_cnt = nullptr; uint16_t* d_range = nullptr;
        if ((st = ws_get(c, 32, (p1 - p0 + 8) * 2, (void**)&d_range))) return st;
        bp.range = d_range;
        if (sparse) {
            if ((st = ws_get(c, 3, (size_t)n_slices * 8, (void**)&d_mask)) || (st = ws_get(c, 4, (size_t)nbp * n_slices * 4, (void**)&d_list)) ||
                (st = ws_get(c, 5, (size_t)nbp * 4, (void**)&d_cnt)))
                return st;
            HIP_TRY(c, hipMemsetAsync(d_mask, 0, (size_t)n_slices * 8, s));
        }
        bp.slice_mask = d_mask;
        hipLaunchKernelGGL(emba_schur_build_kernel, dim3((unsigned)std::min<size_t>((p1 - p0 + kBuildWaves - 1) / kBuildWaves, 4096)), dim3(64 * kBuildWaves), lds_bytes, s, bp);
        int nks = (int)std::max<long>(1, std::min<long>(nks_max, kc / c->opt_syrk_min_cols));   // ... but >= 512 columns each (option syrk_min_cols): a block pays a fixed LDS combine + 32-KB slab write
        // ... and whole rounds of one block per CU: 268 blocks of 512 columns on 256 CUs (K = 21, 137 k columns) took 57 us, 255 blocks of 536 columns 44 us
        if ((long)nbp * nks > c->n_cu) nks = (int)std::max<long>(1, ((long)nbp * nks / c->n_cu) * c->n_cu / nbp);
        SyrkParams sp{};
        sp.A = d_U; sp.lda = lds_; sp.n = n; sp.k = kc; sp.C = d_S; sp.ldc = lds_; sp.slab = d_slab; sp.nbp = nbp; sp.range = d_range;
        // Where it pays: a BANDED U — a pixel is in view for the fraction fov / (path of the optical axis over the window) of the control poses, and an operand block
        // is read once per pair of its band.  With a dense band (every pixel sees the whole window: 1 s at any K) there is nothing to re-use that the lists form does
        // not get from its longer runs (measured: 10 M events / K = 97 over 1 s 3.14 vs 3.38 ms per solve, 1 M / K = 201 over 1 s 2.5 vs 2.75; config 2's shape, 10 s:
        // 4.15 vs 3.93).  Option syrk_lists: 0 auto, 1 the lists form always, 2 the item form always.
        double paz = 0.0, pel = 0.0, in_view = 1.0;
        if (axis_path(c, &paz, &pel)) in_view = std::min(1.0, (paz >= pel ? c->fov_x : c->fov_y) / std::max(std::max(paz, pel), 1e-9));
        const int band_blocks = std::min(nb64, (int)std::ceil(in_view * nb64) + 1);
        const bool items_form = sparse && (c->opt_syrk_lists == 2 || (c->opt_syrk_lists == 0 && 2 * band_blocks <= nb64));
        if (sparse) {
            if (!items_form || c->opt_solve_debug) hipLaunchKernelGGL(emba_syrk_lists_kernel, dim3((unsigned)nbp), dim3(64), 0, s, d_mask, n_slices, nbp, d_list, d_cnt);
            sp.list = d_list; sp.count = d_cnt; sp.n_slices = n_slices;
            nks = std::max(1, std::min(nks_max, n_slices));
        }
        // ITEM form of the block-sparse product (round 5, SyrkParams::items): workgroups = (block pair, chunk of slices) in chunk-major order.  Option syrk_lists = 1
        // keeps the round-3/4 form (a workgroup per (pair, part) walking every nks-th slice of the pair's list).
        int item_chunk = 0, n_item_chunks = 0;
        uint32_t* d_pair_items = nullptr; uint32_t* d_items = nullptr;
        const uint32_t item_cap = (uint32_t)c->opt_syrk_item_cap;      // slabs of the item form: items beyond them add to S by global atomics (option syrk_item_cap: 4096; tests force the overflow branch with 8)
        if (items_form) {
            const long pairs_est = (long)band_blocks * (band_blocks + 1) / 2;
            item_chunk = (int)std::min<long>(64, std::max<long>(4, ((long)n_slices * pairs_est * 3 / 2 + 2999) / 3000));
            n_item_chunks = (n_slices + item_chunk - 1) / item_chunk;
            if (n_item_chunks > 65535 || nbp > 65535) return fail(c, EMBA_ERR_CAPACITY, "too many slice chunks for the item form of the block-sparse product");
            double* d_slab2 = nullptr;
            if ((st = ws_get(c, 12, std::max<size_t>((size_t)nks_max * nbp, item_cap) * 4096 * 8, (void**)&d_slab2)) ||
                (st = ws_get(c, 37, (size_t)nbp * n_item_chunks * 4, (void**)&d_pair_items)) || (st = ws_get(c, 39, ((size_t)nbp * n_item_chunks + 2) * 4, (void**)&d_items)))
                return st;
            sp.slab = d_slab2;
        }
        sp.direct = (nks == 1);
        if (c->opt_solve_debug) {      // diagnostic: how sparse is this chunk?  (products = (block pair, slice) pairs the SYRK forms)
            (void)hipStreamSynchronize(s);
            std::vector<uint16_t> hr(p1 - p0); (void)hipMemcpy(hr.data(), d_range, (p1 - p0) * 2, hipMemcpyDeviceToHost);
            long w16 = 0, w64 = 0, hist[9] = {0};
            for (uint16_t r : hr) { const int lo = r & 255, hi = r >> 8; if (lo > hi) continue; w16 += hi - lo + 1; const int b = (hi >> 2) - (lo >> 2) + 1; w64 += b; hist[b < 8 ? b : 8]++; }
            long prod = 0, prod_diag = 0;
            if (sparse) { std::vector<uint32_t> hc(nbp); (void)hipMemcpy(hc.data(), d_cnt, nbp * 4, hipMemcpyDeviceToHost); for (int b = 0; b < nbp; ++b) { prod += hc[b]; int I, J; I = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5); while ((long)I * (I + 1) / 2 > b) --I; while ((long)(I + 1) * (I + 2) / 2 <= b) ++I; J = b - I * (I + 1) / 2; if (I == J) prod_diag += hc[b]; } }
            for (int R : {1, 2, 4, 8, 16}) {      // union band (16-row groups) of runs of R slices: what a band-resident product would have to hold
                const size_t per = (size_t)R * kSyrkSlicePix; long ng = 0, fit12 = 0, fit16 = 0, wsum = 0; long fit16_pix = 0;
                for (size_t g0 = 0; g0 < hr.size(); g0 += per) {
                    int lo = 999, hi = -1;
                    for (size_t i = g0; i < std::min(hr.size(), g0 + per); ++i) { const int l = hr[i] & 255, h = hr[i] >> 8; if (l > h) continue; lo = std::min(lo, l); hi = std::max(hi, h); }
                    if (hi < 0) continue;
                    ++ng; const int w = hi - lo + 1; wsum += w; if (w <= 12) ++fit12; if (w <= 16) { ++fit16; fit16_pix += (long)std::min(hr.size(), g0 + per) - (long)g0; }
                }
                fprintf(stderr, "[solve debug] runs of %2d slices: %ld runs, mean union band %.1f groups, <= 12 groups: %.1f %%, <= 16 groups: %.1f %% (%.1f %% of the pixels)\n", R, ng,
                        (double)wsum / std::max(1L, ng), 100.0 * fit12 / std::max(1L, ng), 100.0 * fit16 / std::max(1L, ng), 100.0 * fit16_pix / std::max<size_t>(1, hr.size()));
            }
            fprintf(stderr, "[solve debug] pixels %ld slices %d nbp %d nks %d: mean 16-row groups per pixel %.2f, mean 64-row blocks written %.2f (hist 1..8+: %ld %ld %ld %ld %ld %ld %ld %ld), products %ld (diag %ld) = %.2f per slice; U written %.1f MB\n",
                    (long)(p1 - p0), n_slices, nbp, nks, (double)w16 / (p1 - p0), (double)w64 / (p1 - p0), hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8], prod, prod_diag,
                    (double)prod / n_slices, w64 * 64.0 * 2 * 8 / 1e6);
        }
        if (items_form) {
            int32_t* d_ord = reinterpret_cast<int32_t*>(d_list);                  // (nbp * n_slices words >= nbp * n_item_chunks)
            uint32_t* d_n_items = d_items + (size_t)nbp * n_item_chunks;
            hipLaunchKernelGGL(emba_syrk_item_flag_kernel, dim3((unsigned)n_item_chunks), dim3(256), 0, s, d_mask, n_slices, nbp, item_chunk, d_ord);
            hipLaunchKernelGGL(emba_syrk_item_ord_kernel, dim3((unsigned)((nbp + 63) / 64)), dim3(64), 0, s, nbp, n_item_chunks, d_ord, d_cnt);
            hipLaunchKernelGGL(emba_syrk_item_list_kernel, dim3(1), dim3(1024), 0, s, d_ord, nbp, n_item_chunks, d_items, d_pair_items, d_n_items);
            sp.list = nullptr; sp.count = nullptr; sp.n_slices = n_slices; sp.direct = 0;
            sp.items = d_items; sp.n_items = d_n_items; sp.slice_mask = d_mask; sp.item_chunk = item_chunk; sp.item_cap = item_cap;
            hipLaunchKernelGGL(emba_syrk_kernel, dim3((unsigned)((size_t)nbp * n_item_chunks)), dim3(256), 0, s, sp);
            hipLaunchKernelGGL(emba_syrk_item_reduce_kernel, dim3((unsigned)(((size_t)nbp * 4096 + 255) / 256)), dim3(256), 0, s, sp.slab, d_pair_items, d_cnt, n_item_chunks, item_cap,
                               nbp, n, d_S, lds_);
            continue;
        }
        hipLaunchKernelGGL(emba_syrk_kernel, dim3(nbp, nks), dim3(256), 0, s, sp);
        if (nks > 1)
            hipLaunchKernelGGL(emba_syrk_reduce_kernel, dim3((unsigned)(((size_t)nbp * 4096 + 255) / 256), (unsigned)((nks + kSyrkReduceGroup - 1) / kSyrkReduceGroup)),
                               dim3(256), 0, s, d_slab, nks, nbp, n, d_S, lds_);
    }
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

// blocked Cholesky of S[skip:n, skip:n] and the two triangular solves: d_rhs <- x1 (zeros in front of skip).  The reference calls
// Eigen's ldlt (model.cpp:789).
emba_status schur_factor_solve(emba_ctx* c, double* d_S, long lds_, int n, int skip, double* d_rhs, int* d_info)
{
    // The right-hand side sits in S as row n (schur kernels: the augmented block).  The panel loop carries it along as one more row below
    // the matrix — its triangular solve and trailing updates ARE the forward substitution — so only L^T x = z is left for the solve kernel.
    hipStream_t s = c->stream;
    const int m = n - skip;
    double* Sm = d_S + (size_t)lds_ * skip + skip;
    if (m + 1 <= 64 && skip <= 64) {                         // K <= 21: factor + both substitutions in one launch
        hipLaunchKernelGGL(emba_chol_small_kernel, dim3(1), dim3(256), 0, s, Sm, lds_, m, skip, d_rhs, d_info);
        HIP_TRY(c, hipGetLastError());
        return EMBA_OK;
    }
    // per panel: [diagonal factor — a launch of its own for the first panel only] / panel solve / trailing update + the next panel's diagonal factor
    hipLaunchKernelGGL(emba_chol_diag_kernel, dim3(1), dim3(256), 0, s, Sm, lds_, 0, std::min(64, m), d_info);
    for (int jb = 0; jb < m; jb += 64) {
        const int nb = std::min(64, m - jb);
        const int below = m - jb - nb;                      // matrix rows under the panel; the rhs row (index m) comes on top of them
        hipLaunchKernelGGL(emba_chol_trsm_kernel, dim3((below + 1 + 4 * kTrsmRows - 1) / (4 * kTrsmRows)), dim3(256), 0, s, Sm, lds_, m + 1, jb, nb);
        if (below > 0) {
            const int tb = (below + 1 + 63) / 64;
            hipLaunchKernelGGL(emba_chol_trail_kernel, dim3(tb * (tb + 1) / 2), dim3(256), 0, s, Sm, lds_, jb, nb, below + 1, std::min(64, below), d_info);
        }
    }
    hipLaunchKernelGGL(emba_schur_rhs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_S, lds_, n, skip, d_rhs);   // z = row n of the factor
    hipLaunchKernelGGL(emba_chol_trsv_kernel, dim3(1), dim3(kTrsvThreads), 0, s, Sm, lds_, m, d_rhs + skip);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

// The column order of U for the local solve (emba_perm_keys_kernel): panorama columns first when the camera mostly pans (azimuth path of the optical axis
// over the control poses >= its elevation path), the compact order (nullptr) otherwise.  Built once per list build; workspaces 34-36.
// azimuth / elevation path length (rad) of the optical axis over the control poses of the last evaluation (the host's pinned copy)
bool axis_path(const emba_ctx* c, double* path_az_out, double* path_el_out)
{
    if (!c->h_knots || c->K < 2) return false;
    double path_az = 0.0, path_el = 0.0, az0 = 0.0, el0 = 0.0;
    for (int i = 0; i < c->K; ++i) {
        const double x = c->h_knots[4 * i], y = c->h_knots[4 * i + 1], z = c->h_knots[4 * i + 2], w = c->h_knots[4 * i + 3];
        const double ax = 2.0 * (x * z + w * y), ay = 2.0 * (y * z - w * x), az_ = 1.0 - 2.0 * (x * x + y * y);      // R (0, 0, 1)
        const double az = atan2(ax, az_), el = asin(std::max(-1.0, std::min(1.0, ay)));
        if (i) { double d = az - az0; while (d > M_PI) d -= 2.0 * M_PI; while (d < -M_PI) d += 2.0 * M_PI; path_az += fabs(d); path_el += fabs(el - el0); }
        az0 = az; el0 = el;
    }
    *path_az_out = path_az; *path_el_out = path_el;
    return true;
}

emba_status solve_perm(emba_ctx* c, size_t P, const uint32_t** perm, size_t lo = 0)
{   // the pixels [lo, lo + P) of the active set (round 6: a rank's owned range in the sharded solve too — without the order its half of the product took 1.45 ms where
    // the whole single-rank product takes 0.96); perm[j] is relative to lo
    *perm = nullptr;
    // Measured (device time of one solve, with / without; the U build reads its pixels' records out of sequence and pays 5 % for it): config 2's shape (K = 201,
    // 10 M events over 10 s) 3.96 / 4.16 ms — SYRK 1.25 / 1.52, 5.8 / 7.9 products per slice —; 10 M events at K = 97: 2.81 / 2.79; 1 M events over 1 s at
    // K = 201 (every pixel sees the whole window: nothing to gain) 2.27 / 2.24.  From six row blocks (K >= 128) up, unless option solve_perm forces it.
    if (c->solve_perm_mode == 0 || P < 4 * (size_t)kSyrkSlicePix || (c->solve_perm_mode < 0 && 3 * c->K < 384) || 3 * c->K < 256) return EMBA_OK;
    if (c->perm_valid && c->perm_lo == lo && c->perm_n == P) { *perm = c->d_perm; return EMBA_OK; }
    if (c->solve_perm_mode < 0) {
        double path_az = 0.0, path_el = 0.0;
        if (!axis_path(c, &path_az, &path_el)) return EMBA_OK;
        if (path_az < path_el) return EMBA_OK;      // mostly tilting: a panorama row is the better slice already
    }
    hipStream_t s = c->stream;
    uint32_t *d_cnt = nullptr, *d_tick = nullptr, *d_pm = nullptr;
    emba_status st;
    if ((st = ws_get(c, 34, ((size_t)c->W + 2) * 4, (void**)&d_cnt)) || (st = ws_get(c, 35, P * 4, (void**)&d_tick)) || (st = ws_get(c, 36, P * 4, (void**)&d_pm))) return st;
    HIP_TRY(c, hipMemsetAsync(d_cnt, 0, ((size_t)c->W + 1) * 4, s));
    hipLaunchKernelGGL(emba_perm_ticket_kernel, dim3(nblocks(P)), dim3(256), 0, s, c->d_active + lo, (long)P, c->W, d_cnt, d_tick);
    if ((st = dev_scan(c, d_cnt, d_cnt, (size_t)c->W, nullptr))) return st;
    hipLaunchKernelGGL(emba_perm_place_kernel, dim3(nblocks(P)), dim3(256), 0, s, c->d_active + lo, (long)P, c->W, d_cnt, d_tick, d_pm);
    HIP_TRY(c, hipGetLastError());
    uint32_t* v0 = d_pm;
    c->d_perm = v0; c->perm_valid = true; c->perm_lo = lo; c->perm_n = P;
    *perm = v0;
    return EMBA_OK;
}

RecView local_view(emba_ctx* c)
{
    RecView v{};
    v.rec = c->d_rec; v.slot_key = c->d_slot_key; v.compact = c->d_compact; v.stamp = c->set_stamp; v.packed = 0; v.pix_base = 0;
    return v;
}

}  // namespace

extern "C" emba_status emba_solve_normal_eq(emba_ctx* c, double lambda, int32_t fix_first_pose, double* x1_host, double* x2_host)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->finish_done) return fail(c, EMBA_ERR_STATE, "solveNormalEq needs formNormalEq + applyL2Reg (emba_form_finish) first");
    if (c->eq_in_alt) return fail(c, EMBA_ERR_STATE, "an evaluation has been written since these equations were formed (its records are the working set): report the LM decision first — emba_map_reject / emba_trial_reject to go back to them, or form the new ones");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st = resolve_pending(c);
    if (st) return st;
    hipStream_t s = c->stream;
    const int n = 3 * c->K;
    const int skip = fix_first_pose ? 3 : 0;
    const size_t P = c->P, M = c->n_cand;
    if (n - skip <= 0) return fail(c, EMBA_ERR_INVALID_ARG, "nothing to solve for");
    const int na = n + 1;                              // augmented: row n carries y / the right-hand side
    const long lds_ = (na + 15) / 16 * 16;             // leading dimension of S_aug and of U
    double *d_S = nullptr, *d_rhs = nullptr, *d_y = nullptr, *d_cf = nullptr, *d_x2 = nullptr;
    int* d_info = nullptr;
    if ((st = ws_get(c, 6, (size_t)lds_ * na * 8, (void**)&d_S)) || (st = ws_get(c, 7, (size_t)n * 8, (void**)&d_rhs)) || (st = ws_get(c, 9, 2 * (P + 1) * 8, (void**)&d_y)) ||
        (st = ws_get(c, 10, 3 * (P + 1) * 8, (void**)&d_cf)) || (st = ws_get(c, 11, 2 * (P + 1) * 8, (void**)&d_x2)) || (st = ws_get(c, 13, 4, (void**)&d_info)))
        return st;
    if ((st = ensure_compact(c))) return st;
    HIP_TRY(c, hipMemsetAsync(d_info, 0, sizeof(int), s));
    HIP_TRY(c, hipMemsetAsync(d_S, 0, (size_t)lds_ * na * sizeof(double), s));
    const RecView view = local_view(c);
    SolveLists L;
    if ((st = build_lists(c, view, M, P, &L))) return st;
    hipLaunchKernelGGL(emba_schur_init_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, pack_A11(c), pack_b1(c), n, lambda, d_S, lds_);
    const uint32_t* perm = nullptr;
    if ((st = solve_perm(c, P, &perm))) return st;
    if ((st = schur_accumulate(c, L.sorted, L, P, pack_A22b2(c), lambda, n, d_S, lds_, d_y, d_cf, d_info, perm))) return st;
    if ((st = schur_factor_solve(c, d_S, lds_, n, skip, d_rhs, d_info))) return st;
    // x2 = A22m^-1 (b2 - A12^T x1), straight from the records of each pixel
    if (P) {
        if ((st = pose_vector_lds(c, (const void*)emba_schur_x2_kernel, (size_t)n * sizeof(double), "solveNormalEq (x2)"))) return st;
        hipLaunchKernelGGL(emba_schur_x2_kernel, dim3((unsigned)std::min<size_t>((P + 3) / 4, 8192)), dim3(256), (size_t)n * sizeof(double), s, L.sorted, L.off, d_y,
                           d_cf, d_rhs, c->irls, c->eta, (long)P, d_x2, n);
    }
    HIP_TRY(c, hipGetLastError());
    int info = 0;
    HIP_TRY(c, hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    if (x1_host) HIP_TRY(c, hipMemcpyAsync(x1_host, d_rhs, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (x2_host && P) HIP_TRY(c, hipMemcpyAsync(x2_host, d_x2, 2 * P * sizeof(double), hipMemcpyDeviceToHost, s));
    if ((st = keep_x2(c, d_x2, P))) return st;
    HIP_TRY(c, hipStreamSynchronize(s));
    c->solve_info = info;
    // bit 0: a 2x2 block A22m_i is not positive definite — the reference's A22m_i.inverse() (model.cpp:750) returns inf / nan there and the
    // step is lost (NaN cost: rejected); reported as EMBA_ERR_NUMERIC.  bit 1: a pivot of S vanished — handled like Eigen's ldlt (zero
    // update in that component), not an error.
    if (info & 1) return fail(c, EMBA_ERR_NUMERIC, "a 2x2 block of A22 + lambda*diag(A22) is not positive definite (the reference's inverse() gives inf/nan)");
    if (info & 4) return fail(c, EMBA_ERR_NUMERIC, "the Schur complement is indefinite or not finite (a pivot clearly below zero, or NaN): no update is returned");
    return EMBA_OK;
}

extern "C" emba_status emba_last_solve_info(const emba_ctx* c, int32_t* info)
{
    if (!c || !info) return EMBA_ERR_INVALID_ARG;
    *info = c->solve_info;
    return EMBA_OK;
}

// ---- sharded Schur solve (f1 on N GPUs): count / pack / [all-to-all] / partial / [all-reduce] / finish ----------------------------
extern "C" emba_status emba_solve_shard_size(emba_ctx* c, size_t* s_doubles)
{
    if (!c || !s_doubles) return EMBA_ERR_INVALID_ARG;
    const int na = 3 * c->K + 1;
    const long lds_ = (na + 15) / 16 * 16;
    *s_doubles = (size_t)lds_ * na;
    return EMBA_OK;
}

extern "C" emba_status emba_solve_shard_count(emba_ctx* c, int32_t n_ranks, size_t* counts_host)
{
    if (!c || !counts_host || n_ranks < 1 || n_ranks > 1024) return c ? fail(c, EMBA_ERR_INVALID_ARG, "solve_shard_count: bad arguments") : EMBA_ERR_INVALID_ARG;
    if (!c->finish_done) return fail(c, EMBA_ERR_STATE, "the sharded solve needs formNormalEq + applyL2Reg (emba_form_finish) first");
    if (c->eq_in_alt) return fail(c, EMBA_ERR_STATE, "an evaluation has been written since these equations were formed (its records are the working set): report the LM decision first — emba_map_reject / emba_trial_reject to go back to them, or form the new ones");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st = resolve_pending(c);
    if (st) return st;
    if ((st = ensure_compact(c))) return st;
    hipStream_t s = c->stream;
    unsigned long long* d_cnt = nullptr;
    if ((st = ws_get(c, 14, (size_t)3 * n_ranks * 8 + 8, (void**)&d_cnt))) return st;
    HIP_TRY(c, hipMemsetAsync(d_cnt, 0, (size_t)3 * n_ranks * 8 + 8, s));
    if (c->n_cand) hipLaunchKernelGGL(emba_shard_count_kernel, dim3((unsigned)std::min<size_t>(nblocks(c->n_cand), (size_t)4 * c->n_cu)), dim3(256), 0, s, local_view(c), (long)c->n_cand, (long)c->P, (int)n_ranks, d_cnt);
    std::vector<unsigned long long> h(n_ranks);
    HIP_TRY(c, hipMemcpyAsync(h.data(), d_cnt, (size_t)n_ranks * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    unsigned long long run = 0;
    std::vector<unsigned long long> off(n_ranks);
    for (int r = 0; r < n_ranks; ++r) { counts_host[r] = (size_t)h[r]; off[r] = run; run += h[r]; }
    HIP_TRY(c, hipMemcpyAsync(d_cnt + n_ranks, off.data(), (size_t)n_ranks * 8, hipMemcpyHostToDevice, s));   // [n, 2n): offsets; [2n, 3n): cursors (zero)
    HIP_TRY(c, hipStreamSynchronize(s));
    return EMBA_OK;
}

extern "C" emba_status emba_solve_shard_pack(emba_ctx* c, int32_t n_ranks, double* send_dev)
{
    if (!c || n_ranks < 1 || n_ranks > kShardMaxRanks) return EMBA_ERR_INVALID_ARG;
    if (!c->ws[14].p) return fail(c, EMBA_ERR_STATE, "call emba_solve_shard_count first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    unsigned long long* d_cnt = (unsigned long long*)c->ws[14].p;
    HIP_TRY(c, hipMemsetAsync(d_cnt + 2 * n_ranks, 0, (size_t)n_ranks * 8, s));
    if (c->n_cand && send_dev)
        hipLaunchKernelGGL(emba_shard_pack_kernel, dim3(nblocks(c->n_cand, kShardPackRec)), dim3(256), 0, s, local_view(c), (long)c->n_cand, (long)c->P, (int)n_ranks, d_cnt + n_ranks,
                           d_cnt + 2 * n_ranks, send_dev);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

namespace {
void shard_range(size_t P, int rank, int n_ranks, size_t* lo, size_t* hi) { *lo = (P * (size_t)rank) / n_ranks; *hi = (P * ((size_t)rank + 1)) / n_ranks; }
}

// Does this rank still hold, in pixel order, the records it received as the owner of its pixel range for the CURRENT equations (a re-solve with another lambda after a
// rejected trial, solver.cpp:340-352)?  Then emba_solve_shard_count / _pack and the all-to-all can be skipped and emba_solve_shard_partial / _finish take recv_dev = NULL.
extern "C" emba_status emba_solve_shard_cached(emba_ctx* c, int32_t rank, int32_t n_ranks, int32_t* cached, size_t* n_recv)
{
    if (!c || !cached || rank < 0 || rank >= n_ranks) return c ? fail(c, EMBA_ERR_INVALID_ARG, "solve_shard_cached: bad arguments") : EMBA_ERR_INVALID_ARG;
    size_t lo, hi;
    shard_range(c->P, rank, n_ranks, &lo, &hi);
    *cached = (c->finish_done && !c->eq_in_alt && c->lists_valid && c->lists_packed && c->lists_stamp == c->set_stamp && c->lists_P == hi - lo && c->lists_lo == (long)lo &&
               c->ws[0].p && c->ws[33].p && c->ws[9].p) ? 1 : 0;
    if (n_recv) *n_recv = *cached ? c->lists_nrec : 0;
    return EMBA_OK;
}

extern "C" emba_status emba_solve_shard_partial(emba_ctx* c, int32_t rank, int32_t n_ranks, const double* recv_dev, size_t n_recv, double lambda, double* S_part_dev)
{
    if (!c || !S_part_dev || rank < 0 || rank >= n_ranks) return c ? fail(c, EMBA_ERR_INVALID_ARG, "solve_shard_partial: bad arguments") : EMBA_ERR_INVALID_ARG;
    if (!c->finish_done) return fail(c, EMBA_ERR_STATE, "the sharded solve needs formNormalEq + applyL2Reg (emba_form_finish) first");
    if (c->eq_in_alt) return fail(c, EMBA_ERR_STATE, "an evaluation has been written since these equations were formed (its records are the working set): report the LM decision first — emba_map_reject / emba_trial_reject to go back to them, or form the new ones");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = 3 * c->K, na = n + 1;
    const long lds_ = (na + 15) / 16 * 16;
    size_t lo, hi;
    shard_range(c->P, rank, n_ranks, &lo, &hi);
    const size_t n_pix = hi - lo;
    double *d_y = nullptr, *d_cf = nullptr; int* d_info = nullptr;
    emba_status st;
    if ((st = ws_get(c, 9, 2 * (n_pix + 1) * 8, (void**)&d_y)) || (st = ws_get(c, 10, 3 * (n_pix + 1) * 8, (void**)&d_cf)) || (st = ws_get(c, 13, 64, (void**)&d_info))) return st;
    HIP_TRY(c, hipMemsetAsync(d_info, 0, sizeof(int), s));
    HIP_TRY(c, hipMemsetAsync(S_part_dev, 0, (size_t)lds_ * na * sizeof(double), s));
    RecView view{};
    view.rec = recv_dev; view.packed = 1; view.pix_base = (long)lo;
    SolveLists L;
    if ((st = build_lists(c, view, n_recv, n_pix, &L))) return st;
    const uint32_t* perm = nullptr;
    if ((st = solve_perm(c, n_pix, &perm, lo))) return st;
    return schur_accumulate(c, L.sorted, L, n_pix, pack_A22b2(c) + 5 * lo, lambda, n, S_part_dev, lds_, d_y, d_cf, d_info, perm);
}

extern "C" emba_status emba_solve_shard_finish(emba_ctx* c, int32_t rank, int32_t n_ranks, const double* recv_dev, size_t n_recv, double lambda,
                                               int32_t fix_first_pose, double* S_dev, double* x1_host, double* x2_full_dev)
{
    if (!c || !S_dev || rank < 0 || rank >= n_ranks) return c ? fail(c, EMBA_ERR_INVALID_ARG, "solve_shard_finish: bad arguments") : EMBA_ERR_INVALID_ARG;
    if (!c->ws[0].p || !c->ws[9].p) return fail(c, EMBA_ERR_STATE, "call emba_solve_shard_partial first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = 3 * c->K, na = n + 1, skip = fix_first_pose ? 3 : 0;
    const long lds_ = (na + 15) / 16 * 16;
    size_t lo, hi;
    shard_range(c->P, rank, n_ranks, &lo, &hi);
    const size_t n_pix = hi - lo;
    double* d_rhs = nullptr;
    emba_status st;
    if ((st = ws_get(c, 7, (size_t)n * 8, (void**)&d_rhs))) return st;
    int* d_info = (int*)c->ws[13].p;
    double* d_y = (double*)c->ws[9].p; double* d_cf = (double*)c->ws[10].p;
    hipLaunchKernelGGL(emba_schur_add_a11_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, pack_A11(c), pack_b1(c), n, lambda, S_dev, lds_);
    if ((st = schur_factor_solve(c, S_dev, lds_, n, skip, d_rhs, d_info))) return st;
    if (x2_full_dev) {
        HIP_TRY(c, hipMemsetAsync(x2_full_dev, 0, 2 * c->P * sizeof(double), s));
        RecView view{};       // the received records in pixel order, as emba_solve_shard_partial's build_lists left them
        view.rec = (const double*)c->ws[33].p; view.packed = 1; view.pix_base = 0;
        if ((st = pose_vector_lds(c, (const void*)emba_schur_x2_kernel, (size_t)n * sizeof(double), "solveNormalEq (x2)"))) return st;
        if (n_pix)
            hipLaunchKernelGGL(emba_schur_x2_kernel, dim3((unsigned)std::min<size_t>((n_pix + 3) / 4, 8192)), dim3(256), (size_t)n * sizeof(double), s, view, (const uint32_t*)c->ws[0].p,
                               d_y, d_cf, d_rhs, c->irls, c->eta, (long)n_pix, x2_full_dev + 2 * lo, n);
    }
    HIP_TRY(c, hipGetLastError());
    int info = 0;
    HIP_TRY(c, hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    if (x1_host) HIP_TRY(c, hipMemcpyAsync(x1_host, d_rhs, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->solve_info = info;
    if (info & 1) return fail(c, EMBA_ERR_NUMERIC, "a 2x2 block of A22 + lambda*diag(A22) is not positive definite (the reference's inverse() gives inf/nan)");
    if (info & 4) return fail(c, EMBA_ERR_NUMERIC, "the Schur complement is indefinite or not finite (a pivot clearly below zero, or NaN): no update is returned");
    return EMBA_OK;
}

// LEGM::solveNormalEqCG (model.cpp:794-840)
extern "C" emba_status emba_solve_normal_eq_cg(emba_ctx* c, double lambda, int32_t fix_first_pose, int32_t max_iter, double tol, double* x1_host,
                                               double* x2_host, int32_t* iterations, double* error)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if (!c->finish_done) return fail(c, EMBA_ERR_STATE, "solveNormalEqCG needs formNormalEq + applyL2Reg (emba_form_finish) first");
    if (c->eq_in_alt) return fail(c, EMBA_ERR_STATE, "an evaluation has been written since these equations were formed (its records are the working set): report the LM decision first — emba_map_reject / emba_trial_reject to go back to them, or form the new ones");
    HIP_TRY(c, hipSetDevice(c->device));
    emba_status st = resolve_pending(c);
    if (st) return st;
    hipStream_t s = c->stream;
    const int n = 3 * c->K, skip = fix_first_pose ? 3 : 0;
    const size_t P = c->P, M = c->n_cand, N = (size_t)n + 2 * P;
    if (max_iter <= 0) max_iter = 100;     // model.cpp:823-824
    if (!(tol > 0)) tol = 1e-6;
    double *d_x = nullptr, *d_r = nullptr, *d_p = nullptr, *d_z = nullptr, *d_t = nullptr, *d_invd = nullptr, *d_sc = nullptr;
    if ((st = ws_get(c, 6, N * 8, (void**)&d_x)) || (st = ws_get(c, 7, N * 8, (void**)&d_r)) || (st = ws_get(c, 8, N * 8, (void**)&d_p)) ||
        (st = ws_get(c, 9, N * 8, (void**)&d_z)) || (st = ws_get(c, 10, N * 8, (void**)&d_t)) || (st = ws_get(c, 11, N * 8, (void**)&d_invd)) ||
        (st = ws_get(c, 13, 64, (void**)&d_sc)))
        return st;
    if ((st = ensure_compact(c))) return st;
    const RecView view = local_view(c);
    SolveLists L;
    if ((st = build_lists(c, view, M, P, &L))) return st;
    const unsigned gridN = (unsigned)std::min<size_t>(nblocks(N), 1024);
    auto read2 = [&](double* a, double* b) -> emba_status {   // device scalars d_sc[0], d_sc[1] -> host, then cleared
        double h[2] = {0, 0};
        HIP_TRY(c, hipMemcpyAsync(h, d_sc, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        HIP_TRY(c, hipMemsetAsync(d_sc, 0, 16, s));
        if (a) *a = h[0];
        if (b) *b = h[1];
        return EMBA_OK;
    };
    if ((st = pose_vector_lds(c, (const void*)emba_cg_pixel_kernel, (size_t)n * 16, "solveNormalEqCG", 48))) return st;
    CgPixParams pp{};
    pp.view = L.sorted; pp.off = L.off; pp.A22b2 = pack_A22b2(c); pp.lambda = lambda; pp.irls = c->irls; pp.eta = c->eta; pp.n = n; pp.skip = skip;
    pp.P = (long)P;
    auto apply = [&](const double* v, double* y) {   // y = [A11m A12; A12^T A22m] v
        hipLaunchKernelGGL(emba_cg_a11_kernel, dim3((n + 3) / 4), dim3(256), 0, s, pack_A11(c), n, lambda, skip, v, y);
        pp.v = v; pp.y = y;
        if (P) hipLaunchKernelGGL(emba_cg_pixel_kernel, dim3((unsigned)std::min<size_t>((P + 3) / 4, (size_t)8 * c->n_cu)), dim3(256), (size_t)n * 16, s, pp);
    };
    // Eigen/src/IterativeLinearSolvers/ConjugateGradient.h:28-88 (zero initial guess: residual = rhs)
    HIP_TRY(c, hipMemsetAsync(d_sc, 0, 64, s));
    hipLaunchKernelGGL(emba_cg_init_kernel, dim3(gridN), dim3(256), 0, s, pack_A11(c), pack_b1(c), pack_A22b2(c), n, skip, (long)P, lambda, d_x, d_r, d_p, d_invd, d_sc);
    double rhs2 = 0, absNew = 0;
    if ((st = read2(&rhs2, &absNew))) return st;
    int it = 0;
    double err = 0;
    if (rhs2 != 0) {
        const double thr = std::max(tol * tol * rhs2, std::numeric_limits<double>::min());
        double rn2 = rhs2;
        if (rn2 >= thr) {
            while (it < max_iter) {
                apply(d_p, d_t);
                hipLaunchKernelGGL(emba_cg_dot_kernel, dim3(gridN), dim3(256), 0, s, d_p, d_t, (long)N, d_sc);
                double pt = 0;
                if ((st = read2(&pt, nullptr))) return st;
                const double alpha = absNew / pt;
                hipLaunchKernelGGL(emba_cg_xr_kernel, dim3(gridN), dim3(256), 0, s, alpha, d_p, d_t, (long)N, d_x, d_r, d_sc);
                hipLaunchKernelGGL(emba_cg_z_kernel, dim3(gridN), dim3(256), 0, s, d_invd, d_r, (long)N, d_z, d_sc + 1);
                double absNext = 0;
                if ((st = read2(&rn2, &absNext))) return st;
                if (rn2 < thr) break;
                const double beta = absNext / absNew;
                absNew = absNext;
                hipLaunchKernelGGL(emba_cg_p_kernel, dim3(nblocks(N)), dim3(256), 0, s, beta, d_z, (long)N, d_p);
                ++it;
            }
        }
        err = std::sqrt(rn2 / rhs2);
    } else {
        HIP_TRY(c, hipMemsetAsync(d_x, 0, N * 8, s));
    }
    HIP_TRY(c, hipGetLastError());
    if (x1_host) HIP_TRY(c, hipMemcpyAsync(x1_host, d_x, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (x2_host && P) HIP_TRY(c, hipMemcpyAsync(x2_host, d_x + n, 2 * P * sizeof(double), hipMemcpyDeviceToHost, s));
    if ((st = keep_x2(c, d_x + n, P))) return st;
    HIP_TRY(c, hipStreamSynchronize(s));
    if (iterations) *iterations = it;
    if (error) *error = err;
    return EMBA_OK;
}

// ---- LEGM::solveNormalEqCG over a sharded window (round 6, VERDICT r5 missing #3: a launch file with use_CG = true on several GPUs, solver.cpp:190-202) --------------
// The system [A11m A12; A12^T A22m] is applied matrix-free with the PIXELS sharded: rank r owns the active pixels [P r / n, P (r+1) / n) and — after the same
// record exchange as the sharded Schur solve (cached across re-solves) — every record of those pixels.  A CG vector is [pose part, 3K, REPLICATED | this rank's
// 2 x (its pixels) map entries]; the map part of M v is local, the pose part is a sum over the ranks' pixels: ONE all-reduce of 3K + 2 doubles per application
// (the rank's share of t1 and of p.t), one of 2 doubles per iteration for r.r and r.z.  Every rank takes the same scalars from the reduced sums, so the
// replicated pose parts stay bit-identical.  The loop itself (Eigen's ConjugateGradient.h:28-88) is driven by the host: emba_group_solve_cg, emba_amd/sharded.py.
//   begin     lists of the owned pixels' records, x = 0, r = b, p = invd r;  red[n] = r.r, red[n+1] = r.p (partial)      -> all-reduce red (n + 2)
//   apply     t = M p on this rank's pixels;  red[0..n) = its share of t1 (rank 0 adds A11m p1), red[n] = p2.t2          -> all-reduce red (n + 2)
//   pt        t1 = red[0..n);  *pt = p1.t1 + red[n]
//   update    x += alpha p, r -= alpha t, z = invd r;  red[n] = r.r, red[n+1] = r.z (partial)                             -> all-reduce red + n (2)
//   direction p = z + beta p
//   end       x1 -> host, this rank's x2 into x2_full_dev (zeros elsewhere)                                               -> all-reduce x2_full (2P)
// Partial sums count the replicated pose part on rank 0 only.
extern "C" emba_status emba_cg_shard_size(emba_ctx* c, size_t* red_doubles)
{
    if (!c || !red_doubles) return EMBA_ERR_INVALID_ARG;
    *red_doubles = (size_t)3 * c->K + 2;
    return EMBA_OK;
}

extern "C" emba_status emba_cg_shard_begin(emba_ctx* c, int32_t rank, int32_t n_ranks, const double* recv_dev, size_t n_recv, double lambda, int32_t fix_first_pose,
                                           double* red_dev)
{
    if (!c || !red_dev || rank < 0 || rank >= n_ranks) return c ? fail(c, EMBA_ERR_INVALID_ARG, "cg_shard_begin: bad arguments") : EMBA_ERR_INVALID_ARG;
    if (!c->finish_done) return fail(c, EMBA_ERR_STATE, "solveNormalEqCG needs formNormalEq + applyL2Reg (emba_form_finish) first");
    if (c->eq_in_alt) return fail(c, EMBA_ERR_STATE, "an evaluation has been written since these equations were formed: report the LM decision first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    auto& g = c->cg;
    g.active = false;
    g.rank = rank; g.n_ranks = n_ranks; g.n = 3 * c->K; g.skip = fix_first_pose ? 3 : 0; g.lambda = lambda;
    size_t lo, hi;
    shard_range(c->P, rank, n_ranks, &lo, &hi);
    g.lo = lo; g.npix = hi - lo; g.Nl = (size_t)g.n + 2 * g.npix;
    emba_status st;
    if ((st = ws_get(c, 6, g.Nl * 8, (void**)&g.x)) || (st = ws_get(c, 7, g.Nl * 8, (void**)&g.r)) || (st = ws_get(c, 8, g.Nl * 8, (void**)&g.p)) ||
        (st = ws_get(c, 9, g.Nl * 8, (void**)&g.z)) || (st = ws_get(c, 10, g.Nl * 8, (void**)&g.t)) || (st = ws_get(c, 11, g.Nl * 8, (void**)&g.invd)) ||
        (st = ws_get(c, 13, 64, (void**)&g.sc)))
        return st;
    RecView view{};
    view.rec = recv_dev; view.packed = 1; view.pix_base = (long)lo;
    if ((st = build_lists(c, view, n_recv, g.npix, &g.L))) return st;
    if ((st = pose_vector_lds(c, (const void*)emba_cg_pixel_kernel, (size_t)g.n * 16, "solveNormalEqCG", 48))) return st;
    HIP_TRY(c, hipMemsetAsync(red_dev, 0, ((size_t)g.n + 2) * 8, s));
    const unsigned grid = (unsigned)std::min<size_t>(nblocks(g.Nl), 1024);
    hipLaunchKernelGGL(emba_cg_init_kernel, dim3(grid), dim3(256), 0, s, pack_A11(c), pack_b1(c), pack_A22b2(c) + 5 * lo, g.n, g.skip, (long)g.npix, lambda, g.x, g.r, g.p, g.invd,
                       red_dev + g.n, (long)(rank == 0 ? 0 : g.n));
    HIP_TRY(c, hipGetLastError());
    g.active = true;
    return EMBA_OK;
}

extern "C" emba_status emba_cg_shard_apply(emba_ctx* c, double* red_dev)
{
    if (!c || !red_dev) return EMBA_ERR_INVALID_ARG;
    auto& g = c->cg;
    if (!g.active) return fail(c, EMBA_ERR_STATE, "call emba_cg_shard_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = g.n;
    if (g.rank == 0) hipLaunchKernelGGL(emba_cg_a11_kernel, dim3((n + 3) / 4), dim3(256), 0, s, pack_A11(c), n, g.lambda, g.skip, g.p, g.t);
    else HIP_TRY(c, hipMemsetAsync(g.t, 0, (size_t)n * 8, s));
    CgPixParams pp{};
    pp.view = g.L.sorted; pp.off = g.L.off; pp.A22b2 = pack_A22b2(c) + 5 * g.lo; pp.lambda = g.lambda; pp.irls = c->irls; pp.eta = c->eta; pp.n = n; pp.skip = g.skip;
    pp.P = (long)g.npix; pp.v = g.p; pp.y = g.t;
    if (g.npix) hipLaunchKernelGGL(emba_cg_pixel_kernel, dim3((unsigned)std::min<size_t>((g.npix + 3) / 4, (size_t)8 * c->n_cu)), dim3(256), (size_t)n * 16, s, pp);
    HIP_TRY(c, hipMemcpyAsync(red_dev, g.t, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemsetAsync(red_dev + n, 0, 16, s));
    if (g.npix) hipLaunchKernelGGL(emba_cg_dot_kernel, dim3((unsigned)std::min<size_t>(nblocks(2 * g.npix), 1024)), dim3(256), 0, s, g.p + n, g.t + n, (long)(2 * g.npix), red_dev + n);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

extern "C" emba_status emba_cg_shard_pt(emba_ctx* c, const double* red_dev, double* pt)
{
    if (!c || !red_dev || !pt) return EMBA_ERR_INVALID_ARG;
    auto& g = c->cg;
    if (!g.active) return fail(c, EMBA_ERR_STATE, "call emba_cg_shard_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = g.n;
    HIP_TRY(c, hipMemcpyAsync(g.t, red_dev, (size_t)n * 8, hipMemcpyDeviceToDevice, s));      // the reduced t1, identical on every rank
    HIP_TRY(c, hipMemsetAsync(g.sc, 0, 16, s));
    hipLaunchKernelGGL(emba_cg_dot_kernel, dim3(1), dim3(256), 0, s, g.p, g.t, (long)n, g.sc);     // (one block: the same summation order on every rank)
    double h[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&h[0], g.sc, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&h[1], red_dev + n, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *pt = h[0] + h[1];
    return EMBA_OK;
}

extern "C" emba_status emba_cg_shard_update(emba_ctx* c, double alpha, double* red_dev)
{
    if (!c || !red_dev) return EMBA_ERR_INVALID_ARG;
    auto& g = c->cg;
    if (!g.active) return fail(c, EMBA_ERR_STATE, "call emba_cg_shard_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const long from = g.rank == 0 ? 0 : g.n;
    const unsigned grid = (unsigned)std::min<size_t>(nblocks(g.Nl), 1024);
    HIP_TRY(c, hipMemsetAsync(red_dev + g.n, 0, 16, s));
    hipLaunchKernelGGL(emba_cg_xr_kernel, dim3(grid), dim3(256), 0, s, alpha, g.p, g.t, (long)g.Nl, g.x, g.r, red_dev + g.n, from);
    hipLaunchKernelGGL(emba_cg_z_kernel, dim3(grid), dim3(256), 0, s, g.invd, g.r, (long)g.Nl, g.z, red_dev + g.n + 1, from);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

extern "C" emba_status emba_cg_shard_direction(emba_ctx* c, double beta)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    auto& g = c->cg;
    if (!g.active) return fail(c, EMBA_ERR_STATE, "call emba_cg_shard_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(emba_cg_p_kernel, dim3(nblocks(g.Nl)), dim3(256), 0, c->stream, beta, g.z, (long)g.Nl, g.p);
    HIP_TRY(c, hipGetLastError());
    return EMBA_OK;
}

extern "C" emba_status emba_cg_shard_end(emba_ctx* c, double* x1_host, double* x2_full_dev)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    auto& g = c->cg;
    if (!g.active) return fail(c, EMBA_ERR_STATE, "call emba_cg_shard_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (x2_full_dev) {
        HIP_TRY(c, hipMemsetAsync(x2_full_dev, 0, 2 * c->P * sizeof(double), s));
        if (g.npix) HIP_TRY(c, hipMemcpyAsync(x2_full_dev + 2 * g.lo, g.x + g.n, 2 * g.npix * 8, hipMemcpyDeviceToDevice, s));
    }
    if (x1_host) HIP_TRY(c, hipMemcpyAsync(x1_host, g.x, (size_t)g.n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    g.active = false;
    return EMBA_OK;
}

// ---- f3: intensity panorama from the gradient map ---------------------------------------------------------------------------
extern "C" emba_status emba_reconstruct_intensity(emba_ctx* c, const double* Gx_host, const double* Gy_host, double* M_host)
{
    if (!c) return EMBA_ERR_INVALID_ARG;
    if ((Gx_host == nullptr) != (Gy_host == nullptr)) return fail(c, EMBA_ERR_INVALID_ARG, "pass both Gx and Gy, or neither");
    if (!Gx_host && !c->have_map) return fail(c, EMBA_ERR_STATE, "no map resident");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int H = c->H, W = c->W;
    const size_t npix = c->npix;
    emba_status st;
    // option poisson = 1 (dense): both axes by sine-matrix products (four GEMMs: the round-1/2 form, kept for comparison); default: Fourier analysis
    // along H only + tridiagonal solves along W (poisson_kernels.h) — a third of the arithmetic, same solution to rounding.
    const bool dense = c->opt_poisson == 1;      // option poisson: 1 dense sine transforms, 2 no folding (comparison forms)
    if (!c->d_SH) {   // first use: S_H, its eigenvalues, two scratch planes
        if ((st = dev_alloc(c, &c->d_SH, (size_t)H * H)) || (st = dev_alloc(c, &c->d_lamH, (size_t)H)) || (st = dev_alloc(c, &c->d_pF, npix)) || (st = dev_alloc(c, &c->d_pT, npix))) {
            dev_free(c, c->d_SH); dev_free(c, c->d_lamH); dev_free(c, c->d_pF); dev_free(c, c->d_pT);
            return st;
        }
        hipLaunchKernelGGL(emba_sine_matrix_kernel, dim3((unsigned)(((size_t)H * H + 255) / 256)), dim3(256), 0, s, H, c->d_SH);
        hipLaunchKernelGGL(emba_dirichlet_eigen_kernel, dim3((H + 255) / 256), dim3(256), 0, s, H, c->d_lamH);
    }
    if (dense && !c->d_SW) {
        if ((st = dev_alloc(c, &c->d_SW, (size_t)W * W)) || (st = dev_alloc(c, &c->d_lamW, (size_t)W))) { dev_free(c, c->d_SW); dev_free(c, c->d_lamW); return st; }
        hipLaunchKernelGGL(emba_sine_matrix_kernel, dim3((unsigned)(((size_t)W * W + 255) / 256)), dim3(256), 0, s, W, c->d_SW);
        hipLaunchKernelGGL(emba_dirichlet_eigen_kernel, dim3((W + 255) / 256), dim3(256), 0, s, W, c->d_lamW);
    }
    if (!dense && !c->d_thomas) {   // Thomas factors of T_W + lambda1[i] I for every H-frequency i: W x H doubles, once
        if ((st = dev_alloc(c, &c->d_thomas, npix))) return st;
        hipLaunchKernelGGL(emba_thomas_coef_kernel, dim3((H + 63) / 64), dim3(64), 0, s, c->d_lamH, H, W, c->d_thomas);
    }
    const bool fold = !dense && (H % 2 == 0) && H >= 64 && c->opt_poisson != 2;
    if (fold && !c->d_Sfold) {
        if ((st = dev_alloc(c, &c->d_Sfold, (size_t)H * H / 2))) return st;
        hipLaunchKernelGGL(emba_sine_folded_kernel, dim3((unsigned)(((size_t)H * H / 2 + 255) / 256)), dim3(256), 0, s, H, c->d_Sfold);
    }
    const double *gx = c->d_Gx, *gy = c->d_Gy;
    if (Gx_host) {
        if (!c->d_pGx) { if ((st = dev_alloc(c, &c->d_pGx, npix)) || (st = dev_alloc(c, &c->d_pGy, npix))) { dev_free(c, c->d_pGx); return st; } }
        HIP_TRY(c, hipMemcpyAsync(c->d_pGx, Gx_host, npix * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->d_pGy, Gy_host, npix * sizeof(double), hipMemcpyHostToDevice, s));
        gx = c->d_pGx; gy = c->d_pGy;
    }
    if (dense) hipLaunchKernelGGL(emba_divergence_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, gx, gy, H, W, c->d_pF);
    auto gemm = [&](const double* A, const double* B, double* C, int M, int N, int K, int epilogue, double inv_norm, long lda = 0, int a_kstride = 1, int a_koff = 0) {
        GemmParams p{};
        p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda ? lda : K; p.ldb = N; p.ldc = N;
        p.a_kstride = a_kstride; p.a_koff = a_koff;
        p.epilogue = epilogue; p.inv_norm = inv_norm;
        p.lam1 = c->d_lamH; p.lam2 = c->d_lamW;
        p.vec = ((K & 1) == 0 && (N & 1) == 0) ? 1 : 0;
        const long tm = (M + kGemmBM - 1) / kGemmBM, wide = tm * ((N + 127) / 128), narrow = tm * ((N + 63) / 64);
        const long big = ((M + kGemmBM2 - 1) / kGemmBM2) * ((N + kGemmBN2 - 1) / kGemmBN2);
        const bool use_big = big >= (long)c->n_cu && !c->opt_gemm64;
        if (use_big) hipLaunchKernelGGL(emba_dgemm128_kernel, dim3((unsigned)grid8(big)), dim3(512), 0, s, p);
        else if (wide >= 2L * c->n_cu) hipLaunchKernelGGL(emba_dgemm_kernel<128>, dim3((unsigned)grid8(wide)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(emba_dgemm_kernel<64>, dim3((unsigned)grid8(narrow)), dim3(256), 0, s, p);
    };
    if (dense) {
        // rhs -> eigenvector space, solve, back (laplace.cpp:633-758):  M = S_H ((S_H F S_W) o C) S_W
        const double inv_norm = 1.0 / (4.0 * ((double)(H + 1) * (double)(W + 1)));   // fft_norm, laplace.cpp:648
        gemm(c->d_pF, c->d_SW, c->d_pT, H, W, W, 0, 1.0);            // T = F S_W
        gemm(c->d_SH, c->d_pT, c->d_pF, H, W, H, 1, inv_norm);       // U = (S_H T / fft_norm) / (lambda1 + lambda2)
        gemm(c->d_pF, c->d_SW, c->d_pT, H, W, W, 0, 1.0);            // T = U S_W
        gemm(c->d_SH, c->d_pT, c->d_pF, H, W, H, 0, 1.0);            // M = S_H T
    } else {
        const dim3 tg((W + 31) / 32, (H + 31) / 32), tgT((H + 31) / 32, (W + 31) / 32);
        // DST-I along H of every row of the transposed plane: src (W x H) -> dst (W x H), times scale.  Folded (even H): two half-size products on
        // the even- and odd-indexed inputs + a butterfly; tmp holds [E | O] (W x H/2 each).
        // transposed_out: dst receives the result TRANSPOSED (H x W) — the folded form's butterfly writes it that way, saving the last transpose pass
        auto dst_rows = [&](const double* src, double* dst, double* tmp, double scale, bool transposed_out) -> bool {
            if (!fold) { gemm(src, c->d_SH, dst, W, H, H, 2, scale); return false; }
            const int h = H / 2;
            double* E = tmp; double* O = tmp + (size_t)W * h;
            gemm(src, c->d_Sfold, E, W, h, h, 0, 1.0, H, 2, 0);
            gemm(src, c->d_Sfold + (size_t)h * h, O, W, h, h, 0, 1.0, H, 2, 1);
            if (transposed_out) hipLaunchKernelGGL(emba_dst_butterfly_T_kernel, dim3((h + 31) / 32, (W + 31) / 32), dim3(256), 0, s, E, O, W, H, scale, dst);
            else hipLaunchKernelGGL(emba_dst_butterfly_kernel, dim3((unsigned)(((size_t)W * h + 255) / 256)), dim3(256), 0, s, E, O, W, H, scale, dst);
            return transposed_out;
        };
        if (fold && !c->d_pE) { if ((st = dev_alloc(c, &c->d_pE, npix))) return st; }
        hipLaunchKernelGGL(emba_divergence_T_kernel, tg, dim3(256), 0, s, gx, gy, H, W, c->d_pT);                     // F^T  (W x H), straight from the gradient maps
        dst_rows(c->d_pT, c->d_pF, c->d_pE, 1.0, false);                                                                 // (S_H F)^T = F^T S_H
        const unsigned tb = (unsigned)((H + kTriSys - 1) / kTriSys);
        hipLaunchKernelGGL(emba_tridiag_sweep_kernel<false>, dim3(tb), dim3(kTriSys * kTriChunks), 0, s, c->d_thomas, c->d_lamH, c->d_pF, H, W);   // (T_W + lambda1[i] I) x = g, per i
        hipLaunchKernelGGL(emba_tridiag_sweep_kernel<true>, dim3(tb), dim3(kTriSys * kTriChunks), 0, s, c->d_thomas, c->d_lamH, c->d_pF, H, W);
        // M^T = X^T S_H / (2 (H+1)).  Folded: the butterfly writes M itself into d_pF (its input X^T is dead once E and O exist)
        if (fold) dst_rows(c->d_pF, c->d_pF, c->d_pE, 1.0 / (2.0 * (double)(H + 1)), true);
        else {
            dst_rows(c->d_pF, c->d_pT, c->d_pE, 1.0 / (2.0 * (double)(H + 1)), false);
            hipLaunchKernelGGL(emba_transpose_kernel, tgT, dim3(256), 0, s, c->d_pT, W, H, c->d_pF);                  // M
        }
    }
    HIP_TRY(c, hipGetLastError());
    if (M_host) HIP_TRY(c, hipMemcpyAsync(M_host, c->d_pF, npix * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->spun = false; c->knots_in_flight = false;
    return EMBA_OK;
}

// ---- single-process multi-GPU host (emba_group_*) ----------------------------------------------------------------------------
#include "group.h"
