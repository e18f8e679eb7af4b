/* include/emba_hip.h — C ABI of the MI355X (gfx950) implementation of EMBA's hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer; its seam is the
 * C++ object EMBA::LEGM (reference include/emba/model.h:72-133), owned by EMBA::EMBA
 * (include/emba/emba.h:109) and called only from EMBA::solveTimeWindow (src/emba/solver.cpp:63-353).
 * The functions below are what a thin `EMBA::LEGM` adapter binds (INTEGRATION.md shows it):
 *
 *   emba_create / emba_destroy      <- LEGM::LEGM / ~LEGM                 model.cpp:56-70, model.h:80
 *   emba_set_events                 <- the EventPacket argument of evaluateDataError + the
 *                                      pose-independent part of event_map_ (event_map.h:34-47)
 *   emba_eval_data_error            <- LEGM::evaluateDataError            model.cpp:72-258
 *   emba_form_normal_eq             <- LEGM::formNormalEq / formNormalEqIRLS + applyL2Reg
 *                                                                          model.cpp:316-491, 493-687, 689-719
 *   emba_data_cost / emba_reg_cost  <- 0.5*ep.ep, evaluateRobustDataCost, evaluateRegError
 *                                                                          solver.cpp:88-91, model.cpp:260-314
 *   emba_get_A12_sparse             <- the rank-1 factors of the dense A12 of model.cpp:358,483-487
 *
 * plus the phase-level entry points (emba_*_launch / *_finish) that keep every operand resident in
 * HBM, used by bench.py and by the multi-GPU host (one process per GPU; the two exchanges per
 * Gauss-Newton iteration — the int32 pixel-count map and the fp64 normal-equation pack — are
 * all-reduced by the caller between phases on buffers it binds with emba_bind_exchange_buffers).
 *
 * Conventions: plain C, no C++ types, no exceptions across the boundary.  Every function returns an
 * emba_status; the adapter turns non-zero into LOG(FATAL) to keep the reference's fail-fast
 * contract (glog CHECKs at model.cpp:247,350,378).  All floating point is IEEE double, indices are
 * int32/uint32, exactly as in the reference.  Single caller thread per context (non-reentrant), like
 * the reference.  "host" pointers are ordinary host memory; "dev" pointers are HIP device memory of
 * the context's device.  There is NO CPU fallback: without a GPU emba_create fails with
 * EMBA_ERR_NO_DEVICE.
 */
#ifndef EMBA_HIP_H
#define EMBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMBA_ABI_VERSION 2

typedef enum {
    EMBA_OK = 0,
    EMBA_ERR_INVALID_ARG = 1,   /* null pointer, bad size, unsorted timestamps, pixel out of sensor */
    EMBA_ERR_NO_DEVICE = 2,     /* no HIP device / requested ordinal absent */
    EMBA_ERR_HIP = 3,           /* a HIP runtime call failed; see emba_last_error */
    EMBA_ERR_TIME_RANGE = 4,    /* a batch midpoint lies outside the spline's knots
                                   (BASALT_ASSERT at so3_spline.h:221-229 in the reference) */
    EMBA_ERR_STATE = 5,         /* call order violated (e.g. form before eval) */
    EMBA_ERR_CAPACITY = 6,      /* a caller-provided buffer is too small */
    EMBA_ERR_NUMERIC = 7        /* a 2x2 block A22_i + lambda*diag(A22_i) is not positive definite: the reference's A22m_i.inverse()
                                   (model.cpp:750) yields inf/nan there and the LM step is lost.  (A vanishing pivot of the Schur
                                   complement is NOT an error: like Eigen's ldlt, model.cpp:789, it gives a zero update.) */
} emba_status;

typedef struct emba_ctx emba_ctx;

typedef struct {
    int32_t sensor_w, sensor_h;   /* camera_info.width/height        (model.cpp:67-68) */
    int32_t pano_w, pano_h;       /* panorama size, W = 2H           (emba.cpp:552-553) */
    const double* bearing_lut;    /* host, sensor_w*sensor_h*3 f64, row-major by y*w+x: the
                                     precomputed_bearing_vectors_ of event_pano_warper.cpp:27-41 */
    double C_th;                  /* contrast threshold               (model.cpp:60) */
    int32_t event_batch;          /* 0 or 100: the reference hard-codes 100 (model.cpp:78) */
    double outlier_px;            /* 0 -> 10.0                        (model.cpp:200) */
    int32_t device;               /* HIP device ordinal */
    void* stream;                 /* hipStream_t to enqueue on, or NULL for the context's own stream */
} emba_cfg;

/* Library/ABI identification (no GPU needed). */
int emba_abi_version(void);
const char* emba_build_info(void);

emba_status emba_create(const emba_cfg* cfg, emba_ctx** out);
void emba_destroy(emba_ctx* ctx);
/* Human-readable description of the last non-OK status on this context (or of a failed
 * emba_create when ctx == NULL). */
const char* emba_last_error(const emba_ctx* ctx);

/* Once per time window.  Events sorted by time (rosbag_loading.cpp:61-65 order is taken as
 * ground truth, quirk Q10); the last n % 100 events are ignored (quirk Q1, model.cpp:77-79).
 * Builds the pose-independent structure the reference rebuilds on every evaluateDataError call:
 * per-batch midpoint times (model.cpp:116-119), the per-sensor-pixel event lists
 * (event_map.h:34-47) laid out pixel-major in HBM, and uploads them.
 *
 * Sharding: a rank holds a contiguous range of whole global batches and, in
 * halo_*, for each sensor pixel the LAST event before its range (if any) together with the
 * midpoint time of the batch that event belongs to; halo events are warped but are not
 * measurements of this rank.  Pass n_halo = 0 on a single GPU. */
emba_status emba_set_events(emba_ctx* ctx, const uint16_t* x, const uint16_t* y, const uint8_t* pol,
                            const int64_t* t_ns, size_t n,
                            const uint16_t* halo_x, const uint16_t* halo_y,
                            const int64_t* halo_batch_t_ns, size_t n_halo);

/* The same with every array already in HBM (device pointers; t_ns is needed here: validation and the batch midpoints run on the
 * device too).  Nothing of the once-per-window structure is built on the host: the (sensor pixel, time) order is a stable LSD
 * radix sort, the record slots a second one by control-pose pair (emba_amd/csrc/order_kernels.h).  The arrays may be freed
 * when the call returns. */
emba_status emba_set_events_dev(emba_ctx* ctx, const uint16_t* x_dev, const uint16_t* y_dev, const uint8_t* pol_dev,
                                const int64_t* t_ns_dev, size_t n,
                                const uint16_t* halo_x_dev, const uint16_t* halo_y_dev,
                                const int64_t* halo_batch_t_ns_dev, size_t n_halo);

/* Diagnostics of the once-per-window work: wall time of the last emba_set_events[_dev] and of the order preparation done by
 * the first evaluation (control-pose pairs, record slots, optional tile order), whether the tile order is in use (events grouped
 * by the panorama tile the first trajectory sends their chain segment to: option "order"), entries of the device order (events +
 * halo / lead-in copies) and workgroup chunks of the tiled kernel.  Any pointer may be NULL. */
emba_status emba_last_setup_ms(const emba_ctx* ctx, double* set_events_ms, double* prepare_ms, int32_t* tile_order,
                               size_t* n_entries, size_t* n_chunks);

/* What the last pixel / tile order decision of a window saw (at its first evaluation): events per panorama pixel of the occupied cells of the tile-origin grid
 * under the first trajectory, and the fraction of lead-in copies the tile order would add.  Diagnostics; either pointer may be NULL. */
emba_status emba_last_order_stats(const emba_ctx* ctx, double* events_per_pano_px, double* lead_in_frac);
/* ... and the inlier fraction it estimated from the predicted pixels (round 6: the rule prices the pixel order per inlier and the tile order per entry). */
emba_status emba_last_order_inlier_estimate(const emba_ctx* ctx, double* inlier_frac);
/* The LDS tile that decision chose for the window (round 6: one of four shapes of 1152 panorama pixels), the pitch of its origin grid and the reserve kept free
 * on every side of a tile for the drift of trial poses (options tile_shape, tile_fine, tile_reserve).  Diagnostics; any pointer may be NULL. */
emba_status emba_last_tile_geometry(const emba_ctx* ctx, int32_t* tile_w, int32_t* tile_h, int32_t* pitch_x, int32_t* pitch_y, int32_t* reserve);

/* Tile order only: inliers of the last resolved evaluation that the tiled kernel found outside their LDS tile — the
 * trajectory has moved them since the order was built; they are handled correctly, one HBM atomic each — and how many times this
 * context has rebuilt the order of a window for that reason (done at the next evaluation once a fifth of the inliers are outside).
 * No reference counterpart (the reference has no device order).  Either pointer may be NULL. */
emba_status emba_last_tile_drift(const emba_ctx* ctx, size_t* n_outside, int32_t* n_rebin);

/* Number of events actually used (floor(n/100)*100) and of measurement candidates
 * (events that have a predecessor at their sensor pixel). */
emba_status emba_event_counts(const emba_ctx* ctx, size_t* n_used, size_t* n_candidates);

/* ---- one-shot, host-buffer entry points (what the LEGM adapter calls) ------------------------ */

/* LEGM::evaluateDataError(traj, Gx, Gy, events, eval_deriv=true, num_ev_map)   model.cpp:72-258.
 * knots_xyzw: K unit quaternions (x,y,z,w) = traj->getControlPose(i).unit_quaternion();
 * t0_ns/dt_ns: the spline's start_t_ns_/dt_ns_ (trajectory.cpp:59-64).
 * Gx, Gy: host, pano_h*pano_w f64 row-major (CV_64FC1 continuous); both NULL = use the map already resident on the
 * device (emba_upload_map / emba_bind_map_dev / emba_update_map).
 * ep_out: host, capacity >= n_used, receives the residuals in the reference's order (sensor pixel
 * major, then time; model.cpp:179-186,221); *n_inliers = ep.size().  num_ev_map_out: host int32
 * pano_h*pano_w (model.cpp:227) or NULL.  eval_deriv must be non-zero (the reference never passes
 * false, solver.cpp:75,251; quirk Q4). */
emba_status emba_eval_data_error(emba_ctx* ctx, const double* knots_xyzw, int32_t K, int64_t t0_ns,
                                 int64_t dt_ns, const double* Gx, const double* Gy, int32_t eval_deriv,
                                 double* ep_out, size_t* n_inliers, int32_t* num_ev_map_out);

/* LEGM::formNormalEq[IRLS] + LEGM::applyL2Reg   model.cpp:316-491, 493-687, 689-719.
 * Uses the device-resident state of the last emba_eval_data_error (the reference's hidden
 * event_map_ coupling, solver.cpp:99-102).  ep: host residuals to use (same layout as ep_out) or
 * NULL for the device-resident ones.  irls: 0 quadratic, 1 huber, 2 cauchy (model.cpp:599-618);
 * eta its parameter.  alpha: L2 weight; pass 0 to skip applyL2Reg.
 * Outputs (host; any may be NULL): A11 3K*3K col-major, b1 3K, *P active pixels,
 * active_idx (cap pano_h*pano_w, ascending pano index = std::set order, model.cpp:371-377),
 * A22 P*4 (each block [xx xy; xy yy]), b2 2P, A12_dense 3K x 2P col-major (model.cpp:358; only
 * sensible at small sizes — 3K*2P*8 bytes).  cap_P bounds A22/b2/A12 capacity in pixels. */
emba_status emba_form_normal_eq(emba_ctx* ctx, const double* ep, int32_t thres_valid_pixel,
                                int32_t irls, double eta, double alpha, double* A11, double* b1,
                                size_t* P, uint32_t* active_idx, size_t cap_P, double* A22, double* b2,
                                double* A12_dense);

/* Sparse form of A12: one rank-1 factor per measurement candidate, in the device's
 * (control-pose-pair)-sorted order.  A12[3*cp_c+i, 2*pix+d] += w*jc[i]*dp[d];
 * A12[3*cp_p+i, 2*pix+d] += w*jp[i]*dp[d].  pix = -1 marks outliers / inactive pixels.
 * All pointers host, capacity n_candidates (see emba_event_counts); any may be NULL. */
emba_status emba_get_A12_sparse(emba_ctx* ctx, int32_t* cp_c, int32_t* cp_p, int32_t* pix, double* w,
                                double* jc /*n*6*/, double* jp /*n*6*/, double* dp /*n*2*/);

/* Enqueue the compaction of the last evaluation's residuals into the reference-order vector `ep` (model.cpp:221,256) and the per-event inlier
 * numbers, on the device.  Nothing on the device reads `ep`.  The resident step (emba_step, option step_ep) produces it in the tail blocks of its
 * Gram launch (round 5) — this call is then a no-op; the phase-level calls of an LM loop leave it to whoever asks; the one-shot
 * emba_eval_data_error always pays for it. */
emba_status emba_compact_ep(emba_ctx* ctx);
/* The last evaluation's `ep` (what LEGM::evaluateDataError returns, model.cpp:256) to the host: as the resident step left it, or compacted now.
 * ep_host: capacity `cap` doubles (EMBA_ERR_CAPACITY below the inlier count); *n_inliers (may be NULL) its length.  Synchronizes. */
emba_status emba_get_ep(emba_ctx* ctx, double* ep_host, size_t cap, size_t* n_inliers);

/* Sensor pixel (y * sensor_w + x) of every inlier measurement of the last evaluation, in the order of ep (sensor pixel major, then time;
 * model.cpp:179-186): pix_host has capacity n_inliers.  What a multi-GPU host merges the ranks' residual vectors by (emba_group_eval). */
emba_status emba_get_inlier_pixels(emba_ctx* ctx, uint32_t* pix_host);
/* The same in S + 1 words (S = sensor_w * sensor_h): starts_host[p] = number of inlier measurements at sensor pixels < p, i.e. where pixel p's residuals begin in this
 * context's ep; starts_host[S] = the inlier count.  Kept by the context for emba_get_ep_by_pixel.  Synchronizes. */
emba_status emba_get_inlier_pixel_starts(emba_ctx* ctx, uint32_t* starts_host);
/* The last evaluation's ep placed pixel block by pixel block: the residuals of sensor pixel p go to ep_out[dst[p] ...] (dst: S element offsets).  What a multi-GPU host
 * merges time-sharded ranks' residual vectors with (model.cpp:179-186 orders them by sensor pixel, then time: inside a pixel, earlier ranks first): every rank writes
 * its own pieces of ONE output vector, streamed through pinned buffers, and ranks may do so concurrently from their own threads.  Synchronizes. */
emba_status emba_get_ep_by_pixel(emba_ctx* ctx, double* ep_out, const uint64_t* dst);

/* Cost terms of the caller (solver.cpp:88-91,257-268), reduced on the device from the resident
 * residuals / map: data = 0.5*ep.ep (irls 0) or evaluateRobustDataCost (model.cpp:279-314);
 * reg = 0.5*alpha*sum(Gx^2+Gy^2) over all pixels (model.cpp:260-277). */
emba_status emba_data_cost(emba_ctx* ctx, int32_t irls, double eta, double* cost);
emba_status emba_reg_cost(emba_ctx* ctx, double alpha, double* cost);
/* Both terms with one host synchronisation (the LM loop asks for them together at every trial point, solver.cpp:88-91, 265-268).
 * reg_cost may be NULL (a rank of a sharded host whose map is a replica).  emba_costs_launch / emba_costs_finish: the same in two halves, so
 * that a multi-GPU host can enqueue every rank's reductions before it waits for any. */
emba_status emba_costs(emba_ctx* ctx, int32_t irls, double eta, double alpha, double* data_cost, double* reg_cost);
emba_status emba_costs_launch(emba_ctx* ctx, int32_t irls, double eta, int32_t with_reg);
emba_status emba_costs_finish(emba_ctx* ctx, int32_t irls, double eta, double alpha, double* data_cost, double* reg_cost);

/* Per-event state dump in ORIGINAL (time) order for parity tests: what State_LEGM holds after
 * evaluateDataError (state.h:56-83).  All host, capacity n; any may be NULL.
 * pm n*2, D n*12 (dpm_ddrot_cp row-major 2x6), cp_idx n, inlier_idx n (-1 outlier, -2 not a
 * measurement), pm_int n*2 (-1 if not an inlier), dp/Gpm/temp n*2. */
emba_status emba_dump_state(emba_ctx* ctx, double* pm, double* D, int32_t* cp_idx, int32_t* inlier_idx,
                            int32_t* pm_int, double* dp, double* Gpm, double* temp);

/* ---- phase-level, HBM-resident entry points (bench.py, multi-GPU host) ----------------------- */

/* Copy the map to the device (host -> HBM), or adopt caller-owned device planes without a copy. */
emba_status emba_upload_map(emba_ctx* ctx, const double* Gx_host, const double* Gy_host);
emba_status emba_bind_map_dev(emba_ctx* ctx, const double* Gx_dev, const double* Gy_dev);

/* LM-loop residency: LEGM::updateMap(Gx_new, Gy_new, x2, damping, active, inactive), model.cpp:863-903, on
 * the device-resident map with the active set of the last emba_form_active: trial = current; trial[active_i] += damping*x2[2i],
 * x2[2i+1]; trial[every other pixel] = 0.  From then on evaluations use the TRIAL map (pass Gx = Gy = NULL to
 * emba_eval_data_error, or call emba_eval_launch) until the caller reports the LM decision: emba_map_accept makes the trial map
 * current (solver.cpp:299-339), emba_map_reject drops it (:340-352).  x2_host: 2P doubles, or NULL: the x2 the last
 * emba_solve_normal_eq[_cg] on this context left on the device (the reference hands x2 from the solver straight to updateMap,
 * solver.cpp:193-239 — 2P doubles that need not cross to the host and back).  emba_update_map_dev: x2 in device memory (a sharded
 * host's all-reduced x2), NULL as above.  emba_download_map copies the map the next evaluation would use. */
emba_status emba_update_map(emba_ctx* ctx, const double* x2_host, double damping);
emba_status emba_update_map_dev(emba_ctx* ctx, const double* x2_dev, double damping);
emba_status emba_map_accept(emba_ctx* ctx);
emba_status emba_map_reject(emba_ctx* ctx);
/* A rejected LM trial without a re-evaluation (solver.cpp:340-352 simply reuses A and b).  An evaluation that follows emba_form_* writes its
 * factor records into a SECOND record set, so the records, pack and active set the current normal equations consist of stay intact while
 * the trial point is evaluated.  emba_trial_reject (called by emba_map_reject; callable on its own when no map update was involved) makes
 * them current again: emba_solve_normal_eq[_cg], emba_update_map, emba_form_finish downloads work as if the trial evaluation had not
 * happened.  An ACCEPTED trial needs no call: the following emba_form_* forms new equations from the trial's records.  After a rejection
 * formNormalEq needs a new evaluation first (count map and per-pixel sums are the trial's). */
emba_status emba_trial_reject(emba_ctx* ctx);
emba_status emba_download_map(emba_ctx* ctx, double* Gx_host, double* Gy_host);
/* The map the next evaluation would use (the trial map after emba_update_map), at the P pixels of the current active set only:
 * gxy_host[2i] = Gx[active_i], gxy_host[2i+1] = Gy[active_i].  By model.cpp:892-901 the trial map is ZERO at every other pixel, so this is the
 * whole result of LEGM::updateMap in 16 B x P instead of 16 B x H x W (the drop-in adapter zero-fills the caller's Mats around it). */
emba_status emba_get_map_active(emba_ctx* ctx, double* gxy_host, size_t cap_P);

/* Intensity panorama from the gradient map: poisson_reconstruction::reconstructFromGradient,
 * src/image_rec/poisson_reconstruction.cpp:9-50 + pde::poisolve (Dirichlet, zero boundary), src/image_rec/laplace.cpp:587-797,
 * as the reference calls it for its map images (solver.cpp:417,471).  Gx_host/Gy_host: pano_h x pano_w row-major doubles, or
 * both NULL to use the map resident on the device (the one the next evaluation would use).  M_host: pano_h x pano_w doubles,
 * or NULL to leave the result in the context's device buffer (timing).  The 2-D DST-I is applied as products with the sine
 * matrix on the fp64 matrix cores (the transform lengths 2(n+1) have large prime factors); the matrices are built on first
 * use and kept (8 (H^2 + W^2) bytes). */
emba_status emba_reconstruct_intensity(emba_ctx* ctx, const double* Gx_host, const double* Gy_host, double* M_host);

/* Schur-complement solve: LEGM::solveNormalEq(A11, A12, A22_blocks, b1, b2, lambda, x1, x2), model.cpp:721-792,
 * on the device-resident normal equations of the last emba_form_finish (so after applyL2Reg, as in solver.cpp:130,190-202),
 * consuming the SPARSE A12 factors: S = A11m - A12 A22m^-1 A12^T is formed chunk-wise from per-pixel column pairs built from the
 * records, x1 = S \ (b1 - A12 A22m^-1 b2) by a blocked Cholesky that treats a vanishing pivot the way Eigen's ldlt does (model.cpp:789:
 * the pseudo-inverse of D — a control pose no event constrains gets a ZERO update; emba_last_solve_info bit 1 reports it),
 * x2 = A22m^-1 (b2 - A12^T x1).  LM damping as the reference:
 * A11m = A11 + lambda*diag(A11), A22m = A22 + lambda*diag(A22).  fix_first_pose != 0 reproduces the first-window trim of
 * solver.cpp:156-165 (rows/cols 0..2 dropped; x1[0..2] = 0 on return).  x1_host: 3K doubles, x2_host: 2P doubles (either may
 * be NULL).  Works on this context's own records: with a sharded window use emba_solve_shard_* (a pixel's A12 columns are sums over all
 * ranks' records, so the records are first re-distributed by pixel owner). */
emba_status emba_solve_normal_eq(emba_ctx* ctx, double lambda, int32_t fix_first_pose, double* x1_host, double* x2_host);
/* Diagnostics of the last Schur solve of this context: bit 0 = a 2x2 block was not positive definite (EMBA_ERR_NUMERIC was returned),
 * bit 1 = a pivot of S vanished and its component got a zero update (what ldlt.info() == NumericalIssue is in the reference: never read). */
emba_status emba_last_solve_info(const emba_ctx* ctx, int32_t* info);

/* The same Schur solve for a window sharded over n_ranks GPUs (events sharded by time, A12 factors stay sharded).
 * A pixel's A12 columns are sums over ALL ranks' records of that pixel, and S needs the outer products of the summed columns, so the
 * records are first re-distributed by pixel owner (rank r owns the active pixels [P r / n, P (r+1) / n) in ascending panorama order):
 *   emba_solve_shard_count   counts_host[n_ranks] = this rank's valid records on active pixels, by owner
 *   emba_solve_shard_pack    packs them owner-major into send_dev (16 doubles each; tail word = {compact pixel index, pose-pair key})
 *   -- caller: all-to-all of the records (RCCL send/recv, torch.distributed.all_to_all_single) into recv_dev / n_recv --
 *   emba_solve_shard_partial S_part_dev (emba_solve_shard_size doubles) = - sum over the OWNED pixels of U_aug U_aug^T
 *   -- caller: all-reduce(SUM) of S_part_dev --
 *   emba_solve_shard_finish  adds the replicated [A11m | b1], factors, x1 -> x1_host (identical on every rank); x2 of the owned pixels
 *                            into x2_full_dev (2P doubles, zero elsewhere: the caller all-reduces it to get every rank's full x2)
 * Call after emba_form_finish (A22/b2 all-reduced, L2 applied), all on the context's stream.  Tested as rank threads on one GPU
 * against the single-process oracle (tests/test_gpu_sharded.py).
 * Re-solves (round 6): solver.cpp:340-352 solves the SAME equations again with a larger lambda after every rejected trial.  A rank keeps the records it received
 * for its pixels (in pixel order) until new equations are formed; emba_solve_shard_cached says whether it still holds them for the current equations — when every
 * rank does, skip count / pack / all-to-all and pass recv_dev = NULL, n_recv = the cached count to _partial and _finish. */
emba_status emba_solve_shard_size(emba_ctx* ctx, size_t* s_doubles);
emba_status emba_solve_shard_count(emba_ctx* ctx, int32_t n_ranks, size_t* counts_host);
emba_status emba_solve_shard_pack(emba_ctx* ctx, int32_t n_ranks, double* send_dev);
emba_status emba_solve_shard_cached(emba_ctx* ctx, int32_t rank, int32_t n_ranks, int32_t* cached, size_t* n_recv);
emba_status emba_solve_shard_partial(emba_ctx* ctx, int32_t rank, int32_t n_ranks, const double* recv_dev, size_t n_recv,
                                     double lambda, double* S_part_dev);
emba_status emba_solve_shard_finish(emba_ctx* ctx, int32_t rank, int32_t n_ranks, const double* recv_dev, size_t n_recv,
                                    double lambda, int32_t fix_first_pose, double* S_dev, double* x1_host, double* x2_full_dev);

/* LEGM::solveNormalEqCG for a sharded window (round 6; model.cpp:794-840 with the pixels sharded by owner as in the Schur solve): the per-rank steps around the
 * caller's collectives.  A CG vector is [3K pose entries, replicated | this rank's pixels' map entries]; `red_dev` holds emba_cg_shard_size doubles = 3K + 2:
 *   emba_cg_shard_begin      (after the record exchange, or with recv_dev = NULL where emba_solve_shard_cached allows) x = 0, r = b, p = invd r;
 *                            red[3K] = r.r, red[3K+1] = r.p, partial                           -- caller: all-reduce(SUM) red_dev (3K + 2) --
 *   emba_cg_shard_apply      t = M p on the rank's pixels; red[0..3K) = its share of the pose part of t, red[3K] = its p.t    -- all-reduce red_dev (3K + 2) --
 *   emba_cg_shard_pt         takes the reduced pose part; *pt = p.t (host, identical on every rank)        [alpha = r.z / p.t]
 *   emba_cg_shard_update     x += alpha p, r -= alpha t, z = invd r; red[3K] = r.r, red[3K+1] = r.z, partial         -- all-reduce red_dev + 3K (2) --
 *   emba_cg_shard_direction  p = z + beta p                                                                [beta = r.z / previous r.z]
 *   emba_cg_shard_end        x1 -> x1_host; the rank's x2 into x2_full_dev (2P doubles, zero elsewhere)               -- all-reduce x2_full_dev (2P) --
 * The loop and its stopping rule are Eigen's (ConjugateGradient.h:28-88): emba_group_solve_cg, emba_amd/sharded.py: ShardedLEGM.solveNormalEqCG. */
emba_status emba_cg_shard_size(emba_ctx* ctx, size_t* red_doubles);
emba_status emba_cg_shard_begin(emba_ctx* ctx, int32_t rank, int32_t n_ranks, const double* recv_dev, size_t n_recv, double lambda, int32_t fix_first_pose, double* red_dev);
emba_status emba_cg_shard_apply(emba_ctx* ctx, double* red_dev);
emba_status emba_cg_shard_pt(emba_ctx* ctx, const double* red_dev, double* pt);
emba_status emba_cg_shard_update(emba_ctx* ctx, double alpha, double* red_dev);
emba_status emba_cg_shard_direction(emba_ctx* ctx, double beta);
emba_status emba_cg_shard_end(emba_ctx* ctx, double* x1_host, double* x2_full_dev);

/* LEGM::solveNormalEqCG (model.cpp:794-840; selected by BA_config.use_CG at solver.cpp:190-202): Eigen's ConjugateGradient (default
 * diagonal preconditioner, zero initial guess; max_iter <= 0 -> 100, tol <= 0 -> 1e-6 as in the reference) on the full system
 * [A11m A12; A12^T A22m], applied matrix-free through the sparse A12 factors.  iterations / error = cg.iterations() / cg.error(). */
emba_status emba_solve_normal_eq_cg(emba_ctx* ctx, double lambda, int32_t fix_first_pose, int32_t max_iter, double tol,
                                    double* x1_host, double* x2_host, int32_t* iterations, double* error);

/* Bind caller-owned device buffers that the caller all-reduces between phases:
 *   count_map_dev : int32 pano_h*pano_w                       (exchange 1)
 *   pack_dev      : f64, capacity pack_cap doubles, laid out [A11 9K^2 | b1 3K | A22b2 5P]
 *                   with A22b2 = per active pixel {xx, xy, yy, bx, by}   (exchange 2)
 * Pass NULLs to return to context-owned buffers. */
emba_status emba_bind_exchange_buffers(emba_ctx* ctx, int32_t* count_map_dev, double* pack_dev,
                                       size_t pack_cap);

/* Enqueue whatever is still needed for the device count map (own or bound with emba_bind_exchange_buffers) to hold num_ev_map of the
 * last evaluation (model.cpp:227).  The evaluation itself only marks touched pixels there — the count of a pixel is accumulated
 * next to its A22/b2 sums, one atomic request per measurement — and the markers become counts in the first dense pass that follows
 * (emba_form_active, emba_count_compress, a map download).  A host that reads or all-reduces a BOUND count map directly, before
 * any of those, calls this first. */
emba_status emba_count_map_ready(emba_ctx* ctx);

/* Exchange-1 compression for multi-GPU hosts: the merged count map only decides `count >= thres` (model.cpp:333,409), so each rank
 * may send min(count, cap) as ONE BYTE per pixel (cap * world_size <= 255, thres <= cap) — a quarter of the int32 volume on the
 * xGMI links.  emba_count_compress writes the saturated bytes of the context's count map to u8_dev (pano_h*pano_w bytes);
 * emba_count_expand stores the (all-reduced) bytes back as int32.  After that the map holds saturated, not exact, totals. */
emba_status emba_count_compress(emba_ctx* ctx, uint8_t* u8_dev, int32_t cap);
emba_status emba_count_expand(emba_ctx* ctx, const uint8_t* u8_dev);

/* Phase F1 of a RESIDENT step on a rank of a sharded window (round 5: what emba_step does on one GPU, between the two exchanges): emba_eval_finish +
 * emba_form_active in the step's form — launch A leaves per-unit active lists and zeroes the touched-but-inactive accumulator lines, the active-set write +
 * A22 | b2 gather ride inside the Gram launch of the emba_form_accumulate that follows and zero the lines they read, so the next evaluation needs no clearing
 * pass.  global_counts_u8_dev: the all-reduced saturated byte counts of exchange 1 (emba_count_compress on every rank, then the caller's all-reduce) — activity is
 * decided from THEM (model.cpp:333,409 on the global counts) while the context's own count map keeps this rank's counts; NULL: from the count map.  Asynchronous;
 * P comes from emba_last_counts once the Gram launch is enqueued.  No applyL2Reg here: emba_form_finish(alpha) after exchange 2.  This evaluation's per-pixel
 * sums are consumed: a second formNormalEq on it rebuilds A22 | b2 from the records. */
emba_status emba_step_form_active(emba_ctx* ctx, int32_t thres_valid_pixel, const uint8_t* global_counts_u8_dev);

/* Phase E1: pose table, Hessian/texel pack, warp + residual + count + factor records.  Asynchronous
 * on the context's stream.  After it the count map holds THIS rank's counts. */
emba_status emba_eval_launch(emba_ctx* ctx, const double* knots_xyzw_host, int32_t K, int64_t t0_ns,
                             int64_t dt_ns);
/* Phase E2: residual compaction into reference order.  With any non-NULL output it synchronizes, returns the
 * inlier count and copies ep / the (possibly all-reduced) count map to the host; with all three NULL it only
 * enqueues work (the count is available from emba_last_counts after the next synchronizing call). */
emba_status emba_eval_finish(emba_ctx* ctx, double* ep_out_host, size_t* n_inliers,
                             int32_t* num_ev_map_out_host);
/* Phase F1: active set from the (all-reduced) count map.  With P or pack_len non-NULL it synchronizes to return P
 * and the number of doubles of the pack [A11 | b1 | A22b2] that exchange 2 must all-reduce; with both NULL it only
 * enqueues work (single-GPU steps then need ONE host synchronization, in emba_form_finish). */
emba_status emba_form_active(emba_ctx* ctx, int32_t thres_valid_pixel, size_t* P, size_t* pack_len);
/* Phase F2: zero the pack and accumulate this rank's measurements into it.  Asynchronous.
 * ep_host as in emba_form_normal_eq. */
emba_status emba_form_accumulate(emba_ctx* ctx, const double* ep_host, int32_t irls, double eta);
/* Phase F3: applyL2Reg on the (all-reduced) pack — at most once per emba_form_accumulate, later calls with alpha != 0 only
 * download — then optional download (any pointer may be NULL); synchronizes. */
emba_status emba_form_finish(emba_ctx* ctx, double alpha, double* A11, double* b1, uint32_t* active_idx,
                             size_t cap_P, double* A22, double* b2, double* A12_dense);

/* One whole resident step in one call (what bench.py times on a single GPU): emba_eval_launch + emba_eval_finish +
 * emba_form_active + emba_form_accumulate(device-resident ep) + emba_form_finish(alpha, no downloads), one host
 * synchronisation at the end.  n_inliers / P may be NULL. */
emba_status emba_step(emba_ctx* ctx, const double* knots_xyzw_host, int32_t K, int64_t t0_ns, int64_t dt_ns,
                      int32_t thres_valid_pixel, int32_t irls, double eta, double alpha, size_t* n_inliers, size_t* P);

/* Declare the robust cost of the formNormalEq[IRLS] calls that will follow the next evaluations (irls: 0 quadratic, 1 huber,
 * 2 cauchy; eta as in formNormalEqIRLS, model.cpp:493-687).  The reference applies the IRLS weight w(e) to A22/b2 in
 * formNormalEqIRLS (model.cpp:599-636); the weight only depends on the residual, which the evaluation has in hand, so an
 * evaluation that knows the cost accumulates the WEIGHTED per-pixel sums directly and the following emba_form_accumulate with the
 * same (irls, eta) takes A22/b2 from them instead of a second pass over the records.  Purely a speed hint: results are the same
 * with any declaration (a mismatch falls back to the records).  emba_step declares its own cost for its own evaluation. */
emba_status emba_set_cost(emba_ctx* ctx, int32_t irls, double eta);

/* Tuning and A/B switches: options by name, integer values, part of the ABI (EMBA_ABI_VERSION 2); an unknown name or a value out
 * of range is EMBA_ERR_INVALID_ARG.  Every option changes SPEED or the internal form only — results are the same under all of them (tests/test_gpu_parity.py
 * runs the parity cases under the non-default values).  Defaults are what the measurements under profiles/ chose.
 *   order          0 auto | 1 pixel order | 2 tile order of the device's event stream (takes effect at the next evaluation: the order is rebuilt)
 *   texel          0 auto | 1 pack every texel | 2 3x3 stencil on the fly | 3 pack inside the previous footprint's rectangle
 *   segpose        0 auto (= 2) | 1 per-batch pose table in pixel order | 2 per-event pose from the K-1 segment records
 *   gram_tags      1 the pixel order's 8-B tag stream lets the Gram kernel skip dead slots | 0 decide from the records
 *   step_ep        1 emba_step compacts the residuals into ep (what evaluateDataError returns): in the tail of its Gram launch, or for windows of more than
 *                  8.4 M entries by a scan + a compaction launch behind it (2: that form always) | 0 on demand only.  (emba_get_option "ep_valid", read-only: the device holds ep)
 *   step_fast      1 emba_step zeroes the per-pixel sums behind their reader (no clearing pass) | 0 keeps the clearing pass
 *   step_gather    0 sweeping active-set write | 1 list-driven gather as a launch | 2 (default) inside the Gram kernel | 3 inside it at every size
 *   step_one_set   1 emba_step keeps one record set | 0 alternates between two like an LM loop
 *   gram_sparse    -1 auto (by the active-pixel count of the window's last equations) | 0 | 1 the Gram kernel's form for slot streams with few live records (pixel
 *                  order on a large panorama): stages of 128 tags, the live slots compacted, only their records fetched;  gram_sparse_chunk 1 ... 8 (4): its slots per wave, x 1024
 *   gather_waves   0 auto (4) | 1 | 2 | 4 waves of a Gram workgroup do its slice of the gather
 *   chunk_order_bin 0 tile-order chunks longest first | 1 in bin order
 *   tile_shape     -1 auto (fewest entries + chunks) | 0 48x24 | 1 72x16 | 2 96x12 | 3 36x32: the tiled kernel's LDS tile;  tile_fine -1 auto | 0 | 1 the finer grid of tile origins;
 *                  tile_reserve 0 ... 5 (2): pixels kept free on every side of a tile when chains are cut into tile-sized segments;  tile_min_events (1650000): fewest
 *                  events for which order = 0 considers the tile order;  tile_chunk 0 auto | entries per workgroup of the tiled kernel
 *   solve_perm     -1 auto | 0 | 1 U's columns ordered by panorama column in the Schur solve
 *   solve_counts   -1 auto | 0 list lengths counted from the records | 2 both, compared
 *   syrk_dense     1 dense instead of block-sparse SYRK;  syrk_lists 0 auto | 1 the block-sparse SYRK's per-pair slice lists always | 2 its (pair, chunk) items always;
 *                  syrk_min_cols 64 ... 4096 (512): fewest columns of U a workgroup of the dense split-K SYRK takes;  syrk_item_cap 1 ... 65536 (4096): slabs of the item
 *                  form (an item beyond them adds its tile to S by global atomics)
 *   poison         1 (tests) fills every NEW device allocation of the context with 0xFF bytes: a read of never-written workspace then shows as NaN / 0xFFFFFFFF
 *   solve_debug    1 prints the band statistics of a solve
 *   poisson        0 folded Fourier form | 1 dense sine transforms | 2 no folding;  gemm64 1 forces the 64-wide GEMM tiles */
emba_status emba_set_option(emba_ctx* ctx, const char* name, int32_t value);
emba_status emba_get_option(emba_ctx* ctx, const char* name, int32_t* value);

/* Inlier count of the last evaluation and active-pixel count of the last emba_form_active, once resolved
 * (after any synchronizing call, e.g. emba_form_finish or emba_sync). */
emba_status emba_last_counts(emba_ctx* ctx, size_t* n_inliers, size_t* P);

/* Stream / event helpers so a host without HIP bindings can time the phases on the stream the
 * kernels run on (bench.py's roofline leg). */
emba_status emba_sync(emba_ctx* ctx);
emba_status emba_timer_start(emba_ctx* ctx, int32_t slot);   /* records a hipEvent on the stream */
emba_status emba_timer_stop(emba_ctx* ctx, int32_t slot);
emba_status emba_timer_elapsed_ms(emba_ctx* ctx, int32_t slot, float* ms); /* synchronizes on stop */
/* Device-measured duration (ms, HIP events around the launch) of the dominant kernel
 * (warp+residual+record kernel, "emba_warp_residual_kernel") in the most recent emba_eval_launch,
 * valid after a sync when kernel timing was enabled with emba_enable_kernel_timing(ctx, 1). */
emba_status emba_enable_kernel_timing(emba_ctx* ctx, int32_t on);      /* on = 0: off; 1 + slot (slot < 16): the events of the next step go to that slot */
emba_status emba_last_kernel_ms(emba_ctx* ctx, float* warp_ms, float* accum_ms);
/* ... read LATER, after the timed loop: sampled steps then cost their event records only, not a host wait each (bench.py) */
emba_status emba_kernel_ms_slot(emba_ctx* ctx, int32_t slot, float* warp_ms, float* accum_ms);
/* All launches of a sampled step (VERDICT r4 #1a).  emba_kernel_timing_all(ctx, 1): sampled steps also record an event in front of their first
 * launch, so that four consecutive intervals tile the step's device time; emba_kernel_ms_all returns them for a slot:
 * ms4[0] prep || pose || texel, [1] the warp kernel, [2] post-warp launch A (+ the sweeping active write where it is a launch of its own),
 * [3] the Gram kernel (with the active-set gather where it rides inside); -1 where a sampled step did not record them. */
emba_status emba_kernel_timing_all(emba_ctx* ctx, int32_t on);
emba_status emba_kernel_ms_all(emba_ctx* ctx, int32_t slot, float* ms4);
/* What an event bracket costs by itself: the mean HIP-event interval around an EMPTY kernel queued behind running work (reps brackets). */
emba_status emba_bracket_overhead_us(emba_ctx* ctx, int32_t reps, float* us);
/* The clocks this run had: the device attributes (kHz), and a probe kernel — one wave per SIMD running a dependent chain of fp32 adds
 * between two reads each of the constant-rate clock (s_memrealtime) and the shader-cycle counter (s_memtime): *probe_us its duration,
 * *sclk_mhz = shader cycles / duration, *cycles_per_add the chain's cadence (8 on gfx950: a sanity check of the counter).  Any pointer may be NULL. */
emba_status emba_clock_probe(emba_ctx* ctx, double* sclk_mhz, double* probe_us, double* cycles_per_add, int32_t* clock_rate_khz,
                             int32_t* mem_clock_rate_khz, int32_t* wall_clock_rate_khz);
/* "0000:xx:00.0" of the context's device (hipDeviceGetPCIBusId): /sys/bus/pci/devices/<id>/ holds the card's own clock and power report. */
emba_status emba_device_pci_bus_id(emba_ctx* ctx, char* buf, size_t len);

/* ---- Several GPUs behind ONE host thread ------------------------------------------------------------------------
 * The reference front-end is one process that owns one LEGM (src/emba/emba.cpp:378; solver.cpp:63-353 calls it).  An emba_group is
 * that object for a multi-GPU node: n_ranks contexts on devices[0..n_ranks), events sharded by time on the global batch grid with
 * a per-pixel halo, the two exchanges of an iteration as grouped RCCL all-reduces on the contexts' own streams (ncclCommInitAll,
 * librccl bound at run time).  All ranks on ONE device (devices = {0, 0}: what a single-GPU box can test) or n_ranks = 1 exchange
 * through in-library copies and add kernels instead — same protocol, no RCCL.  Same call order as the single-GPU entry points:
 *   emba_group_set_events -> emba_group_upload_map -> { emba_group_step -> [emba_group_download] -> emba_group_solve ->
 *   emba_group_update_map -> emba_group_step ... -> emba_group_map_accept / _reject }.  emba_group_ctx gives a rank's context for
 * anything else (diagnostics, dumps).  emba_group_eval + emba_group_form are the two halves of emba_group_step with the reference's
 * call shape (solver.cpp:75,251 evaluateDataError; :114-130 formNormalEq[IRLS] + applyL2Reg), so the EMBA::LEGM adapter
 * (emba_amd/host/legm_adapter.hpp) sits on a group of any size — one rank included — without a change to solver.cpp.
 * With more than one rank every rank's launches are issued from a host thread of its own (EMBA_GROUP_NO_THREADS: from the caller's). */
typedef struct emba_group emba_group;
emba_status emba_group_create(const emba_cfg* cfg, const int32_t* devices, int32_t n_ranks, emba_group** out);   /* cfg->device, cfg->stream are ignored */
/* ... with flags: EMBA_GROUP_FORCE_RCCL — a ONE-rank group goes through RCCL too (what a one-GPU box can rehearse of that path: run-time binding of
 * librccl, communicator set-up, every collective call, world size 1); EMBA_GROUP_NO_THREADS — the caller's thread drives every rank itself. */
#define EMBA_GROUP_FORCE_RCCL 1u
#define EMBA_GROUP_NO_THREADS 2u
emba_status emba_group_create_flags(const emba_cfg* cfg, const int32_t* devices, int32_t n_ranks, uint32_t flags, emba_group** out);
/* emba_set_option on every rank's context, plus the group's own: x2_split — -1 auto (from 3 M events per rank) | 0 exchange 2 in one piece | 1 its
 * A22 | b2 rows on the side streams under the Gram kernel; group_step_fast — 1 the ranks form as resident steps where they can (emba_step_form_active) |
 * 0 the sweeping forms. */
emba_status emba_group_set_option(emba_group* g, const char* name, int32_t value);
void        emba_group_destroy(emba_group* g);
const char* emba_group_last_error(const emba_group* g);
int32_t     emba_group_size(const emba_group* g);
int32_t     emba_group_uses_rccl(const emba_group* g);
emba_ctx*   emba_group_ctx(emba_group* g, int32_t rank);
emba_status emba_group_set_events(emba_group* g, const uint16_t* x, const uint16_t* y, const uint8_t* pol, const int64_t* t_ns, size_t n);
emba_status emba_group_upload_map(emba_group* g, const double* Gx, const double* Gy);
/* evaluateDataError + formNormalEq[IRLS] + applyL2Reg over all ranks; n_inliers = total over the ranks, P = active pixels */
emba_status emba_group_step(emba_group* g, const double* knots_xyzw, int32_t K, int64_t t0_ns, int64_t dt_ns, int32_t thres_valid_pixel,
                            int32_t irls, double eta, double alpha, size_t* n_inliers, size_t* P);
/* LEGM::evaluateDataError over all ranks.  Gx / Gy: host planes to upload first, or both NULL for the resident (current or trial) map.
 * Outputs (any may be NULL): ep_out (capacity >= events used) = the residuals of all ranks merged into the reference's order, *n_inliers
 * their number, num_ev_map_out the GLOBAL count map (exact int32 exchange; without it exchange 1 is left to emba_group_form, which may
 * send saturated bytes). */
emba_status emba_group_eval(emba_group* g, const double* knots_xyzw, int32_t K, int64_t t0_ns, int64_t dt_ns, const double* Gx, const double* Gy,
                            double* ep_out, size_t* n_inliers, int32_t* num_ev_map_out);
/* The residual vector of the last emba_group_eval (ep_out NULL there, n_inliers asked for) in the reference's order, into memory the caller owns: lets a host
 * that returns it by value (VecXd LEGM::evaluateDataError) size its vector from the count and receive the residuals in it directly.  Host copies of this size go
 * through two pinned 4-MB buffers in pipelined chunks (also for emba_eval_finish / emba_get_ep). */
emba_status emba_group_get_ep(emba_group* g, double* ep_out, size_t cap, size_t* n_inliers);
/* LEGM::formNormalEq[IRLS] + applyL2Reg over all ranks on the state of the last emba_group_eval (device-resident residuals). */
emba_status emba_group_form(emba_group* g, int32_t thres_valid_pixel, int32_t irls, double eta, double alpha, size_t* n_inliers, size_t* P);
/* LEGM::applyL2Reg as a call of its own (after emba_group_form with alpha = 0): once per set of equations, on every rank's replica. */
emba_status emba_group_apply_l2(emba_group* g, double alpha);
/* emba_set_cost on every rank: the robust cost of the emba_group_form calls to come (speed only; emba_group_step declares its own). */
emba_status emba_group_set_cost(emba_group* g, int32_t irls, double eta);
emba_status emba_group_download(emba_group* g, double* A11, double* b1, uint32_t* active_idx, size_t cap_P, double* A22, double* b2);
emba_status emba_group_costs(emba_group* g, int32_t irls, double eta, double alpha, double* data_cost, double* reg_cost);
emba_status emba_group_solve(emba_group* g, double lambda, int32_t fix_first_pose, double* x1_host, double* x2_host);
/* LEGM::solveNormalEqCG over the group (round 6: any number of ranks — the pixels are sharded as in the Schur solve, one all-reduce of 3K + 2 doubles per
 * application of the matrix and one of 2 doubles per iteration; emba_cg_shard_* below are the per-rank steps). */
emba_status emba_group_solve_cg(emba_group* g, double lambda, int32_t fix_first_pose, int32_t max_iter, double tol, double* x1_host, double* x2_host,
                                int32_t* iterations, double* error);
/* Did the last emba_group_solve / _solve_cg run the record exchange (1), or did every rank still hold the records it had received for these equations (0)? */
emba_status emba_group_last_solve_exchanged(const emba_group* g, int32_t* exchanged);
/* x2_host == NULL: every rank applies the (all-reduced) x2 the last emba_group_solve / _solve_cg left in its own device memory. */
emba_status emba_group_update_map(emba_group* g, const double* x2_host, double damping);
emba_status emba_group_map_accept(emba_group* g);
emba_status emba_group_map_reject(emba_group* g);
emba_status emba_group_download_map(emba_group* g, double* Gx_host, double* Gy_host);
emba_status emba_group_get_map_active(emba_group* g, double* gxy_host, size_t cap_P);   /* emba_get_map_active on rank 0's replica */
emba_status emba_group_trial_reject(emba_group* g);   /* emba_trial_reject on every rank (emba_group_map_reject includes it) */

#ifdef __cplusplus
}
#endif
#endif /* EMBA_HIP_H */
