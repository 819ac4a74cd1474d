/*
 * mppi_hip.h -- C ABI of the MI355X-native MPPI rollout engine (libmppi_hip.so).
 *
 * Drop-in seam: the three hot methods of the reference controller's `MPPI` class
 * (moribots/motion_planning, control/src/mppi) plus its tick driver:
 *
 *     MPPI.__init__/initialize   control/src/mppi:62-83     -> mppi_create / mppi_reset
 *     MPPI.get_cost2go           control/src/mppi:127-178   -> mppi_rollout (+ mppi_download_value/_noise)
 *     MPPI.get_cost              control/src/mppi:180-184   -> inside the rollout kernel
 *     MPPI.update_action         control/src/mppi:186-208   -> mppi_update
 *     MPPI.perform_action        control/src/mppi:210-213   -> mppi_plant_step
 *     MPPI.get_path              control/src/mppi:85-102    -> mppi_tick (= tick_begin + tick_finish)
 *
 * The reference has no FFI of its own (the node is one Python script); the binding a
 * maintainer adds is the ctypes shim shown in INTEGRATION.md (shipped as
 * motion_planning_amd/mppi.py).
 *
 * Conventions
 *   - plain C, no C++/torch types; opaque handle owns every device buffer;
 *   - host arrays are caller-owned, row-major float64, laid out exactly like the
 *     reference's numpy arrays, with a leading agent axis A (A = 1 for the stock node):
 *         state, goal [A][3]   uvec [A][2][T]   eps [A][T][2][K]   V [A][T][K];
 *   - every function returns 0 on success or a negative MPPI_E_* code, never throws or
 *     aborts across the ABI; mppi_last_error() gives the message;
 *   - one handle is not re-entrant; it may be driven from any single thread;
 *   - all work is enqueued on the handle's HIP stream.  DEFAULT: a stream the engine creates for
 *     itself with hipStreamNonBlocking -- it does NOT synchronise with the legacy null stream, so a
 *     caller that touches engine-owned device memory (mppi_partials_ptr, the `gathered_dev` buffer
 *     it passes to mppi_tick_finish) from another stream must order the two: either run the engine
 *     on that stream (mppi_set_stream), or use mppi_stream_wait_partials / mppi_wait_for_stream,
 *     or call mppi_synchronize.  Calls that return host data wait for the engine's stream, the rest
 *     are asynchronous.
 *   - every blocking wait is bounded: if the device does not finish within the sync timeout
 *     (mppi_set_sync_timeout, default 10 s; env MPPI_SYNC_TIMEOUT_MS) the call returns
 *     MPPI_E_TIMEOUT instead of blocking the control thread forever; the engine must then be
 *     destroyed.
 *   - a call makes the engine's device current only for its own duration and restores the caller's
 *     current device before returning.
 *   - K is the number of samples owned by THIS handle (one GPU's shard); sample_offset is
 *     the global index of its first sample (device-RNG streams are keyed by global index,
 *     so the NOISE does not depend on the shard count; with mppi_config.samples_total the
 *     arithmetic does not either).
 *
 * Environment: MPPI_SYNC_TIMEOUT_MS (default deadline of the blocking waits) is the only variable the library reads.
 * Measurement and test switches are per-handle options: mppi_set_option / mppi_get_option in include/mppi_hip_diag.h,
 * the measurement surface (kernel timing, shader clock, launch geometry, option switches) that no binding of the node needs.
 */
#ifndef MPPI_HIP_H
#define MPPI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_ABI_VERSION 5

/* error codes */
#define MPPI_OK 0
#define MPPI_E_INVALID (-1)   /* bad argument / bad handle */
#define MPPI_E_HIP (-2)       /* a HIP runtime call failed (no device, OOM, launch error) */
#define MPPI_E_STATE (-3)     /* call order violated (e.g. update before rollout) */
#define MPPI_E_INTERNAL (-4)
#define MPPI_E_TIMEOUT (-5)   /* a bounded wait expired: the device did not finish in time */

/* storage type of the per-(t,k) intermediates eps and V kept in HBM */
#define MPPI_STORE_F32 0 /* eps fp32, V as fp32 offset from the nominal cost-to-go */
#define MPPI_STORE_F64 1 /* eps fp64, V fp64 (offset form as well)                 */

/* noise source for a rollout */
#define MPPI_NOISE_INJECTED 0 /* use the buffer filled by mppi_upload_noise (reference-RNG parity) */
#define MPPI_NOISE_PHILOX 1   /* device Philox4x32-10 + Box-Muller keyed by (seed, tick, agent, sample): hipRAND's
                                 HIPRAND_RNG_PSEUDO_PHILOX4_32_10 words at subsequence = agent << 32 | tick,
                                 offset = 4 * ((t / 3) << 32 | sample_offset + sample) */

/* dynamics + integrator pairs the reference's `model=` argument can select (control/src/mppi:62) */
#define MPPI_MODEL_DIFFDRIVE_RK4 0   /* rk4 :39-54 over dd_dynamics :23-30 -- what the node runs      */
#define MPPI_MODEL_UNICYCLE_EULER 1  /* euler :57-58 over unicycle_dynamics :33-36 (no theta wrap)   */

/* How mppi_tick / mppi_tick_begin map the K x T rollouts onto the chip:
 *   LANES  one lane per sample, T sequential steps (rollout kernel + update kernel): the throughput path;
 *   SCAN   one wave (T <= 64) or block (T <= 256) per sample, lanes = timesteps, the trajectory as prefix
 *          scans, rollout + cost-to-go + softmax partials in ONE kernel: the latency path for small K
 *          (diff-drive rk4 model only; falls back to LANES where it does not apply);
 *   AUTO   SCAN while n_agents * samples <= 14336 (T <= 64) or 5120 (T <= 256) -- the measured
 *          crossovers --, else LANES.
 * Both give the same results to rounding. */
#define MPPI_TICK_AUTO 0
#define MPPI_TICK_LANES 1
#define MPPI_TICK_SCAN 2

/* Co-scheduled shards (mppi_config.co_shards).  A fused mppi_tick with device noise may split its samples over G
 * engines inside this one handle -- same GPU, one stream each, coupled only by device-side mailbox flags -- so that one
 * shard's HBM-bound update kernel runs under another's VALU-bound rollout (config 4: +3-5 % rollouts/s back to back, -7 us on the
 * blocking call; a K-shard's cost prefix, totals and per-wave noise sums are columns of the handle's own rows -- it allocates its
 * small arrays only, and no 128-byte line holds words of two shards: checked when the group is built).  Results equal
 * the unsplit tick to rounding (sample ids are global, the tuple merge is exact); every other call of this ABI keeps
 * working: after such a tick the tick's V is complete IN PLACE in the handle's arrays -- mppi_download_value / mppi_update read the
 * bytes the shards' update kernels consumed -- and the noise is re-drawn on demand as after any tick; whatever the handle is asked
 * between two split ticks is ordered against the shards' streams both ways.  AUTO = 2 shards for n_agents * samples >= 500000
 * on the lane-per-sample path with at least 32768 samples per agent and n_agents * horizon <= 256 rows (beyond that the
 * shards' publish kernels cost more than the overlap gains), fp32 storage only (the all-fp64 mode's two big kernels are both
 * HBM-bound: split it measured slower).  A handle of SEVERAL agents splits its AGENTS instead (AUTO prefers this wherever each
 * half still holds >= 400 000 sample-agents: config 5's 64 x 16 384, 2 x 500 000 ...): two complete engines, agents [0, ceil(A / 2)) and the rest,
 * NOTHING exchanged (agents are independent controllers) -- per agent bit for bit the one engine's results (the noise streams are
 * keyed by the global agent index), config 5 +8-10 % rollouts/s.  Only the fused tick runs split; any other call first copies the
 * second engine's small results of the last tick (nominal / filtered controls, state, outputs) into the handle's own arrays -- its V already is there --, and the
 * next split tick hands over what changed.  Else none.  An AUTO handle builds its shards with its FIRST fused device-noise
 * mppi_tick -- a handle that only runs the caller's own exchange (the ranks of an N > 1 run), graph replays or injected-noise
 * ticks never pays for the second engine (a few megabytes: both kinds of shard live in the handle's own big arrays); mppi_co_info reports the split from the start, mppi_co_note why a handle
 * that should have split did not.  mppi_tick_begin / _finish (the caller's own exchange), mppi_tick_graph and
 * injected-noise ticks always run unsplit; mppi_p2p_create on such a handle dissolves the group. */

typedef struct mppi_engine mppi_engine;

typedef struct mppi_config {
    uint32_t struct_size;  /* sizeof(mppi_config) as the CALLER compiled it (mppi_default_config fills it in).  mppi_create
                              reads exactly that many bytes over its own defaults: a caller built against an older, shorter
                              struct keeps working when fields are appended (they take their defaults), a size the library
                              does not know (0, shorter than MPPI_CONFIG_SIZE_V5, longer than its own) is MPPI_E_INVALID    */
    int32_t n_agents;      /* A >= 1 independent controllers batched in one engine           */
    int32_t samples;       /* K >= 1 rollouts per agent owned by this engine (MPPI samples=)  */
    int32_t horizon;       /* T >= 5 (Savitzky-Golay window T-1 > 3, control/src/mppi:202; odd T as scipy >= 1.x) */
    int32_t storage;       /* MPPI_STORE_F32 | MPPI_STORE_F64                                 */
    int32_t device;        /* HIP device ordinal                                              */
    uint32_t sample_offset;/* global index of local sample 0                                  */
    int32_t model;         /* MPPI_MODEL_*: the `model=` ctor argument (control/src/mppi:62)   */
    int32_t tick_path;     /* MPPI_TICK_*: which kernels a tick runs; default MPPI_TICK_AUTO          */
    int32_t co_shards;     /* co-scheduled shards of the fused mppi_tick: 0 auto | 1 off | 2..8 (above)  */
    int32_t agent_offset;  /* global index of local agent 0 (independent agents split over engines / GPUs: the device-RNG streams
                              are keyed by the GLOBAL agent index, so a replica rank draws what the one big engine would)   */
    int32_t reserved0;     /* 0 (keeps the doubles below 8-byte aligned without compiler padding)                            */
    double dt;             /* <= 0: 1/T  (control/src/mppi:67)                                */
    double sigma;          /* noise std-dev = sig[0,0] (control/src/mppi:145); default 0.9    */
    double lambda;         /* temperature; default 0.001 (control/src/mppi:89)                */
    double q[3];           /* diag Q  (control/src/mppi:69)                                   */
    double r[2];           /* diag R  (control/src/mppi:71)                                   */
    double p1[3];          /* diag P1 (control/src/mppi:73)                                   */
    double u_max;          /* WHEEL_VEL_MAX  (control/src/mppi:18)                            */
    double wheel_radius;   /* WHEEL_RADIUS   (control/src/mppi:19)                            */
    double wheel_base;     /* WHEEL_BASE     (control/src/mppi:20)                            */
    double floor_w;        /* weight floor 1e-8 (control/src/mppi:193)                        */
    /* -- appended in round 6 (struct_size 176; callers compiled against the 168-byte struct get the default 0) -- */
    int64_t samples_total; /* 0: this handle is the whole controller.  > 0: the controller's samples per agent over ALL handles / ranks
                              it is sharded over (this one owns [sample_offset, sample_offset + samples) of them): every choice the
                              engine makes by SIZE -- which rollout kernel (all-fp64 | mixed precision), lane kernels or scan kernel --
                              is then made from this number, so every rank of an N-way split runs the arithmetic the unsplit
                              controller would and N = 1 / 2 / 4 / 8 end every tick with the same controls to rounding (1e-10;
                              SURVEY 8d-4).  bench.py --gpus N and sharded.make_hip_ticker set it.  The price: a small share runs
                              the kernel sized for the whole (125 000 samples on the mixed rollout: its under-filled forms, +0 us)  */
} mppi_config;
/* the struct as ABI version 5 introduced it: the shortest struct_size mppi_create accepts (fields are only ever appended) */
#define MPPI_CONFIG_SIZE_V5 168u

/* Fill *cfg with the reference defaults (K=10, T=100, constants above) and struct_size = sizeof(mppi_config). */
int mppi_default_config(mppi_config *cfg);

int mppi_abi_version(void);

/* Message of the most recent failure on this handle (or, with h == NULL, of the last
 * failed mppi_create on this thread).  Never NULL. */
const char *mppi_last_error(const mppi_engine *h);

/* MPPI.__init__, control/src/mppi:62-77: allocates device buffers, zero nominal controls. */
int mppi_create(const mppi_config *cfg, mppi_engine **out);
int mppi_destroy(mppi_engine *h);

/* Use an existing hipStream_t (e.g. torch's current stream) for all later work. */
int mppi_set_stream(mppi_engine *h, void *hip_stream);
/* The hipStream_t the engine enqueues on (its own non-blocking stream unless mppi_set_stream changed it). */
int mppi_get_stream(mppi_engine *h, void **hip_stream);

/* Per-call sig / lam of get_path (control/src/mppi:88-89), sig = sigma * I. */
int mppi_set_sigma_lambda(mppi_engine *h, double sigma, double lambda);
/* The same with sig as the full row-major 2 x 2 matrix the reference accepts: the noise of BOTH
 * wheels is drawn with std-dev sig[0][0] (control/src/mppi:143-146), the stage cost uses the whole
 * matrix, lam * u . sig . eps (:184). */
int mppi_set_sig_matrix(mppi_engine *h, const double *sig /*[4]*/, double lambda);

/* Q, R, P1 of the stage / terminal cost (control/src/mppi:69-73; instance attributes the reference reads on every
 * call, :168, :183) as their diagonals q[3], r[2], p1[3]; NULL keeps the current values.  Takes effect with the next
 * rollout. */
int mppi_set_weights(mppi_engine *h, const double *q, const double *r, const double *p1);
/* The same as the FULL row-major matrices the reference multiplies -- (state - desired).T.dot(Q).dot(state - desired),
 * u.T.dot(R).dot(u) (control/src/mppi:181-184), .dot(P1) (:168): Q [3][3], R [2][2], P1 [3][3]; NULL keeps the current one.
 * A quadratic form only sees the symmetric part of its matrix, so (M + M') / 2 is what the kernels carry.  Matrices with
 * off-diagonal terms run the general-cost rollout (all fp64); diagonal ones keep whatever kernel mppi_set_weights would pick. */
int mppi_set_weight_matrices(mppi_engine *h, const double *Q /*[9]*/, const double *R /*[4]*/, const double *P1 /*[9]*/);

/* Deadline of the blocking waits, in milliseconds (0 = wait forever).  See MPPI_E_TIMEOUT. */
int mppi_set_sync_timeout(mppi_engine *h, int milliseconds);

/*
 * EXTENSION (not in the reference's cost; off unless called with weight != 0):
 * an obstacle stage cost read from an occupancy grid in the format map::Grid exports
 * (map/src/map/grid.cpp:126-144: int8 0 free / 50 inflation / 100 occupied; row-major
 * index x + y*width, :251-266; origin = map_min; square cells of `resolution`):
 *     stage cost += weight * cell(x_t, y_t) / 100          (0 outside the grid)
 * for the post-step state of every step, shared by all agents.  cells == NULL or weight == 0
 * removes it.  The grid is copied to the device.
 */
int mppi_set_obstacle_grid(mppi_engine *h, const int8_t *cells, int32_t width, int32_t height,
                           double resolution, double origin_x, double origin_y, double weight);

/* MPPI.initialize, control/src/mppi:79-83: zero the nominal controls of one agent (-1: all). */
int mppi_reset(mppi_engine *h, int agent);

/* What the receding-horizon shift writes into the freed last column of one agent's nominal controls: the
 * reference's uvec_init[:, 0] (control/src/mppi:101), fill [2].  Zeros (the reference's default uvec_init) until called. */
int mppi_set_shift_fill(mppi_engine *h, int agent, const double *fill);

/* latest_uvec [2][T] of one agent (control/src/mppi:81, :100-101). */
int mppi_set_nominal(mppi_engine *h, int agent, const double *uvec);
int mppi_get_nominal(mppi_engine *h, int agent, double *uvec);

/* Injected noise, eps [A][T][2][K] float64 (the list `eps` of control/src/mppi:143-146). */
int mppi_upload_noise(mppi_engine *h, const double *eps);
/* The noise of the last rollout as the kernels used it (after rounding to the storage type). */
int mppi_download_noise(mppi_engine *h, double *eps);

/*
 * MPPI.get_cost2go, control/src/mppi:127-178.  state, goal [A][3] (NULL: keep the
 * device-resident values; for state that is the state predicted by the last tick).
 * Leaves eps and V resident in HBM for mppi_update.  tick_id feeds the RNG counter.
 */
int mppi_rollout(mppi_engine *h, const double *state, const double *goal, int noise_mode,
                 uint64_t seed, uint32_t tick_id);

/* value_fcn [A][T][K] of the last rollout (absolute cost-to-go, float64). */
int mppi_download_value(mppi_engine *h, double *V);
/* Replace the resident V (update_action called with a caller-made value_fcn). */
int mppi_upload_value(mppi_engine *h, const double *V);

/*
 * MPPI.update_action, control/src/mppi:186-208: per-timestep softmax weights over this
 * engine's K samples, weighted-noise update, clip, Savitzky-Golay, clip.  The result
 * becomes the engine's nominal sequence (get_path assigns it to latest_uvec, :90) and is
 * copied to uvec_out [A][2][T] when non-NULL.
 */
int mppi_update(mppi_engine *h, double *uvec_out);
/* What the last mppi_update left BEFORE its Savitzky-Golay step: uvec + the weighted noise, clipped -- [A][2][T].  The reference's
 * update_action writes exactly this into its caller's `uvec` IN PLACE (control/src/mppi:196-199) and returns the filtered sequence
 * as a new array (:202); a binding that wants the method's side effects as well as its result copies this into the caller's array
 * (motion_planning_amd.MPPI.update_action does). */
int mppi_get_unfiltered(mppi_engine *h, double *uvec);

/* MPPI.perform_action, control/src/mppi:210-213: one rk4 step with the nominal u[:,0].
 * state [A][3] (NULL: resident), next_state [A][3]. */
int mppi_plant_step(mppi_engine *h, const double *state, double *next_state);

/* Receding-horizon shift, control/src/mppi:100-101. */
int mppi_shift(mppi_engine *h);

/*
 * MPPI.get_path, control/src/mppi:85-102, split at the only cross-GPU exchange:
 *   tick_begin : nominal baseline -> rollout+cost -> softmax partials -> per-(agent,t)
 *                merged partials of THIS shard, left in the device buffer mppi_partials_ptr
 *                reports (pointer and size in bytes), float64 [A][T][8] =
 *                {min V, sum e, sum e*eps0, sum e*eps1, sum eps0, sum eps1, K_shard, 0};
 *   (host side all-gathers those buffers over RCCL when K is sharded over GPUs)
 *   tick_finish: merges n_shards such buffers (device pointer, [n_shards][A][T][8];
 *                NULL = this engine's own), control update, clip, filter, clip, plant
 *                step, shift.  Asynchronous; results stay on the device.
 * mppi_tick = tick_begin + tick_finish(own partials) + mppi_get_outputs.
 */
int mppi_tick_begin(mppi_engine *h, const double *state, const double *goal, int noise_mode,
                    uint64_t seed, uint32_t tick_id);
int mppi_partials_ptr(mppi_engine *h, void **dev_ptr, size_t *bytes);
int mppi_tick_finish(mppi_engine *h, const void *gathered_dev, int n_shards);
/* Cross-stream ordering for callers that run the exchange on a stream of their own:
 *   mppi_stream_wait_partials: work enqueued on `other_stream` after this call starts only when
 *       everything the engine has enqueued so far (e.g. tick_begin's partials) is complete;
 *   mppi_wait_for_stream: everything the engine enqueues after this call starts only when the work
 *       `other_stream` holds now (e.g. the all-gather into gathered_dev) is complete.
 * other_stream is a hipStream_t (NULL = the legacy null stream). */
int mppi_stream_wait_partials(mppi_engine *h, void *other_stream);
int mppi_wait_for_stream(mppi_engine *h, void *other_stream);
/*
 * One-shot peer-to-peer exchange of the partials for K sharded over the (<= 8) GPUs of ONE node -- the alternative
 * to an RCCL all-gather between tick_begin and tick_finish (the message is 3.2 KB: latency is everything):
 *   mppi_p2p_create   allocates this rank's mailbox (fine-grained device memory, [2][n_ranks] slots + flags) and
 *                     returns its HIP IPC handle in ipc_handle_out (MPPI_IPC_HANDLE_BYTES bytes; NULL: not wanted);
 *   mppi_p2p_connect  maps the peers' mailboxes: ipc_handles = n_ranks handles, rank-major (other processes), and / or
 *                     local_ptrs[g] = mppi_p2p_mailbox_ptr of an engine in THIS process (non-NULL entries win) -- on this GPU or on
 *                     another one of the node: peer access from this engine's device is enabled on the way, MPPI_E_INVALID
 *                     names the pair when the two devices have no peer path;
 *   mppi_tick_exchange_p2p   replaces mppi_tick_finish after mppi_tick_begin: a publish kernel stores this rank's
 *                     tuples into every peer's mailbox over xGMI and raises a flag; the finalize kernel waits for its
 *                     own mailbox's n_ranks flags and finishes the tick.  Asynchronous, no host involvement, no collective;
 *                     a peer that never delivers surfaces as MPPI_E_TIMEOUT from mppi_get_outputs.
 *                     = mppi_p2p_publish + mppi_tick_finish_p2p, which a caller driving SEVERAL engines from one
 *                     thread calls separately (publish on all of them first: a finalize kernel that waits for a
 *                     publish enqueued after it can starve it when the runtime maps both streams to one hardware queue);
 *   mppi_p2p_selftest collective round trips of a known pattern, consumed by a kernel that waits on the flags the way
 *                     the finalize kernel does (set-up time check before trusting the path; blocking; one caller per
 *                     rank -- engines of ONE thread cannot run it against each other).
 * Every rank must run the same sequence of exchanges (epochs are counted per rank).
 */
#define MPPI_IPC_HANDLE_BYTES 64
int mppi_p2p_create(mppi_engine *h, int n_ranks, int rank, void *ipc_handle_out);
int mppi_p2p_connect(mppi_engine *h, const void *ipc_handles, void *const *local_ptrs);
/* mppi_p2p_create + mppi_p2p_connect for ranks that are separate processes with nothing but a file system in common (no process
 * group, no torch): rank r leaves its IPC handle in the file "<path_prefix>.<r>" and waits up to timeout_ms (0: forever) for
 * the other ranks' files.  The prefix must be fresh for every group (e.g. inside a directory the launcher made for this run:
 * a stale file of an earlier run would be taken for a peer's handle); the files are left behind for the launcher to remove
 * once every rank has returned. */
int mppi_p2p_rendezvous(mppi_engine *h, const char *path_prefix, int n_ranks, int rank, int timeout_ms);
int mppi_p2p_mailbox_ptr(mppi_engine *h, void **dev_ptr);
int mppi_p2p_selftest(mppi_engine *h, int rounds);
int mppi_p2p_destroy(mppi_engine *h);
int mppi_p2p_publish(mppi_engine *h);
int mppi_tick_finish_p2p(mppi_engine *h);
int mppi_tick_exchange_p2p(mppi_engine *h);
/* Synchronises; next_state [A][3], u_applied [A][2] (either may be NULL). */
int mppi_get_outputs(mppi_engine *h, double *next_state, double *u_applied);
int mppi_tick(mppi_engine *h, const double *state, const double *goal, int noise_mode,
              uint64_t seed, uint32_t tick_id, double *next_state, double *u_applied);

/* Capture begin+finish (device RNG, resident state/goal) into a hipGraph and replay it:
 * one launch per tick for the launch-bound small-K case.  The tick id of a replay is read from a
 * device-resident counter that every replay advances; eager ticks (mppi_tick, mppi_tick_finish)
 * leave it at their own tick_id + 1, so a node that mixes the two keeps drawing fresh noise.
 * mppi_set_tick_counter overrides it (the id the NEXT replay uses). */
int mppi_tick_graph(mppi_engine *h, uint64_t seed);
int mppi_set_tick_counter(mppi_engine *h, uint32_t next_tick_id);

int mppi_synchronize(mppi_engine *h);

/* Savitzky-Golay operator S [T][T] with u_f = u @ S, as savgol_filter(u, T-1, 3, axis=1)
 * (mode='interp') applies it at control/src/mppi:202.  Host only. */
int mppi_savgol_matrix(int horizon, double *S);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_HIP_H */
