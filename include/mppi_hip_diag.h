/*
 * mppi_hip_diag.h -- the MEASUREMENT surface of libmppi_hip.so: kernel timing, the shader-clock probe, launch geometry, which
 * kernels a handle runs, and the per-handle option switches of tests and same-box A/B runs.  Nothing here is needed to drive the
 * controller: the binding a maintainer of control/src/mppi adds uses include/mppi_hip.h alone.  bench.py, tools/
 * and tests/ include this header; its entry points live in the same library and follow the same conventions (int return codes,
 * never throw).
 */
#ifndef MPPI_HIP_DIAG_H
#define MPPI_HIP_DIAG_H

#include "mppi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernels, for mppi_kernel_timing (the scan kernel is timed as MPPI_KERNEL_ROLLOUT) */
#define MPPI_KERNEL_NOMINAL 0
#define MPPI_KERNEL_ROLLOUT 1
#define MPPI_KERNEL_UPDATE 2
#define MPPI_KERNEL_MERGE 3
#define MPPI_KERNEL_FINALIZE 4
#define MPPI_KERNEL_EXCHANGE 5 /* the p2p publish kernel (the wait for the peers is inside MPPI_KERNEL_FINALIZE) */
#define MPPI_KERNEL_COUNT 6

/*
 * Per-handle switches for tests and same-box A/B runs (none changes results beyond rounding); a co-scheduled handle passes them
 * on to its shards.  SEVEN keys -- every one a code path the engine's own rules also take at some size or call pattern, so that a
 * test can force either side; switches that lost everywhere they were measured are gone (EXPERIMENTS.md lists them with their numbers:
 * k_pieces, upd_nv, fin_threads, upd_skip, pk_waves).  Keys of mppi_set_option (value) / mppi_get_option:
 *   "rollout_pk"      0: fp32-storage ticks stay on the all-fp64 rollout (same-box A/B against the mixed-precision one)
 *   "pk_min_samples"  >= 0: a plain size rule for the mixed-precision rollout; -1 (default): chosen by rounds of waves
 *   "store_eps"       1: the tick path stores its noise like mppi_rollout does
 *   "co_cut_pct"      share of shard 0 of a two-shard co-scheduled handle in per cent (default 58); re-cuts the group
 *   "lanes_zero_copy" 1 (default) | 0: a fused lane-per-sample tick with fresh inputs lets its rollout read them from the pinned slot
 *                     (0: one small fetch launch in front of it, as mppi_rollout / mppi_tick_begin do)
 *   "table_hoist"     -1 (default: by size) | 0 | 1: the nominal trajectory's per-step table of a tick whose inputs are the previous
 *                     tick's own outputs is computed by that tick's finalize kernel instead of by every rollout workgroup's prologue
 *                     (AUTO: handles of >= 786 432 sample-agents at T <= 64, where the prologue costs a launch 4-5 us)
 * and one that selects another noise STREAM (same generator, same counters, other use of its bits):
 *   "noise_packing"   0 (default): one Philox4x32-10 call serves three steps (2 x 21-bit uniforms per step: Box-Muller radius
 *                     <= 5.53 sigma, 2^21 directions); 1: four steps (word j of call t / 4 serves step t: its low 16 bits the radius
 *                     uniform, radius <= 4.85 sigma, its high 16 bits the direction) -- a quarter fewer calls, the mixed-precision
 *                     rollout 7 % shorter; 2: hipRAND's own normals -- two steps per call, mppi_download_noise / sigma =
 *                     hiprand_normal4() of a hiprandStatePhilox4_32_10_t initialised with hiprand_init(seed, agent << 32 | tick,
 *                     4 * ((t / 2) << 32 | global sample)), bit for bit: values (x, y) step t even, (z, w) step t odd, wheels 0, 1
 *                     (32-bit uniforms, radius <= 6.66 sigma, the device library's logf / sqrtf: the rollout a third longer).
 *                     1 and 2 are drawn by the mixed-precision rollout only: fp32 storage, the lane kernels, the node's cost and
 *                     model (2: T >= 32 at dt = 1 / T and sigma = 0.9); MPPI_E_INVALID where they cannot be served (from the option
 *                     call, or from the tick that would need another kernel).  mppi_rollout, mppi_download_noise, mppi_update and
 *                     the oracle's twins follow the option.
 * Unknown keys and out-of-range values return MPPI_E_INVALID.
 */
int mppi_set_option(mppi_engine *h, const char *key, int64_t value);
int mppi_get_option(mppi_engine *h, const char *key, int64_t *value);

/*
 * Kernel timing with HIP events on the engine's stream.  mask = OR of (1 << MPPI_KERNEL_*)
 * to time, 0 = off.  mppi_kernel_times synchronises and returns, per kernel, the summed
 * duration (ms) and the number of launches since timing was (re)enabled.  The rollout kernel's
 * events ride on its own launch (dispatch begin / end timestamps, no marker packets in the
 * stream); the small kernels are bracketed by recorded events.
 * On a handle that runs co-scheduled engines (mppi_co_info) the times are those of the handle's OWN engine:
 * shard 0's samples (mppi_co_info's samples[0]), or, on an agent split, the first ceil(n_agents / 2) agents -- scale
 * by what that engine covers, as bench.py does, before comparing with an unsplit handle.
 */
int mppi_kernel_timing(mppi_engine *h, uint32_t mask);
/* Bracket only every `period`-th launch of each selected kernel (default 1 = every launch):
 * an event pair costs a few microseconds of stream time, sampling keeps a timed region honest. */
int mppi_kernel_timing_period(mppi_engine *h, int period);
int mppi_kernel_times(mppi_engine *h, double *ms /*[MPPI_KERNEL_COUNT]*/,
                      int64_t *launches /*[MPPI_KERNEL_COUNT]*/);

/* Shader clock (MHz) the last lane-per-sample rollout launch ran at: one lane of that launch's middle workgroup reads
 * the shader cycle counter (s_memtime) and the constant-rate counter (s_memrealtime) when its wave starts and ends.
 * 0 before the first such launch.  Synchronises.  (Measurement aid: prices the VALU-issue roofline of bench.py.) */
int mppi_shader_clock(mppi_engine *h, double *mhz);

/* How the fused device-noise mppi_tick of this handle runs: n_shards co-scheduled engines (1: unsplit) and the samples
 * each owns (samples [8], zero-filled behind n_shards).  A handle that splits its AGENTS reports mppi_config.samples for every
 * engine (each rolls out all samples of its agents: engine 0 the first ceil(n_agents / 2) of them). */
int mppi_co_info(mppi_engine *h, int32_t *n_shards, int32_t *samples);
/* Why this handle runs unsplit although co_shards AUTO would have split it (the second set of buffers could not be
 * built), or why a group was dissolved (a co-scheduled tick failed half-way): "" when there is nothing to report.  Never NULL. */
const char *mppi_co_note(const mppi_engine *h);

/* Which rollout kernel this handle's last tick / mppi_rollout launched (a co-scheduled handle: its shards all take the same
 * one; the re-run behind a later mppi_download_value does not count):
 * MPPI_ROLLOUT_NONE before the first; _FP64 the one-sample-per-lane kernel (all arithmetic fp64); _MIXED the
 * mixed-precision two-samples-per-lane kernel (fp32 storage, device noise, the node's cost and model, T <= 256, at the sizes
 * where it is the faster of the two -- from about 262 000 samples); _SCAN the single-kernel small-K tick.  What tests and bench.py label their numbers with. */
#define MPPI_ROLLOUT_NONE 0
#define MPPI_ROLLOUT_FP64 1
#define MPPI_ROLLOUT_MIXED 2
#define MPPI_ROLLOUT_SCAN 3
#define MPPI_ROLLOUT_FUSED 4 /* fp64 storage, device noise: rollout + cost-to-go + softmax partials in one kernel, V never stored (rollout_fused.hpp) */
int mppi_rollout_kernel(mppi_engine *h, int32_t *kind);

/* Bytes of HBM held by the engine, and the launch geometry (blocks) of a tick's kernels: rollout +
 * update on the lane-per-sample path; the scan kernel and update_blocks = 0 on the small-K path. */
int mppi_engine_info(mppi_engine *h, size_t *hbm_bytes, int32_t *rollout_blocks,
                     int32_t *update_blocks);

/* Diagnostic builds only (make -C motion_planning_amd/csrc PROBE=1 -> lib/libmppi_hip_probe.so): the probe wave of the last
 * lane-per-sample rollout launch (thread 0 of the middle workgroup) stamps the shader cycle counter behind its prologue's
 * barrier, behind every chunk of steps and at its end; cycles [n] receives the stamps relative to the wave's start (0 where
 * the wave never got to), *total the wave's whole life.  The product library returns zeros.  Synchronises. */
#define MPPI_PROBE_MARKS 30
int mppi_probe_timeline(mppi_engine *h, uint64_t *cycles /*[MPPI_PROBE_MARKS]*/, uint64_t *total);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_HIP_DIAG_H */
