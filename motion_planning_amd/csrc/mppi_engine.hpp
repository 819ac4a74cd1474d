// mppi_engine.hpp -- the engine object behind the C ABI of include/mppi_hip.h: members, the small inline helpers, and the declarations
// of what the four translation units of libmppi_hip.so define:
//     mppi_engine.hip   the core: everything that launches a kernel -- buffers, launch geometry, the pipelines of a tick
//     mppi_co.hip       co-scheduled engines inside one handle (split by samples / by agents)
//     mppi_p2p.hip      the caller's peer-to-peer exchange: mailboxes, IPC handles, rendezvous (C ABI mppi_p2p_*)
//     mppi_abi.hip      the rest of the C ABI: argument checks, call order, error translation
// Mirrors the reference's `MPPI` object (moribots/motion_planning control/src/mppi:61-213): the nominal control sequence
// `latest_uvec` lives on the device between ticks exactly like the Python attribute does.
#pragma once
#include <hip/hip_runtime.h>

#include <errno.h>
#include <signal.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mppi_hip.h"
#include "../../include/mppi_hip_diag.h"
// only the core translation unit emits the non-template kernels of mppi_kernels.hpp; the others see its types and templates
#ifndef MPPI_ENGINE_CORE_TU
#define MPPI_ROLLOUT_TU 1
#endif
#include "mppi_kernels.hpp"
#include "savgol.hpp"

namespace mppi_detail {

struct EngineError {
    int code;
    std::string msg;
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw EngineError{code, buf};
}

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            fail(MPPI_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// Makes the engine's device current for the duration of one ABI call and puts the caller's device back
// afterwards (a process may drive engines on several GPUs, or run torch on another device, from one thread).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev == dev) return;
        hipError_t e = hipSetDevice(dev);
        if (e != hipSuccess) fail(MPPI_E_HIP, "hipSetDevice(%d): %s", dev, hipGetErrorString(e));
        switched = prev >= 0;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

template <typename T>
T* dev_alloc(size_t n, size_t& tally) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(hipMalloc(&p, bytes));
    tally += bytes;
    return static_cast<T*>(p);
}

}  // namespace mppi_detail
using mppi_detail::EngineError;
using mppi_detail::fail;
using mppi_detail::DeviceGuard;
using mppi_detail::dev_alloc;


struct mppi_engine {
    mppi_config cfg{};
    mppi::DevParams P{};
    std::string err = "";
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    size_t hbm_bytes = 0;

    // launch geometry
    int roll_bs = 256, roll_blocks = 0, nterm = 4;
    int NCH = 1, CH = 1024;
    int upd_nv = mppi::kUpdNV;   // the update kernel's vectors per lane (8 | 16: mppi::UpdCfg), by size (pick_update_shape)
    // small-K tick: ONE scan_tick_kernel (lanes = timesteps) instead of rollout + update
    int small_nb = 0, small_spw = 1, small_nw = 1;  // blocks (0 = path not used), samples per unit, waves per unit
    double* d_prev = nullptr;                        // pre-tick {unom [A][2][T], state [A][3], goal [A][3]}
    bool value_lazy = false;                         // the last tick's V exists only as that snapshot + its noise
    const double *ro_state = nullptr, *ro_goal = nullptr, *ro_unom = nullptr;  // rollout inputs override

    // device buffers
    void* d_eps = nullptr;   // S [A][T][2][Ks]
    void* d_dP = nullptr;    // S [A][T][Ks]  exclusive prefix of (stage cost - nominal stage cost)
    void* d_stot = nullptr;  // S [A][Ks]     per-sample total of the same
    void* d_epart = nullptr; // S [A][T][2][Ks/64]  per-wave sums of eps (E of the floor term)
    bool epart_ready = false;
    // device noise that was drawn but not stored (tick path): regenerated on demand from these
    bool eps_lazy = false, lazy_from_counter = false, lazy_counter_bumped = false;
    uint64_t lazy_seed = 0;
    uint32_t lazy_tick = 0;
    bool store_eps_always = false;  // option "store_eps": the tick path writes eps like mppi_rollout does
    long pk_min_samples = 400000;   // the size rule of co-scheduled shards (and of every engine whose option "pk_min_samples" is set)
    bool pk_min_set = false;
    int force_pk = -1;             // >= 0: the size rule is overridden (every shard of a co-scheduled tick takes shard 0's kernel)
    bool last_rollout_pk = false;  // which kernel the last rollout launch was
    int last_rollout_kind = MPPI_ROLLOUT_NONE;   // ... as mppi_rollout_kernel reports it
    int noise_pack = 0;             // option "noise_packing": how a Philox call's bits become normals (mppi::NoisePack): 0 three steps per call, 1 four, 2 hipRAND's normals (two)
    bool use_pk = true;             // option "rollout_pk" = 0: keep the all-fp64 rollout kernel on the tick path (same-box A/B measurements)
    // The fused tick of fp64 storage (rollout_fused.hpp): rollout + cost-to-go + softmax partials in one kernel, V never stored.
    // fused_nb: its waves per agent (0: this engine can never run it -- storage, model, cost, horizon).  It serves the regime of the
    // closed loop in which a row has a handful of samples with weight; parked at the goal (every sample of a row within a few lambda)
    // each of them would cost it a Philox call, so the engine looks at the largest row sum of weights the last finished tick reported
    // (finalize writes it next to the outputs in pinned memory: a host read, no synchronisation, stale by a tick at most) and keeps the
    // two-kernel tick above kFusedRegimeCut.  Option "pk_min_samples" >= 0 replaces both rules by a plain size rule (tests, A/B).
    int fused_nb = 0;
    static constexpr double kFusedRegimeCut = 64.0;
    // (same box, fp64 storage, T = 50, tick us fused / two kernels: 10^6 samples 176.3 / 194-201, 500 000 99.9 / 113.0, 250 000 61.4 / 61.6,
    // 125 000 40.9 / 40.1 -- profiles/r6_ab_fused_f64.txt)
    static constexpr long kFusedMinSamples = 200000;
    // A caller that enqueues ticks without waiting (mppi_tick with NULL outputs) runs ahead of the device: what it reads here is a tick
    // that finished a while ago.  That lag is harmless while the regime drifts (the robot parks over hundreds of ticks), not when the
    // caller starts something new: mppi_set_nominal / mppi_reset (a new plan) put the estimate back to "under way" until a tick
    // enqueued behind them has finished and says otherwise (regime_fresh_start; sequence numbers of the finalize launches).
    bool regime_reset = false;
    uint32_t regime_reset_seq = 0;
    void regime_fresh_start() { regime_reset = true; regime_reset_seq = out_seq + 1u; }
    double regime_weight() {   // the largest row sum of softmax weights the last finished tick reported (0 before the first)
        if (regime_reset) {
            if ((int32_t)(__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) - regime_reset_seq) < 0) return 0.0;
            regime_reset = false;
        }
        double d = 0.0;
        for (int a = 0; a < cfg.n_agents; ++a) {
            const uint64_t bits = __atomic_load_n(reinterpret_cast<const uint64_t*>(h_out) + (size_t)a * 8 + 5, __ATOMIC_RELAXED);
            double v;
            std::memcpy(&v, &bits, sizeof(v));
            d = std::max(d, v);
        }
        return d;
    }
    bool pick_fused(bool ph, bool store);
    void launch_fused(uint64_t seed, uint32_t tick, const uint32_t* tick_ptr);
    double* d_tc = nullptr;  // [A][T][8]   the nominal trajectory's table the LAST rollout used (= tcb[tab])
    double* d_base = nullptr;   //             (= baseb[tab])
    // The table exists twice: a tick's finalize kernel writes the NEXT tick's table (nominal_table_lanes) into the other set while
    // d_tc / d_base still describe the tick just run (mppi_download_value: V = base + Stot - dP); a rollout launch that loads its
    // table takes that set and makes it the current one.
    double* tcb[2] = {nullptr, nullptr};
    double* baseb[2] = {nullptr, nullptr};
    mppi::PkRow* pkb[2] = {nullptr, nullptr};   // [A][T] the deviation-form rows (rollout_pk.hpp) of the same tables
    int tab = 0;
    bool table_valid = false;   // set tab ^ 1 holds the table of (d_state, d_goal, d_unom) as they are now
    int hoist_opt = -1;         // option "table_hoist": -1 by size (hoist_on), 0, 1
    // Where the table pays (same box, tick us without / with it, profiles/r5_ab_table_hoist.jsonl): the prologue it takes out of every
    // rollout workgroup is worth 4-5 us of a launch that runs several rounds of workgroups (config 4 co-scheduled 136.7 -> 133.4,
    // config 5 149.2 -> 146.7; config 4 on one engine 146.4 -> 146.0) and under 1 us of an under-filled one (the other workgroups'
    // waves fill the SIMD while one wave runs the prologue), while the finalize kernel's one wave per agent takes 1.9 us for it at
    // T = 50 and 5.7 at T = 100: 125 000 samples 39.7 -> 40.4, 250 000 54.6 -> 55.7, 500 000 84.4 -> 85.0, config 3 57.9 -> 62.2.
    // AUTO: handles of >= 786 432 sample-agents with T <= 64 -- decided ONCE from the handle's full size (hoist_auto, set by init): the
    // views a co-scheduled tick puts over cfg (ShardView, AgentView) shrink cfg.samples / cfg.n_agents for the duration of shard 0's
    // launches, and its finalize launch runs outside them -- both must take the same decision.  A co-scheduled group
    // decides for its shards (hoist_opt of a sub is the handle's decision).
    bool hoist_auto = false;
    bool hoist_on() const { return hoist_opt >= 0 ? hoist_opt != 0 : hoist_auto; }
    int graph_tab = 0;
    bool table_taken = false;   // this tick's rollout launches already switched to the set they load
    void use_table_set(int t) { tab = t; d_tc = tcb[t]; d_base = baseb[t]; }
    void invalidate_table() { table_valid = false; table_taken = false; }   // (whatever changes d_state / d_goal / d_unom or what the table derives from them)
    double* d_unom = nullptr;
    double* d_ufilt = nullptr;
    double* d_state = nullptr;
    double* d_goal = nullptr;
    double* d_part = nullptr;
    double* d_merged = nullptr;
    double* d_S = nullptr;
    double* d_out = nullptr;
    uint32_t* d_tick = nullptr;
    double* d_fill = nullptr;               // [A][2] what the shift puts into the freed column (uvec_init[:, 0], control/src/mppi:101)
    unsigned long long* d_clk = nullptr;   // {shader cycles, wall-clock ticks} of the last rollout launch's probe wave
    signed char* d_grid = nullptr;
    size_t grid_bytes = 0;
    double* d_tmp = nullptr;
    size_t tmp_elems = 0;

    // pinned staging ring for state/goal uploads
    static constexpr int kRing = 16;
    double* h_stage = nullptr;  // [kRing][A*6]  pinned AND device-mapped: the scan tick reads its inputs straight from here
    double* d_stage_view = nullptr;  // the same ring as the device sees it
    double* h_out = nullptr;    // [A][8] pinned landing zone of mppi_get_outputs / mppi_plant_step (device-mapped: finalize writes it)
    double* d_out_view = nullptr;
    uint32_t* h_seq = nullptr;  // [A] sequence words finalize raises behind its host-side outputs
    uint32_t* d_seq_view = nullptr;
    uint32_t out_seq = 0;       // sequence number of the last finalize that wrote the host-side outputs
    bool out_via_host = false;  // mppi_get_outputs: poll h_seq instead of copying d_out
    const double *in_state = nullptr, *in_goal = nullptr;  // what the FIRST kernel of this tick reads (pinned slot or d_state / d_goal)
    int in_slot = -1;
    // A zero-copy input slot is free again once the tick that read it has finished.  That tick's finalize kernel raises
    // h_seq anyway, so the slot remembers the sequence number to look for (no event record in the stream of a
    // latency-bound tick); a tick that never reaches such a finalize falls back to an event.
    uint32_t slot_seq[kRing]{};
    bool slot_seq_valid[kRing]{};
    int slot_unclaimed = -1;   // slot read by a kernel already enqueued, not yet tied to a finalize's sequence number
    hipEvent_t ring_ev[kRing]{};
    bool ring_used[kRing]{};
    int ring_pos = 0;

    bool noise_ready = false, value_ready = false, partials_ready = false, have_state = false, have_goal = false;
    bool injected_ready = false;   // d_eps holds noise a MPPI_NOISE_INJECTED rollout may read (uploaded, or stored by a rollout)
    double w_off[7] = {0, 0, 0, 0, 0, 0, 0};  // off-diagonal terms of the symmetric parts of Q (01, 02, 12), R (01), P1 (01, 02, 12): mppi_set_weight_matrices
    double sig_cost[4] = {0, 0, 0, 0};  // the sig matrix of the stage cost (sigma * I unless mppi_set_sig_matrix)
    bool sig_is_matrix = false;
    uint32_t last_tick_id = 0;     // id of the last eager tick (its successor is written to d_tick by tick_finish)
    bool last_tick_eager = false;
    hipEvent_t ev_partials = nullptr, ev_foreign = nullptr;  // cross-stream ordering helpers (mppi_stream_wait_*)

    // Blocking waits are polls with a deadline: a control thread must get an error back, not hang, if the
    // device stops answering (MPPI_E_TIMEOUT; mppi_set_sync_timeout, default 10 s, 0 = wait forever).
    int sync_timeout_ms = 10000;
    template <typename Query>
    void bounded_wait(Query query, const char* what) {
        using clock = std::chrono::steady_clock;
        const auto t0 = clock::now();
        for (unsigned spins = 1;; ++spins) {
            const hipError_t e = query();
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) fail(MPPI_E_HIP, "%s: %s", what, hipGetErrorString(e));
            if ((spins & 7u) == 0) {
                const auto us = std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t0).count();
                if (sync_timeout_ms > 0 && us > (long long)sync_timeout_ms * 1000)
                    fail(MPPI_E_TIMEOUT, "%s: the device did not finish within %d ms (the engine must be destroyed)", what, sync_timeout_ms);
                if (us > 2000) { struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }  // long waits: stop burning the core
            }
        }
    }
    void wait_stream(const char* what) {
        hipStream_t st = stream;
        bounded_wait([st] { return hipStreamQuery(st); }, what);
    }
    void wait_event(hipEvent_t ev, const char* what) {
        bounded_wait([ev] { return hipEventQuery(ev); }, what);
    }

    // kernel timing
    uint32_t time_mask = 0;
    int time_period = 1;
    int64_t time_seen[MPPI_KERNEL_COUNT]{};
    struct Pending { int kid; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
    double t_ms[MPPI_KERNEL_COUNT]{};
    int64_t t_n[MPPI_KERNEL_COUNT]{};

    // peer-to-peer exchange of the shard tuples (mppi_p2p_*): see P2PWait in mppi_kernels.hpp
    int p2p_n = 0, p2p_rank = 0;
    char* p2p_mbox = nullptr;            // this rank's mailbox (fine-grained device memory)
    size_t p2p_bytes = 0, p2p_slot = 0;  // total size; bytes of one [n] slot
    char* p2p_peer[8] = {};              // every rank's mailbox as this process sees it ([rank] = own)
    bool p2p_peer_ipc[8] = {};
    bool p2p_connected = false;
    int wall_clock_khz = 100000;         // rate of the device's wall_clock64() (hipDeviceAttributeWallClockRate)
    uint32_t p2p_epoch = 0;
    bool p2p_published = false;          // this epoch's tuples are on their way; mppi_tick_finish_p2p may follow
    mppi::P2PWait p2p_wait{};
    size_t p2p_n_f64() const { return (size_t)cfg.n_agents * cfg.horizon * mppi::kTupleW; }
    double* p2p_data(char* base, int parity, int slot) const {
        return reinterpret_cast<double*>(base + ((size_t)parity * p2p_n + slot) * p2p_slot);
    }
    uint32_t* p2p_flag(char* base, int parity, int slot) const {
        return reinterpret_cast<uint32_t*>(base + (size_t)2 * p2p_n * p2p_slot) + ((size_t)parity * p2p_n + slot) * mppi::kFlagStride;
    }
    void p2p_release() {
        for (int g = 0; g < 8; ++g) {
            if (p2p_peer[g] && p2p_peer_ipc[g]) hipIpcCloseMemHandle(p2p_peer[g]);
            p2p_peer[g] = nullptr; p2p_peer_ipc[g] = false;
        }
        if (p2p_mbox) { hipFree(p2p_mbox); hbm_bytes -= p2p_bytes; p2p_mbox = nullptr; }
        p2p_connected = false; p2p_n = 0;
    }
    mppi::P2PWait p2p_publish(const double* src);

    // Co-scheduled shards (mppi_config.co_shards): the samples of a big engine split over G engines on this one GPU, every
    // engine on its own stream, coupled only through the p2p mailboxes their finalize kernels poll (no event, no host wait
    // between them): one shard's HBM-bound update kernel runs under another's VALU-bound rollout.  THIS engine runs shard
    // 0 of such a tick (so its nominal controls, state and outputs stay the handle's), `subs` the other shards; every
    // other call of the ABI keeps working on this engine's own full-size buffers.
    std::vector<mppi_engine*> subs;
    int co_k0 = 0;             // samples of shard 0
    int co_cut_pct = 58;       // two shards: shard 0's share in per cent (option "co_cut_pct" rebuilds the group)
    bool co_synced = false;    // the subs hold this engine's nominal controls / state / goal
    hipEvent_t ev_co = nullptr;
    // Stream ordering between this engine's stream and the subs' (the subs' big arrays are regions of this engine's own).
    // Back-to-back split ticks need none (every engine's launches follow its own earlier ones).  Anything ELSE this handle is asked
    // to do runs on this engine's stream over the whole arrays, so
    //   co_subs_inflight  the subs have launches enqueued that this engine's stream has not waited for: the next call that is not a
    //                     split tick first makes this stream wait for them (co_join_subs: it may read or rewrite their regions);
    //   co_parent_dirty   this engine's stream has been given such other work since: the next split tick makes every sub's stream
    //                     wait for it before the sub's first launch (co_fence_subs: a re-draw or re-run still writing the sub's
    //                     columns must not meet the sub's next rollout there).
    bool co_subs_inflight = false, co_parent_dirty = false;
    void co_join_subs() {
        if (!co_subs_inflight) return;
        for (auto* e : subs) {
            HIPCHK(hipEventRecord(ev_co, e->stream));
            HIPCHK(hipStreamWaitEvent(stream, ev_co, 0));
        }
        co_subs_inflight = false;
    }
    void co_fence_subs() {
        if (!co_parent_dirty) return;
        HIPCHK(hipEventRecord(ev_co, stream));
        for (auto* e : subs) HIPCHK(hipStreamWaitEvent(e->stream, ev_co, 0));
        co_parent_dirty = false;
    }
    void co_other_call() {   // every ABI call but the split tick itself, the outputs' read-back and the read-only queries (API_BEGIN)
        if (subs.empty()) return;
        co_join_subs();
        co_parent_dirty = true;
    }
    bool co_active() const { return !subs.empty(); }
    void co_release() {
        for (auto* e : subs) delete e;
        subs.clear();
        if (p2p_internal) { p2p_release(); p2p_internal = false; }
        co_agents = false; co_dirty = false; co_value_dirty = false; co_subs_inflight = false; co_parent_dirty = false;
    }
    bool p2p_internal = false;  // the mailboxes belong to the co-scheduled group, not to a caller's cross-GPU exchange
    std::string co_fallback = "";   // why this handle runs unsplit although co-scheduling was possible (mppi_co_note)
    bool is_co_sub = false;     // this engine is a co-scheduled shard inside another handle
    mppi_engine* alias_parent = nullptr; int alias_k0 = 0, alias_a0 = 0;   // (set before init) a co-scheduled shard lives in the parent's big arrays: from column k0 (K split) / from agent a0 (agent split)
    // this engine's view of its own shard while a co-scheduled tick is enqueued: K, chunk count and launch geometry of shard 0
    struct ShardView {
        mppi_engine* e; int K, samples, NCH, roll_blocks;
        explicit ShardView(mppi_engine* e_) : e(e_), K(e_->P.K), samples(e_->cfg.samples), NCH(e_->NCH), roll_blocks(e_->roll_blocks) {
            e->P.K = e->co_k0; e->cfg.samples = e->co_k0; e->NCH = (e->co_k0 + e->CH - 1) / e->CH;
            e->roll_blocks = (e->co_k0 + e->roll_bs - 1) / e->roll_bs;
        }
        ~ShardView() { e->P.K = K; e->cfg.samples = samples; e->NCH = NCH; e->roll_blocks = roll_blocks; }
    };
    void co_sync_subs();
    // ---- the second way of co-scheduling: by AGENTS (a handle of many independent agents, config 5) -------------------------
    // Two complete engines, agents [0, co_a0) on this one and the rest on the sub: nothing is exchanged -- agents are independent
    // (control/src/mppi:296-342: one controller per robot) -- each engine runs rollout, update and finalize for its own agents
    // on its own stream, and one engine's HBM-bound update runs under the other's VALU-bound rollout.  The handle stays the one
    // owner of every per-agent array towards the API: only the fused device-noise mppi_tick runs split; whatever else is called
    // first pulls the sub's results into this engine's arrays (co_pull), and the next split tick pushes what changed (co_push_agents).
    bool co_agents = false;    // the group splits the agents, not the samples
    int co_a0 = 0;             // agents of this engine while a split tick is enqueued
    bool co_dirty = false;     // the sub holds newer nominal / filtered controls, state and outputs of its agents than this engine's arrays (a few KB: pulled by whatever is called next)
    bool co_value_dirty = false;   // ... and a newer V (cost prefix, totals, table, eps sums: ~100 MB at config 5): pulled only by what reads V (co_pull_value)
    double* out_view_ext = nullptr;   // (a sub of an agent split) where its finalize drops the outputs: the handle's pinned rows
    uint32_t* seq_view_ext = nullptr;
    uint32_t seq_ext = 0;
    struct AgentView {   // this engine's view of its own agents while a split tick is enqueued
        mppi_engine* e; int A;
        explicit AgentView(mppi_engine* e_) : e(e_), A(e_->cfg.n_agents) { e->cfg.n_agents = e->co_a0; e->P.A = e->co_a0; e->in_agent_view = true; }
        ~AgentView() { e->cfg.n_agents = A; e->P.A = A; e->in_agent_view = false; }
    };
    bool in_agent_view = false;
    void co_push_agents();
    void co_pull();
    void co_pull_value();
    void co_tick_agents(const double* state, const double* goal, uint64_t seed, uint32_t tick);
    bool co_pending = false;   // co_shards AUTO decided to split: the shards are built with the first fused device-noise tick
    int co_plan(bool& wanted, bool* by_agents = nullptr) const;
    void co_cuts(int G, std::vector<int>& cuts) const;
    void co_check_regions(const mppi_engine* sub) const;
    void co_hand_switches(mppi_engine* e) const {   // what the handle was told since its creation: the deadline and the option switches
        e->sync_timeout_ms = sync_timeout_ms;
        e->store_eps_always = store_eps_always; e->use_pk = use_pk;
        e->pk_min_set = pk_min_set; e->pk_min_samples = pk_min_samples; e->noise_pack = noise_pack;
        e->lanes_zero_copy = lanes_zero_copy; e->hoist_opt = hoist_on() ? 1 : 0;
    }
    void co_build();   // creates the subs
    void co_tick(const double* state, const double* goal, uint64_t seed, uint32_t tick);

    // hipGraph of a whole tick
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    uint64_t graph_seed = 0;
    bool capturing = false;

    bool f64() const { return cfg.storage == MPPI_STORE_F64; }
    size_t esz() const { return f64() ? 8 : 4; }

    hipEvent_t get_event() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        return e;
    }
    void drain_timing() {
        if (pending.empty()) return;
        wait_stream("kernel-timing drain");
        for (auto& p : pending) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
            t_ms[p.kid] += ms; t_n[p.kid] += 1;
            ev_pool.push_back(p.a); ev_pool.push_back(p.b);
        }
        pending.clear();
    }
    struct Scope {  // brackets one kernel launch with events (on the stream it goes to) when its bit is set
        mppi_engine* e; int kid; hipStream_t st; hipEvent_t a = nullptr;
        Scope(mppi_engine* e_, int kid_, hipStream_t st_ = nullptr) : e(e_), kid(kid_), st(st_ ? st_ : e_->stream) {
            if ((e->time_mask & (1u << kid)) && (e->time_seen[kid]++ % e->time_period) == 0) {
                a = e->get_event();
                HIPCHK(hipEventRecord(a, st));
            }
        }
        ~Scope() {  // never throws: a failed end marker only loses one timing sample
            if (!a) return;
            hipEvent_t b = nullptr;
            if (!e->ev_pool.empty()) { b = e->ev_pool.back(); e->ev_pool.pop_back(); }
            else if (hipEventCreate(&b) != hipSuccess) b = nullptr;
            if (b && hipEventRecord(b, st) == hipSuccess) {
                try { e->pending.push_back({kid, a, b}); } catch (...) { hipEventDestroy(a); hipEventDestroy(b); return; }
                if (e->pending.size() >= 4096) { try { e->drain_timing(); } catch (...) {} }
            } else {
                hipEventDestroy(a);
                if (b) hipEventDestroy(b);
            }
        }
    };

    void ensure_tmp(size_t elems) {
        if (elems <= tmp_elems) return;
        if (d_tmp) { wait_stream("staging-buffer regrow"); HIPCHK(hipFree(d_tmp)); hbm_bytes -= tmp_elems * 8; d_tmp = nullptr; tmp_elems = 0; }
        d_tmp = dev_alloc<double>(elems, hbm_bytes);
        tmp_elems = elems;
    }

    void wait_slot_free(int slot) {  // whoever used this ring slot last (a copy, or a kernel reading it in place) is done with it
        if (slot_seq_valid[slot]) {
            const uint32_t want = slot_seq[slot];
            const uint32_t* seqw = h_seq;
            bounded_wait([seqw, want] { return (int32_t)(__atomic_load_n(seqw, __ATOMIC_ACQUIRE) - want) >= 0 ? hipSuccess : hipErrorNotReady; },
                         "state/goal staging ring");
            slot_seq_valid[slot] = false;
        }
        if (ring_used[slot]) { wait_event(ring_ev[slot], "state/goal staging ring"); ring_used[slot] = false; }
    }
    void stage_upload(const double* src, double* dst, size_t n) {
        release_unclaimed_slot();
        const int slot = ring_pos;
        ring_pos = (ring_pos + 1) % kRing;
        wait_slot_free(slot);
        double* h = h_stage + (size_t)slot * cfg.n_agents * 6;
        std::memcpy(h, src, n * sizeof(double));
        HIPCHK(hipMemcpyAsync(dst, h, n * sizeof(double), hipMemcpyHostToDevice, stream));
        HIPCHK(hipEventRecord(ring_ev[slot], stream));
        ring_used[slot] = true;
    }

    // zero_copy: the caller's state / goal are written into a pinned, device-mapped ring slot and the tick's first
    // kernel (scan_tick_kernel) reads them from there over PCIe -- no H2D copy in front of a latency-bound tick
    // (two copies were 14 of the 47 us of a K = 10 tick); that kernel refreshes d_state / d_goal for the later ones.
    // zero_copy on the LANE kernels (the fused tick only: a finalize kernel follows): the rollout's workgroups read the pose / goal
    // straight from the pinned slot too (a few hundred to a few thousand 64-byte reads over PCIe, all in flight at once) and workgroup
    // 0 leaves them in the pre-tick snapshot, where the finalize kernel finds this tick's pose (and refreshes the device-resident
    // goal) -- no fetch launch in front of a blocking tick (its life + the launch boundary: ~3.5 us of the node's call).
    bool lanes_fresh_state = false, lanes_fresh_goal = false;   // this tick's pose / goal live in the snapshot (d_prev), not in d_state / d_goal yet
    void set_inputs(const double* state, const double* goal, bool zero_copy = false);
    // the lane kernels take fresh inputs from the pinned slot when the nominal trajectory is computed inside the rollout (no
    // nominal_kernel reading d_state in front of it) and the launch is not so big that thousands of workgroups would queue on PCIe
    bool lanes_zero_copy = true;   // option "lanes_zero_copy" (0: the fetch launch in front of the rollout, as every other call takes it)
    bool lanes_zero_copy_ok() {
        // (the fused fp64 tick loads its table: its inputs go through the fetch launch)
        return lanes_zero_copy && small_nb == 0 && inline_nominal() && (long)cfg.n_agents * roll_blocks <= 4096 && !capturing &&
               !pick_fused(true, store_eps_always);
    }
    void inputs_consumed() {  // the kernel that reads the pinned slot has been enqueued: the slot is free once it has run
        if (in_slot >= 0) { slot_unclaimed = in_slot; in_slot = -1; }
        in_state = d_state; in_goal = d_goal;
    }
    void release_unclaimed_slot() {  // no finalize took the slot over: guard it with an event after all
        if (slot_unclaimed >= 0) {
            HIPCHK(hipEventRecord(ring_ev[slot_unclaimed], stream));
            ring_used[slot_unclaimed] = true;
            slot_unclaimed = -1;
        }
    }

    void launch_rollout(hipStream_t st, int k0, int k1, bool ph, bool store, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr);
    // write the lazily-drawn noise of the last tick into d_eps (bit-identical re-draw)
    uint32_t lazy_tick_now() {  // the tick id the last (lazy) tick drew its noise with
        uint32_t tick = lazy_tick;
        if (lazy_from_counter) {
            HIPCHK(hipMemcpyAsync(&tick, d_tick, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            wait_stream("tick counter read-back");
            if (lazy_counter_bumped) tick -= 1u;
        }
        return tick;
    }
    void materialise_eps() {
        if (!eps_lazy) return;
        launch_regen(stream, lazy_seed, lazy_tick_now(), nullptr);
        eps_lazy = false; injected_ready = true;
    }
    void materialise_value();
    void launch_scan_tick(bool ph, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr);
    void launch_regen(hipStream_t st, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr);
    void ensure_epart(hipStream_t st);
    void pick_update_shape();
    void launch_update(hipStream_t st, int ch0, int nch, const uint32_t* tick_ptr = nullptr);
    void launch_merge(int nch);
    void check_noise_mode(int noise_mode, bool tick_path = true);
    bool general_cost() const {
        // the lean rollout instantiations are written for the node's cost: Q = diag(q, q, 0), q > 0 (and sane: they scale
        // positions by sqrt(q/2)), no obstacle grid
        return P.q2 != 0.0 || P.grid_weight != 0.0 || P.q0 != P.q1 || !(P.q0 > 1e-100 && P.q0 < 1e100) || P.offdiag != 0;
    }
    bool pick_pk(bool ph, bool store, int k0, int k1) const;
    // rollout + update + merge of one tick
    // The merge launch is skipped when whoever consumes the tuples can merge a handful per row itself -- one launch and
    // one boundary less per tick: the finalize kernel (skip_small_merge: the fused mppi_tick, no exchange follows) or the
    // merging publish kernel of the p2p exchange.  "A handful" = at most kDirectTuples chunk / scan-block tuples per row
    // (K <= 131072 samples on the lane kernels).
    bool merge_skipped = false;
    int direct_n = 0;   // tuples per row in d_part when the merge was skipped
    static constexpr int kDirectTuples = 16;
    void run_pipeline(int noise_mode, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr, bool skip_small_merge = false);
    // T <= 256: the nominal rollout runs inside every rollout block (lanes = timesteps)
    bool inline_nominal() const { return cfg.horizon <= 256 && cfg.model == MPPI_MODEL_DIFFDRIVE_RK4; }
    void run_nominal();
    void run_rollout(int noise_mode, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr);
    void run_update();
    void run_finalize(const double* gathered, int G, int flags, mppi::P2PWait wait = mppi::P2PWait{}, size_t shard_stride = 0);

    void refresh_weights() {   // the cost weights and what the lean rollout step derives from them
        P.q0 = cfg.q[0]; P.q1 = cfg.q[1]; P.q2 = cfg.q[2];
        P.r0 = cfg.r[0]; P.r1 = cfg.r[1];
        P.p0 = cfg.p1[0]; P.p1 = cfg.p1[1]; P.p2 = cfg.p1[2];
        P.q01 = w_off[0]; P.q02 = w_off[1]; P.q12 = w_off[2]; P.r01 = w_off[3]; P.p01 = w_off[4]; P.p02 = w_off[5]; P.p12 = w_off[6];
        P.offdiag = 0;
        for (double v : w_off) if (v != 0.0) P.offdiag = 1;
        P.lean_f = std::sqrt(0.5 * P.q0);
        P.lean_rho = P.lean_f * (P.dt * P.rhalf * (1.0 / 6.0)) / (0.5 * P.kth * P.dt);
        P.lean_inv_f = 1.0 / P.lean_f;
    }
    void refresh_params() {
        P.sigma = cfg.sigma; P.lambda = cfg.lambda; P.inv_lambda = 1.0 / cfg.lambda;
        if (!sig_is_matrix) { sig_cost[0] = sig_cost[3] = cfg.sigma; sig_cost[1] = sig_cost[2] = 0.0; }
        P.sg00 = sig_cost[0]; P.sg01 = sig_cost[1]; P.sg10 = sig_cost[2]; P.sg11 = sig_cost[3];
    }
    // The last tick's noise / V may exist only as "re-draw with these parameters" (eps_lazy, value_lazy):
    // anything that changes what a re-draw or re-run would produce must materialise them first, so that
    // mppi_download_noise / _value keep returning what the last rollout really used.
    void settle_lazy_state() {
        if (co_value_dirty) co_pull_value();   // (an agent split: what is about to change must not change the meaning of the sub's V)
        if (value_lazy && have_state && have_goal) materialise_value();
        materialise_eps();
    }

    void init(const mppi_config& c);

    // the kernels behind the stand-alone calls of the ABI (definitions: mppi_engine.hip)
    void launch_noise_rows(bool pack);
    void launch_value_rows(bool pack);
    void launch_plant();
    void launch_shift();
    void launch_p2p_check(const mppi::P2PWait& w, const double* slots, int n, int slot_f64, double* d_got, int* d_status);

    void destroy_graph() {
        if (graph_exec) { hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        if (graph) { hipGraphDestroy(graph); graph = nullptr; }
    }

    ~mppi_engine();
};

// --------------------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------------------
#define API_BEGIN_FAST(h)                              \
    if (!(h)) return MPPI_E_INVALID;                   \
    try {                                              \
        DeviceGuard dev_guard__((h)->device);
// every call but the split tick itself, the outputs' read-back and the read-only queries first makes this engine's arrays whole again
#define API_BEGIN(h)                                   \
    API_BEGIN_FAST(h)                                  \
        (h)->co_other_call();                          \
        if ((h)->co_dirty) (h)->co_pull();
#define API_END(h)                                                                  \
        return MPPI_OK;                                                             \
    } catch (const EngineError& e) { (h)->err = e.msg; return e.code; }             \
    catch (const std::bad_alloc&) { (h)->err = "host allocation failed"; return MPPI_E_INTERNAL; } \
    catch (const std::exception& e) { (h)->err = e.what(); return MPPI_E_INTERNAL; } \
    catch (...) { (h)->err = "unknown error"; return MPPI_E_INTERNAL; }
