// mppi_engine.hip -- host side of libmppi_hip.so: the engine object behind the C ABI of
// include/mppi_hip.h.  Owns the HBM buffers, picks launch geometry for gfx950 and enqueues
// the kernels of mppi_kernels.hpp on one HIP stream.  Mirrors the reference's `MPPI` object
// (moribots/motion_planning control/src/mppi:61-213): the nominal control sequence
// `latest_uvec` lives on the device between ticks exactly like the Python attribute does.
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
#include "mppi_kernels.hpp"
#include "rollout_launch.hpp"
#include "rollout_pk.hpp"
#include "savgol.hpp"

namespace {

struct EngineError {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const char* fmt, ...) {
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

thread_local std::string g_create_error = "";

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

}  // namespace

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
    // publish this rank's tuples for the next epoch and return the wait descriptor for the consumer.
    // src = merged tuples [A][T][8]; src == nullptr: merge d_part's direct_n tuples per row on the way (merge_skipped)
    mppi::P2PWait p2p_publish(const double* src) {
        if (!p2p_connected) fail(MPPI_E_STATE, "p2p exchange is not connected (mppi_p2p_create + mppi_p2p_connect)");
        p2p_epoch += 1u;
        const int par = (int)(p2p_epoch & 1u);
        mppi::P2PPeers peers{};
        for (int g = 0; g < p2p_n; ++g) { peers.data[g] = p2p_data(p2p_peer[g], par, p2p_rank); peers.flag[g] = p2p_flag(p2p_peer[g], par, p2p_rank); }
        Scope sc(this, MPPI_KERNEL_EXCHANGE);
        if (src) hipLaunchKernelGGL(mppi::p2p_publish_kernel, dim3(p2p_n), dim3(256), 0, stream, src, (int)p2p_n_f64(), peers, p2p_epoch);
        else hipLaunchKernelGGL(mppi::p2p_publish_merge_kernel, dim3(p2p_n), dim3(256), 0, stream, P, (const double*)d_part, direct_n,
                                cfg.n_agents * cfg.horizon, peers, p2p_epoch);
        HIPCHK(hipGetLastError());
        mppi::P2PWait w{};
        w.flags = p2p_flag(p2p_mbox, par, 0); w.n = p2p_n; w.epoch = p2p_epoch;
        w.timeout_ticks = sync_timeout_ms > 0 ? (unsigned long long)sync_timeout_ms * (unsigned long long)wall_clock_khz : 0ull;
        return w;
    }

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
    void co_sync_subs() {   // (rare) something other than a co-scheduled tick changed this engine's nominal controls / state / goal
        if (co_synced) return;
        const size_t A_ = cfg.n_agents, T_ = cfg.horizon;
        HIPCHK(hipEventRecord(ev_co, stream));
        for (auto* e : subs) {
            HIPCHK(hipStreamWaitEvent(e->stream, ev_co, 0));
            HIPCHK(hipMemcpyAsync(e->d_unom, d_unom, A_ * 2 * T_ * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->d_state, d_state, A_ * 3 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->d_goal, d_goal, A_ * 3 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
            e->have_state = have_state; e->have_goal = have_goal;
            e->invalidate_table();
        }
        co_synced = true;
    }
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
    void set_inputs(const double* state, const double* goal, bool zero_copy = false) {
        const size_t n = (size_t)cfg.n_agents * 3;
        if (state || goal) invalidate_table();   // a fresh pose / goal: not what the last finalize kernel prepared the table for
        in_state = d_state; in_goal = d_goal; in_slot = -1;
        lanes_fresh_state = lanes_fresh_goal = false;
        if (zero_copy && (state || goal)) {
            release_unclaimed_slot();
            const int slot = ring_pos;
            ring_pos = (ring_pos + 1) % kRing;
            wait_slot_free(slot);
            double* h = h_stage + (size_t)slot * cfg.n_agents * 6;
            const double* dv = d_stage_view + (size_t)slot * cfg.n_agents * 6;
            if (state) { std::memcpy(h, state, n * sizeof(double)); in_state = dv; have_state = true; }
            if (goal) { std::memcpy(h + n, goal, n * sizeof(double)); in_goal = dv + n; have_goal = true; }
            in_slot = slot;
            if (small_nb == 0) { lanes_fresh_state = state != nullptr; lanes_fresh_goal = goal != nullptr; }
        } else if (state || goal) {
            // lane-per-sample tick: the inputs go into a pinned slot as well, and ONE small kernel moves them to d_state /
            // d_goal (two H2D copies cost ~10 us more in front of a blocking tick)
            release_unclaimed_slot();
            const int slot = ring_pos;
            ring_pos = (ring_pos + 1) % kRing;
            wait_slot_free(slot);
            double* h = h_stage + (size_t)slot * cfg.n_agents * 6;
            const double* dv = d_stage_view + (size_t)slot * cfg.n_agents * 6;
            if (state) { std::memcpy(h, state, n * sizeof(double)); have_state = true; }
            if (goal) { std::memcpy(h + n, goal, n * sizeof(double)); have_goal = true; }
            hipLaunchKernelGGL(mppi::fetch_inputs_kernel, dim3(((int)n + 63) / 64), dim3(64), 0, stream, state ? dv : nullptr,
                               goal ? dv + n : nullptr, d_state, d_goal, (int)n);
            HIPCHK(hipGetLastError());
            slot_unclaimed = slot;   // free once that kernel has run: tied to this tick's finalize, or to an event
        }
        if (!have_state || !have_goal) fail(MPPI_E_STATE, "state/goal passed as NULL before ever being set");
    }
    // the lane kernels take fresh inputs from the pinned slot when the nominal trajectory is computed inside the rollout (no
    // nominal_kernel reading d_state in front of it) and the launch is not so big that thousands of workgroups would queue on PCIe
    bool lanes_zero_copy = true;   // option "lanes_zero_copy" (0: the fetch launch in front of the rollout, as every other call takes it)
    bool lanes_zero_copy_ok() const {
        return lanes_zero_copy && small_nb == 0 && inline_nominal() && (long)cfg.n_agents * roll_blocks <= 4096 && !capturing;
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

    void launch_rollout(hipStream_t st, int k0, int k1, bool ph, bool store, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
        mppi::RolloutArgs a{};
        // timing: the launch carries its own start / stop events (no marker packets in the stream)
        if ((time_mask & (1u << MPPI_KERNEL_ROLLOUT)) && (time_seen[MPPI_KERNEL_ROLLOUT]++ % time_period) == 0) {
            a.ev_start = get_event();
            a.ev_stop = get_event();
        }
        a.P = P; a.stream = st; a.k0 = k0; a.k1 = k1; a.philox = ph; a.store_eps = store;
        a.model = cfg.model;
        // the table the previous tick's finalize kernel left for exactly these inputs: load it (no prologue); that set becomes the
        // current one (d_tc / d_base: what mppi_download_value adds to the stored offsets)
        const bool load_table = hoist_on() && table_valid && inline_nominal() && !ro_state && !ro_goal && !ro_unom && !capturing;
        if (load_table && !table_taken) { use_table_set(tab ^ 1); table_taken = true; }
        a.inline_nominal = !inline_nominal() || load_table ? 0 : (cfg.horizon <= 64 ? 1 : 2);
        a.general = general_cost();
        a.seed = seed; a.tick = tick; a.tick_ptr = tick_ptr;
        a.state = ro_state ? ro_state : (in_state ? in_state : d_state); a.goal = ro_goal ? ro_goal : (in_goal ? in_goal : d_goal);
        a.unom = ro_unom ? ro_unom : d_unom; a.tc = d_tc; a.base = d_base;
        a.eps = d_eps; a.dP = d_dP; a.stot = d_stot; a.epart = d_epart;
        hipError_t e;
        // the tick path of an fp32-storage engine with the node's own cost and model: the mixed-precision kernel, two
        // samples per lane on the packed-fp32 pipe (rollout_pk.hpp); its heading series need the noise's reach bounded
        // ... and enough waves: it halves their number and doubles their length, which only pays when every SIMD still gets
        // several (same-box A/B at T = 50, rollout_kernel vs this one: 10^6 samples 106.8 vs 101.8 us, 750 000 83.0 vs 79.4,
        // 500 000 58.2 vs 56.9, 375 000 47.1 vs 46.3, 250 000 34.5 vs 36.1, 125 000 24.6 vs 28.3)
        // Which of the two is faster at a given size is a matter of ROUNDS OF WAVES: a launch takes as long as its busiest SIMD,
        // i.e. ceil(blocks / 256 CUs) waves of the kernel's length, and a wave of this kernel (128 samples) costs 1.9 waves of the
        // other (64 samples).  One engine, T = 50, tick us all-fp64 / mixed, sizes chosen around whole rounds (r = blocks / 256):
        //   393 216 (r 3.00) 75.2 / 73.8   400 000 (3.05) 78.5 / 83.0   430 000 (3.28) 80.2 / 83.8   460 000 (3.51) 85.7 / 83.6
        //   560 000 (4.27) 98.1 / 100.3    600 000 (4.58) 103.8 / 100.5  700 000 (5.34) 114.5 / 116.0  750 000 (5.72) 122.3 / 117.7
        //   850 000 (6.49) 128.8 / 130.8   900 000 (6.87) 136.6 / 134.1  10^6 (7.63) 154.8 / 150.1   1 048 576 (8.00) 157.3 / 150.7
        // -- 1.9 ceil(r_mixed) < ceil(r_fp64) picks the faster kernel at 21 of the 23 sizes measured (300 000 ... 1 200 000; 0.7 % and 2.2 % slower at the other two).
        // Below three rounds the long waves lose to latency whatever the rounds say (250 000: 34.5 vs 36.1 us).  Shards of a
        // co-scheduled handle fill each other's gaps and keep the plain size rule (measured: 138-139 us against 142-145 per tick);
        // so does an engine whose option "pk_min_samples" is set (tests, A/B runs).  (pick_pk)
        const bool pk = pick_pk(ph, store, k0, k1);
        if (noise_pack && ph && !pk)
            fail(MPPI_E_INVALID, "noise_packing 1 / 2 is drawn by the mixed-precision rollout only: all samples of an fp32-storage engine, noise not stored "
                                 "(option store_eps 0), the node's cost (Q = diag(q, q, 0), no obstacle grid), rk4 / diff drive, T <= 256 with sigma small enough for its series");
        last_rollout_pk = pk;
        last_rollout_kind = pk ? MPPI_ROLLOUT_MIXED : MPPI_ROLLOUT_FP64;
        if (pk) {
            mppi::RolloutPkArgs b{};
            b.P = P; b.stream = st; b.inline_nominal = a.inline_nominal; b.seed = seed; b.tick = tick; b.tick_ptr = tick_ptr;
            b.state = a.state; b.goal = a.goal; b.unom = a.unom; b.tc = d_tc; b.base = d_base;
            b.dP = static_cast<float*>(d_dP); b.stot = static_cast<float*>(d_stot); b.epart = static_cast<float*>(d_epart);
            b.al_guard = mppi::rollout_pk_guard(P.kth, P.dt, P.sigma, noise_pack);
            b.pkrows = pkb[tab];
            b.noise_pack = noise_pack;
            b.ev_start = a.ev_start; b.ev_stop = a.ev_stop;
            e = mppi::launch_rollout_pk(b);
        } else
        if (f64()) e = nterm == 4 ? mppi::launch_rollout_typed<double, 4>(a) : nterm == 7 ? mppi::launch_rollout_typed<double, 7>(a) : mppi::launch_rollout_typed<double, 0>(a);
        else e = nterm == 4 ? mppi::launch_rollout_typed<float, 4>(a) : nterm == 7 ? mppi::launch_rollout_typed<float, 7>(a) : mppi::launch_rollout_typed<float, 0>(a);
        if (a.ev_start) {
            if (e == hipSuccess) {
                pending.push_back({MPPI_KERNEL_ROLLOUT, a.ev_start, a.ev_stop});
                if (pending.size() >= 4096) drain_timing();
            } else {
                ev_pool.push_back(a.ev_start); ev_pool.push_back(a.ev_stop);
            }
        }
        if (e != hipSuccess) fail(MPPI_E_HIP, "rollout launch failed: %s", hipGetErrorString(e));
    }
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
    // The small-K tick keeps V in registers.  When a caller asks for it afterwards (mppi_download_value,
    // mppi_update), the lane-per-sample rollout kernel re-runs the tick's rollout from the pre-tick
    // snapshot the scan kernel left behind, with the same noise (re-drawn bit-identically, or the
    // injected buffer), and leaves dP / Stot / base / epart as any rollout does.  (The scan tick only: a co-scheduled
    // tick's V is complete in this handle's own arrays -- its shards fill columns of them -- and is read in place.)
    void materialise_value() {
        if (!value_lazy) return;
        const bool ph = eps_lazy;
        const uint32_t tick = ph ? lazy_tick_now() : 0u;
        ro_unom = d_prev;
        ro_state = d_prev + (size_t)cfg.n_agents * 2 * cfg.horizon;
        ro_goal = ro_state + (size_t)cfg.n_agents * 3;
        const int kind_of_the_tick = last_rollout_kind;   // (the re-run is not what mppi_rollout_kernel reports)
        try {
            launch_rollout(stream, 0, cfg.samples, ph, true, lazy_seed, tick, nullptr);
        } catch (...) {
            ro_unom = ro_state = ro_goal = nullptr;
            throw;
        }
        ro_unom = ro_state = ro_goal = nullptr;
        last_rollout_kind = kind_of_the_tick;
        if (ph) { eps_lazy = false; injected_ready = true; }
        value_lazy = false; value_ready = true; epart_ready = true;
    }
    void launch_scan_tick(bool ph, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
        Scope sc(this, MPPI_KERNEL_ROLLOUT);
        dim3 grid(small_nb, cfg.n_agents);
#define LAUNCH_SCAN(TYPE, NW, PH)                                                                                       \
    hipLaunchKernelGGL((mppi::scan_tick_kernel<TYPE, NW, PH>), grid, dim3(256), 0, stream, P, in_state ? in_state : (const double*)d_state, \
                       in_goal ? in_goal : (const double*)d_goal, (const double*)d_unom, static_cast<const TYPE*>(d_eps), seed, tick,      \
                       tick_ptr, small_spw, d_part, small_nb, d_prev, d_state, d_goal)
#define LAUNCH_SCAN_T(TYPE)                                                                \
    do {                                                                                   \
        if (small_nw == 1) { if (ph) LAUNCH_SCAN(TYPE, 1, true); else LAUNCH_SCAN(TYPE, 1, false); } \
        else { if (ph) LAUNCH_SCAN(TYPE, 4, true); else LAUNCH_SCAN(TYPE, 4, false); }     \
    } while (0)
        if (f64()) LAUNCH_SCAN_T(double); else LAUNCH_SCAN_T(float);
#undef LAUNCH_SCAN_T
#undef LAUNCH_SCAN
        HIPCHK(hipGetLastError());
        last_rollout_kind = MPPI_ROLLOUT_SCAN;
        inputs_consumed();
    }
    void launch_regen(hipStream_t st, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
        const int spd = noise_pack == 1 ? mppi::NoisePack<1>::kSteps : (noise_pack == 2 ? mppi::NoisePack<2>::kSteps : mppi::NoisePack<0>::kSteps);
        dim3 g((cfg.samples + 255) / 256, (cfg.horizon + spd - 1) / spd, cfg.n_agents);
        if (f64()) hipLaunchKernelGGL(mppi::eps_regen_kernel<double>, g, dim3(256), 0, st, P, static_cast<double*>(d_eps), seed, tick, tick_ptr);
        else if (noise_pack == 1) hipLaunchKernelGGL((mppi::eps_regen_kernel<float, 1>), g, dim3(256), 0, st, P, static_cast<float*>(d_eps), seed, tick, tick_ptr);
        else if (noise_pack == 2) hipLaunchKernelGGL((mppi::eps_regen_kernel<float, 2>), g, dim3(256), 0, st, P, static_cast<float*>(d_eps), seed, tick, tick_ptr);
        else hipLaunchKernelGGL(mppi::eps_regen_kernel<float>, g, dim3(256), 0, st, P, static_cast<float*>(d_eps), seed, tick, tick_ptr);
        HIPCHK(hipGetLastError());
    }
    void ensure_epart(hipStream_t st) {
        if (epart_ready) return;  // noise uploaded by the caller and never rolled out: sum it now
        dim3 g((cfg.samples + 255) / 256, cfg.horizon * 2, cfg.n_agents);
        if (f64()) hipLaunchKernelGGL(mppi::eps_wavesum_kernel<double>, g, dim3(256), 0, st, P, static_cast<const double*>(d_eps), static_cast<double*>(d_epart));
        else hipLaunchKernelGGL(mppi::eps_wavesum_kernel<float>, g, dim3(256), 0, st, P, static_cast<const float*>(d_eps), static_cast<float*>(d_epart));
        HIPCHK(hipGetLastError());
        epart_ready = true;
    }
    // chunk length of the update kernel: twice the streaming shape's where that is what takes a row to <= kDirectTuples chunk tuples
    // (no merge launch in the fused tick); d_part is sized for the SHORT chunks, so the choice can change with the option
    void pick_update_shape() {
        const int ch8 = f64() ? mppi::UpdCfg<double, 8>::CH : mppi::UpdCfg<float, 8>::CH;
        const int n8 = (cfg.samples + ch8 - 1) / ch8, n16 = (cfg.samples + 2 * ch8 - 1) / (2 * ch8);
        upd_nv = (n8 > kDirectTuples && n16 <= kDirectTuples) ? 16 : 8;
        if (noise_pack) upd_nv = 8;   // (the other noise packings' re-draws are built into the streaming shape only)
        CH = ch8 * upd_nv / 8;
        NCH = (cfg.samples + CH - 1) / CH;
    }
    void launch_update(hipStream_t st, int ch0, int nch, const uint32_t* tick_ptr = nullptr) {
        ensure_epart(st);
        Scope sc(this, MPPI_KERNEL_UPDATE, st);
        dim3 grid(8 * cfg.horizon, (cfg.n_agents * nch + 7) / 8);  // XCD-aware decode inside the kernel
#define LAUNCH_UPD(TYPE, REGEN)                                                                                  \
    hipLaunchKernelGGL((mppi::update_kernel<TYPE, REGEN>), grid, dim3(256), 0, st, P, static_cast<const TYPE*>(d_eps), \
                       static_cast<const TYPE*>(d_dP), static_cast<const TYPE*>(d_stot), d_part, NCH, ch0, nch,    \
                       static_cast<const TYPE*>(d_epart), lazy_seed, lazy_tick, tick_ptr)
#define LAUNCH_UPD_PACK(PK)                                                                                               \
    hipLaunchKernelGGL((mppi::update_kernel<float, true, PK>), grid, dim3(256), 0, st, P, static_cast<const float*>(d_eps), \
                       static_cast<const float*>(d_dP), static_cast<const float*>(d_stot), d_part, NCH, ch0, nch,           \
                       static_cast<const float*>(d_epart), lazy_seed, lazy_tick, tick_ptr)
#define LAUNCH_UPD16(TYPE, REGEN)                                                                                  \
    hipLaunchKernelGGL((mppi::update_kernel<TYPE, REGEN, 0, 16>), grid, dim3(256), 0, st, P, static_cast<const TYPE*>(d_eps), \
                       static_cast<const TYPE*>(d_dP), static_cast<const TYPE*>(d_stot), d_part, NCH, ch0, nch,    \
                       static_cast<const TYPE*>(d_epart), lazy_seed, lazy_tick, tick_ptr)
        if (upd_nv == 16) {
            if (f64()) { if (eps_lazy) LAUNCH_UPD16(double, true); else LAUNCH_UPD16(double, false); }
            else { if (eps_lazy) LAUNCH_UPD16(float, true); else LAUNCH_UPD16(float, false); }
        } else
        if (f64()) { if (eps_lazy) LAUNCH_UPD(double, true); else LAUNCH_UPD(double, false); }
        else if (eps_lazy && noise_pack == 1) LAUNCH_UPD_PACK(1);
        else if (eps_lazy && noise_pack == 2) LAUNCH_UPD_PACK(2);
        else { if (eps_lazy) LAUNCH_UPD(float, true); else LAUNCH_UPD(float, false); }
#undef LAUNCH_UPD16
#undef LAUNCH_UPD_PACK
#undef LAUNCH_UPD
        HIPCHK(hipGetLastError());
    }
    void launch_merge(int nch) {
        Scope sc(this, MPPI_KERNEL_MERGE);
        hipLaunchKernelGGL(mppi::merge_kernel, dim3(cfg.horizon, cfg.n_agents), dim3(nch > 128 ? 256 : 64), 0, stream, P, d_part, nch, d_merged);
        HIPCHK(hipGetLastError());
    }
    // Everything a tick / rollout can refuse for, checked BEFORE any state of the handle changes (inputs staged, lazy-noise bookkeeping,
    // a co-scheduled group half way through its launches): a refused call leaves the handle exactly as it was.
    // tick_path: the call is a tick (its device noise is not stored unless option store_eps says so); else mppi_rollout
    void check_noise_mode(int noise_mode, bool tick_path = true) {
        if (noise_mode == MPPI_NOISE_INJECTED && !injected_ready)
            fail(MPPI_E_STATE, "MPPI_NOISE_INJECTED but no noise is resident (mppi_upload_noise, or a rollout that stored its noise)");
        if (noise_mode != MPPI_NOISE_INJECTED && noise_mode != MPPI_NOISE_PHILOX)
            fail(MPPI_E_INVALID, "unknown noise_mode %d", noise_mode);
        if (noise_mode == MPPI_NOISE_PHILOX && noise_pack && small_nb == 0) {
            // (mppi_rollout draws with the mixed kernel and re-draws the noise into d_eps: it never asks that kernel to store)
            const bool store = tick_path && store_eps_always;
            const int forced = force_pk;
            force_pk = -1;
            const bool ok = pick_pk(true, store, 0, cfg.samples);
            force_pk = forced;
            if (!ok)
                fail(MPPI_E_INVALID, "noise_packing 1 / 2 is drawn by the mixed-precision rollout only: all samples of an fp32-storage engine, noise not stored "
                                     "(option store_eps 0), the node's cost (Q = diag(q, q, 0), no obstacle grid), rk4 / diff drive, T <= 256 with sigma small enough for its series");
        }
    }
    bool general_cost() const {
        // the lean rollout instantiations are written for the node's cost: Q = diag(q, q, 0), q > 0 (and sane: they scale
        // positions by sqrt(q/2)), no obstacle grid
        return P.q2 != 0.0 || P.grid_weight != 0.0 || P.q0 != P.q1 || !(P.q0 > 1e-100 && P.q0 < 1e100) || P.offdiag != 0;
    }
    // which of the two lane-per-sample rollouts a device-noise tick of this engine takes (see launch_rollout)
    bool pick_pk(bool ph, bool store, int k0, int k1) const {
        bool pk_size;
        if (noise_pack) pk_size = true;   // (the only kernel that draws that stream)
        else if (force_pk >= 0) pk_size = force_pk != 0;
        // a shard of a controller split over handles / ranks (mppi_config.samples_total): the size that decides is the WHOLE controller's
        else if (cfg.samples_total > 0) pk_size = (long)cfg.n_agents * cfg.samples_total >= pk_min_samples;
        else if (pk_min_set || co_active() || is_co_sub) pk_size = (long)cfg.n_agents * cfg.samples >= pk_min_samples;
        else {
            const long r_pk = ((long)cfg.n_agents * ((cfg.samples + 511) / 512) + 255) / 256;
            const long r_64 = ((long)cfg.n_agents * ((cfg.samples + 255) / 256) + 255) / 256;
            pk_size = r_pk >= 3 && 19 * r_pk < 10 * r_64;
        }
        return (use_pk || noise_pack) && !f64() && ph && !store && inline_nominal() && !general_cost() && k0 == 0 && k1 == cfg.samples && pk_size &&
               mppi::rollout_pk_applies(P.kth, P.dt, P.sigma, cfg.horizon, noise_pack);
    }
    // rollout + update + merge of one tick
    // The merge launch is skipped when whoever consumes the tuples can merge a handful per row itself -- one launch and
    // one boundary less per tick: the finalize kernel (skip_small_merge: the fused mppi_tick, no exchange follows) or the
    // merging publish kernel of the p2p exchange.  "A handful" = at most kDirectTuples chunk / scan-block tuples per row
    // (K <= 131072 samples on the lane kernels).
    bool merge_skipped = false;
    int direct_n = 0;   // tuples per row in d_part when the merge was skipped
    static constexpr int kDirectTuples = 16;
    void run_pipeline(int noise_mode, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr, bool skip_small_merge = false) {
        check_noise_mode(noise_mode);
        if (!in_agent_view) co_value_dirty = false;   // (V of EVERY agent is about to be this engine's own: nothing of a sub's is wanted any more)
        const bool ph = noise_mode == MPPI_NOISE_PHILOX;
        const bool store = !ph || store_eps_always;
        eps_lazy = ph && !store;
        if (ph && store) injected_ready = true;   // this rollout leaves its noise in d_eps
        lazy_seed = seed; lazy_tick = tick; lazy_from_counter = tick_ptr != nullptr; lazy_counter_bumped = false;
        last_tick_id = tick; last_tick_eager = tick_ptr == nullptr;
        epart_ready = true;  // every rollout launch below writes its waves' eps sums
        if (small_nb > 0) {  // small K: rollout + cost-to-go + softmax partials in one kernel, V stays in registers
            eps_lazy = ph;
            if (ph) injected_ready = false;  // the scan kernel never writes d_eps
            launch_scan_tick(ph, seed, tick, tick_ptr);
            merge_skipped = (skip_small_merge || (p2p_connected && !p2p_internal)) && small_nb <= kDirectTuples;
            direct_n = small_nb;
            if (!merge_skipped) launch_merge(small_nb);
            noise_ready = true; value_ready = false; value_lazy = true; partials_ready = true; epart_ready = false;
            return;
        }
        merge_skipped = (skip_small_merge || (p2p_connected && !p2p_internal)) && NCH <= kDirectTuples;
        direct_n = NCH;
        {
            double* const snap_was = P.snap;
            if (lanes_fresh_state || lanes_fresh_goal) P.snap = d_prev;   // (workgroup 0 keeps the inputs it read from the pinned slot)
            try { launch_rollout(stream, 0, cfg.samples, ph, store, seed, tick, tick_ptr); } catch (...) { P.snap = snap_was; throw; }
            P.snap = snap_was;
            if (in_slot >= 0) inputs_consumed();
        }
        launch_update(stream, 0, NCH, tick_ptr);
        if (!merge_skipped) launch_merge(NCH);
        noise_ready = true; value_ready = true; value_lazy = false; partials_ready = true; epart_ready = true;
    }
    // T <= 256: the nominal rollout runs inside every rollout block (lanes = timesteps)
    bool inline_nominal() const { return cfg.horizon <= 256 && cfg.model == MPPI_MODEL_DIFFDRIVE_RK4; }
    void run_nominal() {
        if (inline_nominal()) return;
        Scope sc(this, MPPI_KERNEL_NOMINAL);
        hipLaunchKernelGGL(mppi::nominal_kernel, dim3(cfg.n_agents), dim3(mppi::kNomThreads),
                           (size_t)cfg.horizon * sizeof(double), stream, P, d_state, d_goal, d_unom, d_tc, d_base);
        HIPCHK(hipGetLastError());
    }
    void run_rollout(int noise_mode, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
        check_noise_mode(noise_mode, /*tick_path=*/false);
        co_value_dirty = false;   // (V of every agent is about to be this engine's own)
        const bool ph = noise_mode == MPPI_NOISE_PHILOX;
        // (the 16-bit packing is drawn by the mixed-precision kernel, which does not store its noise: the re-draw kernel leaves the same bits in d_eps)
        launch_rollout(stream, 0, cfg.samples, ph, !(ph && noise_pack), seed, tick, tick_ptr);
        if (ph && noise_pack) launch_regen(stream, seed, tick, tick_ptr);
        eps_lazy = false; injected_ready = true;
        noise_ready = true; value_ready = true; value_lazy = false; partials_ready = false; epart_ready = true;
    }
    void run_update() {
        materialise_value();
        if (!noise_ready || !value_ready) fail(MPPI_E_STATE, "update needs a rollout (or uploaded V and eps) first");
        materialise_eps();
        launch_update(stream, 0, NCH);
        launch_merge(NCH);
        merge_skipped = false;  // (an earlier fused tick may have left its tuples unmerged: these are merged)
        partials_ready = true;
    }
    // shard_stride: elements between consecutive shards' [A][T][8] blocks in `gathered` (0: packed)
    void run_finalize(const double* gathered, int G, int flags, mppi::P2PWait wait = mppi::P2PWait{}, size_t shard_stride = 0) {
        const int A_ = cfg.n_agents, T_ = cfg.horizon;
        mppi::TupleLayout lay{(unsigned)(shard_stride ? shard_stride : (size_t)A_ * T_ * mppi::kTupleW), (unsigned)(T_ * mppi::kTupleW),
                              (unsigned)mppi::kTupleW};
        if (!gathered) {
            if (!partials_ready) fail(MPPI_E_STATE, "no partials: call mppi_tick_begin first");
            gathered = d_merged; G = 1;
            if (merge_skipped) {  // the chunk / scan-block tuples, merged by the finalize kernel itself
                gathered = d_part; G = direct_n;
                lay = mppi::TupleLayout{(unsigned)mppi::kTupleW, (unsigned)(T_ * direct_n * mppi::kTupleW), (unsigned)(direct_n * mppi::kTupleW)};
            }
        } else if (merge_skipped && !wait.flags) {
            fail(MPPI_E_STATE, "this tick's partials were not merged (fused mppi_tick, or an engine connected to the p2p "
                               "exchange): nothing for a caller-side exchange to gather");
        }
        if (G < 1) fail(MPPI_E_INVALID, "n_shards must be >= 1");
        Scope sc(this, MPPI_KERNEL_FINALIZE);
        const int T = cfg.horizon;
        size_t lds = (size_t)4 * T * sizeof(double);
        if (lds + (size_t)(4 * (T - 1) + 4) * sizeof(double) + 1024 <= 64 * 1024) {  // the filter's basis fits next to the control rows: stage it
            lds += (size_t)(4 * (T - 1) + 4) * sizeof(double);
            flags |= 8;
        }
        // 16 lanes per row for the tuple merge, one wave per filter coefficient (16 of them): T = 50 -> 1024 threads; at least 256
        int fin_threads = std::min(1024, std::max(256, ((2 * T * std::max(1, 1024 / (2 * T)) + 63) / 64) * 64));
        // co-scheduled engines (shards, or the two halves of an agent split): a 1024-thread workgroup needs four free waves on EVERY SIMD
        // of a CU at once and waits for the other engine's rollout waves to drain; 512 threads start in the gaps (config 5 on its two
        // engines 134.5 -> 129.5 us per tick; one engine alone prefers 1024: 145.8 against 147.1, profiles/r5_ab_fin_threads.jsonl)
        if ((co_active() || is_co_sub) && fin_threads > 512) fin_threads = 512;
        // the next tick's nominal table on the way out (lane-per-sample ticks that run the plant step and the shift; a graph replay
        // keeps its prologue: its launches are frozen)
        if ((flags & 3) == 3 && !(flags & 4) && hoist_on() && inline_nominal() && small_nb == 0 && !capturing) flags |= 32;
        uint32_t tick_set = 0;
        if ((flags & 1) && !(flags & 4) && last_tick_eager) { flags |= 16; tick_set = last_tick_id + 1u; }
        // eager ticks also drop their outputs into the pinned host buffer (a graph replay cannot: its sequence number
        // would be frozen at capture time -- it keeps the D2H copy)
        const bool host_out = (flags & 1) && !(flags & 4) && !capturing;
        const bool ext = host_out && out_view_ext != nullptr;   // a sub of an agent split: the outputs land in the handle's pinned rows
        if (host_out && !ext) out_seq += 1u;
        if (slot_unclaimed >= 0) {
            if (host_out && !ext) { slot_seq[slot_unclaimed] = out_seq; slot_seq_valid[slot_unclaimed] = true; slot_unclaimed = -1; }
            else release_unclaimed_slot();
        }
        hipLaunchKernelGGL(mppi::finalize_kernel, dim3(cfg.n_agents), dim3(fin_threads), lds,
                           stream, P, gathered, G, lay, d_S, d_unom, d_ufilt, d_state, d_out, d_tick, flags, tick_set,
                           ext ? out_view_ext : (host_out ? d_out_view : nullptr), ext ? seq_view_ext : d_seq_view, ext ? seq_ext : out_seq, wait,
                           lanes_fresh_goal ? (const double*)(d_prev + (size_t)cfg.n_agents * 2 * cfg.horizon + (size_t)cfg.n_agents * 3) : (const double*)d_goal,
                           tcb[tab ^ 1], baseb[tab ^ 1], pkb[tab ^ 1],
                           lanes_fresh_state ? (const double*)(d_prev + (size_t)cfg.n_agents * 2 * cfg.horizon) : (const double*)nullptr, d_goal);
        lanes_fresh_state = lanes_fresh_goal = false;
        if (flags & 1) out_via_host = host_out;
        HIPCHK(hipGetLastError());
        partials_ready = false;
        table_valid = (flags & 32) != 0;   // (this launch rewrote the nominal controls: a table it did not refresh is stale)
        table_taken = false;
    }

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

    void init(const mppi_config& c) {
        cfg = c;
        // the one environment variable the library reads (a deployment knob: the default deadline of the blocking waits);
        // every measurement / test switch is an mppi_set_option key
        if (const char* v = std::getenv("MPPI_SYNC_TIMEOUT_MS")) sync_timeout_ms = std::atoi(v);
        if (cfg.n_agents < 1 || cfg.samples < 1) fail(MPPI_E_INVALID, "n_agents and samples must be >= 1");
        if (cfg.n_agents > 65535) fail(MPPI_E_INVALID, "n_agents %d: agents are a grid dimension (<= 65535)", cfg.n_agents);
        if (cfg.horizon < 5)
            fail(MPPI_E_INVALID, "horizon=%d: the Savitzky-Golay window horizon-1 must be > 3 "
                 "(scipy.signal.savgol_filter(u, horizon - 1, 3) at control/src/mppi:202)", cfg.horizon);
        if ((size_t)cfg.horizon * 40 + 128 > 64 * 1024)   // + the rollout kernel's few static LDS words
            fail(MPPI_E_INVALID, "horizon %d: the per-step table (40 B/step) must fit 64 KB of LDS (horizon <= 1634)", cfg.horizon);
        if (cfg.storage != MPPI_STORE_F32 && cfg.storage != MPPI_STORE_F64) fail(MPPI_E_INVALID, "bad storage %d", cfg.storage);
        // one row of dP / eps is addressed through a 32-bit buffer descriptor and 32-bit lane offsets
        if ((size_t)cfg.samples * (cfg.storage == MPPI_STORE_F64 ? 8 : 4) >= ((size_t)1 << 31))
            fail(MPPI_E_INVALID, "samples %d: a row of %d-byte elements must stay below 2 GiB", cfg.samples,
                 cfg.storage == MPPI_STORE_F64 ? 8 : 4);
        // the per-wave eps sums [A][T][2][NWp] (NWp: Ks / 64 rounded up to 32) are written through ONE 32-bit buffer descriptor (2 GiB of records)
        if ((size_t)cfg.n_agents * cfg.horizon * 2 * ((((size_t)cfg.samples + 63) / 64 + 31) / 32 * 32) * (cfg.storage == MPPI_STORE_F64 ? 8 : 4) >= ((size_t)1 << 31))
            fail(MPPI_E_INVALID, "n_agents * horizon * samples = %d * %d * %d: the per-wave noise sums must stay below 2 GiB", cfg.n_agents,
                 cfg.horizon, cfg.samples);
        if (cfg.model != MPPI_MODEL_DIFFDRIVE_RK4 && cfg.model != MPPI_MODEL_UNICYCLE_EULER)
            fail(MPPI_E_INVALID, "unknown model %d (rk4 + dd_dynamics = 0, euler + unicycle_dynamics = 1)", cfg.model);
        if (cfg.co_shards < 0 || cfg.co_shards > 8) fail(MPPI_E_INVALID, "co_shards must be 0 (auto), 1 (off) or 2..8");
        if (cfg.tick_path != MPPI_TICK_AUTO && cfg.tick_path != MPPI_TICK_LANES && cfg.tick_path != MPPI_TICK_SCAN)
            fail(MPPI_E_INVALID, "bad tick_path %d", cfg.tick_path);
        if (!(cfg.lambda > 0.0) || !(cfg.sigma >= 0.0)) fail(MPPI_E_INVALID, "lambda must be > 0 and sigma >= 0");
        if (!(cfg.dt > 0.0)) cfg.dt = 1.0 / (double)cfg.horizon;  // control/src/mppi:67
        int ndev = 0;
        HIPCHK(hipGetDeviceCount(&ndev));
        if (ndev < 1) fail(MPPI_E_HIP, "no HIP device visible: libmppi_hip has no CPU fallback");
        if (cfg.device < 0 || cfg.device >= ndev) fail(MPPI_E_INVALID, "device %d out of range (%d visible)", cfg.device, ndev);
        device = cfg.device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
        {
            int khz = 0;
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) wall_clock_khz = khz;
        }
        stream = own_stream;

        const int A = cfg.n_agents, K = cfg.samples, T = cfg.horizon;
        P.A = A; P.K = K; P.T = T; P.Ks = (K + 63) / 64 * 64;
        P.NWp = (P.Ks / 64 + 31) / 32 * 32;   // rows of the per-wave eps sums: whole 128-byte lines (in either storage type)
        if (alias_parent) { P.Ks = alias_parent->P.Ks; P.NWp = alias_parent->P.NWp; }   // (a co-scheduled shard: its rows are columns of the handle's own)
        hoist_auto = T <= 64 && (long)A * K >= 786432;
        P.sample_offset = cfg.sample_offset;
        if (cfg.agent_offset < 0) fail(MPPI_E_INVALID, "agent_offset must be >= 0");
        if (cfg.samples_total != 0 && (cfg.samples_total < (int64_t)cfg.sample_offset + cfg.samples || cfg.samples_total > 0xFFFFFFFFll))
            fail(MPPI_E_INVALID, "samples_total = %lld: 0 (this handle is the whole controller) or >= sample_offset + samples = %lld (global sample ids are 32-bit)",
                 (long long)cfg.samples_total, (long long)cfg.sample_offset + cfg.samples);
        P.agent_offset = (uint32_t)cfg.agent_offset;
        P.dt = cfg.dt;
        P.u_max = cfg.u_max;
        P.kth = cfg.wheel_radius / cfg.wheel_base;
        P.rhalf = cfg.wheel_radius / 2.0;
        P.floor_w = cfg.floor_w;
        refresh_weights();
        refresh_params();

        roll_bs = 256;
        roll_blocks = (K + roll_bs - 1) / roll_bs;
        P.model = cfg.model;
        P.grid = nullptr; P.grid_w = 0; P.grid_h = 0; P.grid_res = 1.0; P.grid_ox = 0.0; P.grid_oy = 0.0; P.grid_weight = 0.0;
        // largest rotation of the heading vector in one step: h/2 <= kth*dt*u_max (rk4), dt*u_max (euler)
        const double phi_max = (cfg.model == MPPI_MODEL_UNICYCLE_EULER ? 1.0 : P.kth) * P.dt * P.u_max;
        nterm = phi_max <= 0.03 ? 4 : (phi_max <= 0.25 ? 7 : 0);

        // update geometry: each block keeps one chunk of a row in registers
        pick_update_shape();
        const size_t Ks = (size_t)P.Ks;
        if (alias_parent) {
            // A co-scheduled K-shard fills COLUMNS [alias_k0, alias_k0 + K) of the handle's own rows (same row stride; the cut is a
            // multiple of the update kernel's chunk): ONE layout in memory whatever the number of engines that fill it -- the same
            // DRAM pages as the one-engine tick -- and nothing to allocate.
            // (An agent-split shard: the handle's arrays from agent alias_a0 on -- whole rows, every array line-aligned per agent.)
            // The per-wave eps sums too: their rows are padded to whole lines (P.NWp) and the cut is a multiple of 2048 samples, so the
            // shard's slots of a row start on a line of their own (co_check_regions verifies every array before the group is used).
            const size_t es = esz(), k0 = (size_t)alias_k0, a0 = (size_t)alias_a0;
            d_eps = static_cast<char*>(alias_parent->d_eps) + (a0 * T * 2 * Ks + k0) * es;
            d_dP = static_cast<char*>(alias_parent->d_dP) + (a0 * T * Ks + k0) * es;
            d_stot = static_cast<char*>(alias_parent->d_stot) + (a0 * Ks + k0) * es;
            d_epart = static_cast<char*>(alias_parent->d_epart) + (a0 * T * 2 * (size_t)P.NWp + k0 / 64) * es;
        } else {
            void* p = nullptr;
            size_t bytes = (size_t)A * T * 2 * Ks * esz();
            HIPCHK(hipMalloc(&p, bytes)); hbm_bytes += bytes; d_eps = p;
            bytes = (size_t)A * T * Ks * esz();
            HIPCHK(hipMalloc(&p, bytes)); hbm_bytes += bytes; d_dP = p;
            bytes = (size_t)A * Ks * esz();
            HIPCHK(hipMalloc(&p, bytes)); hbm_bytes += bytes; d_stot = p;
            bytes = (size_t)A * T * 2 * (size_t)P.NWp * esz();
            HIPCHK(hipMalloc(&p, bytes)); hbm_bytes += bytes; d_epart = p;
            HIPCHK(hipMemsetAsync(d_epart, 0, bytes, stream));
        }
        for (int i = 0; i < 2; ++i) {
            tcb[i] = dev_alloc<double>((size_t)A * T * mppi::kTcW, hbm_bytes);
            baseb[i] = dev_alloc<double>((size_t)A * T, hbm_bytes);
            pkb[i] = dev_alloc<mppi::PkRow>((size_t)A * T, hbm_bytes);
        }
        use_table_set(0);
        d_unom = dev_alloc<double>((size_t)A * 2 * T, hbm_bytes);
        d_ufilt = dev_alloc<double>((size_t)A * 2 * T, hbm_bytes);
        d_state = dev_alloc<double>((size_t)A * 3, hbm_bytes);
        d_goal = dev_alloc<double>((size_t)A * 3, hbm_bytes);
        {   // small-K path: lanes = timesteps, one wave (T <= 64) or one block (T <= 256) per sample
            // AUTO: measured ticks, scan vs lane kernels: K = 1000 16.1 vs 27.0 us, 4000 20.0 vs 28.4, 10000 26.8 vs 30.6, 16000 31.5 vs 31.3 (T = 50, a wave per sample);
            // T = 100 (a block per sample): K = 500 19.0 vs 39.8, 2000 26.1 vs 40.4, 5000 36.9 vs 41.3, 10000 54.8 vs 43.4.
            // Round 2 (back to back | blocking call, which only the scan path serves zero-copy), T = 50: K = 8000 24.7 vs 28.4 | 48 vs 61,
            // 12000 27.5 vs 29.7 | 51 vs 62, 16000 31.8 vs 30.3 | 56 vs 62, 24000 36.4 vs 31.5 | 61 vs 67; T = 100: 4000 34.9 vs 38.1 | 59 vs 75,
            // 6000 40.5 vs 39.8 | 65 vs 72, 10000 55.5 vs 44.0 | 79 vs 76  ->  16384 / 6144
            // Round 5 (the lane rollout's rows stored write-through, its fresh inputs zero-copy as well: profiles/r5_ab_tick_path_small_k.jsonl,
            // r5_blocking_tick_scan_vs_lanes.txt), T = 50: 12000 26.2 vs 26.8 | 34.7 vs 38.2, 14000 27.9 vs 26.8 | 36.5 vs 37.8, 16000 30.5 vs 27.1 |
            // 39.1 vs 38.4; T = 100: 4000 32.0 vs 35.0 | 40.3 vs 43.2, 5000 35.0 vs 35.2 | 43.2 vs 43.3, 6000 38.1 vs 35.5 | 46.1 vs 43.8  ->  14336 / 5120
            const bool applies = T <= 256 && cfg.model == MPPI_MODEL_DIFFDRIVE_RK4;
            const long k_rule = cfg.samples_total > 0 ? (long)cfg.samples_total : (long)K;   // (a shard decides by the whole controller's size)
            const bool want = cfg.tick_path == MPPI_TICK_SCAN || (cfg.tick_path == MPPI_TICK_AUTO && (long)A * k_rule <= (T <= 64 ? 14336 : 5120));
            if (applies && want) {
                small_nw = T <= 64 ? 1 : 4;
                // units the chip keeps resident at once (152 VGPRs: 3 waves per SIMD x 1024 SIMDs): the kernel is
                // latency-bound (46 % VALU-busy at K = 10^4), so a second round of waves would double its time
                // (forcing the fp32 variant into 128 VGPRs for a 4th wave per SIMD -- 4096 units, 3 instead of 4 samples per
                // wave at K = 10^4 -- changed nothing: 16.9 vs 17.2 us; each wave just runs slower)
                const long unit_cap = small_nw == 1 ? 3072 : 768;
                small_spw = (int)std::max(1L, ((long)A * K + unit_cap - 1) / unit_cap);
                const int units = (K + small_spw - 1) / small_spw;
                small_nb = small_nw == 1 ? (units + 3) / 4 : units;
            }
        }
        {
            const int ch8 = f64() ? mppi::UpdCfg<double, 8>::CH : mppi::UpdCfg<float, 8>::CH;
            d_part = dev_alloc<double>((size_t)A * T * std::max((K + ch8 - 1) / ch8, small_nb) * mppi::kTupleW, hbm_bytes);
        }
        d_prev = dev_alloc<double>((size_t)A * (2 * T + 6), hbm_bytes);
        d_merged = dev_alloc<double>((size_t)A * T * mppi::kTupleW, hbm_bytes);
        d_S = dev_alloc<double>((size_t)4 * (T - 1) + 4, hbm_bytes);   // the Savitzky-Golay operator's orthonormal basis [4][T-1] (+ its four values at the even window's half-integer position)
        d_out = dev_alloc<double>((size_t)A * 8, hbm_bytes);
        d_tick = dev_alloc<uint32_t>(1, hbm_bytes);
        d_clk = dev_alloc<unsigned long long>(2 + mppi::kProbeMarks, hbm_bytes);
        HIPCHK(hipMemsetAsync(d_clk, 0, (2 + mppi::kProbeMarks) * sizeof(unsigned long long), stream));
        P.clk = d_clk;
        d_fill = dev_alloc<double>((size_t)A * 2, hbm_bytes);
        HIPCHK(hipMemsetAsync(d_fill, 0, (size_t)A * 2 * sizeof(double), stream));
        P.shift_fill = d_fill;
        HIPCHK(hipMemsetAsync(d_unom, 0, (size_t)A * 2 * T * sizeof(double), stream));  // uvec_init, :65
        HIPCHK(hipMemsetAsync(d_ufilt, 0, (size_t)A * 2 * T * sizeof(double), stream));
        HIPCHK(hipMemsetAsync(d_out, 0, (size_t)A * 8 * sizeof(double), stream));
        HIPCHK(hipMemsetAsync(d_tick, 0, sizeof(uint32_t), stream));
        if (!alias_parent) {
            HIPCHK(hipMemsetAsync(d_eps, 0, (size_t)A * T * 2 * Ks * esz(), stream));
            HIPCHK(hipMemsetAsync(d_dP, 0, (size_t)A * T * Ks * esz(), stream));
            HIPCHK(hipMemsetAsync(d_stot, 0, (size_t)A * Ks * esz(), stream));
        }

        std::vector<double> S;
        if (!mppi::savgol_basis(T, S)) fail(MPPI_E_INVALID, "cannot build the Savitzky-Golay operator for horizon %d", T);
        HIPCHK(hipMemcpyAsync(d_S, S.data(), S.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        wait_stream("engine initialisation");
        HIPCHK(hipEventCreateWithFlags(&ev_partials, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_foreign, hipEventDisableTiming));

        // pinned + mapped + coherent: the device reads inputs from / writes outputs to these buffers directly
        const unsigned pin = hipHostMallocMapped | hipHostMallocCoherent;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_stage), (size_t)kRing * A * 6 * sizeof(double), pin));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_stage_view), h_stage, 0));
        for (int i = 0; i < kRing; ++i) HIPCHK(hipEventCreateWithFlags(&ring_ev[i], hipEventDisableTiming));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_out), (size_t)A * 8 * sizeof(double), pin));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_out_view), h_out, 0));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_seq), (size_t)A * sizeof(uint32_t), pin));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_seq_view), h_seq, 0));
        std::memset(h_out, 0, (size_t)A * 8 * sizeof(double));
        std::memset(h_seq, 0, (size_t)A * sizeof(uint32_t));
    }

    void destroy_graph() {
        if (graph_exec) { hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        if (graph) { hipGraphDestroy(graph); graph = nullptr; }
    }

    ~mppi_engine() {
        int prev = -1;
        const bool back = hipGetDevice(&prev) == hipSuccess && prev != device;
        hipSetDevice(device);
        struct Restore { bool on; int dev; ~Restore() { if (on) (void)hipSetDevice(dev); } } restore{back, prev};
        try { wait_stream("engine teardown"); } catch (...) {}  // a dead device must not hang the destructor either
        for (auto* e : subs) delete e;
        subs.clear();
        if (ev_co) hipEventDestroy(ev_co);
        p2p_release();
        if (ev_partials) hipEventDestroy(ev_partials);
        if (ev_foreign) hipEventDestroy(ev_foreign);
        destroy_graph();
        for (auto& p : pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
        for (auto e : ev_pool) hipEventDestroy(e);
        for (int i = 0; i < kRing; ++i) if (ring_ev[i]) hipEventDestroy(ring_ev[i]);
        if (h_stage) hipHostFree(h_stage);
        if (h_out) hipHostFree(h_out);
        if (h_seq) hipHostFree(h_seq);
        if (alias_parent) d_eps = d_dP = d_stot = d_epart = nullptr;   // (the handle's)
        void* bufs[] = {d_eps, d_dP, d_stot, d_epart, tcb[0], tcb[1], baseb[0], baseb[1], pkb[0], pkb[1], d_unom, d_ufilt, d_state, d_goal, d_part, d_merged, d_S, d_out, d_tick, d_tmp, d_grid, d_prev, d_clk, d_fill};
        for (void* b : bufs) if (b) hipFree(b);
        if (own_stream) hipStreamDestroy(own_stream);
    }
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

// How many engines a fused device-noise tick of this handle runs on (1: unsplit).  wanted: asked for by name (co_shards >= 2).
int mppi_engine::co_plan(bool& wanted, bool* by_agents) const {
    int G = cfg.co_shards;
    const bool lanes = small_nb == 0;
    wanted = G > 1;
    if (by_agents) *by_agents = false;
    // AUTO, several agents: the AGENTS are split -- two engines of A / 2 agents, nothing exchanged (agents are independent controllers).
    // It beats the split by samples wherever both apply, and applies where that one does not pay (config 5: every shard's publish
    // would walk all A * T rows; asked for by name it measured 0.297 against 0.154 ms).  Same box, one process, tick us, one engine |
    // split by samples | by agents: 2 x 500 000 147.7 | 135.2 | 128.9; 4 x 250 000 147.9 | 146.2 | 129.4; 8 x 131 072 148.8 | -- | 132.1;
    // 64 x 16 384 150.8 | -- | 134.5 (38 + 26 agents 139.3, 40 + 24 140.1, three engines 22 + 21 + 21 132.4).
    // Each half must still be a size the mixed-precision rollout is chosen for (shards choose it by size: 400 000 sample-agents).
    if (G == 0 && lanes && !f64() && cfg.n_agents >= 2 && (long)(cfg.n_agents / 2) * cfg.samples >= 400000 && hbm_bytes < ((size_t)48 << 30)) {
        if (by_agents) *by_agents = true;
        return 2;
    }
    // AUTO: two shards where the pair measured faster than the one engine (config 4: +7-9 % rollouts/s; nothing below
    // ~5e5 samples), on the lane-per-sample path only
    // (and while a second set of buffers is small change against the 288 GB: the subs hold another half of this engine's)
    // and while the shards' publish kernels (one block per peer walking all A * T rows, 16 per pass) stay small change: measured
    // on one box, tick us one engine / two shards: A = 1 T = 100 K = 1e6 300 / 287, T = 25 99.4 / 96.5; A = 2 x 500 000 148 / 137;
    // A = 4 x 250 000 149 / 145; A = 8 x 131 072 (400 rows) 150 / 159 -- no longer a gain
    // ... and in fp32 storage only: the all-fp64 mode's two big kernels are both bound by HBM traffic (400 MB written, 400 MB read),
    // there is nothing complementary to overlap -- measured 0.277 ms split against 0.250 ms unsplit (profiles/r4_bench_c4_f64_*.json)
    if (G == 0) G = (lanes && !f64() && (long)cfg.n_agents * cfg.samples >= 500000 && cfg.samples >= 4 * CH && cfg.n_agents * cfg.horizon <= 256 &&
                     hbm_bytes < ((size_t)48 << 30)) ? 2 : 1;
    if (G <= 1) return 1;
    if (!lanes || cfg.samples < G * CH) {
        if (wanted) fail(MPPI_E_INVALID, "co_shards = %d needs the lane-per-sample tick path and at least %d samples per shard", G, CH);
        return 1;
    }
    return G;
}
// shard boundaries [0, c1, ..., K] on multiples of the update kernel's chunk (no shard ends in a ragged chunk)
void mppi_engine::co_cuts(int G, std::vector<int>& cuts) const {
    cuts.assign(G + 1, 0);
    for (int g = 1; g < G; ++g) cuts[g] = (int)(((long)g * cfg.samples / G + CH / 2) / CH) * CH;
    // Two shards: 58 / 42.  Shard 0's launches go first, so the tick ends with shard 1's update + merge + finalize with nothing
    // left to hide them under; a smaller shard 1 shortens that tail as long as its rollout still covers shard 0's update
    // (same box, K = 10^6, tick us: 50/50 139.0, 55/45 136.4, 58/42 134.2, 60/40 135.4, 62/38 136, 65/35 137.1; 45/55 141.6)
    if (G == 2) cuts[1] = std::max(CH, (int)(((long)cfg.samples * co_cut_pct / 100 + CH / 2) / CH) * CH);
    cuts[G] = cfg.samples;
    for (int g = 0; g < G; ++g) if (cuts[g + 1] <= cuts[g]) fail(MPPI_E_INVALID, "co_shards = %d: %d samples do not split", G, cfg.samples);
}

// A co-scheduled engine fills REGIONS of this handle's own big arrays while this engine's kernels fill and read the rest, the two
// streams unordered: no 128-byte line may hold words of both regions (kernels of two streams writing and reading words of one line
// through the eight XCDs' separate L2s is not something this layout leans on), and no write of one engine may land in the other's
// region at all.  Verified here for every array, for whatever (K, A, cut) the group was built with -- a violated invariant refuses
// the group (AUTO: the one engine serves every call) instead of computing on.
//   K split at column k0:  rows of dP / eps / Stot: pitch Ks * es, the shard's columns from k0 * es;  eps sums: pitch NWp * es, from (k0 / 64) * es
//   agent split at a0:     every array from agent a0: per-agent sizes T Ks es, 2 T Ks es, Ks es, 2 T NWp es
void mppi_engine::co_check_regions(const mppi_engine* sub) const {
    constexpr size_t kLine = 128;
    const size_t es = esz(), T_ = (size_t)cfg.horizon, Ks = (size_t)P.Ks, NWp = (size_t)P.NWp;
    auto on_line = [&](const void* p, const char* what) {
        if (reinterpret_cast<uintptr_t>(p) % kLine != 0) fail(MPPI_E_INTERNAL, "co-scheduled shard: its region of %s does not start on a %zu-byte line", what, kLine);
    };
    auto pitch_ok = [&](size_t bytes, const char* what) {
        if (bytes % kLine != 0) fail(MPPI_E_INTERNAL, "co-scheduled shard: %s (%zu bytes) is not a whole number of %zu-byte lines", what, bytes, kLine);
    };
    if (sub->alias_parent != this || sub->P.Ks != P.Ks || sub->P.NWp != P.NWp) fail(MPPI_E_INTERNAL, "co-scheduled shard: not a region of this handle's arrays");
    on_line(d_dP, "dP (base)"); on_line(d_eps, "eps (base)"); on_line(d_stot, "Stot (base)"); on_line(d_epart, "the eps sums (base)");
    on_line(sub->d_dP, "dP"); on_line(sub->d_eps, "eps"); on_line(sub->d_stot, "Stot"); on_line(sub->d_epart, "the eps sums");
    pitch_ok(Ks * es, "a row of dP / eps / Stot"); pitch_ok(NWp * es, "a row of the eps sums");
    if (sub->alias_k0 > 0) {
        // the shard's slots of an eps-sum row are [k0 / 64, k0 / 64 + ceil(K_sub / 64)): inside the row, behind shard 0's
        if (sub->alias_k0 % 64 != 0 || (size_t)sub->alias_k0 / 64 + ((size_t)sub->cfg.samples + 63) / 64 > NWp || (size_t)sub->alias_k0 + (size_t)sub->cfg.samples > Ks)
            fail(MPPI_E_INTERNAL, "co-scheduled shard: columns [%d, %d) do not fit the handle's rows", sub->alias_k0, sub->alias_k0 + sub->cfg.samples);
    } else {
        pitch_ok(T_ * Ks * es, "an agent's dP"); pitch_ok(Ks * es, "an agent's Stot"); pitch_ok(T_ * 2 * NWp * es, "an agent's eps sums");
        if (sub->alias_a0 < 1 || sub->alias_a0 + sub->cfg.n_agents > cfg.n_agents) fail(MPPI_E_INTERNAL, "co-scheduled shard: agents out of range");
    }
}

// Builds the shards.  Asked for by name (co_shards >= 2): at mppi_create, errors reported there.  AUTO: with the first fused
// device-noise mppi_tick (co_pending) -- a handle that only ever runs the caller's own exchange (mppi_tick_begin / _finish: the
// ranks of an N > 1 run), graph replays or injected-noise ticks never pays for the second set of buffers.
void mppi_engine::co_build() {
    co_pending = false;
    bool wanted = false, by_agents = false;
    const int G = co_plan(wanted, &by_agents);
    if (G <= 1) return;
    if (by_agents) {   // (AUTO only)
        try {
            mppi_config c = cfg;
            co_a0 = (cfg.n_agents + 1) / 2;
            c.n_agents = cfg.n_agents - co_a0;
            c.agent_offset = cfg.agent_offset + (uint32_t)co_a0;   // the noise streams are keyed by the global agent index
            c.co_shards = 1;
            c.tick_path = MPPI_TICK_LANES;
            mppi_engine* e = new mppi_engine();
            subs.push_back(e);
            e->is_co_sub = true;
            // the second engine's agents are agents [co_a0, A) of the handle's own big arrays (whole rows, every array's per-agent size
            // a multiple of a line) -- its V is where every other call of the ABI looks for it, nothing to pull
            e->alias_parent = this; e->alias_a0 = co_a0;
            e->init(c);
            co_check_regions(e);
            if (sig_is_matrix) { for (int i = 0; i < 4; ++i) e->sig_cost[i] = sig_cost[i]; e->sig_is_matrix = true; e->refresh_params(); }
            e->P.grid = P.grid; e->P.grid_w = P.grid_w; e->P.grid_h = P.grid_h; e->P.grid_res = P.grid_res; e->P.grid_ox = P.grid_ox;
            e->P.grid_oy = P.grid_oy; e->P.grid_weight = P.grid_weight;
            e->sync_timeout_ms = sync_timeout_ms;
            co_hand_switches(e);
            for (int i = 0; i < 7; ++i) e->w_off[i] = w_off[i];
            e->refresh_weights();
            e->out_view_ext = d_out_view + (size_t)co_a0 * 8;
            e->seq_view_ext = d_seq_view + co_a0;
            if (!ev_co) HIPCHK(hipEventCreateWithFlags(&ev_co, hipEventDisableTiming));
            co_agents = true; co_synced = false; co_dirty = false; co_value_dirty = false;
        } catch (const EngineError& er) {
            co_release();
            co_fallback = "co_shards AUTO (agents) fell back to one engine: " + er.msg;
        } catch (...) {
            co_release();
            co_fallback = "co_shards AUTO (agents) fell back to one engine (allocation failed)";
        }
        return;
    }
    try {
        std::vector<int> cuts;
        co_cuts(G, cuts);
        co_k0 = cuts[1];
        for (int g = 1; g < G; ++g) {
            mppi_config c = cfg;
            c.samples = cuts[g + 1] - cuts[g];
            c.sample_offset = cfg.sample_offset + (uint32_t)cuts[g];
            c.samples_total = 0;   // (a co-scheduled shard takes shard 0's kernel: force_pk)
            c.co_shards = 1;
            c.tick_path = MPPI_TICK_LANES;
            mppi_engine* e = new mppi_engine();
            subs.push_back(e);
            e->is_co_sub = true;
            e->alias_parent = this; e->alias_k0 = cuts[g];
            e->init(c);
            co_check_regions(e);
            // what the handle was told since its creation (the shards may be built long after): the cost's sig matrix, the obstacle grid
            // (shared: same device; a later mppi_set_obstacle_grid reaches the shards first and gives them their own copy), the shift
            // fill, the deadline and the measurement switches.  Nominal controls / state / goal follow with the first tick (co_sync_subs).
            if (sig_is_matrix) { for (int i = 0; i < 4; ++i) e->sig_cost[i] = sig_cost[i]; e->sig_is_matrix = true; e->refresh_params(); }
            e->P.grid = P.grid; e->P.grid_w = P.grid_w; e->P.grid_h = P.grid_h; e->P.grid_res = P.grid_res; e->P.grid_ox = P.grid_ox;
            e->P.grid_oy = P.grid_oy; e->P.grid_weight = P.grid_weight;
            HIPCHK(hipMemcpyAsync(e->d_fill, d_fill, (size_t)cfg.n_agents * 2 * sizeof(double), hipMemcpyDeviceToDevice, stream));
            wait_stream("co-scheduled shard set-up");
            co_hand_switches(e);
            for (int i = 0; i < 7; ++i) e->w_off[i] = w_off[i];
            e->refresh_weights();
        }
        std::vector<void*> ptrs(G, nullptr);
        std::vector<mppi_engine*> all{this};
        all.insert(all.end(), subs.begin(), subs.end());
        for (int g = 0; g < G; ++g) {
            if (mppi_p2p_create(all[g], G, g, nullptr)) fail(MPPI_E_HIP, "co-scheduled shard %d: %s", g, all[g]->err.c_str());
            ptrs[g] = all[g]->p2p_mbox;
        }
        for (int g = 0; g < G; ++g)
            if (mppi_p2p_connect(all[g], nullptr, ptrs.data())) fail(MPPI_E_HIP, "co-scheduled shard %d: %s", g, all[g]->err.c_str());
        p2p_internal = true;
        if (!ev_co) HIPCHK(hipEventCreateWithFlags(&ev_co, hipEventDisableTiming));
        co_synced = false;
    } catch (const EngineError& er) {
        co_release();
        if (wanted) throw;   // asked for by name: report; AUTO: the one engine serves every call anyway -- and says why (mppi_co_note)
        co_fallback = "co_shards AUTO fell back to one engine: " + er.msg;
    } catch (...) {
        co_release();
        if (wanted) throw;
        co_fallback = "co_shards AUTO fell back to one engine (allocation failed)";
    }
}

// what this engine's arrays hold for the sub's agents -> the sub (only after something other than a split tick touched them)
void mppi_engine::co_push_agents() {
    if (co_synced) return;
    mppi_engine* e = subs[0];
    const size_t A1 = e->cfg.n_agents, T_ = cfg.horizon, a0 = (size_t)co_a0;
    HIPCHK(hipEventRecord(ev_co, stream));
    HIPCHK(hipStreamWaitEvent(e->stream, ev_co, 0));
    HIPCHK(hipMemcpyAsync(e->d_unom, d_unom + a0 * 2 * T_, A1 * 2 * T_ * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_state, d_state + a0 * 3, A1 * 3 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_goal, d_goal + a0 * 3, A1 * 3 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_fill, d_fill + a0 * 2, A1 * 2 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    e->have_state = have_state; e->have_goal = have_goal;
    e->invalidate_table();
    co_synced = true;
}

// The sub's results of the last split tick(s) -> this engine's arrays, in two parts (the V set of config 5 is ~100 MB
// device-to-device; a caller that reads its nominal controls between ticks must not pay for it):
//   co_pull        nominal and filtered controls, state, outputs -- a few KB, by every call but the split tick itself and the read-only
//                  queries.  The sub still holds the same values afterwards: co_synced stays as it is (the calls that CHANGE this
//                  engine's controls / state clear it themselves, and the next split tick hands those over).
//   co_pull_value  the tick's V (cost prefix, totals, nominal cost-to-go, per-step table, per-wave eps sums) -- only by what reads V:
//                  mppi_download_value, mppi_update, and whatever settles the lazy state before a parameter change.
// The noise is never copied: it is a function of (seed, tick, GLOBAL agent, sample, t) and re-drawn here on demand.
void mppi_engine::co_pull() {
    if (!co_dirty) return;
    co_dirty = false;
    invalidate_table();   // (the sub's agents' controls / poses arrive in this engine's arrays: its own table knows nothing of them)
    mppi_engine* e = subs[0];
    const size_t A1 = e->cfg.n_agents, T_ = cfg.horizon, a0 = (size_t)co_a0;
    HIPCHK(hipEventRecord(ev_co, e->stream));
    HIPCHK(hipStreamWaitEvent(stream, ev_co, 0));
    auto pull = [&](void* dst, const void* src, size_t per_agent_bytes) {
        HIPCHK(hipMemcpyAsync(static_cast<char*>(dst) + a0 * per_agent_bytes, src, A1 * per_agent_bytes, hipMemcpyDeviceToDevice, stream));
    };
    pull(d_unom, e->d_unom, 2 * T_ * sizeof(double));
    pull(d_ufilt, e->d_ufilt, 2 * T_ * sizeof(double));
    pull(d_state, e->d_state, 3 * sizeof(double));
    pull(d_out, e->d_out, 8 * sizeof(double));
    // (the sub's stream must not run ahead of these copies: its next launches come after co_push_agents' event)
    out_via_host = false;   // d_out is whole; the pinned rows are too, but a later non-split finalize rewrites only d_out's sequence
    wait_stream("co-scheduled agents: results pulled");
}
void mppi_engine::co_pull_value() {
    co_pull();
    if (!co_value_dirty) return;
    co_value_dirty = false;
    mppi_engine* e = subs[0];
    const size_t A1 = e->cfg.n_agents, T_ = cfg.horizon, a0 = (size_t)co_a0;
    HIPCHK(hipEventRecord(ev_co, e->stream));
    HIPCHK(hipStreamWaitEvent(stream, ev_co, 0));
    auto pull = [&](void* dst, const void* src, size_t per_agent_bytes) {
        HIPCHK(hipMemcpyAsync(static_cast<char*>(dst) + a0 * per_agent_bytes, src, A1 * per_agent_bytes, hipMemcpyDeviceToDevice, stream));
    };
    // (the second engine's cost prefix, totals, eps sums and stored noise already are where this handle keeps them: its big arrays
    // are agents [co_a0, A) of this engine's own; what it computed into arrays of its own are the two small per-step tables)
    pull(d_base, e->d_base, T_ * sizeof(double));
    pull(d_tc, e->d_tc, T_ * mppi::kTcW * sizeof(double));
    wait_stream("co-scheduled agents: V pulled");
}

// the fused device-noise tick of a handle whose agents are split over two engines
void mppi_engine::co_tick_agents(const double* state, const double* goal, uint64_t seed, uint32_t tick) {
    mppi_engine* e = subs[0];
    co_push_agents();
    set_inputs(state, goal);
    co_fence_subs();   // (whatever else this handle was asked to do since the last split tick ran on this engine's stream, over all agents' rows)
    e->set_inputs(state ? state + (size_t)3 * co_a0 : nullptr, goal ? goal + (size_t)3 * co_a0 : nullptr);
    {
        AgentView view(this);
        run_nominal();
        run_pipeline(MPPI_NOISE_PHILOX, seed, tick, nullptr, /*skip_small_merge=*/true);
        run_finalize(nullptr, 1, 1 | 2);
    }
    e->seq_ext = out_seq;   // the one sequence number mppi_get_outputs waits for, on every agent's row
    e->run_nominal();
    e->run_pipeline(MPPI_NOISE_PHILOX, seed, tick, nullptr, /*skip_small_merge=*/true);
    e->run_finalize(nullptr, 1, 1 | 2);
    co_dirty = true; co_value_dirty = true;
    co_subs_inflight = true;
}

void mppi_engine::co_tick(const double* state, const double* goal, uint64_t seed, uint32_t tick) {
    if (co_agents) { co_tick_agents(state, goal, seed, tick); return; }
    co_sync_subs();
    set_inputs(state, goal);
    co_fence_subs();   // (whatever else this handle was asked to do since the last split tick ran on this engine's stream, over all columns)
    for (auto* e : subs) e->set_inputs(state, goal);
    bool shards_pk;
    {
        ShardView view(this);
        run_nominal();
        run_pipeline(MPPI_NOISE_PHILOX, seed, tick, nullptr, /*skip_small_merge=*/true);   // the publish kernel merges a handful of tuples itself
        shards_pk = last_rollout_pk;
    }
    for (auto* e : subs) {   // every shard takes shard 0's kernel (sizes differ by a chunk at most): one kernel's arithmetic for every column of V
        e->force_pk = shards_pk ? 1 : 0;
        e->run_nominal();
        e->run_pipeline(MPPI_NOISE_PHILOX, seed, tick, nullptr, /*skip_small_merge=*/true);
    }
    // one thread drives all engines: every publish is enqueued before any finalize that waits for it
    p2p_wait = p2p_publish(merge_skipped ? nullptr : d_merged);
    for (auto* e : subs) e->p2p_wait = e->p2p_publish(e->merge_skipped ? nullptr : e->d_merged);
    const int par = (int)(p2p_epoch & 1u);
    run_finalize(p2p_data(p2p_mbox, par, 0), p2p_n, 1 | 2, p2p_wait, p2p_slot / sizeof(double));
    for (auto* e : subs) e->run_finalize(e->p2p_data(e->p2p_mbox, par, 0), e->p2p_n, 1 | 2, e->p2p_wait, e->p2p_slot / sizeof(double));
    // The tick's V is complete in this handle's own arrays: the shards' cost prefixes, totals and eps sums are columns of this engine's
    // rows, base / tc are shard 0's (every shard derives the same table bit for bit).  mppi_download_value / mppi_update read them IN
    // PLACE -- the bytes the shards' update kernels consumed -- once this engine's stream has waited for the shards' (co_join_subs, made
    // by whatever is called next).  The noise is a function of (seed, tick, global sample): re-drawn on demand as after any tick.
    noise_ready = true; value_ready = true; value_lazy = false; epart_ready = true;
    eps_lazy = !store_eps_always; injected_ready = store_eps_always;   // (option store_eps: the shards stored their columns of it)
    lazy_seed = seed; lazy_tick = tick; lazy_from_counter = false; lazy_counter_bumped = false;
    co_subs_inflight = true;
}

extern "C" {

int mppi_abi_version(void) { return MPPI_ABI_VERSION; }

int mppi_default_config(mppi_config* cfg) {
    if (!cfg) return MPPI_E_INVALID;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(*cfg);
    cfg->n_agents = 1;
    cfg->samples = 10;   // control/src/mppi:62
    cfg->horizon = 100;  // control/src/mppi:62
    cfg->storage = MPPI_STORE_F32;
    cfg->device = 0;
    cfg->sample_offset = 0;
    cfg->model = MPPI_MODEL_DIFFDRIVE_RK4;  // MPPI(model=rk4), control/src/mppi:62
    cfg->tick_path = MPPI_TICK_AUTO;
    cfg->co_shards = 0;   // auto
    cfg->dt = 0.0;
    cfg->sigma = 0.9;     // control/src/mppi:88
    cfg->lambda = 0.001;  // control/src/mppi:89
    cfg->q[0] = 1e3; cfg->q[1] = 1e3; cfg->q[2] = 0.0;       // :69
    cfg->r[0] = 1.0; cfg->r[1] = 1.0;                        // :71
    cfg->p1[0] = 1e3; cfg->p1[1] = 1e3; cfg->p1[2] = 1e3;    // :73
    cfg->u_max = 6.35492;       // :18
    cfg->wheel_radius = 0.033;  // :19
    cfg->wheel_base = 0.16;     // :20
    cfg->floor_w = 1e-8;        // :193
    cfg->samples_total = 0;     // this handle is the whole controller
    return MPPI_OK;
}

const char* mppi_last_error(const mppi_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mppi_create(const mppi_config* cfg, mppi_engine** out) {
    if (!cfg || !out) { g_create_error = "mppi_create: NULL argument"; return MPPI_E_INVALID; }
    *out = nullptr;
    mppi_engine* e = nullptr;
    int prev_dev = -1;
    const bool have_prev = hipGetDevice(&prev_dev) == hipSuccess;
    struct Restore { bool on; int dev; ~Restore() { if (on) (void)hipSetDevice(dev); } } restore{have_prev, prev_dev};
    try {
        // the caller's struct may be an older, shorter one: its bytes over this library's defaults (fields are only ever appended)
        static_assert(sizeof(mppi_config) >= MPPI_CONFIG_SIZE_V5, "mppi_config shrank");
        if (cfg->struct_size < MPPI_CONFIG_SIZE_V5 || cfg->struct_size > sizeof(mppi_config))
            fail(MPPI_E_INVALID, "mppi_config.struct_size = %u: this library knows %u ... %zu bytes (start from mppi_default_config; "
                                 "a caller compiled against a NEWER header than the library it loads?)", cfg->struct_size, MPPI_CONFIG_SIZE_V5, sizeof(mppi_config));
        mppi_config full;
        mppi_default_config(&full);
        std::memcpy(&full, cfg, cfg->struct_size);
        full.struct_size = (uint32_t)sizeof(full);
        e = new mppi_engine();
        e->init(full);
        if (full.co_shards == 0) { bool w; e->co_pending = e->co_plan(w) > 1; }   // AUTO: built with the first fused device-noise tick
        else e->co_build();
        *out = e;
        return MPPI_OK;
    } catch (const EngineError& er) { g_create_error = er.msg; delete e; return er.code; }
    catch (const std::exception& ex) { g_create_error = ex.what(); delete e; return MPPI_E_INTERNAL; }
    catch (...) { g_create_error = "unknown error"; delete e; return MPPI_E_INTERNAL; }
}

int mppi_destroy(mppi_engine* h) {
    if (!h) return MPPI_E_INVALID;
    delete h;  // the destructor restores the caller's current device
    return MPPI_OK;
}

int mppi_set_stream(mppi_engine* h, void* hip_stream) {
    API_BEGIN(h)
    h->drain_timing();
    h->wait_stream(__func__);
    h->destroy_graph();
    h->stream = static_cast<hipStream_t>(hip_stream);
    API_END(h)
}

int mppi_get_stream(mppi_engine* h, void** hip_stream) {
    API_BEGIN_FAST(h)
    if (!hip_stream) fail(MPPI_E_INVALID, "hip_stream is NULL");
    *hip_stream = static_cast<void*>(h->stream);
    API_END(h)
}

int mppi_set_sigma_lambda(mppi_engine* h, double sigma, double lambda) {
    API_BEGIN(h)
    h->invalidate_table();
    for (auto* sub__ : h->subs) if (int rc__ = mppi_set_sigma_lambda(sub__, sigma, lambda)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    if (!(lambda > 0.0) || !(sigma >= 0.0)) fail(MPPI_E_INVALID, "lambda must be > 0 and sigma >= 0");
    h->settle_lazy_state();
    h->cfg.sigma = sigma; h->cfg.lambda = lambda;
    h->sig_is_matrix = false;
    h->refresh_params();
    h->destroy_graph();
    API_END(h)
}

int mppi_set_sig_matrix(mppi_engine* h, const double* sig, double lambda) {
    API_BEGIN(h)
    h->invalidate_table();
    if (!sig) fail(MPPI_E_INVALID, "sig is NULL");
    for (auto* sub__ : h->subs) if (int rc__ = mppi_set_sig_matrix(sub__, sig, lambda)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    if (!(lambda > 0.0) || !(sig[0] >= 0.0)) fail(MPPI_E_INVALID, "lambda must be > 0 and sig[0][0] >= 0");
    for (int i = 0; i < 4; ++i) if (!std::isfinite(sig[i])) fail(MPPI_E_INVALID, "sig[%d] is not finite", i);
    h->settle_lazy_state();
    h->cfg.sigma = sig[0]; h->cfg.lambda = lambda;  // the noise of BOTH wheels is drawn with sig[0,0] (control/src/mppi:145)
    for (int i = 0; i < 4; ++i) h->sig_cost[i] = sig[i];
    h->sig_is_matrix = true;
    h->refresh_params();
    h->destroy_graph();
    API_END(h)
}

int mppi_set_weights(mppi_engine* h, const double* q, const double* r, const double* p1) {
    API_BEGIN(h)
    for (auto* sub__ : h->subs) if (int rc__ = mppi_set_weights(sub__, q, r, p1)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    for (int i = 0; i < 3; ++i) if ((q && !std::isfinite(q[i])) || (p1 && !std::isfinite(p1[i]))) fail(MPPI_E_INVALID, "cost weights must be finite");
    for (int i = 0; i < 2; ++i) if (r && !std::isfinite(r[i])) fail(MPPI_E_INVALID, "cost weights must be finite");
    h->settle_lazy_state();   // the last tick's V may exist only as "re-run with these weights"
    // (a matrix given by its diagonal IS diagonal: whatever mppi_set_weight_matrices left off it goes)
    if (q) { for (int i = 0; i < 3; ++i) h->cfg.q[i] = q[i]; h->w_off[0] = h->w_off[1] = h->w_off[2] = 0.0; }
    if (r) { for (int i = 0; i < 2; ++i) h->cfg.r[i] = r[i]; h->w_off[3] = 0.0; }
    if (p1) { for (int i = 0; i < 3; ++i) h->cfg.p1[i] = p1[i]; h->w_off[4] = h->w_off[5] = h->w_off[6] = 0.0; }
    h->refresh_weights();
    h->invalidate_table();
    h->destroy_graph();
    API_END(h)
}

int mppi_set_weight_matrices(mppi_engine* h, const double* Q, const double* R, const double* P1) {
    API_BEGIN(h)
    for (auto* sub__ : h->subs) if (int rc__ = mppi_set_weight_matrices(sub__, Q, R, P1)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    for (int i = 0; i < 9; ++i) if ((Q && !std::isfinite(Q[i])) || (P1 && !std::isfinite(P1[i]))) fail(MPPI_E_INVALID, "cost weights must be finite");
    for (int i = 0; i < 4; ++i) if (R && !std::isfinite(R[i])) fail(MPPI_E_INVALID, "cost weights must be finite");
    h->settle_lazy_state();   // the last tick's V may exist only as "re-run with these weights"
    // x' M x sees the symmetric part of M only: diagonal as given, off-diagonal (M[i][j] + M[j][i]) / 2
    if (Q) {
        for (int i = 0; i < 3; ++i) h->cfg.q[i] = Q[4 * i];
        h->w_off[0] = 0.5 * (Q[1] + Q[3]); h->w_off[1] = 0.5 * (Q[2] + Q[6]); h->w_off[2] = 0.5 * (Q[5] + Q[7]);
    }
    if (R) { h->cfg.r[0] = R[0]; h->cfg.r[1] = R[3]; h->w_off[3] = 0.5 * (R[1] + R[2]); }
    if (P1) {
        for (int i = 0; i < 3; ++i) h->cfg.p1[i] = P1[4 * i];
        h->w_off[4] = 0.5 * (P1[1] + P1[3]); h->w_off[5] = 0.5 * (P1[2] + P1[6]); h->w_off[6] = 0.5 * (P1[5] + P1[7]);
    }
    h->refresh_weights();
    h->invalidate_table();
    h->destroy_graph();
    API_END(h)
}

int mppi_set_sync_timeout(mppi_engine* h, int milliseconds) {
    API_BEGIN_FAST(h)
    for (auto* sub__ : h->subs) if (int rc__ = mppi_set_sync_timeout(sub__, milliseconds)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    if (milliseconds < 0) fail(MPPI_E_INVALID, "timeout must be >= 0 (0 = wait forever)");
    h->sync_timeout_ms = milliseconds;
    API_END(h)
}

int mppi_set_tick_counter(mppi_engine* h, uint32_t next_tick_id) {
    API_BEGIN(h)
    h->settle_lazy_state();   // a graph replay's lazily re-drawn noise / V are addressed through this counter: materialise them first
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_tick), (int)next_tick_id, 1, h->stream));
    h->last_tick_eager = false;  // the counter now holds what the caller put there
    API_END(h)
}

int mppi_stream_wait_partials(mppi_engine* h, void* other_stream) {
    API_BEGIN(h)
    HIPCHK(hipEventRecord(h->ev_partials, h->stream));
    HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(other_stream), h->ev_partials, 0));
    API_END(h)
}

int mppi_wait_for_stream(mppi_engine* h, void* other_stream) {
    API_BEGIN(h)
    HIPCHK(hipEventRecord(h->ev_foreign, static_cast<hipStream_t>(other_stream)));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_foreign, 0));
    API_END(h)
}

int mppi_set_obstacle_grid(mppi_engine* h, const int8_t* cells, int32_t width, int32_t height, double resolution,
                           double origin_x, double origin_y, double weight) {
    API_BEGIN(h)
    for (auto* sub__ : h->subs) if (int rc__ = mppi_set_obstacle_grid(sub__, cells, width, height, resolution, origin_x, origin_y, weight)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    h->settle_lazy_state();
    h->wait_stream(__func__);
    h->destroy_graph();
    if (!cells || weight == 0.0) {
        h->P.grid = nullptr; h->P.grid_weight = 0.0;
    } else {
        if (width < 1 || height < 1 || !(resolution > 0.0)) fail(MPPI_E_INVALID, "bad grid geometry %d x %d @ %g", width, height, resolution);
        const size_t bytes = (size_t)width * height;
        if (bytes > h->grid_bytes) {
            if (h->d_grid) { HIPCHK(hipFree(h->d_grid)); h->hbm_bytes -= h->grid_bytes; h->d_grid = nullptr; h->grid_bytes = 0; }
            h->d_grid = dev_alloc<signed char>(bytes, h->hbm_bytes);
            h->grid_bytes = bytes;
        }
        HIPCHK(hipMemcpy(h->d_grid, cells, bytes, hipMemcpyHostToDevice));
        h->P.grid = h->d_grid; h->P.grid_w = width; h->P.grid_h = height;
        h->P.grid_res = resolution; h->P.grid_ox = origin_x; h->P.grid_oy = origin_y; h->P.grid_weight = weight;
    }
    API_END(h)
}

int mppi_reset(mppi_engine* h, int agent) {
    API_BEGIN(h)
    h->invalidate_table();
    h->co_synced = h->co_synced && !h->co_agents;   // (an agent split takes this engine's arrays over with its next tick)
    if (!h->co_agents) for (auto* sub__ : h->subs) if (int rc__ = mppi_reset(sub__, agent)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    const size_t row = (size_t)2 * h->cfg.horizon * sizeof(double);
    if (agent < 0) HIPCHK(hipMemsetAsync(h->d_unom, 0, row * h->cfg.n_agents, h->stream));
    else if (agent < h->cfg.n_agents) HIPCHK(hipMemsetAsync(h->d_unom + (size_t)agent * 2 * h->cfg.horizon, 0, row, h->stream));
    else fail(MPPI_E_INVALID, "agent %d out of range", agent);
    API_END(h)
}

int mppi_set_shift_fill(mppi_engine* h, int agent, const double* fill) {
    API_BEGIN(h)
    h->invalidate_table();
    h->co_synced = h->co_synced && !h->co_agents;   // (an agent split takes this engine's arrays over with its next tick)
    if (!h->co_agents) for (auto* sub__ : h->subs) if (int rc__ = mppi_set_shift_fill(sub__, agent, fill)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    if (!fill || agent < 0 || agent >= h->cfg.n_agents) fail(MPPI_E_INVALID, "bad agent/fill");
    HIPCHK(hipMemcpyAsync(h->d_fill + (size_t)agent * 2, fill, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->wait_stream(__func__);
    API_END(h)
}

int mppi_set_nominal(mppi_engine* h, int agent, const double* uvec) {
    API_BEGIN(h)
    h->invalidate_table();
    h->co_synced = h->co_synced && !h->co_agents;   // (an agent split takes this engine's arrays over with its next tick)
    if (!h->co_agents) for (auto* sub__ : h->subs) if (int rc__ = mppi_set_nominal(sub__, agent, uvec)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    if (!uvec || agent < 0 || agent >= h->cfg.n_agents) fail(MPPI_E_INVALID, "bad agent/uvec");
    const size_t n = (size_t)2 * h->cfg.horizon;
    HIPCHK(hipMemcpyAsync(h->d_unom + agent * n, uvec, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->wait_stream(__func__);
    API_END(h)
}

int mppi_get_nominal(mppi_engine* h, int agent, double* uvec) {
    API_BEGIN(h)
    if (!uvec || agent < 0 || agent >= h->cfg.n_agents) fail(MPPI_E_INVALID, "bad agent/uvec");
    const size_t n = (size_t)2 * h->cfg.horizon;
    HIPCHK(hipMemcpyAsync(uvec, h->d_unom + agent * n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    h->wait_stream(__func__);
    API_END(h)
}

int mppi_upload_noise(mppi_engine* h, const double* eps) {
    API_BEGIN(h)
    if (!eps) fail(MPPI_E_INVALID, "eps is NULL");
    const int A = h->cfg.n_agents, T = h->cfg.horizon, K = h->cfg.samples;
    const size_t n = (size_t)A * T * 2 * K;
    h->ensure_tmp(n);
    HIPCHK(hipMemcpyAsync(h->d_tmp, eps, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    dim3 grid(std::min((K + 255) / 256, 1024), A * T * 2);
    if (h->f64()) hipLaunchKernelGGL(mppi::pack_rows_kernel<double>, grid, dim3(256), 0, h->stream, h->d_tmp, static_cast<double*>(h->d_eps), K, h->P.Ks, (const double*)nullptr, 1);
    else hipLaunchKernelGGL(mppi::pack_rows_kernel<float>, grid, dim3(256), 0, h->stream, h->d_tmp, static_cast<float*>(h->d_eps), K, h->P.Ks, (const double*)nullptr, 1);
    HIPCHK(hipGetLastError());
    h->wait_stream(__func__);
    h->noise_ready = true; h->injected_ready = true;
    h->epart_ready = false;
    h->eps_lazy = false;
    h->value_lazy = false;  // the snapshot no longer matches the resident noise
    API_END(h)
}

int mppi_download_noise(mppi_engine* h, double* eps) {
    API_BEGIN(h)
    if (!eps) fail(MPPI_E_INVALID, "eps is NULL");
    if (!h->noise_ready) fail(MPPI_E_STATE, "no noise resident");
    if (!h->eps_lazy) h->co_pull_value();   // (an agent split whose ticks STORED their noise, option store_eps: the second engine's rows are pulled with its V)
    h->materialise_eps();
    const int A = h->cfg.n_agents, T = h->cfg.horizon, K = h->cfg.samples;
    const size_t n = (size_t)A * T * 2 * K;
    h->ensure_tmp(n);
    dim3 grid(std::min((K + 255) / 256, 1024), A * T * 2);
    if (h->f64()) hipLaunchKernelGGL(mppi::unpack_rows_kernel<double>, grid, dim3(256), 0, h->stream, static_cast<const double*>(h->d_eps), h->d_tmp, K, h->P.Ks, (const double*)nullptr, 1);
    else hipLaunchKernelGGL(mppi::unpack_rows_kernel<float>, grid, dim3(256), 0, h->stream, static_cast<const float*>(h->d_eps), h->d_tmp, K, h->P.Ks, (const double*)nullptr, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(eps, h->d_tmp, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    h->wait_stream(__func__);
    API_END(h)
}

int mppi_rollout(mppi_engine* h, const double* state, const double* goal, int noise_mode, uint64_t seed, uint32_t tick_id) {
    API_BEGIN(h)
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    h->check_noise_mode(noise_mode, /*tick_path=*/false);   // refuse before the inputs are staged: nothing half-set on failure
    h->set_inputs(state, goal);
    h->run_nominal();
    h->run_rollout(noise_mode, seed, tick_id, nullptr);
    API_END(h)
}

int mppi_download_value(mppi_engine* h, double* V) {
    API_BEGIN(h)
    if (!V) fail(MPPI_E_INVALID, "V is NULL");
    h->co_pull_value();
    h->materialise_value();
    if (!h->value_ready) fail(MPPI_E_STATE, "no value function resident");
    const int A = h->cfg.n_agents, T = h->cfg.horizon, K = h->cfg.samples;
    const size_t n = (size_t)A * T * K;
    h->ensure_tmp(n);
    dim3 grid(std::min((K + 255) / 256, 1024), A * T);
    if (h->f64()) hipLaunchKernelGGL(mppi::value_unpack_kernel<double>, grid, dim3(256), 0, h->stream, static_cast<const double*>(h->d_dP), static_cast<const double*>(h->d_stot), (const double*)h->d_base, h->d_tmp, K, h->P.Ks, T);
    else hipLaunchKernelGGL(mppi::value_unpack_kernel<float>, grid, dim3(256), 0, h->stream, static_cast<const float*>(h->d_dP), static_cast<const float*>(h->d_stot), (const double*)h->d_base, h->d_tmp, K, h->P.Ks, T);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(V, h->d_tmp, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    h->wait_stream(__func__);
    API_END(h)
}

int mppi_upload_value(mppi_engine* h, const double* V) {
    API_BEGIN(h)
    if (!V) fail(MPPI_E_INVALID, "V is NULL");
    h->co_value_dirty = false;   // (every agent's V is replaced: nothing of the sub's is wanted any more)
    const int A = h->cfg.n_agents, T = h->cfg.horizon, K = h->cfg.samples;
    const size_t n = (size_t)A * T * K;
    h->ensure_tmp(n);
    HIPCHK(hipMemcpyAsync(h->d_tmp, V, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // baseline := per-row minimum, so the stored offsets are >= 0 and small near the minimum
    hipLaunchKernelGGL(mppi::row_min_kernel, dim3(A * T), dim3(256), 0, h->stream, h->d_tmp, K, h->d_base);
    HIPCHK(hipGetLastError());
    dim3 grid(std::min((K + 255) / 256, 1024), A * T);
    if (h->f64()) hipLaunchKernelGGL(mppi::value_pack_kernel<double>, grid, dim3(256), 0, h->stream, (const double*)h->d_tmp, (const double*)h->d_base, static_cast<double*>(h->d_dP), static_cast<double*>(h->d_stot), K, h->P.Ks, T);
    else hipLaunchKernelGGL(mppi::value_pack_kernel<float>, grid, dim3(256), 0, h->stream, (const double*)h->d_tmp, (const double*)h->d_base, static_cast<float*>(h->d_dP), static_cast<float*>(h->d_stot), K, h->P.Ks, T);
    HIPCHK(hipGetLastError());
    h->wait_stream(__func__);
    h->value_ready = true; h->value_lazy = false;
    API_END(h)
}

int mppi_update(mppi_engine* h, double* uvec_out) {
    API_BEGIN(h)
    h->co_pull_value();
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    h->run_update();
    h->run_finalize(nullptr, 1, 0);
    if (uvec_out) {
        const size_t n = (size_t)h->cfg.n_agents * 2 * h->cfg.horizon;
        HIPCHK(hipMemcpyAsync(uvec_out, h->d_ufilt, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        h->wait_stream(__func__);
    }
    API_END(h)
}

int mppi_plant_step(mppi_engine* h, const double* state, double* next_state) {
    API_BEGIN(h)
    h->invalidate_table();
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    const int A = h->cfg.n_agents;
    if (state) { h->stage_upload(state, h->d_state, (size_t)A * 3); h->have_state = true; }
    if (!h->have_state) fail(MPPI_E_STATE, "no state resident");
    hipLaunchKernelGGL(mppi::plant_kernel, dim3((A + 63) / 64), dim3(64), 0, h->stream, h->P, h->d_state, h->d_unom, h->d_out);
    HIPCHK(hipGetLastError());
    h->out_via_host = false;  // d_out now holds the plant step's result, not the last tick's
    if (next_state) {
        const double* o = h->h_out;
        if (h->out_seq) h->wait_stream("mppi_plant_step");  // a finalize still in flight may write h_out: let it land first
        HIPCHK(hipMemcpyAsync(h->h_out, h->d_out, (size_t)A * 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        h->wait_stream(__func__);
        for (int a = 0; a < A; ++a) for (int i = 0; i < 3; ++i) next_state[a * 3 + i] = o[(size_t)a * 8 + i];
    }
    API_END(h)
}

int mppi_shift(mppi_engine* h) {
    API_BEGIN(h)
    h->invalidate_table();
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    hipLaunchKernelGGL(mppi::shift_kernel, dim3(h->cfg.n_agents * 2), dim3(256), (size_t)h->cfg.horizon * sizeof(double), h->stream, h->P, h->d_unom);
    HIPCHK(hipGetLastError());
    API_END(h)
}

int mppi_tick_begin(mppi_engine* h, const double* state, const double* goal, int noise_mode, uint64_t seed, uint32_t tick_id) {
    API_BEGIN(h)
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    h->check_noise_mode(noise_mode);   // refuse before the inputs are staged: nothing half-set on failure
    h->set_inputs(state, goal, /*zero_copy=*/h->small_nb > 0);
    h->run_nominal();
    h->run_pipeline(noise_mode, seed, tick_id, nullptr);
    API_END(h)
}

int mppi_partials_ptr(mppi_engine* h, void** dev_ptr, size_t* bytes) {
    API_BEGIN(h)
    if (dev_ptr) *dev_ptr = h->d_merged;
    if (bytes) *bytes = (size_t)h->cfg.n_agents * h->cfg.horizon * mppi::kTupleW * sizeof(double);
    API_END(h)
}

int mppi_tick_finish(mppi_engine* h, const void* gathered_dev, int n_shards) {
    API_BEGIN(h)
    h->run_finalize(static_cast<const double*>(gathered_dev), n_shards, 1 | 2);
    API_END(h)
}

int mppi_p2p_create(mppi_engine* h, int n_ranks, int rank, void* ipc_handle_out) {
    API_BEGIN(h)
    h->co_pending = false;   // (a handle on a caller's cross-GPU exchange runs unsplit)
    if ((h->p2p_internal || h->co_agents) && h->co_active()) { h->wait_stream(__func__); for (auto* e__ : h->subs) e__->wait_stream(__func__); h->co_release(); }
    if (n_ranks < 1 || n_ranks > 8 || rank < 0 || rank >= n_ranks) fail(MPPI_E_INVALID, "p2p: 1 <= n_ranks <= 8, 0 <= rank < n_ranks");
    static_assert(sizeof(hipIpcMemHandle_t) <= MPPI_IPC_HANDLE_BYTES, "IPC handle does not fit the ABI's buffer");
    h->wait_stream(__func__);
    h->p2p_release();
    h->p2p_n = n_ranks; h->p2p_rank = rank;
    h->p2p_slot = (h->p2p_n_f64() * sizeof(double) + 255) / 256 * 256;
    h->p2p_bytes = (size_t)2 * n_ranks * h->p2p_slot + (size_t)2 * n_ranks * mppi::kFlagStride * sizeof(uint32_t);
    void* p = nullptr;
    // fine-grained: stores from a peer GPU become visible to a kernel that is already running here
    HIPCHK(hipExtMallocWithFlags(&p, h->p2p_bytes, hipDeviceMallocFinegrained));
    h->p2p_mbox = static_cast<char*>(p); h->hbm_bytes += h->p2p_bytes;
    HIPCHK(hipMemset(p, 0, h->p2p_bytes));
    h->p2p_epoch = 0;
    if (ipc_handle_out) {
        hipIpcMemHandle_t hd;
        HIPCHK(hipIpcGetMemHandle(&hd, p));
        std::memset(ipc_handle_out, 0, MPPI_IPC_HANDLE_BYTES);
        std::memcpy(ipc_handle_out, &hd, sizeof(hd));
    }
    API_END(h)
}

int mppi_p2p_connect(mppi_engine* h, const void* ipc_handles, void* const* local_ptrs) {
    API_BEGIN(h)
    if (!h->p2p_mbox || h->p2p_internal) fail(MPPI_E_STATE, "mppi_p2p_create first");   // (a co-scheduled group's mailboxes are wired before they are marked internal)
    if (!ipc_handles && !local_ptrs) fail(MPPI_E_INVALID, "p2p connect needs IPC handles or mailbox pointers");
    h->wait_stream(__func__);
    for (int g = 0; g < 8; ++g) {  // connecting again: unmap what an earlier connect opened
        if (h->p2p_peer[g] && h->p2p_peer_ipc[g]) hipIpcCloseMemHandle(h->p2p_peer[g]);
        h->p2p_peer[g] = nullptr; h->p2p_peer_ipc[g] = false;
    }
    h->p2p_connected = false;
    for (int g = 0; g < h->p2p_n; ++g) {
        if (g == h->p2p_rank) { h->p2p_peer[g] = h->p2p_mbox; continue; }
        if (local_ptrs && local_ptrs[g]) {   // an engine of THIS process -- possibly on another GPU of the node
            hipPointerAttribute_t at{};
            if (hipPointerGetAttributes(&at, local_ptrs[g]) != hipSuccess || at.type != hipMemoryTypeDevice) {
                (void)hipGetLastError();
                fail(MPPI_E_INVALID, "p2p connect: local_ptrs[%d] is not a device pointer (pass mppi_p2p_mailbox_ptr of the peer engine)", g);
            }
            if (at.device != h->device) {
                int can = 0;
                HIPCHK(hipDeviceCanAccessPeer(&can, h->device, at.device));
                if (!can) fail(MPPI_E_INVALID, "p2p connect: device %d cannot access device %d (rank %d's mailbox): no peer path between the two", h->device, at.device, g);
                const hipError_t pe = hipDeviceEnablePeerAccess(at.device, 0);   // (the engine's device is current: DeviceGuard)
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) fail(MPPI_E_HIP, "hipDeviceEnablePeerAccess(%d) from device %d: %s", at.device, h->device, hipGetErrorString(pe));
                (void)hipGetLastError();
            }
            h->p2p_peer[g] = static_cast<char*>(local_ptrs[g]);
            continue;
        }
        if (!ipc_handles) fail(MPPI_E_INVALID, "p2p connect: no handle for rank %d", g);
        hipIpcMemHandle_t hd;
        std::memcpy(&hd, static_cast<const char*>(ipc_handles) + (size_t)g * MPPI_IPC_HANDLE_BYTES, sizeof(hd));
        void* p = nullptr;
        HIPCHK(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
        h->p2p_peer[g] = static_cast<char*>(p); h->p2p_peer_ipc[g] = true;
    }
    h->p2p_connected = true;
    API_END(h)
}

// mppi_p2p_create + exchange of the IPC handles through files + mppi_p2p_connect, for ranks that are separate PROCESSES of one
// node and have no process group to carry the handles (a plain C++ / ROS node needs no torch for this): rank r writes its
// handle to "<prefix>.<r>" (written under a temporary name and renamed, so a reader never sees half a file), waits until all
// n_ranks files exist, connects.
int mppi_p2p_rendezvous(mppi_engine* h, const char* prefix, int n_ranks, int rank, int timeout_ms) {
    if (!h) return MPPI_E_INVALID;
    if (!prefix || !*prefix) { h->err = "p2p rendezvous: empty path prefix"; return MPPI_E_INVALID; }
    unsigned char mine[MPPI_IPC_HANDLE_BYTES];
    // this rank's file of an EARLIER run goes first: a fast peer must not find it while this rank is still creating its mailbox
    if (rank >= 0) (void)std::remove((std::string(prefix) + "." + std::to_string(rank)).c_str());
    if (int rc = mppi_p2p_create(h, n_ranks, rank, mine)) return rc;
    API_BEGIN(h)
    const std::string base(prefix);
    auto name = [&](int r) { return base + "." + std::to_string(r); };
    // file = {magic, n_ranks, rank, bytes of one mailbox, writer's pid} + the handle.  A reader refuses a file of another SHAPE (not
    // this group's) and keeps waiting over a file whose writer is no longer alive (a stale file of an earlier run of the same
    // shape -- the normal relaunch case: its handle would name a dead process's memory)
    struct Head { char magic[8]; int32_t n_ranks, rank; uint64_t mbox_bytes; int64_t pid; };
    auto head_of = [&](int r) { Head hd{}; std::memcpy(hd.magic, "MPPIMBX2", 8); hd.n_ranks = n_ranks; hd.rank = r; hd.mbox_bytes = h->p2p_bytes; hd.pid = (int64_t)getpid(); return hd; };
    {
        const std::string tmp = name(rank) + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) fail(MPPI_E_INVALID, "p2p rendezvous: cannot write %s", tmp.c_str());
        const Head hd = head_of(rank);
        const size_t w = std::fwrite(&hd, 1, sizeof(hd), f) + std::fwrite(mine, 1, sizeof(mine), f);
        if (std::fclose(f) != 0 || w != sizeof(hd) + sizeof(mine) || std::rename(tmp.c_str(), name(rank).c_str()) != 0)
            fail(MPPI_E_INVALID, "p2p rendezvous: cannot publish %s", name(rank).c_str());
    }
    std::vector<unsigned char> all((size_t)n_ranks * MPPI_IPC_HANDLE_BYTES, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < n_ranks; ++r) {
        if (r == rank) { std::memcpy(all.data() + (size_t)r * MPPI_IPC_HANDLE_BYTES, mine, sizeof(mine)); continue; }
        for (;;) {
            FILE* f = std::fopen(name(r).c_str(), "rb");
            if (f) {
                Head hd{};
                const size_t got = std::fread(&hd, 1, sizeof(hd), f) + std::fread(all.data() + (size_t)r * MPPI_IPC_HANDLE_BYTES, 1, MPPI_IPC_HANDLE_BYTES, f);
                std::fclose(f);
                const bool writer_alive = hd.pid > 0 && (kill((pid_t)hd.pid, 0) == 0 || errno == EPERM);
                if (got == sizeof(hd) + MPPI_IPC_HANDLE_BYTES && writer_alive) {
                    Head want = head_of(r);
                    want.pid = hd.pid;
                    if (std::memcmp(&hd, &want, sizeof(hd)) != 0)
                        fail(MPPI_E_INVALID, "p2p rendezvous: %s belongs to another group (ranks %d / rank %d / mailbox %llu bytes; this group: %d / %d / %llu): "
                             "a stale file of an earlier run, or engines of different shapes", name(r).c_str(), hd.n_ranks, hd.rank,
                             (unsigned long long)hd.mbox_bytes, n_ranks, r, (unsigned long long)h->p2p_bytes);
                    break;
                }
            }
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (timeout_ms > 0 && ms > timeout_ms) fail(MPPI_E_TIMEOUT, "p2p rendezvous: rank %d's handle (%s) did not appear within %d ms", r, name(r).c_str(), timeout_ms);
            struct timespec ts = {0, 2000000};
            nanosleep(&ts, nullptr);
        }
    }
    if (int rc = mppi_p2p_connect(h, all.data(), nullptr)) fail(rc, "%s", h->err.c_str());
    API_END(h)
}

int mppi_p2p_mailbox_ptr(mppi_engine* h, void** dev_ptr) {
    API_BEGIN(h)
    if (!dev_ptr) fail(MPPI_E_INVALID, "dev_ptr is NULL");
    *dev_ptr = h->p2p_internal ? nullptr : h->p2p_mbox;   // (the co-scheduled group's mailboxes are not the caller's)
    API_END(h)
}

int mppi_p2p_destroy(mppi_engine* h) {
    API_BEGIN(h)
    // a handle that never called mppi_p2p_create may still carry mailboxes: those of its co-scheduled group, which the caller
    // does not own -- leave them alone (before round 4 this freed shard 0's mailbox under the other shards' raw pointers)
    if (h->p2p_internal) return MPPI_OK;
    h->wait_stream(__func__);
    h->p2p_release();
    API_END(h)
}

// the caller's view of the exchange: connected by the caller's own mppi_p2p_create + mppi_p2p_connect (the co-scheduled group's
// internal mailboxes do not count -- publishing on them would desynchronise the group's epochs)
static void need_callers_exchange(mppi_engine* h) {
    if (h->p2p_internal || !h->p2p_connected)
        fail(MPPI_E_STATE, "p2p exchange is not connected (mppi_p2p_create + mppi_p2p_connect)");
}

int mppi_p2p_publish(mppi_engine* h) {
    API_BEGIN(h)
    need_callers_exchange(h);
    if (!h->partials_ready) fail(MPPI_E_STATE, "no partials: call mppi_tick_begin first");
    if (h->p2p_published) fail(MPPI_E_STATE, "this tick's partials were already published");
    h->p2p_wait = h->p2p_publish(h->merge_skipped ? nullptr : h->d_merged);
    h->p2p_published = true;
    API_END(h)
}

int mppi_tick_finish_p2p(mppi_engine* h) {
    API_BEGIN(h)
    need_callers_exchange(h);
    if (!h->p2p_published) fail(MPPI_E_STATE, "mppi_p2p_publish first");
    const int par = (int)(h->p2p_epoch & 1u);
    h->p2p_published = false;
    h->run_finalize(h->p2p_data(h->p2p_mbox, par, 0), h->p2p_n, 1 | 2, h->p2p_wait, h->p2p_slot / sizeof(double));  // mailbox slots are padded
    API_END(h)
}

int mppi_tick_exchange_p2p(mppi_engine* h) {
    const int rc = mppi_p2p_publish(h);
    return rc ? rc : mppi_tick_finish_p2p(h);
}

// Round trips of a known pattern through the mailboxes (no rollouts): every rank publishes, and a consumer kernel on
// every rank does exactly what the finalize kernel does -- polls this rank's flags from the device, acquires, reads the
// slots -- before the host compares what arrived with what every peer must have sent.  Collective: all ranks must call
// it with the same `rounds`.
int mppi_p2p_selftest(mppi_engine* h, int rounds) {
    API_BEGIN(h)
    need_callers_exchange(h);
    const size_t n = h->p2p_n_f64();
    const size_t slot_f64 = h->p2p_slot / sizeof(double);  // slots are padded to 256 bytes
    std::vector<double> pat(n), got((size_t)h->p2p_n * n);
    h->ensure_tmp(n + got.size() + 1);
    double* d_pat = h->d_tmp;
    double* d_got = h->d_tmp + n;
    int* d_status = reinterpret_cast<int*>(h->d_tmp + n + got.size());
    for (int r = 0; r < rounds; ++r) {
        const uint32_t e = h->p2p_epoch + 1u;
        for (size_t i = 0; i < n; ++i) pat[i] = 1e6 * (h->p2p_rank + 1) + 1e3 * e + (double)(i % 997);
        HIPCHK(hipMemcpyAsync(d_pat, pat.data(), n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        const mppi::P2PWait w = h->p2p_publish(d_pat);
        const int par = (int)(h->p2p_epoch & 1u);
        hipLaunchKernelGGL(mppi::p2p_check_kernel, dim3(1), dim3(256), 0, h->stream, w,
                           (const double*)h->p2p_data(h->p2p_mbox, par, 0), (int)n, (int)slot_f64, d_got, d_status);
        HIPCHK(hipGetLastError());
        int status = -1;
        HIPCHK(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(got.data(), d_got, got.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        h->wait_stream("p2p selftest");
        if (status != 0) fail(MPPI_E_TIMEOUT, "p2p selftest: round %d: a peer's flag did not reach the consumer kernel in time", r);
        for (int g = 0; g < h->p2p_n; ++g)
            for (size_t i = 0; i < n; ++i)
                if (got[(size_t)g * n + i] != 1e6 * (g + 1) + 1e3 * e + (double)(i % 997))
                    fail(MPPI_E_INTERNAL, "p2p selftest: round %d, slot %d, element %zu holds %.17g", r, g, i, got[(size_t)g * n + i]);
    }
    API_END(h)
}

int mppi_get_outputs(mppi_engine* h, double* next_state, double* u_applied) {
    API_BEGIN_FAST(h)
    const int A = h->cfg.n_agents;
    const double* o = h->h_out;
    if (h->out_via_host) {
        // the last finalize wrote its results into h_out itself and raised h_seq[a] behind them: wait for the words
        const uint32_t want = h->out_seq;
        const uint32_t* seqw = h->h_seq;
        h->bounded_wait([seqw, want, A] {
            for (int a = 0; a < A; ++a)
                if (__atomic_load_n(seqw + a, __ATOMIC_ACQUIRE) != want) return hipErrorNotReady;
            return hipSuccess;
        }, "mppi_get_outputs");
    } else {
        HIPCHK(hipMemcpyAsync(h->h_out, h->d_out, (size_t)A * 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        h->wait_stream(__func__);
    }
    for (int a = 0; a < A; ++a)
        if (o[(size_t)a * 8 + 7] != 0.0)
            fail(MPPI_E_TIMEOUT, "p2p exchange: a peer's tuples did not arrive within %d ms (the engine must be destroyed)", h->sync_timeout_ms);
    for (int a = 0; a < A; ++a) {
        if (next_state) for (int i = 0; i < 3; ++i) next_state[a * 3 + i] = o[(size_t)a * 8 + i];
        if (u_applied) for (int i = 0; i < 2; ++i) u_applied[a * 2 + i] = o[(size_t)a * 8 + 3 + i];
    }
    API_END(h)
}

// mppi_tick_begin with the knowledge that no exchange follows (the fused call)
static int tick_begin_fused(mppi_engine* h, const double* state, const double* goal, int noise_mode, uint64_t seed, uint32_t tick_id) {
    API_BEGIN(h)
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    h->check_noise_mode(noise_mode);   // refuse before the inputs are staged: nothing half-set on failure
    h->set_inputs(state, goal, /*zero_copy=*/h->small_nb > 0 || h->lanes_zero_copy_ok());
    h->run_nominal();
    h->run_pipeline(noise_mode, seed, tick_id, nullptr, /*skip_small_merge=*/true);
    API_END(h)
}

static int co_build_now(mppi_engine* h) {
    API_BEGIN_FAST(h)
    h->wait_stream("co-scheduled shard set-up");
    h->co_build();   // (AUTO: never throws -- on failure the one engine serves every call and mppi_co_note says why)
    API_END(h)
}

// the fused tick of a handle that carries co-scheduled shards (device noise; injected noise lives in this engine's own buffer)
static int tick_co(mppi_engine* h, const double* state, const double* goal, uint64_t seed, uint32_t tick_id) {
    API_BEGIN_FAST(h)
    h->check_noise_mode(MPPI_NOISE_PHILOX);   // a plain configuration refusal must not cost the handle its group (the catch below dissolves it)
    try {
        h->co_tick(state, goal, seed, tick_id);
    } catch (...) {
        // a throw between the shards' publishes / finalizes leaves their mailbox epochs and nominal controls out of step:
        // dissolve the group -- the one engine serves every later call (bounded waits: a dead device cannot hang this)
        bool lost = false;
        try {
            try { h->wait_stream("co-scheduled tick unwinding"); } catch (...) {}
            // an agent split: the sub may hold the only current controls / poses of its agents -- fetch them before it goes
            if (h->co_agents && h->co_dirty) {
                try { h->subs[0]->wait_stream("co-scheduled tick unwinding"); h->co_pull(); }
                catch (...) {
                    // they are gone: those agents start over from zero controls, and the caller must pass their poses again
                    lost = true;
                    const size_t T_ = h->cfg.horizon, a0 = (size_t)h->co_a0, A1 = (size_t)h->cfg.n_agents - a0;
                    (void)hipMemsetAsync(h->d_unom + a0 * 2 * T_, 0, A1 * 2 * T_ * sizeof(double), h->stream);
                    h->have_state = false;
                }
            }
            h->out_via_host = false;   // (a deleted sub will never raise its agents' sequence words)
            h->co_release();
        } catch (...) {}
        h->co_synced = false;
        h->invalidate_table();
        h->co_fallback = lost ? "a co-scheduled tick failed and the second engine's results could not be fetched: the group was dissolved, the agents it "
                                "carried were reset (zero nominal controls; pass every agent's state with the next call)"
                              : "a co-scheduled tick failed: the group was dissolved";
        throw;
    }
    API_END(h)
}

int mppi_tick(mppi_engine* h, const double* state, const double* goal, int noise_mode, uint64_t seed, uint32_t tick_id,
              double* next_state, double* u_applied) {
    int rc;
    if (h && h->co_pending && noise_mode == MPPI_NOISE_PHILOX) {
        rc = co_build_now(h);
        if (rc) return rc;
    }
    if (h && h->co_active() && noise_mode == MPPI_NOISE_PHILOX) {
        rc = tick_co(h, state, goal, seed, tick_id);
    } else {
        rc = tick_begin_fused(h, state, goal, noise_mode, seed, tick_id);
        if (rc) return rc;
        rc = mppi_tick_finish(h, nullptr, 1);
    }
    if (rc) return rc;
    if (next_state || u_applied) rc = mppi_get_outputs(h, next_state, u_applied);
    return rc;
}

int mppi_tick_graph(mppi_engine* h, uint64_t seed) {
    API_BEGIN(h)
    h->invalidate_table();
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    if (!h->have_state || !h->have_goal) fail(MPPI_E_STATE, "tick_graph needs a resident state and goal (run one mppi_tick first)");
    if (h->stream == nullptr) fail(MPPI_E_STATE, "graph capture is not possible on the null stream");
    h->check_noise_mode(MPPI_NOISE_PHILOX);
    if (h->graph_exec && h->graph_seed != seed) h->destroy_graph();
    if (!h->graph_exec) {
        const uint32_t saved = h->time_mask;
        h->time_mask = 0;
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        h->capturing = true;
        h->graph_tab = h->tab;   // the captured launches write THIS table set
        try {
            h->run_nominal();
            h->run_pipeline(MPPI_NOISE_PHILOX, seed, 0, h->d_tick, /*skip_small_merge=*/true);
            h->run_finalize(nullptr, 1, 1 | 2 | 4);
        } catch (...) {
            hipGraph_t g = nullptr;
            hipStreamEndCapture(h->stream, &g);
            if (g) hipGraphDestroy(g);
            h->time_mask = saved;
            h->capturing = false;
            throw;
        }
        h->capturing = false;
        HIPCHK(hipStreamEndCapture(h->stream, &h->graph));
        HIPCHK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
        h->graph_seed = seed;
        h->time_mask = saved;
    }
    h->use_table_set(h->graph_tab);   // (eager ticks in between may have switched sets: the replay's rollout rewrites the captured one)
    HIPCHK(hipGraphLaunch(h->graph_exec, h->stream));
    h->out_via_host = false;
    const bool small = h->small_nb > 0;   // (the scan kernel: V not resident, re-run from the snapshot on demand)
    h->noise_ready = true; h->value_ready = !small; h->value_lazy = small; h->partials_ready = false; h->epart_ready = !small;
    h->eps_lazy = small || !h->store_eps_always; h->lazy_seed = seed; h->lazy_from_counter = true; h->lazy_counter_bumped = true;
    h->injected_ready = !h->eps_lazy; h->last_tick_eager = false;
    API_END(h)
}

// Measurement / test switches (include/mppi_hip.h lists the keys); none changes results beyond rounding.
int mppi_set_option(mppi_engine* h, const char* key, int64_t value) {
    API_BEGIN(h)
    if (!key) fail(MPPI_E_INVALID, "option key is NULL");
    const std::string k(key);
    for (auto* sub__ : h->subs)
        if (k != "co_cut_pct" && k != "table_hoist") if (int rc__ = mppi_set_option(sub__, key, value)) fail(rc__, "co-scheduled shard: %s", sub__->err.c_str());
    if (k == "store_eps") { h->settle_lazy_state(); h->store_eps_always = value != 0; h->destroy_graph(); }
    else if (k == "rollout_pk") { h->settle_lazy_state(); h->use_pk = value != 0; h->destroy_graph(); }
    else if (k == "noise_packing") {
        if (value < 0 || value > 2) fail(MPPI_E_INVALID, "noise_packing: 0 (three steps per Philox call, the default stream), 1 (four) or 2 (hipRAND's normals: two)");
        if (value && (h->f64() || h->small_nb > 0 || !h->inline_nominal()))
            fail(MPPI_E_INVALID, "noise_packing 1 / 2 is drawn by the mixed-precision rollout only: fp32 storage, the lane kernels (tick_path lanes), rk4 / diff drive, T <= 256");
        h->settle_lazy_state(); h->wait_stream(__func__); h->noise_pack = (int)value; h->pick_update_shape(); h->partials_ready = false; h->destroy_graph();
    }
    else if (k == "table_hoist") {
        if (value < -1 || value > 1) fail(MPPI_E_INVALID, "table_hoist: -1 (by size), 0 or 1");
        if (!h->is_co_sub) { h->hoist_opt = (int)value; for (auto* e : h->subs) e->hoist_opt = h->hoist_on() ? 1 : 0; }
        h->invalidate_table();
    }
    else if (k == "lanes_zero_copy") h->lanes_zero_copy = value != 0;
    else if (k == "pk_min_samples") { h->settle_lazy_state(); h->pk_min_set = value >= 0; h->pk_min_samples = value >= 0 ? (long)value : 400000; h->destroy_graph(); }
    else if (k == "co_cut_pct") {
        if (value < 1 || value > 99) fail(MPPI_E_INVALID, "co_cut_pct: 1..99");
        if (h->is_co_sub) fail(MPPI_E_INVALID, "co_cut_pct is a property of the handle");
        h->co_cut_pct = (int)value;
        if (h->co_active() && h->p2p_internal) {   // re-cut the group
            h->wait_stream(__func__);
            for (auto* e : h->subs) e->wait_stream(__func__);
            const int G = 1 + (int)h->subs.size();
            h->co_release();
            const int asked = h->cfg.co_shards;
            h->cfg.co_shards = G;
            try { h->co_build(); } catch (...) { h->cfg.co_shards = asked; throw; }
            h->cfg.co_shards = asked;
            // (co_build hands the new shards this handle's switches)
        }
    }
    else fail(MPPI_E_INVALID, "unknown option '%s'", key);
    API_END(h)
}

int mppi_get_option(mppi_engine* h, const char* key, int64_t* value) {
    API_BEGIN_FAST(h)
    if (!key || !value) fail(MPPI_E_INVALID, "NULL argument");
    const std::string k(key);
    if (k == "store_eps") *value = h->store_eps_always;
    else if (k == "rollout_pk") *value = h->use_pk;
    else if (k == "noise_packing") *value = h->noise_pack;
    else if (k == "lanes_zero_copy") *value = h->lanes_zero_copy;
    else if (k == "table_hoist") *value = h->hoist_opt;
    else if (k == "pk_min_samples") *value = h->pk_min_set ? h->pk_min_samples : -1;
    else if (k == "co_cut_pct") *value = h->co_cut_pct;
    else fail(MPPI_E_INVALID, "unknown option '%s'", key);
    API_END(h)
}

int mppi_synchronize(mppi_engine* h) {
    API_BEGIN_FAST(h)
    h->wait_stream(__func__);
    for (auto* e : h->subs) e->wait_stream(__func__);
    API_END(h)
}

int mppi_savgol_matrix(int horizon, double* S) {
    if (!S || horizon < 1) return MPPI_E_INVALID;
    try {
        std::vector<double> v;
        if (!mppi::savgol_operator(horizon, v)) return MPPI_E_INVALID;
        std::memcpy(S, v.data(), v.size() * sizeof(double));
        return MPPI_OK;
    } catch (...) { return MPPI_E_INTERNAL; }
}

int mppi_kernel_timing(mppi_engine* h, uint32_t mask) {
    API_BEGIN_FAST(h)
    h->drain_timing();
    h->time_mask = mask;
    for (int i = 0; i < MPPI_KERNEL_COUNT; ++i) { h->t_ms[i] = 0.0; h->t_n[i] = 0; h->time_seen[i] = 0; }
    API_END(h)
}

int mppi_kernel_timing_period(mppi_engine* h, int period) {
    API_BEGIN_FAST(h)
    if (period < 1) fail(MPPI_E_INVALID, "period must be >= 1");
    h->time_period = period;
    API_END(h)
}

int mppi_kernel_times(mppi_engine* h, double* ms, int64_t* launches) {
    API_BEGIN_FAST(h)
    h->drain_timing();
    for (int i = 0; i < MPPI_KERNEL_COUNT; ++i) {
        if (ms) ms[i] = h->t_ms[i];
        if (launches) launches[i] = h->t_n[i];
    }
    API_END(h)
}

int mppi_shader_clock(mppi_engine* h, double* mhz) {
    API_BEGIN_FAST(h)
    if (!mhz) fail(MPPI_E_INVALID, "mhz is NULL");
    unsigned long long v[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(v, h->d_clk, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    h->wait_stream(__func__);
    // v[1] counts the constant-rate wall clock (hipDeviceAttributeWallClockRate, kHz)
    *mhz = v[1] ? (double)v[0] / (double)v[1] * (double)h->wall_clock_khz * 1e-3 : 0.0;
    API_END(h)
}

int mppi_probe_timeline(mppi_engine* h, uint64_t* cycles, uint64_t* total) {
    API_BEGIN_FAST(h)
    static_assert(MPPI_PROBE_MARKS == mppi::kProbeMarks, "header and kernels disagree on the number of stamps");
    if (!cycles) fail(MPPI_E_INVALID, "cycles is NULL");
    unsigned long long v[2 + mppi::kProbeMarks] = {};
    HIPCHK(hipMemcpyAsync(v, h->d_clk, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    h->wait_stream(__func__);
    for (int i = 0; i < mppi::kProbeMarks; ++i) cycles[i] = v[2 + i];
    if (total) *total = v[0];
    HIPCHK(hipMemsetAsync(h->d_clk + 2, 0, mppi::kProbeMarks * sizeof(unsigned long long), h->stream));
    API_END(h)
}

int mppi_co_info(mppi_engine* h, int32_t* n_shards, int32_t* samples) {
    API_BEGIN_FAST(h)
    int G = 1 + (int)h->subs.size();
    std::vector<int> cuts;
    bool by_agents = h->co_agents;
    if (h->co_pending) {   // the shards are built with the first fused device-noise tick: report what that tick will run on
        bool w;
        G = h->co_plan(w, &by_agents);
        if (G > 1 && !by_agents) h->co_cuts(G, cuts);
    }
    if (n_shards) *n_shards = G;
    if (samples) {
        for (int g = 0; g < 8; ++g) samples[g] = 0;
        if (!cuts.empty()) { for (int g = 0; g < G; ++g) samples[g] = cuts[g + 1] - cuts[g]; }
        else if (by_agents) { for (int g = 0; g < G; ++g) samples[g] = h->cfg.samples; }   // the AGENTS are split: every engine rolls out all samples of its agents
        else {
            samples[0] = h->co_active() ? h->co_k0 : h->cfg.samples;
            for (int g = 1; g < G; ++g) samples[g] = h->subs[g - 1]->cfg.samples;
        }
    }
    API_END(h)
}

const char* mppi_co_note(const mppi_engine* h) { return h ? h->co_fallback.c_str() : ""; }

int mppi_rollout_kernel(mppi_engine* h, int32_t* kind) {
    API_BEGIN_FAST(h)
    if (!kind) fail(MPPI_E_INVALID, "NULL argument");
    *kind = h->last_rollout_kind;
    API_END(h)
}

int mppi_engine_info(mppi_engine* h, size_t* hbm_bytes, int32_t* rollout_blocks, int32_t* update_blocks) {
    API_BEGIN_FAST(h)
    if (hbm_bytes) { *hbm_bytes = h->hbm_bytes; for (auto* e : h->subs) *hbm_bytes += e->hbm_bytes; }
    // what a tick launches: the scan kernel alone (no update kernel), or rollout + update
    const bool scan = h->small_nb > 0;
    if (rollout_blocks) *rollout_blocks = (scan ? h->small_nb : h->roll_blocks) * h->cfg.n_agents;
    if (update_blocks) *update_blocks = scan ? 0 : h->NCH * h->cfg.horizon * h->cfg.n_agents;
    API_END(h)
}

}  // extern "C"
