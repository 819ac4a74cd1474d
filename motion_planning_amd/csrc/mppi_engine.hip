// mppi_engine.hip -- the core of libmppi_hip.so: owns the HBM buffers, picks launch geometry for gfx950 and enqueues the kernels
// of mppi_kernels.hpp / rollout_pk.hpp on the engine's HIP stream.  Every kernel launch of the library is in this file (the
// non-template kernels are emitted by this translation unit only); the engine object is declared in mppi_engine.hpp.
#define MPPI_ENGINE_CORE_TU 1
#include "mppi_engine.hpp"
#include "rollout_launch.hpp"
#include "rollout_pk.hpp"
#include "rollout_fused.hpp"

// publish this rank's tuples for the next epoch and return the wait descriptor for the consumer.
// src = merged tuples [A][T][8]; src == nullptr: merge d_part's direct_n tuples per row on the way (merge_skipped)
mppi::P2PWait mppi_engine::p2p_publish(const double* src) {
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

void mppi_engine::set_inputs(const double* state, const double* goal, bool zero_copy) {
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

void mppi_engine::launch_rollout(hipStream_t st, int k0, int k1, bool ph, bool store, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
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

// The small-K tick keeps V in registers.  When a caller asks for it afterwards (mppi_download_value,
// mppi_update), the lane-per-sample rollout kernel re-runs the tick's rollout from the pre-tick
// snapshot the scan kernel left behind, with the same noise (re-drawn bit-identically, or the
// injected buffer), and leaves dP / Stot / base / epart as any rollout does.  (The scan tick only: a co-scheduled
// tick's V is complete in this handle's own arrays -- its shards fill columns of them -- and is read in place.)
void mppi_engine::materialise_value() {
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

void mppi_engine::launch_scan_tick(bool ph, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
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

void mppi_engine::launch_regen(hipStream_t st, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
    const int spd = noise_pack == 1 ? mppi::NoisePack<1>::kSteps : (noise_pack == 2 ? mppi::NoisePack<2>::kSteps : mppi::NoisePack<0>::kSteps);
    dim3 g((cfg.samples + 255) / 256, (cfg.horizon + spd - 1) / spd, cfg.n_agents);
    if (f64()) hipLaunchKernelGGL(mppi::eps_regen_kernel<double>, g, dim3(256), 0, st, P, static_cast<double*>(d_eps), seed, tick, tick_ptr);
    else if (noise_pack == 1) hipLaunchKernelGGL((mppi::eps_regen_kernel<float, 1>), g, dim3(256), 0, st, P, static_cast<float*>(d_eps), seed, tick, tick_ptr);
    else if (noise_pack == 2) hipLaunchKernelGGL((mppi::eps_regen_kernel<float, 2>), g, dim3(256), 0, st, P, static_cast<float*>(d_eps), seed, tick, tick_ptr);
    else hipLaunchKernelGGL(mppi::eps_regen_kernel<float>, g, dim3(256), 0, st, P, static_cast<float*>(d_eps), seed, tick, tick_ptr);
    HIPCHK(hipGetLastError());
}

void mppi_engine::ensure_epart(hipStream_t st) {
    if (epart_ready) return;  // noise uploaded by the caller and never rolled out: sum it now
    dim3 g((cfg.samples + 255) / 256, cfg.horizon * 2, cfg.n_agents);
    if (f64()) hipLaunchKernelGGL(mppi::eps_wavesum_kernel<double>, g, dim3(256), 0, st, P, static_cast<const double*>(d_eps), static_cast<double*>(d_epart));
    else hipLaunchKernelGGL(mppi::eps_wavesum_kernel<float>, g, dim3(256), 0, st, P, static_cast<const float*>(d_eps), static_cast<float*>(d_epart));
    HIPCHK(hipGetLastError());
    epart_ready = true;
}

// chunk length of the update kernel: twice the streaming shape's where that is what takes a row to <= kDirectTuples chunk tuples
// (no merge launch in the fused tick); d_part is sized for the SHORT chunks, so the choice can change with the option
void mppi_engine::pick_update_shape() {
    const int ch8 = f64() ? mppi::UpdCfg<double, 8>::CH : mppi::UpdCfg<float, 8>::CH;
    const int n8 = (cfg.samples + ch8 - 1) / ch8, n16 = (cfg.samples + 2 * ch8 - 1) / (2 * ch8);
    upd_nv = (n8 > kDirectTuples && n16 <= kDirectTuples) ? 16 : 8;
    if (noise_pack) upd_nv = 8;   // (the other noise packings' re-draws are built into the streaming shape only)
    CH = ch8 * upd_nv / 8;
    NCH = (cfg.samples + CH - 1) / CH;
}

void mppi_engine::launch_update(hipStream_t st, int ch0, int nch, const uint32_t* tick_ptr) {
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

void mppi_engine::launch_merge(int nch) {
    Scope sc(this, MPPI_KERNEL_MERGE);
    hipLaunchKernelGGL(mppi::merge_kernel, dim3(cfg.horizon, cfg.n_agents), dim3(nch > 128 ? 256 : 64), 0, stream, P, d_part, nch, d_merged);
    HIPCHK(hipGetLastError());
}

// Everything a tick / rollout can refuse for, checked BEFORE any state of the handle changes (inputs staged, lazy-noise bookkeeping,
// a co-scheduled group half way through its launches): a refused call leaves the handle exactly as it was.
// tick_path: the call is a tick (its device noise is not stored unless option store_eps says so); else mppi_rollout
void mppi_engine::check_noise_mode(int noise_mode, bool tick_path) {
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

// which of the two lane-per-sample rollouts a device-noise tick of this engine takes (see launch_rollout)
bool mppi_engine::pick_pk(bool ph, bool store, int k0, int k1) const {
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

// whether a device-noise tick of this engine runs the fused fp64 kernel (rollout_fused.hpp; rules: mppi_engine.hpp fused_nb)
bool mppi_engine::pick_fused(bool ph, bool store) {
    if (fused_nb == 0 || !ph || store || !use_pk || capturing || noise_pack || general_cost() || ro_state || ro_goal || ro_unom) return false;
    const long k_rule = cfg.samples_total > 0 ? (long)cfg.samples_total : (long)cfg.samples;
    if (pk_min_set) return (long)cfg.n_agents * k_rule >= pk_min_samples;
    return (long)cfg.n_agents * k_rule >= kFusedMinSamples && regime_weight() < kFusedRegimeCut;
}
void mppi_engine::launch_fused(uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
    // the nominal trajectory's per-step table: the set the previous tick's finalize kernel left for exactly these inputs, else nominal_kernel now
    if (table_valid) {
        if (!table_taken) { use_table_set(tab ^ 1); table_taken = true; }
    } else {
        Scope sc(this, MPPI_KERNEL_NOMINAL);
        hipLaunchKernelGGL(mppi::nominal_kernel, dim3(cfg.n_agents), dim3(mppi::kNomThreads), (size_t)cfg.horizon * sizeof(double), stream, P,
                           in_state ? in_state : (const double*)d_state, in_goal ? in_goal : (const double*)d_goal, d_unom, d_tc, d_base);
        HIPCHK(hipGetLastError());
    }
    mppi::RolloutFusedArgs a{};
    a.P = P; a.P.snap = d_prev;   // (V is never stored: mppi_download_value / mppi_update re-run the tick from this snapshot)
    a.stream = stream; a.seed = seed; a.tick = tick; a.tick_ptr = tick_ptr;
    a.state = in_state ? in_state : d_state; a.goal = in_goal ? in_goal : d_goal; a.unom = d_unom;
    a.tc = d_tc; a.part = d_part; a.NB = fused_nb; a.nterm = nterm;
    a.split = mppi::rollout_fused_lds(cfg.horizon, true) <= (size_t)160 * 1024;   // (T <= 56: the pairs' noise buffers fit next to the prefix rows)
    {
        Scope sc(this, MPPI_KERNEL_ROLLOUT);
        const hipError_t e = mppi::launch_rollout_fused(a);
        if (e != hipSuccess) fail(MPPI_E_HIP, "fused rollout launch failed: %s", hipGetErrorString(e));
    }
    last_rollout_pk = false;
    last_rollout_kind = MPPI_ROLLOUT_FUSED;
    if (in_slot >= 0) inputs_consumed();
}

void mppi_engine::run_pipeline(int noise_mode, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr, bool skip_small_merge) {
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
    if (pick_fused(ph, store)) {   // fp64 storage, under way: rollout + cost-to-go + partials in one kernel, V stays on the chip
        eps_lazy = true; injected_ready = false;
        launch_fused(seed, tick, tick_ptr);
        // (one tuple per row and workgroup of four waves; a handful of them -- small engines -- the consumer merges itself)
        const int n_tuples = fused_nb / 4;
        merge_skipped = (skip_small_merge || (p2p_connected && !p2p_internal)) && n_tuples <= kDirectTuples;
        direct_n = n_tuples;
        if (!merge_skipped) launch_merge(n_tuples);
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

void mppi_engine::run_nominal() {
    if (inline_nominal()) return;
    Scope sc(this, MPPI_KERNEL_NOMINAL);
    hipLaunchKernelGGL(mppi::nominal_kernel, dim3(cfg.n_agents), dim3(mppi::kNomThreads),
                       (size_t)cfg.horizon * sizeof(double), stream, P, d_state, d_goal, d_unom, d_tc, d_base);
    HIPCHK(hipGetLastError());
}

void mppi_engine::run_rollout(int noise_mode, uint64_t seed, uint32_t tick, const uint32_t* tick_ptr) {
    check_noise_mode(noise_mode, /*tick_path=*/false);
    co_value_dirty = false;   // (V of every agent is about to be this engine's own)
    const bool ph = noise_mode == MPPI_NOISE_PHILOX;
    // (the 16-bit packing is drawn by the mixed-precision kernel, which does not store its noise: the re-draw kernel leaves the same bits in d_eps)
    launch_rollout(stream, 0, cfg.samples, ph, !(ph && noise_pack), seed, tick, tick_ptr);
    if (ph && noise_pack) launch_regen(stream, seed, tick, tick_ptr);
    eps_lazy = false; injected_ready = true;
    noise_ready = true; value_ready = true; value_lazy = false; partials_ready = false; epart_ready = true;
}

void mppi_engine::run_update() {
    materialise_value();
    if (!noise_ready || !value_ready) fail(MPPI_E_STATE, "update needs a rollout (or uploaded V and eps) first");
    materialise_eps();
    launch_update(stream, 0, NCH);
    launch_merge(NCH);
    merge_skipped = false;  // (an earlier fused tick may have left its tuples unmerged: these are merged)
    partials_ready = true;
}

// shard_stride: elements between consecutive shards' [A][T][8] blocks in `gathered` (0: packed)
void mppi_engine::run_finalize(const double* gathered, int G, int flags, mppi::P2PWait wait, size_t shard_stride) {
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
    // (a fused fp64 tick always wants it: that kernel only ever LOADS its table)
    if ((flags & 3) == 3 && !(flags & 4) && (hoist_on() || last_rollout_kind == MPPI_ROLLOUT_FUSED) && inline_nominal() && small_nb == 0 && !capturing) flags |= 32;
    if (fused_nb > 0) flags |= 64;   // (the regime report the fused fp64 tick's rule reads)
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
                       tcb[tab ^ 1], baseb[tab ^ 1], f64() ? nullptr : pkb[tab ^ 1],   // (the deviation-form rows are the mixed rollout's: fp32 storage only)
                       
                       lanes_fresh_state ? (const double*)(d_prev + (size_t)cfg.n_agents * 2 * cfg.horizon) : (const double*)nullptr, d_goal);
    lanes_fresh_state = lanes_fresh_goal = false;
    if (flags & 1) out_via_host = host_out;
    HIPCHK(hipGetLastError());
    partials_ready = false;
    table_valid = (flags & 32) != 0;   // (this launch rewrote the nominal controls: a table it did not refresh is stale)
    table_taken = false;
}

void mppi_engine::init(const mppi_config& c) {
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
    if (cfg.samples_total != 0 && (cfg.samples_total < (int64_t)cfg.sample_offset + cfg.samples || cfg.samples_total > 0xFFFFFFFFll))
        fail(MPPI_E_INVALID, "samples_total = %lld: 0 (this handle is the whole controller) or >= sample_offset + samples = %lld (global sample ids are 32-bit)",
             (long long)cfg.samples_total, (long long)cfg.sample_offset + cfg.samples);
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
    d_ufilt = dev_alloc<double>((size_t)2 * A * 2 * T, hbm_bytes);   // [A][2][T] filtered | [A][2][T] updated + clipped, not filtered (mppi_update)
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
    // the fused fp64 tick: one wave per SIMD of the chip (four to a workgroup, a workgroup per CU: its LDS), split over the agents
    if (f64() && small_nb == 0 && inline_nominal() && T <= 64 && nterm != 0 && !is_co_sub) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        const int groups = (K + 63) / 64;
        fused_nb = std::max(4, std::min((groups + 3) / 4 * 4, std::max(1, cus / A) * 4));
        if (mppi::rollout_fused_lds(T) > (size_t)160 * 1024) fused_nb = 0;
    }
    {
        const int ch8 = f64() ? mppi::UpdCfg<double, 8>::CH : mppi::UpdCfg<float, 8>::CH;
        d_part = dev_alloc<double>((size_t)A * T * std::max(std::max((K + ch8 - 1) / ch8, small_nb), fused_nb) * mppi::kTupleW, hbm_bytes);
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
    HIPCHK(hipMemsetAsync(d_ufilt, 0, (size_t)2 * A * 2 * T * sizeof(double), stream));
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

mppi_engine::~mppi_engine() {
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

// ---- the kernels behind the stand-alone calls of the ABI (mppi_abi.hip / mppi_p2p.hip hold the calls themselves) -------------------
// noise rows between the caller's layout (d_tmp: [A][T][2][K] float64) and the engine's (d_eps: S [A][T][2][Ks])
void mppi_engine::launch_noise_rows(bool pack) {
    const int A = cfg.n_agents, T = cfg.horizon, K = cfg.samples;
    dim3 grid(std::min((K + 255) / 256, 1024), A * T * 2);
    if (pack) {
        if (f64()) hipLaunchKernelGGL(mppi::pack_rows_kernel<double>, grid, dim3(256), 0, stream, d_tmp, static_cast<double*>(d_eps), K, P.Ks, (const double*)nullptr, 1);
        else hipLaunchKernelGGL(mppi::pack_rows_kernel<float>, grid, dim3(256), 0, stream, d_tmp, static_cast<float*>(d_eps), K, P.Ks, (const double*)nullptr, 1);
    } else {
        if (f64()) hipLaunchKernelGGL(mppi::unpack_rows_kernel<double>, grid, dim3(256), 0, stream, static_cast<const double*>(d_eps), d_tmp, K, P.Ks, (const double*)nullptr, 1);
        else hipLaunchKernelGGL(mppi::unpack_rows_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float*>(d_eps), d_tmp, K, P.Ks, (const double*)nullptr, 1);
    }
    HIPCHK(hipGetLastError());
}
// V between the caller's layout (d_tmp: [A][T][K] float64, absolute) and the engine's (base + Stot - dP); pack: base := the row minimum
void mppi_engine::launch_value_rows(bool pack) {
    const int A = cfg.n_agents, T = cfg.horizon, K = cfg.samples;
    dim3 grid(std::min((K + 255) / 256, 1024), A * T);
    if (pack) {
        hipLaunchKernelGGL(mppi::row_min_kernel, dim3(A * T), dim3(256), 0, stream, d_tmp, K, d_base);
        HIPCHK(hipGetLastError());
        if (f64()) hipLaunchKernelGGL(mppi::value_pack_kernel<double>, grid, dim3(256), 0, stream, (const double*)d_tmp, (const double*)d_base, static_cast<double*>(d_dP), static_cast<double*>(d_stot), K, P.Ks, T);
        else hipLaunchKernelGGL(mppi::value_pack_kernel<float>, grid, dim3(256), 0, stream, (const double*)d_tmp, (const double*)d_base, static_cast<float*>(d_dP), static_cast<float*>(d_stot), K, P.Ks, T);
    } else {
        if (f64()) hipLaunchKernelGGL(mppi::value_unpack_kernel<double>, grid, dim3(256), 0, stream, static_cast<const double*>(d_dP), static_cast<const double*>(d_stot), (const double*)d_base, d_tmp, K, P.Ks, T);
        else hipLaunchKernelGGL(mppi::value_unpack_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float*>(d_dP), static_cast<const float*>(d_stot), (const double*)d_base, d_tmp, K, P.Ks, T);
    }
    HIPCHK(hipGetLastError());
}
void mppi_engine::launch_plant() {   // MPPI.perform_action, control/src/mppi:210-213
    hipLaunchKernelGGL(mppi::plant_kernel, dim3((cfg.n_agents + 63) / 64), dim3(64), 0, stream, P, d_state, d_unom, d_out);
    HIPCHK(hipGetLastError());
}
void mppi_engine::launch_shift() {   // control/src/mppi:100-101
    hipLaunchKernelGGL(mppi::shift_kernel, dim3(cfg.n_agents * 2), dim3(256), (size_t)cfg.horizon * sizeof(double), stream, P, d_unom);
    HIPCHK(hipGetLastError());
}
void mppi_engine::launch_p2p_check(const mppi::P2PWait& w, const double* slots, int n, int slot_f64, double* d_got, int* d_status) {
    hipLaunchKernelGGL(mppi::p2p_check_kernel, dim3(1), dim3(256), 0, stream, w, slots, n, slot_f64, d_got, d_status);
    HIPCHK(hipGetLastError());
}
