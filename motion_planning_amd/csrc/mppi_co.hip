// mppi_co.hip -- co-scheduled engines inside ONE handle (mppi_config.co_shards): the fused device-noise tick split over engines on
// this one GPU, by samples (shards coupled through the p2p mailboxes) or by agents (nothing exchanged).  No kernel is launched from
// this file: the engines' own pipelines (mppi_engine.hip) do that.
#include "mppi_engine.hpp"

void mppi_engine::co_sync_subs() {
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
