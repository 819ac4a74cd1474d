// mppi_abi.hip -- the C ABI of include/mppi_hip.h / mppi_hip_diag.h (all but mppi_p2p_*: mppi_p2p.hip): argument checks, call order,
// error translation (nothing throws across the ABI), the option switches.  The work itself is the engine's (mppi_engine.hip).
#include "mppi_engine.hpp"

namespace {
thread_local std::string g_create_error = "";
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
    h->regime_fresh_start();
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
    h->regime_fresh_start();
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
    h->launch_noise_rows(/*pack=*/true);
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
    h->launch_noise_rows(/*pack=*/false);
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
    h->launch_value_rows(/*pack=*/false);
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
    h->launch_value_rows(/*pack=*/true);   // (baseline := per-row minimum, so the stored offsets are >= 0 and small near the minimum)
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

int mppi_get_unfiltered(mppi_engine* h, double* uvec) {
    API_BEGIN(h)
    if (!uvec) fail(MPPI_E_INVALID, "uvec is NULL");
    const size_t n = (size_t)h->cfg.n_agents * 2 * h->cfg.horizon;
    HIPCHK(hipMemcpyAsync(uvec, h->d_ufilt + n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    h->wait_stream(__func__);
    API_END(h)
}

int mppi_plant_step(mppi_engine* h, const double* state, double* next_state) {
    API_BEGIN(h)
    h->invalidate_table();
    h->co_synced = false;   // (the co-scheduled shards no longer hold this engine's nominal controls / state)
    const int A = h->cfg.n_agents;
    if (state) { h->stage_upload(state, h->d_state, (size_t)A * 3); h->have_state = true; }
    if (!h->have_state) fail(MPPI_E_STATE, "no state resident");
    h->launch_plant();
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
    h->launch_shift();
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
