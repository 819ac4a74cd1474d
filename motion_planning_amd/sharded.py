"""K sharded over GPUs: one process per GPU, one tiny exchange per control tick.

The reference is single-process (control/src/mppi); its only cross-sample coupling is the
per-timestep softmax of update_action (:187-196).  Splitting K over G ranks therefore needs
exactly one exchange per tick: each rank reduces its samples to the tuple
{min V, sum e, sum e*eps, sum eps, count} per (agent, t) -- [A][T][8] float64, 3.2 KB at
T=50 -- the ranks all-gather those tuples and every rank finishes the tick identically
(merge, control update, clip, filter, clip, plant step, shift).  The message is
latency-bound, so it is ONE exchange of the whole [A][T][8] block, not one per timestep.

Exchange back-ends (``exchange=``):
  "rccl"  torch.distributed.all_gather_into_tensor (backend "nccl" = RCCL over xGMI; "gloo" on
          CPU in the tests) on the stream the engine runs on;
  "p2p"   the engine's own one-shot all-gather (mppi_p2p_*): every rank's publish kernel stores its
          tuples straight into each peer's IPC-mapped mailbox over xGMI and raises a flag; the
          finalize kernel waits for the G flags.  No collective launch, no host involvement per tick.
  "auto"  p2p when its probe passes on every rank (sharded.probe_p2p), else rccl.
Independent agents (BASELINE config 5) are replicas: no exchange at all (make_replica_ticker).
Co-scheduled shards (make_co_scheduled_ticker): the same K-split with the shards on ONE GPU in one process, each engine on
its own stream, coupled only by the p2p mailboxes -- one shard's HBM-bound update kernel runs under the other's
VALU-bound rollout (+5 % rollouts/s at K = 10^6 on one MI355X).
"""
TUPLE_W = 8


class _DevBuf(object):
    """Zero-copy view of a device buffer owned by libmppi_hip for torch.as_tensor."""

    def __init__(self, ptr, n_f64):
        self.__cuda_array_interface__ = {"shape": (n_f64,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _one_hip_runtime():
    """torch wheels bundle a libamdhip64.so of their own.  Loaded first, it also serves libmppi_hip.so (same SONAME) and
    everything is one runtime; but an engine created BEFORE `import torch` binds the system runtime and torch then maps
    its own copy next to it -- two HIP runtimes whose pointers and streams mean nothing to each other.  This module
    hands engine memory to torch (zero-copy tensor, collectives on it), so it refuses to run in such a process."""
    try:
        paths = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
    except OSError:
        return
    if len(paths) > 1:
        raise RuntimeError("two HIP runtimes are mapped in this process (%s): import torch before creating the first "
                           "motion_planning_amd engine, so that libmppi_hip binds the runtime torch uses" % ", ".join(paths))


class HipShard(object):
    """Adapter: one libmppi_hip Engine as a shard (partials exposed as a torch CUDA tensor)."""

    def __init__(self, engine, torch_device, use_torch_stream=True):
        import torch
        _one_hip_runtime()
        self.engine = engine
        self.device = torch_device
        ptr, nbytes = engine.partials()
        self._part = torch.as_tensor(_DevBuf(ptr, nbytes // 8), device=torch_device)
        assert self._part.data_ptr() == ptr, "torch copied the partials buffer instead of aliasing it"
        if use_torch_stream:
            # all work of the engine goes to torch's current stream so that collectives order with it
            engine.set_stream(torch.cuda.current_stream(torch_device).cuda_stream)

    def tick_begin(self, state, goal, noise, seed, tick_id):
        self.engine.tick_begin(state, goal, noise=noise, seed=seed, tick_id=tick_id)

    def partials_tensor(self):
        return self._part

    def tick_finish(self, gathered, n_shards):
        if gathered is None:
            self.engine.tick_finish()
        else:
            self.engine.tick_finish(gathered.data_ptr(), n_shards)

    def tick_fused(self, state, goal, noise, seed, tick_id):
        """No exchange follows (single process / replicas): the engine's fused tick, asynchronous."""
        self.engine.tick_async(state, goal, noise=noise, seed=seed, tick_id=tick_id)

    def get_outputs(self):
        return self.engine.get_outputs()


def shard_range(samples_total, world_size, rank):
    """Contiguous, balanced split of the global sample index range (first ranks get the remainder)."""
    base, rem = divmod(int(samples_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedTicker(object):
    """Drives one shard per rank through tick_begin -> exchange -> tick_finish.

    ``shard`` is anything with the HipShard interface (the CPU tests plug the oracle in);
    ``group`` a torch.distributed process group (None: the default group, or single process);
    ``exchange``: "rccl" (the collective), "p2p" (the engine's own mailbox all-gather, set up by
    make_hip_ticker), "none" (replicas / single process: every shard finishes on its own partials).
    """

    def __init__(self, shard, group=None, exchange="rccl"):
        import torch
        import torch.distributed as dist
        self.shard = shard
        self.group = group
        self.dist = dist if (exchange != "none" and dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.exchange = exchange if self.dist else "none"
        self._gathered = None
        if self.exchange == "rccl":
            part = shard.partials_tensor()
            # flat [world * A*T*8] receive buffer (rank-major), the layout mppi_tick_finish expects
            # (a 1-rank process group still goes through the collective: same code path as N ranks)
            self._gathered = torch.empty(self.world * part.numel(), dtype=part.dtype, device=part.device)
        self._timing = False
        self._ev = []

    # -- optional timing of the exchange step (bench.py diagnostics) ----------------------------------
    def time_exchange(self, on):
        self._timing = bool(on) and self.exchange == "rccl"
        self._ev = []

    def exchange_times_us(self):
        """Mean duration of the exchange step over the ticks run since time_exchange(True); None when there is
        no bracketable exchange (no group; or p2p, whose wait lives inside the finalize kernel)."""
        if not self._ev:
            return None
        self._ev[-1][1].synchronize()
        return float(sum(1e3 * a.elapsed_time(b) for a, b in self._ev) / len(self._ev))

    def tick_async(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        if self.exchange == "none" and hasattr(self.shard, "tick_fused"):
            self.shard.tick_fused(state, goal, noise, seed, tick_id)
            return
        self.shard.tick_begin(state, goal, noise, seed, tick_id)
        if self.exchange == "rccl":
            if self._timing:
                import torch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            # one all-gather of [A][T][8] f64 per tick (RCCL over xGMI on GPUs)
            self.dist.all_gather_into_tensor(self._gathered, self.shard.partials_tensor(), group=self.group)
            if self._timing:
                b.record()
                self._ev.append((a, b))
            self.shard.tick_finish(self._gathered, self.world)
        elif self.exchange == "p2p":
            self.shard.engine.tick_exchange_p2p()   # publish into every peer's mailbox + finalize behind the flags
        else:
            self.shard.tick_finish(None, 1)

    def tick(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        self.tick_async(state, goal, noise, seed, tick_id)
        return self.shard.get_outputs()


def make_hip_ticker(samples_total, horizon, n_agents=1, storage="f32", local_rank=0, group=None, exchange="auto",
                    pinned_total=None, **engine_kw):
    """Engine for this rank's slice of the samples + the ticker around it.
    pinned_total: what the engine's size rules are given as the whole controller's sample count (mppi_config.samples_total).  None
    (default) = samples_total: every rank picks its kernels by the WHOLE controller's size -- N = 1 / 2 / 4 / 8 then end every tick
    with the same controls to rounding; 0: every rank chooses by its own share (faster small shares, results equal to ~1e-6 only);
    another number: this group is itself one share of a larger controller."""
    import torch
    import torch.distributed as dist
    from .mppi import Engine
    in_group = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if in_group else 1
    rank = dist.get_rank(group) if in_group else 0
    lo, hi = shard_range(samples_total, world, rank)
    torch.cuda.set_device(local_rank)
    eng = Engine(hi - lo, horizon, n_agents=n_agents, storage=storage, device=local_rank, sample_offset=lo,
                 samples_total=samples_total if pinned_total is None else pinned_total, **engine_kw)
    kind = "none" if not in_group else "rccl"
    report = {"requested": exchange}
    if in_group and world > 1 and exchange in ("auto", "p2p"):
        from . import p2p
        if p2p.setup(eng, group, rank, world, local_rank, required=(exchange == "p2p"), report=report):
            kind = "p2p"
    # only the RCCL collective has to order with torch's stream; a single process and the p2p exchange keep the
    # engine's own stream (which is also what hipGraph capture needs -- the null stream cannot be captured)
    shard = HipShard(eng, torch.device("cuda", local_rank), use_torch_stream=(kind == "rccl"))
    ticker = ShardedTicker(shard, group, exchange=kind)
    report["ran"] = ticker.exchange
    ticker.exchange_report = report     # what the set-up of the exchange did on this rank (bench.py prints it per rank)
    return ticker, eng


class P2PTicker(object):
    """One rank of a K-sharded controller on the engine's own peer-to-peer exchange, WITHOUT torch: the rank's engine, connected
    to its peers by mppi_p2p_rendezvous (IPC handles through files), ticking tick_begin -> publish -> finalize-behind-the-flags.
    What ShardedTicker(exchange="p2p") does once a process group has carried the handles -- for launchers that have no process
    group (a ROS launch file starting one node per GPU, mpirun, a shell loop): rank and world size come from the caller."""

    exchange = "p2p"

    def __init__(self, engine, world, rank):
        self.engine, self.world, self.rank = engine, int(world), int(rank)

    def tick_async(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        self.engine.tick_begin(state, goal, noise=noise, seed=seed, tick_id=tick_id)
        self.engine.tick_exchange_p2p()

    def tick(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        self.tick_async(state, goal, noise, seed, tick_id)
        return self.engine.get_outputs()


def make_p2p_ticker(samples_total, horizon, rank, world, rendezvous_prefix, n_agents=1, storage="f32", device=0,
                    timeout_ms=60000, selftest_rounds=2, pinned_total=None, **engine_kw):
    """This rank's slice of the samples (global sample offsets) + the exchange, no torch and no process group: every rank calls this
    with the same `rendezvous_prefix` (a path inside a directory the launcher made for THIS run; the files `<prefix>.<rank>` are left
    for the launcher to remove).  Returns (ticker, engine)."""
    from .mppi import Engine
    lo, hi = shard_range(samples_total, world, rank)
    eng = Engine(hi - lo, horizon, n_agents=n_agents, storage=storage, device=device, sample_offset=lo, co_shards=1,
                 samples_total=samples_total if pinned_total is None else pinned_total, **engine_kw)   # (size rules: see make_hip_ticker)
    try:
        eng.p2p_rendezvous(rendezvous_prefix, world, rank, timeout_ms)
        if selftest_rounds:
            eng.p2p_selftest(selftest_rounds)   # collective: round trips of a known pattern, consumed by a kernel the way finalize does
    except Exception:
        eng.close()
        raise
    return P2PTicker(eng, world, rank), eng


def make_replica_ticker(samples, horizon, n_agents, storage="f32", local_rank=0, **engine_kw):
    """Independent agents (BASELINE config 5): this rank's agents in one engine, no exchange.  Pass ``agent_offset`` = the global
    index of this rank's first agent: the device-noise streams are keyed by the global agent index, so the ranks together draw
    exactly what one engine holding all agents would."""
    import torch
    from .mppi import Engine
    torch.cuda.set_device(local_rank)
    eng = Engine(samples, horizon, n_agents=n_agents, storage=storage, device=local_rank, **engine_kw)
    shard = HipShard(eng, torch.device("cuda", local_rank), use_torch_stream=False)
    return ShardedTicker(shard, None, exchange="none"), eng


class CoScheduledTicker(object):
    """G engines on one GPU in this process, the samples split between them (global sample offsets, so the noise streams
    are those of the unsharded controller), every engine on its own stream; the only coupling is the p2p mailbox
    exchange, i.e. flags polled by the finalize kernels -- no event, no host wait between the engines.  Every engine
    finishes every tick identically; outputs are read from the first."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.exchange = "p2p (co-scheduled x%d)" % len(self.engines)
        for g, e in enumerate(self.engines):
            e.p2p_create(len(self.engines), g)
        ptrs = [e.p2p_mailbox_ptr() for e in self.engines]
        for e in self.engines:
            e.p2p_connect(local_ptrs=ptrs)

    def set_nominal(self, uvec, agent=0):
        for e in self.engines:
            e.set_nominal(uvec, agent=agent)

    def tick_async(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        for e in self.engines:
            e.tick_begin(state, goal, noise=noise, seed=seed, tick_id=tick_id)
        for e in self.engines:   # one thread drives all engines: every publish is enqueued before any finalize that waits for it
            e.p2p_publish()
        for e in self.engines:
            e.tick_finish_p2p()

    def tick(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        self.tick_async(state, goal, noise, seed, tick_id)
        return self.engines[0].get_outputs()

    def get_outputs(self):
        return self.engines[0].get_outputs()

    def get_nominal(self, agent=0):
        return self.engines[0].get_nominal(agent)

    def synchronize(self):
        for e in self.engines:
            e.synchronize()

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def co_scheduled_cuts(samples_total, n_shards, chunk=8192):
    """Shard boundaries [0, c1, ..., K]: as equal as multiples of `chunk` allow (equal halves measured best), every shard
    non-empty; interior boundaries are multiples of `chunk` whenever K has room for them."""
    K = int(samples_total)
    if not 2 <= n_shards <= 8:
        raise ValueError("2..8 co-scheduled shards")
    if K < n_shards:
        raise ValueError("%d samples do not split into %d shards" % (K, n_shards))
    cuts = [0]
    for g in range(1, n_shards):
        c = int(round(g * K / n_shards / chunk)) * chunk
        lo, hi = cuts[-1] + 1, K - (n_shards - g)      # leave at least one sample for every shard before and behind
        if not lo <= c <= hi:                          # K too small for chunk-aligned boundaries: plain balanced split
            c = min(hi, max(lo, (g * K) // n_shards))
        cuts.append(c)
    cuts.append(K)
    return cuts


def make_co_scheduled_ticker(samples_total, horizon, n_shards=2, n_agents=1, storage="f32", device=0, chunk=8192, **engine_kw):
    """K split into n_shards engines on one GPU (boundaries on multiples of `chunk`, the update kernel's chunk of
    fp32-storage samples, so that no shard ends in a ragged chunk)."""
    from .mppi import Engine
    cuts = co_scheduled_cuts(samples_total, n_shards, chunk)
    engine_kw.setdefault("tick_path", "lanes")
    engines = []
    try:
        for g in range(n_shards):
            engines.append(Engine(cuts[g + 1] - cuts[g], horizon, n_agents=n_agents, storage=storage, device=device,
                                  sample_offset=cuts[g], co_shards=1, **engine_kw))
        return CoScheduledTicker(engines)
    except Exception:
        for e in engines:
            e.close()
        raise
