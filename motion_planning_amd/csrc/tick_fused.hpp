// tick_fused.hpp -- tick_fused_kernel: the rollout AND the per-timestep softmax partials of one control tick in ONE launch
// (MPPI.get_cost2go, control/src/mppi:127-178, followed by the K-reduction of MPPI.update_action, :187-196).
//
// Why: the two big kernels of a tick sit on different roofs -- the rollout on VALU issue, the update on HBM -- and as two
// launches they run strictly one after the other.  Here they are work items of one launch, handed to workgroups through
// ordered tickets: a chunk column (8192 samples of one agent) is produced by RB rollout work items and consumed by T update
// work items (one per timestep), and the update items of a column are queued a few columns BEHIND its rollout items, so the
// HBM-bound update of the columns that are done runs on the same CUs, at the same time, as the VALU-bound rollout of the
// columns that are not.
//
// Queues.  One ticket counter per XCD (a single word saturates at ~90 dequeues/us; 8 heads also keep a column on ONE XCD:
// its Stot chunk is re-read by T update items, from that XCD's L2).  Queue x owns the columns col = x, x + 8, ...; its ticket
// order is
//     rollout items of its first L columns | for every column j: [rollout items of column j + L] [update items of column j]
// A workgroup takes ONE ticket (from its own XCD's queue; from the next queue with items left when its own is exhausted)
// and does that one item -- the hardware's workgroup dispatcher is the scheduler, there is no loop.
//
// No deadlock, whatever the dispatch order or placement: an update item waits (one lane polling the column's arrival
// counter, bounded by a device-side deadline) only for rollout items of ITS OWN queue with SMALLER ticket numbers; those
// tickets were handed out before, i.e. to workgroups that are already running, and rollout items never wait.
//
// Hand-off (cdna_hip_programming.md Guideline 16, R1): the rollout items store dP / Stot / the per-wave eps sums
// write-through (sc1), every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup barriers, one lane adds 1 to the
// column's counter (agent scope, no return); the update items poll that word relaxed and read with sc1 loads.
// The counters are double-buffered by launch parity: the workgroup that takes ticket 0 of queue 0 zeroes the OTHER parity's
// words for the next launch (launches of one engine are ordered by its stream).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "mppi_kernels.hpp"

namespace mppi {

constexpr int kFusedQueues = 8;        // one per XCD
constexpr int kFusedHeadStride = 16;   // uint32 words between ticket counters (one 64-byte line each)

struct FusedGeom {
    int n_cols;   // A * NCH chunk columns, col = a * NCH + ch
    int NCH;      // chunk columns per agent
    int RB;       // rollout work items per column (chunk samples / samples per rollout block)
    int T;        // update work items per column
    int L;        // a column's update items are queued behind the rollout items of the L next columns of its queue
};
enum { kFusedRollout = 0, kFusedUpdate = 1, kFusedNone = 2 };
struct FusedItem { int kind, col, idx; };   // idx: rollout block within the column | timestep

__host__ __device__ inline int fused_queue_cols(const FusedGeom& g, int x) { return g.n_cols > x ? (g.n_cols - x + kFusedQueues - 1) / kFusedQueues : 0; }
__host__ __device__ inline int fused_queue_len(const FusedGeom& g, int x) { return fused_queue_cols(g, x) * (g.RB + g.T); }
// ticket n of queue x -> work item
__host__ __device__ inline FusedItem fused_decode(const FusedGeom& g, int x, int n) {
    const int nx = fused_queue_cols(g, x);
    const int Lx = g.L < nx ? (g.L > 0 ? g.L : 0) : nx;
    FusedItem it{kFusedNone, 0, 0};
    if (n < 0 || n >= nx * (g.RB + g.T)) return it;
    int j;
    const int pro = Lx * g.RB;
    if (n < pro) { it.kind = kFusedRollout; j = n / g.RB; it.idx = n % g.RB; }
    else {
        const int m = n - pro, full = nx - Lx, seg = g.RB + g.T;
        if (m < full * seg) {
            const int grp = m / seg, r = m % seg;
            if (r < g.RB) { it.kind = kFusedRollout; j = grp + Lx; it.idx = r; }
            else { it.kind = kFusedUpdate; j = grp; it.idx = r - g.RB; }
        } else {
            const int m2 = m - full * seg;
            it.kind = kFusedUpdate; j = full + m2 / g.T; it.idx = m2 % g.T;
        }
    }
    it.col = x + kFusedQueues * j;
    return it;
}

struct FusedArgs {
    FusedGeom g;
    uint32_t* heads;       // [8 * kFusedHeadStride] ticket counters of this launch's parity
    uint32_t* done;        // [n_cols] rollout items that have arrived, per column (this parity)
    uint32_t* zero_base;   // the other parity's words (heads + done), zeroed for the next launch
    int zero_n;
    uint32_t* status;      // set != 0 when a wait ran into its deadline (finalize reports MPPI_E_TIMEOUT)
    unsigned long long timeout_ticks;   // of wall_clock64(); 0 = wait forever
    int prio_mode;         // bit 0: update items at the highest wave priority; bit 1: rollout waves by progress (prio_by_progress)
    // rollout
    const double *state, *goal, *unom;
    double *tc, *base;
    float *dP, *stot, *epart;
    uint64_t seed;
    uint32_t tick;
    float al_guard;
    // update
    double* part;
    int skip_light;
    // --- the update STREAM (rollout_arrive_kernel + update_stream_kernel, below) ---
    uint32_t epoch;          // this engine's tick number on the update stream: 1, 2, ...
    uint32_t* epoch_flag;    // the rollout launch's first workgroup stores `epoch` here: the gate of the update stream's workgroups
    uint32_t* col_done;      // [n_cols] rollout workgroups that have arrived, per column, MONOTONIC over the ticks (epoch * count)
    uint32_t* items_done;    // [8 lines] update items finished, by queue, monotonic; this tick's are done when the SUM reaches items_target
    uint32_t items_target;
    uint32_t* uheads;        // [8 * kFusedHeadStride] update-ticket counters, zeroed by the rollout launch's first workgroup
    uint32_t* exits;         // [8 lines] update-stream workgroups that have left their ticket loop for good (by XCD), monotonic over the ticks
    uint32_t exit_target;    // ... of all EARLIER ticks: nobody touches the ticket heads any more once this is reached
    int join;                // the tail launch: workgroup 0 leaves only when this tick's items are all done
    int bs;                  // samples per rollout workgroup (256 | 512)
    int ch;                  // samples per chunk column (8192)
};

struct FusedLaunch {
    DevParams P;
    FusedArgs F;
    hipStream_t stream;
    int inline_nominal;   // 1: one wave (T <= 64), 2: four waves (T <= 256)
    hipEvent_t ev_start, ev_stop;
};
// the two families of instantiations (one translation unit each, see tick_fused_*.hip)
hipError_t launch_tick_fused_pk(const FusedLaunch& a);                 // rollout items = rollout_pk_body (512 samples each)
template <int NTERM> hipError_t launch_tick_fused_f32(const FusedLaunch& a);   // rollout items = rollout_body<float, ...> (256 samples each)

// ---- The update stream: TWO concurrent launches instead of one fused one --------------------------------------------------
// Measured (profiles/r4_fused_single_launch_ab.jsonl): the single fused launch is SLOWER than rollout + update one after the other.
// A launch has ONE register allocation -- the rollout's 107 VGPRs -- so an update work item occupies a rollout-sized slot, the
// rollout already fills every slot (4 waves per SIMD), and queueing update items between rollout items only delays the rollout.
// What does fit next to four rollout waves of 112 VGPRs is ONE wave of <= 64: exactly an update workgroup per CU -- if it is a
// launch of its own.  So:
//   main stream   rollout_arrive_kernel   the stand-alone rollout grid; write-through stores; each workgroup adds 1 to its
//                                         column's arrival counter when its stores have drained; workgroup 0 opens the gate
//                 update_stream_kernel    (tail, big grid, join = 1) takes whatever update tickets are left -- by stream order
//                                         its columns are all complete -- and its workgroup 0 waits until every item of the
//                                         tick is done before the launch ends: the kernels behind it (merge, publish, finalize)
//                                         are ordered by the stream as they always were
//   second stream update_stream_kernel    (head start, one workgroup per CU) enqueued right behind the rollout launch: waits at
//                                         the gate, then takes update tickets in column order, each waiting for its column
// No host synchronisation and no event between the two streams; the coupling is a handful of device words, all monotonic over
// the ticks except the ticket heads: the gate (= the tick number), the per-column arrival counters, the items-done counter and
// an exit counter.  The ticket heads are re-armed by the rollout launch's first workgroup BEFORE it opens the gate, and only
// after every update-stream workgroup of earlier ticks has counted itself out (the exit counter; the host knows how many it
// launched) -- so a workgroup of an old tick that the hardware starts late can never take a ticket of a newer one: it finds the
// gate past its own tick number and leaves.  Whatever the order in which the hardware starts the launches, nothing can
// deadlock: rollout workgroups never wait (the first one only for workgroups that are on their way out), the head-start launch
// has at most one workgroup per CU (it cannot crowd the rollout out), and the tail launch starts after the rollout launch has
// ended.
struct UpdateStreamLaunch {
    DevParams P;
    FusedArgs F;
    hipStream_t stream;
    int blocks;
    hipEvent_t ev_start, ev_stop;
};
hipError_t launch_update_stream(const UpdateStreamLaunch& a);
hipError_t launch_rollout_arrive_pk(const FusedLaunch& a);
template <int NTERM> hipError_t launch_rollout_arrive_f32(const FusedLaunch& a);
// rollout workgroups with samples in chunk column ch of an agent (the ragged last column has fewer)
__host__ __device__ inline int fused_col_blocks(int K, int ch, int ch_samples, int bs) {
    const long lo = (long)ch * ch_samples, hi = lo + ch_samples < (long)K ? lo + ch_samples : (long)K;
    return hi > lo ? (int)((hi - lo + bs - 1) / bs) : 0;
}

#ifdef MPPI_FUSED_TU
__device__ __forceinline__ int fused_xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, width 4): the XCD this wave runs on (0..7 on gfx950)
    return (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & (kFusedQueues - 1);
}

// Rollout families: kWaves = waves per SIMD the kernel is compiled for, kBS = samples per rollout work item
#ifdef MPPI_FUSED_PK_TU
template <int IN> struct FusedRollPk {
    static constexpr int kWaves = 4, kBS = 512, kLdsPerStep = (int)sizeof(PkRow);
    static __device__ __forceinline__ void run(const DevParams& P, const FusedArgs& F, int bx, int a, bool probe, int prio) {
        rollout_pk_body<IN, true>(P, F.state, F.goal, F.tc, F.dP, F.stot, F.seed, F.tick, nullptr, F.epart, F.unom, F.base, F.al_guard, bx, a,
                                  probe, prio);
    }
};
#endif
template <int NTERM, int IN> struct FusedRollF32 {
    static constexpr int kWaves = 5, kBS = 256, kLdsPerStep = 5 * (int)sizeof(double);
    static __device__ __forceinline__ void run(const DevParams& P, const FusedArgs& F, int bx, int a, bool probe, int prio) {
        rollout_body<float, NTERM, true, false, IN, 0, false, true>(P, F.state, F.goal, F.tc, static_cast<float*>(nullptr), F.dP, F.stot, F.seed,
                                                                    F.tick, nullptr, 0, P.K, F.epart, F.unom, F.base, bx, a, probe, prio);
    }
};

template <class RO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RO::kWaves, 8))) void tick_fused_kernel(DevParams P, FusedArgs F) {
    __shared__ int item_sh[4];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int x0 = fused_xcc_id();
        FusedItem it{kFusedNone, 0, 0};
        int first = 0;
        for (int i = 0; i < kFusedQueues; ++i) {   // own XCD's queue first; the next one with items left when it is exhausted
            const int x = (x0 + i) & (kFusedQueues - 1);
            const int len = fused_queue_len(F.g, x);
            if (len == 0) continue;
            const uint32_t n = __hip_atomic_fetch_add(F.heads + x * kFusedHeadStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n < (uint32_t)len) { it = fused_decode(F.g, x, (int)n); first = (x == 0 && n == 0u); break; }
        }
        item_sh[0] = it.kind; item_sh[1] = it.col; item_sh[2] = it.idx; item_sh[3] = first;
    }
    __syncthreads();
    const int kind = item_sh[0], col = item_sh[1], idx = item_sh[2];
    if (item_sh[3])   // (uniform) the next launch's counters: nobody reads or writes the other parity during this launch
        for (int i = tid; i < F.zero_n; i += 256) F.zero_base[i] = 0u;
    if (kind == kFusedNone) return;
    const int a = col / F.g.NCH, ch = col - a * F.g.NCH;
    if (kind == kFusedRollout) {
        const int bx = ch * F.g.RB + idx;
        if (bx * RO::kBS < P.K)   // (uniform) the ragged last column of an agent has fewer blocks with samples: the others only arrive
            RO::run(P, F, bx, a, col == (F.g.n_cols >> 1) && idx == 0, F.prio_mode);
        // release: every wave's write-through stores have left, then ONE agent-scope add on the column's counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(F.done + col, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // update item: wait until the column's rollout items have all arrived (they hold smaller tickets of this queue)
    if (F.prio_mode & 1) __builtin_amdgcn_s_setprio(3);
    __syncthreads();   // (item_sh is re-used below)
    if (tid == 0) {
        int late = 0;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(F.done + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)F.g.RB) {
            if (F.timeout_ticks && wall_clock64() - t0 > F.timeout_ticks) { late = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        item_sh[0] = late;
        if (late) __hip_atomic_store(F.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (item_sh[0]) return;   // (uniform) the column never completed: the status word poisons the tick's outputs
    asm volatile("" ::: "memory");
    update_body<float, true, true>(P, static_cast<const float*>(nullptr), F.dP, F.stot, F.part, F.g.NCH, a, idx, ch, F.epart, F.seed, F.tick,
                                   nullptr, F.skip_light);
}

// wrap-safe "counter has reached target" for monotonic 32-bit counters
__device__ __forceinline__ bool reached(uint32_t v, uint32_t target) { return (int32_t)(v - target) >= 0; }
// Counters that thousands of workgroups add to are kept as 8 words, a 64-byte line apart, one per XCD queue: atomics on ONE
// address retire at ~12 ns each (6150 update items counting themselves on one word took 74 us of a 77-us tail launch,
// profiles/r4_update_stream_ab_v1.jsonl); the reader sums the eight.
__device__ __forceinline__ uint32_t load_sum8(const uint32_t* w) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < kFusedQueues; ++i) v += __hip_atomic_load(w + i * kFusedHeadStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
__device__ __forceinline__ bool poll_sum8_reached(const uint32_t* w, uint32_t target, unsigned long long timeout_ticks) {
    const unsigned long long t0 = wall_clock64();
    while (!((int32_t)(load_sum8(w) - target) >= 0)) {
        if (timeout_ticks && wall_clock64() - t0 > timeout_ticks) return false;
        __builtin_amdgcn_s_sleep(8);
    }
    return true;
}
// one lane polls a word until it reaches `target` (bounded by the deadline); returns false when it gave up
__device__ __forceinline__ bool poll_reached(const uint32_t* w, uint32_t target, unsigned long long timeout_ticks, int sleep) {
    const unsigned long long t0 = wall_clock64();
    while (!reached(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), target)) {
        if (timeout_ticks && wall_clock64() - t0 > timeout_ticks) return false;
        if (sleep > 8) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(8);
    }
    return true;
}

template <class RO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RO::kWaves, 8))) void rollout_arrive_kernel(DevParams P, FusedArgs F) {
    const int bx = (int)blockIdx.x, a = (int)blockIdx.y;
    if (bx == 0 && a == 0 && threadIdx.x == 0) {
        // the gate.  Everything before this launch on its stream is done; the update-stream workgroups of earlier ticks are done with
        // their items (the tail launch joined them) but may still be on their way out: wait until they have all counted themselves
        // out, re-arm the ticket heads, and only then let this tick's update workgroups in
        if (!poll_sum8_reached(F.exits, F.exit_target, F.timeout_ticks)) __hip_atomic_store(F.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < kFusedQueues; ++i) __hip_atomic_store(F.uheads + i * kFusedHeadStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(F.epoch_flag, F.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RO::run(P, F, bx, a, blockIdx.x == (gridDim.x >> 1) && a == 0, F.prio_mode);
    // release: every wave's write-through stores have left, then ONE agent-scope add on the column's counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(F.col_done + a * F.g.NCH + (bx * RO::kBS) / F.ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class RO>
static hipError_t rollout_arrive_go(const FusedLaunch& a) {
    dim3 grid((a.P.K + RO::kBS - 1) / RO::kBS, a.P.A);
    const unsigned lds = (unsigned)((size_t)a.P.T * RO::kLdsPerStep);
    auto kern = rollout_arrive_kernel<RO>;
    if (a.ev_start) hipExtLaunchKernelGGL(kern, grid, dim3(256), lds, a.stream, a.ev_start, a.ev_stop, 0, a.P, a.F);
    else hipLaunchKernelGGL(kern, grid, dim3(256), lds, a.stream, a.P, a.F);
    return hipGetLastError();
}

#ifdef MPPI_UPDATE_STREAM_TU
// Persistent update workgroups: take update tickets (queue x = this XCD's columns first, then the others'), wait for the
// ticket's column, run update_body on it, count the item.  <= 64 VGPRs, so that a workgroup fits next to four rollout waves.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void update_stream_kernel(DevParams P, FusedArgs F) {
    __shared__ int item_sh[4];
    const int tid = threadIdx.x;
    if (tid == 0) {
        // the gate: this tick's rollout launch has re-armed the ticket heads.  (The tail launch sits behind that launch on its
        // stream and finds the gate open; a head-start workgroup that the hardware starts after its tick is over finds the gate
        // PAST its tick number: its items were done by others, it only counts itself out.)
        int ok = poll_reached(F.epoch_flag, F.epoch, F.timeout_ticks, 32);
        if (!ok) __hip_atomic_store(F.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ok && __hip_atomic_load(F.epoch_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != F.epoch) ok = 0;
        item_sh[3] = ok;
        if (!ok) __hip_atomic_fetch_add(F.exits + fused_xcc_id() * kFusedHeadStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!item_sh[3]) return;
    if (F.prio_mode & 1) __builtin_amdgcn_s_setprio(3);
    const int x0 = fused_xcc_id();
    const int T = F.g.T;
    unsigned dead = 0u;   // (lane 0) queues this workgroup has found exhausted: not probed again
    for (;;) {
        __syncthreads();   // (item_sh: the previous round's readers are through)
        if (tid == 0) {
            int kind = kFusedNone, col = 0, t = 0;
            for (int i = 0; i < kFusedQueues; ++i) {
                const int x = (x0 + i) & (kFusedQueues - 1);
                if (dead & (1u << x)) continue;
                const int len = fused_queue_cols(F.g, x) * T;
                uint32_t n = (uint32_t)len;
                // (a look before the read-modify-write: exhausted queues are found by a load, which the L2 serves in parallel, instead
                // of one more serialised atomic on a word thousands of workgroups hit at the end of the launch)
                if (len && __hip_atomic_load(F.uheads + x * kFusedHeadStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)len)
                    n = __hip_atomic_fetch_add(F.uheads + x * kFusedHeadStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n < (uint32_t)len) { kind = kFusedUpdate; col = x + kFusedQueues * (int)(n / (uint32_t)T); t = (int)(n % (uint32_t)T); break; }
                dead |= 1u << x;
            }
            if (kind == kFusedUpdate) {   // the column's rollout workgroups: epoch * (their number), monotonic
                const int a = col / F.g.NCH, ch = col - a * F.g.NCH;
                const uint32_t target = F.epoch * (uint32_t)fused_col_blocks(P.K, ch, F.ch, F.bs);
                if (!poll_reached(F.col_done + col, target, F.timeout_ticks, 8)) {
                    __hip_atomic_store(F.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    kind = kFusedNone;
                }
            }
            item_sh[0] = kind; item_sh[1] = col; item_sh[2] = t;
        }
        __syncthreads();
        const int kind = item_sh[0], col = item_sh[1], t = item_sh[2];
        if (kind != kFusedUpdate) break;   // (uniform) no tickets left
        asm volatile("" ::: "memory");
        const int a = col / F.g.NCH, ch = col - a * F.g.NCH;
        update_body<float, true, true>(P, static_cast<const float*>(nullptr), F.dP, F.stot, F.part, F.g.NCH, a, t, ch, F.epart, F.seed, F.tick,
                                       nullptr, F.skip_light);
        // the tuple went out write-through: drained, then counted
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(F.items_done + (col & (kFusedQueues - 1)) * kFusedHeadStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        __hip_atomic_fetch_add(F.exits + x0 * kFusedHeadStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // done with the ticket heads for good
        if (F.join && blockIdx.x == 0)   // the tail launch ends only when the head-start launch's items are in as well
            if (!poll_sum8_reached(F.items_done, F.items_target, F.timeout_ticks)) __hip_atomic_store(F.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
hipError_t launch_update_stream(const UpdateStreamLaunch& a) {
    if (a.ev_start) hipExtLaunchKernelGGL(update_stream_kernel, dim3(a.blocks), dim3(256), 0, a.stream, a.ev_start, a.ev_stop, 0, a.P, a.F);
    else hipLaunchKernelGGL(update_stream_kernel, dim3(a.blocks), dim3(256), 0, a.stream, a.P, a.F);
    return hipGetLastError();
}
#endif  // MPPI_UPDATE_STREAM_TU

template <class RO>
static hipError_t tick_fused_go(const FusedLaunch& a) {
    const int items = a.F.g.n_cols * (a.F.g.RB + a.F.g.T);
    const unsigned lds = (unsigned)((size_t)a.P.T * RO::kLdsPerStep);
    auto kern = tick_fused_kernel<RO>;
    if (a.ev_start) hipExtLaunchKernelGGL(kern, dim3(items), dim3(256), lds, a.stream, a.ev_start, a.ev_stop, 0, a.P, a.F);
    else hipLaunchKernelGGL(kern, dim3(items), dim3(256), lds, a.stream, a.P, a.F);
    return hipGetLastError();
}
#endif  // MPPI_FUSED_TU

}  // namespace mppi
