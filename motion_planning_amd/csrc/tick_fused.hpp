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
