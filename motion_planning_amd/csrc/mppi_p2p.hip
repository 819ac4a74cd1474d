// mppi_p2p.hip -- the caller's peer-to-peer exchange of the shard tuples (include/mppi_hip.h mppi_p2p_*): mailboxes in fine-grained
// device memory, HIP IPC handles, the file rendezvous of ranks that share nothing but a file system, the self-test.  The kernels
// (publish, the self-test's consumer) are launched by the core (mppi_engine.hip: p2p_publish, launch_p2p_check).
#include "mppi_engine.hpp"

#include <sys/stat.h>

extern "C" {

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
    // (the pid only means something inside the writer's pid namespace -- ranks in separate containers see each other as pid 1, or not at
    // all: the file carries the namespace's inode too, and a reader in ANOTHER namespace skips the liveness test -- ADVICE r5)
    struct Head { char magic[8]; int32_t n_ranks, rank; uint64_t mbox_bytes; int64_t pid; uint64_t pid_ns; };
    struct stat ns_st{};
    const uint64_t my_ns = stat("/proc/self/ns/pid", &ns_st) == 0 ? (uint64_t)ns_st.st_ino : 0ull;
    auto head_of = [&](int r) { Head hd{}; std::memcpy(hd.magic, "MPPIMBX3", 8); hd.n_ranks = n_ranks; hd.rank = r; hd.mbox_bytes = h->p2p_bytes; hd.pid = (int64_t)getpid(); hd.pid_ns = my_ns; return hd; };
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
                const bool same_ns = my_ns != 0 && hd.pid_ns == my_ns;
                const bool writer_alive = !same_ns || (hd.pid > 0 && (kill((pid_t)hd.pid, 0) == 0 || errno == EPERM));
                if (got == sizeof(hd) + MPPI_IPC_HANDLE_BYTES && writer_alive) {
                    Head want = head_of(r);
                    want.pid = hd.pid; want.pid_ns = hd.pid_ns;
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
        h->launch_p2p_check(w, h->p2p_data(h->p2p_mbox, par, 0), (int)n, (int)slot_f64, d_got, d_status);
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

}  // extern "C"
