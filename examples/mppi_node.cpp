// mppi_node -- ROS-less C++ caller of libmppi_hip.so (include/mppi_hip.h).
//
// The shape of the reference's MPPI node shell (moribots/motion_planning, `Controller`,
// control/src/mppi:296-389) in a compiled host program: one pose in per odometry message, one
// (vx, wz) twist out, waypoint cycling / parallel parking in between.  Everything heavy happens
// behind the C ABI; this file needs no HIP, no torch, no Python: g++ and the header are enough,
// which is what a C++ ROS node binding the library would look like.
//
//   make -C examples            (g++ only; links motion_planning_amd/lib/libmppi_hip.so)
//   build/mppi_node --task pentagon --samples 4096 --horizon 50 --callbacks 80 --seed 3
//
// K sharded over the GPUs of a node (split K, one exchange of the [A][T][8] tuples per tick), still without Python:
//   --handles G   ONE process drives G engines (device g where the node has that many, device 0 otherwise), samples split
//                 G ways, mailboxes connected by pointer (mppi_p2p_connect local_ptrs; peer access across devices is enabled
//                 by the library): mppi_tick_begin on all -> mppi_p2p_publish on all -> mppi_tick_finish_p2p on all;
//   --procs G     G processes (forked before anything touches the GPU), one engine each (device rank where the node has that many
//                 GPUs), the IPC handles exchanged through files in a fresh temporary directory (mppi_p2p_rendezvous); every rank
//                 runs the same node shell on the same plant, rank 0 prints.
//   --exchange auto|p2p|rccl   (with --procs) how the [A][T][8] tuples cross ranks: p2p = the engine's own mailboxes; rccl = ONE
//                 ncclAllGather per tick between mppi_tick_begin and mppi_tick_finish (include/mppi_hip.h: the caller's own
//                 exchange; librccl linked directly, the unique id passed through a file of the same directory); auto = p2p when
//                 EVERY rank's rendezvous and mailbox self-test pass, else rccl -- the ranks agree through status files, and rank
//                 0 says on stderr which exchange runs and why ("exchange: ...").  --force-p2p-fail makes this rank report a failed
//                 p2p set-up (tests of the fallback).  --procs 1 --exchange rccl runs the whole sharded call sequence in a world of one.
// Either way the twists equal the one-handle node's to the split-invariance bound of the tuple merge (sample ids are global).
//
// Output: one line per odometry callback,
//   i  x y theta  gx gy gtheta  ul ur  vx wz  idx done init
// (the columns of tests/golden ctl_* rows).  The plant between callbacks is the reference's own
// rk4 of the diff-drive model (control/src/mppi:23-30, :39-54), integrated on the host.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <thread>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "mppi_hip.h"

namespace {

struct Pose { double x, y, th; };

const double kPi = 3.14159265358979323846;

// control/src/mppi:23-30
void dd_dynamics(const mppi_config& c, const Pose& p, const double u[2], double out[3]) {
    out[0] = c.wheel_radius / 2.0 * std::cos(p.th) * (u[0] + u[1]);
    out[1] = c.wheel_radius / 2.0 * std::sin(p.th) * (u[0] + u[1]);
    out[2] = c.wheel_radius / c.wheel_base * (u[1] - u[0]);
}

// control/src/mppi:39-54 (k_i scaled by dt, heading wrapped into (-pi, pi])
Pose rk4(const mppi_config& c, const Pose& p, const double u[2], double dt) {
    double k1[3], k2[3], k3[3], k4[3];
    dd_dynamics(c, p, u, k1);
    for (double& v : k1) v *= dt;
    dd_dynamics(c, Pose{p.x + k1[0] / 2, p.y + k1[1] / 2, p.th + k1[2] / 2}, u, k2);
    for (double& v : k2) v *= dt;
    dd_dynamics(c, Pose{p.x + k2[0] / 2, p.y + k2[1] / 2, p.th + k2[2] / 2}, u, k3);
    for (double& v : k3) v *= dt;
    dd_dynamics(c, Pose{p.x + k3[0], p.y + k3[1], p.th + k3[2]}, u, k4);
    for (double& v : k4) v *= dt;
    Pose n{p.x + (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0]) / 6.0,
           p.y + (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1]) / 6.0,
           p.th + (k1[2] + 2 * k2[2] + 2 * k3[2] + k4[2]) / 6.0};
    n.th -= (std::ceil((n.th + kPi) / (2 * kPi)) - 1.0) * 2 * kPi;
    return n;
}

// tf.transformations.euler_from_quaternion(q)[2] (control/src/mppi:330-335)
double yaw_from_quaternion(double qx, double qy, double qz, double qw) {
    return std::atan2(2.0 * (qw * qz + qx * qy), 1.0 - 2.0 * (qy * qy + qz * qz));
}

#define MPPI_CALL(call)                                                                  \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != 0) {                                                                \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mppi_last_error(eng_err_)); \
            std::exit(2);                                                              \
        }                                                                              \
    } while (0)

// The node shell.  `publish` is stdout.
class Controller {
public:
    // handles > 1: this process drives that many engines, the samples split between them.  n_ranks > 1: this process is rank
    // `rank` of a group of processes, one engine each, IPC handles exchanged through files under `rendezvous`.
    // exchange: 0 auto, 1 p2p, 2 rccl (ranks that are processes only)
    Controller(const mppi_config& cfg, std::vector<std::pair<double, double>> waypoints, double thresh, uint64_t seed, int handles = 1,
               int n_ranks = 1, int rank = 0, const std::string& rendezvous = "", int exchange = 0, bool force_p2p_fail = false)
        : cfg_(cfg), waypoints_(std::move(waypoints)), thresh_(thresh), seed_(seed), n_ranks_(n_ranks), rank_(rank) {
        const bool world_of_one = n_ranks == 1 && handles == 1 && exchange == 2;   // the sharded call sequence with one rank
        const int G = n_ranks > 1 ? n_ranks : handles;
        const int first = n_ranks > 1 ? rank : 0, count = n_ranks > 1 ? 1 : handles;
        for (int g = first; g < first + count; ++g) {
            mppi_config c = cfg_;
            const int base = cfg_.samples / G, rem = cfg_.samples % G;           // contiguous, balanced split of the global sample range
            c.samples = base + (g < rem ? 1 : 0);
            c.sample_offset = cfg_.sample_offset + (uint32_t)(g * base + (g < rem ? g : rem));
            c.co_shards = G > 1 ? 1 : cfg_.co_shards;
            // every shard is told the whole controller's size: all of them pick the kernels the unsplit controller runs, and any number
            // of handles / ranks ends every tick with the same controls to rounding (mppi_config.samples_total)
            if (G > 1) c.samples_total = (int64_t)cfg_.sample_offset + cfg_.samples;
            mppi_engine* e = nullptr;
            if (handles > 1 || n_ranks > 1) c.device = cfg_.device + g;            // one GPU per handle / rank where the node has them ...
            int rc = mppi_create(&c, &e);
            if (rc == MPPI_E_INVALID && c.device != cfg_.device) {                 // ... all on the one device otherwise
                c.device = cfg_.device;
                rc = mppi_create(&c, &e);
            }
            if (rc != 0) {
                std::fprintf(stderr, "mppi_create: %s\n", mppi_last_error(nullptr));
                std::exit(2);
            }
            engs_.push_back(e);
            dev_of_engine_ = c.device;
        }
        eng_ = eng_err_ = engs_[0];
        if (n_ranks > 1 || world_of_one) {
            // p2p unless told otherwise; every rank must take the SAME exchange: each leaves its verdict in a status file and reads the others'
            std::string why = exchange == 2 ? "asked for by name" : "";
            bool p2p_ok = exchange != 2;
            if (p2p_ok) {
                int rc = force_p2p_fail ? MPPI_E_INTERNAL : mppi_p2p_rendezvous(eng_, rendezvous.c_str(), n_ranks, rank, 20000);
                if (rc == 0) rc = mppi_p2p_selftest(eng_, 2);
                if (rc != 0) { p2p_ok = false; why = force_p2p_fail ? "p2p set-up failure forced (--force-p2p-fail)" : std::string("p2p set-up failed: ") + mppi_last_error(eng_); }
                if (n_ranks > 1 && !agree(rendezvous, p2p_ok)) { p2p_ok = false; if (why.empty()) why = "another rank's p2p set-up failed"; }
                if (!p2p_ok && exchange == 1) { std::fprintf(stderr, "exchange: p2p asked for by name, and %s\n", why.c_str()); std::exit(2); }
                if (!p2p_ok) (void)mppi_p2p_destroy(eng_);
            }
            if (!p2p_ok) rccl_init(rendezvous, why);
            if (rank == 0) std::fprintf(stderr, "exchange: %s%s%s\n", p2p_ok ? "p2p" : "rccl", why.empty() ? "" : " -- ", why.c_str());
        } else if (handles > 1) {
            std::vector<void*> ptrs(handles, nullptr);
            for (int g = 0; g < handles; ++g) {
                eng_err_ = engs_[g];
                MPPI_CALL(mppi_p2p_create(engs_[g], handles, g, nullptr));
                MPPI_CALL(mppi_p2p_mailbox_ptr(engs_[g], &ptrs[g]));
            }
            for (int g = 0; g < handles; ++g) {
                eng_err_ = engs_[g];
                MPPI_CALL(mppi_p2p_connect(engs_[g], nullptr, ptrs.data()));
            }
            eng_err_ = eng_;
        }
        sharded_ = G > 1 || world_of_one;
        parallel_park_ = waypoints_.empty();  // :305-309
    }
    ~Controller() {
        for (mppi_engine* e : engs_) mppi_synchronize(e);   // nobody unmaps a mailbox a peer's kernel may still be writing to
        if (comm_) ncclCommDestroy(comm_);
        if (gathered_) (void)hipFree(gathered_);
        for (mppi_engine* e : engs_) mppi_destroy(e);
    }

    // one odometry message -> one twist (control/src/mppi:327-389)
    void odom_cb(double px, double py, double qx, double qy, double qz, double qw, double twist[2]) {
        start_ = Pose{px, py, yaw_from_quaternion(qx, qy, qz, qw)};
        if (parallel_park_) goal_ = Pose{0.0, -1.0, 0.0};  // :336-337
        const bool far = std::hypot(start_.x - goal_.x, start_.y - goal_.y) > thresh_;
        if (far && !init_) {  // :339-343  MPPI.get_path, :85-102
            const double s[3] = {start_.x, start_.y, start_.th}, g[3] = {goal_.x, goal_.y, goal_.th};
            double nxt[3];
            if (!sharded_) {
                MPPI_CALL(mppi_tick(eng_, s, g, MPPI_NOISE_PHILOX, seed_, tick_++, nxt, u_last_));
            } else {
                // K sharded: every engine rolls its samples out and reduces them to one tuple per timestep; ALL publishes are enqueued
                // before any finalize that waits for them (one thread drives the engines of this process, include/mppi_hip.h)
                for (mppi_engine* e : engs_) { eng_err_ = e; MPPI_CALL(mppi_tick_begin(e, s, g, MPPI_NOISE_PHILOX, seed_, tick_)); }
                if (comm_) {
                    // the caller's own exchange (include/mppi_hip.h, mppi_tick_begin ... mppi_tick_finish): ONE all-gather of this rank's
                    // [A][T][8] tuples on the engine's stream, every rank then merges the same n_ranks blocks
                    void* part = nullptr; size_t bytes = 0; void* st = nullptr;
                    MPPI_CALL(mppi_partials_ptr(eng_, &part, &bytes));
                    MPPI_CALL(mppi_get_stream(eng_, &st));
                    const ncclResult_t nr = ncclAllGather(part, gathered_, bytes / sizeof(double), ncclDouble, comm_, static_cast<hipStream_t>(st));
                    if (nr != ncclSuccess) { std::fprintf(stderr, "ncclAllGather: %s\n", ncclGetErrorString(nr)); std::exit(2); }
                    MPPI_CALL(mppi_tick_finish(eng_, gathered_, n_ranks_));
                } else {
                    for (mppi_engine* e : engs_) { eng_err_ = e; MPPI_CALL(mppi_p2p_publish(e)); }
                    for (mppi_engine* e : engs_) { eng_err_ = e; MPPI_CALL(mppi_tick_finish_p2p(e)); }
                }
                eng_err_ = eng_;
                MPPI_CALL(mppi_get_outputs(eng_, nxt, u_last_));
                for (size_t i = 1; i < engs_.size(); ++i) {   // every engine finishes every tick identically (the merge is one formula on the same tuples)
                    double n2[3], u2[2];
                    eng_err_ = engs_[i];
                    MPPI_CALL(mppi_get_outputs(engs_[i], n2, u2));
                    if (u2[0] != u_last_[0] || u2[1] != u_last_[1] || n2[0] != nxt[0] || n2[1] != nxt[1] || n2[2] != nxt[2]) {
                        std::fprintf(stderr, "engine %zu finished tick %u differently from engine 0\n", i, tick_);
                        std::exit(3);
                    }
                }
                eng_err_ = eng_;
                ++tick_;
            }
            done_ = false;
        } else if (init_) {  // :344-355
            initialize();
            if (!parallel_park_) goal_from_waypoint();
            init_ = false;
        } else {  // :356-375
            if (!parallel_park_) {
                idx_ = idx_ + 1 >= waypoints_.size() ? 0 : idx_ + 1;
                initialize();
                goal_from_waypoint();
            } else {
                done_ = true;
            }
        }
        const double ul = done_ ? 0.0 : u_last_[0], ur = done_ ? 0.0 : u_last_[1];  // :377-381
        twist[0] = cfg_.wheel_radius * (ul + ur) / 2.0;                             // wheelsToTwist :319-325
        twist[1] = cfg_.wheel_radius * (ur - ul) / cfg_.wheel_base;
    }

    const Pose& start() const { return start_; }
    const Pose& goal() const { return goal_; }
    void applied(double u[2]) const { u[0] = done_ ? 0.0 : u_last_[0]; u[1] = done_ ? 0.0 : u_last_[1]; }
    size_t idx() const { return idx_; }
    bool done() const { return done_; }
    bool init() const { return init_; }
    double dt() const { return cfg_.dt; }
    const mppi_config& cfg() const { return cfg_; }

private:
    // every rank writes "<prefix>.ok.<rank>" = its verdict and reads the others': true when ALL said yes (a rank that never answers counts as no)
    bool agree(const std::string& prefix, bool mine) {
        auto name = [&](int r) { return prefix + ".ok." + std::to_string(r); };
        if (FILE* f = std::fopen((name(rank_) + ".tmp").c_str(), "w")) { std::fputc(mine ? '1' : '0', f); std::fclose(f); std::rename((name(rank_) + ".tmp").c_str(), name(rank_).c_str()); }
        bool all = mine;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < n_ranks_; ++r) {
            if (r == rank_) continue;
            for (;;) {
                if (FILE* f = std::fopen(name(r).c_str(), "r")) { const int c = std::fgetc(f); std::fclose(f); if (c == '0' || c == '1') { all = all && c == '1'; break; } }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) { all = false; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        return all;
    }
    // RCCL communicator of the ranks: rank 0's unique id travels through "<prefix>.ncclid"
    void rccl_init(const std::string& prefix, const std::string& why) {
        ncclUniqueId id;
        const std::string path = prefix + ".ncclid";
        if (rank_ == 0) {
            ncclResult_t r = ncclGetUniqueId(&id);
            if (r != ncclSuccess) { std::fprintf(stderr, "exchange: rccl unavailable (%s); p2p: %s\n", ncclGetErrorString(r), why.c_str()); std::exit(2); }
            if (n_ranks_ > 1) {
                FILE* f = std::fopen((path + ".tmp").c_str(), "wb");
                if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1 || std::fclose(f) != 0 || std::rename((path + ".tmp").c_str(), path.c_str()) != 0) {
                    std::fprintf(stderr, "exchange: cannot publish the RCCL id at %s\n", path.c_str()); std::exit(2);
                }
            }
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                if (FILE* f = std::fopen(path.c_str(), "rb")) { const size_t got = std::fread(&id, sizeof(id), 1, f); std::fclose(f); if (got == 1) break; }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) { std::fprintf(stderr, "exchange: rank 0's RCCL id did not appear\n"); std::exit(2); }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        // the communicator lives on the device this rank's engine was created on (the engine restores the caller's device after every call)
        (void)hipSetDevice(dev_of_engine_);
        ncclResult_t r = ncclCommInitRank(&comm_, n_ranks_, id, rank_);
        if (r != ncclSuccess) {
            // (two ranks on ONE GPU is the case a one-GPU test box produces: RCCL refuses it by design)
            std::fprintf(stderr, "exchange: rccl communicator of %d ranks failed: %s (%s); p2p: %s\n", n_ranks_, ncclGetErrorString(r), ncclGetLastError(nullptr), why.c_str());
            std::exit(5);
        }
        void* part = nullptr; size_t bytes = 0;
        MPPI_CALL(mppi_partials_ptr(eng_, &part, &bytes));
        if (hipMalloc(&gathered_, bytes * (size_t)n_ranks_) != hipSuccess) { std::fprintf(stderr, "exchange: hipMalloc of the gather buffer failed\n"); std::exit(2); }
    }

    void initialize() {  // MPPI.initialize, :79-83
        for (mppi_engine* e : engs_) { eng_err_ = e; MPPI_CALL(mppi_reset(e, -1)); }
        eng_err_ = eng_;
        u_last_[0] = u_last_[1] = 0.0;
    }
    void goal_from_waypoint() {  // :347-352
        const auto& w = waypoints_[idx_];
        goal_ = Pose{w.first, w.second, std::atan2(w.second - start_.y, w.first - start_.x)};
    }

    mppi_config cfg_;
    ncclComm_t comm_ = nullptr;        // non-null: the tuples cross ranks by ncclAllGather (else the engine's p2p mailboxes)
    void* gathered_ = nullptr;         // [n_ranks][A][T][8] float64 on this rank's device
    int dev_of_engine_ = 0;
    mppi_engine* eng_ = nullptr;       // engine 0: where the outputs are read
    mppi_engine* eng_err_ = nullptr;   // the engine of the call in flight (error messages)
    std::vector<mppi_engine*> engs_;
    bool sharded_ = false;
    std::vector<std::pair<double, double>> waypoints_;
    double thresh_;
    uint64_t seed_;
    int n_ranks_ = 1, rank_ = 0;
    uint32_t tick_ = 0;
    bool parallel_park_ = true, init_ = true, done_ = false;
    size_t idx_ = 0;
    Pose start_{0, 0, 0}, goal_{0, 0, 0};
    double u_last_[2] = {0.0, 0.0};
};

}  // namespace

int main(int argc, char** argv) {
    mppi_config cfg;
    mppi_default_config(&cfg);
    std::string task = "park";
    int callbacks = 40, handles = 1, procs = 1, exchange = 0;
    bool force_p2p_fail = false;
    double thresh = 0.05;
    uint64_t seed = 0;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](const char* name) -> const char* {
            if (std::strcmp(argv[i], name) != 0) return nullptr;
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(1); }
            return argv[++i];
        };
        if (const char* v = val("--task")) task = v;
        else if (const char* v = val("--samples")) cfg.samples = std::atoi(v);
        else if (const char* v = val("--horizon")) cfg.horizon = std::atoi(v);
        else if (const char* v = val("--callbacks")) callbacks = std::atoi(v);
        else if (const char* v = val("--thresh")) thresh = std::atof(v);
        else if (const char* v = val("--seed")) seed = std::strtoull(v, nullptr, 10);
        else if (const char* v = val("--storage")) cfg.storage = std::strcmp(v, "f64") == 0 ? MPPI_STORE_F64 : MPPI_STORE_F32;
        else if (const char* v = val("--device")) cfg.device = std::atoi(v);
        else if (const char* v = val("--handles")) handles = std::atoi(v);
        else if (const char* v = val("--procs")) procs = std::atoi(v);
        else if (const char* v = val("--exchange")) exchange = std::strcmp(v, "p2p") == 0 ? 1 : std::strcmp(v, "rccl") == 0 ? 2 : 0;
        else if (std::strcmp(argv[i], "--force-p2p-fail") == 0) force_p2p_fail = true;
        else if (const char* v = val("--tick-path"))
            cfg.tick_path = std::strcmp(v, "lanes") == 0 ? MPPI_TICK_LANES : std::strcmp(v, "scan") == 0 ? MPPI_TICK_SCAN : MPPI_TICK_AUTO;
        else {
            std::fprintf(stderr, "usage: mppi_node [--task park|pentagon] [--samples K] [--horizon T] [--callbacks N]\n"
                                 "                 [--thresh m] [--seed s] [--storage f32|f64] [--device d]\n"
                                 "                 [--tick-path auto|lanes|scan] [--handles G | --procs G [--exchange auto|p2p|rccl] [--force-p2p-fail]]\n");
            return 1;
        }
    }
    cfg.dt = 1.0 / cfg.horizon;  // control/src/mppi:67
    if (handles < 1 || handles > 8 || procs < 1 || procs > 8 || (handles > 1 && procs > 1) || cfg.samples < std::max(handles, procs)) {
        std::fprintf(stderr, "--handles / --procs: 1..8, one of the two, at least one sample each\n");
        return 1;
    }
    // --procs G: fork the other ranks BEFORE anything touches the GPU; they meet through files in a fresh directory
    int rank = 0;
    std::string rendezvous;
    std::vector<pid_t> children;
    if (exchange == 2 && handles > 1) { std::fprintf(stderr, "--exchange rccl serves ranks that are processes (--procs)\n"); return 1; }
    if (procs > 1 || exchange == 2) {
        char dir[] = "/tmp/mppi_node_XXXXXX";
        if (!mkdtemp(dir)) { std::perror("mkdtemp"); return 1; }
        rendezvous = std::string(dir) + "/mbox";
        for (int r = 1; r < procs; ++r) {
            const pid_t pid = fork();
            if (pid < 0) { std::perror("fork"); return 1; }
            if (pid == 0) { rank = r; children.clear(); if (!std::freopen("/dev/null", "w", stdout)) return 1; break; }
            children.push_back(pid);
        }
    }
    std::vector<std::pair<double, double>> wp;
    if (task == "pentagon")  // the `waypoints` parameter of the node (control/config/waypoints.yaml)
        wp = {{1.0, 0.0}, {2.0, 1.0}, {1.0, 2.0}, {0.0, 2.0}, {0.0, 0.0}};
    else if (task != "park") { std::fprintf(stderr, "unknown task %s\n", task.c_str()); return 1; }

    // MPPI_NODE_TRACE=<file>: a progress word in a memory-mapped file (no system call per update), so that
    // a watchdog can tell where a run stopped: -1 creating, -2 created, i >= 0 callback i done, -3 destroying,
    // -4 destroyed
    volatile int* progress = nullptr;
    if (const char* path = std::getenv("MPPI_NODE_TRACE")) {
        const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd >= 0 && ftruncate(fd, 64) == 0) {
            void* m = mmap(nullptr, 64, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) progress = static_cast<volatile int*>(m);
        }
        if (fd >= 0) close(fd);
    }
    const bool trace = progress != nullptr;
    if (trace) *progress = -1;
    {
    Controller node(cfg, wp, thresh, seed, handles, procs, rank, rendezvous, exchange, force_p2p_fail);
    if (trace) *progress = -2;
    Pose plant{0.0, 0.0, 0.0};
    for (int i = 0; i < callbacks; ++i) {
        double twist[2], u[2];
        node.odom_cb(plant.x, plant.y, 0.0, 0.0, std::sin(plant.th / 2.0), std::cos(plant.th / 2.0), twist);
        node.applied(u);
        std::printf("%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %zu %d %d\n", i,
                    node.start().x, node.start().y, node.start().th, node.goal().x, node.goal().y, node.goal().th,
                    u[0], u[1], twist[0], twist[1], node.idx(), (int)node.done(), (int)node.init());
        plant = rk4(node.cfg(), plant, u, node.dt());
        if (trace) *progress = i;
    }
    if (trace) *progress = -3;
    }
    if (trace) *progress = -4;
    int status = 0;
    for (pid_t pid : children) {   // rank 0 collects the other ranks, then removes the rendezvous files
        int st = 0;
        if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) status = 4;
    }
    if (!rendezvous.empty() && rank == 0) {
        for (int r = 0; r < procs; ++r) { std::remove((rendezvous + "." + std::to_string(r)).c_str()); std::remove((rendezvous + ".ok." + std::to_string(r)).c_str()); }
        std::remove((rendezvous + ".ncclid").c_str());
        rmdir(rendezvous.substr(0, rendezvous.rfind('/')).c_str());
    }
    return status;
}
