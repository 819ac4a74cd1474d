// tools/node_tail.cpp -- where does the p99 of the node's blocking call come from?  (task 8)
//   g++ -O2 -std=c++17 -Iinclude tools/node_tail.cpp -o tools/node_tail -Lmotion_planning_amd/lib -lmppi_hip -Wl,-rpath,$PWD/motion_planning_amd/lib
//   tools/node_tail [K] [T] [calls] [gap_us] [co_shards: 0 auto | 1 one engine | 2..8] [storage: 0 f32 | 1 f64]
// The stock node's call (mppi_tick: host state in, blocking, host controls out) N times from plain C++ -- no Python in the
// loop --, every call's wall time kept: percentiles, a histogram, and for the slow calls their positions (periodic? bursts?)
// and the time the host spent INSIDE the enqueue part vs waiting for the device (mppi_tick with NULL outputs, then
// mppi_get_outputs).  gap_us > 0 sleeps between calls like a real odometry stream does.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "mppi_hip.h"

int main(int argc, char** argv) {
    const int K = argc > 1 ? std::atoi(argv[1]) : 10, T = argc > 2 ? std::atoi(argv[2]) : 100;
    const int N = argc > 3 ? std::atoi(argv[3]) : 5000, gap_us = argc > 4 ? std::atoi(argv[4]) : 0;
    mppi_config cfg;
    mppi_default_config(&cfg);
    cfg.samples = K; cfg.horizon = T;
    if (argc > 5) cfg.co_shards = std::atoi(argv[5]);
    if (argc > 6) cfg.storage = std::atoi(argv[6]) ? MPPI_STORE_F64 : MPPI_STORE_F32;
    mppi_engine* h = nullptr;
    if (mppi_create(&cfg, &h)) { std::fprintf(stderr, "create: %s\n", mppi_last_error(nullptr)); return 1; }
    double st[3] = {0, 0, 0}, goal[3] = {0, -1, 0}, nxt[3], ua[2];
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int i = 0; i < 200; ++i) { if (mppi_tick(h, st, goal, MPPI_NOISE_PHILOX, 0, i, nxt, ua)) { std::fprintf(stderr, "%s\n", mppi_last_error(h)); return 1; } for (int j = 0; j < 3; ++j) st[j] = nxt[j]; }
    std::vector<double> total(N), enq(N), wait(N);
    for (int i = 0; i < N; ++i) {
        if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
        const auto t0 = clk::now();
        mppi_tick(h, st, goal, MPPI_NOISE_PHILOX, 0, 1000 + i, nullptr, nullptr);   // enqueue only
        const auto t1 = clk::now();
        mppi_get_outputs(h, nxt, ua);                                                // wait for the device
        const auto t2 = clk::now();
        total[i] = us(t0, t2); enq[i] = us(t0, t1); wait[i] = us(t1, t2);
        for (int j = 0; j < 3; ++j) st[j] = nxt[j];
    }
    auto pct = [&](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[std::min<size_t>(v.size() - 1, (size_t)(p * v.size()))]; };
    std::printf("K=%d T=%d co_shards=%d storage=%s calls=%d gap=%d us: total median %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f | enqueue median %.1f p99 %.1f | device wait median %.1f p99 %.1f\n",
                K, T, cfg.co_shards, cfg.storage == MPPI_STORE_F64 ? "f64" : "f32", N, gap_us, pct(total, .5), pct(total, .9), pct(total, .99), pct(total, .999), pct(total, 1.0), pct(enq, .5), pct(enq, .99), pct(wait, .5), pct(wait, .99));
    const double med = pct(total, .5);
    int bins[16] = {0};
    for (double v : total) bins[std::min(15, (int)(v / 5.0))]++;
    std::printf("histogram (5 us bins from 0; last = 75+):");
    for (int b = 0; b < 16; ++b) std::printf(" %d", bins[b]);
    std::printf("\nslow calls (> 1.5 x median): index : total = enqueue + wait\n");
    int shown = 0, last = -1;
    for (int i = 0; i < N && shown < 40; ++i)
        if (total[i] > 1.5 * med) { std::printf("  %d (+%d): %.1f = %.1f + %.1f\n", i, last < 0 ? 0 : i - last, total[i], enq[i], wait[i]); last = i; ++shown; }
    mppi_destroy(h);
    return 0;
}
