// Stress of gph::ThreadPool (gp_emulator_amd/csrc/gp_host_pool.hpp), built by
// tests/test_abi_cpu.py with -fsanitize=thread: every task of every job runs exactly once, a
// job's writes are visible to the caller when run() returns, run_on_worker really runs on a
// worker, and the pool shuts down cleanly with jobs of every size in between.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "gp_host_pool.hpp"

int main(int argc, char** argv) {
  const int n_threads = argc > 1 ? atoi(argv[1]) : 6;
  const int n_jobs = argc > 2 ? atoi(argv[2]) : 3000;
  unsigned long long total = 0, expect = 0;
  unsigned seed = 12345u;
  auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  {
    gph::ThreadPool pool(n_threads);
    if (pool.size() != (n_threads > 1 ? n_threads : 1)) { printf("size %d\n", pool.size()); return 2; }
    for (int j = 0; j < n_jobs; ++j) {
      const int n = 1 + (int)(rnd() % 24);
      std::vector<int> hits(n, 0);                 // plain ints: the pool's own ordering must make this race-free
      std::vector<unsigned long long> val(n, 0);
      pool.run(n, [&](int t) {
        hits[t] += 1;
        unsigned long long s = 0;
        for (int k = 0; k <= (t * 37 + j) % 200; ++k) s += (unsigned long long)k * (t + 1);
        val[t] = s;
      });
      for (int t = 0; t < n; ++t) {
        if (hits[t] != 1) { printf("job %d task %d ran %d times\n", j, t, hits[t]); return 3; }
        unsigned long long s = 0;
        for (int k = 0; k <= (t * 37 + j) % 200; ++k) s += (unsigned long long)k * (t + 1);
        expect += s;
        total += val[t];
      }
      if (j % 7 == 0) {
        std::thread::id where;
        int ran = 0;
        pool.run_on_worker([&] { where = std::this_thread::get_id(); ran += 1; });
        if (ran != 1) { printf("run_on_worker ran %d times\n", ran); return 4; }
        if (n_threads > 1 && where == std::this_thread::get_id()) { printf("run_on_worker ran on the caller\n"); return 5; }
      }
    }
  }   // ~ThreadPool: joins
  if (total != expect) { printf("sum mismatch\n"); return 6; }
  printf("ok %d jobs, %d threads\n", n_jobs, n_threads);
  return 0;
}
