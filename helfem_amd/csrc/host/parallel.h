// Tiny std::thread parallel-for used by the setup code (the reference uses OpenMP there,
// e.g. basis.cpp:1175-1178); no dependency on an OpenMP runtime.
#pragma once
#include <atomic>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

namespace helfem {

inline int host_threads() {
  const char *e = getenv("HELFEM_NUM_THREADS");
  if (e && atoi(e) > 0) return atoi(e);
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

inline void parallel_for(size_t n, const std::function<void(size_t)> &fn, int nthreads = 0) {
  if (nthreads <= 0) nthreads = host_threads();
  if ((size_t)nthreads > n) nthreads = (int)n;
  if (nthreads <= 1) {
    for (size_t i = 0; i < n; i++) fn(i);
    return;
  }
  std::atomic<size_t> next(0);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&]() {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= n) break;
        fn(i);
      }
    });
  for (auto &t : th) t.join();
}

}  // namespace helfem
