// Tiny std::thread parallel-for used by the setup code (the reference uses OpenMP there,
// e.g. basis.cpp:1175-1178); no dependency on an OpenMP runtime.
#pragma once
#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
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
  // An exception thrown by fn on a worker thread (std::logic_error of the special-function code, say) must not reach
  // the thread's top level (std::terminate): the first one is kept, the remaining work is abandoned, and it is rethrown
  // on the calling thread after the join -- so that the C ABI still turns it into a status code.
  std::atomic<size_t> next(0);
  std::atomic<bool> failed(false);
  std::exception_ptr first;
  std::mutex mtx;
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&]() {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= n || failed.load()) break;
        try {
          fn(i);
        } catch (...) {
          std::lock_guard<std::mutex> lock(mtx);
          if (!first) first = std::current_exception();
          failed.store(true);
        }
      }
    });
  for (auto &t : th) t.join();
  if (first) std::rethrow_exception(first);
}

}  // namespace helfem
