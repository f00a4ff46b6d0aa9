// workers.hpp -- what the host-side stages (tail.cpp, fastq.cpp) share: the error type that
// crosses into the C ABI as a status + message, a persistent worker pool, and the number of
// CPUs the process may really use.
#ifndef KSLAM_HOST_WORKERS_HPP_
#define KSLAM_HOST_WORKERS_HPP_
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#if defined(__linux__)
#include <sys/mman.h>
#endif

#include "../../include/kslam.h"

namespace kslam_host {

struct HostError {
  kslam_status code;
  std::string msg;
};
[[noreturn]] inline void fail(kslam_status c, const std::string &m) { throw HostError{c, m}; }

inline thread_local std::string g_err;

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

// ---------------------------------------------------------------- worker pool --
// Persistent workers, woken per parallel region; the caller is worker 0.
class Pool {
 public:
  static Pool &get() {
    static Pool *p = new Pool();  // never destroyed: workers are detached
    return *p;
  }
  void run(int n, const std::function<void(int)> &f) {
    if (n <= 1) {
      f(0);
      return;
    }
    std::lock_guard<std::mutex> region(region_);
    {
      std::unique_lock<std::mutex> lk(m_);
      while ((int)workers_ < n - 1) {
        int id = workers_++;
        std::thread([this, id] { loop(id); }).detach();
      }
      job_ = &f;
      want_ = n - 1;
      active_ = n - 1;
      failed_ = false;
      gen_++;
    }
    start_.notify_all();
    try {
      f(0);
    } catch (const HostError &e) {
      note(e);
    } catch (const std::exception &e) {
      note(HostError{KSLAM_ERR_INTERNAL, e.what()});
    }
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return active_ == 0; });
    job_ = nullptr;
    if (failed_) throw error_;
  }
  // dynamic schedule of n_tasks over n_threads
  void tasks(int n_threads, size_t n_tasks, const std::function<void(size_t)> &f) {
    std::atomic<size_t> next(0);
    run((int)std::min<size_t>(n_threads, std::max<size_t>(n_tasks, 1)), [&](int) {
      for (;;) {
        size_t t = next.fetch_add(1, std::memory_order_relaxed);
        if (t >= n_tasks) break;
        f(t);
      }
    });
  }

 private:
  void note(const HostError &e) {
    std::lock_guard<std::mutex> lk(m_);
    if (!failed_) {
      failed_ = true;
      error_ = e;
    }
  }
  void loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)> *job;
      {
        std::unique_lock<std::mutex> lk(m_);
        start_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (id >= want_) continue;
        job = job_;
      }
      try {
        (*job)(id + 1);
      } catch (const HostError &e) {
        note(e);
      } catch (const std::exception &e) {
        note(HostError{KSLAM_ERR_INTERNAL, e.what()});
      }
      std::lock_guard<std::mutex> lk(m_);
      if (--active_ == 0) done_.notify_one();
    }
  }
  std::mutex region_, m_;
  std::condition_variable start_, done_;
  const std::function<void(int)> *job_ = nullptr;
  uint64_t gen_ = 0;
  int want_ = 0, active_ = 0;
  size_t workers_ = 0;
  bool failed_ = false;
  HostError error_;
};

// Asks for transparent huge pages under a large heap block (the boxes run THP in "madvise" mode).
// The host stages walk multi-hundred-megabyte arrays at random -- overlap records, alignment pairs,
// the genome columns -- and with 4 KiB pages nearly every such access also misses the TLB.  Must be
// called before the block is first touched to take effect at once; harmless otherwise.
// Allocator of the big column blocks the host stages hand out (FASTQ columns).  Plain malloc + huge-page
// advice by default; once a GPU context exists the HIP side of the library installs page-locked
// allocation (kslam_api.hip: pinned_alloc), so that those columns go to the device by DMA straight from
// where the parser wrote them (kslam_submit_batch_columns) instead of through a gather copy.
struct BigAlloc {
  void *(*alloc)(size_t) = nullptr;   // nullptr: malloc
  void (*release)(void *, size_t) = nullptr;
};
// One pointer to an immutable struct, so that a parser thread reading the hook while a context is being
// created or destroyed sees either no hook or a complete one (never `alloc` without `release`).
inline std::atomic<const BigAlloc *> &big_alloc_hook() {
  static std::atomic<const BigAlloc *> h{nullptr};
  return h;
}

inline void advise_huge(void *p, size_t bytes) {
#if defined(__linux__) && defined(MADV_HUGEPAGE)
  constexpr uintptr_t HP = 2u << 20;
  static const bool off = getenv("KSLAM_NO_THP") != nullptr;
  if (off || !p || bytes < 2 * HP) return;
  const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + HP - 1) & ~(HP - 1);
  const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(HP - 1);
  if (e > a) (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
#else
  (void)p;
  (void)bytes;
#endif
}

// CPUs this process may actually use: the hardware threads, capped by a cgroup v2
// CPU quota when there is one (more runnable threads than quota only get throttled)
inline int usable_cpus() {
  static const int n = [] {
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char quota[32];
      long period = 0;
      if (fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && strcmp(quota, "max") != 0) {
        long q = atol(quota);
        if (q > 0) hw = std::min<long>(hw, std::max<long>(1, (q + period - 1) / period));
      }
      fclose(f);
    }
    return hw;
  }();
  return n;
}


template <typename F>
kslam_status guarded(F &&f) {
  try {
    f();
    return KSLAM_OK;
  } catch (const HostError &e) {
    g_err = e.msg;
    return e.code;
  } catch (const std::bad_alloc &) {
    g_err = "out of host memory";
    return KSLAM_ERR_OOM;
  } catch (const std::exception &e) {
    g_err = e.what();
    return KSLAM_ERR_INTERNAL;
  }
}

}  // namespace kslam_host
#endif  // KSLAM_HOST_WORKERS_HPP_
