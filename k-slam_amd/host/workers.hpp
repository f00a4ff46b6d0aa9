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
#include <pthread.h>
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

// Names the calling thread (top -H, /proc/<pid>/task/*/comm: bench.py's cpu_s_by_thread groups CPU time by these).
inline void name_thread(const char *name) {
#if defined(__linux__)
  (void)pthread_setname_np(pthread_self(), name);   // at most 15 characters
#else
  (void)name;
#endif
}

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

// ---------------------------------------------------------------- worker pool --
// Persistent workers shared by every parallel loop of the host stages.  A loop (`tasks`) is a job: a counter of
// tasks handed out and one of tasks finished; the calling thread works on its own job, the workers on whichever
// active job has tasks left and the fewest threads on it.  Several jobs may be active at once -- the SAM text of a
// batch on one thread and its taxonomy part on another (stream.cpp): the serial stretches between the loops of one
// run under the loops of the other, instead of every loop owning the pool while the rest of the process waits for it.
class Pool {
 public:
  static Pool &get() {
    static Pool *p = new Pool();  // never destroyed: workers are detached
    return *p;
  }
  // dynamic schedule of n_tasks over at most n_threads threads (the caller included)
  void tasks(int n_threads, size_t n_tasks, const std::function<void(size_t)> &f) {
    if (n_tasks == 0) return;
    Job job;
    job.f = &f;
    job.n = n_tasks;
    job.max_threads = (int)std::min<size_t>((size_t)std::min(std::max(n_threads, 1), cap_.load(std::memory_order_relaxed)), n_tasks);
    if (job.max_threads <= 1) {
      for (size_t t = 0; t < n_tasks; t++) f(t);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      while ((int)workers_ < job.max_threads - 1) {
        const int index = (int)workers_++;
        std::thread([this, index] { loop(index); }).detach();
      }
      job.threads = 1;   // the caller
      active_.push_back(&job);
    }
    work_.notify_all();
    drain(job);
    std::unique_lock<std::mutex> lk(m_);
    job.threads--;
    done_.wait(lk, [&] { return job.finished == job.n && job.threads == 0; });
    active_.erase(std::find(active_.begin(), active_.end(), &job));
    if (job.failed) throw job.error;
  }
  // At most `n` threads on any loop from now on and at most n - 1 workers awake (0: no limit).  For a caller that runs
  // other busy threads next to the loops -- the batch loop's writer, lanes and second host thread -- inside a CPU
  // quota: more runnable threads than the quota has CPUs get the whole process throttled (cgroup cpu.max).
  // Callers nest and overlap (one batch loop per GPU context in one process): every caller registers its cap and takes it
  // back when it is done; the strictest registered cap is the one in force, none registered = no limit.
  void add_cap(int n) {
    std::lock_guard<std::mutex> lk(cap_m_);
    caps_.push_back(n > 0 ? n : (1 << 30));
    cap_.store(*std::min_element(caps_.begin(), caps_.end()), std::memory_order_relaxed);
  }
  void remove_cap(int n) {
    std::lock_guard<std::mutex> lk(cap_m_);
    auto it = std::find(caps_.begin(), caps_.end(), n > 0 ? n : (1 << 30));
    if (it != caps_.end()) caps_.erase(it);
    cap_.store(caps_.empty() ? (1 << 30) : *std::min_element(caps_.begin(), caps_.end()), std::memory_order_relaxed);
  }

 private:
  struct Job {
    const std::function<void(size_t)> *f = nullptr;
    size_t n = 0;
    std::atomic<size_t> next{0};
    size_t finished = 0;      // guarded by m_
    int threads = 0;          // threads inside drain(), guarded by m_
    int max_threads = 1;
    bool failed = false;
    HostError error;
  };
  // takes tasks of `job` until none is left; returns the number it ran
  void drain(Job &job) {
    size_t ran = 0;
    bool failed = false;
    HostError error;
    for (;;) {
      const size_t t = job.next.fetch_add(1, std::memory_order_relaxed);
      if (t >= job.n) break;
      try {
        if (!failed) (*job.f)(t);
      } catch (const HostError &e) {
        failed = true;
        error = e;
      } catch (const std::exception &e) {
        failed = true;
        error = HostError{KSLAM_ERR_INTERNAL, e.what()};
      }
      ran++;
    }
    std::lock_guard<std::mutex> lk(m_);
    job.finished += ran;
    if (failed && !job.failed) {
      job.failed = true;
      job.error = error;
    }
  }
  void loop(int index) {
    name_thread("kslam-pool");
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      Job *pick = nullptr;
      if (index < cap_.load(std::memory_order_relaxed) - 1)
      for (Job *j : active_)
        if (j->next.load(std::memory_order_relaxed) < j->n && j->threads < j->max_threads && (!pick || j->threads < pick->threads)) pick = j;
      if (!pick) {
        work_.wait(lk);
        continue;
      }
      pick->threads++;
      lk.unlock();
      drain(*pick);
      lk.lock();
      pick->threads--;
      if (pick->finished == pick->n && pick->threads == 0) done_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable work_, done_;
  std::vector<Job *> active_;
  size_t workers_ = 0;
  std::atomic<int> cap_{1 << 30};
  std::mutex cap_m_;
  std::vector<int> caps_;   // guarded by cap_m_
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
