// pool_check.cpp -- the host stages' worker pool (k-slam_amd/host/workers.hpp) under concurrent jobs: several threads
// run parallel loops at the same time (as stream.cpp's SAM-text and taxonomy threads do); every task must run exactly
// once, a failing task must surface as the loop's error in the thread that started it and in no other, and a loop
// started from inside a task must not deadlock.  Built with -fsanitize=thread by tests/test_tail.py.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../k-slam_amd/host/workers.hpp"

using kslam_host::Pool;

int main() {
  std::atomic<int> bad{0};
  auto user = [&](int who) {
    for (int round = 0; round < 200; round++) {
      const size_t n = 1 + (size_t)((round * 37 + who * 11) % 500);
      std::vector<int> hit(n, 0);
      bool want_fail = round % 17 == who;
      bool failed = false;
      try {
        Pool::get().tasks(2 + (round + who) % 7, n, [&](size_t t) {
          hit[t]++;
          if (want_fail && t == n / 2) kslam_host::fail(KSLAM_ERR_ARG, "planned");
          if (t == 0 && round % 50 == 0) {   // a loop inside a task
            std::atomic<int> inner{0};
            Pool::get().tasks(4, 64, [&](size_t) { inner++; });
            if (inner != 64) bad++;
          }
        });
      } catch (const kslam_host::HostError &e) {
        failed = true;
        if (e.msg != "planned") bad++;
      }
      if (failed != want_fail) bad++;
      if (!failed)
        for (size_t t = 0; t < n; t++)
          if (hit[t] != 1) bad++;
    }
  };
  std::vector<std::thread> th;
  for (int w = 0; w < 4; w++) th.emplace_back(user, w);
  for (auto &t : th) t.join();
  printf(bad ? "FAILED %d\n" : "ok\n", (int)bad);
  return bad ? 1 : 0;
}
