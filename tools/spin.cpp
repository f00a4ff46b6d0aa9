// spin.cpp -- how many CPUs does this job really get?  N threads each do a fixed amount of
// integer work; prints wall time and effective parallelism.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
  for (int n : {1, 8, 16, 32, 64, 128, 256}) {
    std::vector<std::thread> th;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++)
      th.emplace_back([i] {
        volatile unsigned long x = i;
        for (long k = 0; k < 400000000L; k++) x = x * 6364136223846793005UL + 1442695040888963407UL;
      });
    for (auto &t : th) t.join();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    static double one = 0;
    if (n == 1) one = s;
    printf("%3d threads: %.3f s  -> effective CPUs %.1f\n", n, s, n * one / s);
  }
}
