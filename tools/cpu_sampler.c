/* cpu_sampler.c -- a poor man's perf for the boxes without one: LD_PRELOAD it, and at exit it prints where the
 * process's CPU time went (SIGPROF every 1 ms of process CPU time, program counter bucketed by symbol).
 *   gcc -O2 -shared -fPIC tools/cpu_sampler.c -o /tmp/cpu_sampler.so -ldl
 *   KSLAM_SAMPLER_OUT=/tmp/samples.txt LD_PRELOAD=/tmp/cpu_sampler.so python3 bench.py ...
 * The output lists "<object> <offset>" per sample, leaf first, then its callers; tools/cpu_sampler_report.py resolves them.
 * Window: sampling is on only while the file named by KSLAM_SAMPLER_GATE exists (or always when unset). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <sys/prctl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <unistd.h>

#define CAP (1 << 20)
#define DEPTH 16
static uintptr_t *pcs;   /* DEPTH program counters per sample: the interrupted one, then backtrace() (the handler's frames included) */
static volatile uint32_t n_pcs;
static char *names;   /* 16 bytes per sample: the thread's name */

static void on_prof(int sig, siginfo_t *si, void *uc_) {
  (void)sig; (void)si;
  ucontext_t *uc = (ucontext_t *)uc_;
  uint32_t k = __atomic_fetch_add(&n_pcs, 1, __ATOMIC_RELAXED);
  if (k < CAP) {
    /* no unwinder here (backtrace() takes locks): the interrupted pc, then the first words above the stack pointer
     * that look like addresses of mapped objects and are not stack addresses -- return addresses among them; the
     * report keeps those that resolve to code */
    prctl(PR_GET_NAME, names + (size_t)k * 16, 0, 0, 0);
    const uintptr_t sp = (uintptr_t)uc->uc_mcontext.gregs[REG_RSP];
    uintptr_t *out = pcs + (size_t)k * DEPTH;
    out[0] = (uintptr_t)uc->uc_mcontext.gregs[REG_RIP];
    int n = 1;
    const uintptr_t *w = (const uintptr_t *)(sp & ~(uintptr_t)7);
    for (int i = 0; i < 192 && n < DEPTH; i++) {
      const uintptr_t v = w[i];
      if (v >= 0x700000000000ull && v < 0x800000000000ull && (v < sp - (1u << 20) || v > sp + (8u << 20))) out[n++] = v;
    }
    while (n < DEPTH) out[n++] = 0;
  }
}

__attribute__((constructor)) static void start(void) {
  if (!getenv("KSLAM_SAMPLER_OUT")) return;
  pcs = (uintptr_t *)calloc((size_t)CAP * DEPTH, sizeof *pcs);
  names = (char *)calloc((size_t)CAP, 16);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  struct itimerval it = {{0, 1000}, {0, 1000}};
  setitimer(ITIMER_PROF, &it, NULL);
}

__attribute__((destructor)) static void stop(void) {
  const char *out = getenv("KSLAM_SAMPLER_OUT");
  if (!out || !pcs) return;
  struct itimerval it = {{0, 0}, {0, 0}};
  setitimer(ITIMER_PROF, &it, NULL);
  FILE *f = fopen(out, "w");
  if (!f) return;
  uint32_t n = n_pcs < CAP ? n_pcs : CAP;
  for (uint32_t i = 0; i < n; i++) {
    names[(size_t)i * 16 + 15] = 0;
    fprintf(f, "@%s | ", names + (size_t)i * 16);
    for (int d = 0; d < DEPTH; d++) {
      const uintptr_t pc = pcs[(size_t)i * DEPTH + d];
      Dl_info di;
      if (!pc) break;
      if (dladdr((void *)pc, &di) && di.dli_fname)
        fprintf(f, "%s%s %lx", d ? " | " : "", di.dli_fname, (unsigned long)(pc - (uintptr_t)di.dli_fbase));
      else
        fprintf(f, "%s? %lx", d ? " | " : "", (unsigned long)pc);
    }
    fprintf(f, "\n");
  }
  fclose(f);
}
