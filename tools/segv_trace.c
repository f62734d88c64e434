// LD_PRELOAD diagnostic: on SIGSEGV/SIGBUS print the faulting address and the native backtrace
// (module + offset) to stderr, then re-raise.  Build: gcc -shared -fPIC -O1 -o segv_trace.so segv_trace.c
// Use:   LD_PRELOAD=$PWD/tools/segv_trace.so python -X faulthandler -m pytest ...
// (python's faulthandler installs its handler later and chains to nothing, so set
//  PYTHONFAULTHANDLER= empty and do not pass -X faulthandler when using this.)
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t *si, void *uc) {
  (void)uc;
  char line[128];
  int n = snprintf(line, sizeof line, "\n[segv_trace] signal %d at address %p\n", sig, si->si_addr);
  if (n > 0) (void)!write(2, line, (size_t)n);
  void *frames[64];
  int depth = backtrace(frames, 64);
  backtrace_symbols_fd(frames, depth, 2);
  FILE *maps = fopen("/proc/self/maps", "r");
  if (maps) {  // the mappings of the modules named above, so offsets can be resolved offline
    char buf[512];
    while (fgets(buf, sizeof buf, maps))
      if (strstr(buf, "r-xp") && (strstr(buf, "lbfgsb") || strstr(buf, "gpdla") || strstr(buf, "openblas") || strstr(buf, "libc.so")))
        (void)!write(2, buf, strlen(buf));
    fclose(maps);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  static char stack[1 << 16];
  stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
  sigaltstack(&ss, NULL);
  sigaction(SIGSEGV, &sa, NULL);
  sigaction(SIGBUS, &sa, NULL);
}
