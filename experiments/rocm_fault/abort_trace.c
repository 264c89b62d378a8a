/* abort_trace.c -- LD_PRELOAD helper: prints the NATIVE backtrace of the thread that raises
 * SIGABRT / SIGSEGV / SIGBUS (name, tid, frames as module(+offset) -- resolvable with addr2line on
 * the same image), then hands over to whatever handler was installed before (or the default).
 *
 * Why: the round-4 GPU suite died one run in eleven with a bare SIGABRT raised on a thread that
 * has no Python state (faulthandler only shows the main thread parked inside an unrelated copy).
 * Python's faulthandler, enabled later by pytest, chains to the handler it replaced, i.e. to this
 * one, in the raising thread.
 *
 *   gcc -O1 -g -shared -fPIC tools/abort_trace.c -o tools/libabort_trace.so
 *   LD_PRELOAD=tools/libabort_trace.so python -m pytest tests -m gpu ...
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

static struct sigaction g_prev[65];
/* Where the report goes: the file named by ABORT_TRACE_LOG, opened when the library loads (pytest
 * points file descriptor 2 at its capture file while a test runs, so a report written there is
 * lost with the process -- which is what happened to the first crash this was built for), and
 * file descriptor 2 as well. */
static int g_log_fd = -1;

static void put(const char* s) {
  ssize_t r = write(2, s, strlen(s));
  if (g_log_fd >= 0) r = write(g_log_fd, s, strlen(s));
  (void)r;
}

static void put_num(long v) {
  char b[32];
  int i = 31;
  b[i] = 0;
  const int neg = v < 0;
  if (neg) v = -v;
  if (v == 0) b[--i] = '0';
  while (v > 0 && i > 1) {
    b[--i] = (char)('0' + v % 10);
    v /= 10;
  }
  if (neg) b[--i] = '-';
  put(b + i);
}

static const char* g_maps_path; /* ABORT_TRACE_MAPS, read when the library loads */

static void copy_maps(void) {
  const char* path = g_maps_path;
  if (!path) return;
  int in = open("/proc/self/maps", O_RDONLY), out = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (in >= 0 && out >= 0) {
    char buf[4096];
    ssize_t n;
    while ((n = read(in, buf, sizeof buf)) > 0) {
      ssize_t r = write(out, buf, (size_t)n);
      (void)r;
    }
  }
  if (in >= 0) close(in);
  if (out >= 0) close(out);
}

static void on_signal(int sig, siginfo_t* info, void* ctx) {
  char name[32] = "?";
  prctl(PR_GET_NAME, name, 0, 0, 0);
  put("\n=== abort_trace: signal ");
  put_num(sig);
  put(" on thread '");
  put(name);
  put("' tid ");
  put_num((long)syscall(SYS_gettid));
  put(" (pid ");
  put_num((long)getpid());
  put(") -- native backtrace ===\n");
  put("si_code ");
  put_num(info ? (long)info->si_code : -1);
  put(" si_pid ");
  put_num(info ? (long)info->si_pid : -1);
  put("\n");
  void* frames[128];
  int n = backtrace(frames, 128);
  backtrace_symbols_fd(frames, n, 2);
  if (g_log_fd >= 0) backtrace_symbols_fd(frames, n, g_log_fd);
  put("=== abort_trace: end ===\n");
  copy_maps();
  /* hand over */
  struct sigaction* prev = &g_prev[sig];
  if ((prev->sa_flags & SA_SIGINFO) && prev->sa_sigaction) {
    prev->sa_sigaction(sig, info, ctx);
    return;
  }
  if (!(prev->sa_flags & SA_SIGINFO) && prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN &&
      prev->sa_handler) {
    prev->sa_handler(sig);
    return;
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  void* warm[4];
  (void)backtrace(warm, 4); /* loads libgcc_s now, not inside the handler */
  g_maps_path = getenv("ABORT_TRACE_MAPS");
  const char* log = getenv("ABORT_TRACE_LOG");
  if (log) g_log_fd = open(log, O_WRONLY | O_CREAT | O_APPEND | O_CLOEXEC, 0644);
  const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS};
  for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; ++i) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_signal;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(sigs[i], &sa, &g_prev[sigs[i]]);
  }
}
