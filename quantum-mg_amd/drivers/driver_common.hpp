// driver_common.hpp -- what every driver does before and after its work, so that a fault locates itself
// (a driver died with SIGSEGV and empty stdout AND stderr -- its output sat in a block-buffered pipe).
//   * stdout is line-buffered (std::cout shares the C stream's buffer), so every finished line is in the pipe;
//   * phase markers: `[QMG-PHASE]: <name>` on stdout (root rank) and the CURRENT phase kept for the fault handler;
//   * SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL handler on an alternate stack: the signal, the current phase and a
//     backtrace (backtrace_symbols_fd; the drivers are linked -g -rdynamic) go to stderr with async-signal-safe writes, then the
//     default action is restored and the signal re-raised (the exit status stays the signal);
//   * leave(rc): flush, ordered teardown of the library (qmg_shutdown: device sync, per-thread workspaces and pinned
//     buffers released), phase "exit" -- whatever faults after that is the runtime's own static teardown, and says so.
// QMG_MALLOC_POISON=1 in the environment switches the "malloc_poison" tuning key on (every qmg_malloc filled with NaNs).
#ifndef QMG_DRIVER_COMMON_HPP
#define QMG_DRIVER_COMMON_HPP

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "../../include/qmg_hip.h"

namespace qmg_driver {

inline const char*& current_phase() { static const char* p = "startup"; return p; }   // process-wide: the handler reads it

inline void phase(const char* name, bool print = true) {
  current_phase() = name;
  if (print) { std::cout << "[QMG-PHASE]: " << name << "\n"; std::cout.flush(); }
}

inline void sig_write(const char* s) { ssize_t r = write(2, s, strlen(s)); (void)r; }

inline void fault_handler(int sig) {
  char num[16];
  int n = sig, k = 0;
  char tmp[16];
  do { tmp[k++] = (char)('0' + n % 10); n /= 10; } while (n && k < 15);
  for (int i = 0; i < k; i++) num[i] = tmp[k - 1 - i];
  num[k] = 0;
  sig_write("\n[QMG-FATAL]: signal ");
  sig_write(num);
  sig_write(sig == SIGSEGV ? " (SIGSEGV)" : sig == SIGABRT ? " (SIGABRT)" : sig == SIGBUS ? " (SIGBUS)" : sig == SIGFPE ? " (SIGFPE)" : "");
  sig_write(" in phase '");
  sig_write(current_phase());
  sig_write("'; backtrace:\n");
  void* frames[64];
  const int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, 2);
  sig_write("[QMG-FATAL]: end of backtrace\n");
  signal(sig, SIG_DFL);
  raise(sig);
}

inline void install() {
  static bool done = false;
  if (done) return;
  done = true;
  setvbuf(stdout, nullptr, _IOLBF, 1 << 16);   // std::cout is synchronised with stdout: line-buffered from here on
  { void* warm[4]; backtrace(warm, 4); }        // loads libgcc's unwinder now, not inside the handler
  static char altstack[64 * 1024];
  stack_t ss;
  ss.ss_sp = altstack; ss.ss_size = sizeof(altstack); ss.ss_flags = 0;
  sigaltstack(&ss, nullptr);
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = fault_handler;
  sa.sa_flags = SA_ONSTACK | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
  for (int s : sigs) sigaction(s, &sa, nullptr);
  if (const char* p = getenv("QMG_MALLOC_POISON")) qmg_set_tuning("malloc_poison", atoi(p));
}

// first statement of main(): installs everything above
struct Guard { Guard() { install(); } };

// last statement of main(): `return qmg_driver::leave(rc);`
inline int leave(int rc) {
  phase("teardown", false);
  std::cout.flush();
  fflush(stdout);
  qmg_shutdown();
  phase("exit (after qmg_shutdown: static destructors of the process)", false);
  return rc;
}

// R ranks as host threads of this process on the one GPU (csrc/qmg_comm.hip: ThreadWorld): every thread attaches as a rank,
// gets its own stream, runs `run_rank(r)`, and leaves through the library's ordered teardown for ITS workspaces.
// set_stream(st): the caller's way to make `st` the facade's current stream of that thread (qmg::current_stream() = st).
template <typename Run, typename SetStream>
inline int emulate_ranks(int R, Run run_rank, SetStream set_stream) {
  if (qmg_comm_emulate_begin(R) != QMG_SUCCESS) { std::cout << "[QMG-ERROR]: qmg_comm_emulate_begin failed\n"; return 2; }
  std::vector<int> rc(R, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < R; r++)
    th.emplace_back([&, r] {
      qmg_comm_emulate_attach(r);
      void* st = 0;
      qmg_stream_create(&st);
      set_stream(st);
      rc[r] = run_rank(r);
      qmg_stream_sync(st);
      set_stream((void*)0);
      qmg_shutdown();
      qmg_stream_destroy(st);
    });
  for (auto& t : th) t.join();
  qmg_comm_emulate_end();
  for (int r = 0; r < R; r++) if (rc[r]) return rc[r];
  return 0;
}

}  // namespace qmg_driver

#endif
