/*
 * cmio_error.c - error reporting of the CPU oracle (TEST INFRASTRUCTURE, see
 * cmio.h). The oracle never abort()s: where the reference raises cmac_error
 * (src/Error.hpp) the oracle records the first message here, returns a
 * harmless value and goes on; tests/oracle_lib.py reads the flag after every
 * call and raises. A run of the test suite thus survives an oracle-side error
 * and reports it as the failure of one test.
 */
#include "cmio.h"

#include <stdarg.h>
#include <stdio.h>

static char g_message[512];
static int g_set = 0;

void cmio_set_error(const char *fmt, ...) {
#pragma omp critical(cmio_error)
  {
    if (!g_set) {
      va_list ap;
      va_start(ap, fmt);
      vsnprintf(g_message, sizeof g_message, fmt, ap);
      va_end(ap);
      g_set = 1;
    }
  }
}

const char *cmio_last_error(void) {
  const char *m = NULL;
#pragma omp critical(cmio_error)
  {
    if (g_set)
      m = g_message;
  }
  return m;
}

void cmio_clear_error(void) {
#pragma omp critical(cmio_error)
  { g_set = 0; }
}

/* ------------------------------------------------------------------------
 * Diagnosis of a SIGABRT raised by ANY library of a test process (a GPU fault
 * makes the HSA runtime abort() on one of its own threads; Python's
 * faulthandler then shows only where the Python threads happened to be).
 * tests/conftest.py installs this handler: the native backtrace of the
 * aborting thread goes to `fd`, then the handler that was there before
 * (faulthandler's) runs. */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static int g_abort_fd = -1;
static struct sigaction g_previous_abort;

static void abort_backtrace(int sig, siginfo_t *info, void *context) {
  static const char head[] =
      "\n=== SIGABRT: native backtrace of the aborting thread ===\n";
  if (g_abort_fd >= 0) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    if (write(g_abort_fd, head, sizeof head - 1) < 0) {
    }
    backtrace_symbols_fd(frames, n, g_abort_fd);
  }
  if (g_previous_abort.sa_flags & SA_SIGINFO) {
    if (g_previous_abort.sa_sigaction)
      g_previous_abort.sa_sigaction(sig, info, context);
  } else if (g_previous_abort.sa_handler != SIG_DFL &&
             g_previous_abort.sa_handler != SIG_IGN) {
    g_previous_abort.sa_handler(sig);
  }
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
}

void cmio_install_abort_backtrace(int fd) {
  /* (the first call of backtrace() loads libgcc: not from the handler) */
  void *warm[2];
  (void)backtrace(warm, 2);
  g_abort_fd = fd;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = abort_backtrace;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGABRT, &sa, &g_previous_abort);
}
