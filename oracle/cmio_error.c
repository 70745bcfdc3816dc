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
