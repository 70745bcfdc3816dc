/* How many cores does this process really get? Pure-compute OpenMP scaling
 * (no memory traffic): iterations/s with T threads, T = 1, 2, 4, ... max.
 * gcc -O2 -fopenmp tools/host_cores_probe.c -o /tmp/probe -lm */
#include <math.h>
#include <omp.h>
#include <stdio.h>
int main(void) {
  const int maxt = omp_get_max_threads();
  printf("omp_get_max_threads %d\n", maxt);
  double base = 0.;
  for (int t = 1; t <= maxt; t *= 2) {
    omp_set_num_threads(t);
    const long per = 40000000;
    double sum = 0.;
    const double t0 = omp_get_wtime();
#pragma omp parallel reduction(+ : sum)
    {
      double x = 1.0 + omp_get_thread_num();
      for (long i = 0; i < per; ++i)
        x = x * 1.0000001 + 1e-9;
      sum += x;
    }
    const double dt = omp_get_wtime() - t0;
    const double rate = (double)per * t / dt;
    if (t == 1)
      base = rate;
    printf("threads %4d  %.3g it/s  speedup %.1f  (%g)\n", t, rate,
           rate / base, sum);
  }
  return 0;
}
