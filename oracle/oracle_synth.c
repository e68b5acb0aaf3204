/*
 * oracle_synth.c -- TEST INFRASTRUCTURE (see oracle.h).
 * Deterministic synthetic clouds of SURVEY.md 8(d) / BASELINE.md section 3:
 *   u(k) = (splitmix64(seed ^ k) >> 11) * 2^-53
 *   centres x_i[c] = u(i*d+c), seed 0xC0FFEE01
 *   targets y_k[c] = 0.02 + 0.96*u(k*d+c), seed 0xC0FFEE02
 *   data    f_i = sum_c sin(3(c+1) x_i[c])
 */
#include "oracle.h"
#include <math.h>

uint64_t oracle_splitmix64(uint64_t z)
{
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

static double unit(uint64_t seed, uint64_t k)
{
  return (double)(oracle_splitmix64(seed ^ k) >> 11) * 0x1.0p-53;
}

void oracle_synth_centres(double *x, size_t n, int dim)
{
  for (size_t i = 0; i < n * (size_t)dim; i++) x[i] = unit(0xC0FFEE01ULL, i);
}

void oracle_synth_targets(double *y, size_t first, size_t m, int dim)
{
  for (size_t i = 0; i < m * (size_t)dim; i++)
    y[i] = 0.02 + 0.96 * unit(0xC0FFEE02ULL, first * (size_t)dim + i);
}

void oracle_synth_response(const double *x, size_t n, int dim, double *f)
{
  for (size_t i = 0; i < n; i++) {
    double s = 0.0;
    for (int c = 0; c < dim; c++) s += sin(3.0 * (c + 1) * x[i * dim + c]);
    f[i] = s;
  }
}
