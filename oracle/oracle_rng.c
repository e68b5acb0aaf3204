/*
 * oracle_rng.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * MT19937 as the reference's default generator, the rejection-sampled
 * integer draw and the Durstenfeld shuffle: together they fix the insertion
 * order of the Delaunay build, hence the DAG node numbering.
 *   rng/mt.c:79-129   (generate + temper)
 *   rng/mt.c:131-152  (seeding; seed 0 -> 4357)
 *   rng/gsl_rng.h:190-212 (uniform_int with scale = range / n, reject k >= n)
 *   randist/shuffle.c:69-79 (for i = n-1 .. 1: swap(i, uniform_int(i+1)))
 */
#include "oracle.h"
#include <stdlib.h>

#define MT_N 624
#define MT_M 397

struct oracle_mt {
  uint32_t w[MT_N];
  int pos;
};

oracle_mt *oracle_mt_alloc(unsigned long seed)
{
  oracle_mt *r = (oracle_mt *)malloc(sizeof *r);
  if (!r) return NULL;
  uint32_t s = (uint32_t)(seed & 0xffffffffUL);
  if (seed == 0) s = 4357u;                       /* rng/mt.c:137-138 */
  r->w[0] = s;
  for (int i = 1; i < MT_N; i++) {
    uint32_t prev = r->w[i - 1];
    r->w[i] = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;  /* rng/mt.c:145-148 */
  }
  r->pos = MT_N;
  return r;
}

void oracle_mt_free(oracle_mt *r) { free(r); }

static void mt_refill(oracle_mt *r)
{
  uint32_t *w = r->w;
  for (int k = 0; k < MT_N; k++) {
    uint32_t y = (w[k] & 0x80000000u) | (w[(k + 1) % MT_N] & 0x7fffffffu);
    uint32_t twist = (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    w[k] = w[(k + MT_M) % MT_N] ^ twist;
  }
  r->pos = 0;
}

unsigned long oracle_mt_get(oracle_mt *r)
{
  if (r->pos >= MT_N) mt_refill(r);
  uint32_t k = r->w[r->pos++];
  k ^= k >> 11;
  k ^= (k << 7) & 0x9d2c5680u;
  k ^= (k << 15) & 0xefc60000u;
  k ^= k >> 18;
  return (unsigned long)k;
}

unsigned long oracle_mt_uniform_int(oracle_mt *r, unsigned long n)
{
  const unsigned long range = 0xffffffffUL;       /* max - min, rng/mt.c:207-214 */
  if (n == 0 || n > range) return 0;
  unsigned long scale = range / n;
  unsigned long k;
  do {
    k = oracle_mt_get(r) / scale;
  } while (k >= n);
  return k;
}

void oracle_shuffle_sizet(oracle_mt *r, size_t *base, size_t n)
{
  if (n < 2) return;
  for (size_t i = n - 1; i > 0; i--) {
    size_t j = (size_t)oracle_mt_uniform_int(r, (unsigned long)(i + 1));
    size_t tmp = base[i];
    base[i] = base[j];
    base[j] = tmp;
  }
}
