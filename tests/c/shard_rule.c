/*
 * shard_rule.c -- CPU-side unit test (plain C, no GPU) of the target-shard rule behind the multi-GPU
 * entries (gsl_sinterp_hip_shard_bounds): contiguous ceil-sized shards, every target
 * covered exactly once, in order, for ragged totals, more ranks than targets and empty batches; plus
 * the scatter/gather bookkeeping the host driver does with it (shard r of a packed M x d target
 * array starts at first*d; the values of shard r land at out[first .. first+count)).
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include "gsl_sinterp.h"

static void check(size_t m, int world)
{
  size_t next = 0, per = (m + (size_t)world - 1) / (size_t)world;
  for (int r = 0; r < world; r++) {
    size_t first = 12345, count = 12345;
    gsl_sinterp_hip_shard_bounds(m, world, r, &first, &count);
    assert(first == next);                 /* contiguous, in rank order */
    assert(count <= per);
    assert(first + count <= m);
    if (first + count < m) assert(count == per);   /* only trailing shards are short */
    next = first + count;
  }
  assert(next == m);                       /* every target exactly once */
}

int main(void)
{
  const size_t totals[] = {0, 1, 2, 7, 8, 9, 63, 64, 65, 1000, 4097, 1000000, 10000000, 10000019};
  for (size_t t = 0; t < sizeof totals / sizeof totals[0]; t++)
    for (int world = 1; world <= 64; world++) check(totals[t], world);
  /* gather emulation: every rank "evaluates" its shard of k -> 3k+1 and writes it at its offset */
  const size_t m = 100003;
  const int world = 8;
  double *y = malloc(m * 2 * sizeof(double)), *out = malloc(m * sizeof(double));
  for (size_t k = 0; k < m; k++) { y[2 * k] = (double)k; y[2 * k + 1] = -(double)k; out[k] = -1.0; }
  for (int r = 0; r < world; r++) {
    size_t first, count;
    gsl_sinterp_hip_shard_bounds(m, world, r, &first, &count);
    const double *shard = y + first * 2;
    for (size_t i = 0; i < count; i++) out[first + i] = 3.0 * shard[2 * i] + 1.0 + 0.0 * shard[2 * i + 1];
  }
  for (size_t k = 0; k < m; k++) assert(out[k] == 3.0 * (double)k + 1.0);
  /* NULL outputs are allowed, nonsensical world / rank are clamped instead of dividing by zero */
  gsl_sinterp_hip_shard_bounds(10, 0, 0, NULL, NULL);
  size_t f, c;
  gsl_sinterp_hip_shard_bounds(10, 0, -3, &f, &c);
  assert(f == 0 && c == 10);
  free(y); free(out);
  puts("shard rule: ok");
  return 0;
}
