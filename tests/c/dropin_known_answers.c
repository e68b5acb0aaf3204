/*
 * dropin_known_answers.c -- a C program written against include/gsl_sinterp.h the way a user of the
 * reference writes against interpolation/linear_simplex.h: same type names, macros (SIMP / LINK /
 * POINT / LEAF / N_CHILDREN / DATA_POINT / FIND), call sequence and view types.  It checks
 *   (1) the known answers the reference asserts in interpolation/scattered_interp_example.c:38-77,
 *   (2) the outputs of the reference captured at survey time on its 50-station dataset
 *       (tests/golden/survey_known_answers.json, cfg1: init(data, NULL, NULL, 0, mt19937 seed 0)).
 * Host API only (no GPU needed).  Usage: dropin_known_answers <weather_stations.csv>
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gsl_sinterp.h"

static void first_insertion_and_location(void)
{
  simplex_tree *tree = simplex_tree_alloc(2, 10);
  simplex_tree_free(tree);

  simplex_tree_accel *accel = simplex_tree_accel_alloc(2);
  tree = simplex_tree_alloc(2, 50);
  assert(GSL_SUCCESS == simplex_tree_init(tree, NULL, NULL, NULL, SIMPLEX_TREE_NOSTANDARDIZE, NULL));

  double xy[4] = {-88, 41, -89, 41};
  gsl_matrix_view data = gsl_matrix_view_array(xy, 2, 2);
  gsl_vector_view p = gsl_matrix_row(&data.matrix, 0);
  simplex_index leaf = find_leaf(tree, NULL, &p.vector, NULL);
  assert(leaf == 0);
  /* interpolating on the empty cage gives exactly 0 */
  assert(0 == interp_point(tree, leaf, &data.matrix, NULL, &p.vector, accel));

  assert(GSL_SUCCESS == insert_point(tree, leaf, &data.matrix, &p.vector, accel));
  assert(!LEAF(leaf));
  const int expect[3][3] = {{0, -2, -3}, {0, -1, -3}, {0, -1, -2}};
  for (int c = 0; c < 3; c++)
    for (int v = 0; v < 3; v++) assert(expect[c][v] == POINT(LINK(leaf, c), v));
  assert(1 == in_hypersphere(tree, 0, &data.matrix, 0, accel));

  p = gsl_matrix_row(&data.matrix, 1);
  leaf = find_leaf(tree, &data.matrix, &p.vector, accel);
  assert(0 == POINT(leaf, 0) && -2 == POINT(leaf, 1) && -3 == POINT(leaf, 2));

  simplex_tree_free(tree);
  simplex_tree_accel_free(accel);
}

static void weather_dataset(const char *csv)
{
  double tab[150];
  int n = 0;
  FILE *f = fopen(csv, "r");
  assert(f);
  char line[256];
  while (fgets(line, sizeof line, f) && n < 50) {
    if (line[0] == '#') continue;
    assert(3 == sscanf(line, "%lf,%lf,%lf", &tab[3 * n], &tab[3 * n + 1], &tab[3 * n + 2]));
    n++;
  }
  fclose(f);
  assert(n == 50);
  /* 50 x 2 submatrix of a 50 x 3 array (tda = 3) and a stride-3 response column */
  gsl_matrix_view all = gsl_matrix_view_array(tab, 50, 3);
  gsl_vector_view response = gsl_matrix_column(&all.matrix, 2);
  gsl_matrix_view data = gsl_matrix_submatrix(&all.matrix, 0, 0, 50, 2);

  gsl_rng_env_setup();
  gsl_rng *rng = gsl_rng_alloc(gsl_rng_default);
  simplex_tree_accel *accel = simplex_tree_accel_alloc(2);
  simplex_tree *tree = simplex_tree_alloc(2, 50);
  assert(GSL_SUCCESS == simplex_tree_init(tree, &data.matrix, NULL, NULL, 0, rng));
  assert(tree->n_simplexes == 369);

  const double q[3][2] = {{-88, 41}, {-88, 42}, {-89, 42.5}};
  const int leaf_expect[3] = {275, 204, 201};
  const char *value_expect[3] = {"274.08608080148747", "277.65796408212475", "271.75363625014228"};
  for (int k = 0; k < 3; k++) {
    double pt[2] = {q[k][0], q[k][1]};
    gsl_vector_view point = gsl_vector_view_array(pt, 2);
    simplex_index leaf = find_leaf(tree, &data.matrix, &point.vector, accel);
    double v = interp_point(tree, leaf, &data.matrix, &response.vector, &point.vector, accel);
    char got[64];
    snprintf(got, sizeof got, "%.17g", v);
    assert(leaf == leaf_expect[k]);
    assert(0 == strcmp(got, value_expect[k]));
  }
  /* the reference header's N_CHILDREN / DATA_POINT / FIND macros (linear_simplex.h:67-102), used the
     way linear_simplex.c:365,452,614 uses them: the root was split by the first insertion (d+1
     children), leaves have none; every neighbour link of a leaf has a reverse link; DATA_POINT
     resolves cage seeds (negative ids) and data rows through the shuffle. */
  assert(N_CHILDREN(0) == 3);
  int n_leaves = 0;
  for (simplex_index node = 1; node < tree->n_simplexes; node++) {
    if (!LEAF(node)) { assert(N_CHILDREN(node) == 2 || N_CHILDREN(node) == 3); continue; }
    n_leaves++;
    assert(N_CHILDREN(node) == 0);
    for (int i = 0; i < 3; i++) {
      simplex_index neighbor = LINK(node, i);
      if (!neighbor) continue;
      int j;
      FIND(j, LINK(neighbor, j) == node, "no reverse link");
      assert(LEAF(neighbor));
    }
  }
  assert(n_leaves == 2 * 50 + 1);
  gsl_vector_view seed0 = DATA_POINT(&data.matrix, -1);
  assert(seed0.vector.data == tree->seed_points->data && seed0.vector.size == 2);
  gsl_vector_view first = DATA_POINT(&data.matrix, 0);
  assert(first.vector.data == tab + 3 * gsl_permutation_get(tree->shuffle, 0));
  simplex_tree_free(tree);
  simplex_tree_accel_free(accel);
  gsl_rng_free(rng);
}

int main(int argc, char **argv)
{
  first_insertion_and_location();
  if (argc > 1) weather_dataset(argv[1]);
  puts("dropin known answers: ok");
  return 0;
}
