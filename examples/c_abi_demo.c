/* The C-ABI from plain C (C99, gcc, no C++ and no HIP headers): one frame pair, solve, print the pose.
 *   c_abi_demo n grid_rows grid_cols fx fy cx cy points.f64 grid.f64
 * points.f64: n x 3 doubles (x y z of frame A's edge points); grid.f64: grid_rows x grid_cols doubles, the Grid2D view
 * the reference builds at standalone_edge_align.cpp:258 (rows = image width).  Prints
 *   q0 q1 q2 q3 t0 t1 t2 iterations termination final_cost
 * with 17 significant digits, so that a caller can compare bit for bit. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "ea_hip.h"

static double *read_doubles(const char *path, size_t count) {
  FILE *f = fopen(path, "rb");
  double *buf = (double *)malloc(count * sizeof(double));
  if (!f || !buf || fread(buf, sizeof(double), count, f) != count) {
    fprintf(stderr, "cannot read %lu doubles from %s\n", (unsigned long)count, path);
    exit(2);
  }
  fclose(f);
  return buf;
}

int main(int argc, char **argv) {
  if (argc != 10) {
    fprintf(stderr, "usage: %s n grid_rows grid_cols fx fy cx cy points.f64 grid.f64\n", argv[0]);
    return 2;
  }
  const long n = atol(argv[1]);
  const int rows = atoi(argv[2]), cols = atoi(argv[3]);
  ea_camera cam;
  cam.fx = atof(argv[4]); cam.fy = atof(argv[5]); cam.cx = atof(argv[6]); cam.cy = atof(argv[7]);
  double *xyz = read_doubles(argv[8], (size_t)n * 3);
  double *grid = read_doubles(argv[9], (size_t)rows * (size_t)cols);

  ea_problem *p = NULL;
  ea_options opt;
  ea_summary s;
  double q[4] = {1.0, 0.0, 0.0, 0.0}, t[3] = {0.0, 0.0, 0.0};
  int rc = ea_problem_create(&p, &cam, EA_F64, 0);
  if (rc == EA_OK) rc = ea_problem_set_points(p, xyz, n, 3);
  if (rc == EA_OK) rc = ea_problem_set_dt(p, grid, rows, cols);
  if (rc == EA_OK) rc = ea_problem_set_loss(p, EA_LOSS_CAUCHY, 1.0);
  ea_default_options(&opt);
  if (rc == EA_OK) rc = ea_solve(p, &opt, q, t, &s);
  if (rc != EA_OK) {
    fprintf(stderr, "libea_hip error %d: %s\n", rc, ea_last_error());
    return 1;
  }
  /* materialised mode at the solved pose: the rows themselves (r and the 1x6 row of every point, loss-corrected); their
   * J^T r must be the gradient ea_eval reports */
  double jtr_err = -1.0;
  {
    int64_t rows = 0, bad = 0, nb = 0;
    double cost = 0.0, JtJ[36], Jtr[6], g[6] = {0, 0, 0, 0, 0, 0}, gmax = 0.0;
    rc = ea_problem_num_rows(p, &rows);
    double *r = (double *)malloc((size_t)(rows > 0 ? rows : 1) * sizeof(double));
    double *J = (double *)malloc((size_t)(rows > 0 ? rows : 1) * 6 * sizeof(double));
    if (rc == EA_OK) rc = ea_eval_rows(p, q, t, 1, 0, r, J, rows, &bad);
    if (rc == EA_OK) rc = ea_eval(p, q, t, &cost, JtJ, Jtr, &nb);
    if (rc != EA_OK || rows != n) {
      fprintf(stderr, "libea_hip error %d: %s\n", rc, ea_last_error());
      return 1;
    }
    for (int64_t i = 0; i < rows; ++i)
      for (int a = 0; a < 6; ++a) g[a] += J[6 * i + a] * r[i];
    jtr_err = 0.0;
    for (int a = 0; a < 6; ++a) {
      const double d = g[a] > Jtr[a] ? g[a] - Jtr[a] : Jtr[a] - g[a], m = Jtr[a] > 0 ? Jtr[a] : -Jtr[a];
      if (d > jtr_err) jtr_err = d;
      if (m > gmax) gmax = m;
    }
    /* relative to the scale of the terms that cancel in a gradient near its minimum: sqrt(trace JtJ * 2 cost) */
    {
      double tr = 0.0;
      for (int a = 0; a < 6; ++a) tr += JtJ[6 * a + a];
      const double scale = tr * 2.0 * cost;
      jtr_err = scale > 0 ? jtr_err / (scale > 1 ? scale : 1) : jtr_err;
    }
    free(r); free(J);
  }
  printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d %.17g %.17g\n", q[0], q[1], q[2], q[3], t[0], t[1], t[2],
         s.num_iterations, s.termination, s.final_cost, jtr_err);
  ea_problem_destroy(p);
  free(xyz); free(grid);
  return 0;
}
