/* The C-ABI from plain C (C99, gcc, no C++ and no HIP headers): one frame pair, solve, print the pose.
 *   c_abi_demo n grid_rows grid_cols fx fy cx cy points.f64 grid.f64
 * points.f64: n x 3 doubles (x y z of frame A's edge points); grid.f64: grid_rows x grid_cols doubles, the Grid2D view
 * the reference builds at standalone_edge_align.cpp:258 (rows = image width).  Prints
 *   q0 q1 q2 q3 t0 t1 t2 iterations termination final_cost
 * with 17 significant digits, so that a caller can compare bit for bit. */
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
  printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d %.17g\n", q[0], q[1], q[2], q[3], t[0], t[1], t[2],
         s.num_iterations, s.termination, s.final_cost);
  ea_problem_destroy(p);
  free(xyz); free(grid);
  return 0;
}
