/* Separable symmetric Gaussian smoothing of a float32 volume, scipy.ndimage order
 * (tmp = x[l] * w[0]; for j = WR..1: tmp += (x[l-j] + x[l+j]) * w[j]; float32 store per
 * axis; 'reflect' boundary), in two forms: products and sums rounded separately (what
 * scipy's C does on x86-64) and fused multiply-adds (what a contracting compiler makes
 * of it).  Used by tools/dev/find_fma_witness.py to find small volumes on which the two
 * (fused = 1: every product fused into the running sum; fused = 2: what hipcc made of
 * the kernels before contraction was switched off - x[0] * w[0] fused onto the rounded
 * product of the outermost pair, then every further pair fused) differ after the
 * float32 store - the regression inputs of
 * tests/test_gpu_voxel2obj.py::test_smoothing_rounds_products_and_sums_separately.
 *   gcc -O2 -ffp-contract=off -shared -fPIC -o /tmp/fma_witness.so fma_witness.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int64_t reflect(int64_t i, int64_t n) { return i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i); }

static void pass(const float *in, float *out, const int64_t d[3], int axis, const double *w,
                 int wr, int fused) {
  int64_t stride[3] = {d[1] * d[2], d[2], 1};
  const int64_t n = d[axis], st = stride[axis];
  double *line = malloc(sizeof(double) * (size_t)(n + 2 * wr));
  for (int64_t a = 0; a < d[0]; ++a)
    for (int64_t b = 0; b < d[1]; ++b)
      for (int64_t c = 0; c < d[2]; ++c) {
        const int64_t idx[3] = {a, b, c};
        if (idx[axis] != 0) continue;
        const int64_t base = a * stride[0] + b * stride[1] + c * stride[2];
        for (int64_t i = -wr; i < n + wr; ++i) line[i + wr] = (double)in[base + reflect(i, n) * st];
        for (int64_t l = 0; l < n; ++l) {
          const double *x = line + l + wr;
          double tmp;
          int j = wr;
          if (fused == 2 && wr >= 1) {
            const double p = (x[-wr] + x[wr]) * w[wr];
            tmp = fma(x[0], w[0], p);
            j = wr - 1;
          } else {
            tmp = x[0] * w[0];
          }
          for (; j >= 1; --j) {
            const double s = x[-j] + x[j];
            if (fused) tmp = fma(s, w[j], tmp);
            else { const double p = s * w[j]; tmp = tmp + p; }
          }
          out[base + l * st] = (float)tmp;
        }
      }
  free(line);
}

/* w[0..wr]: weight by distance */
void smooth3(const float *in, float *out, float *scratch, const int64_t d[3], const double *w,
             int wr, int fused) {
  pass(in, out, d, 0, w, wr, fused);
  pass(out, scratch, d, 1, w, wr, fused);
  pass(scratch, out, d, 2, w, wr, fused);
}
