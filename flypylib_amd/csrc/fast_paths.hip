// Dispatch of fused fast paths.  (vgg_like kernels land in vgg_fused.hip.)
#include "fast_paths.h"

int fpl_fast_infer_volume(fpl_ctx *ctx, fpl_program *prog, const void *src,
                          int src_dtype, float mean, float sd,
                          const int64_t dims[3], const int32_t tile_in[3],
                          const int32_t offset[3], int precision,
                          const std::vector<int32_t> origins[3],
                          const int32_t out_sz[3], int32_t zb, int32_t ze,
                          float *dst, bool *handled) {
  *handled = false;
  return 0;
}
