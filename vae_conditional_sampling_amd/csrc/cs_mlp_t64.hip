// 16-row-tile split-bf16 fused MLP kernel, hidden width 64.
#include "cs_mlp_split16_kernel.inc"

int cs_mlp_split16_launch_h64(const CsMlpArgs& a, hipStream_t stream, int waves, int terms) {
  return cs_mlp_split16_launch_nt<4>(a, stream, waves, terms);
}
