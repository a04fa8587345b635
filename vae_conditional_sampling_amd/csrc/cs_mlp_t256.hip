// 16-row-tile split-bf16 fused MLP kernel, hidden width 256.
#include "cs_mlp_split16_kernel.inc"

int cs_mlp_split16_launch_h256(const CsMlpArgs& a, hipStream_t stream, int waves, int terms) {
  return cs_mlp_split16_launch_nt<16>(a, stream, waves, terms);
}
