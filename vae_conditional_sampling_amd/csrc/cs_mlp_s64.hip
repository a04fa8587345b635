// Split-bf16 fused residual-MLP kernel instances for hidden width 64 (see cs_mlp_split_kernel.inc).
#include "cs_mlp_split_kernel.inc"

int cs_mlp_split_launch_h64(const CsMlpArgs& a, hipStream_t stream) { return cs_mlp_split_launch_nt<2>(a, stream); }
