// Split-bf16 fused residual-MLP kernel instances for hidden width 128 (see cs_mlp_split_kernel.inc).
#include "cs_mlp_split_kernel.inc"

int cs_mlp_split_launch_h128(const CsMlpArgs& a, hipStream_t stream) { return cs_mlp_split_launch_nt<4>(a, stream); }
