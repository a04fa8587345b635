// Split-bf16 fused residual-MLP kernel instances for hidden width 256 (see cs_mlp_split_kernel.inc).
#include "cs_mlp_split_kernel.inc"

int cs_mlp_split_launch_h256(const CsMlpArgs& a, hipStream_t stream) { return cs_mlp_split_launch_nt<8>(a, stream); }
