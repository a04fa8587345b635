// Fused residual-MLP kernel instances for hidden width 256 (see cs_mlp_kernel.inc).
#include "cs_mlp_kernel.inc"

int cs_mlp_launch_h256(const CsMlpArgs& a, hipStream_t stream) { return cs_mlp_launch_nt<16>(a, stream); }
