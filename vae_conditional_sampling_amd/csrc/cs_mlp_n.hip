// Small-launch fused MLP kernel (one row tile per workgroup, output features split over the wavefronts), all widths.
#include "cs_mlp_narrow_kernel.inc"

bool cs_mlp_narrow_supported(const CsMlpArgs& a) {
  return a.net.wstream16h != nullptr && (cs_mlp_narrow_has<4>(a) || cs_mlp_narrow_has<8>(a) || cs_mlp_narrow_has<16>(a));
}

int cs_mlp_narrow_launch(const CsMlpArgs& a, hipStream_t stream) {
  switch (a.net.nt) {
    case 4: return cs_mlp_narrow_launch_nt<4>(a, stream);
    case 8: return cs_mlp_narrow_launch_nt<8>(a, stream);
    case 16: return cs_mlp_narrow_launch_nt<16>(a, stream);
  }
  return cs_fail(CS_EUNSUPPORTED, "no narrow kernel instance for hidden=" + std::to_string(a.net.nt * 16));
}
