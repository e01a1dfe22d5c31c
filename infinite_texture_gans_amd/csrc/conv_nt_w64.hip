// conv_nt_kernel instantiations with blocked fp64 accumulation (NT_W64, fp32 operands): the uniform-class GEMMs of the
// Winograd paths (conv_wino.hip; D's 256 -> 512 layer, reference models/discriminators.py:196-206), whose output
// transform amplifies the fp32 accumulation chain's rounding ~16 x.  See conv_nt_kernel.h.
#include "conv_nt_kernel.h"

namespace itgk {

int launch_nt_w64(int bco, int bpix, const ConvP& p, int k, hipStream_t s) {
  if (k != 16) return ITG_ERR_ARG;
  return launch_nt_shape_w64<NT_W64>(bco, bpix, p, k, s);
}

// ... and with the block sums in a second fp32 accumulator (NT_W32): the F(4 x 4, 2 x 2) forward GEMMs of D's stride-2 layers
int launch_nt_w32(int bco, int bpix, const ConvP& p, int k, hipStream_t s) {
  if (k != 16) return ITG_ERR_ARG;
  return launch_nt_shape_w64<NT_W32>(bco, bpix, p, k, s);
}

}  // namespace itgk
