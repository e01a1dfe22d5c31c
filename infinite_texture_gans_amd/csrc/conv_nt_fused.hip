// conv_nt_kernel instantiations with the BatchNorm folded in (fp32 operands): NT_XF applies alpha * x + beta', the
// activation and the nearest x2 upsample in the tile loader (forward of reference models/layers.py:301-311 without the
// intermediate tensor), NT_BNS accumulates the BatchNorm backward sums in the input-gradient epilogue.
#include "conv_nt_kernel.h"

namespace itgk {

int launch_nt_fused(int mode, int bco, int bpix, const ConvP& p, int k, hipStream_t s) {
  if (k != 16) return ITG_ERR_ARG;
  return mode == NT_XF ? launch_nt_shape<NT_XF>(bco, bpix, p, k, s) : launch_nt_shape<NT_BNS>(bco, bpix, p, k, s);
}

}  // namespace itgk
