// Instantiations of the register-fed Winograd F(4x4,3x3) kernel (fs_wino4t_kernel.h; description in fs_wino4t.hip): the FLATTENED 16-tile form
// (M = 4) with the input-gradient epilogues of the transform net's residual convs -- the residual gradient added in the interior (EPI 2), and the
// two forms that also leave the instance-norm-backward partial sums of the unit below (EPI 5 / 6).
#include "fs_wino4t_kernel.h"

namespace fs {

#ifdef FS_WINO4T_TRACE
extern "C" int fs_debug_wino4t_trace_4b(long long* out, int n_wg) { return wino4t_trace_read(out, n_wg); }
#endif

int wino4t_launch_4b(const ConvArgs& a, int epi, long grid, hipStream_t s) { return wino4t_launch_part_fb<4>(a, epi, grid, s); }

}  // namespace fs
