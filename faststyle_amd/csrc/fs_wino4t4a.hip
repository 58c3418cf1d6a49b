// Instantiations of the register-fed Winograd F(4x4,3x3) kernel (fs_wino4t_kernel.h; description in fs_wino4t.hip): the FLATTENED 16-tile form
// (M = 4: items over the sample's row-major tile list, a 6 x 6 patch per tile) with the forward epilogues of the transform net's residual convs
// (im_transf_net.py:250-276) -- raw / per-item statistics, with and without the producer's instance norm + ReLU on load.
#include "fs_wino4t_kernel.h"

namespace fs {

#ifdef FS_WINO4T_TRACE
extern "C" int fs_debug_wino4t_trace_4a(long long* out, int n_wg) { return wino4t_trace_read(out, n_wg); }
#endif

int wino4t_launch_4a(const ConvArgs& a, int epi, long grid, hipStream_t s) { return wino4t_launch_part_fa<4>(a, epi, grid, s); }

}  // namespace fs
