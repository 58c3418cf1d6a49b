// Instantiations of the register-fed Winograd F(4x4,3x3) kernel (fs_wino4t_kernel.h; description in fs_wino4t.hip): 16-tile items whose epilogue
// also leaves the instance-norm-backward partial sums of the unit whose output gradient the launch writes (EPI 5: raw, EPI 6: + the residual
// gradient) -- the transform net's residual input gradients (im_transf_net.py:250-276 adjoint, fs_tnet_backward), round 5.
#include "fs_wino4t_kernel.h"

namespace fs {

#ifdef FS_WINO4T_TRACE
extern "C" int fs_debug_wino4t_trace_1d(long long* out, int n_wg) { return wino4t_trace_read(out, n_wg); }
#endif

int wino4t_launch_1d(const ConvArgs& a, int epi, long grid, hipStream_t s) { return wino4t_launch_part_d<1>(a, epi, grid, s); }

}  // namespace fs
