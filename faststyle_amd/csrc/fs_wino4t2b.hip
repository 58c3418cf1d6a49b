// Instantiations of the register-fed Winograd F(4x4,3x3) kernel (fs_wino4t_kernel.h; description in fs_wino4t.hip): 32-tile items with the residual-gradient / bias + ReLU + pool / consumer-mask epilogues -- the VGG16 convs of fs_perceptual_loss and their input gradients.
#include "fs_wino4t_kernel.h"

namespace fs {

#ifdef FS_WINO4T_TRACE
extern "C" int fs_debug_wino4t_trace_2b(long long* out, int n_wg) { return wino4t_trace_read(out, n_wg); }
#endif

int wino4t_launch_2b(const ConvArgs& a, int epi, long grid, hipStream_t s) { return wino4t_launch_part_b<2>(a, epi, grid, s); }

}  // namespace fs
