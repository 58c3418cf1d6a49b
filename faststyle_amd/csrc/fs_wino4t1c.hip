// Instantiations of the register-fed Winograd F(4x4,3x3) kernel (fs_wino4t_kernel.h; description in fs_wino4t.hip): the 128-channel item form
// (16 tiles x two channel blocks per wave) with the VGG16 epilogues -- the layers of fs_perceptual_loss with >= 128 output channels.
#include "fs_wino4t_kernel.h"

namespace fs {

#ifdef FS_WINO4T_TRACE
extern "C" int fs_debug_wino4t_trace_1c(long long* out, int n_wg) { return wino4t_trace_read(out, n_wg); }
#endif

int wino4t_launch_1c(const ConvArgs& a, int epi, long grid, hipStream_t s) { return wino4t_launch_part_c<3>(a, epi, grid, s); }

}  // namespace fs
