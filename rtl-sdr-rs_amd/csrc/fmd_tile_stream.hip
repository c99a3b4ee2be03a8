// fmd_tile_stream.hip -- register-streaming demodulation kernels (downsample 2 and 4, banks of >= 8 channels in one phase class).
// (device code: fmd_tile_body.h; launcher: fmd_tile_launch.hip)
#include "fmd_tile_body.h"

namespace fmd_tk {
template void launch_stream<1>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_stream<2>(const FmdLaunch&, dim3, size_t, hipStream_t);
}  // namespace fmd_tk
