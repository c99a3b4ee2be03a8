// fmd_tile_lds_wide.hip -- LDS-DMA demodulation kernels for downsample 16 ... 32 (even), 64, 128 and the catch-all (33 ... 127).
// (device code: fmd_tile_body.h; launcher: fmd_tile_launch.hip)
#include "fmd_tile_body.h"

namespace fmd_tk {
template void launch_lds<8>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<9>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<10>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<11>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<12>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<13>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<14>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<15>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<16>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<32>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<64>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<0>(const FmdLaunch&, dim3, size_t, hipStream_t);
}  // namespace fmd_tk
