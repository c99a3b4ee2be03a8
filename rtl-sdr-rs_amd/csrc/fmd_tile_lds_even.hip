// fmd_tile_lds_even.hip -- LDS-DMA demodulation kernels for the even downsample factors 2 ... 14 (whole-dword windows, adjacent-window rounds).
// (device code: fmd_tile_body.h; launcher: fmd_tile_launch.hip)
#include "fmd_tile_body.h"

namespace fmd_tk {
template void launch_lds<1>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<2>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<3>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<4>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<5>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<6>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<7>(const FmdLaunch&, dim3, size_t, hipStream_t);
}  // namespace fmd_tk
