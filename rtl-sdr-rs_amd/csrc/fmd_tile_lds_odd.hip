// fmd_tile_lds_odd.hip -- LDS-DMA demodulation kernels for the odd downsample factors 1 ... 31 (masked-window rounds).
// (device code: fmd_tile_body.h; launcher: fmd_tile_launch.hip)
#include "fmd_tile_body.h"

namespace fmd_tk {
template void launch_lds<-1>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-3>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-5>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-7>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-9>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-11>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-13>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-15>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-17>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-19>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-21>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-23>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-25>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-27>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-29>(const FmdLaunch&, dim3, size_t, hipStream_t);
template void launch_lds<-31>(const FmdLaunch&, dim3, size_t, hipStream_t);
}  // namespace fmd_tk
