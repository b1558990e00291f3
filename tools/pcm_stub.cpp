// tools/pcm_stub.cpp -- development builds only (tools/variant.sh): stands in for the 13 MB object with the planar-PCM
// twins of the kernels, so that a one-kernel tuning variant is a couple of megabytes on its way to the GPU box.  Every
// entry throws: tuning runs use fp64 buffers.
#include <stdexcept>
#include "r8b_launch.h"
namespace r8bhip {
#define STUB(name, T) void name(const T&, void*) { throw std::runtime_error(#name ": development build without the PCM kernels"); }
#define STUBM(name, T) void name(const T&, int, void*) { throw std::runtime_error(#name ": development build without the PCM kernels"); }
STUB(launch_conv_pcm, ConvLaunch) STUB(launch_hbup_pcm, HBLaunch) STUB(launch_poly_pcm, PolyLaunch)
STUB(launch_tail_pcm, TailLaunch) STUBM(launch_convp_pcm, ConvxLaunch) STUBM(launch_convx_pcm, ConvxLaunch)
STUB(launch_whole_pcm, WholeLaunch) STUB(launch_hbdown_pcm, HBLaunch) STUB(launch_hbcascade_pcm, HBCascadeLaunch)
STUB(launch_hbdcascade_pcm, HBCascadeLaunch)
}
