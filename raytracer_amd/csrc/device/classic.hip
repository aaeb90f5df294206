// classic.hip — placeholder launchers, replaced below by the per-material kernels.
#include "device_util.h"
namespace mirt {
void launch_connect(const LaunchCfg &, const DScene &, const DState &, const DFrameParams &) {}
void launch_shade_classic(const LaunchCfg &, const DScene &, const DState &, const DFrameParams &) {}
}  // namespace mirt
