// placeholder until the fused narrow path lands
#include "gnx_device.h"
namespace gnx {
int32_t launch_block_narrow(const gnx_graphs*, const BlockArgs&, int64_t, hipStream_t) { return 1; }
}
