// placeholder until the MFMA wide path lands
#include "gnx_device.h"
namespace gnx {
int32_t launch_block_wide(const gnx_graphs*, const BlockArgs&, int64_t, hipStream_t) { return 1; }
size_t wide_workspace_bytes(const gnx_graphs*, const gnx_block_params*, int64_t) { return 0; }
}
