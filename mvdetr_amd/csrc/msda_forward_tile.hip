// LDS-tiled encoder forward (placeholder until the tiled kernel lands: never selected).
#include "common.h"
#include "msda_dispatch.h"

namespace mvdetr {

bool msda_tile_supported(int, int, int, int, int, int, int, bool) { return false; }

int msda_forward_tile(hipStream_t, const float *, const int64_t *, const int64_t *, const float *,
                      const float *, int, int, int, int, int, int, int, float *)
{
    return (int)hipErrorNotSupported;
}

}  // namespace mvdetr
