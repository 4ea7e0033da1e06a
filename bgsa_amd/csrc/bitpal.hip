// bitpal.hip — BitPAl packed (2,-3,-5) (placeholder until the kernel lands).
#include "bgsa_common.h"
namespace bgsa {
const char *bitpal_kernel_name(int) { return "bitpal_kernel"; }
int launch_bitpal(const char *, const uint32_t *, int16_t *, int, int, int64_t, int, int, int, hipStream_t)
{
    set_error_text("bitpal: kernel not built yet");
    return BGSA_HIP_EUNSUPPORTED;
}
}  // namespace bgsa
