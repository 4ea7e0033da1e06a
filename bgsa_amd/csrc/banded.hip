// banded.hip — banded Myers filter (placeholder until the kernel lands).
#include "bgsa_common.h"
namespace bgsa {
const char *banded_kernel_name(int) { return "banded_kernel"; }
int launch_banded(const char *, const uint32_t *, int8_t *, int, int, int64_t, int, int, int, int, hipStream_t)
{
    set_error_text("banded: kernel not built yet");
    return BGSA_HIP_EUNSUPPORTED;
}
}  // namespace bgsa
