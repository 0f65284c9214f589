// Launch-variant overrides for equivalence tests and tuning sweeps (see common.h: DebugOption).
#include "common.h"

namespace chitu {
int g_debug_options[kOptCount] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static_assert(kOptCount == 17, "one initialiser per option");
}

extern "C" int chitu_hip_debug_option(int32_t option, int32_t value) {
    CHITU_REQUIRE(option >= 0 && option < chitu::kOptCount && value >= -1);
    chitu::g_debug_options[option] = value;
    return CHITU_OK;
}

