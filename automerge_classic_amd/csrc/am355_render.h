// Host-side rendering of the patch IR as JSON text identical to JSON.stringify(Backend.getPatch(state)).
#pragma once
#include "../../include/am355.h"
#include <string>

namespace am355 {
bool render_patch_json(const am355_patch_ir& ir, std::string& out, std::string& err);
}
