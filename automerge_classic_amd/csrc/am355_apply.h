// Host side of the incremental patch of Backend.applyChanges: setupPatches (backend/new.js:1461-1528) over the
// object table, and the assembly of the patch's record tables from the device's delta tables (am355_delta.hip) and the
// whole-document tables of the same replay.
#pragma once
#include "am355_delta.h"

#include <string>
#include <unordered_map>
#include <vector>

namespace am355 {

struct ApplyPatch {
  std::vector<am355_ir_object> objects;
  std::vector<am355_ir_map> map;
  std::vector<am355_ir_edit> edits;  // + sentinel
  am355_patch_ir ir{};               // points into the vectors above and into the envelope / arena of the whole-document IR
};

// whole: the whole-document IR of the new state (host).  link / d_map / d_edits: the delta tables copied from the device.
// Returns 0, or AM355_E_UNSUPPORTED with *err set (an edit inside an object the document no longer reaches, where what the children
// tables of the reference's objectMeta hold could not be told).
// known: what objectMeta.children holds for the property of some objects (object index -> KeyHistory, am355_delta.h), found by the
// device from the history of the rows on it; need: filled with the objects whose property the walk had to know about and did not (the
// caller asks the device and calls again). With need empty on return the result is final. ask_always: the state has seen a merge call
// that skipped values of a property with a child object -- no property is taken to list its visible values without asking.
int assemble_apply_patch(const am355_patch_ir& whole, const ObjLink* link, const am355_ir_map* d_map, uint32_t n_dmap, const am355_ir_edit* d_edits,
                         uint32_t n_dedits, const std::unordered_map<uint32_t, KeyHistory>& known, bool ask_always, std::vector<uint32_t>& need, ApplyPatch& out,
                         std::string& err);

}  // namespace am355
