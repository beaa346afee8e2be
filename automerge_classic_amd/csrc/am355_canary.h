// Debug mode AM355_CANARY=1 (VERDICT r3 "find the crash"): every carve-out of the device blocks gets a red zone behind it -- the
// slack up to the 256-byte boundary plus 256 more bytes -- filled with a pattern when it is carved and verified by one kernel at
// the end of every API call that launches kernels (am355_replay, am355_apply_changes, am355_load_document, am355_save,
// am355_doc_changes). A kernel that writes past its array therefore fails the CALL THAT DID IT, with the name of the block and the
// number of the carve-out, instead of corrupting the neighbour and faulting somewhere else some of the time.
// Off (the default): carve_round() is the plain 256-byte rounding and nothing else runs.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "am355_device.h"

namespace am355 {

bool canary_on();
// bytes a carve-out of `bytes` occupies in its block (every *_bytes() / *_carve() pair of the engine rounds with this)
static inline size_t carve_round(size_t bytes) {
  size_t r = (bytes + 255) & ~(size_t)255;
  return canary_on() ? r + 256 : r;
}
// names the block the following carve-outs belong to (restarts their numbering)
void canary_scope(const char* block);
// [base + used, base + carve_round(used)) is a red zone from now on
void canary_note(const void* base, size_t used);
// the device range [base, base + bytes) is about to be freed or carved again from scratch: its zones are forgotten
void canary_forget(const void* base, size_t bytes);
// [base, base + bytes) is written as ONE range on purpose (several carve-outs cleared by one fill): the zones inside it are dropped,
// a zone that straddles its end is cut down to what lies behind it
void canary_allow(const void* base, size_t bytes);
// fills the zones noted since the last call (device-wide synchronisation: debug mode only)
void canary_arm();
// verifies every zone; false: `msg` names the first damaged one. Synchronises the device.
bool canary_check(char* msg, size_t msg_len);

}  // namespace am355
