// Stage 1 launchers (see am355_decode.hip).
#pragma once
#include "am355_internal.h"

namespace am355 {
void launch_parse_changes(const uint8_t* arena, const uint64_t* offsets, uint32_t n_changes, ChangeMeta* metas, hipStream_t st);
void launch_decode_columns(const uint8_t* arena, const ChangeMeta* metas, const ChangePlan* plans, uint32_t n_plans, const uint32_t* amap,
                           OpCols cols, uint32_t* flags, hipStream_t st);
}  // namespace am355
