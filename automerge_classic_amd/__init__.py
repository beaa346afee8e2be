"""MI355X-native bulk change-replay engine behind automerge-classic's Backend API.

Only the hot path named in BASELINE.json is implemented here: binary change decode,
causal scheduling, per-object op-set merge (RGA list order, multi-value registers) and whole-document
patch generation, as hand-written HIP kernels for gfx950 behind the C ABI declared in include/am355.h.
"""
