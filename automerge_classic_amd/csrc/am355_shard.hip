// objectId sharding across GPUs (include/am355.h am355_set_shard / _export_fragment / _import_fragments). See am355_ctx.h.
#include "am355_ctx.h"

// ---------------------------------------------------------------------------------------------------------
// objectId sharding across GPUs. Every rank stages and decodes the whole batch (rows keep their global
// indexes, so op id -> row stays arithmetic) and merges only the objects it owns; the object table is identical on all
// ranks. What is exchanged is the OUTPUT: each rank's record tables (map records / edit records / values of its objects)
// as one contiguous fragment, all-gathered by the host binding over RCCL (xGMI) and stitched by object index.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct FragmentHeader {
  uint32_t magic, world, rank, n_objects, n_map, n_erecs, n_values, reserved;
  uint64_t off_objects, off_map, off_edits, total;
};
constexpr uint32_t FRAGMENT_MAGIC = 0x46333535;  // "553F"
inline size_t frag_align(size_t x) { return (x + 255) & ~(size_t)255; }
FragmentHeader fragment_layout(uint32_t world, uint32_t rank, const Counts& k) {
  FragmentHeader h{};
  h.magic = FRAGMENT_MAGIC; h.world = world; h.rank = rank;
  h.n_objects = k.n_objects; h.n_map = k.n_map_emit; h.n_erecs = k.n_erecs; h.n_values = k.n_edits;
  h.off_objects = frag_align(sizeof(FragmentHeader));
  h.off_map = h.off_objects + frag_align((size_t)h.n_objects * sizeof(am355_ir_object));
  h.off_edits = h.off_map + frag_align((size_t)h.n_map * sizeof(am355_ir_map));
  h.total = h.off_edits + frag_align(((size_t)h.n_erecs + 1) * sizeof(am355_ir_edit));
  return h;
}
}  // namespace

extern "C" int am355_set_shard(am355_ctx* c, uint32_t rank, uint32_t world) {
  if (!c || world == 0 || rank >= world) return c ? fail(c, AM355_E_ARG, "bad shard (rank %u of %u)", rank, world) : AM355_E_ARG;
  c->shard_rank = rank;
  c->shard_world = world;
  c->replayed = c->ir_fetched = false;
  return AM355_OK;
}

extern "C" int am355_fragment_size(am355_ctx* c, size_t* bytes) {
  if (!c || !bytes) return AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  *bytes = (size_t)fragment_layout(c->shard_world, c->shard_rank, c->counts).total;
  return AM355_OK;
}

extern "C" int am355_export_fragment(am355_ctx* c, void* dst, size_t cap, int dst_is_device, size_t* len) {
  if (!c || !dst || !len) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  (void)hipSetDevice(c->device);
  FragmentHeader h = fragment_layout(c->shard_world, c->shard_rank, c->counts);
  if (h.total > cap) return fail(c, AM355_E_ARG, "fragment needs %llu bytes, buffer has %llu", (unsigned long long)h.total, (unsigned long long)cap);
  hipStream_t st = c->stream;
  uint8_t* d = (uint8_t*)dst;
  hipMemcpyKind from_dev = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (!c->h_words.ensure(sizeof(FragmentHeader) + 64)) return fail(c, AM355_E_NOMEM, "host allocation failed");
  FragmentHeader* hh = (FragmentHeader*)(c->h_words.as<uint8_t>() + 64);  // (pinned; the first 64 bytes are the replay's words)
  *hh = h;
  HIPCHK(c, hipMemcpyAsync(d, hh, sizeof h, dst_is_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost, st));
  if (h.n_objects) HIPCHK(c, hipMemcpyAsync(d + h.off_objects, c->ir.obj, (size_t)h.n_objects * sizeof(am355_ir_object), from_dev, st));
  if (h.n_map) HIPCHK(c, hipMemcpyAsync(d + h.off_map, c->ir.map, (size_t)h.n_map * sizeof(am355_ir_map), from_dev, st));
  HIPCHK(c, hipMemcpyAsync(d + h.off_edits, c->ir.edit, ((size_t)h.n_erecs + 1) * sizeof(am355_ir_edit), from_dev, st));
  HIPCHK(c, hipStreamSynchronize(st));
  *len = (size_t)h.total;
  return AM355_OK;
}

// frags: `world` fragments back to back, fragment r = frags[offsets[r] .. offsets[r+1]) (host memory). Builds the patch IR of
// the whole document in this context (the one whose am355_patch_json / am355_fetch_ir the caller then uses: it must have
// replayed the same batch, its envelope and arena serve the stitched patch).
int import_fragments_impl(am355_ctx* c, const uint8_t* frags, const uint64_t* offsets, uint32_t world) {
  if (!c || !frags || !offsets || !world) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  std::vector<FragmentHeader> hs(world);
  uint64_t n_map = 0, n_erecs = 0, n_values = 0;
  for (uint32_t r = 0; r < world; r++) {
    if (offsets[r + 1] < offsets[r] || offsets[r + 1] - offsets[r] < sizeof(FragmentHeader)) return fail(c, AM355_E_ARG, "fragment %u too short", r);
    memcpy(&hs[r], frags + offsets[r], sizeof(FragmentHeader));
    const FragmentHeader& h = hs[r];
    Counts k{};
    k.n_objects = h.n_objects; k.n_map_emit = h.n_map; k.n_erecs = h.n_erecs; k.n_edits = h.n_values;
    FragmentHeader want = fragment_layout(world, r, k);
    if (h.magic != FRAGMENT_MAGIC || h.world != world || h.rank != r || h.total != want.total || h.total > offsets[r + 1] - offsets[r] ||
        h.n_objects != hs[0].n_objects)
      return fail(c, AM355_E_ARG, "fragment %u is malformed or from another batch", r);
    n_map += h.n_map; n_erecs += (uint64_t)h.n_erecs + 1; n_values += h.n_values;
  }
  if (n_map >= 0xfffffff0ull || n_erecs >= 0xfffffff0ull || n_values >= 0xfffffff0ull) return fail(c, AM355_E_UNSUPPORTED, "stitched patch too large");
  const uint32_t NO = hs[0].n_objects;
  size_t o_obj = 0, o_map = o_obj + frag_align((size_t)NO * sizeof(am355_ir_object)), o_edit = o_map + frag_align(n_map * sizeof(am355_ir_map)),
         total = o_edit + frag_align(n_erecs * sizeof(am355_ir_edit));
  c->stitched.assign(total + 64, 0);
  uint8_t* base = c->stitched.data();
  base += (64 - ((uintptr_t)base & 63)) & 63;
  am355_ir_object* obj = (am355_ir_object*)(base + o_obj);
  am355_ir_map* map = (am355_ir_map*)(base + o_map);
  am355_ir_edit* edit = (am355_ir_edit*)(base + o_edit);
  std::vector<uint32_t> map_base(world), edit_base(world), val_base(world);
  uint32_t mb = 0, eb = 0, vb = 0;
  for (uint32_t r = 0; r < world; r++) {
    const FragmentHeader& h = hs[r];
    const uint8_t* f = frags + offsets[r];
    map_base[r] = mb; edit_base[r] = eb; val_base[r] = vb;
    if (h.n_map) memcpy(map + mb, f + h.off_map, (size_t)h.n_map * sizeof(am355_ir_map));
    memcpy(edit + eb, f + h.off_edits, ((size_t)h.n_erecs + 1) * sizeof(am355_ir_edit));
    for (uint32_t k = 0; k <= h.n_erecs; k++) edit[eb + k].first += vb;  // (the sentinel of rank r then points at rank r+1's first value)
    mb += h.n_map; eb += h.n_erecs + 1; vb += h.n_values;
  }
  // object table: identical on every rank but for the ranges, which the owner knows
  const am355_ir_object* obj0 = (const am355_ir_object*)(frags + offsets[0] + hs[0].off_objects);
  for (uint32_t oi = 0; oi < NO; oi++) {
    uint32_t owner = oi == 0 ? 0u : shard_owner(obj0[oi].id_actor, obj0[oi].id_ctr, world);
    const am355_ir_object& src = ((const am355_ir_object*)(frags + offsets[owner] + hs[owner].off_objects))[oi];
    if (src.id_ctr != obj0[oi].id_ctr || src.id_actor != obj0[oi].id_actor || src.map_end > hs[owner].n_map || src.edit_end > hs[owner].n_erecs)
      return fail(c, AM355_E_ARG, "fragments disagree on object %u", oi);
    obj[oi] = src;
    obj[oi].map_begin += map_base[owner]; obj[oi].map_end += map_base[owner];
    obj[oi].edit_begin += edit_base[owner]; obj[oi].edit_end += edit_base[owner];
  }
  int rc = fetch_ir_impl(c, nullptr);  // envelope (clock, heads, actors, arena) of this context
  if (rc) return rc;
  am355_patch_ir& h = c->hir;
  h.n_objects = NO; h.n_map = mb; h.n_edits = eb; h.n_values = vb;
  h.objects = obj; h.map = map; h.edits = edit;
  // (edit record eb is never read: the last fragment's own sentinel is record eb - 1)
  return AM355_OK;
}

