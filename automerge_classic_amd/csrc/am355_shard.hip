// objectId sharding across GPUs (include/am355.h am355_set_shard / _export_fragment / _import_fragments). See am355_ctx.h.
#include "am355_ctx.h"
#include <dlfcn.h>

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
  c->h_tables_current = false;
  c->resident_valid = false;
  return AM355_OK;
}

extern "C" int am355_fragment_size(am355_ctx* c, size_t* bytes) {
  if (!c || !bytes) return AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  { int frc = ensure_ir_fresh(c); if (frc) return frc; }
  *bytes = (size_t)fragment_layout(c->shard_world, c->shard_rank, c->counts).total;
  return AM355_OK;
}

extern "C" int am355_export_fragment(am355_ctx* c, void* dst, size_t cap, int dst_is_device, size_t* len) {
  if (!c || !dst || !len) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must succeed first");
  (void)hipSetDevice(c->device);
  { int frc = ensure_ir_fresh(c); if (frc) return frc; }
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


// ---------------------------------------------------------------------------------------------------------
// The collective inside the library: RCCL over xGMI (include/am355.h am355_shard_init / am355_sharded_replay)
// ---------------------------------------------------------------------------------------------------------
namespace {
// The five entry points of RCCL this path calls, as nccl.h declares them (this image ships librccl.so.1 without its header;
// NCCL's C ABI: ncclUniqueId is 128 opaque bytes passed BY VALUE to ncclCommInitRank, ncclUint8 == 1, ncclSuccess == 0).
struct RcclUniqueId { char internal[AM355_SHARD_ID_BYTES]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(void** comm, int nranks, RcclUniqueId id, int rank) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t st) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
constexpr int RCCL_UINT8 = 1;

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, []() {
    const char* env = getenv("AM355_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
      api.err = dlerror();
      if (n == env) break;   // (an explicit name that does not load is an error, not a reason to try another library)
    }
    if (!api.lib) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) { api.err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy"; api.lib = nullptr; }
  });
  return &api;
}
const char* rccl_error(int rc) {
  RcclApi* a = rccl();
  return a->GetErrorString ? a->GetErrorString(rc) : "RCCL error";
}
#define RCCLCHK(ctx, call)                                                                                  \
  do {                                                                                                      \
    int r_ = (call);                                                                                        \
    if (r_ != 0) return fail(ctx, AM355_E_DEVICE, "%s: %s", #call, rccl_error(r_));                         \
  } while (0)
}  // namespace

extern "C" int am355_shard_unique_id(uint8_t id[AM355_SHARD_ID_BYTES]) {
  if (!id) return AM355_E_ARG;
  RcclApi* a = rccl();
  if (!a->lib) return AM355_E_DEVICE;
  RcclUniqueId u{};
  if (a->GetUniqueId(&u) != 0) return AM355_E_DEVICE;
  memcpy(id, u.internal, AM355_SHARD_ID_BYTES);
  return AM355_OK;
}

extern "C" int am355_shard_finalize(am355_ctx* c) {
  if (!c) return AM355_E_ARG;
  if (c->shard_comm) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)rccl()->CommDestroy(c->shard_comm);
    c->shard_comm = nullptr;
  }
  c->shard_fragment_bytes.clear();
  return am355_set_shard(c, 0, 1);
}

extern "C" int am355_shard_init(am355_ctx* c, const uint8_t id[AM355_SHARD_ID_BYTES], uint32_t rank, uint32_t world) {
  if (!c || !id || world == 0 || rank >= world) return c ? fail(c, AM355_E_ARG, "bad shard (rank %u of %u)", rank, world) : AM355_E_ARG;
  RcclApi* a = rccl();
  if (!a->lib) return fail(c, AM355_E_DEVICE, "RCCL is not available: %s", a->err.c_str());
  (void)hipSetDevice(c->device);
  if (c->shard_comm) { int frc = am355_shard_finalize(c); if (frc) return frc; }
  RcclUniqueId u{};
  memcpy(u.internal, id, AM355_SHARD_ID_BYTES);
  void* comm = nullptr;
  RCCLCHK(c, a->CommInitRank(&comm, (int)world, u, (int)rank));
  c->shard_comm = comm;
  return am355_set_shard(c, rank, world);
}

extern "C" int am355_shard_fragment_bytes(am355_ctx* c, uint64_t* bytes, uint32_t capacity) {
  if (!c || !bytes) return AM355_E_ARG;
  if (c->shard_fragment_bytes.size() > capacity) return fail(c, AM355_E_ARG, "%zu ranks, room for %u", c->shard_fragment_bytes.size(), capacity);
  for (size_t r = 0; r < c->shard_fragment_bytes.size(); r++) bytes[r] = c->shard_fragment_bytes[r];
  return AM355_OK;
}

extern "C" int am355_sharded_replay(am355_ctx* c, int stitch_on_all_ranks) {
  if (!c) return AM355_E_ARG;
  if (!c->shard_comm) return fail(c, AM355_E_STATE, "am355_shard_init must succeed first");
  return guarded(c, [&]() -> int {
    (void)hipSetDevice(c->device);
    RcclApi* a = rccl();
    hipStream_t st = c->stream;
    const uint32_t world = c->shard_world, rank = c->shard_rank;
    // this rank's part; a failure is carried into the first collective, which every rank enters whatever happened to it
    int own_rc = c->staged ? replay_impl(c) : fail(c, AM355_E_STATE, "am355_load_changes / am355_load_document must be called first");
    const std::string own_err = c->err;
    const uint32_t own_flags = c->flags;
    size_t need = 0;
    if (!own_rc) own_rc = am355_fragment_size(c, &need);
    if (!c->d_shard_sizes.ensure(16 * ((size_t)world + 1)) || !c->h_shard_sizes.ensure(16 * ((size_t)world + 1))) return fail(c, AM355_E_NOMEM, "allocation failed (shard sizes)");
    uint64_t* h_sizes = c->h_shard_sizes.as<uint64_t>();        // [0..1]: mine {failed, bytes}; [2 ..]: everybody's
    uint64_t* d_sizes = c->d_shard_sizes.as<uint64_t>();
    h_sizes[0] = own_rc ? 1 : 0;
    h_sizes[1] = need;
    HIPCHK(c, hipMemcpyAsync(d_sizes, h_sizes, 16, hipMemcpyHostToDevice, st));
    RCCLCHK(c, a->AllGather(d_sizes, d_sizes + 2, 16, RCCL_UINT8, c->shard_comm, st));
    HIPCHK(c, hipMemcpyAsync(h_sizes + 2, d_sizes + 2, 16 * (size_t)world, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->shard_fragment_bytes.assign(world, 0);
    uint64_t cap = 0;
    int first_failed = -1;
    for (uint32_t r = 0; r < world; r++) {
      if (h_sizes[2 + 2 * r] && first_failed < 0) first_failed = (int)r;
      c->shard_fragment_bytes[r] = h_sizes[2 + 2 * r + 1];
      cap = std::max<uint64_t>(cap, h_sizes[2 + 2 * r + 1]);
    }
    if (first_failed >= 0) {
      c->replayed = false;
      if (own_rc) { c->err = own_err; c->flags = own_flags; return own_rc; }
      return fail(c, AM355_E_INVALID, "rank %d rejected the batch", first_failed);
    }
    // the data-path collective: every rank's fragment, HBM -> HBM over xGMI, at one stride
    const size_t stride = frag_align((size_t)cap);
    if (!c->d_shard_send.ensure(stride) || !c->d_shard_recv.ensure(stride * (size_t)world)) return fail(c, AM355_E_NOMEM, "device allocation failed (shard fragments)");
    size_t wrote = 0;
    int erc = am355_export_fragment(c, c->d_shard_send.p, stride, 1, &wrote);
    if (erc) return erc;
    RCCLCHK(c, a->AllGather(c->d_shard_send.p, c->d_shard_recv.p, stride, RCCL_UINT8, c->shard_comm, st));
    if (rank != 0 && !stitch_on_all_ranks) { HIPCHK(c, hipStreamSynchronize(st)); return AM355_OK; }
    if (!c->h_shard_frags.ensure(stride * (size_t)world)) return fail(c, AM355_E_NOMEM, "host allocation failed (shard fragments)");
    HIPCHK(c, hipMemcpyAsync(c->h_shard_frags.p, c->d_shard_recv.p, stride * (size_t)world, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    std::vector<uint64_t> offsets((size_t)world + 1);
    for (uint32_t r = 0; r <= world; r++) offsets[r] = (uint64_t)r * stride;   // (fragment r fills the front of its stride)
    return import_fragments_impl(c, c->h_shard_frags.as<uint8_t>(), offsets.data(), world);
  });
}
