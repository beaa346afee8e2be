// Host side of the incremental patch of Backend.applyChanges: see am355_apply.h.
//
// setupPatches (backend/new.js:1461-1528) links every object the batch touched to its parent, and that one to its parent, until it
// reaches _root or an object whose patch already lists the child: in a map parent `props[key]` gets every value of
// objectMeta.children[key] it does not hold yet, in a list parent the element gets one `update` edit per value at the index the
// element has at the end of the call.  children[key] is the set of visible `set` / make values of the property as of the last
// call that looked at it (new.js:916-931) -- for a property that holds a VISIBLE child object that is its current visible set
// (am355_delta.hip refuses the one case where the merge loop skips values of such a property), which the whole-document tables of
// the same replay list.  A touched object whose make op has a successor is refused: whether children[key] is empty or still
// lists other values then depends on calls this engine has no record of.
#include "am355_apply.h"

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <unordered_set>

namespace am355 {
namespace {

inline uint32_t utf16_order_byte_host(uint32_t x) {
  if (x >= 0xf0 && x <= 0xf4) return x - 2;
  if (x == 0xee || x == 0xef) return x + 5;
  return x;
}
// order of the whole-document map records inside one object (am355_merge.hip map_emission_less / the LSD passes)
inline int key_cmp(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
  uint32_t n = la < lb ? la : lb;
  for (uint32_t k = 0; k < n; k++) {
    uint32_t x = utf16_order_byte_host(a[k]), y = utf16_order_byte_host(b[k]);
    if (x != y) return x < y ? -1 : 1;
  }
  return la < lb ? -1 : la > lb;
}

struct KeyRef {
  uint32_t oi;
  const uint8_t* p;
  uint32_t len;
  bool operator==(const KeyRef& o) const { return oi == o.oi && len == o.len && memcmp(p, o.p, len) == 0; }
};
struct KeyRefHash {
  size_t operator()(const KeyRef& k) const {
    uint64_t h = 0xcbf29ce484222325ull ^ k.oi;
    for (uint32_t i = 0; i < k.len; i++) h = (h ^ k.p[i]) * 0x100000001b3ull;
    return (size_t)h;
  }
};

}  // namespace

int assemble_apply_patch(const am355_patch_ir& whole, const ObjLink* link, const am355_ir_map* d_map, uint32_t n_dmap, const am355_ir_edit* d_edits,
                         uint32_t n_dedits, const std::unordered_map<uint32_t, KeyHistory>& known, bool ask_always, std::vector<uint32_t>& need, ApplyPatch& out,
                         std::string& err) {
  need.clear();
  const uint32_t NO = whole.n_objects;
  const uint8_t* arena = whole.arena;
  // ---- objectIds: the objects the batch touched, in the order it first touched them ----
  std::vector<uint32_t> touched;
  for (uint32_t oi = 0; oi < NO; oi++)
    if (link[oi].touch != NONE32) touched.push_back(oi);
  std::sort(touched.begin(), touched.end(), [&](uint32_t a, uint32_t b) { return link[a].touch < link[b].touch; });

  std::unordered_map<uint32_t, std::vector<am355_ir_map>> extra_map;
  std::unordered_map<uint32_t, std::vector<am355_ir_edit>> extra_edits;
  std::unordered_set<KeyRef, KeyRefHash> present_keys;   // keys of the delta tables (of parents looked at) and appended ones
  std::unordered_set<uint32_t> keys_indexed;             // parents whose delta keys are in present_keys
  std::unordered_set<uint64_t> appended_elems;           // (parent, element id) that received update edits -- key below
  std::unordered_map<uint32_t, std::unordered_map<uint64_t, uint32_t>> elem_index;  // list parent -> element id -> first whole-document record
  auto elem_key = [](uint32_t ctr, uint32_t actor) { return (uint64_t)ctr << 32 | actor; };
  // list parent -> op ids of the batch's own edits that carry one (insert / update: new.js:1477-1481 looks for them)
  std::unordered_map<uint32_t, std::unordered_set<uint64_t>> edit_ops;
  auto batch_edit_ops = [&](uint32_t o) -> const std::unordered_set<uint64_t>& {
    auto it = edit_ops.find(o);
    if (it != edit_ops.end()) return it->second;
    auto& set = edit_ops[o];
    for (uint32_t r = link[o].edit_begin; r < link[o].edit_end && r < n_dedits; r++) {
      const am355_ir_edit& ed = d_edits[r];
      if (ed.flags & (AM355_EDIT_REMOVE | AM355_EDIT_CONT | AM355_EDIT_MULTI)) continue;
      if (d_edits[r + 1].first - ed.first != 1 || (r + 1 < link[o].edit_end && (d_edits[r + 1].flags & AM355_EDIT_CONT))) continue;  // multi-insert: no opId
      set.insert(elem_key(ed.id_ctr, ed.id_actor));
    }
    return set;
  };

  for (uint32_t start : touched) {
    uint32_t o = start, child = NONE32;
    bool patch_exists = false;
    for (;;) {
      if (child != NONE32) {
        const ObjLink& L = link[child];
        if (o >= NO) { err = "internal: parent index out of range"; return AM355_E_DEVICE; }
        // A property the batch itself touched is already in the patch with everything that is visible, and children[key] was
        // refreshed by that visit (or is empty): setupPatches adds nothing and stops here whatever became of the child. Otherwise
        // children[key] is what an earlier call left, which is the visible set only while the child is visible.
        // children[key] of a property that holds plain values only is what the last visit left: those values, or nothing for good
        // (new.js:916-931) -- the device tells which from the rows on the property (delta_key_history). 0: go on, `dead` says how.
        // `listed`: the op ids children[key] lists (KH_LIVE)
        const KeyHistory* listed = nullptr;
        auto stale = [&](bool& dead) {
          auto it = known.find(child);
          if (it == known.end()) { need.push_back(child); dead = true; return AM355_OK; }  // (asked for; this pass goes on as if nothing were listed)
          if (it->second.state == KH_DEAD) { dead = true; return AM355_OK; }
          if (it->second.state == KH_LIVE) { dead = false; listed = &it->second; return AM355_OK; }
          err = "unsupported: the batch edits an object that is no longer a visible value of its parent (objectMeta history)";
          return AM355_E_UNSUPPORTED;
        };
        auto is_listed = [&](uint32_t ctr, uint32_t actor) {
          if (!listed) return true;
          for (uint32_t k = 0; k < listed->n; k++)
            if (listed->ctr[k] == ctr && listed->actor[k] == actor) return true;
          return false;
        };
        auto refuse_history = [&]() {
          err = "unsupported: the batch edits an object that is no longer a visible value of its parent (objectMeta history)";
          return AM355_E_UNSUPPORTED;
        };
        const am355_ir_object& po = whole.objects[o];
        if (L.flags & OL_LIST_PARENT) {
          uint64_t ek = elem_key(L.elem_ctr, L.elem_actor);
          uint64_t pk = (uint64_t)o * 0x9e3779b97f4a7c15ull ^ ek;  // (set key: collisions only cost a missed dedup of identical work)
          bool exists = (L.flags & OL_ELEM_NEW) != 0 || appended_elems.count(pk) != 0;
          bool none_visible = false;
          if (!exists) {
            if (!whole.edits) { need.clear(); err = "edit records needed"; return AM355_E_UNSUPPORTED; }  // (the caller fetches them and calls again)
            auto it = elem_index.find(o);
            if (it == elem_index.end()) {
              auto& m = elem_index[o];
              for (uint32_t r = po.edit_begin; r < po.edit_end; r++) {
                const am355_ir_edit& ed = whole.edits[r];
                if (ed.flags & AM355_EDIT_CONT) continue;
                m.emplace(elem_key(ed.elem_ctr, ed.elem_actor), r);
              }
              it = elem_index.find(o);
            }
            auto hit = it->second.find(ek);
            // the visible values of the element (an element inside a multi-insert record holds one plain value: not looked for)
            std::vector<am355_ir_edit> vals;
            bool any_child = false;
            if (hit != it->second.end())
              for (uint32_t r = hit->second; r < po.edit_end; r++) {
                const am355_ir_edit& ed = whole.edits[r];
                if (ed.elem_ctr != L.elem_ctr || ed.elem_actor != L.elem_actor || (ed.flags & AM355_EDIT_CONT)) break;
                if (r > hit->second && !(ed.flags & AM355_EDIT_UPDATE)) break;
                // (an element that holds a counter completed by increments, or shows rows without a value: what objectMeta.children
                // lists for it -- the counter's `set` value, not its total, new.js:919-926 -- is not restated)
                if (ed.flags & (AM355_EDIT_COUNTER | AM355_EDIT_REMOVE)) return refuse_history();
                am355_ir_edit u = ed;
                u.flags = AM355_EDIT_UPDATE | (ed.flags & AM355_EDIT_CHILD);
                any_child = any_child || (ed.flags & AM355_EDIT_CHILD);
                vals.push_back(u);
              }
            if (L.flags & OL_VISIBLE) {
              if (vals.empty()) { err = "internal: element of a visible child object not found in the document patch"; return AM355_E_DEVICE; }
            } else if (hit == it->second.end() && !vals.size()) {
              // the element shows no value of its own record: either it is invisible (children = {}: the walk ends here) or it sits
              // inside a multi-insert run with a plain value -- then whether children still lists that value is history
              bool in_run = false;
              for (uint32_t r = po.edit_begin; r < po.edit_end && !in_run; r++) {
                const am355_ir_edit& ed = whole.edits[r];
                uint32_t count = whole.edits[r + 1].first - ed.first;
                if (!(ed.flags & (AM355_EDIT_UPDATE | AM355_EDIT_CHILD)) && ed.elem_actor == L.elem_actor && L.elem_ctr >= ed.elem_ctr && L.elem_ctr - ed.elem_ctr < count + 0u) in_run = true;
              }
              if (in_run) return refuse_history();
              none_visible = true;
            } else if (!any_child) { int rc = stale(none_visible); if (rc) return rc; }
            if (!none_visible && listed) {
              std::vector<am355_ir_edit> kept;
              for (const am355_ir_edit& u : vals) if (is_listed(u.id_ctr, u.id_actor)) kept.push_back(u);
              if (kept.size() != listed->n) { err = "unsupported: objectMeta lists values of a list element that the document patch does not show"; return AM355_E_UNSUPPORTED; }
              vals.swap(kept);
            }
            if (!none_visible) {
              // an edit of the batch already shows one of the listed values: the child is linked through it (new.js:1477-1481)
              const auto& ops = batch_edit_ops(o);
              for (const am355_ir_edit& u : vals)
                if (ops.count(elem_key(u.id_ctr, u.id_actor))) exists = true;
            }
            if (!none_visible && !exists) {
              auto& dst = extra_edits[o];
              dst.insert(dst.end(), vals.begin(), vals.end());
              appended_elems.insert(pk);
            }
          }
          if (none_visible) break;   // children[elemId] is empty: `childMeta && !hasChildren` (new.js:1521)
          patch_exists = exists;
        } else {
          if (!keys_indexed.count(o)) {
            keys_indexed.insert(o);
            for (uint32_t r = link[o].map_begin; r < link[o].map_end && r < n_dmap; r++) present_keys.insert(KeyRef{o, arena + d_map[r].key_off, d_map[r].key_len});
          }
          KeyRef kr{o, arena + L.key_off, L.key_len};
          bool exists = present_keys.count(kr) != 0;
          bool none_visible = false;
          if (!exists) {
            // the records of (o, key) in the whole-document table: sorted by key inside the object
            uint32_t lo = po.map_begin, hi = po.map_end;
            while (lo < hi) {
              uint32_t mid = lo + (hi - lo) / 2;
              if (key_cmp(arena + whole.map[mid].key_off, whole.map[mid].key_len, kr.p, kr.len) < 0) lo = mid + 1; else hi = mid;
            }
            std::vector<am355_ir_map> vals;
            bool any_child = false, any_value = false;
            for (uint32_t r = lo; r < po.map_end; r++) {
              const am355_ir_map& m = whole.map[r];
              if (m.key_len != kr.len || memcmp(arena + m.key_off, kr.p, kr.len) != 0) break;
              any_value = true;
              if (m.flags & AM355_MAP_COUNTER) continue;  // a counter with increments is not among objectMeta.children (new.js:921)
              any_child = any_child || (m.flags & AM355_MAP_CHILD);
              vals.push_back(m);
            }
            if (ask_always && any_value) { int rc = stale(none_visible); if (rc) return rc; }  // (a call may have skipped values of the property)
            else if (L.flags & OL_VISIBLE) {
              if (vals.empty()) { err = "internal: key of a visible child object not found in the document patch"; return AM355_E_DEVICE; }
            } else if (!any_value) none_visible = true;   // children[key] is empty whatever happened before
            else if (!any_child) { int rc = stale(none_visible); if (rc) return rc; }  // plain values only: children[key] lists them, or went empty once and stayed so
            if (!none_visible && listed) {
              std::vector<am355_ir_map> kept;
              for (const am355_ir_map& m : vals) if (is_listed(m.id_ctr, m.id_actor)) kept.push_back(m);
              if (kept.size() != listed->n) { err = "unsupported: objectMeta lists values of a property that the document patch does not show"; return AM355_E_UNSUPPORTED; }
              vals.swap(kept);
            }
            if (!none_visible) {
              auto& dst = extra_map[o];
              dst.insert(dst.end(), vals.begin(), vals.end());
              present_keys.insert(kr);
            }
          }
          if (none_visible) break;   // `childMeta && !hasChildren` (new.js:1521)
          patch_exists = exists;
        }
      }
      if (patch_exists || o == 0) break;
      child = o;
      o = link[o].parent;
      if (o == NONE32) break;
    }
  }

  if (!need.empty()) {
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    err = "unsupported: the batch edits an object that is no longer a visible value of its parent (objectMeta history)";
    return AM355_E_UNSUPPORTED;
  }
  // ---- the record tables of the patch ----
  out.objects.assign(whole.objects, whole.objects + NO);
  out.map.clear();
  out.edits.clear();
  uint32_t ordinal = 0;
  for (uint32_t oi = 0; oi < NO; oi++) {
    am355_ir_object& ob = out.objects[oi];
    ob.map_begin = (uint32_t)out.map.size();
    for (uint32_t r = link[oi].map_begin; r < link[oi].map_end && r < n_dmap; r++) out.map.push_back(d_map[r]);
    auto em = extra_map.find(oi);
    if (em != extra_map.end()) out.map.insert(out.map.end(), em->second.begin(), em->second.end());
    ob.map_end = (uint32_t)out.map.size();
    ob.edit_begin = (uint32_t)out.edits.size();
    for (uint32_t r = link[oi].edit_begin; r < link[oi].edit_end && r < n_dedits; r++) {
      am355_ir_edit ed = d_edits[r];
      uint32_t count = d_edits[r + 1].first - ed.first;
      ed.first = ordinal;
      ordinal += count;
      out.edits.push_back(ed);
    }
    auto ee = extra_edits.find(oi);
    if (ee != extra_edits.end())
      for (am355_ir_edit ed : ee->second) { ed.first = ordinal++; out.edits.push_back(ed); }
    ob.edit_end = (uint32_t)out.edits.size();
  }
  out.edits.push_back(am355_ir_edit{0, 0, 0, 0, 0, 0, ordinal, 0, 0, 0});
  out.ir = whole;
  out.ir.n_objects = NO;
  out.ir.n_map = (uint32_t)out.map.size();
  out.ir.n_edits = (uint32_t)out.edits.size() - 1;
  out.ir.n_values = ordinal;
  out.ir.objects = out.objects.data();
  out.ir.map = out.map.data();
  out.ir.edits = out.edits.data();
  return AM355_OK;
}

}  // namespace am355
