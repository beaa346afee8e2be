// Patch IR -> JSON text, byte-identical to JSON.stringify of the reference's patch object.
//
// Reference shapes reproduced (paths relative to the reference tree): envelope key order new.js:2064-2067;
// object patches new.js:726-732; map props new.js:1035-1039; list edits new.js:747-782 (key order of
// insert / multi-insert / update edits as produced by appendEdit); values columnar.js:300-329 decodeValue with
// `Object.assign({type: 'value'}, ...)` at new.js:971 and the counter form at new.js:963; JS own-property
// enumeration order (canonical array-index keys first, ascending) for `props` and `clock`.
// This is presentation of results that were computed on the GPU; it performs no merge logic.
#include "am355_render.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace am355 {
namespace {

struct R {
  const am355_patch_ir& ir;
  std::string& out;
  std::string err;
  int depth = 0;
  explicit R(const am355_patch_ir& i, std::string& o) : ir(i), out(o) {}

  bool fail(const char* m) { if (err.empty()) err = m; return false; }

  static bool utf8_valid(const uint8_t* p, size_t n) {
    size_t i = 0;
    while (i < n) {
      uint8_t b = p[i];
      if (b < 0x80) { i++; continue; }
      if (b >= 0xc2 && b <= 0xdf && i + 1 < n && (p[i + 1] & 0xc0) == 0x80) { i += 2; continue; }
      if (b >= 0xe0 && b <= 0xef && i + 2 < n && (p[i + 1] & 0xc0) == 0x80 && (p[i + 2] & 0xc0) == 0x80) {
        uint32_t c = (b & 0x0f) << 12 | (p[i + 1] & 0x3f) << 6 | (p[i + 2] & 0x3f);
        if (c < 0x800 || (c >= 0xd800 && c <= 0xdfff)) return false;
        i += 3;
        continue;
      }
      if (b >= 0xf0 && b <= 0xf4 && i + 3 < n && (p[i + 1] & 0xc0) == 0x80 && (p[i + 2] & 0xc0) == 0x80 && (p[i + 3] & 0xc0) == 0x80) {
        uint32_t c = (b & 0x07) << 18 | (p[i + 1] & 0x3f) << 12 | (p[i + 2] & 0x3f) << 6 | (p[i + 3] & 0x3f);
        if (c < 0x10000 || c > 0x10ffff) return false;
        i += 4;
        continue;
      }
      return false;
    }
    return true;
  }

  bool json_string(const uint8_t* p, size_t n) {
    if (!utf8_valid(p, n)) return fail("unsupported: malformed UTF-8 in a string");
    out.push_back('"');
    for (size_t i = 0; i < n; i++) {
      uint8_t ch = p[i];
      switch (ch) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default:
          if (ch < 0x20) { char t[8]; snprintf(t, sizeof t, "\\u%04x", ch); out += t; }
          else out.push_back((char)ch);
      }
    }
    out.push_back('"');
    return true;
  }

  void hex(const uint8_t* p, size_t n) {
    static const char hx[] = "0123456789abcdef";
    for (size_t i = 0; i < n; i++) { out.push_back(hx[p[i] >> 4]); out.push_back(hx[p[i] & 15]); }
  }

  bool op_id(uint32_t ctr, uint32_t a) {
    if (a >= ir.n_actors) return fail("internal: actor rank out of range");
    char t[24];
    snprintf(t, sizeof t, "\"%u@", ctr);
    out += t;
    hex(ir.actor_bytes + ir.actor_off[a], ir.actor_off[a + 1] - ir.actor_off[a]);
    out.push_back('"');
    return true;
  }

  // ECMA-262 Number::toString
  void json_double(double x) {
    if (std::isnan(x) || std::isinf(x)) { out += "null"; return; }
    if (x == 0) { out += "0"; return; }
    char digits[40];
    int exp10 = 0, k = 0;
    for (int prec = 1; prec <= 17; prec++) {
      char t[48];
      snprintf(t, sizeof t, "%.*e", prec - 1, x);
      if (strtod(t, nullptr) == x) {
        const char* p = t;
        if (*p == '-') p++;
        k = 0;
        for (; *p && *p != 'e'; p++) if (*p != '.') digits[k++] = *p;
        exp10 = atoi(p + 1);
        break;
      }
    }
    while (k > 1 && digits[k - 1] == '0') k--;
    digits[k] = 0;
    int n = exp10 + 1;
    if (x < 0) out.push_back('-');
    if (k <= n && n <= 21) {
      out += digits;
      out.append((size_t)(n - k), '0');
    } else if (0 < n && n <= 21) {
      out.append(digits, (size_t)n);
      out.push_back('.');
      out += digits + n;
    } else if (-6 < n && n <= 0) {
      out += "0.";
      out.append((size_t)(-n), '0');
      out += digits;
    } else {
      out.push_back(digits[0]);
      if (k > 1) { out.push_back('.'); out += digits + 1; }
      char t[16];
      snprintf(t, sizeof t, "e%c%d", n - 1 >= 0 ? '+' : '-', std::abs(n - 1));
      out += t;
    }
  }

  bool leb_value(uint32_t tl, uint32_t off, int64_t& v) {
    uint32_t tag = tl & 15, len = tl >> 4;
    const uint8_t* p = ir.arena + off;
    uint64_t u = 0;
    int shift = 0;
    for (uint32_t i = 0; i < len; i++) {
      uint8_t b = p[i];
      if (shift == 63 && (tag == 3 ? (b & 0xfe) != 0 : (b != 0 && b != 0x7f))) return fail("number out of range");
      u |= (uint64_t)(b & 0x7f) << shift;
      shift += 7;
      if (!(b & 0x80)) {
        if (tag != 3 && (b & 0x40) && shift < 64) u |= ~0ull << shift;
        v = (int64_t)u;
        const int64_t MAXS = 9007199254740991LL;
        if (tag == 3 ? u > (uint64_t)MAXS : (v > MAXS || v < -MAXS)) return fail("number out of range");
        return true;  // trailing bytes after the number are ignored, as `new Decoder(bytes).readInt53()` does
      }
    }
    return fail("buffer ended with incomplete number");
  }

  bool prim_value(uint32_t tl, uint32_t off) {
    uint32_t tag = tl & 15, len = tl >> 4;
    if (tl > 2 && (uint64_t)off + len > ir.arena_len) return fail("internal: value outside the arena");
    const uint8_t* p = ir.arena + off;
    if (tl == 0) { out += "null"; return true; }
    if (tl == 1) { out += "false"; return true; }
    if (tl == 2) { out += "true"; return true; }
    char t[40];
    switch (tag) {
      case 6:
        // utf8ToString = TextDecoder('utf-8').decode (encoding.js:9-17): a leading U+FEFF is dropped
        if (len >= 3 && p[0] == 0xef && p[1] == 0xbb && p[2] == 0xbf) return json_string(p + 3, len - 3);
        return json_string(p, len);
      case 3: case 4: case 8: case 9: {
        int64_t v;
        if (!leb_value(tl, off, v)) return false;
        snprintf(t, sizeof t, "%lld", (long long)v);
        out += t;
        return true;
      }
      case 5: {
        if (len != 8) return fail("Invalid length for floating point number");
        double x;
        memcpy(&x, p, 8);
        json_double(x);
        return true;
      }
      default:  // Uint8Array -> {"0":b0,"1":b1,...}
        out.push_back('{');
        for (uint32_t i = 0; i < len; i++) {
          snprintf(t, sizeof t, "%s\"%u\":%u", i ? "," : "", i, p[i]);
          out += t;
        }
        out.push_back('}');
        return true;
    }
  }

  // datatype property of decodeValue's result: absent for null/bool/string
  static bool has_datatype(uint32_t tl) { return !(tl == 0 || tl == 1 || tl == 2 || (tl & 15) == 6); }

  void datatype(uint32_t tag) {
    switch (tag) {
      case 3: out += "\"uint\""; break;
      case 4: out += "\"int\""; break;
      case 5: out += "\"float64\""; break;
      case 8: out += "\"counter\""; break;
      case 9: out += "\"timestamp\""; break;
      default: { char t[8]; snprintf(t, sizeof t, "%u", tag); out += t; }
    }
  }

  // value of a single-value edit record
  bool edit_value(const am355_ir_edit& ed) {
    if (ed.flags & AM355_EDIT_COUNTER) {  // {type:'value', datatype:'counter', value} (new.js:963)
      char t[80];
      snprintf(t, sizeof t, "{\"type\":\"value\",\"datatype\":\"counter\",\"value\":%lld}", (long long)((uint64_t)ed.pad << 32 | ed.val_off));
      out += t;
      return true;
    }
    return value(ed.val_tl, ed.val_off, (ed.flags & AM355_EDIT_CHILD) != 0);
  }

  bool value(uint32_t tl, uint32_t off, bool child) {
    if (child) return object(off);
    out += "{\"type\":\"value\",\"value\":";
    if (!prim_value(tl, off)) return false;
    if (has_datatype(tl)) { out += ",\"datatype\":"; datatype(tl & 15); }
    out.push_back('}');
    return true;
  }

  static bool array_index_key(const uint8_t* k, size_t n, uint64_t& v) {
    if (n == 0 || n > 10 || (n > 1 && k[0] == '0')) return false;
    v = 0;
    for (size_t i = 0; i < n; i++) {
      if (k[i] < '0' || k[i] > '9') return false;
      v = v * 10 + (k[i] - '0');
    }
    return v <= 4294967294ull;
  }

  bool same_key(const am355_ir_map& x, const am355_ir_map& y) {
    return x.key_len == y.key_len && memcmp(ir.arena + x.key_off, ir.arena + y.key_off, x.key_len) == 0;
  }

  bool prop(uint32_t begin, uint32_t end) {  // map records [begin, end) share one key
    const am355_ir_map& m0 = ir.map[begin];
    // (a key that starts with U+FEFF loses it in the reference's utf8ToString and then collides with other keys: JS path)
    if (m0.key_len >= 3 && ir.arena[m0.key_off] == 0xef && ir.arena[m0.key_off + 1] == 0xbb && ir.arena[m0.key_off + 2] == 0xbf)
      return fail("unsupported: map key starts with a byte order mark");
    if (!json_string(ir.arena + m0.key_off, m0.key_len)) return false;
    out += ":{";
    for (uint32_t i = begin; i < end; i++) {
      const am355_ir_map& m = ir.map[i];
      if (m.flags & AM355_MAP_EMPTY) continue;  // incremental patch: `props[key] = {}` (new.js:1037)
      if (i > begin) out.push_back(',');
      if (!op_id(m.id_ctr, m.id_actor)) return false;
      out.push_back(':');
      if (m.flags & AM355_MAP_COUNTER) {
        char t[80];
        snprintf(t, sizeof t, "{\"type\":\"value\",\"datatype\":\"counter\",\"value\":%lld}", (long long)m.counter);
        out += t;
      } else if (!value(m.val_tl, m.val_off, (m.flags & AM355_MAP_CHILD) != 0)) {
        return false;
      }
    }
    out.push_back('}');
    return true;
  }

  // one record's values (all with the type/length word val_tl, back to back in the arena), comma separated
  bool record_values(const am355_ir_edit& ed, uint32_t count) {
    if (ed.flags & AM355_EDIT_COUNTER) {  // one value per record: the total of a counter inside a list
      char t[32];
      snprintf(t, sizeof t, "%lld", (long long)((uint64_t)ed.pad << 32 | ed.val_off));
      out += t;
      return count == 1 || fail("internal: counter record with several values");
    }
    uint32_t len = ed.val_tl >> 4;
    for (uint32_t k = 0; k < count; k++) {
      if (k) out.push_back(',');
      if (!prim_value(ed.val_tl, ed.val_off + k * len)) return false;
    }
    return true;
  }

  // edit records [b, e) of one list object, comma separated; b is not a continuation record
  bool edits_serial(uint32_t b, uint32_t e) {
    char t[64];
    for (uint32_t i = b; i < e;) {
      const am355_ir_edit& ed = ir.edits[i];
      uint32_t count = ir.edits[i + 1].first - ed.first;
      uint32_t j = i + 1;
      while (j < e && (ir.edits[j].flags & AM355_EDIT_CONT)) j++;  // further records of the same multi-insert
      if (ir.edits[i + 1].first <= ed.first || ir.edits[j].first > ir.n_values) return fail("internal: edit record without values");
      if (i > b) out.push_back(',');
      if (ed.flags & AM355_EDIT_REMOVE) {  // incremental patch (new.js:1029, 775-777)
        snprintf(t, sizeof t, "{\"action\":\"remove\",\"index\":%u,\"count\":%u}", ed.index, count);
        out += t;
      } else if (count >= 2 || j > i + 1 || (ed.flags & AM355_EDIT_MULTI)) {
        snprintf(t, sizeof t, "{\"action\":\"multi-insert\",\"index\":%u,\"elemId\":", ed.index);
        out += t;
        if (!op_id(ed.elem_ctr, ed.elem_actor)) return false;
        uint32_t tl = ed.val_tl;
        if (has_datatype(tl) && (tl & 15) != 0) { out += ",\"datatype\":"; datatype(tl & 15); }  // only truthy datatypes (new.js:762)
        out += ",\"values\":[";
        for (uint32_t r = i; r < j; r++) {
          if (r > i) out.push_back(',');
          if (!record_values(ir.edits[r], ir.edits[r + 1].first - ir.edits[r].first)) return false;
        }
        out += "]}";
      } else if (ed.flags & AM355_EDIT_UPDATE) {
        snprintf(t, sizeof t, "{\"action\":\"update\",\"index\":%u,\"opId\":", ed.index);
        out += t;
        if (!op_id(ed.id_ctr, ed.id_actor)) return false;
        out += ",\"value\":";
        if (!edit_value(ed)) return false;
        out.push_back('}');
      } else {
        snprintf(t, sizeof t, "{\"action\":\"insert\",\"index\":%u,\"elemId\":", ed.index);
        out += t;
        if (!op_id(ed.elem_ctr, ed.elem_actor)) return false;
        out += ",\"opId\":";
        if (!op_id(ed.id_ctr, ed.id_actor)) return false;
        out += ",\"value\":";
        if (!edit_value(ed)) return false;
        out.push_back('}');
      }
      i = j;
    }
    return true;
  }

  // Long edit lists (a Text object holds one edit per character run) are rendered by several host threads, each into its
  // own buffer, split where no multi-insert is cut; the pieces are then joined. Same text as the serial walk.
  bool edits(uint32_t b, uint32_t e) {
    // (AM355_RENDER_CHUNK: edits per thread below which the walk stays serial; the tests lower it to exercise the join)
    const char* env = getenv("AM355_RENDER_CHUNK");
    const uint32_t kMinPerThread = env && atoi(env) > 0 ? (uint32_t)atoi(env) : 1u << 15;
    unsigned hw = std::thread::hardware_concurrency();
    uint32_t want = (e - b) / kMinPerThread;
    if (want > 16) want = 16;
    if (hw && want > hw) want = hw;
    if (want < 2) return edits_serial(b, e);
    std::vector<uint32_t> cut{b};
    for (uint32_t k = 1; k < want; k++) {
      uint32_t c = b + (uint32_t)((uint64_t)(e - b) * k / want);
      while (c < e && (ir.edits[c].flags & AM355_EDIT_CONT)) c++;  // never between the records of one multi-insert
      if (c > cut.back() && c < e) cut.push_back(c);
    }
    cut.push_back(e);
    size_t parts = cut.size() - 1;
    std::vector<std::string> piece(parts), perr(parts);
    std::vector<char> ok(parts, 1);
    std::vector<std::thread> workers;
    for (size_t k = 0; k < parts; k++)
      workers.emplace_back([&, k]() {
        R sub(ir, piece[k]);
        ok[k] = sub.edits_serial(cut[k], cut[k + 1]) ? 1 : 0;
        perr[k] = sub.err;
      });
    for (auto& w : workers) w.join();
    size_t total = out.size() + parts;
    for (auto& pc : piece) total += pc.size();
    out.reserve(total + 64);
    for (size_t k = 0; k < parts; k++) {
      if (!ok[k]) return fail(perr[k].c_str());
      if (k && !piece[k].empty()) out.push_back(',');
      out += piece[k];
    }
    return true;
  }

  bool object(uint32_t oi) {
    if (oi >= ir.n_objects) return fail("internal: object index out of range");
    if (++depth > 100000) return fail("unsupported: object nesting too deep");
    const am355_ir_object& ob = ir.objects[oi];
    uint32_t type = oi == 0 ? 0 : ob.type;
    out += "{\"objectId\":";
    if (oi == 0) out += "\"_root\""; else if (!op_id(ob.id_ctr, ob.id_actor)) return false;
    out += ",\"type\":";
    switch (type) {
      case 0: out += "\"map\""; break;
      case 2: out += "\"list\""; break;
      case 4: out += "\"text\""; break;
      case 6: out += "\"table\""; break;
      default: out += "null";
    }
    if (oi != 0 && (type == 2 || type == 4)) {
      out += ",\"edits\":[";
      uint32_t b = ob.edit_begin, e = ob.edit_end;
      if (b > e || e > ir.n_edits) return fail("internal: edit range out of bounds");
      if (!edits(b, e)) return false;
      out += "]}";
    } else {
      out += ",\"props\":{";
      uint32_t b = ob.map_begin, e = ob.map_end;
      if (b > e || e > ir.n_map) return fail("internal: map range out of bounds");
      // group by key; integer-like keys first in numeric order, then the rest in (already sorted) key order
      struct Grp { uint32_t b, e; uint64_t num; bool is_index; };
      std::vector<Grp> groups;
      for (uint32_t i = b; i < e;) {
        uint32_t j = i + 1;
        while (j < e && same_key(ir.map[i], ir.map[j])) j++;
        Grp g{i, j, 0, false};
        g.is_index = array_index_key(ir.arena + ir.map[i].key_off, ir.map[i].key_len, g.num);
        groups.push_back(g);
        i = j;
      }
      std::vector<Grp> idx;
      for (auto& g : groups) if (g.is_index) idx.push_back(g);
      std::stable_sort(idx.begin(), idx.end(), [](const Grp& x, const Grp& y) { return x.num < y.num; });
      bool first = true;
      for (auto& g : idx) { if (!first) out.push_back(','); first = false; if (!prop(g.b, g.e)) return false; }
      for (auto& g : groups) {
        if (g.is_index) continue;
        if (!first) out.push_back(',');
        first = false;
        if (!prop(g.b, g.e)) return false;
      }
      out += "}}";
    }
    depth--;
    return true;
  }

  bool run() {
    char t[64];
    snprintf(t, sizeof t, "{\"maxOp\":%llu,\"clock\":{", (unsigned long long)ir.max_op);
    out += t;
    {
      struct CK { std::string hex; uint64_t seq, num; bool is_index; };
      std::vector<CK> ck;
      for (uint32_t i = 0; i < ir.n_clock; i++) {
        uint32_t a = ir.clock_actor[i];
        CK k;
        std::string tmp;
        std::swap(tmp, out);
        hex(ir.actor_bytes + ir.actor_off[a], ir.actor_off[a + 1] - ir.actor_off[a]);
        std::swap(tmp, out);
        k.hex = tmp;
        k.seq = ir.clock_seq[i];
        k.is_index = array_index_key((const uint8_t*)k.hex.data(), k.hex.size(), k.num);
        ck.push_back(k);
      }
      std::vector<CK> idx;
      for (auto& k : ck) if (k.is_index) idx.push_back(k);
      std::stable_sort(idx.begin(), idx.end(), [](const CK& x, const CK& y) { return x.num < y.num; });
      bool first = true;
      auto put = [&](const CK& k) {
        if (!first) out.push_back(',');
        first = false;
        out.push_back('"'); out += k.hex;
        snprintf(t, sizeof t, "\":%llu", (unsigned long long)k.seq);
        out += t;
      };
      for (auto& k : idx) put(k);
      for (auto& k : ck) if (!k.is_index) put(k);
    }
    out += "},\"deps\":[";
    for (uint32_t i = 0; i < ir.n_heads; i++) {
      if (i) out.push_back(',');
      out.push_back('"');
      hex(ir.heads + 32 * i, 32);
      out.push_back('"');
    }
    snprintf(t, sizeof t, "],\"pendingChanges\":%u,\"diffs\":", ir.pending);
    out += t;
    if (!object(0)) return false;
    out.push_back('}');
    return true;
  }
};

}  // namespace

bool render_patch_json(const am355_patch_ir& ir, std::string& out, std::string& err) {
  R r(ir, out);
  bool ok = r.run();
  if (!ok) err = r.err;
  return ok;
}

}  // namespace am355
