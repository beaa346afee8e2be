// Documents out (include/am355.h): Backend.save (row order + column encoders on the device, document assembly on the host) and the
// history of a loaded document (am355_doc_changes: device stages of am355_hist.hip, host finish of am355_history.cpp). See am355_ctx.h.
#include "am355_ctx.h"

// ---------------------------------------------------------------------------------------------------------
// Backend.save (new.js:2033-2055, columnar.js:983-1004)
// ---------------------------------------------------------------------------------------------------------
namespace {

struct HostOut : std::vector<uint8_t> {
  void uleb(uint64_t v) { while (v >= 0x80) { push_back((uint8_t)(v | 0x80)); v >>= 7; } push_back((uint8_t)v); }
  void sleb(int64_t v) {
    for (;;) {
      uint8_t b = (uint8_t)(v & 0x7f);
      v >>= 7;
      if ((v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40))) { push_back(b); return; }
      push_back(b | 0x80);
    }
  }
  void bytes(const void* p, size_t n) { insert(end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};

// RLE of the (small) change-metadata columns: same output as the device encoders (am355_encode.hip), values never null
template <class T, class Put>
void host_rle(HostOut& out, const std::vector<T>& v, Put put) {
  size_t n = v.size(), i = 0;
  std::vector<size_t> lit;
  auto flush = [&]() {
    if (lit.empty()) return;
    out.sleb(-(int64_t)lit.size());
    for (size_t k : lit) put(out, v[k]);
    lit.clear();
  };
  while (i < n) {
    size_t j = i + 1;
    while (j < n && v[j] == v[i]) j++;
    if (j - i >= 2) { flush(); out.sleb((int64_t)(j - i)); put(out, v[i]); }
    else lit.push_back(i);
    i = j;
  }
  flush();
}
void host_rle_uint(HostOut& out, const std::vector<int64_t>& v) {
  host_rle(out, v, [](HostOut& o, int64_t x) { o.uleb((uint64_t)x); });
}
void host_delta(HostOut& out, const std::vector<int64_t>& v) {
  std::vector<int64_t> d(v.size());
  int64_t prev = 0;
  for (size_t i = 0; i < v.size(); i++) { d[i] = v[i] - prev; prev = v[i]; }
  host_rle(out, d, [](HostOut& o, int64_t x) { o.sleb(x); });
}
void host_rle_str(HostOut& out, const std::vector<std::string>& v) {
  host_rle(out, v, [](HostOut& o, const std::string& x) { o.uleb(x.size()); o.bytes(x.data(), x.size()); });
}

struct SaveColumn {
  uint32_t id;
  std::vector<uint8_t> data;
};

// columns of >= 256 bytes are stored DEFLATEd with bit 3 of the id set (columnar.js:1052-1057, DEFLATE_MIN_SIZE)
bool deflate_column(SaveColumn& col) {
  if (col.data.size() < 256) return true;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  std::vector<uint8_t> out(deflateBound(&zs, (uLong)col.data.size()) + 64);
  zs.next_in = col.data.data(); zs.avail_in = (uInt)col.data.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
  int rc = deflate(&zs, Z_FINISH);
  size_t got = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return false;
  out.resize(got);
  col.data.swap(out);
  col.id |= 8;
  return true;
}

struct ChangeInfo {  // header fields of one applied change, parsed on the host (a few thousand small headers)
  std::vector<Hash32> deps;
  uint64_t seq = 0, start_op = 0;
  int64_t time = 0;
  std::string message, extra;
};

bool parse_change_info(const uint8_t* p, size_t len, ChangeInfo& ci) {
  size_t off = 9;
  uint64_t clen, v;
  if (len < 10 || !read_uleb_host(p, len, off, clen)) return false;
  if (!read_uleb_host(p, len, off, v) || v * 32 > len - off) return false;
  ci.deps.resize((size_t)v);
  for (auto& d : ci.deps) { memcpy(d.b, p + off, 32); off += 32; }
  if (!read_uleb_host(p, len, off, v) || v > len - off) return false;
  off += (size_t)v;  // actor
  if (!read_uleb_host(p, len, off, ci.seq) || !read_uleb_host(p, len, off, ci.start_op)) return false;
  {  // time: signed LEB128
    uint64_t u = 0;
    int shift = 0;
    for (;;) {
      if (off >= len || shift > 63) return false;
      uint8_t b = p[off++];
      u |= (uint64_t)(b & 0x7f) << shift;
      shift += 7;
      if (!(b & 0x80)) { if ((b & 0x40) && shift < 64) u |= ~0ull << shift; break; }
    }
    ci.time = (int64_t)u;
  }
  if (!read_uleb_host(p, len, off, v) || v > len - off) return false;
  ci.message.assign((const char*)p + off, (size_t)v);
  off += (size_t)v;
  if (!read_uleb_host(p, len, off, v)) return false;
  for (uint64_t k = 0; k < v; k++) {
    uint64_t l;
    if (!read_uleb_host(p, len, off, l) || l > len - off) return false;
    off += (size_t)l;
  }
  uint64_t ncols, total = 0;
  if (!read_uleb_host(p, len, off, ncols)) return false;
  for (uint64_t k = 0; k < ncols; k++) {
    uint64_t id, l;
    if (!read_uleb_host(p, len, off, id) || !read_uleb_host(p, len, off, l)) return false;
    total += l;
  }
  if (total > len - off) return false;
  off += (size_t)total;
  ci.extra.assign((const char*)p + off, len - off);  // extraBytes (columnar.js:757-760)
  return true;
}

__global__ __launch_bounds__(BLOCK) void k_save_identity(uint32_t n, uint32_t* __restrict__ v) {
  uint32_t i = gtid();
  if (i < n) v[i] = i;
}
// loaded document: rows are canonical already; only the actor fields change representation (rank -> document index)
__global__ __launch_bounds__(BLOCK) void k_save_doc_rows(OpCols in, uint32_t n, uint32_t n_succ, const uint32_t* __restrict__ doc_actor, OpCols out) {
  uint32_t f = gtid();
  if (f < n_succ) { out.pred_actor[f] = doc_actor[in.pred_actor[f]]; out.pred_ctr[f] = in.pred_ctr[f]; }
  if (f >= n) return;
  bool root = in.obj_actor[f] == NONE32;
  out.obj_actor[f] = root ? NONE32 : doc_actor[in.obj_actor[f]];
  out.obj_ctr[f] = root ? NONE32 : in.obj_ctr[f];
  out.key_actor[f] = in.key_actor[f] == NONE32 ? NONE32 : doc_actor[in.key_actor[f]];
  out.key_ctr[f] = in.key_ctr[f];
  out.key_off[f] = in.key_off[f];
  out.key_len[f] = in.key_len[f];
  out.id_actor[f] = doc_actor[in.id_actor[f]];
  out.id_ctr[f] = in.id_ctr[f];
  out.insert[f] = in.insert[f];
  out.action[f] = in.action[f];
  out.val_tl[f] = in.val_tl[f];
  out.val_off[f] = in.val_off[f];
  out.pred_num[f] = in.pred_num[f];
}

}  // namespace

int save_impl(am355_ctx* c, uint32_t flags, const uint8_t** out_bytes, size_t* out_len) {
  if (!c || !out_bytes || !out_len) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed) return fail(c, AM355_E_STATE, "am355_replay must be called first");
  (void)hipSetDevice(c->device);
  if (c->is_document && !(flags & 1)) {  // unchanged document: the reference returns the bytes it was given (new.js:2034)
    *out_bytes = c->doc_bytes.data();
    *out_len = c->doc_bytes.size();
    return AM355_OK;
  }
  // (a sharded replay decodes the rows of foreign objects only as far as their object columns: their key / value / pred fields are
  // whatever an earlier replay left there)
  if (c->shard_world > 1) return fail(c, AM355_E_UNSUPPORTED, "am355_save on a sharded context");
  if (c->n_pending) return fail(c, AM355_E_UNSUPPORTED, "changes are queued: the document is saved by the JS path");
  { int frc = ensure_ir_fresh(c); if (frc) return frc; }   // (the save's row order reads the whole-document tables: rebuilt if in-place list merges left them stale)
  if (!c->is_document && c->has_unknown_cols) return fail(c, AM355_E_UNSUPPORTED, "a change carries columns this engine does not model: the document is saved by the JS path");
  hipStream_t st = c->stream;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "am355_save: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
    t_start = now;
  };
  const uint32_t N = (uint32_t)c->n_ops, P = (uint32_t)c->n_preds, NA = (uint32_t)c->actors.size();
  const uint32_t n_obj = c->counts.n_objects, n_ins = c->counts.n_list_ins;
  // ---- actor table of the document: order of first appearance (new.js:1434-1441); documents keep theirs ----
  std::vector<uint32_t> doc_actor(std::max(NA, 1u), 0);  // rank -> document index
  std::vector<uint32_t> actor_by_doc(NA, 0);
  if (c->is_document) {
    for (uint32_t i = 0; i < NA; i++) { doc_actor[c->doc_actor_rank[i]] = i; actor_by_doc[i] = c->doc_actor_rank[i]; }
  } else {
    if (c->clock_actor.size() != NA) return fail(c, AM355_E_UNSUPPORTED, "actors without an applied change");
    for (uint32_t i = 0; i < NA; i++) { doc_actor[c->clock_actor[i]] = i; actor_by_doc[i] = c->clock_actor[i]; }
  }
  // ---- device buffers ----
  size_t n1 = (size_t)N + 2, p1 = (size_t)P + 2, o1 = (size_t)n_obj + 2;
  auto al = [](size_t b) { return carve_round(b); };
  size_t save_bytes = 10 * al(4 * n1) + 3 * al(4 * o1) + al(16 * o1) + 2 * al(8 * p1) + 2 * al(4 * p1) + al(64) + 13 * al(4 * n1) + al(n1) + 2 * al(4 * p1) + al(4 * (size_t)std::max(NA, 1u));
  if (!c->d_save.ensure(save_bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed (save)");
  SaveBufs s;
  uint32_t* d_doc_actor;
  {
    uint8_t* p = c->d_save.as<uint8_t>();
    canary_scope("save buffers (d_save)");
    canary_forget(c->d_save.p, c->d_save.cap);
    auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al(bytes); return r; };
    uint32_t** a10[] = {&s.map_flag, &s.map_ex, &s.upd_flag, &s.upd_ex, &s.upd_cnt, &s.pos_of, &s.list_off, &s.final_pos, &s.src_of, &s.map_perm};
    for (uint32_t** a : a10) *a = (uint32_t*)take(4 * n1);
    s.obj_rank = (uint32_t*)take(4 * o1); s.rank_obj = (uint32_t*)take(4 * o1); s.base_by_rank = (uint32_t*)take(4 * o1);
    uint32_t* bounds = (uint32_t*)take(16 * o1);
    s.map_begin = bounds; s.map_end = bounds + (n_obj + 1); s.list_begin = bounds + 2 * (size_t)(n_obj + 1); s.list_end = bounds + 3 * (size_t)(n_obj + 1);
    s.succ_key_a = (uint64_t*)take(8 * p1); s.succ_key_b = (uint64_t*)take(8 * p1);
    s.succ_val_a = (uint32_t*)take(4 * p1); s.succ_val_b = (uint32_t*)take(4 * p1);
    s.words = (uint32_t*)take(64);
    OpCols& o = s.out;
    uint32_t** cols13[] = {&o.obj_actor, &o.obj_ctr, &o.key_actor, &o.key_ctr, &o.key_off, &o.key_len, &o.action, &o.val_tl, &o.val_off, &o.pred_first, &o.pred_num, &o.id_ctr, &o.id_actor};
    for (uint32_t** a : cols13) *a = (uint32_t*)take(4 * n1);
    o.insert = (uint8_t*)take(n1);
    o.pred_actor = (uint32_t*)take(4 * p1); o.pred_ctr = (uint32_t*)take(4 * p1);
    d_doc_actor = (uint32_t*)take(4 * (size_t)std::max(NA, 1u));
    canary_arm();
  }
  HIPCHK(c, hipMemcpyAsync(d_doc_actor, doc_actor.data(), 4 * (size_t)std::max(NA, 1u), hipMemcpyHostToDevice, st));
  // ---- rows in saved-document order ----
  uint32_t n_doc, n_succ = P;
  if (c->is_document) {
    n_doc = N;
    uint32_t n = std::max(N, P);
    if (n) AM355_LAUNCH_INDEPENDENT(k_save_doc_rows, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), st, c->cols, N, P, (const uint32_t*)d_doc_actor, s.out);
  } else {
    uint32_t words[8];
    save_phase1(c->mb, s, st);
    HIPCHK(c, hipMemcpyAsync(c->h_words.p, s.words, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    memcpy(words, c->h_words.p, 32);
    n_doc = words[0] + words[1] + n_ins;
    save_phase2(c->mb, c->ir, s, words, n_obj, n_ins, d_doc_actor, st);
  }
  // ---- encode the op columns on the device ----
  enum { E_OBJ_ACTOR, E_OBJ_CTR, E_KEY_ACTOR, E_KEY_CTR, E_KEY_STR, E_ID_ACTOR, E_ID_CTR, E_INSERT, E_ACTION, E_VAL_LEN, E_VAL_RAW, E_SUCC_NUM, E_SUCC_ACTOR, E_SUCC_CTR, E_NUM };
  static const uint32_t col_id[E_NUM] = {0x01, 0x02, 0x11, 0x13, 0x15, 0x21, 0x23, 0x34, 0x42, 0x56, 0x57, 0x80, 0x81, 0x83};
  uint32_t nmax = std::max(n_doc, n_succ);
  size_t raw_bound = c->raw.size() + 16;  // value bytes and key strings come out of the arena: never more than all of it
  size_t bound[E_NUM];
  for (int k = 0; k < E_NUM; k++) bound[k] = al(enc_numbers_bound(k >= E_SUCC_ACTOR ? n_succ : n_doc));
  bound[E_KEY_STR] = al(enc_numbers_bound(n_doc) + raw_bound);
  bound[E_VAL_RAW] = al(raw_bound);
  size_t out_total = 0;
  for (int k = 0; k < E_NUM; k++) out_total += bound[k];
  size_t enc_bytes = al(enc_work_bytes(nmax)) + al(4 * ((size_t)nmax + 2)) + al((size_t)nmax + 2) + al(4 * E_NUM);
  if (!c->d_enc.ensure(enc_bytes) || !c->d_encout.ensure(out_total + 256) || !c->h_words.ensure(256)) return fail(c, AM355_E_NOMEM, "device allocation failed (save columns)");
  EncWork w;
  canary_forget(c->d_enc.p, c->d_enc.cap);
  enc_carve(w, c->d_enc.p, nmax);
  canary_arm();
  uint32_t* deltas = (uint32_t*)(c->d_enc.as<uint8_t>() + al(enc_work_bytes(nmax)));
  uint8_t* nullmask = (uint8_t*)deltas + al(4 * ((size_t)nmax + 2));
  uint32_t* d_lens = (uint32_t*)(nullmask + al((size_t)nmax + 2));
  uint8_t* outp[E_NUM];
  {
    uint8_t* p = c->d_encout.as<uint8_t>();
    for (int k = 0; k < E_NUM; k++) { outp[k] = p; p += bound[k]; }
  }
  const OpCols& o = s.out;
  const uint8_t* arena = c->d_arena.as<uint8_t>();
  enc_rle_numbers(o.obj_actor, nullptr, n_doc, false, w, outp[E_OBJ_ACTOR], d_lens + E_OBJ_ACTOR, st);
  enc_rle_numbers(o.obj_ctr, nullptr, n_doc, false, w, outp[E_OBJ_CTR], d_lens + E_OBJ_CTR, st);
  enc_rle_numbers(o.key_actor, nullptr, n_doc, false, w, outp[E_KEY_ACTOR], d_lens + E_KEY_ACTOR, st);
  enc_delta_prepare(o.key_ctr, n_doc, deltas, nullmask, w, st);
  enc_rle_numbers(deltas, nullmask, n_doc, true, w, outp[E_KEY_CTR], d_lens + E_KEY_CTR, st);
  enc_rle_strings(arena, o.key_off, o.key_len, n_doc, w, outp[E_KEY_STR], d_lens + E_KEY_STR, st);
  enc_rle_numbers(o.id_actor, nullptr, n_doc, false, w, outp[E_ID_ACTOR], d_lens + E_ID_ACTOR, st);
  enc_delta_prepare(o.id_ctr, n_doc, deltas, nullmask, w, st);
  enc_rle_numbers(deltas, nullmask, n_doc, true, w, outp[E_ID_CTR], d_lens + E_ID_CTR, st);
  enc_boolean(o.insert, n_doc, w, outp[E_INSERT], d_lens + E_INSERT, st);
  enc_rle_numbers(o.action, nullptr, n_doc, false, w, outp[E_ACTION], d_lens + E_ACTION, st);
  enc_rle_numbers(o.val_tl, nullptr, n_doc, false, w, outp[E_VAL_LEN], d_lens + E_VAL_LEN, st);
  enc_raw_values(arena, o.val_off, o.val_tl, n_doc, w, outp[E_VAL_RAW], d_lens + E_VAL_RAW, st);
  enc_rle_numbers(o.pred_num, nullptr, n_doc, false, w, outp[E_SUCC_NUM], d_lens + E_SUCC_NUM, st);
  enc_rle_numbers(o.pred_actor, nullptr, n_succ, false, w, outp[E_SUCC_ACTOR], d_lens + E_SUCC_ACTOR, st);
  enc_delta_prepare(o.pred_ctr, n_succ, deltas, nullmask, w, st);
  enc_rle_numbers(deltas, nullmask, n_succ, true, w, outp[E_SUCC_CTR], d_lens + E_SUCC_CTR, st);
  HIPCHK(c, hipMemcpyAsync(c->h_words.p, d_lens, 4 * E_NUM, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  uint32_t lens[E_NUM];
  memcpy(lens, c->h_words.p, 4 * E_NUM);
  lap("row order + column encode");
  std::vector<SaveColumn> ops_cols;
  {
    size_t total = 0;
    for (int k = 0; k < E_NUM; k++) { if (lens[k] > bound[k]) return fail(c, AM355_E_UNSUPPORTED, "encoded column larger than its bound"); total += lens[k]; }
    if (!c->h_encout.ensure(total + 16)) return fail(c, AM355_E_NOMEM, "host allocation failed (save)");
    uint8_t* h = c->h_encout.as<uint8_t>();
    size_t at = 0;
    for (int k = 0; k < E_NUM; k++) {
      if (lens[k]) HIPCHK(c, hipMemcpyAsync(h + at, outp[k], lens[k], hipMemcpyDeviceToHost, st));
      at += lens[k];
    }
    HIPCHK(c, hipStreamSynchronize(st));
    at = 0;
    for (int k = 0; k < E_NUM; k++) {
      ops_cols.push_back(SaveColumn{col_id[k], std::vector<uint8_t>(h + at, h + at + lens[k])});
      at += lens[k];
    }
  }
  lap("columns to host");
  {
    // A key that starts with U+FEFF: the reference decodes keys on their way into a document (TextDecoder drops a leading byte
    // order mark, encoding.js:9-17) and writes the shortened key; such documents are saved by the JS path. (Walk of the encoded key
    // column: one step per run or literal.)
    const std::vector<uint8_t>& kc = ops_cols[E_KEY_STR].data;
    size_t o = 0;
    auto uleb = [&](uint64_t& v) { return read_uleb_host(kc.data(), kc.size(), o, v); };
    auto sleb = [&](int64_t& v) {
      uint64_t u = 0; int shift = 0;
      while (o < kc.size() && shift < 64) {
        uint8_t b = kc[o++];
        u |= (uint64_t)(b & 0x7f) << shift; shift += 7;
        if (!(b & 0x80)) { if ((b & 0x40) && shift < 64) u |= ~0ull << shift; v = (int64_t)u; return true; }
      }
      return false;
    };
    bool bom = false, ok = true;
    while (ok && o < kc.size() && !bom) {
      int64_t n;
      if (!(ok = sleb(n))) break;
      uint64_t strings = n > 0 ? 1 : n < 0 ? (uint64_t)-n : 0, l;
      if (n == 0) { ok = uleb(l); continue; }
      for (uint64_t k = 0; ok && k < strings; k++) {
        ok = uleb(l) && l <= kc.size() - o;
        if (ok) { bom = bom || (l >= 3 && kc[o] == 0xef && kc[o + 1] == 0xbb && kc[o + 2] == 0xbf); o += (size_t)l; }
      }
    }
    if (!ok) return fail(c, AM355_E_DEVICE, "internal: encoded key column does not parse");
    if (bom) return fail(c, AM355_E_UNSUPPORTED, "a map key starts with a byte order mark: the document is saved by the JS path");
  }
  // ---- change metadata columns (columnar.js:86-96, new.js:1680-1692) ----
  std::vector<SaveColumn> chg_cols;
  std::vector<uint8_t> tail;  // headsIndexes (+ extraBytes of a loaded document)
  if (c->is_document) {
    // diagnostic re-encode of a loaded document (flags & 1): its change metadata and trailer are kept as loaded
    if (c->doc_other_ops_cols) return fail(c, AM355_E_UNSUPPORTED, "document has op columns this engine does not model");
    for (auto& col : c->doc_chg_cols) chg_cols.push_back(SaveColumn{col.first, col.second});
    tail = c->doc_tail;
  } else {
    size_t na = c->applied_change.size();
    std::vector<int64_t> v_actor(na), v_seq(na), v_maxop(na), v_time(na), v_depsnum(na), v_depsidx, v_extralen(na);
    std::vector<std::string> v_msg(na);
    std::string extra_raw;
    std::unordered_map<Hash32, uint32_t, Hash32Hasher> index_of;
    const uint8_t* hs = c->h_hashes.as<uint8_t>();
    for (size_t i = 0; i < na; i++) {
      Hash32 h;
      memcpy(h.b, hs + 32 * (size_t)c->applied_change[i], 32);
      index_of.emplace(h, (uint32_t)i);
    }
    // author of each applied change: rank of its actor id
    std::unordered_map<std::string, uint32_t> rank_of;
    for (uint32_t r = 0; r < NA; r++) rank_of[c->actors[r]] = r;
    for (size_t i = 0; i < na; i++) {
      uint32_t ci = c->applied_change[i];
      const uint8_t* p = c->raw.data() + c->raw_off[ci];
      size_t len = (size_t)(c->raw_off[ci + 1] - c->raw_off[ci]);
      ChangeInfo info;
      if (!parse_change_info(p, len, info)) return fail(c, AM355_E_INVALID, "change %u: malformed header", ci);
      // actor id bytes: after the deps
      size_t off = 9;
      uint64_t clen, nd, al2;
      read_uleb_host(p, len, off, clen);
      read_uleb_host(p, len, off, nd);
      off += (size_t)nd * 32;
      read_uleb_host(p, len, off, al2);
      auto it = rank_of.find(std::string((const char*)p + off, (size_t)al2));
      if (it == rank_of.end()) return fail(c, AM355_E_STATE, "change %u: unknown author", ci);
      uint32_t n_ops_i = (i + 1 < na ? c->applied_op_base[i + 1] : N) - c->applied_op_base[i];
      v_actor[i] = doc_actor[it->second];
      v_seq[i] = (int64_t)info.seq;
      v_maxop[i] = (int64_t)(info.start_op + n_ops_i) - 1;
      v_time[i] = info.time;
      v_msg[i] = info.message;
      v_depsnum[i] = (int64_t)info.deps.size();
      for (const Hash32& d : info.deps) {
        auto di = index_of.find(d);
        if (di == index_of.end()) return fail(c, AM355_E_STATE, "change %u: dependency is not an applied change", ci);
        v_depsidx.push_back(di->second);
      }
      v_extralen[i] = (int64_t)(info.extra.size() << 4 | 7);  // VALUE_TYPE.BYTES
      extra_raw += info.extra;
    }
    HostOut a, sq, mo, tm, ms, dn, dx, el;
    host_rle_uint(a, v_actor); host_delta(sq, v_seq); host_delta(mo, v_maxop); host_delta(tm, v_time); host_rle_str(ms, v_msg);
    host_rle_uint(dn, v_depsnum); host_delta(dx, v_depsidx); host_rle_uint(el, v_extralen);
    chg_cols.push_back(SaveColumn{0x01, a}); chg_cols.push_back(SaveColumn{0x03, sq}); chg_cols.push_back(SaveColumn{0x13, mo});
    chg_cols.push_back(SaveColumn{0x23, tm}); chg_cols.push_back(SaveColumn{0x35, ms}); chg_cols.push_back(SaveColumn{0x40, dn});
    chg_cols.push_back(SaveColumn{0x43, dx}); chg_cols.push_back(SaveColumn{0x56, el});
    chg_cols.push_back(SaveColumn{0x57, std::vector<uint8_t>(extra_raw.begin(), extra_raw.end())});
    HostOut t;
    for (size_t k = 0; k + 32 <= c->heads.size(); k += 32) {
      Hash32 h;
      memcpy(h.b, &c->heads[k], 32);
      auto hi = index_of.find(h);
      if (hi == index_of.end()) return fail(c, AM355_E_STATE, "head is not an applied change");
      t.uleb(hi->second);
    }
    tail = t;
  }
  lap("change metadata");
  // ---- document chunk (columnar.js:983-1004): actors, heads, the two column directories, column data, head indexes ----
  {
    // DEFLATE is the one sequential codec the format imposes; columns are independent streams, so the big ones get a host
    // thread each (output per column is unchanged)
    std::vector<SaveColumn*> all;
    for (auto& col : chg_cols) all.push_back(&col);
    for (auto& col : ops_cols) all.push_back(&col);
    std::vector<std::thread> workers;
    std::vector<int> ok(all.size(), 1);
    for (size_t k = 0; k < all.size(); k++) {
      if (all[k]->data.size() >= (64u << 10)) workers.emplace_back([&, k]() { ok[k] = deflate_column(*all[k]) ? 1 : 0; });
      else ok[k] = deflate_column(*all[k]) ? 1 : 0;
    }
    for (auto& t : workers) t.join();
    for (int v : ok)
      if (!v) return fail(c, AM355_E_NOMEM, "deflate failed");
  }
  lap("deflate");
  HostOut body;
  body.uleb(NA);
  for (uint32_t i = 0; i < NA; i++) { const std::string& id = c->actors[actor_by_doc[i]]; body.uleb(id.size()); body.bytes(id.data(), id.size()); }
  body.uleb(c->heads.size() / 32);
  body.bytes(c->heads.data(), c->heads.size());
  auto directory = [&](const std::vector<SaveColumn>& cols) {
    size_t n = 0;
    for (auto& col : cols) n += col.data.empty() ? 0 : 1;
    body.uleb(n);
    for (auto& col : cols)
      if (!col.data.empty()) { body.uleb(col.id); body.uleb(col.data.size()); }
  };
  directory(chg_cols);
  directory(ops_cols);
  for (auto& col : chg_cols) body.bytes(col.data.data(), col.data.size());
  for (auto& col : ops_cols) body.bytes(col.data.data(), col.data.size());
  body.bytes(tail.data(), tail.size());
  HostOut chunk;  // [type][LEB len][body]: the part the checksum covers (columnar.js:664-686)
  chunk.push_back(0);
  chunk.uleb(body.size());
  chunk.bytes(body.data(), body.size());
  uint8_t digest[32];
  sha256_digest(chunk.data(), chunk.size(), digest);
  c->saved.clear();
  static const uint8_t magic[4] = {0x85, 0x6f, 0x4a, 0x83};
  c->saved.insert(c->saved.end(), magic, magic + 4);
  c->saved.insert(c->saved.end(), digest, digest + 4);
  c->saved.insert(c->saved.end(), chunk.begin(), chunk.end());
  lap("assembly + checksum");
  *out_bytes = c->saved.data();
  *out_len = c->saved.size();
  return AM355_OK;
}

// ---- C ABI entry points of the calls that allocate with the input size ----
// ---------------------------------------------------------------------------------------------------------
// history of a loaded document (am355_history.cpp does the host work; the rows come from the device decode)
// ---------------------------------------------------------------------------------------------------------
int doc_changes_impl(am355_ctx* c, uint32_t flags, const uint8_t** arena, const uint64_t** offsets, uint32_t* n_changes, const uint8_t** hashes) {
  if (!c || !arena || !offsets || !n_changes || !hashes) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  if (!c->replayed || !c->is_document) return fail(c, AM355_E_STATE, "am355_load_document and am355_replay must be called first");
  (void)hipSetDevice(c->device);
  if (c->shard_world > 1) return fail(c, AM355_E_UNSUPPORTED, "am355_doc_changes on a sharded context");
  if (!(c->history_ok && c->history_flags == (flags & 1))) {
    if (c->doc_other_ops_cols) return fail(c, AM355_E_UNSUPPORTED, "document has op columns this engine does not model (child / link / unknown): history comes from the JS path");
    if (c->doc_col_rows.size() != BIG_NCOL) return fail(c, AM355_E_UNSUPPORTED, "history needs the parallel column decode (AM355_DOC_SERIAL is set)");
    const bool trace = getenv("AM355_TRACE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!trace) return;
      auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "am355_doc_changes: %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
      t0 = now;
    };
    const uint32_t N = (uint32_t)c->n_ops, P = (uint32_t)c->n_preds;
    hipStream_t st = c->stream;
    HistoryInput in;
    in.n_rows = N; in.n_succ = P;
    in.actors = &c->actors;
    in.change_columns = &c->doc_chg_cols;
    in.doc_actor_rank = &c->doc_actor_rank;
    in.heads = c->heads.data(); in.n_heads = (uint32_t)(c->heads.size() / 32);
    {
      // the reference reads rows until EVERY column is exhausted (columnar.js:577-590 decodeColumns): a column holding more values
      // than the action column makes extra, empty rows there. The per-row columns must hold N values or none.
      static const int per_row[] = {BC_OBJ_ACTOR, BC_OBJ_CTR, BC_KEY_ACTOR, BC_KEY_CTR, BC_ID_ACTOR, BC_ID_CTR, BC_INSERT, BC_ACTION, BC_VAL_LEN, BC_SUCC_NUM};
      for (int k : per_row)
        if (c->doc_col_rows[k] != N && c->doc_col_rows[k] != 0) return fail(c, AM355_E_UNSUPPORTED, "op columns of unequal length: the JS path decides");
      if (c->doc_col_rows[BC_SUCC_ACTOR] != P || c->doc_col_rows[BC_SUCC_CTR] != P) return fail(c, AM355_E_UNSUPPORTED, "succ columns do not match succNum: the JS path decides");
      in.key_column = c->raw.data() + c->doc_meta.col_off[C_KEY_STR];
      in.key_column_len = c->doc_meta.col_len[C_KEY_STR];
      in.val_raw_len = c->doc_meta.col_len[C_VAL_RAW];
    }
    // ---- host: the change metadata columns (a few thousand values) ----
    std::string err;
    HistoryMeta meta;
    int rc = history_metadata(in, meta, err);
    if (rc == HISTORY_INVALID) return fail(c, AM355_E_INVALID, "%s", err.c_str());
    if (rc) return fail(c, AM355_E_UNSUPPORTED, "%s", err.c_str());
    lap("change metadata");
    // the dependency indexes (one value per edge of the hash graph: a quarter of a million for the headline log, 1.3 ms of one host
    // thread) are needed only by the hash chain at the end: decoded on a thread of their own while the device stages run
    struct DepTask {
      std::thread t; int rc = 0; std::string err;
      ~DepTask() { if (t.joinable()) t.join(); }
    } deps;
    deps.t = std::thread([&in, &meta, &deps]() {
      // (an exception must not leave this thread: the C ABI reports it like any other failure of the call)
      try { deps.rc = history_dependencies(in, meta, deps.err); }
      catch (const std::bad_alloc&) { deps.rc = -1; }   // (out of memory is not "unsupported": the JS host must not quietly retry on its own path)
      catch (const std::exception& e) { deps.rc = HISTORY_UNSUPPORTED; deps.err = std::string("history: ") + e.what(); }
    });
    // ---- device, stage 1: ids -> slots, preds by slot, the changes' slot ranges (am355_hist.hip) ----
    const uint32_t NC = (uint32_t)meta.chg.size(), NA = (uint32_t)c->actors.size(), W = meta.word_base[NA];
    // (the keyStr output is sized from the document's key column first; a key that many changes write is held once there and once
    // per change here: the encoder then stops at the bound and reports the size it needs, and the stages run again with it)
    size_t key_bytes = c->doc_meta.col_len[C_KEY_STR];
    const size_t val_bytes = c->doc_meta.col_len[C_VAL_RAW];
    int bind_attempt = 0;
  bind_again:
    if (!c->d_hist.ensure(hist_bytes(N, P, NC, NA, W, key_bytes, val_bytes))) return fail(c, AM355_E_NOMEM, "device allocation failed (history)");
    HistBufs hb;
    hist_bind(hb, c->d_hist.p, N, P, NC, NA, W, key_bytes, val_bytes);
    canary_arm();
    const size_t AW = hb.AW, c1 = (size_t)NC + 1;
    // pinned staging: [word_base | act_max | chg_actor | chg_prev_max | chg_max] up, [flags | chg_base | chg_nops] down, then
    // [sorted_base | sorted_chg] up and [flags | col_len | col_off | abits | column bytes] down
    size_t col_total_cap = 0;
    for (int k = 0; k < HIST_NCOL; k++) col_total_cap += hb.col_cap[k] + 256;
    const size_t up_words = 2 * ((size_t)NA + 1) + 5 * c1, down_words = 8 + 2 * c1 + HIST_NCOL + (size_t)HIST_NCOL * 2 * (c1) + c1 * AW;
    if (!c->h_rows.ensure(4 * (up_words + down_words) + col_total_cap + 4096)) return fail(c, AM355_E_NOMEM, "host allocation failed (history)");
    uint32_t* up = c->h_rows.as<uint32_t>();
    uint32_t *u_word_base = up, *u_act_max = up + NA + 1, *u_actor = u_act_max + NA + 1, *u_prev = u_actor + c1, *u_max = u_prev + c1, *u_sbase = u_max + c1, *u_schg = u_sbase + c1;
    uint32_t* down = up + up_words;
    uint32_t *d_flags = down, *d_base = down + 8, *d_nops = d_base + c1, *d_col_len = d_nops + c1, *d_col_off = d_col_len + HIST_NCOL, *d_abits = d_col_off + (size_t)HIST_NCOL * 2 * c1;
    uint8_t* d_cols = (uint8_t*)(down + down_words);
    memcpy(u_word_base, meta.word_base.data(), 4 * ((size_t)NA + 1));
    if (NA) memcpy(u_act_max, meta.act_max.data(), 4 * (size_t)NA);
    for (uint32_t k = 0; k < NC; k++) {
      const HistoryChange& ch = meta.chg[k];
      u_actor[k] = ch.actor;
      u_prev[k] = ch.prev_same_actor == NONE32 ? 0u : (uint32_t)meta.chg[ch.prev_same_actor].max_op;
      u_max[k] = (uint32_t)ch.max_op;
    }
    HIPCHK(c, hipMemcpyAsync(hb.word_base, u_word_base, 4 * ((size_t)NA + 1), hipMemcpyHostToDevice, st));
    if (NA) HIPCHK(c, hipMemcpyAsync(hb.act_max, u_act_max, 4 * (size_t)NA, hipMemcpyHostToDevice, st));
    if (NC) {
      HIPCHK(c, hipMemcpyAsync(hb.chg_actor, u_actor, 4 * (size_t)NC, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(hb.chg_prev_max, u_prev, 4 * (size_t)NC, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(hb.chg_max, u_max, 4 * (size_t)NC, hipMemcpyHostToDevice, st));
    }
    hist_stage1(c->cols, hb, st);
    HIPCHK(c, hipMemcpyAsync(d_flags, hb.flags, 16, hipMemcpyDeviceToHost, st));
    if (NC) {
      HIPCHK(c, hipMemcpyAsync(d_base, hb.chg_base, 4 * (size_t)NC, hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipMemcpyAsync(d_nops, hb.chg_nops, 4 * (size_t)NC, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    lap("ids -> slots, preds (device)");
    if (d_flags[0] & HF_INVALID) return fail(c, AM355_E_INVALID, "operation ids of the document contradict its change metadata");
    const uint32_t M = d_flags[2], PT = d_flags[3];
    // the changes that own slots, in slot order = (actor, seq) order; every slot must belong to one of them
    uint32_t n_sorted = 0;
    {
      std::vector<uint32_t> order;
      order.reserve(NC);
      for (uint32_t k = 0; k < NC; k++) if (d_nops[k]) order.push_back(k);
      std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return d_base[x] < d_base[y]; });
      uint64_t covered = 0;
      for (uint32_t k : order) {
        if (d_base[k] != covered) return fail(c, AM355_E_INVALID, "operation ids that no change of the document accounts for");
        covered += d_nops[k];
        u_sbase[n_sorted] = d_base[k];
        u_schg[n_sorted++] = k;
      }
      if (covered != M) return fail(c, AM355_E_INVALID, "operation ids that no change of the document accounts for");
    }
    if (n_sorted) {
      HIPCHK(c, hipMemcpyAsync(hb.sorted_base, u_sbase, 4 * (size_t)n_sorted, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(hb.sorted_chg, u_schg, 4 * (size_t)n_sorted, hipMemcpyHostToDevice, st));
    }
    // ---- device, stage 2: actor tables, the changes' op columns, the twelve column encodes segmented by change ----
    c->pool->prewake(c->pool->size(), 4000);   // (the host threads assemble and hash right behind it: they poll instead of sleeping until then)
    hb.P = PT;   // (pred entries = succ entries the slots account for)
    hist_stage2(c->cols, c->d_arena.as<uint8_t>(), c->raw.size(), hb, n_sorted, M, st);
    HIPCHK(c, hipMemcpyAsync(d_flags, hb.flags, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(d_col_len, hb.col_len, 4 * HIST_NCOL, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(d_col_off, hb.col_off, 4 * (size_t)HIST_NCOL * 2 * c1, hipMemcpyDeviceToHost, st));
    if (NC) HIPCHK(c, hipMemcpyAsync(d_abits, hb.abits, 4 * (size_t)NC * AW, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (d_col_len[4] > hb.col_cap[4] && bind_attempt == 0) {
      bind_attempt = 1;
      key_bytes = (size_t)d_col_len[4] + 64;
      lap("key column beyond its bound: bound again");
      goto bind_again;
    }
    if (d_flags[0] & HF_INVALID) return fail(c, AM355_E_INVALID, "the document's rows do not re-encode (an operation the reference throws on)");
    if (d_flags[0] & HF_UNSUPPORTED) return fail(c, AM355_E_UNSUPPORTED, "a value or key the reference does not re-encode byte for byte: the JS path decides");
    // (the value bytes of the rows must cover the valRaw column exactly: a longer column makes extra rows in the reference)
    if (d_col_len[8] != in.val_raw_len) return fail(c, AM355_E_UNSUPPORTED, "value bytes do not cover the valRaw column: the JS path decides");
    HistoryPieces pc;
    pc.chg_nops = d_nops; pc.abits = d_abits; pc.aw = (uint32_t)AW; pc.col_off = d_col_off;
    {
      uint8_t* q = d_cols;
      for (int k = 0; k < HIST_NCOL; k++) {
        pc.col_bytes[k] = q;
        if (d_col_len[k] > hb.col_cap[k]) return fail(c, AM355_E_DEVICE, "internal: encoded column larger than its bound");
        if (d_col_len[k]) HIPCHK(c, hipMemcpyAsync(q, hb.col_out[k], d_col_len[k], hipMemcpyDeviceToHost, st));
        q += ((size_t)d_col_len[k] + 255) & ~(size_t)255;
      }
      HIPCHK(c, hipStreamSynchronize(st));
    }
    lap("columns of all changes (device)");
    c->history = HistoryOutput{};
    deps.t.join();
    if (deps.rc == -1) return fail(c, AM355_E_NOMEM, "history: out of host memory");
    if (deps.rc == HISTORY_INVALID) return fail(c, AM355_E_INVALID, "%s", deps.err.c_str());
    if (deps.rc) return fail(c, AM355_E_UNSUPPORTED, "%s", deps.err.c_str());
    lap("dependency indexes joined");
    rc = history_finish(in, meta, pc, (flags & 1) != 0, [&](unsigned k, const std::function<void(unsigned)>& fn) { c->pool->run(k, fn); }, c->history, err);
    lap("headers + hash chain");
    if (rc == HISTORY_INVALID) return fail(c, AM355_E_INVALID, "%s", err.c_str());
    if (rc) return fail(c, AM355_E_UNSUPPORTED, "%s", err.c_str());
    c->history_ok = true;
    c->history_flags = flags & 1;
  }
  *arena = c->history.arena.data();
  *offsets = c->history.offsets.data();
  *n_changes = (uint32_t)(c->history.offsets.size() - 1);
  *hashes = c->history.hashes.data();
  return AM355_OK;
}

