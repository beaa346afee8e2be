// Staging of the C ABI (include/am355.h): am355_load_changes (gather into the pinned arena, inflate, H2D) and the host part of
// am355_load_document (container, header, checksum, inflate of the columns, change metadata scan). See am355_ctx.h.
#include "am355_ctx.h"
#include "am355_pinflate.h"

// ---------------------------------------------------------------------------------------------------------
// staging
// ---------------------------------------------------------------------------------------------------------
bool read_uleb_host(const uint8_t* p, size_t len, size_t& off, uint64_t& out) {
  uint64_t v = 0;
  int shift = 0;
  while (off < len && shift < 64) {
    uint8_t b = p[off++];
    v |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    if (!(b & 0x80)) { out = v; return true; }
  }
  return false;
}

constexpr size_t INFLATE_CAP = 0xfff00000ull;  // one staged batch / document is addressed with 32-bit arena offsets

// keep_staged: the changes staged so far stay where they are -- in the pinned arena and in HBM -- and the batch goes behind them
// (am355_apply_changes onto a state whose changes were all applied in the order they are staged: only the batch crosses the link)
int flush_uploads(am355_ctx* c) {
  if (!c->pending_up.n) return AM355_OK;
  launch_copy_ranges(c->pending_up, c->stream);
  c->pending_up.n = 0;
  HIPCHK(c, hipGetLastError());
  return AM355_OK;
}

int queue_upload(am355_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!bytes) return AM355_OK;
  static const bool off = getenv("AM355_NO_UPLOAD_QUEUE") != nullptr;   // (A/B: one copy command per upload, as before)
  if (off || bytes > ((size_t)256 << 10)) {
    int rc = flush_uploads(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return AM355_OK;
  }
  if (c->pending_up.n == 8) { int rc = flush_uploads(c); if (rc) return rc; }
  c->pending_up.add(dst, src, bytes);
  return AM355_OK;
}

int upload_offsets(am355_ctx* c) {
  if (c->offsets_on_device) return AM355_OK;
  const size_t bytes = sizeof(uint64_t) * ((size_t)c->n_changes + 1);
  if (!c->d_offsets.ensure(bytes) || !c->h_offsets.ensure(bytes)) return fail(c, AM355_E_NOMEM, "device allocation failed");
  memcpy(c->h_offsets.p, c->raw_off.data(), bytes);  // (pinned mirror: the copy below must not bounce through the driver)
  HIPCHK(c, hipMemcpyAsync(c->d_offsets.p, c->h_offsets.p, bytes, hipMemcpyHostToDevice, c->stream));
  c->offsets_on_device = true;
  return AM355_OK;
}

int load_changes_impl(am355_ctx* c, const uint8_t* arena, const uint64_t* offsets, uint32_t n, bool keep_staged) {
  if (!c || (!arena && n) || !offsets) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  (void)hipSetDevice(c->device);
  (void)c->doc_sum.wait();
  const uint32_t k0 = keep_staged ? c->n_changes : 0;  // changes and bytes kept in front of the batch
  const size_t b0 = keep_staged ? c->raw.size() : 0;
  { int frc = flush_uploads(c); if (frc) return frc; }   // (none expected: a call that queued uploads launched them)
  if (c->staging_in_flight) { c->staging_in_flight = false; HIPCHK(c, hipStreamSynchronize(c->stream)); }  // (copies of the previous batch still read the pinned arena)
  c->staged = c->replayed = c->ir_fetched = false;
  if (!keep_staged) { c->resident_valid = false; c->arena_epoch++; }
  c->apply_ready = false;
  c->state_checked = false;
  c->is_document = false;
  c->flags = 0;
  if (n && offsets[n] - offsets[0] >= ((uint64_t)1 << 20))
    c->pool->prewake(offsets[1] - offsets[0] > 9 && arena[offsets[0] + 8] == 2 ? c->pool->size() : 4);  // (compressed changes: every thread inflates)
  for (uint32_t i = 0; i < n; i++)
    if (offsets[i] > offsets[i + 1]) return fail(c, AM355_E_ARG, "change offsets must be ascending (offsets[%u] > offsets[%u])", i, i + 1);
  if (offsets[n] - offsets[0] + b0 >= INFLATE_CAP) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "batch larger than 4 GiB (32-bit arena offsets)"); }
  // ---- gather into the pinned raw arena + H2D, in slices handled by the host pool ----
  // Slice k covers a contiguous run of changes of about equal bytes. Phase A (parallel): changes of chunk type 2 are inflated
  // and their uncompressed containers rebuilt (columnar.js:813-823; checksum / hash are over that form) into a slice-local
  // buffer; a slice without compressed changes has nothing to do. Then the slice sizes are summed (host, O(slices)) and
  // phase B (parallel) copies every slice to its place in the pinned arena, fills its offsets and enqueues its H2D copy, so
  // that the DMA engine works on early slices while the host threads are still gathering later ones.
  const size_t in_bytes = (size_t)(offsets[n] - offsets[0]);
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "load_changes: %-22s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  unsigned n_slices = 1;
  // (AM355_SLICE_BYTES: bytes per slice, 2 MiB by default -- every H2D copy has a fixed cost of some microseconds --; the tests lower it to exercise the sliced path on small inputs)
  const char* slice_env = getenv("AM355_SLICE_BYTES");
  const size_t slice_bytes = slice_env && atol(slice_env) > 0 ? (size_t)atol(slice_env) : (size_t)1 << 21;
  // compressed changes (chunk type 2; the first and the middle change are taken as representative) are inflated slice by slice on
  // the host threads: many small slices keep all of them busy (zlib runs at a few hundred MB/s per thread)
  const bool deflated = n && ((offsets[1] - offsets[0] > 9 && arena[offsets[0] + 8] == 2) || (offsets[n / 2 + 1] - offsets[n / 2] > 9 && arena[offsets[n / 2] + 8] == 2));
  const size_t per_slice = deflated && !slice_env ? (size_t)64 << 10 : slice_bytes;
  if (in_bytes >= 2 * per_slice && n >= 16)
    n_slices = (unsigned)std::min<size_t>({(size_t)(c->pool->size() + 1) * (deflated ? 8 : 2), in_bytes / per_slice, (size_t)n / 8});
  if (n_slices < 1) n_slices = 1;
  struct Slice { uint32_t c0 = 0, c1 = 0; size_t out_bytes = 0, base = 0; bool any_deflated = false; int err = 0; uint32_t err_change = 0; std::vector<uint8_t> tmp; std::vector<uint32_t> tmp_len; };
  std::vector<Slice> slices(n_slices);
  {
    uint32_t ci = 0;
    for (unsigned k = 0; k < n_slices; k++) {
      slices[k].c0 = ci;
      uint64_t target = offsets[0] + (uint64_t)in_bytes * (k + 1) / n_slices;
      while (ci < n && (k + 1 == n_slices || offsets[ci + 1] <= target)) ci++;
      slices[k].c1 = ci;
    }
    slices[n_slices - 1].c1 = n;
  }
  auto phase_a = [&](unsigned k) {
    Slice& sl = slices[k];
    for (uint32_t i = sl.c0; i < sl.c1; i++) {
      size_t len = (size_t)(offsets[i + 1] - offsets[i]);
      if (len > 9 && arena[offsets[i] + 8] == 2) { sl.any_deflated = true; break; }
    }
    if (!sl.any_deflated) { sl.out_bytes = (size_t)(offsets[sl.c1] - offsets[sl.c0]); return; }
    sl.tmp_len.resize(sl.c1 - sl.c0);
    std::vector<uint8_t> out;
    for (uint32_t i = sl.c0; i < sl.c1 && !sl.err; i++) {
      const uint8_t* p = arena + offsets[i];
      size_t len = (size_t)(offsets[i + 1] - offsets[i]);
      size_t before = sl.tmp.size();
      if (len > 9 && p[8] == 2) {
        size_t off = 9;
        uint64_t clen;
        if (!read_uleb_host(p, len, off, clen) || clen > len - off) { sl.err = 10; sl.err_change = i; break; }
        int irc = inflate_raw(p + off, (size_t)clen, out, INFLATE_CAP);
        if (irc) { sl.err = irc; sl.err_change = i; break; }
        sl.tmp.insert(sl.tmp.end(), p, p + 8);
        sl.tmp.push_back(1);
        uint64_t v = out.size();
        do { uint8_t x = v & 0x7f; v >>= 7; if (v) x |= 0x80; sl.tmp.push_back(x); } while (v);
        sl.tmp.insert(sl.tmp.end(), out.begin(), out.end());
      } else {
        sl.tmp.insert(sl.tmp.end(), p, p + len);
      }
      if (sl.tmp.size() >= INFLATE_CAP) { sl.err = 2; sl.err_change = i; break; }
      sl.tmp_len[i - sl.c0] = (uint32_t)(sl.tmp.size() - before);
    }
    sl.out_bytes = sl.tmp.size();
  };
  c->pool->run(n_slices, phase_a);
  lap("inflate / sizes");
  size_t total = 0;
  for (Slice& sl : slices) {
    if (sl.err == 10) { c->flags |= AM355_F_BAD_CHUNK; return fail(c, AM355_E_INVALID, "change %u: bad deflate container", sl.err_change); }
    if (sl.err == 3) return fail(c, AM355_E_NOMEM, "inflate: out of memory");
    if (sl.err == 2) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "change %u: inflated size beyond the 4 GiB batch limit", sl.err_change); }
    if (sl.err) { c->flags |= AM355_F_BAD_DEFLATE; return fail(c, AM355_E_INVALID, "change %u: invalid or truncated deflate data", sl.err_change); }
    sl.base = total;
    total += sl.out_bytes;
    if (total + b0 >= INFLATE_CAP) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "batch larger than 4 GiB (32-bit arena offsets)"); }
  }
  const uint32_t n_all = k0 + n;
  c->raw.resize(b0 + total);
  c->raw_off.resize((size_t)n_all + 1);
  c->raw_off[n_all] = b0 + total;
  c->n_changes = n_all;
  if (!(keep_staged ? c->d_arena.ensure_keep(b0 + total + 64, b0) : c->d_arena.ensure(total + 64)) || !c->d_offsets.ensure(sizeof(uint64_t) * ((size_t)n_all + 1)) ||
      !(keep_staged ? c->d_metas.ensure_keep(sizeof(ChangeMeta) * (size_t)std::max(n_all, 1u), sizeof(ChangeMeta) * (size_t)k0)
                    : c->d_metas.ensure(sizeof(ChangeMeta) * (size_t)std::max(n_all, 1u))) || !c->h_metas.ensure(sizeof(ChangeMeta) * (size_t)std::max(n_all, 1u)) ||
      !c->d_counts.ensure(sizeof(Counts)) || !c->h_counts.ensure(2 * sizeof(Counts)) || !c->h_offsets.ensure(sizeof(uint64_t) * ((size_t)n_all + 1)))
    return fail(c, AM355_E_NOMEM, "device allocation failed");
  // (from here on `raw`, `d_raw` and `roff` address the batch's part: byte b0 of the arena, entry k0 of the offsets)
  uint8_t* raw = c->raw.data() + b0;
  uint8_t* d_raw = c->d_arena.as<uint8_t>() + b0;
  uint64_t* roff = c->raw_off.data() + k0;
  std::vector<hipError_t> h2d(n_slices, hipSuccess);
  bool any_deflated = false;
  for (const Slice& sl : slices) any_deflated = any_deflated || sl.any_deflated;
  const char* gather_env = getenv("AM355_GATHER_UNIT");  // (tests: bytes per copy unit, lowered to run the grouped gather on small inputs)
  if (!any_deflated && (total >= ((size_t)4 << 20) || (gather_env && total > 0))) {
    // Plain changes: the arena is one contiguous copy of the input, pageable -> pinned by host threads, pinned -> HBM by the DMA
    // engine, pipelined. Measured on the EPYC 9575F host (tools/micro/pinned_memcpy.cpp, profiles/r02_ab_staging_*): one thread
    // copies 16 MiB into pinned memory in 0.27 ms, one 16 MiB H2D command takes 0.30 ms (56 GB/s: the link), every H2D command
    // costs ~10 us whatever its size, and a sleeping pool thread needs ~0.1 ms to start working. So: the CALLING thread starts
    // copying at once and is the one that enqueues; units of 256 KiB are drawn from a shared counter by the caller and four pool
    // threads (a thread's FIRST unit runs at a fraction of the later rate: cold source lines); the first DMA command goes out
    // after one unit, every following one covers twice as much, up to 8 MiB -- few commands, none waiting for its bytes.
    const size_t unit = gather_env && atol(gather_env) > 0 ? (size_t)atol(gather_env) : (size_t)256 << 10;
    const size_t n_units = (total + unit - 1) / unit;
    std::vector<uint32_t> group_of(n_units);
    std::vector<size_t> group_first;  // first unit of each group (+ end)
    for (size_t u = 0, span = 1; u < n_units; span = std::min<size_t>(span * 2, 64)) {
      group_first.push_back(u);
      for (size_t k = 0; k < span && u < n_units; k++, u++) group_of[u] = (uint32_t)group_first.size() - 1;
    }
    const size_t n_groups = group_first.size();
    group_first.push_back(n_units);
    std::vector<std::atomic<uint32_t>> left(n_groups);
    for (size_t g = 0; g < n_groups; g++) left[g].store((uint32_t)(group_first[g + 1] - group_first[g]));
    h2d.assign(n_groups, hipSuccess);
    const uint8_t* src = arena + offsets[0];
    const uint64_t off0 = offsets[0];
    std::atomic<size_t> next_unit{0};
    const unsigned n_helpers = (unsigned)std::min<size_t>(4, std::min<size_t>(n_units > 1 ? n_units - 1 : 0, c->pool->size()));
    lap("  buffers ready");
    c->pool->run(n_helpers + 1, [&](unsigned task) {
      const bool issuer = task == 0;  // (the calling thread: it draws the first index)
      size_t next_group = 0;
      auto issue_ready = [&]() {
        while (next_group < n_groups && left[next_group].load(std::memory_order_acquire) == 0) {
          size_t gb = group_first[next_group] * unit, ge = std::min(total, group_first[next_group + 1] * unit);
          h2d[next_group] = hipMemcpyAsync(d_raw + gb, raw + gb, ge - gb, hipMemcpyHostToDevice, c->stream);
          if (trace) fprintf(stderr, "load_changes:   group %zu (%zu KiB) enqueued +%8.3f ms\n", next_group, (ge - gb) >> 10, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
          next_group++;
        }
      };
      if (issuer) (void)hipSetDevice(c->device);
      for (;;) {
        size_t u = next_unit.fetch_add(1, std::memory_order_relaxed);
        if (u >= n_units) break;
        size_t b = u * unit, e = std::min(total, b + unit);
        memcpy(raw + b, src + b, e - b);
        left[group_of[u]].fetch_sub(1, std::memory_order_acq_rel);
        if (issuer) issue_ready();
      }
      if (issuer) {
        while (next_group < n_groups) {
          issue_ready();
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      }
    });
    for (uint32_t i = 0; i < n; i++) roff[i] = b0 + (offsets[i] - off0);
  } else {
    // slices (inflated or plain) to their place in the arena in parallel; the H2D copies go out in few large commands: consecutive
    // slices are grouped to >= 2 MiB and one thread (task 0) enqueues a group as soon as its slices have landed
    std::vector<uint32_t> group_of(n_slices);
    std::vector<size_t> group_begin{0};
    {
      size_t acc = 0;
      for (unsigned k = 0; k < n_slices; k++) {
        group_of[k] = (uint32_t)group_begin.size() - 1;
        acc += slices[k].out_bytes;
        if (acc >= ((size_t)2 << 20) && k + 1 < n_slices) { group_begin.push_back(slices[k + 1].base); acc = 0; }
      }
      group_begin.push_back(total);
    }
    const size_t n_groups = group_begin.size() - 1;
    std::vector<std::atomic<uint32_t>> left(n_groups);
    for (auto& x : left) x.store(0);
    for (unsigned k = 0; k < n_slices; k++) left[group_of[k]].fetch_add(1);
    h2d.assign(n_groups, hipSuccess);
    if (n_slices == 1) {
      // a small batch (Backend.applyChanges with a change or two): copied and enqueued by the calling thread -- handing one of the two
      // tasks to a pool thread means waiting ~50 us for a sleeping thread to wake up and copy a kilobyte
      Slice& sl = slices[0];
      if (sl.any_deflated) {
        if (sl.out_bytes) memcpy(raw, sl.tmp.data(), sl.out_bytes);
        size_t o = b0;
        for (uint32_t i = 0; i < n; i++) { roff[i] = o; o += sl.tmp_len[i]; }
      } else {
        if (sl.out_bytes) memcpy(raw, arena + offsets[0], sl.out_bytes);
        for (uint32_t i = 0; i < n; i++) roff[i] = b0 + (offsets[i] - offsets[0]);
      }
      if (total) {
        // (behind a kept state the bytes wait for the replay's first launch -- its tables ride in the same copy kernel)
        if (keep_staged) { int qrc = queue_upload(c, d_raw, raw, total); if (qrc) return qrc; }
        else h2d[0] = hipMemcpyAsync(d_raw, raw, total, hipMemcpyHostToDevice, c->stream);
      }
    } else
    c->pool->run(n_slices + 1, [&](unsigned task) {
      if (task == 0) {
        (void)hipSetDevice(c->device);
        for (size_t g = 0; g < n_groups; g++) {
          while (left[g].load(std::memory_order_acquire) != 0) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
          }
          size_t gb = group_begin[g], ge = group_begin[g + 1];
          if (ge > gb) h2d[g] = hipMemcpyAsync(d_raw + gb, raw + gb, ge - gb, hipMemcpyHostToDevice, c->stream);
        }
        return;
      }
      Slice& sl = slices[task - 1];
      if (sl.any_deflated) {
        if (sl.out_bytes) memcpy(raw + sl.base, sl.tmp.data(), sl.out_bytes);
        size_t o = b0 + sl.base;
        for (uint32_t i = sl.c0; i < sl.c1; i++) { roff[i] = o; o += sl.tmp_len[i - sl.c0]; }
      } else {
        if (sl.out_bytes) memcpy(raw + sl.base, arena + offsets[sl.c0], sl.out_bytes);
        for (uint32_t i = sl.c0; i < sl.c1; i++) roff[i] = b0 + sl.base + (offsets[i] - offsets[sl.c0]);
      }
      left[group_of[task - 1]].fetch_sub(1, std::memory_order_acq_rel);
    });
  }
  lap("gathered, H2D enqueued");
  for (hipError_t e : h2d)
    if (e != hipSuccess) return fail(c, AM355_E_DEVICE, "hipMemcpyAsync (arena): %s", hipGetErrorString(e));
  // the offsets table is what the device's own parse / hash kernels walk: a batch that goes behind a kept state may never need it
  // (am355_replay.hip replay_resident reads the headers on the host) -- upload_offsets() when a full replay does
  c->offsets_on_device = false;
  if (!keep_staged) { int orc = upload_offsets(c); if (orc) return orc; }
  // No wait here: am355_replay enqueues behind these copies on the same stream, so its host-side set-up runs beside the tail of
  // the DMA instead of after a wake-up. The pinned arena is only rewritten by the next load, which waits first.
  c->staging_in_flight = true;
  static const bool stage_sync = getenv("AM355_STAGE_SYNC") != nullptr;  // (diagnostic: wait for the copies here, as round 1 did)
  if (trace || stage_sync) { c->staging_in_flight = false; HIPCHK(c, hipStreamSynchronize(c->stream)); lap("H2D done"); }
  c->staged = true;
  c->stats = am355_stats{};
  c->stats.n_changes = n_all;
  c->stats.raw_bytes = c->raw.size();
  return AM355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// document staging (Backend.load): columnar.js:1006-1038 decodeDocumentHeader, 1062-1067 inflateColumn,
// new.js:1645-1675 readDocumentChanges.  Host work is the container/header parse, the chunk checksum (one SHA-256
// over the whole chunk is sequential by construction), zlib inflate of the columns and the scan of the change
// metadata (clock); the op columns go to HBM for the device decode + patch.
// ---------------------------------------------------------------------------------------------------------
namespace {
// host-side RLE-uint / delta reader for the (small) change-metadata columns
struct HostRle {
  const uint8_t* p; size_t len, off = 0; int64_t count = 0; int state = 0; int64_t last = 0; bool last_null = true; bool is_signed;
  HostRle(const uint8_t* p_, size_t l, bool sg) : p(p_), len(l), is_signed(sg) {}
  bool done() const { return count == 0 && off >= len; }
  bool leb(bool sg, int64_t& out) {
    uint64_t v = 0; int shift = 0;
    while (off < len && shift < 64) {
      uint8_t b = p[off++];
      v |= (uint64_t)(b & 0x7f) << shift; shift += 7;
      if (!(b & 0x80)) { if (sg && (b & 0x40) && shift < 64) v |= ~0ull << shift; out = (int64_t)v; return true; }
    }
    return false;
  }
  bool next(bool& is_null, int64_t& v) {
    if (done()) { is_null = true; v = 0; return true; }
    if (count == 0) {
      int64_t n;
      if (!leb(true, n)) return false;
      if (n > 1) { if (!leb(is_signed, last)) return false; last_null = false; state = 1; count = n; }
      else if (n == 1) return false;
      else if (n < 0) { state = 2; count = -n; }
      else { int64_t z; if (!leb(false, z) || z <= 0) return false; state = 3; count = z; last_null = true; }
    }
    count--;
    if (state == 2) { if (!leb(is_signed, last)) return false; last_null = false; }
    is_null = last_null; v = last;
    return true;
  }
};
}  // namespace

int load_document_impl(am355_ctx* c, const uint8_t* doc, size_t len, bool defer_checksum) {
  if (!c || !doc) return c ? fail(c, AM355_E_ARG, "null argument") : AM355_E_ARG;
  (void)hipSetDevice(c->device);
  (void)c->doc_sum.wait();  // (the checksum thread of an earlier document reads doc_bytes)
  c->resident_valid = false;
  if (c->staging_in_flight) { c->staging_in_flight = false; HIPCHK(c, hipStreamSynchronize(c->stream)); }
  c->doc_graph_known = false;
  c->staged = c->replayed = c->ir_fetched = false;
  c->history_ok = false;
  c->is_document = true;
  c->flags = 0;
  const bool trace = getenv("AM355_TRACE") != nullptr;
  auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (trace) fprintf(stderr, "load_document: %-30s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  auto bad0 = [&](uint32_t flag, const char* msg) { c->flags |= flag; return fail(c, AM355_E_INVALID, "%s", msg); };
  if (len < 10 || doc[0] != 0x85 || doc[1] != 0x6f || doc[2] != 0x4a || doc[3] != 0x83) return bad0(AM355_F_BAD_MAGIC, "Data does not begin with magic bytes 85 6f 4a 83");
  size_t off = 9;
  uint64_t clen;
  if (!read_uleb_host(doc, len, off, clen) || clen != len - off) return bad0(AM355_F_BAD_CHUNK, "Encoded document has trailing data or is truncated");
  if (doc[8] != 0) return bad0(AM355_F_BAD_CHUNK, "Unexpected chunk type");
  // The chunk checksum (one SHA-256 over the whole chunk: a dependent chain, ~20 ms for 44 MB with the SHA extensions) runs on a
  // thread of its own from the first moment on, over the context's copy of the document (Backend.save of an unchanged document
  // returns these bytes, new.js:2034) as the copy tasks of the staging job below complete its pieces. The reference verifies it before
  // it reads the header (columnar.js:699-705), so whatever this function finds wrong is only reported once the checksum is known to
  // match; with `defer_checksum` a well-formed document is staged and returned without the verdict, which am355_replay then asks for.
  c->doc_bytes.resize(len);
  {
    am355_ctx::DocSum& ds = c->doc_sum;
    ds.n_pieces = (len + am355_ctx::DocSum::PIECE - 1) / am355_ctx::DocSum::PIECE;
    ds.copied.reset(new std::atomic<uint8_t>[ds.n_pieces]);
    for (size_t k = 0; k < ds.n_pieces; k++) ds.copied[k].store(0, std::memory_order_relaxed);
    ds.ok = false;
    ds.pending = true;
    const uint8_t* bytes = c->doc_bytes.data();
    const uint8_t want[4] = {doc[4], doc[5], doc[6], doc[7]};
    ds.t = std::thread([&ds, bytes, len, want0 = want[0], want1 = want[1], want2 = want[2], want3 = want[3]]() {
      uint32_t hs[8];
      sha256_initial(hs);
      size_t pos = 8;
      for (size_t k = 0; k < ds.n_pieces; k++) {
        while (!ds.copied[k].load(std::memory_order_acquire)) std::this_thread::yield();
        const size_t end = std::min(len, (k + 1) * am355_ctx::DocSum::PIECE);
        const size_t nb = end > pos ? (end - pos) / 64 : 0;
        sha256_blocks(hs, bytes + pos, nb);
        pos += nb * 64;
      }
      uint8_t digest[32];
      sha256_finish(hs, bytes + pos, len - pos, len - 8, digest);
      ds.ok = digest[0] == want0 && digest[1] == want1 && digest[2] == want2 && digest[3] == want3;
    });
  }
  // (an early exit copies what the staging job has not: the thread waits for every piece)
  auto copy_piece = [&](size_t k) {
    const size_t b = k * am355_ctx::DocSum::PIECE, e = std::min(len, b + am355_ctx::DocSum::PIECE);
    memcpy(c->doc_bytes.data() + b, doc + b, e - b);
    c->doc_sum.copied[k].store(1, std::memory_order_release);
  };
  bool copy_done = false;
  auto sum_verdict = [&]() {
    if (!copy_done) {
      for (size_t k = 0; k < c->doc_sum.n_pieces; k++)
        if (!c->doc_sum.copied[k].load(std::memory_order_acquire)) copy_piece(k);
      copy_done = true;
    }
    return c->doc_sum.wait();
  };
  auto bad = [&](uint32_t flag, const char* msg) {
    if (!sum_verdict()) { flag = AM355_F_BAD_CHECKSUM; msg = "checksum does not match data"; }
    c->flags |= flag;
    return fail(c, AM355_E_INVALID, "%s", msg);
  };
  // (failures that are not the document's fault -- memory, device -- still have to leave no thread behind)
  struct SumGuard {
    std::function<bool()>& verdict; bool armed = true;
    ~SumGuard() { if (armed) (void)verdict(); }
  };
  std::function<bool()> sum_verdict_fn = sum_verdict;
  SumGuard sum_guard{sum_verdict_fn};
  const uint8_t* h = doc + off;
  size_t hl = (size_t)clen, ho = 0;
  uint64_t na, nh;
  if (!read_uleb_host(h, hl, ho, na) || na > hl) return bad(AM355_F_BAD_LEB, "bad document header");
  c->actors.clear();
  for (uint64_t i = 0; i < na; i++) {
    uint64_t l;
    if (!read_uleb_host(h, hl, ho, l) || l > hl - ho) return bad(AM355_F_BAD_LEB, "bad document header");
    c->actors.emplace_back((const char*)h + ho, (size_t)l);
    ho += (size_t)l;
  }
  if (!read_uleb_host(h, hl, ho, nh) || nh > (hl - ho) / 32) return bad(AM355_F_BAD_LEB, "bad document header");
  c->heads.assign(h + ho, h + ho + nh * 32);
  ho += (size_t)nh * 32;
  struct Col { uint64_t id, len; std::vector<uint8_t>* data = nullptr; const uint8_t* p = nullptr; size_t n = 0; PInflateJob* pj = nullptr; uint8_t last = 0; };  // data: inflated bytes (a scratch vector of the context); pj: inflated in parallel, bytes not resolved yet (p == nullptr)
  auto read_dir = [&](std::vector<Col>& cols) -> bool {
    uint64_t n;
    if (!read_uleb_host(h, hl, ho, n) || n > hl) return false;
    int64_t last = -1;
    for (uint64_t i = 0; i < n; i++) {
      Col col;
      if (!read_uleb_host(h, hl, ho, col.id) || !read_uleb_host(h, hl, ho, col.len)) return false;
      if ((int64_t)(col.id & ~8ull) <= last) return false;  // Columns must be in ascending order (deflate bit ignored)
      last = (int64_t)(col.id & ~8ull);
      cols.push_back(std::move(col));
    }
    return true;
  };
  std::vector<Col> ccols, ocols;
  if (!read_dir(ccols) || !read_dir(ocols)) return bad(AM355_F_BAD_COLUMNS, "bad column directory");
  // column slices, then: checksum | copy of the document bytes (Backend.save of an unchanged document returns them, new.js:2034) |
  // raw-DEFLATE of every compressed column (columnar.js:1062-1067), all on the host pool. A column is ONE DEFLATE stream: the
  // 34 MB key column of the config-5 document took one thread 55 ms while sixty others had nothing to do, so long streams are
  // decoded in chunks (am355_pinflate.h: block starts found by search, back-references beyond a chunk's start as markers that
  // are resolved once the chunk in front is known) and their bytes land in the pinned arena directly.
  std::vector<Col*> all_cols;
  for (Col& col : ccols) all_cols.push_back(&col);
  for (Col& col : ocols) all_cols.push_back(&col);
  for (Col* col : all_cols) {
    if (col->len > hl - ho) return bad(AM355_F_BAD_CHUNK, "document columns exceed the chunk");
    col->p = h + ho;
    col->n = (size_t)col->len;
    ho += (size_t)col->len;
  }
  {
    std::vector<Col*> deflated, par;
    for (Col* col : all_cols)
      if (col->id & 8) deflated.push_back(col);
    std::sort(deflated.begin(), deflated.end(), [](const Col* x, const Col* y) { return x->len > y->len; });
    std::vector<int> irc(deflated.size(), 0);
    // (inflate buffers live in the context: the k-th longest column of the next document finds its pages already mapped -- a fresh
    // 34 MB vector costs ~10 ms of page faults on the thread that is the critical path of this call)
    if (c->inflate_scratch.size() < deflated.size()) c->inflate_scratch.resize(deflated.size());
    for (size_t k = 0; k < deflated.size(); k++) deflated[k]->data = &c->inflate_scratch[k];
    // which streams are worth chunks (AM355_PINFLATE_MIN: compressed bytes, tests lower it; AM355_PINFLATE=0 switches the path off)
    const char* pe = getenv("AM355_PINFLATE");
    const char* pmin_env = getenv("AM355_PINFLATE_MIN");
    const char* pchunk_env = getenv("AM355_PINFLATE_CHUNK");
    const size_t par_min = pmin_env && atol(pmin_env) > 0 ? (size_t)atol(pmin_env) : (size_t)256 << 10;
    const bool par_on = c->pool->size() >= 1 && !(pe && *pe == '0');
    size_t par_bytes = 0;
    for (Col* col : deflated)
      if (par_on && col->len >= par_min) { par.push_back(col); par_bytes += (size_t)col->len; }
    size_t chunk_bytes = std::min<size_t>((size_t)1 << 20, std::max<size_t>((size_t)128 << 10, par_bytes / (3 * ((size_t)c->pool->size() + 1))));
    if (pchunk_env && atol(pchunk_env) > 0) chunk_bytes = (size_t)atol(pchunk_env);
    while (c->pinflate_jobs.size() < par.size()) c->pinflate_jobs.emplace_back(new PInflateJob);
    std::vector<std::atomic<unsigned>> chunks_left(par.size());
    struct Task { int kind; unsigned a, b; };  // 0 search (stream a, chunk b) | 2 copy of piece a of the document | 3 decode (a, b) | 4 whole stream (deflated[a])
    std::vector<Task> tasks;
    for (size_t k = 0; k < c->doc_sum.n_pieces; k++) tasks.push_back(Task{2, (unsigned)k, 0});  // (first: the checksum thread is waiting for them)
    for (size_t s = 0; s < par.size(); s++) {
      PInflateJob* job = c->pinflate_jobs[s].get();
      job->prepare(par[s]->p, (size_t)par[s]->len, INFLATE_CAP, chunk_bytes);
      par[s]->pj = job;
      chunks_left[s].store(job->n_chunks);
      for (unsigned k = 1; k < job->n_chunks; k++) tasks.push_back(Task{0, (unsigned)s, k});
    }
    // (every search task in front of every decode task: a decode task spins for the searches of the chunks behind its own, and
    // the pool hands tasks out in index order -- all of them have been taken by a thread when a decode task starts)
    // (whole streams -- the shorter columns, longest first -- in front of the chunk decodes: a single long task must not be the last one drawn)
    for (size_t k = 0; k < deflated.size(); k++)
      if (!deflated[k]->pj) tasks.push_back(Task{4, (unsigned)k, 0});
    for (size_t s = 0; s < par.size(); s++)
      for (unsigned k = 0; k < c->pinflate_jobs[s]->n_chunks; k++) tasks.push_back(Task{3, (unsigned)s, k});
    std::atomic<int64_t> kind_end_us[5];  // (trace: when the last task of each kind ended, microseconds from the start of the call)
    for (auto& x : kind_end_us) x.store(0);
    c->pool->run((unsigned)tasks.size(), [&](unsigned t) {
      const Task& tk = tasks[t];
      struct Stamp {
        std::atomic<int64_t>* slot; std::chrono::steady_clock::time_point t0;
        ~Stamp() {
          if (!slot) return;
          int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(), prev = slot->load();
          while (prev < us && !slot->compare_exchange_weak(prev, us)) {}
        }
      } stamp{trace ? &kind_end_us[tk.kind] : nullptr, t_begin};
      switch (tk.kind) {
        case 0: c->pinflate_jobs[tk.a]->search(tk.b); break;
        case 2: copy_piece(tk.a); break;
        case 3: {
          PInflateJob* job = c->pinflate_jobs[tk.a].get();
          job->decode(tk.b);
          if (chunks_left[tk.a].fetch_sub(1, std::memory_order_acq_rel) == 1) job->link();  // (the stream's last chunk to finish links it)
          break;
        }
        default: irc[tk.a] = inflate_raw(deflated[tk.a]->p, (size_t)deflated[tk.a]->len, *deflated[tk.a]->data, INFLATE_CAP);
      }
    });
    copy_done = true;
    lap("inflate | copy");
    if (trace)
      fprintf(stderr, "load_document:   %zu tasks on %u threads (%zu streams in chunks of %zu KiB): last search +%.3f, copy +%.3f, chunk decode +%.3f, whole stream +%.3f ms\n",
              tasks.size(), c->pool->size() + 1, par.size(), chunk_bytes >> 10, kind_end_us[0].load() / 1e3, kind_end_us[2].load() / 1e3, kind_end_us[3].load() / 1e3,
              kind_end_us[4].load() / 1e3);
    // streams the chunked decode gave up on (no chain of block starts, output beyond the cap, damaged data): the ordinary
    // single-stream inflate decides what they are
    {
      std::vector<size_t> again;
      for (size_t k = 0; k < deflated.size(); k++)
        if (deflated[k]->pj && !deflated[k]->pj->ok) { deflated[k]->pj = nullptr; again.push_back(k); }
      if (!again.empty())
        c->pool->run((unsigned)again.size(), [&](unsigned t) { size_t k = again[t]; irc[k] = inflate_raw(deflated[k]->p, (size_t)deflated[k]->len, *deflated[k]->data, INFLATE_CAP); });
    }
    int rd = 0;
    for (Col* col : all_cols) {  // (errors in column order, as a sequential reader would meet them)
      if (!(col->id & 8)) continue;
      size_t k = (size_t)(std::find(deflated.begin(), deflated.end(), col) - deflated.begin());
      if (irc[k]) { rd = irc[k] == 1 ? 2 : irc[k] == 2 ? 3 : 4; break; }
      if (col->pj) { col->p = nullptr; col->n = col->pj->out_len; col->last = col->pj->last_byte; }
      else { col->p = col->data->data(); col->n = col->data->size(); col->last = col->n ? col->p[col->n - 1] : 0; }
      col->id ^= 8;
    }
    if (rd == 2) return bad(AM355_F_BAD_DEFLATE, "invalid or truncated deflate data in a document column");
    if (rd == 3) { c->flags |= AM355_F_OVERFLOW; return fail(c, AM355_E_UNSUPPORTED, "document column inflates beyond the 4 GiB limit"); }
    if (rd == 4) return fail(c, AM355_E_NOMEM, "inflate: out of memory");
  }
  // bytes of a chunk-decoded column into a vector (columns that do not go to the arena: change metadata, unknown op columns)
  auto resolve_to_vector = [&](Col& col) -> bool {
    if (!col.pj) return true;
    PInflateJob* job = col.pj;
    col.data->resize(job->out_len);
    std::vector<std::pair<unsigned, unsigned>> ps;
    for (unsigned ci = 0; ci < job->chain.size(); ci++)
      for (unsigned r = 0; r < job->n_pieces(ci); r++) ps.emplace_back(ci, r);
    c->pool->run((unsigned)ps.size(), [&](unsigned t) { job->resolve(ps[t].first, ps[t].second, col.data->data()); });
    col.pj = nullptr;
    col.p = col.data->data();
    return !job->resolve_failed.load();
  };
  for (Col& col : ccols)
    if (!resolve_to_vector(col)) return bad(AM355_F_BAD_DEFLATE, "invalid or truncated deflate data in a document column");
  // (headsIndexes and extraBytes follow; neither influences the patch: kept for am355_save. The reference reads one index per
  // head when anything follows the columns, columnar.js:1032-1034)
  if (ho < hl) {
    size_t to = ho;
    for (uint64_t i = 0; i < nh; i++) {
      uint64_t ix;
      if (!read_uleb_host(h, hl, to, ix) || ix >= (1ull << 53)) return bad(AM355_F_BAD_LEB, "bad head index after the columns");
    }
  }
  c->doc_tail.assign(h + ho, h + hl);
  c->doc_chg_cols.clear();
  for (Col& col : ccols) c->doc_chg_cols.emplace_back((uint32_t)col.id, std::vector<uint8_t>(col.p, col.p + col.n));
  c->doc_other_ops_cols = false;
  for (Col& col : ocols) {
    static const uint64_t known[] = {0x01, 0x02, 0x11, 0x13, 0x15, 0x21, 0x23, 0x34, 0x42, 0x56, 0x57, 0x80, 0x81, 0x83};
    bool k = false;
    for (uint64_t id : known) k = k || id == col.id;
    if (!k && col.n) c->doc_other_ops_cols = true;
    if (!k && col.pj) {
      // its bytes are not looked at (the patch does not depend on them and am355_save refuses such documents) -- but the reference
      // inflates every column (columnar.js:1026-1030) and throws on a back-reference in front of the stream's first byte, which the
      // chunked decode only notices when symbols are resolved: scan them for such a marker
      if (col.pj->has_too_far_marker()) return bad(AM355_F_BAD_DEFLATE, "invalid or truncated deflate data in a document column");
      col.pj = nullptr;
    }
  }

  // ---- change metadata: clock in first-appearance order, seq continuity (new.js:1645-1675) ----
  auto find = [](std::vector<Col>& cols, uint64_t id) -> Col* { for (Col& x : cols) if (x.id == id) return &x; return nullptr; };
  {
    Col* ca = find(ccols, 0x01);
    Col* cs = find(ccols, 0x03);
    HostRle ra(ca ? ca->p : nullptr, ca ? ca->n : 0, false), rs(cs ? cs->p : nullptr, cs ? cs->n : 0, true);
    std::vector<uint64_t> clock(na, 0);
    std::vector<uint8_t> seen(na, 0);
    c->clock_actor.clear();
    int64_t seq_abs = 0;
    uint32_t n_changes = 0;
    while (!ra.done()) {
      bool an, sn;
      int64_t a, dv;
      if (!ra.next(an, a) || !rs.next(sn, dv)) return bad(AM355_F_BAD_RLE, "malformed change metadata columns");
      if (an || a < 0 || (uint64_t)a >= na) return bad(AM355_F_BAD_ROW, "bad actor index in change metadata");
      if (!sn) seq_abs += dv;
      uint64_t seq = sn ? 0 : (uint64_t)seq_abs;
      if (seq != 1 && seq != clock[a] + 1 && !sum_verdict()) return bad(AM355_F_BAD_CHECKSUM, "checksum does not match data");
      if (seq != 1 && seq != clock[a] + 1) { c->flags |= AM355_F_BAD_SEQ; return fail(c, AM355_E_INVALID, "Expected seq %llu, got %llu", (unsigned long long)clock[a] + 1, (unsigned long long)seq); }
      if (!seen[a]) { seen[a] = 1; c->clock_actor.push_back((uint32_t)a); }  // document actor index for now, ranks below
      clock[a] = seq;
      n_changes++;
      if (n_changes > (1u << 26)) return fail(c, AM355_E_UNSUPPORTED, "more than 2^26 changes in one document");  // (a run length can claim any count)
    }
    c->n_changes = n_changes;
    c->clock_seq.clear();
    for (uint32_t a : c->clock_actor) c->clock_seq.push_back(clock[a]);
  }
  lap("change metadata");
  // ---- actor ranks: op-id comparison on the device is numeric on (ctr, rank) ----
  {
    std::vector<uint32_t> order(na);
    for (uint32_t i = 0; i < na; i++) order[i] = i;
    std::vector<std::string> names = c->actors;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return names[x] < names[y]; });
    c->doc_actor_rank.assign(na, 0);
    for (uint32_t r = 0; r < na; r++) { c->doc_actor_rank[order[r]] = r; c->actors[r] = names[order[r]]; }
    for (uint32_t& a : c->clock_actor) a = c->doc_actor_rank[a];
  }
  // ---- op columns -> one arena; layout recorded like a change's column directory ----
  c->raw.clear();
  c->arena_epoch++;
  c->raw_off.assign(1, 0);
  ChangeMeta& m = c->doc_meta;
  memset(&m, 0, sizeof m);
  m.n_entries = (uint32_t)na;
  // (placement first -- offsets only --, the bytes follow in parallel pieces; a column's H2D copy is enqueued by the thread that
  // finishes its last piece)
  struct Piece { const uint8_t* src; size_t dst, n; PInflateJob* pj; unsigned ci, r; uint32_t place; };
  struct Placed { size_t off, n; PInflateJob* pj; };
  std::vector<Piece> pieces;
  std::vector<Placed> placed;
  size_t arena_bytes = 0;
  uint8_t last_byte = 0;
  auto place = [&](int slot, uint64_t id) {
    Col* col = find(ocols, id);
    if (col && arena_bytes + col->n >= 0xfff00000ull) { arena_bytes = 0xfff00000ull; return; }
    m.col_off[slot] = (uint32_t)arena_bytes;
    m.col_len[slot] = col ? (uint32_t)col->n : 0;
    if (col && col->n) {
      const uint32_t pl = (uint32_t)placed.size();
      placed.push_back(Placed{arena_bytes, col->n, col->pj});
      if (col->pj) {
        for (unsigned ci = 0; ci < col->pj->chain.size(); ci++)
          for (unsigned r = 0; r < col->pj->n_pieces(ci); r++) pieces.push_back(Piece{nullptr, arena_bytes, 0, col->pj, ci, r, pl});
      } else {
        for (size_t o = 0; o < col->n; o += (size_t)4 << 20) pieces.push_back(Piece{col->p + o, arena_bytes + o, std::min<size_t>((size_t)4 << 20, col->n - o), nullptr, 0, 0, pl});
      }
      last_byte = col->last;
      arena_bytes += col->n;
    }
  };
  // the LEB-tokenisable columns first (BigCol order), the two byte-string columns after them
  static const struct { int slot; uint64_t id; uint32_t kind; } big[BIG_NCOL] = {
      {C_OBJ_ACTOR, 0x01, BK_UINT}, {C_OBJ_CTR, 0x02, BK_UINT}, {C_KEY_ACTOR, 0x11, BK_UINT}, {C_KEY_CTR, 0x13, BK_DELTA}, {C_ID_ACTOR, 0x21, BK_UINT},
      {C_ID_CTR, 0x23, BK_DELTA}, {C_INSERT, 0x34, BK_BOOL}, {C_ACTION, 0x42, BK_UINT}, {C_VAL_LEN, 0x56, BK_UINT}, {C_PRED_NUM, 0x80, BK_UINT},
      {C_PRED_ACTOR, 0x81, BK_UINT}, {C_PRED_CTR, 0x83, BK_DELTA}};
  for (int k = 0; k < BIG_NCOL; k++) {
    place(big[k].slot, big[k].id);
    c->doc_cols.off[k] = m.col_off[big[k].slot];
    c->doc_cols.len[k] = m.col_len[big[k].slot];
    c->doc_cols.kind[k] = big[k].kind;
    // every column must end on the last byte of a number (the device finds numbers by their terminating bytes)
    if (m.col_len[big[k].slot] && (last_byte & 0x80)) return bad(AM355_F_BAD_LEB, "incomplete number");
  }
  c->doc_cols.tok_bytes = (uint32_t)arena_bytes;
  place(C_KEY_STR, 0x15); place(C_VAL_RAW, 0x57);
  {
    const char* e = getenv("AM355_DOC_SERIAL");
    c->doc_serial = e && *e == '1';
  }
  if (arena_bytes >= 0xfff00000ull) return fail(c, AM355_E_UNSUPPORTED, "document larger than 4 GiB (32-bit arena offsets)");
  lap("columns placed");
  c->raw.resize(arena_bytes);
  c->raw_off.push_back(arena_bytes);
  m.len = (uint32_t)arena_bytes;
  if (!c->d_arena.ensure(arena_bytes + 64) || !c->d_metas.ensure(sizeof(ChangeMeta)) || !c->h_metas.ensure(sizeof(ChangeMeta)) ||
      !c->d_counts.ensure(sizeof(Counts)) || !c->h_counts.ensure(2 * sizeof(Counts)))
    return fail(c, AM355_E_NOMEM, "device allocation failed");
  {
    std::vector<hipError_t> h2d(placed.size(), hipSuccess);
    std::vector<std::atomic<uint32_t>> left(placed.size());
    for (auto& x : left) x.store(0);
    for (const Piece& pc : pieces) left[pc.place].fetch_add(1);
    uint8_t* raw = c->raw.data();
    lap("arena ready");
    c->pool->run((unsigned)pieces.size(), [&](unsigned k) {
      const Piece& pc = pieces[k];
      if (pc.pj) pc.pj->resolve(pc.ci, pc.r, raw + pc.dst);
      else memcpy(raw + pc.dst, pc.src, pc.n);
      if (left[pc.place].fetch_sub(1, std::memory_order_acq_rel) == 1) {
        (void)hipSetDevice(c->device);
        const Placed& pl = placed[pc.place];
        h2d[pc.place] = hipMemcpyAsync(c->d_arena.as<uint8_t>() + pl.off, raw + pl.off, pl.n, hipMemcpyHostToDevice, c->stream);
      }
    });
    for (hipError_t e : h2d)
      if (e != hipSuccess) return fail(c, AM355_E_DEVICE, "hipMemcpyAsync (document columns): %s", hipGetErrorString(e));
    for (const Placed& pl : placed)
      if (pl.pj && pl.pj->resolve_failed.load()) {  // a back-reference in front of the stream's first byte
        (void)hipStreamSynchronize(c->stream);
        return bad(AM355_F_BAD_DEFLATE, "invalid or truncated deflate data in a document column");
      }
  }
  lap("gathered, H2D enqueued");
  // (the symbol buffers of the chunked inflate -- 2 bytes per inflated byte -- are kept between loads for speed, up to 32 MB per chunk
  // and 512 MB per stream; what a large or hostile column grew beyond that goes back to the process now, beside the H2D copies: ADVICE r5)
  for (auto& job : c->pinflate_jobs) job->trim((size_t)32 << 20, (size_t)512 << 20);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  lap("H2D done");
  if (defer_checksum) sum_guard.armed = false;  // (am355_replay asks for the verdict: doc_sum.pending)
  else {
    sum_guard.armed = false;
    if (!sum_verdict()) return bad(AM355_F_BAD_CHECKSUM, "checksum does not match data");
    lap("checksum joined");
  }
  c->staged = true;
  c->stats = am355_stats{};
  c->stats.n_changes = c->n_changes;
  c->stats.raw_bytes = c->raw.size();
  return AM355_OK;
}

