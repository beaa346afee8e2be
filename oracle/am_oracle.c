/*
 * am_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See am_oracle.h.
 *
 * Sequential CPU restatement of automerge-classic's bulk change-replay path. Every function cites the
 * reference lines it follows (paths relative to the reference tree). The reference keeps the op store as
 * RLE-compressed column blocks and merges by re-encoding blocks; that storage machinery is a
 * representation detail (new.js:1304-1380, 370-561) and is replaced here by plain linked rows. What is
 * restated is the *semantics*: wire-format decoding and validation, causal scheduling, op placement
 * (object order, key order, RGA insertion rule), pred->succ resolution, and the whole-document patch
 * state machine including its key-order and coalescing behaviour.
 */
#include "am_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ===================================================================================================
 * small utilities: error handling, bump allocator, growable buffers
 * =================================================================================================*/

typedef struct {
  char msg[512];
  int set;
} err_t;

static int fail(err_t *e, const char *fmt, ...) {
  if (!e->set) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e->msg, sizeof e->msg, fmt, ap);
    va_end(ap);
    e->set = 1;
  }
  return -1;
}

typedef struct chunk {
  struct chunk *next;
  size_t used, cap;
} chunk_t;

typedef struct {
  chunk_t *head;
} pool_t;

static void *pool_alloc(pool_t *p, size_t n) {
  n = (n + 15) & ~(size_t)15;
  if (!p->head || p->head->used + n > p->head->cap) {
    size_t cap = n > (1u << 20) ? n : (1u << 20);
    chunk_t *c = (chunk_t *)malloc(sizeof(chunk_t) + 16 + cap);
    if (!c) abort();
    c->next = p->head;
    c->used = 0;
    c->cap = cap;
    p->head = c;
  }
  void *r = (char *)(p->head + 1) + p->head->used;
  p->head->used += n;
  memset(r, 0, n);
  return r;
}

static void pool_free(pool_t *p) {
  chunk_t *c = p->head;
  while (c) {
    chunk_t *n = c->next;
    free(c);
    c = n;
  }
  p->head = NULL;
}

typedef struct {
  char *p;
  size_t len, cap;
} sbuf_t;

static void sb_reserve(sbuf_t *b, size_t extra) {
  if (b->len + extra + 1 > b->cap) {
    size_t cap = b->cap ? b->cap * 2 : 4096;
    while (cap < b->len + extra + 1) cap *= 2;
    b->p = (char *)realloc(b->p, cap);
    if (!b->p) abort();
    b->cap = cap;
  }
}
static void sb_put(sbuf_t *b, const char *s, size_t n) {
  sb_reserve(b, n);
  memcpy(b->p + b->len, s, n);
  b->len += n;
  b->p[b->len] = 0;
}
static void sb_puts(sbuf_t *b, const char *s) { sb_put(b, s, strlen(s)); }
static void sb_putc(sbuf_t *b, char c) { sb_put(b, &c, 1); }

/* ===================================================================================================
 * SHA-256 (FIPS 180-4). Reference call sites: columnar.js:676-679 (encode), 699-701 (verify).
 * =================================================================================================*/

static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha_block(uint32_t h[8], const uint8_t *p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t t1 = hh + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i];
    uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void amo_sha256(const uint8_t *data, size_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t i = 0;
  for (; i + 64 <= len; i += 64) sha_block(h, data + i);
  uint8_t tail[128];
  size_t rem = len - i;
  memset(tail, 0, sizeof tail);
  memcpy(tail, data + i, rem);
  tail[rem] = 0x80;
  size_t tl = rem < 56 ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
  sha_block(h, tail);
  if (tl == 128) sha_block(h, tail + 64);
  for (int k = 0; k < 8; k++) { out[4 * k] = h[k] >> 24; out[4 * k + 1] = h[k] >> 16; out[4 * k + 2] = h[k] >> 8; out[4 * k + 3] = h[k]; }
}

/* ===================================================================================================
 * Byte decoders.  encoding.js:293-534 (Decoder), 789-920 (RLEDecoder), 1004-1051 (DeltaDecoder),
 * 1141-1207 (BooleanDecoder).
 * =================================================================================================*/

typedef struct {
  const uint8_t *buf;
  size_t len, off;
} dec_t;

#define MAX_SAFE 9007199254740991LL /* 2^53 - 1 */

/* encoding.js:389-396 readUint53 over :410-436 readUint64: LEB128, at most 10 bytes / 64 bits, then the
 * result must fit 53 bits.  Non-minimal encodings are accepted, as in the reference. */
static int read_u53(dec_t *d, uint64_t *out, err_t *e) {
  uint64_t v = 0;
  int shift = 0;
  while (d->off < d->len) {
    uint8_t b = d->buf[d->off];
    if (shift == 63 && (b & 0xfe) != 0) return fail(e, "number out of range");
    v |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    d->off++;
    if (!(b & 0x80)) {
      if (v > (uint64_t)MAX_SAFE) return fail(e, "number out of range");
      *out = v;
      return 0;
    }
  }
  return fail(e, "buffer ended with incomplete number");
}

/* encoding.js:398-408 readInt53 over :438-488 readInt64 */
static int read_i53(dec_t *d, int64_t *out, err_t *e) {
  uint64_t v = 0;
  int shift = 0;
  while (d->off < d->len) {
    uint8_t b = d->buf[d->off];
    if (shift == 63 && b != 0 && b != 0x7f) return fail(e, "number out of range");
    v |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    d->off++;
    if (!(b & 0x80)) {
      if ((b & 0x40) && shift < 64) v |= ~(uint64_t)0 << shift; /* sign extend */
      int64_t s = (int64_t)v;
      if (s > MAX_SAFE || s < -MAX_SAFE) return fail(e, "number out of range");
      *out = s;
      return 0;
    }
  }
  return fail(e, "buffer ended with incomplete number");
}

static int read_bytes(dec_t *d, size_t n, const uint8_t **out, err_t *e) {
  if (n > d->len - d->off) return fail(e, "subarray exceeds buffer size");
  *out = d->buf + d->off;
  d->off += n;
  return 0;
}

/* RLE record state machine (encoding.js:865-887 readRecord, 820-834 readValue).  type: 0 uint, 1 int, 2 utf8 */
typedef struct {
  dec_t d;
  int type;
  int state; /* 0 none, 1 repetition, 2 literal, 3 nulls */
  int64_t count;
  int last_null, have_last;
  int64_t last_int;
  const uint8_t *last_str;
  uint64_t last_len;
} rle_t;

static void rle_init(rle_t *r, int type, const uint8_t *buf, size_t len) {
  memset(r, 0, sizeof *r);
  r->d.buf = buf;
  r->d.len = len;
  r->type = type;
}
static int rle_done(const rle_t *r) { return r->count == 0 && r->d.off == r->d.len; }

typedef struct {
  int is_null;
  int64_t i;
  const uint8_t *s;
  uint64_t slen;
} rval_t;

static int rle_raw(rle_t *r, rval_t *v, err_t *e) {
  v->is_null = 0;
  if (r->type == 0) {
    uint64_t u;
    if (read_u53(&r->d, &u, e)) return -1;
    v->i = (int64_t)u;
  } else if (r->type == 1) {
    if (read_i53(&r->d, &v->i, e)) return -1;
  } else {
    uint64_t n;
    if (read_u53(&r->d, &n, e)) return -1;
    if (read_bytes(&r->d, n, &v->s, e)) return -1;
    v->slen = n;
  }
  return 0;
}

static int rle_same_as_last(const rle_t *r, const rval_t *v) {
  if (!r->have_last || r->last_null) return 0;
  if (r->type == 2) return r->last_len == v->slen && memcmp(r->last_str, v->s, v->slen) == 0;
  return r->last_int == v->i;
}
static void rle_set_last(rle_t *r, const rval_t *v) {
  r->have_last = 1;
  r->last_null = v->is_null;
  r->last_int = v->i;
  r->last_str = v->s;
  r->last_len = v->slen;
}

static int rle_read(rle_t *r, rval_t *v, err_t *e) {
  if (rle_done(r)) { v->is_null = 1; return 0; } /* reading past the end yields null (encoding.js:821) */
  if (r->count == 0) {
    int64_t n;
    if (read_i53(&r->d, &n, e)) return -1;
    if (n > 1) {
      rval_t x;
      if (rle_raw(r, &x, e)) return -1;
      if ((r->state == 1 || r->state == 2) && rle_same_as_last(r, &x)) return fail(e, "Successive repetitions with the same value are not allowed");
      r->state = 1;
      rle_set_last(r, &x);
      r->count = n;
    } else if (n == 1) {
      return fail(e, "Repetition count of 1 is not allowed, use a literal instead");
    } else if (n < 0) {
      if (r->state == 2) return fail(e, "Successive literals are not allowed");
      r->state = 2;
      r->count = -n;
    } else {
      if (r->state == 3) return fail(e, "Successive null runs are not allowed");
      uint64_t c;
      if (read_u53(&r->d, &c, e)) return -1;
      if (c == 0) return fail(e, "Zero-length null runs are not allowed");
      r->count = (int64_t)c;
      r->state = 3;
      rval_t x = {1, 0, NULL, 0};
      rle_set_last(r, &x);
    }
  }
  r->count--;
  if (r->state == 2) {
    rval_t x;
    if (rle_raw(r, &x, e)) return -1;
    if (rle_same_as_last(r, &x)) return fail(e, "Repetition of values is not allowed in literal");
    rle_set_last(r, &x);
    *v = x;
  } else {
    v->is_null = r->last_null;
    v->i = r->last_int;
    v->s = r->last_str;
    v->slen = r->last_len;
  }
  return 0;
}

/* DeltaDecoder (encoding.js:1025-1030): running sum of the non-null deltas */
typedef struct {
  rle_t r;
  int64_t abs;
} delta_t;
static void delta_init(delta_t *d, const uint8_t *buf, size_t len) { rle_init(&d->r, 1, buf, len); d->abs = 0; }
static int delta_read(delta_t *d, rval_t *v, err_t *e) {
  if (rle_read(&d->r, v, e)) return -1;
  if (!v->is_null) {
    d->abs += v->i;
    if (d->abs > MAX_SAFE || d->abs < -MAX_SAFE) return fail(e, "number out of range");
    v->i = d->abs;
  }
  return 0;
}

/* BooleanDecoder (encoding.js:1171-1183) */
typedef struct {
  dec_t d;
  int last, first;
  uint64_t count;
} bool_t;
static void bool_init(bool_t *b, const uint8_t *buf, size_t len) { memset(b, 0, sizeof *b); b->d.buf = buf; b->d.len = len; b->last = 1; b->first = 1; }
static int bool_done(const bool_t *b) { return b->count == 0 && b->d.off == b->d.len; }
static int bool_read(bool_t *b, int *v, err_t *e) {
  if (bool_done(b)) { *v = 0; return 0; }
  while (b->count == 0) {
    if (read_u53(&b->d, &b->count, e)) return -1;
    b->last = !b->last;
    if (b->count == 0 && !b->first) return fail(e, "Zero-length runs are not allowed");
    b->first = 0;
  }
  b->count--;
  *v = b->last;
  return 0;
}

/* ===================================================================================================
 * Change container + header.  columnar.js:688-708 decodeContainerHeader, 635-652 decodeChangeHeader,
 * 609-624 decodeColumnInfo, 741-765 decodeChangeColumns, 813-823 inflateChange.
 * =================================================================================================*/

#define COL_OBJ_ACTOR 0x01
#define COL_OBJ_CTR 0x02
#define COL_KEY_ACTOR 0x11
#define COL_KEY_CTR 0x13
#define COL_KEY_STR 0x15
#define COL_INSERT 0x34
#define COL_ACTION 0x42
#define COL_VAL_LEN 0x56
#define COL_VAL_RAW 0x57
#define COL_PRED_NUM 0x70
#define COL_PRED_ACTOR 0x71
#define COL_PRED_CTR 0x73

typedef struct {
  const uint8_t *p;
  size_t len;
} span_t;

typedef struct {
  const uint8_t *raw;   /* uncompressed container */
  size_t raw_len;
  uint8_t hash[32];
  uint32_t n_deps;
  const uint8_t *deps;  /* n_deps * 32 */
  span_t *actors;       /* actors[0] = author */
  uint32_t n_actors;
  uint64_t seq, start_op;
  span_t col[16];       /* indexed by slot below; absent columns have len 0 */
  int unknown_cols;
} change_t;

enum { S_OBJ_ACTOR, S_OBJ_CTR, S_KEY_ACTOR, S_KEY_CTR, S_KEY_STR, S_INSERT, S_ACTION, S_VAL_LEN, S_VAL_RAW, S_PRED_NUM, S_PRED_ACTOR, S_PRED_CTR, S_NUM };

static int col_slot(uint64_t id) {
  switch (id) {
    case COL_OBJ_ACTOR: return S_OBJ_ACTOR;
    case COL_OBJ_CTR: return S_OBJ_CTR;
    case COL_KEY_ACTOR: return S_KEY_ACTOR;
    case COL_KEY_CTR: return S_KEY_CTR;
    case COL_KEY_STR: return S_KEY_STR;
    case COL_INSERT: return S_INSERT;
    case COL_ACTION: return S_ACTION;
    case COL_VAL_LEN: return S_VAL_LEN;
    case COL_VAL_RAW: return S_VAL_RAW;
    case COL_PRED_NUM: return S_PRED_NUM;
    case COL_PRED_ACTOR: return S_PRED_ACTOR;
    case COL_PRED_CTR: return S_PRED_CTR;
    default: return -1;
  }
}

static const uint8_t MAGIC[4] = {0x85, 0x6f, 0x4a, 0x83};

static int parse_change(pool_t *pool, const uint8_t *buf, size_t len, change_t *c, err_t *e) {
  memset(c, 0, sizeof *c);
  /* columnar.js:742 -- a DEFLATEd change (chunk type 2) is first rebuilt in uncompressed form */
  if (len > 8 && buf[8] == 2) {
    if (memcmp(buf, MAGIC, 4) != 0) return fail(e, "Data does not begin with magic bytes 85 6f 4a 83");
    dec_t d = {buf, len, 9};
    uint64_t clen;
    const uint8_t *cdata;
    if (read_u53(&d, &clen, e) || read_bytes(&d, clen, &cdata, e)) return -1;
    size_t cap = clen * 4 + 1024;
    uint8_t *out = NULL;
    size_t outlen = 0;
    for (;;) {
      out = (uint8_t *)malloc(cap);
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (inflateInit2(&zs, -15) != Z_OK) { free(out); return fail(e, "inflate init failed"); }
      zs.next_in = (Bytef *)cdata; zs.avail_in = (uInt)clen;
      zs.next_out = out; zs.avail_out = (uInt)cap;
      int rc = inflate(&zs, Z_FINISH);
      outlen = zs.total_out;
      inflateEnd(&zs);
      if (rc == Z_STREAM_END) break;
      free(out);
      if (rc == Z_BUF_ERROR || rc == Z_OK) { cap *= 4; continue; }
      return fail(e, "invalid deflate data");
    }
    /* header: magic + checksum + type 1 + uLEB(len) */
    uint8_t lenb[10];
    int nl = 0;
    uint64_t v = outlen;
    do { uint8_t x = v & 0x7f; v >>= 7; if (v) x |= 0x80; lenb[nl++] = x; } while (v);
    size_t total = 9 + nl + outlen;
    uint8_t *raw = (uint8_t *)pool_alloc(pool, total);
    memcpy(raw, buf, 8);
    raw[8] = 1;
    memcpy(raw + 9, lenb, nl);
    memcpy(raw + 9 + nl, out, outlen);
    free(out);
    buf = raw;
    len = total;
  }
  c->raw = buf;
  c->raw_len = len;
  dec_t d = {buf, len, 0};
  const uint8_t *magic, *sum, *body;
  if (read_bytes(&d, 4, &magic, e)) return -1;
  if (memcmp(magic, MAGIC, 4) != 0) return fail(e, "Data does not begin with magic bytes 85 6f 4a 83");
  if (read_bytes(&d, 4, &sum, e)) return -1;
  size_t hash_start = d.off;
  const uint8_t *typ;
  if (read_bytes(&d, 1, &typ, e)) return -1;
  uint64_t clen;
  if (read_u53(&d, &clen, e) || read_bytes(&d, clen, &body, e)) return -1;
  amo_sha256(buf + hash_start, d.off - hash_start, c->hash);
  if (memcmp(c->hash, sum, 4) != 0) return fail(e, "checksum does not match data");
  if (d.off != len) return fail(e, "Encoded change has trailing data");
  if (*typ != 1) return fail(e, "Unexpected chunk type: %d", *typ);

  dec_t h = {body, clen, 0};
  uint64_t nd;
  if (read_u53(&h, &nd, e)) return -1;
  c->n_deps = (uint32_t)nd;
  if (read_bytes(&h, nd * 32, &c->deps, e)) return -1;
  uint64_t alen;
  const uint8_t *abytes;
  if (read_u53(&h, &alen, e) || read_bytes(&h, alen, &abytes, e)) return -1;
  int64_t time;
  uint64_t mlen, nother;
  const uint8_t *m;
  if (read_u53(&h, &c->seq, e) || read_u53(&h, &c->start_op, e) || read_i53(&h, &time, e)) return -1;
  if (read_u53(&h, &mlen, e) || read_bytes(&h, mlen, &m, e)) return -1;
  if (read_u53(&h, &nother, e)) return -1;
  if (nother > h.len) return fail(e, "subarray exceeds buffer size");
  c->n_actors = (uint32_t)nother + 1;
  c->actors = (span_t *)pool_alloc(pool, sizeof(span_t) * c->n_actors);
  c->actors[0].p = abytes;
  c->actors[0].len = alen;
  for (uint64_t i = 0; i < nother; i++) {
    uint64_t l;
    if (read_u53(&h, &l, e) || read_bytes(&h, l, &c->actors[i + 1].p, e)) return -1;
    c->actors[i + 1].len = l;
  }
  /* column directory (columnar.js:609-624): ids strictly ascending ignoring the deflate bit */
  uint64_t ncols;
  if (read_u53(&h, &ncols, e)) return -1;
  if (ncols > h.len) return fail(e, "subarray exceeds buffer size");
  uint64_t *ids = (uint64_t *)pool_alloc(pool, sizeof(uint64_t) * (ncols + 1) * 2);
  uint64_t *lens = ids + ncols + 1;
  int64_t last = -1;
  for (uint64_t i = 0; i < ncols; i++) {
    if (read_u53(&h, &ids[i], e) || read_u53(&h, &lens[i], e)) return -1;
    int64_t masked = (int64_t)(ids[i] & ~(uint64_t)8);
    if (masked <= last) return fail(e, "Columns must be in ascending order");
    last = masked;
  }
  for (uint64_t i = 0; i < ncols; i++) {
    if (ids[i] & 8) return fail(e, "change must not contain deflated columns");
    const uint8_t *p;
    if (read_bytes(&h, lens[i], &p, e)) return -1;
    int s = col_slot(ids[i]);
    if (s >= 0) { c->col[s].p = p; c->col[s].len = lens[i]; }
    else c->unknown_cols = 1; /* carried through untouched by the reference; irrelevant to patches */
  }
  /* anything left is `extraBytes` (columnar.js:757-760) */
  return 0;
}

/* ===================================================================================================
 * Decoded op rows.  new.js:570-610 readOperation, 678-724 readNextChangeOp.
 * =================================================================================================*/

#define NUL64 UINT64_MAX
#define NUL32 UINT32_MAX

typedef struct {
  uint64_t obj_ctr, key_ctr, val_tag_len;
  uint32_t obj_actor, key_actor, action, pred_num;
  const uint8_t *key, *val;
  uint32_t key_len; /* NUL32 = null */
  uint8_t insert;
  uint64_t pred_first; /* index into pred arrays */
} dop_t;

typedef struct {
  dop_t *ops;
  uint64_t n_ops, cap_ops;
  uint64_t *pred_ctr;
  uint32_t *pred_actor;
  uint64_t n_preds, cap_preds;
} dops_t;

static void dops_free(dops_t *o) {
  free(o->ops);
  free(o->pred_ctr);
  free(o->pred_actor);
  memset(o, 0, sizeof *o);
}

static int decode_ops(const change_t *c, dops_t *out, err_t *e) {
  rle_t objA, objC, keyA, keyS, act, vlen, pnum, pact;
  delta_t keyC, pctr;
  bool_t ins;
  rle_init(&objA, 0, c->col[S_OBJ_ACTOR].p, c->col[S_OBJ_ACTOR].len);
  rle_init(&objC, 0, c->col[S_OBJ_CTR].p, c->col[S_OBJ_CTR].len);
  rle_init(&keyA, 0, c->col[S_KEY_ACTOR].p, c->col[S_KEY_ACTOR].len);
  delta_init(&keyC, c->col[S_KEY_CTR].p, c->col[S_KEY_CTR].len);
  rle_init(&keyS, 2, c->col[S_KEY_STR].p, c->col[S_KEY_STR].len);
  bool_init(&ins, c->col[S_INSERT].p, c->col[S_INSERT].len);
  rle_init(&act, 0, c->col[S_ACTION].p, c->col[S_ACTION].len);
  rle_init(&vlen, 0, c->col[S_VAL_LEN].p, c->col[S_VAL_LEN].len);
  rle_init(&pnum, 0, c->col[S_PRED_NUM].p, c->col[S_PRED_NUM].len);
  rle_init(&pact, 0, c->col[S_PRED_ACTOR].p, c->col[S_PRED_ACTOR].len);
  delta_init(&pctr, c->col[S_PRED_CTR].p, c->col[S_PRED_CTR].len);
  dec_t raw = {c->col[S_VAL_RAW].p, c->col[S_VAL_RAW].len, 0};
  memset(out, 0, sizeof *out);
  /* rows exist while the action column has data (new.js:681,700) */
  while (!rle_done(&act)) {
    if (out->n_ops == out->cap_ops) {
      out->cap_ops = out->cap_ops ? out->cap_ops * 2 : 256;
      out->ops = (dop_t *)realloc(out->ops, out->cap_ops * sizeof(dop_t));
    }
    dop_t *op = &out->ops[out->n_ops];
    memset(op, 0, sizeof *op);
    rval_t v;
    if (rle_read(&objA, &v, e)) return -1;
    op->obj_actor = v.is_null ? NUL32 : (uint32_t)v.i;
    if (!v.is_null && (uint64_t)v.i >= c->n_actors) return fail(e, "actor index %lld out of range", (long long)v.i);
    if (rle_read(&objC, &v, e)) return -1;
    op->obj_ctr = v.is_null ? NUL64 : (uint64_t)v.i;
    if (rle_read(&keyA, &v, e)) return -1;
    op->key_actor = v.is_null ? NUL32 : (uint32_t)v.i;
    if (!v.is_null && (uint64_t)v.i >= c->n_actors) return fail(e, "actor index %lld out of range", (long long)v.i);
    if (delta_read(&keyC, &v, e)) return -1;
    op->key_ctr = v.is_null ? NUL64 : (uint64_t)v.i;
    if (!v.is_null && v.i < 0) return fail(e, "unsupported: negative key counter");
    if (rle_read(&keyS, &v, e)) return -1;
    op->key = v.is_null ? NULL : v.s;
    op->key_len = v.is_null ? NUL32 : (uint32_t)v.slen;
    int b;
    if (bool_read(&ins, &b, e)) return -1;
    op->insert = (uint8_t)b;
    if (rle_read(&act, &v, e)) return -1;
    if (v.is_null) return fail(e, "unsupported: null action");
    op->action = (uint32_t)v.i;
    if (rle_read(&vlen, &v, e)) return -1;
    op->val_tag_len = v.is_null ? 0 : (uint64_t)v.i;
    if (read_bytes(&raw, op->val_tag_len >> 4, &op->val, e)) return -1;
    if (rle_read(&pnum, &v, e)) return -1;
    op->pred_num = v.is_null ? 0 : (uint32_t)v.i;
    op->pred_first = out->n_preds;
    for (uint32_t i = 0; i < op->pred_num; i++) {
      if (out->n_preds == out->cap_preds) {
        out->cap_preds = out->cap_preds ? out->cap_preds * 2 : 256;
        out->pred_ctr = (uint64_t *)realloc(out->pred_ctr, out->cap_preds * 8);
        out->pred_actor = (uint32_t *)realloc(out->pred_actor, out->cap_preds * 4);
      }
      rval_t a, cc;
      if (rle_read(&pact, &a, e) || delta_read(&pctr, &cc, e)) return -1;
      if (a.is_null || cc.is_null) return fail(e, "unsupported: null pred");
      if ((uint64_t)a.i >= c->n_actors) return fail(e, "actor index %lld out of range", (long long)a.i);
      out->pred_actor[out->n_preds] = (uint32_t)a.i;
      out->pred_ctr[out->n_preds] = (uint64_t)cc.i;
      out->n_preds++;
    }
    /* new.js:715-723 */
    if ((op->obj_ctr == NUL64) != (op->obj_actor == NUL32)) return fail(e, "Mismatched object reference");
    if ((op->key_ctr == NUL64 && op->key_actor != NUL32) || (op->key_ctr == 0 && op->key_actor != NUL32) ||
        (op->key_ctr != NUL64 && op->key_ctr > 0 && op->key_actor == NUL32))
      return fail(e, "Mismatched operation key");
    out->n_ops++;
  }
  return 0;
}

/* ===================================================================================================
 * Document state
 * =================================================================================================*/

typedef struct {
  uint64_t ctr; /* 0 = none (_root / _head) */
  uint32_t actor;
} opid_t;

typedef struct row {
  opid_t id;
  uint8_t insert;
  uint32_t action;
  uint64_t val_tag_len;
  const uint8_t *val;
  opid_t *succ;
  uint32_t n_succ, cap_succ;
  struct row *next; /* next row of the same key / list element, ascending opId */
} row_t;

typedef struct elem {
  row_t *rows; /* first row is the insert op that created the element */
  struct elem *next;
  /* session documents (am_oracle_apply.c): node of the list's order-statistic tree -- the elements in list order, every subtree
   * knowing how many visible elements it holds, so that "visible elements in front of this one" (the visibleCount of seekToOp,
   * new.js:111-115) costs O(log n) instead of a walk from the head */
  struct elem *t_left, *t_right, *t_parent;
  uint32_t t_prio, t_vis, t_sum;
} elem_t;

typedef struct slot {
  const uint8_t *key;
  uint32_t key_len;
  row_t *rows;
  struct slot *hnext;
} slot_t;

struct chkey;
typedef struct obj {
  opid_t id;      /* ctr 0 = _root */
  int type;       /* action code of the make op; 0 for root (map) */
  elem_t head;    /* list sentinel */
  slot_t **slots; /* hash buckets */
  uint32_t n_slots, n_buckets;
  uint64_t n_elems;
  /* ---- session documents only (amo_apply_changes, am_oracle_apply.c) ---- */
  int has_meta;                 /* docState.objectMeta[objectId] exists (new.js:894-897) */
  struct obj *parent;           /* objectMeta.parentObj (NULL: _root) */
  opid_t parent_elem;           /* objectMeta.parentKey when the parent is a list / text object ... */
  const uint8_t *parent_key;    /* ... or a string key */
  uint32_t parent_key_len;
  struct chkey *children;       /* objectMeta.children: key / elemId -> { opId: value } */
  elem_t *t_root;               /* order-statistic tree over the elements (NULL until the first seek builds it) */
  int t_built;
  slot_t **sorted;              /* map keys in document order (UTF-16 code unit order, new.js:84) */
  uint32_t n_sorted, cap_sorted;
  uint32_t touch_epoch;         /* member of the `objectIds` set of the running applyChanges call */
} obj_t;

/* generic open-addressing table: 64-bit key -> pointer */
typedef struct {
  uint64_t *keys;
  void **vals;
  uint64_t cap, n;
} tab_t;

static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static void **tab_slot(tab_t *t, uint64_t key, int create) {
  if (create && (t->n + 1) * 2 > t->cap) {
    uint64_t ncap = t->cap ? t->cap * 2 : 1024;
    uint64_t *nk = (uint64_t *)calloc(ncap, 8);
    void **nv = (void **)calloc(ncap, sizeof(void *));
    for (uint64_t i = 0; i < t->cap; i++)
      if (t->vals[i]) {
        uint64_t j = mix64(t->keys[i]) & (ncap - 1);
        while (nv[j]) j = (j + 1) & (ncap - 1);
        nk[j] = t->keys[i];
        nv[j] = t->vals[i];
      }
    free(t->keys);
    free(t->vals);
    t->keys = nk;
    t->vals = nv;
    t->cap = ncap;
  }
  if (!t->cap) return NULL;
  uint64_t j = mix64(key) & (t->cap - 1);
  while (t->vals[j]) {
    if (t->keys[j] == key) return &t->vals[j];
    j = (j + 1) & (t->cap - 1);
  }
  if (!create) return NULL;
  t->keys[j] = key;
  t->n++;
  return &t->vals[j];
}
static void tab_free(tab_t *t) { free(t->keys); free(t->vals); memset(t, 0, sizeof *t); }

#define MAX_ACTORS (1u << 20)
static uint64_t idkey(opid_t id) { return id.ctr << 20 | id.actor; }

struct amo_doc {
  pool_t pool;
  uint32_t n_changes, n_applied, n_pending;
  uint8_t *hashes;        /* 32 * n_changes */
  span_t *actors;         /* document actor table, first-applied order (new.js:1434-1441) */
  uint64_t *clock;        /* seq per document actor */
  uint32_t *clock_order;  /* actors with a clock entry, in JS property order of `clock` (NULL: all actors in table order) */
  uint32_t n_clock;
  uint32_t n_actors, cap_actors;
  uint8_t *heads;         /* sorted, 32 bytes each */
  uint32_t n_heads;
  uint64_t max_op, n_ops, n_rows;
  obj_t root;
  tab_t objs;             /* idkey(make op) -> obj_t* */
  tab_t elems;            /* per (object, element id): key = mix(objkey) ^ idkey(elem) -> chain node */
  obj_t **obj_list;
  uint64_t n_objs, cap_objs;
  sbuf_t json;
  int json_done;
  /* ---- session documents (amo_init / amo_apply_changes): what a BackendDoc keeps between calls ---- */
  int session, loaded, meta_built;
  tab_t known;            /* changeIndexByHash: hashes of the applied changes */
  /* a document made by amo_load_document: haveHashGraph (new.js:1697), the hashes of the document's changes as the reference's
   * computeHashGraph would rebuild them (given by the test: amo_set_document_history), the changes applied by the calls since */
  int have_graph;
  const uint8_t *doc_hashes;
  uint32_t n_doc_hashes;
  const uint8_t **since;
  uint32_t n_since, cap_since;
  struct qchange *queue;  /* this.queue: changes waiting for a dependency */
  uint32_t n_queue;
  uint32_t actors_read;   /* actors whose first change has been read (getActorTable, new.js:1434-1451) */
  uint32_t epoch;
  sbuf_t apply_json;
};

/* actor order = order of the hex strings = bytewise order, shorter prefix first (new.js:65,1180,1198) */
static int cmp_span(span_t a, span_t b) {
  size_t n = a.len < b.len ? a.len : b.len;
  int c = n ? memcmp(a.p, b.p, n) : 0;
  if (c) return c;
  return a.len < b.len ? -1 : a.len > b.len;
}
static int cmp_opid(const amo_doc *d, opid_t a, opid_t b) {
  if (a.ctr != b.ctr) return a.ctr < b.ctr ? -1 : 1;
  if (a.actor == b.actor) return 0;
  return cmp_span(d->actors[a.actor], d->actors[b.actor]);
}

/* element index: (object, elemId) -> elem_t*.  Chained by exact comparison on collision. */
typedef struct enode {
  opid_t obj, id;
  elem_t *el;
  struct enode *next;
} enode_t;

static elem_t *find_elem(amo_doc *d, opid_t obj, opid_t id) {
  void **s = tab_slot(&d->elems, mix64(idkey(obj)) ^ idkey(id), 0);
  if (!s) return NULL;
  for (enode_t *n = (enode_t *)*s; n; n = n->next)
    if (n->obj.ctr == obj.ctr && n->obj.actor == obj.actor && n->id.ctr == id.ctr && n->id.actor == id.actor) return n->el;
  return NULL;
}
static void index_elem(amo_doc *d, opid_t obj, opid_t id, elem_t *el) {
  void **s = tab_slot(&d->elems, mix64(idkey(obj)) ^ idkey(id), 1);
  enode_t *n = (enode_t *)pool_alloc(&d->pool, sizeof *n);
  n->obj = obj; n->id = id; n->el = el; n->next = (enode_t *)*s;
  *s = n;
}

static uint32_t hash_bytes(const uint8_t *p, uint32_t n) {
  uint32_t h = 2166136261u;
  for (uint32_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}

static slot_t *find_slot(amo_doc *d, obj_t *o, const uint8_t *key, uint32_t len, int create) {
  if (o->n_buckets == 0 || (create && o->n_slots >= o->n_buckets)) {
    uint32_t nb = o->n_buckets ? o->n_buckets * 4 : 16;
    slot_t **b = (slot_t **)calloc(nb, sizeof(slot_t *));
    for (uint32_t i = 0; i < o->n_buckets; i++)
      for (slot_t *s = o->slots[i]; s;) {
        slot_t *nx = s->hnext;
        uint32_t j = hash_bytes(s->key, s->key_len) & (nb - 1);
        s->hnext = b[j];
        b[j] = s;
        s = nx;
      }
    free(o->slots);
    o->slots = b;
    o->n_buckets = nb;
  }
  uint32_t j = hash_bytes(key, len) & (o->n_buckets - 1);
  for (slot_t *s = o->slots[j]; s; s = s->hnext)
    if (s->key_len == len && memcmp(s->key, key, len) == 0) return s;
  if (!create) return NULL;
  slot_t *s = (slot_t *)pool_alloc(&d->pool, sizeof *s);
  s->key = key;
  s->key_len = len;
  s->hnext = o->slots[j];
  o->slots[j] = s;
  o->n_slots++;
  return s;
}

static int is_list_type(int action) { return action == 2 || action == 4; } /* makeList, makeText */

static void fmt_opid(const amo_doc *d, opid_t id, char *out, size_t cap) {
  int n = snprintf(out, cap, "%llu@", (unsigned long long)id.ctr);
  span_t a = d->actors[id.actor];
  for (size_t i = 0; i < a.len && (size_t)n + 2 < cap; i++) n += snprintf(out + n, cap - n, "%02x", a.p[i]);
}

/* succ lists stay sorted by (ctr, actorId) -- new.js:1173-1188 */
static void add_succ(amo_doc *d, row_t *r, opid_t id) {
  if (r->n_succ == r->cap_succ) {
    uint32_t nc = r->cap_succ ? r->cap_succ * 2 : 2;
    opid_t *ns = (opid_t *)pool_alloc(&d->pool, sizeof(opid_t) * nc);
    if (r->n_succ) memcpy(ns, r->succ, sizeof(opid_t) * r->n_succ);
    r->succ = ns;
    r->cap_succ = nc;
  }
  uint32_t j = 0;
  while (j < r->n_succ && cmp_opid(d, r->succ[j], id) < 0) j++;
  memmove(r->succ + j + 1, r->succ + j, sizeof(opid_t) * (r->n_succ - j));
  r->succ[j] = id;
  r->n_succ++;
}

/*
 * Apply one op of a change to the op store.  This is the net effect of seekToOp/seekWithinBlock
 * (new.js:227-317, 50-192) followed by mergeDocChangeOps (new.js:1052-1290) for that op.
 */
static int apply_op(amo_doc *d, const dop_t *op, opid_t id, const uint32_t *atab, const uint64_t *pctr,
                    const uint32_t *pactor, err_t *e) {
  char buf[160];
  obj_t *o = &d->root;
  opid_t objid = {0, 0};
  if (op->obj_ctr != NUL64) {
    objid.ctr = op->obj_ctr;
    objid.actor = atab[op->obj_actor];
    void **s = tab_slot(&d->objs, idkey(objid), 0);
    if (!s) { fmt_opid(d, objid, buf, sizeof buf); return fail(e, "unsupported: operation on unknown object %s", buf); }
    o = (obj_t *)*s;
  }
  row_t **chain; /* head pointer of the row list of this key / element */
  if (op->key_len != NUL32) {
    /* string key: map or table object (key order: new.js:84) */
    if (is_list_type(o->type)) return fail(e, "unsupported: string key used in a list object");
    if (op->insert) return fail(e, "unsupported: insert flag on a map operation");
    chain = &find_slot(d, o, op->key, op->key_len, 1)->rows;
  } else {
    if (!is_list_type(o->type)) return fail(e, "unsupported: list operation on a map object");
    if (op->key_ctr == NUL64) return fail(e, "unsupported: operation without key");
    if (op->insert) {
      /* RGA insertion rule (new.js:144-163): go to just after the reference element, then skip every
       * following element whose id is greater than the new one. */
      if (op->pred_num) { opid_t p = {pctr[0], atab[pactor[0]]}; fmt_opid(d, p, buf, sizeof buf); return fail(e, "no matching operation for pred: %s", buf); }
      elem_t *ref = &o->head;
      if (op->key_ctr != 0) {
        opid_t rid = {op->key_ctr, atab[op->key_actor]};
        ref = find_elem(d, objid, rid);
        if (!ref) { fmt_opid(d, rid, buf, sizeof buf); return fail(e, "Reference element not found: %s", buf); }
      }
      if (find_elem(d, objid, id)) { fmt_opid(d, id, buf, sizeof buf); return fail(e, "duplicate operation ID: %s", buf); }
      while (ref->next && cmp_opid(d, ref->next->rows->id, id) > 0) ref = ref->next;
      elem_t *el = (elem_t *)pool_alloc(&d->pool, sizeof *el);
      row_t *r = (row_t *)pool_alloc(&d->pool, sizeof *r);
      r->id = id; r->insert = 1; r->action = op->action; r->val_tag_len = op->val_tag_len; r->val = op->val;
      el->rows = r;
      el->next = ref->next;
      ref->next = el;
      o->n_elems++;
      d->n_rows++;
      index_elem(d, objid, id, el);
      goto made_row;
    }
    if (op->key_ctr == 0) return fail(e, "unsupported: non-insert operation on _head");
    opid_t eid = {op->key_ctr, atab[op->key_actor]};
    elem_t *el = find_elem(d, objid, eid);
    if (!el) { fmt_opid(d, eid, buf, sizeof buf); return fail(e, "could not find list element with ID: %s", buf); }
    chain = &el->rows;
  }
  /* pred -> succ (new.js:1173-1188); every pred must name a row of the same key/element (:1252-1258) */
  for (uint32_t i = 0; i < op->pred_num; i++) {
    opid_t p = {pctr[i], atab[pactor[i]]};
    row_t *r = *chain;
    while (r && !(r->id.ctr == p.ctr && r->id.actor == p.actor)) r = r->next;
    if (!r) { fmt_opid(d, p, buf, sizeof buf); return fail(e, "no matching operation for pred: %s", buf); }
    add_succ(d, r, id);
  }
  if (op->action == 3) {
    /* del: no row of its own, only succ entries (new.js:1205-1217) */
    if (op->pred_num == 0) return fail(e, "unsupported: del operation without pred");
    return 0;
  }
  {
    /* rows of one key/element ascend by opId (new.js:1197-1224) */
    row_t **pp = chain;
    while (*pp && cmp_opid(d, (*pp)->id, id) < 0) pp = &(*pp)->next;
    if (*pp && cmp_opid(d, (*pp)->id, id) == 0) { fmt_opid(d, id, buf, sizeof buf); return fail(e, "duplicate operation ID: %s", buf); }
    row_t *r = (row_t *)pool_alloc(&d->pool, sizeof *r);
    r->id = id; r->insert = 0; r->action = op->action; r->val_tag_len = op->val_tag_len; r->val = op->val;
    r->next = *pp;
    *pp = r;
    d->n_rows++;
  }
made_row:
  if ((op->action & 1) == 0) {
    /* make*: a new object comes into existence (objectMeta, new.js:894-897) */
    void **s = tab_slot(&d->objs, idkey(id), 1);
    if (*s) { fmt_opid(d, id, buf, sizeof buf); return fail(e, "duplicate operation ID: %s", buf); }
    obj_t *no = (obj_t *)pool_alloc(&d->pool, sizeof *no);
    no->id = id;
    no->type = (int)op->action;
    *s = no;
    if (d->n_objs == d->cap_objs) {
      d->cap_objs = d->cap_objs ? d->cap_objs * 2 : 64;
      d->obj_list = (obj_t **)realloc(d->obj_list, d->cap_objs * sizeof(obj_t *));
    }
    d->obj_list[d->n_objs++] = no;
  }
  return 0;
}

/* ===================================================================================================
 * Causal scheduling.  new.js:1550-1597 applyChanges (module function), 1797-1879 BackendDoc.applyChanges.
 * =================================================================================================*/

static uint64_t hash_prefix(const uint8_t *h) {
  uint64_t v;
  memcpy(&v, h, 8);
  return v;
}

typedef struct hnode {
  const uint8_t *hash;
  struct hnode *next;
} hnode_t;

static int hset_has(tab_t *t, const uint8_t *h) {
  void **s = tab_slot(t, hash_prefix(h), 0);
  if (!s) return 0;
  for (hnode_t *n = (hnode_t *)*s; n; n = n->next)
    if (memcmp(n->hash, h, 32) == 0) return 1;
  return 0;
}
static void hset_add(pool_t *p, tab_t *t, const uint8_t *h) {
  void **s = tab_slot(t, hash_prefix(h), 1);
  hnode_t *n = (hnode_t *)pool_alloc(p, sizeof *n);
  n->hash = h;
  n->next = (hnode_t *)*s;
  *s = n;
}
static void hset_del(tab_t *t, const uint8_t *h) {
  void **s = tab_slot(t, hash_prefix(h), 0);
  if (!s) return;
  hnode_t **pp = (hnode_t **)s;
  /* removed entries stay in the chain as tombstones (hash == NULL) */
  for (hnode_t *n = *pp; n; n = n->next)
    if (n->hash && memcmp(n->hash, h, 32) == 0) { n->hash = NULL; return; }
}
static int hset_has_live(tab_t *t, const uint8_t *h) {
  void **s = tab_slot(t, hash_prefix(h), 0);
  if (!s) return 0;
  for (hnode_t *n = (hnode_t *)*s; n; n = n->next)
    if (n->hash && memcmp(n->hash, h, 32) == 0) return 1;
  return 0;
}

static int cmp_hash32(const void *a, const void *b) { return memcmp(a, b, 32); }

static int doc_actor_index(const amo_doc *d, span_t a) {
  for (uint32_t i = 0; i < d->n_actors; i++)
    if (cmp_span(d->actors[i], a) == 0) return (int)i;
  return -1;
}

amo_doc *amo_replay(const uint8_t *arena, const uint64_t *offsets, uint32_t n, char *errbuf, size_t errcap) {
  err_t e = {{0}, 0};
  amo_doc *d = (amo_doc *)calloc(1, sizeof *d);
  d->n_changes = n;
  d->hashes = (uint8_t *)calloc(n ? n : 1, 32);
  change_t *ch = (change_t *)calloc(n ? n : 1, sizeof(change_t));
  uint32_t *queue = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t *next_q = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t *applied = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  tab_t known = {0}, heads = {0};
  uint32_t *atab = NULL, actors_read = 0;
  size_t atab_cap = 0;
  int rc = 0;

  /* decodeChangeColumns for every buffer up front (new.js:1806-1810) */
  for (uint32_t i = 0; i < n && !rc; i++) {
    rc = parse_change(&d->pool, arena + offsets[i], offsets[i + 1] - offsets[i], &ch[i], &e);
    if (!rc) memcpy(d->hashes + 32 * i, ch[i].hash, 32);
  }
  uint32_t qn = n;
  for (uint32_t i = 0; i < n; i++) queue[i] = i;
  uint32_t head_count = 0;
  const uint8_t **head_list = (const uint8_t **)malloc(sizeof(uint8_t *) * (n ? n : 1));

  while (!rc && qn > 0) {
    /* one pass of applyChanges() over the queue (new.js:1550-1597) */
    uint32_t na = 0, nq = 0;
    for (uint32_t qi = 0; qi < qn && !rc; qi++) {
      change_t *c = &ch[queue[qi]];
      if (hset_has(&known, c->hash)) continue; /* duplicate (new.js:1557) */
      int ai = doc_actor_index(d, c->actors[0]);
      uint64_t expected = (ai >= 0 ? d->clock[ai] : 0) + 1;
      int ready = 1;
      for (uint32_t k = 0; k < c->n_deps; k++)
        if (!hset_has(&known, c->deps + 32 * k)) ready = 0;
      if (!ready) { next_q[nq++] = queue[qi]; continue; }
      char hex[80];
      size_t hl = 0;
      for (size_t k = 0; k < c->actors[0].len && hl + 2 < sizeof hex; k++) hl += snprintf(hex + hl, sizeof hex - hl, "%02x", c->actors[0].p[k]);
      hex[hl] = 0;
      if (c->seq < expected) { rc = fail(&e, "Reuse of sequence number %llu for actor %s", (unsigned long long)c->seq, hex); break; }
      if (c->seq > expected) { rc = fail(&e, "Skipped sequence number %llu for actor %s", (unsigned long long)expected, hex); break; }
      if (ai < 0) {
        /* the clock gains a key now; the actor-table entry is created when its ops are read
         * (getActorTable new.js:1434-1441) -- same position, because changes are read in applied order */
        if (d->n_actors == d->cap_actors) {
          d->cap_actors = d->cap_actors ? d->cap_actors * 2 : 16;
          d->actors = (span_t *)realloc(d->actors, sizeof(span_t) * d->cap_actors);
          d->clock = (uint64_t *)realloc(d->clock, 8 * d->cap_actors);
        }
        if (d->n_actors >= MAX_ACTORS) { rc = fail(&e, "unsupported: too many actors"); break; }
        ai = (int)d->n_actors++;
        d->actors[ai] = c->actors[0];
        d->clock[ai] = 0;
      }
      d->clock[ai] = c->seq;
      hset_add(&d->pool, &known, c->hash);
      for (uint32_t k = 0; k < c->n_deps; k++) hset_del(&heads, c->deps + 32 * k);
      if (!hset_has_live(&heads, c->hash)) { hset_add(&d->pool, &heads, c->hash); head_list[head_count++] = c->hash; }
      applied[na++] = queue[qi];
    }
    /* apply the ops of every change accepted in this pass, in order (new.js:1587-1591) */
    for (uint32_t k = 0; k < na && !rc; k++) {
      change_t *c = &ch[applied[k]];
      if (c->n_actors > atab_cap) { atab_cap = c->n_actors * 2; atab = (uint32_t *)realloc(atab, atab_cap * 4); }
      /* every actor the change references must already be known to the document when the change is
       * read (new.js:1442-1449); authors join the table as their first change is read (:1435-1441) */
      for (uint32_t a = 0; a < c->n_actors && !rc; a++) {
        int idx = doc_actor_index(d, c->actors[a]);
        if (a == 0 && (uint32_t)idx + 1 > actors_read) actors_read = (uint32_t)idx + 1;
        if (idx < 0 || (uint32_t)idx >= actors_read) rc = fail(&e, "actorId is not known to document");
        else atab[a] = (uint32_t)idx;
      }
      if (rc) break;
      dops_t ops;
      rc = decode_ops(c, &ops, &e);
      for (uint64_t i = 0; i < ops.n_ops && !rc; i++) {
        opid_t id = {c->start_op + i, atab[0]};
        if (id.ctr >= ((uint64_t)1 << 44)) { rc = fail(&e, "unsupported: op counter too large"); break; }
        if (id.ctr > d->max_op) d->max_op = id.ctr; /* new.js:711 */
        rc = apply_op(d, &ops.ops[i], id, atab, ops.pred_ctr + ops.ops[i].pred_first, ops.pred_actor + ops.ops[i].pred_first, &e);
      }
      d->n_ops += ops.n_ops;
      dops_free(&ops);
    }
    d->n_applied += na;
    memcpy(queue, next_q, sizeof(uint32_t) * nq);
    qn = nq;
    if (na == 0) break; /* no progress: the rest stays queued (new.js:1833-1840) */
  }
  d->n_pending = qn;

  if (!rc) {
    /* heads = applied hashes nobody depends on, sorted (new.js:1582-1583,1593) */
    d->heads = (uint8_t *)pool_alloc(&d->pool, 32 * (size_t)(head_count ? head_count : 1));
    for (uint32_t i = 0; i < head_count; i++)
      if (hset_has_live(&heads, head_list[i])) memcpy(d->heads + 32 * d->n_heads++, head_list[i], 32);
    qsort(d->heads, d->n_heads, 32, cmp_hash32);
  }
  free(ch); free(queue); free(next_q); free(applied); free(atab); free(head_list);
  tab_free(&known);
  tab_free(&heads);
  if (rc) {
    if (errbuf && errcap) snprintf(errbuf, errcap, "%s", e.msg);
    amo_free(d);
    return NULL;
  }
  return d;
}

static void session_free(amo_doc *d);
void amo_free(amo_doc *d) {
  if (!d) return;
  session_free(d);
  for (uint64_t i = 0; i < d->n_objs; i++) free(d->obj_list[i]->slots);
  free(d->root.slots);
  free(d->obj_list);
  tab_free(&d->objs);
  tab_free(&d->elems);
  free(d->hashes);
  free(d->actors);
  free(d->clock);
  free(d->clock_order);
  free(d->json.p);
  pool_free(&d->pool);
  free(d);
}

uint32_t amo_num_changes(const amo_doc *d) { return d->n_changes; }
uint32_t amo_num_applied(const amo_doc *d) { return d->n_applied; }
const uint8_t *amo_change_hashes(const amo_doc *d) { return d->hashes; }
uint64_t amo_num_ops(const amo_doc *d) { return d->n_ops; }
uint64_t amo_max_op(const amo_doc *d) { return d->max_op; }
uint32_t amo_num_actors(const amo_doc *d) { return d->n_actors; }
const uint8_t *amo_actor(const amo_doc *d, uint32_t i, uint32_t *len) { *len = (uint32_t)d->actors[i].len; return d->actors[i].p; }

/* ===================================================================================================
 * Canonical order helpers.  Objects: _root first, then ascending (ctr, actorId) (new.js:64-65).
 * Map keys: ascending by JS string comparison = UTF-16 code unit order (new.js:84).
 * =================================================================================================*/

static const amo_doc *g_sort_doc; /* qsort context (single-threaded test tool) */

static int cmp_obj(const void *a, const void *b) {
  const obj_t *x = *(obj_t *const *)a, *y = *(obj_t *const *)b;
  return cmp_opid(g_sort_doc, x->id, y->id);
}

/* decode one UTF-8 scalar; returns length or 0 if malformed */
static int utf8_next(const uint8_t *p, size_t n, uint32_t *cp) {
  if (n == 0) return 0;
  uint8_t b = p[0];
  if (b < 0x80) { *cp = b; return 1; }
  if (b >= 0xc2 && b <= 0xdf && n >= 2 && (p[1] & 0xc0) == 0x80) { *cp = (b & 0x1f) << 6 | (p[1] & 0x3f); return 2; }
  if (b >= 0xe0 && b <= 0xef && n >= 3 && (p[1] & 0xc0) == 0x80 && (p[2] & 0xc0) == 0x80) {
    uint32_t c = (b & 0x0f) << 12 | (p[1] & 0x3f) << 6 | (p[2] & 0x3f);
    if (c < 0x800 || (c >= 0xd800 && c <= 0xdfff)) return 0;
    *cp = c;
    return 3;
  }
  if (b >= 0xf0 && b <= 0xf4 && n >= 4 && (p[1] & 0xc0) == 0x80 && (p[2] & 0xc0) == 0x80 && (p[3] & 0xc0) == 0x80) {
    uint32_t c = (b & 0x07) << 18 | (p[1] & 0x3f) << 12 | (p[2] & 0x3f) << 6 | (p[3] & 0x3f);
    if (c < 0x10000 || c > 0x10ffff) return 0;
    *cp = c;
    return 4;
  }
  return 0;
}

static int utf8_valid(const uint8_t *p, size_t n) {
  size_t i = 0;
  while (i < n) {
    uint32_t cp;
    int l = utf8_next(p + i, n - i, &cp);
    if (!l) return 0;
    i += l;
  }
  return 1;
}

/* compare two valid UTF-8 strings in UTF-16 code-unit order */
static int cmp_utf16(const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
  size_t i = 0, j = 0;
  while (i < na && j < nb) {
    uint32_t x, y;
    int lx = utf8_next(a + i, na - i, &x), ly = utf8_next(b + j, nb - j, &y);
    if (x != y) {
      /* first differing scalar: compare leading UTF-16 units, then (both supplementary) the scalar */
      uint32_t ux = x >= 0x10000 ? 0xd800 + ((x - 0x10000) >> 10) : x;
      uint32_t uy = y >= 0x10000 ? 0xd800 + ((y - 0x10000) >> 10) : y;
      if (ux != uy) return ux < uy ? -1 : 1;
      return x < y ? -1 : 1;
    }
    i += lx;
    j += ly;
  }
  if (i < na) return 1;
  if (j < nb) return -1;
  return 0;
}

static int cmp_slot(const void *a, const void *b) {
  const slot_t *x = *(slot_t *const *)a, *y = *(slot_t *const *)b;
  return cmp_utf16(x->key, x->key_len, y->key, y->key_len);
}

static slot_t **sorted_slots(const obj_t *o) {
  slot_t **v = (slot_t **)malloc(sizeof(slot_t *) * (o->n_slots ? o->n_slots : 1));
  uint32_t k = 0;
  for (uint32_t i = 0; i < o->n_buckets; i++)
    for (slot_t *s = o->slots[i]; s; s = s->hnext) v[k++] = s;
  qsort(v, k, sizeof(slot_t *), cmp_slot);
  return v;
}

static obj_t **sorted_objs(amo_doc *d) {
  obj_t **v = (obj_t **)malloc(sizeof(obj_t *) * (d->n_objs + 1));
  v[0] = &d->root;
  memcpy(v + 1, d->obj_list, sizeof(obj_t *) * d->n_objs);
  g_sort_doc = d;
  qsort(v + 1, d->n_objs, sizeof(obj_t *), cmp_obj);
  return v;
}

uint64_t amo_num_rows(const amo_doc *d) { return d->n_rows; }

void amo_rows(const amo_doc *cd, uint64_t *id_ctr, uint32_t *id_actor, uint64_t *obj_ctr, uint32_t *obj_actor,
              uint8_t *insert, uint32_t *action, uint32_t *succ_num) {
  amo_doc *d = (amo_doc *)cd;
  obj_t **objs = sorted_objs(d);
  uint64_t k = 0;
  for (uint64_t oi = 0; oi <= d->n_objs; oi++) {
    obj_t *o = objs[oi];
#define EMIT(r)                                                                     \
  do {                                                                              \
    id_ctr[k] = (r)->id.ctr; id_actor[k] = (r)->id.actor; obj_ctr[k] = o->id.ctr;   \
    obj_actor[k] = o->id.actor; insert[k] = (r)->insert; action[k] = (r)->action;   \
    succ_num[k] = (r)->n_succ; k++;                                                 \
  } while (0)
    if (is_list_type(o->type)) {
      for (elem_t *el = o->head.next; el; el = el->next)
        for (row_t *r = el->rows; r; r = r->next) EMIT(r);
    } else {
      slot_t **sl = sorted_slots(o);
      for (uint32_t i = 0; i < o->n_slots; i++)
        for (row_t *r = sl[i]->rows; r; r = r->next) EMIT(r);
      free(sl);
    }
#undef EMIT
  }
  free(objs);
}

/* ===================================================================================================
 * Whole-document patch.  new.js:1604-1635 documentPatch, 884-1040 updatePatchProperty (isWholeDoc),
 * 747-782 appendEdit, 797-823 appendUpdate, columnar.js:300-329 decodeValue.
 * =================================================================================================*/

typedef struct pobj pobj_t;

typedef struct {
  int kind; /* 0 primitive (tag/bytes), 1 counter total, 2 child object */
  uint64_t tag_len;
  const uint8_t *bytes;
  int64_t counter;
  pobj_t *obj;
} pval_t;

typedef struct {
  opid_t opid;
  pval_t val;
} pent_t;

typedef struct {
  const uint8_t *key;
  uint32_t key_len;
  pent_t *ents;
  uint32_t n, cap;
} pprop_t;

enum { E_INSERT, E_MULTI, E_UPDATE, E_REMOVE };

typedef struct {
  int action;
  uint64_t index, count;
  opid_t elem, opid;
  pval_t val;     /* insert / update */
  pval_t *vals;   /* multi-insert */
  uint64_t nvals, capvals;
} pedit_t;

struct pobj {
  opid_t id;
  int is_root, type; /* type = make action code (0 map, 2 list, 4 text, 6 table, other -> null) */
  pprop_t *props;
  uint64_t n_props, cap_props;
  pedit_t *edits;
  uint64_t n_edits, cap_edits;
};

/* value class used by appendEdit's `datatype` and `typeof` comparisons (new.js:759-760,768-769) */
static void val_class(const pval_t *v, int *datatype, int *jstype) {
  uint64_t t = v->tag_len;
  if (v->kind == 1) { *datatype = 8; *jstype = 1; return; }
  if (t == 0) { *datatype = -1; *jstype = 3; return; }               /* null: typeof 'object' */
  if (t == 1 || t == 2) { *datatype = -1; *jstype = 2; return; }      /* boolean */
  switch (t & 15) {
    case 6: *datatype = -1; *jstype = 0; return;                      /* string */
    case 3: case 4: case 5: case 8: case 9: *datatype = (int)(t & 15); *jstype = 1; return; /* number */
    default: *datatype = (int)(t & 15); *jstype = 3; return;          /* Uint8Array: object, numeric datatype */
  }
}

static pedit_t *push_edit(pobj_t *p) {
  if (p->n_edits == p->cap_edits) {
    p->cap_edits = p->cap_edits ? p->cap_edits * 2 : 16;
    p->edits = (pedit_t *)realloc(p->edits, p->cap_edits * sizeof(pedit_t));
  }
  pedit_t *e = &p->edits[p->n_edits++];
  memset(e, 0, sizeof *e);
  return e;
}

static int same_id(opid_t a, opid_t b) { return a.ctr == b.ctr && a.actor == b.actor; }

/* new.js:747-782 */
static void append_edit(pobj_t *p, const pedit_t *next) {
  if (p->n_edits > 0) {
    pedit_t *last = &p->edits[p->n_edits - 1];
    if (last->action == E_INSERT && next->action == E_INSERT && last->index + 1 == next->index && last->val.kind != 2 &&
        next->val.kind != 2 && same_id(last->elem, last->opid) && same_id(next->elem, next->opid) &&
        last->elem.actor == next->elem.actor && last->elem.ctr + 1 == next->elem.ctr) {
      int d1, t1, d2, t2;
      val_class(&last->val, &d1, &t1);
      val_class(&next->val, &d2, &t2);
      if (d1 == d2 && t1 == t2) {
        last->action = E_MULTI;
        last->capvals = 8;
        last->vals = (pval_t *)malloc(sizeof(pval_t) * last->capvals);
        last->vals[0] = last->val;
        last->vals[1] = next->val;
        last->nvals = 2;
        return;
      }
    } else if (last->action == E_MULTI && next->action == E_INSERT && last->index + last->nvals == next->index &&
               next->val.kind != 2 && same_id(next->elem, next->opid) && last->elem.actor == next->elem.actor &&
               last->elem.ctr + last->nvals == next->elem.ctr) {
      int d1, t1, d2, t2;
      val_class(&last->vals[0], &d1, &t1);
      val_class(&next->val, &d2, &t2);
      /* lastEdit.datatype was set from the SECOND value at creation (new.js:762); with equal datatypes
       * required at every step it equals the first value's */
      if (d1 == d2 && t1 == t2) {
        if (last->nvals == last->capvals) {
          last->capvals *= 2;
          last->vals = (pval_t *)realloc(last->vals, sizeof(pval_t) * last->capvals);
        }
        last->vals[last->nvals++] = next->val;
        return;
      }
    } else if (last->action == E_REMOVE && next->action == E_REMOVE && last->index == next->index) {
      last->count += next->count;
      return;
    }
  }
  *push_edit(p) = *next;
}

/* new.js:797-823 */
static void append_update(pobj_t *p, uint64_t index, opid_t elem, opid_t opid, pval_t val, int first) {
  int insert = 0;
  if (first) {
    while (!insert && p->n_edits > 0) {
      pedit_t *last = &p->edits[p->n_edits - 1];
      if ((last->action == E_INSERT || last->action == E_UPDATE) && last->index == index) {
        insert = last->action == E_INSERT;
        p->n_edits--;
      } else if (last->action == E_MULTI && last->index + last->nvals - 1 == index) {
        last->nvals--; /* NB: a 1-element multi-insert may remain, exactly as in the reference */
        insert = 1;
      } else break;
    }
  }
  pedit_t e;
  memset(&e, 0, sizeof e);
  e.index = index;
  e.opid = opid;
  e.val = val;
  if (insert) { e.action = E_INSERT; e.elem = elem; }
  else e.action = E_UPDATE;
  append_edit(p, &e);
}

typedef struct cstate {
  opid_t opid;
  int64_t value;
  uint32_t outstanding;
} cstate_t;

typedef struct {
  opid_t succ;
  cstate_t *st;
} cmap_t;

typedef struct {
  int action; /* 0 none, 1 insert, 2 update, 3 remove */
  cmap_t *cmap;
  uint32_t n_cmap, cap_cmap;
  int touched; /* propState[elemId] exists */
} pstate_t;

typedef struct {
  amo_doc *d;
  tab_t patches; /* idkey -> pobj_t* */
  pobj_t *root;
  pool_t pool;
  err_t *e;
} pctx_t;

static pobj_t *get_patch(pctx_t *c, opid_t id, int type) {
  if (id.ctr == 0) return c->root;
  void **s = tab_slot(&c->patches, idkey(id), 1);
  if (!*s) {
    pobj_t *p = (pobj_t *)pool_alloc(&c->pool, sizeof *p);
    p->id = id;
    p->type = type;
    *s = p;
  }
  return (pobj_t *)*s;
}

static int decode_int_value(const pval_t *v, int64_t *out, err_t *e) {
  dec_t d = {v->bytes, (size_t)(v->tag_len >> 4), 0};
  if ((v->tag_len & 15) == 3) {
    uint64_t u;
    if (read_u53(&d, &u, e)) return -1;
    *out = (int64_t)u;
    return 0;
  }
  return read_i53(&d, out, e);
}

/* one call of updatePatchProperty(patches, null, objectId, op, docState, propState, listIndex, succNum) */
static int update_patch_property(pctx_t *c, obj_t *o, pobj_t *patch, const slot_t *slot, opid_t elem_id, row_t *r,
                                 pstate_t *ps, uint64_t list_index) {
  int overwritten = r->n_succ > 0;
  int have_val = 0;
  opid_t patch_key = r->id;
  pval_t pv;
  memset(&pv, 0, sizeof pv);
  ps->touched = 1;
  if (overwritten && r->action == 1 && (r->val_tag_len & 15) == 8) {
    /* counter `set` with successors: open a counter state (new.js:937-951) */
    cstate_t *st = (cstate_t *)pool_alloc(&c->pool, sizeof *st);
    pval_t tmp = {0, r->val_tag_len, r->val, 0, NULL};
    st->opid = r->id;
    if (decode_int_value(&tmp, &st->value, c->e)) return -1;
    st->outstanding = r->n_succ;
    for (uint32_t i = 0; i < r->n_succ; i++) {
      /* counterStates[succOp] = counterState : later assignment wins */
      uint32_t k = 0;
      while (k < ps->n_cmap && !same_id(ps->cmap[k].succ, r->succ[i])) k++;
      if (k == ps->n_cmap) {
        if (ps->n_cmap == ps->cap_cmap) {
          ps->cap_cmap = ps->cap_cmap ? ps->cap_cmap * 2 : 4;
          ps->cmap = (cmap_t *)realloc(ps->cmap, sizeof(cmap_t) * ps->cap_cmap);
        }
        ps->n_cmap++;
        ps->cmap[k].succ = r->succ[i];
      }
      ps->cmap[k].st = st;
    }
  } else if (r->action == 5) {
    /* inc (new.js:953-967) */
    uint32_t k = 0;
    while (k < ps->n_cmap && !same_id(ps->cmap[k].succ, r->id)) k++;
    if (k == ps->n_cmap) {
      char b[160];
      fmt_opid(c->d, r->id, b, sizeof b);
      return fail(c->e, "increment operation %s for unknown counter", b);
    }
    cstate_t *st = ps->cmap[k].st;
    pval_t tmp = {0, r->val_tag_len, r->val, 0, NULL};
    int64_t inc = 0;
    uint64_t tag = r->val_tag_len & 15;
    if (tag == 3 || tag == 4 || tag == 8 || tag == 9) { if (decode_int_value(&tmp, &inc, c->e)) return -1; }
    else return fail(c->e, "unsupported: non-integer increment");
    st->value += inc;
    /* delete counterState.succs[opId]: only has an effect if opId is (still) in that set */
    if (st->outstanding > 0) {
      /* membership: the inc id is in st's succ set iff the map entry was created from st's own succ list;
       * entries overridden by a later counter point at the later state, so reaching `st` through the map
       * means the id is a successor of st */
      st->outstanding--;
      ps->cmap[k].succ.ctr = 0; /* consumed: a second inc row with the same id cannot exist */
      ps->cmap[k].succ.actor = NUL32;
    }
    if (st->outstanding == 0) {
      have_val = 1;
      patch_key = st->opid;
      pv.kind = 1;
      pv.counter = st->value;
    }
  } else if (!overwritten) {
    if (r->action == 1) {
      have_val = 1;
      pv.kind = 0;
      pv.tag_len = r->val_tag_len;
      pv.bytes = r->val;
    } else if ((r->action & 1) == 0) {
      have_val = 1;
      pv.kind = 2;
      pv.obj = get_patch(c, r->id, (int)r->action);
    }
  }

  if (slot == NULL) {
    /* list / text element (new.js:983-1033, whole-document branches only) */
    if (have_val) {
      pedit_t e;
      memset(&e, 0, sizeof e);
      if (ps->action == 0) {
        ps->action = 1;
        e.action = E_INSERT; e.index = list_index; e.elem = elem_id; e.opid = patch_key; e.val = pv;
        append_edit(patch, &e);
      } else if (ps->action == 3) {
        if (patch->n_edits == 0 || patch->edits[patch->n_edits - 1].action != E_REMOVE) return fail(c->e, "last edit has unexpected type");
        pedit_t *last = &patch->edits[patch->n_edits - 1];
        if (last->count > 1) last->count--; else patch->n_edits--;
        ps->action = 2;
        append_update(patch, list_index, elem_id, patch_key, pv, 1);
      } else {
        append_update(patch, list_index, elem_id, patch_key, pv, 0);
      }
    } else if (r->n_succ == 0 && ps->action == 0) {
      pedit_t e;
      memset(&e, 0, sizeof e);
      ps->action = 3;
      e.action = E_REMOVE; e.index = list_index; e.count = 1;
      append_edit(patch, &e);
    }
  } else if (have_val) {
    /* map / table (new.js:1035-1039): props[key][opId] = value, insertion-ordered */
    pprop_t *pp = NULL;
    if (patch->n_props > 0) {
      pprop_t *lastp = &patch->props[patch->n_props - 1];
      if (lastp->key_len == slot->key_len && memcmp(lastp->key, slot->key, slot->key_len) == 0) pp = lastp;
    }
    if (!pp) {
      if (patch->n_props == patch->cap_props) {
        patch->cap_props = patch->cap_props ? patch->cap_props * 2 : 8;
        patch->props = (pprop_t *)realloc(patch->props, patch->cap_props * sizeof(pprop_t));
      }
      pp = &patch->props[patch->n_props++];
      memset(pp, 0, sizeof *pp);
      pp->key = slot->key;
      pp->key_len = slot->key_len;
    }
    /* assigning to an existing opId key keeps its position (JS object semantics) */
    uint32_t k = 0;
    while (k < pp->n && !same_id(pp->ents[k].opid, patch_key)) k++;
    if (k == pp->n) {
      if (pp->n == pp->cap) {
        pp->cap = pp->cap ? pp->cap * 2 : 2;
        pp->ents = (pent_t *)realloc(pp->ents, pp->cap * sizeof(pent_t));
      }
      pp->n++;
      pp->ents[k].opid = patch_key;
    }
    pp->ents[k].val = pv;
  }
  (void)o;
  return 0;
}

/* ---- JSON.stringify-compatible rendering --------------------------------------------------------- */

static void json_string(sbuf_t *b, const uint8_t *p, size_t n) {
  sb_putc(b, '"');
  for (size_t i = 0; i < n; i++) {
    uint8_t ch = p[i];
    switch (ch) {
      case '"': sb_puts(b, "\\\""); break;
      case '\\': sb_puts(b, "\\\\"); break;
      case '\b': sb_puts(b, "\\b"); break;
      case '\f': sb_puts(b, "\\f"); break;
      case '\n': sb_puts(b, "\\n"); break;
      case '\r': sb_puts(b, "\\r"); break;
      case '\t': sb_puts(b, "\\t"); break;
      default:
        if (ch < 0x20) { char t[8]; snprintf(t, sizeof t, "\\u%04x", ch); sb_puts(b, t); }
        else sb_putc(b, (char)ch);
    }
  }
  sb_putc(b, '"');
}

static void json_opid(sbuf_t *b, const amo_doc *d, opid_t id) {
  char t[32];
  int n = snprintf(t, sizeof t, "\"%llu@", (unsigned long long)id.ctr);
  sb_put(b, t, n);
  span_t a = d->actors[id.actor];
  static const char hx[] = "0123456789abcdef";
  for (size_t i = 0; i < a.len; i++) { sb_putc(b, hx[a.p[i] >> 4]); sb_putc(b, hx[a.p[i] & 15]); }
  sb_putc(b, '"');
}

/* Number::toString for a double (ECMA-262 7.1.12.1): shortest digits that round-trip */
static void json_double(sbuf_t *b, double x) {
  if (isnan(x) || isinf(x)) { sb_puts(b, "null"); return; }
  if (x == 0) { sb_puts(b, "0"); return; }
  char digits[32];
  int prec, exp10 = 0;
  for (prec = 1; prec <= 17; prec++) {
    char t[40];
    snprintf(t, sizeof t, "%.*e", prec - 1, x);
    if (strtod(t, NULL) == x) {
      /* t = [-]d.ddd...e[+-]XX */
      const char *p = t;
      if (*p == '-') p++;
      int k = 0;
      for (; *p && *p != 'e'; p++) if (*p != '.') digits[k++] = *p;
      digits[k] = 0;
      exp10 = atoi(p + 1);
      break;
    }
  }
  int k = (int)strlen(digits);
  while (k > 1 && digits[k - 1] == '0') k--;
  digits[k] = 0;
  int n = exp10 + 1; /* decimal point position */
  if (x < 0) sb_putc(b, '-');
  if (k <= n && n <= 21) {
    sb_puts(b, digits);
    for (int i = k; i < n; i++) sb_putc(b, '0');
  } else if (0 < n && n <= 21) {
    sb_put(b, digits, n);
    sb_putc(b, '.');
    sb_puts(b, digits + n);
  } else if (-6 < n && n <= 0) {
    sb_puts(b, "0.");
    for (int i = n; i < 0; i++) sb_putc(b, '0');
    sb_puts(b, digits);
  } else {
    char t[16];
    sb_putc(b, digits[0]);
    if (k > 1) { sb_putc(b, '.'); sb_puts(b, digits + 1); }
    snprintf(t, sizeof t, "e%c%d", n - 1 >= 0 ? '+' : '-', abs(n - 1));
    sb_puts(b, t);
  }
}

static int json_pobj(pctx_t *c, sbuf_t *b, pobj_t *p);

/* the JS value of a decoded op value (columnar.js:300-329), rendered by JSON.stringify */
static int json_prim_value(pctx_t *c, sbuf_t *b, const pval_t *v) {
  uint64_t t = v->tag_len;
  size_t n = (size_t)(t >> 4);
  char tmp[40];
  if (v->kind == 1) { snprintf(tmp, sizeof tmp, "%lld", (long long)v->counter); sb_puts(b, tmp); return 0; }
  if (t == 0) { sb_puts(b, "null"); return 0; }
  if (t == 1) { sb_puts(b, "false"); return 0; }
  if (t == 2) { sb_puts(b, "true"); return 0; }
  switch (t & 15) {
    case 6:
      if (!utf8_valid(v->bytes, n)) return fail(c->e, "unsupported: malformed UTF-8 in string value");
      /* utf8ToString = TextDecoder('utf-8').decode (encoding.js:9-17): a leading U+FEFF is dropped */
      if (n >= 3 && v->bytes[0] == 0xef && v->bytes[1] == 0xbb && v->bytes[2] == 0xbf) json_string(b, v->bytes + 3, n - 3);
      else json_string(b, v->bytes, n);
      return 0;
    case 3: case 4: case 8: case 9: {
      int64_t x;
      if (decode_int_value(v, &x, c->e)) return -1;
      snprintf(tmp, sizeof tmp, "%lld", (long long)x);
      sb_puts(b, tmp);
      return 0;
    }
    case 5: {
      if (n != 8) return fail(c->e, "Invalid length for floating point number: %zu", n);
      double x;
      memcpy(&x, v->bytes, 8);
      json_double(b, x);
      return 0;
    }
    default: /* Uint8Array -> {"0":b0,"1":b1,...} */
      sb_putc(b, '{');
      for (size_t i = 0; i < n; i++) {
        snprintf(tmp, sizeof tmp, "%s\"%zu\":%u", i ? "," : "", i, v->bytes[i]);
        sb_puts(b, tmp);
      }
      sb_putc(b, '}');
      return 0;
  }
}

static const char *datatype_name(int dt) {
  switch (dt) {
    case 3: return "\"uint\"";
    case 4: return "\"int\"";
    case 5: return "\"float64\"";
    case 8: return "\"counter\"";
    case 9: return "\"timestamp\"";
    default: return NULL;
  }
}

static void json_datatype(sbuf_t *b, int dt) {
  const char *nm = datatype_name(dt);
  if (nm) sb_puts(b, nm);
  else { char t[8]; snprintf(t, sizeof t, "%d", dt); sb_puts(b, t); }
}

/* has a `datatype` property?  decodeValue sets it for every non-null/bool/string value; Object.assign
 * copies it even when it is the number 0 (tag 0 with a length), matching `sizeTag % 16`. */
static int has_datatype(const pval_t *v) {
  uint64_t t = v->tag_len;
  if (v->kind == 1) return 1;
  if (t == 0 || t == 1 || t == 2) return 0;
  return (t & 15) != 6;
}

static int json_value(pctx_t *c, sbuf_t *b, const pval_t *v) {
  if (v->kind == 2) return json_pobj(c, b, v->obj);
  if (v->kind == 1) {
    /* {type:'value', datatype:'counter', value} (new.js:963) */
    char t[64];
    snprintf(t, sizeof t, "{\"type\":\"value\",\"datatype\":\"counter\",\"value\":%lld}", (long long)v->counter);
    sb_puts(b, t);
    return 0;
  }
  sb_puts(b, "{\"type\":\"value\",\"value\":");
  if (json_prim_value(c, b, v)) return -1;
  if (has_datatype(v)) {
    sb_puts(b, ",\"datatype\":");
    json_datatype(b, (int)(v->tag_len & 15));
  }
  sb_putc(b, '}');
  return 0;
}

/* JS own-property order: canonical array-index keys (0..2^32-2) ascending first, then insertion order */
static int array_index_key(const uint8_t *k, size_t n, uint64_t *out) {
  if (n == 0 || n > 10) return 0;
  if (n > 1 && k[0] == '0') return 0;
  uint64_t v = 0;
  for (size_t i = 0; i < n; i++) {
    if (k[i] < '0' || k[i] > '9') return 0;
    v = v * 10 + (k[i] - '0');
  }
  if (v > 4294967294ull) return 0;
  *out = v;
  return 1;
}

typedef struct {
  uint64_t num;
  uint64_t pos;
} idxkey_t;
static int cmp_idxkey(const void *a, const void *b) {
  const idxkey_t *x = (const idxkey_t *)a, *y = (const idxkey_t *)b;
  return x->num < y->num ? -1 : x->num > y->num;
}

static int json_prop(pctx_t *c, sbuf_t *b, pprop_t *pp) {
  if (!utf8_valid(pp->key, pp->key_len)) return fail(c->e, "unsupported: malformed UTF-8 in key");
  /* a key that starts with U+FEFF loses it in utf8ToString and then collides with / repeats other keys (the reference's own
   * RLE decoder throws on the repeated literal): left to the JS path */
  if (pp->key_len >= 3 && pp->key[0] == 0xef && pp->key[1] == 0xbb && pp->key[2] == 0xbf) return fail(c->e, "unsupported: map key starts with a byte order mark");
  json_string(b, pp->key, pp->key_len);
  sb_puts(b, ":{");
  for (uint32_t i = 0; i < pp->n; i++) {
    if (i) sb_putc(b, ',');
    json_opid(b, c->d, pp->ents[i].opid);
    sb_putc(b, ':');
    if (json_value(c, b, &pp->ents[i].val)) return -1;
  }
  sb_putc(b, '}');
  return 0;
}

static const char *type_name(int type, int is_root) {
  if (is_root) return "\"map\"";
  switch (type) {
    case 0: return "\"map\"";
    case 2: return "\"list\"";
    case 4: return "\"text\"";
    case 6: return "\"table\"";
    default: return "null";
  }
}

static int json_pobj(pctx_t *c, sbuf_t *b, pobj_t *p) {
  sb_puts(b, "{\"objectId\":");
  if (p->is_root) sb_puts(b, "\"_root\"");
  else json_opid(b, c->d, p->id);
  sb_puts(b, ",\"type\":");
  sb_puts(b, type_name(p->type, p->is_root));
  if (!p->is_root && is_list_type(p->type)) {
    sb_puts(b, ",\"edits\":[");
    char t[64];
    for (uint64_t i = 0; i < p->n_edits; i++) {
      pedit_t *e = &p->edits[i];
      if (i) sb_putc(b, ',');
      switch (e->action) {
        case E_INSERT:
          snprintf(t, sizeof t, "{\"action\":\"insert\",\"index\":%llu,\"elemId\":", (unsigned long long)e->index);
          sb_puts(b, t);
          json_opid(b, c->d, e->elem);
          sb_puts(b, ",\"opId\":");
          json_opid(b, c->d, e->opid);
          sb_puts(b, ",\"value\":");
          if (json_value(c, b, &e->val)) return -1;
          sb_putc(b, '}');
          break;
        case E_MULTI: {
          snprintf(t, sizeof t, "{\"action\":\"multi-insert\",\"index\":%llu,\"elemId\":", (unsigned long long)e->index);
          sb_puts(b, t);
          json_opid(b, c->d, e->elem);
          /* `if (nextEdit.value.datatype) lastEdit.datatype = ...` -- truthy datatypes only (new.js:762) */
          pval_t second = e->vals[e->nvals > 1 ? 1 : 0];
          /* (a counter total -- kind 1 -- has no type/length word of its own: its datatype is 'counter') */
          if (has_datatype(&second) && (second.kind == 1 || (second.tag_len & 15) != 0)) {
            sb_puts(b, ",\"datatype\":");
            json_datatype(b, second.kind == 1 ? 8 : (int)(second.tag_len & 15));
          }
          sb_puts(b, ",\"values\":[");
          for (uint64_t k = 0; k < e->nvals; k++) {
            if (k) sb_putc(b, ',');
            if (json_prim_value(c, b, &e->vals[k])) return -1;
          }
          sb_puts(b, "]}");
          break;
        }
        case E_UPDATE:
          snprintf(t, sizeof t, "{\"action\":\"update\",\"index\":%llu,\"opId\":", (unsigned long long)e->index);
          sb_puts(b, t);
          json_opid(b, c->d, e->opid);
          sb_puts(b, ",\"value\":");
          if (json_value(c, b, &e->val)) return -1;
          sb_putc(b, '}');
          break;
        default:
          snprintf(t, sizeof t, "{\"action\":\"remove\",\"index\":%llu,\"count\":%llu}", (unsigned long long)e->index,
                   (unsigned long long)e->count);
          sb_puts(b, t);
      }
    }
    sb_puts(b, "]}");
  } else {
    sb_puts(b, ",\"props\":{");
    /* integer-like keys first, numerically; then the rest in insertion (= ascending key) order */
    idxkey_t *idx = (idxkey_t *)malloc(sizeof(idxkey_t) * (p->n_props ? p->n_props : 1));
    uint64_t ni = 0;
    for (uint64_t i = 0; i < p->n_props; i++) {
      uint64_t v;
      if (array_index_key(p->props[i].key, p->props[i].key_len, &v)) { idx[ni].num = v; idx[ni].pos = i; ni++; }
    }
    qsort(idx, ni, sizeof(idxkey_t), cmp_idxkey);
    int first = 1, rc = 0;
    for (uint64_t i = 0; i < ni && !rc; i++) {
      if (!first) sb_putc(b, ',');
      first = 0;
      rc = json_prop(c, b, &p->props[idx[i].pos]);
    }
    for (uint64_t i = 0; i < p->n_props && !rc; i++) {
      uint64_t v;
      if (array_index_key(p->props[i].key, p->props[i].key_len, &v)) continue;
      if (!first) sb_putc(b, ',');
      first = 0;
      rc = json_prop(c, b, &p->props[i]);
    }
    free(idx);
    if (rc) return -1;
    sb_puts(b, "}}");
  }
  return 0;
}

/* {"maxOp":..,"clock":{..},"deps":[..],"pendingChanges":n,"diffs":  -- new.js:1870-1873, 2064-2067 */
static void json_envelope(amo_doc *d, sbuf_t *b) {
  char t[64];
  /* envelope key order: maxOp, clock, deps, pendingChanges, diffs (new.js:2064-2067) */
  snprintf(t, sizeof t, "{\"maxOp\":%llu,\"clock\":{", (unsigned long long)d->max_op);
  sb_puts(b, t);
  {
    /* clock keys are hex actor ids in first-applied order, except integer-like keys go first */
    static const char hx[] = "0123456789abcdef";
    uint32_t nc = d->clock_order ? d->n_clock : d->n_actors;
    idxkey_t *idx = (idxkey_t *)malloc(sizeof(idxkey_t) * (nc ? nc : 1));
    char **hex = (char **)malloc(sizeof(char *) * (nc ? nc : 1));
    uint64_t ni = 0;
    for (uint32_t k = 0; k < nc; k++) {
      uint32_t i = d->clock_order ? d->clock_order[k] : k;
      hex[k] = (char *)malloc(d->actors[i].len * 2 + 1);
      for (size_t j = 0; j < d->actors[i].len; j++) { hex[k][2 * j] = hx[d->actors[i].p[j] >> 4]; hex[k][2 * j + 1] = hx[d->actors[i].p[j] & 15]; }
      hex[k][d->actors[i].len * 2] = 0;
      uint64_t v;
      if (array_index_key((const uint8_t *)hex[k], d->actors[i].len * 2, &v)) { idx[ni].num = v; idx[ni].pos = k; ni++; }
    }
    qsort(idx, ni, sizeof(idxkey_t), cmp_idxkey);
    int first = 1;
    for (uint64_t q = 0; q < ni; q++) {
      uint32_t k = (uint32_t)idx[q].pos, i = d->clock_order ? d->clock_order[k] : k;
      snprintf(t, sizeof t, "\":%llu", (unsigned long long)d->clock[i]);
      if (!first) sb_putc(b, ',');
      first = 0;
      sb_putc(b, '"'); sb_puts(b, hex[k]); sb_puts(b, t);
    }
    for (uint32_t k = 0; k < nc; k++) {
      uint32_t i = d->clock_order ? d->clock_order[k] : k;
      uint64_t v;
      if (array_index_key((const uint8_t *)hex[k], d->actors[i].len * 2, &v)) continue;
      snprintf(t, sizeof t, "\":%llu", (unsigned long long)d->clock[i]);
      if (!first) sb_putc(b, ',');
      first = 0;
      sb_putc(b, '"'); sb_puts(b, hex[k]); sb_puts(b, t);
    }
    for (uint32_t k = 0; k < nc; k++) free(hex[k]);
    free(hex);
    free(idx);
  }
  sb_puts(b, "},\"deps\":[");
  for (uint32_t i = 0; i < d->n_heads; i++) {
    static const char hx[] = "0123456789abcdef";
    if (i) sb_putc(b, ',');
    sb_putc(b, '"');
    for (int k = 0; k < 32; k++) { sb_putc(b, hx[d->heads[32 * i + k] >> 4]); sb_putc(b, hx[d->heads[32 * i + k] & 15]); }
    sb_putc(b, '"');
  }
  snprintf(t, sizeof t, "],\"pendingChanges\":%u,\"diffs\":", d->n_pending);
  sb_puts(b, t);
}

const char *amo_patch_json(amo_doc *d, size_t *len, char *errbuf, size_t errcap) {
  if (d->json_done) { if (len) *len = d->json.len; return d->json.p; }
  err_t e = {{0}, 0};
  pctx_t c;
  memset(&c, 0, sizeof c);
  c.d = d;
  c.e = &e;
  pobj_t root;
  memset(&root, 0, sizeof root);
  root.is_root = 1;
  c.root = &root;
  int rc = 0;

  /* documentPatch (new.js:1604-1635): one pass over all rows in canonical order */
  obj_t **objs = sorted_objs(d);
  for (uint64_t oi = 0; oi <= d->n_objs && !rc; oi++) {
    obj_t *o = objs[oi];
    pobj_t *patch = get_patch(&c, o->id, o->type);
    if (is_list_type(o->type)) {
      uint64_t list_index = 0;
      int elem_visible = 0;
      for (elem_t *el = o->head.next; el && !rc; el = el->next) {
        pstate_t ps;
        memset(&ps, 0, sizeof ps);
        for (row_t *r = el->rows; r && !rc; r = r->next) {
          if (r->insert && elem_visible) { elem_visible = 0; list_index++; }
          if (r->n_succ == 0) elem_visible = 1;
          rc = update_patch_property(&c, o, patch, NULL, el->rows->id, r, &ps, list_index);
        }
        free(ps.cmap);
      }
    } else {
      slot_t **sl = sorted_slots(o);
      for (uint32_t i = 0; i < o->n_slots && !rc; i++) {
        pstate_t ps;
        memset(&ps, 0, sizeof ps);
        opid_t none = {0, 0};
        for (row_t *r = sl[i]->rows; r && !rc; r = r->next) rc = update_patch_property(&c, o, patch, sl[i], none, r, &ps, 0);
        free(ps.cmap);
      }
      free(sl);
    }
  }
  free(objs);

  if (!rc) {
    sbuf_t *b = &d->json;
    json_envelope(d, b);
    rc = json_pobj(&c, b, &root);
    if (!rc) sb_putc(b, '}');
  }

  /* release patch scaffolding */
  for (uint64_t i = 0; i < c.patches.cap; i++)
    if (c.patches.vals && c.patches.vals[i]) {
      pobj_t *p = (pobj_t *)c.patches.vals[i];
      for (uint64_t k = 0; k < p->n_props; k++) free(p->props[k].ents);
      for (uint64_t k = 0; k < p->n_edits; k++) if (p->edits[k].action == E_MULTI) free(p->edits[k].vals);
      free(p->props);
      free(p->edits);
    }
  for (uint64_t k = 0; k < root.n_props; k++) free(root.props[k].ents);
  free(root.props);
  tab_free(&c.patches);
  pool_free(&c.pool);

  if (rc) {
    if (errbuf && errcap) snprintf(errbuf, errcap, "%s", e.msg);
    d->json.len = 0;
    return NULL;
  }
  d->json_done = 1;
  if (len) *len = d->json.len;
  return d->json.p;
}

/* ===================================================================================================
 * Document load.  columnar.js:1006-1038 decodeDocumentHeader, 1062-1067 inflateColumn; new.js:1645-1675
 * readDocumentChanges, 1695-1750 BackendDoc constructor (whole-document patch = documentPatch over the stored
 * rows, which a saved document holds in canonical order with their succ lists: columnar.js:892, new.js:2047).
 * =================================================================================================*/

static int inflate_raw(const uint8_t *in, size_t inlen, uint8_t **out, size_t *outlen, err_t *e) {
  size_t cap = inlen * 4 + 1024;
  for (;;) {
    uint8_t *buf = (uint8_t *)malloc(cap);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) { free(buf); return fail(e, "inflate init failed"); }
    zs.next_in = (Bytef *)in; zs.avail_in = (uInt)inlen;
    zs.next_out = buf; zs.avail_out = (uInt)cap;
    int rc = inflate(&zs, Z_FINISH);
    size_t got = zs.total_out;
    inflateEnd(&zs);
    if (rc == Z_STREAM_END) { *out = buf; *outlen = got; return 0; }
    free(buf);
    if (rc == Z_BUF_ERROR || rc == Z_OK) { cap *= 4; continue; }
    return fail(e, "invalid deflate data");
  }
}

typedef struct {
  uint64_t id;
  span_t data;
} dcol_t;

static span_t find_col(const dcol_t *cols, uint64_t n, uint64_t id) {
  span_t none = {NULL, 0};
  for (uint64_t i = 0; i < n; i++)
    if (cols[i].id == id) return cols[i].data;
  return none;
}

/* read a column directory and return the columns (inflated into pool memory) */
static int read_doc_columns(pool_t *pool, dec_t *h, dcol_t **out, uint64_t *n_out, uint64_t **lens_out, err_t *e) {
  uint64_t n;
  if (read_u53(h, &n, e)) return -1;
  if (n > h->len) return fail(e, "subarray exceeds buffer size");
  dcol_t *cols = (dcol_t *)pool_alloc(pool, sizeof(dcol_t) * (n + 1));
  uint64_t *lens = (uint64_t *)pool_alloc(pool, sizeof(uint64_t) * (n + 1));
  int64_t last = -1;
  for (uint64_t i = 0; i < n; i++) {
    if (read_u53(h, &cols[i].id, e) || read_u53(h, &lens[i], e)) return -1;
    int64_t masked = (int64_t)(cols[i].id & ~(uint64_t)8);
    if (masked <= last) return fail(e, "Columns must be in ascending order");
    last = masked;
  }
  *out = cols;
  *n_out = n;
  *lens_out = lens;
  return 0;
}

static int load_doc_column_data(pool_t *pool, dec_t *h, dcol_t *cols, uint64_t n, const uint64_t *lens, err_t *e) {
  for (uint64_t i = 0; i < n; i++) {
    const uint8_t *p;
    if (read_bytes(h, lens[i], &p, e)) return -1;
    if (cols[i].id & 8) {
      uint8_t *buf;
      size_t blen;
      if (inflate_raw(p, lens[i], &buf, &blen, e)) return -1;
      uint8_t *keep = (uint8_t *)pool_alloc(pool, blen ? blen : 1);
      memcpy(keep, buf, blen);
      free(buf);
      cols[i].data.p = keep;
      cols[i].data.len = blen;
      cols[i].id ^= 8;
    } else {
      cols[i].data.p = p;
      cols[i].data.len = lens[i];
    }
  }
  return 0;
}

amo_doc *amo_load_document(const uint8_t *buf, size_t len, char *errbuf, size_t errcap) {
  err_t e = {{0}, 0};
  amo_doc *d = (amo_doc *)calloc(1, sizeof *d);
  d->hashes = (uint8_t *)calloc(1, 32);
  int rc = 0;
  uint8_t *copy = (uint8_t *)pool_alloc(&d->pool, len ? len : 1); /* rows keep pointers into the document */
  memcpy(copy, buf, len);
  buf = copy;
  do {
    dec_t c = {buf, len, 0};
    const uint8_t *magic, *sum, *typ, *body;
    uint64_t clen;
    if ((rc = read_bytes(&c, 4, &magic, &e))) break;
    if (memcmp(magic, MAGIC, 4) != 0) { rc = fail(&e, "Data does not begin with magic bytes 85 6f 4a 83"); break; }
    if ((rc = read_bytes(&c, 4, &sum, &e))) break;
    size_t hash_start = c.off;
    if ((rc = read_bytes(&c, 1, &typ, &e)) || (rc = read_u53(&c, &clen, &e)) || (rc = read_bytes(&c, clen, &body, &e))) break;
    uint8_t hash[32];
    amo_sha256(buf + hash_start, c.off - hash_start, hash);
    if (memcmp(hash, sum, 4) != 0) { rc = fail(&e, "checksum does not match data"); break; }
    if (c.off != len) { rc = fail(&e, "Encoded document has trailing data"); break; }
    if (*typ != 0) { rc = fail(&e, "Unexpected chunk type: %d", *typ); break; }

    dec_t h = {body, clen, 0};
    uint64_t na, nh;
    if ((rc = read_u53(&h, &na, &e))) break;
    if (na > clen || na >= MAX_ACTORS) { rc = fail(&e, "unsupported: too many actors"); break; }
    d->n_actors = d->cap_actors = (uint32_t)na;
    d->actors = (span_t *)calloc(na ? na : 1, sizeof(span_t));
    d->clock = (uint64_t *)calloc(na ? na : 1, 8);
    d->clock_order = (uint32_t *)calloc(na ? na : 1, 4);
    for (uint64_t i = 0; i < na && !rc; i++) {
      uint64_t l;
      if ((rc = read_u53(&h, &l, &e)) || (rc = read_bytes(&h, l, &d->actors[i].p, &e))) break;
      d->actors[i].len = l;
    }
    if (rc) break;
    if ((rc = read_u53(&h, &nh, &e))) break;
    const uint8_t *heads;
    if ((rc = read_bytes(&h, nh * 32, &heads, &e))) break;
    d->n_heads = (uint32_t)nh;
    d->heads = (uint8_t *)pool_alloc(&d->pool, nh ? nh * 32 : 1);
    memcpy(d->heads, heads, nh * 32);
    dcol_t *ccols, *ocols;
    uint64_t ncc, noc, *clens, *olens;
    if ((rc = read_doc_columns(&d->pool, &h, &ccols, &ncc, &clens, &e)) || (rc = read_doc_columns(&d->pool, &h, &ocols, &noc, &olens, &e))) break;
    if ((rc = load_doc_column_data(&d->pool, &h, ccols, ncc, clens, &e)) || (rc = load_doc_column_data(&d->pool, &h, ocols, noc, olens, &e))) break;
    /* headsIndexes and extraBytes follow (columnar.js:1032-1036); neither influences the patch, but the head indexes are read */
    if (h.off < h.len) {
      for (uint64_t i = 0; i < nh && !rc; i++) { uint64_t ix; rc = read_u53(&h, &ix, &e); }
      if (rc) break;
    }

    /* ---- readDocumentChanges (new.js:1645-1675): clock in first-appearance order, seq continuity ---- */
    {
      span_t ca = find_col(ccols, ncc, 0x01), cs = find_col(ccols, ncc, 0x03);
      rle_t actorD;
      delta_t seqD;
      rle_init(&actorD, 0, ca.p, ca.len);
      delta_init(&seqD, cs.p, cs.len);
      while (!rle_done(&actorD) && !rc) {
        rval_t a, q;
        if ((rc = rle_read(&actorD, &a, &e)) || (rc = delta_read(&seqD, &q, &e))) break;
        if (a.is_null || (uint64_t)a.i >= na) { rc = fail(&e, "unsupported: bad actor index in change metadata"); break; }
        uint64_t seq = q.is_null ? 0 : (uint64_t)q.i;
        if (seq != 1 && seq != d->clock[a.i] + 1) { rc = fail(&e, "Expected seq %llu, got %llu", (unsigned long long)d->clock[a.i] + 1, (unsigned long long)seq); break; }
        int seen = 0;
        for (uint32_t k = 0; k < d->n_clock; k++) seen |= d->clock_order[k] == (uint32_t)a.i;
        if (!seen) d->clock_order[d->n_clock++] = (uint32_t)a.i;
        d->clock[a.i] = seq;
        d->n_changes++;
      }
      if (rc) break;
      d->n_applied = d->n_changes;
    }

    /* ---- op rows, in file order ---- */
    rle_t objA, objC, keyA, keyS, idA, act, vlen, snum, sact;
    delta_t keyC, idC, sctr;
    bool_t ins;
#define COL(id) find_col(ocols, noc, id)
    span_t t;
    t = COL(0x01); rle_init(&objA, 0, t.p, t.len);
    t = COL(0x02); rle_init(&objC, 0, t.p, t.len);
    t = COL(0x11); rle_init(&keyA, 0, t.p, t.len);
    t = COL(0x13); delta_init(&keyC, t.p, t.len);
    t = COL(0x15); rle_init(&keyS, 2, t.p, t.len);
    t = COL(0x21); rle_init(&idA, 0, t.p, t.len);
    t = COL(0x23); delta_init(&idC, t.p, t.len);
    t = COL(0x34); bool_init(&ins, t.p, t.len);
    t = COL(0x42); rle_init(&act, 0, t.p, t.len);
    t = COL(0x56); rle_init(&vlen, 0, t.p, t.len);
    t = COL(0x80); rle_init(&snum, 0, t.p, t.len);
    t = COL(0x81); rle_init(&sact, 0, t.p, t.len);
    t = COL(0x83); delta_init(&sctr, t.p, t.len);
    span_t rawc = COL(0x57);
#undef COL
    dec_t raw = {rawc.p, rawc.len, 0};
    obj_t *cur_obj = NULL;
    opid_t cur_objid = {0, 0};
    int have_obj = 0;
    elem_t *tail = NULL;     /* last element of the current list object */
    slot_t *cur_slot = NULL;
    row_t *last_row = NULL;  /* last row of the current slot / element */
    while (!rle_done(&act) && !rc) {
      rval_t vOA, vOC, vKA, vKC, vKS, vIA, vIC, vAct, vLen, vSN;
      int b;
      if ((rc = rle_read(&objA, &vOA, &e)) || (rc = rle_read(&objC, &vOC, &e)) || (rc = rle_read(&keyA, &vKA, &e)) ||
          (rc = delta_read(&keyC, &vKC, &e)) || (rc = rle_read(&keyS, &vKS, &e)) || (rc = rle_read(&idA, &vIA, &e)) ||
          (rc = delta_read(&idC, &vIC, &e)) || (rc = bool_read(&ins, &b, &e)) || (rc = rle_read(&act, &vAct, &e)) ||
          (rc = rle_read(&vlen, &vLen, &e)) || (rc = rle_read(&snum, &vSN, &e)))
        break;
      if (vIA.is_null || vIC.is_null || vAct.is_null || (uint64_t)vIA.i >= na || vIC.i <= 0) { rc = fail(&e, "unsupported: document row without a valid id"); break; }
      if (vOA.is_null != vOC.is_null || (!vOA.is_null && (uint64_t)vOA.i >= na)) { rc = fail(&e, "unsupported: bad object reference in document"); break; }
      opid_t id = {(uint64_t)vIC.i, (uint32_t)vIA.i};
      opid_t objid = {0, 0};
      if (!vOC.is_null) { objid.ctr = (uint64_t)vOC.i; objid.actor = (uint32_t)vOA.i; }
      if (id.ctr >= ((uint64_t)1 << 44)) { rc = fail(&e, "unsupported: op counter too large"); break; }
      uint64_t tl = vLen.is_null ? 0 : (uint64_t)vLen.i;
      const uint8_t *val;
      if ((rc = read_bytes(&raw, tl >> 4, &val, &e))) break;
      if (!have_obj || objid.ctr != cur_objid.ctr || objid.actor != cur_objid.actor) {
        /* object order: _root, then ascending (ctr, actorId) */
        if (have_obj && !(cur_objid.ctr == 0 ? objid.ctr != 0 : cmp_opid(d, cur_objid, objid) < 0)) { rc = fail(&e, "unsupported: document objects not in canonical order"); break; }
        if (objid.ctr == 0) cur_obj = &d->root;
        else {
          void **sl = tab_slot(&d->objs, idkey(objid), 0);
          if (!sl) { rc = fail(&e, "unsupported: document row for an unknown object"); break; }
          cur_obj = (obj_t *)*sl;
        }
        cur_objid = objid;
        have_obj = 1;
        tail = &cur_obj->head;
        cur_slot = NULL;
        last_row = NULL;
      }
      row_t *r = (row_t *)pool_alloc(&d->pool, sizeof *r);
      r->id = id; r->insert = (uint8_t)b; r->action = (uint32_t)vAct.i; r->val_tag_len = tl; r->val = val;
      uint64_t ns = vSN.is_null ? 0 : (uint64_t)vSN.i;
      if (ns) {
        r->succ = (opid_t *)pool_alloc(&d->pool, sizeof(opid_t) * ns);
        r->cap_succ = r->n_succ = (uint32_t)ns;
        for (uint64_t k = 0; k < ns && !rc; k++) {
          rval_t sa, sc;
          if ((rc = rle_read(&sact, &sa, &e)) || (rc = delta_read(&sctr, &sc, &e))) break;
          if (sa.is_null || sc.is_null || (uint64_t)sa.i >= na) { rc = fail(&e, "unsupported: bad succ entry"); break; }
          r->succ[k].ctr = (uint64_t)sc.i;
          r->succ[k].actor = (uint32_t)sa.i;
          if ((uint64_t)sc.i > d->max_op) d->max_op = (uint64_t)sc.i; /* new.js:1628-1630 */
        }
        if (rc) break;
      }
      if (id.ctr > d->max_op) d->max_op = id.ctr; /* new.js:1627 */
      if (!vKS.is_null) {
        if (is_list_type(cur_obj->type) || b) { rc = fail(&e, "unsupported: string key used in a list object"); break; }
        if (!cur_slot || cur_slot->key_len != vKS.slen || memcmp(cur_slot->key, vKS.s, vKS.slen) != 0) {
          if (cur_slot && cmp_utf16(cur_slot->key, cur_slot->key_len, vKS.s, vKS.slen) >= 0) { rc = fail(&e, "unsupported: document keys not in canonical order"); break; }
          if (find_slot(d, cur_obj, vKS.s, (uint32_t)vKS.slen, 0)) { rc = fail(&e, "unsupported: document keys not in canonical order"); break; }
          cur_slot = find_slot(d, cur_obj, vKS.s, (uint32_t)vKS.slen, 1);
          last_row = NULL;
        }
        if (last_row && cmp_opid(d, last_row->id, id) >= 0) { rc = fail(&e, "unsupported: document ops not in ascending order"); break; }
        if (last_row) last_row->next = r; else cur_slot->rows = r;
        last_row = r;
      } else {
        if (!is_list_type(cur_obj->type)) { rc = fail(&e, "unsupported: list operation on a map object"); break; }
        if (b) {
          elem_t *el = (elem_t *)pool_alloc(&d->pool, sizeof *el);
          el->rows = r;
          tail->next = el;
          tail = el;
          cur_obj->n_elems++;
          index_elem(d, cur_objid, id, el);
          last_row = r;
        } else {
          if (vKC.is_null || vKA.is_null || vKC.i <= 0 || tail == &cur_obj->head || tail->rows->id.ctr != (uint64_t)vKC.i ||
              tail->rows->id.actor != (uint32_t)vKA.i) { rc = fail(&e, "unsupported: list update does not follow its element"); break; }
          if (cmp_opid(d, last_row->id, id) >= 0) { rc = fail(&e, "unsupported: document ops not in ascending order"); break; }
          last_row->next = r;
          last_row = r;
        }
      }
      d->n_rows++;
      d->n_ops++;
      if ((r->action & 1) == 0) {
        void **sl = tab_slot(&d->objs, idkey(id), 1);
        if (*sl) { rc = fail(&e, "duplicate operation ID in document"); break; }
        obj_t *no = (obj_t *)pool_alloc(&d->pool, sizeof *no);
        no->id = id;
        no->type = (int)r->action;
        *sl = no;
        if (d->n_objs == d->cap_objs) {
          d->cap_objs = d->cap_objs ? d->cap_objs * 2 : 64;
          d->obj_list = (obj_t **)realloc(d->obj_list, d->cap_objs * sizeof(obj_t *));
        }
        d->obj_list[d->n_objs++] = no;
      }
    }
  } while (0);
  if (rc) {
    if (errbuf && errcap) snprintf(errbuf, errcap, "%s", e.msg);
    amo_free(d);
    return NULL;
  }
  return d;
}

/* ===================================================================================================
 * Stand-alone change decode for kernel-level parity tests
 * =================================================================================================*/

amo_decoded_t *amo_decode_change(const uint8_t *change, size_t len, char *errbuf, size_t errcap) {
  err_t e = {{0}, 0};
  pool_t pool = {0};
  change_t c;
  dops_t ops;
  memset(&ops, 0, sizeof ops);
  amo_decoded_t *o = NULL;
  if (parse_change(&pool, change, len, &c, &e) || decode_ops(&c, &ops, &e)) {
    if (errbuf && errcap) snprintf(errbuf, errcap, "%s", e.msg);
    dops_free(&ops);
    pool_free(&pool);
    return NULL;
  }
  o = (amo_decoded_t *)calloc(1, sizeof *o);
  uint64_t n = ops.n_ops, np = ops.n_preds, a = n ? n : 1, ap = np ? np : 1;
  o->n_ops = n; o->n_preds = np; o->seq = c.seq; o->start_op = c.start_op; o->n_deps = c.n_deps; o->n_actors = c.n_actors;
  memcpy(o->hash, c.hash, 32);
  o->raw_len = c.raw_len;
  o->raw = (uint8_t *)malloc(c.raw_len ? c.raw_len : 1);
  memcpy(o->raw, c.raw, c.raw_len);
  o->obj_ctr = (uint64_t *)malloc(8 * a); o->key_ctr = (uint64_t *)malloc(8 * a); o->val_tag_len = (uint64_t *)malloc(8 * a);
  o->val_off = (uint64_t *)malloc(8 * a); o->obj_actor = (uint32_t *)malloc(4 * a); o->key_actor = (uint32_t *)malloc(4 * a);
  o->action = (uint32_t *)malloc(4 * a); o->pred_num = (uint32_t *)malloc(4 * a); o->key_off = (uint32_t *)malloc(4 * a);
  o->key_len = (uint32_t *)malloc(4 * a); o->insert = (uint8_t *)malloc(a); o->pred_ctr = (uint64_t *)malloc(8 * ap);
  o->pred_actor = (uint32_t *)malloc(4 * ap);
  for (uint64_t i = 0; i < n; i++) {
    dop_t *p = &ops.ops[i];
    o->obj_ctr[i] = p->obj_ctr; o->key_ctr[i] = p->key_ctr; o->val_tag_len[i] = p->val_tag_len;
    o->val_off[i] = (uint64_t)(p->val - c.raw); o->obj_actor[i] = p->obj_actor; o->key_actor[i] = p->key_actor;
    o->action[i] = p->action; o->pred_num[i] = p->pred_num; o->key_len[i] = p->key_len;
    o->key_off[i] = p->key_len == NUL32 ? 0 : (uint32_t)(p->key - c.raw); o->insert[i] = p->insert;
  }
  memcpy(o->pred_ctr, ops.pred_ctr, 8 * np);
  memcpy(o->pred_actor, ops.pred_actor, 4 * np);
  dops_free(&ops);
  pool_free(&pool);
  return o;
}

void amo_decoded_free(amo_decoded_t *o) {
  if (!o) return;
  free(o->obj_ctr); free(o->key_ctr); free(o->val_tag_len); free(o->val_off); free(o->obj_actor); free(o->key_actor);
  free(o->action); free(o->pred_num); free(o->key_off); free(o->key_len); free(o->insert); free(o->pred_ctr);
  free(o->pred_actor); free(o->raw);
  free(o);
}

#include "am_oracle_apply.c"
