/*
 * am_oracle_apply.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Included at the end of am_oracle.c (one translation unit).
 *
 * Backend.applyChanges(state, changes) with its INCREMENTAL patch: a sequential restatement of
 *   new.js:1797-1879  BackendDoc.applyChanges (queue, retry loop, envelope)
 *   new.js:1550-1597  applyChanges (one scheduling pass)
 *   new.js:1304-1380  applyOps            (only its control flow: seek, merge, repeat; the block store is not reproduced)
 *   new.js:1052-1290  mergeDocChangeOps   (the merge loop, line by line: which change ops travel together, which document
 *                                          ops are looked at, in what order updatePatchProperty sees them)
 *   new.js:884-1040   updatePatchProperty in its incremental mode (newBlock != null), objectMeta included
 *   new.js:747-869    appendEdit / appendUpdate / convertInsertToUpdate
 *   new.js:1461-1528  setupPatches
 * over the linked-row document of am_oracle.c.  A document op is visited through a cursor that walks one object in document
 * order (map keys ascending in UTF-16 order, list elements in RGA order, rows of a key / element ascending by op id); whatever
 * lies beyond the object is "not in the correct object" to the merge loop, which is all it ever asks about it.
 *
 * Pinned by tests/golden/ref_apply_vectors.json.gz: every applyChanges call the reference's own suites make, with the patch the
 * unmodified reference returned (oracle/make_apply_vectors.py).
 *
 * A session document is built ONLY by amo_init() + amo_apply_changes() (or amo_load_document() + amo_apply_changes()), call by
 * call as the reference was called, because objectMeta is history: it records what earlier calls saw.
 */

/* ---------------------------------------------------------------------------------------------------
 * objectMeta.children (new.js:894-931): per object, key / elemId -> insertion-ordered { opId: value }
 * -------------------------------------------------------------------------------------------------*/
typedef struct {
  opid_t opid;
  int is_obj, type;      /* child object reference {objectId, type} ... */
  uint64_t tag_len;      /* ... or {type:'value', value, datatype} */
  const uint8_t *bytes;
} chval_t;

typedef struct chkey {
  int is_elem;
  opid_t elem;
  const uint8_t *key;
  uint32_t key_len;
  chval_t *vals;
  uint32_t n;
  struct chkey *next;
} chkey_t;

typedef struct {  /* the key of a property: a string (maps, tables) or an element id (lists, texts) */
  int is_elem;
  opid_t elem;
  const uint8_t *key;
  uint32_t key_len;
} pkey_t;

static int pkey_eq(const pkey_t *a, int is_elem, opid_t elem, const uint8_t *key, uint32_t key_len) {
  if (a->is_elem != is_elem) return 0;
  if (is_elem) return same_id(a->elem, elem);
  return a->key_len == key_len && memcmp(a->key, key, key_len) == 0;
}

static chkey_t *children_get(obj_t *o, const pkey_t *k, int create, pool_t *pool) {
  for (chkey_t *c = o->children; c; c = c->next)
    if (pkey_eq(k, c->is_elem, c->elem, c->key, c->key_len)) return c;
  if (!create) return NULL;
  chkey_t *c = (chkey_t *)pool_alloc(pool, sizeof *c);
  c->is_elem = k->is_elem; c->elem = k->elem; c->key = k->key; c->key_len = k->key_len;
  c->next = o->children;
  o->children = c;
  return c;
}

/* ---------------------------------------------------------------------------------------------------
 * session state
 * -------------------------------------------------------------------------------------------------*/
typedef struct qchange {
  change_t c;
} qchange_t;

static void session_free(amo_doc *d) {
  for (uint64_t i = 0; i < d->n_objs; i++) free(d->obj_list[i]->sorted);
  free(d->root.sorted);
  tab_free(&d->known);
  free(d->queue);
  free(d->since);
  free(d->apply_json.p);
}

amo_doc *amo_init(void) {
  amo_doc *d = (amo_doc *)calloc(1, sizeof *d);
  d->hashes = (uint8_t *)calloc(1, 32);
  d->session = 1;
  d->have_graph = 1;
  d->meta_built = 1;
  d->root.has_meta = 1;
  d->heads = (uint8_t *)pool_alloc(&d->pool, 32);
  return d;
}

/* a change op with its actors translated to document actor indexes (new.js:598-600, 708-709) */
typedef struct {
  opid_t id, obj, key;   /* obj.ctr 0 = _root; key (element id, ctr 0 = _head) when key_str == NULL */
  const uint8_t *key_str;
  uint32_t key_len;
  uint8_t insert;
  uint32_t action;
  uint64_t val_tag_len;
  const uint8_t *val;
  uint32_t pred_num;
  opid_t *preds;
} cop_t;

/* propState[elemId] of one mergeDocChangeOps call (new.js:899-910, 937-966, 988-1032) */
typedef struct {
  pkey_t k;
  row_t **visible;
  uint32_t n_vis, cap_vis;
  int has_child;
  pstate_t ps;  /* action + counter states */
} iprop_t;

typedef struct {
  amo_doc *d;
  pctx_t pc;          /* patches */
  err_t *e;
  obj_t **touched;    /* objectIds, in insertion order */
  uint32_t n_touched, cap_touched;
  iprop_t *props;     /* propState of the running call */
  uint32_t n_props, cap_props;
} ictx_t;

static obj_t *obj_by_id(amo_doc *d, opid_t id) {
  if (id.ctr == 0) return &d->root;
  void **s = tab_slot(&d->objs, idkey(id), 0);
  return s ? (obj_t *)*s : NULL;
}

static pobj_t *patch_of(ictx_t *x, obj_t *o) {
  if (o->id.ctr == 0) return x->pc.root;
  return get_patch(&x->pc, o->id, o->type);
}

/* sorted key table of a map object */
static void ensure_sorted(obj_t *o) {
  if (o->sorted || o->n_slots == 0) return;
  o->sorted = sorted_slots(o);
  o->n_sorted = o->cap_sorted = o->n_slots;
}
static uint32_t sorted_lower_bound(obj_t *o, const uint8_t *key, uint32_t len) {
  uint32_t lo = 0, hi = o->n_sorted;
  while (lo < hi) {
    uint32_t mid = (lo + hi) / 2;
    if (cmp_utf16(o->sorted[mid]->key, o->sorted[mid]->key_len, key, len) < 0) lo = mid + 1; else hi = mid;
  }
  return lo;
}

static int elem_visible(const elem_t *el) {
  for (const row_t *r = el->rows; r; r = r->next)
    if (r->n_succ == 0) return 1;
  return 0;
}

/* ---- order-statistic tree over the elements of one list (a treap in list order, subtree sums of visible elements) ---- */
static uint32_t t_rand(void) {
  static uint64_t s = 0x9e3779b97f4a7c15ull;
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return (uint32_t)(s >> 32);
}
static uint32_t t_sum(const elem_t *n) { return n ? n->t_sum : 0; }
static void t_pull(elem_t *n) { n->t_sum = t_sum(n->t_left) + t_sum(n->t_right) + n->t_vis; }
static void t_rotate_up(obj_t *o, elem_t *n) {  /* n moves above its parent */
  elem_t *p = n->t_parent, *g = p->t_parent;
  if (p->t_left == n) { p->t_left = n->t_right; if (n->t_right) n->t_right->t_parent = p; n->t_right = p; }
  else { p->t_right = n->t_left; if (n->t_left) n->t_left->t_parent = p; n->t_left = p; }
  p->t_parent = n;
  n->t_parent = g;
  if (!g) o->t_root = n;
  else if (g->t_left == p) g->t_left = n; else g->t_right = n;
  t_pull(p);
  t_pull(n);
}
/* `n` becomes the in-order successor of `prev` (NULL: the first element) */
static void t_insert_after(obj_t *o, elem_t *prev, elem_t *n) {
  n->t_left = n->t_right = n->t_parent = NULL;
  n->t_prio = t_rand();
  n->t_vis = (uint32_t)elem_visible(n);
  n->t_sum = n->t_vis;
  if (!o->t_root) { o->t_root = n; return; }
  if (!prev) {
    elem_t *x = o->t_root;
    while (x->t_left) x = x->t_left;
    x->t_left = n; n->t_parent = x;
  } else if (!prev->t_right) { prev->t_right = n; n->t_parent = prev; }
  else {
    elem_t *x = prev->t_right;
    while (x->t_left) x = x->t_left;
    x->t_left = n; n->t_parent = x;
  }
  for (elem_t *x = n->t_parent; x; x = x->t_parent) t_pull(x);
  while (n->t_parent && n->t_parent->t_prio < n->t_prio) t_rotate_up(o, n);
}
static void t_build(obj_t *o) {
  if (o->t_built) return;
  o->t_built = 1;
  o->t_root = NULL;
  elem_t *prev = NULL;
  for (elem_t *x = o->head.next; x; x = x->next) { t_insert_after(o, prev, x); prev = x; }
}
/* the element's visibility changed (a row gained a successor, or a visible row was added) */
static void t_refresh(obj_t *o, elem_t *el) {
  if (!o->t_built) return;
  uint32_t v = (uint32_t)elem_visible(el);
  if (v == el->t_vis) return;
  el->t_vis = v;
  for (elem_t *x = el; x; x = x->t_parent) t_pull(x);
}
/* number of visible elements in front of `el` (NULL: of the whole list) -- the visibleCount of seekToOp, new.js:111-115, 148-152, 170-173 */
static uint64_t vis_before(obj_t *o, const elem_t *el) {
  t_build(o);
  if (!el) return t_sum(o->t_root);
  uint64_t n = t_sum(el->t_left);
  for (const elem_t *x = el; x->t_parent; x = x->t_parent)
    if (x->t_parent->t_right == x) n += t_sum(x->t_parent->t_left) + x->t_parent->t_vis;
  return n;
}

/* ---------------------------------------------------------------------------------------------------
 * cursor over the document ops of ONE object (readNextDocOp, new.js:658-670, confined to the object)
 * -------------------------------------------------------------------------------------------------*/
typedef struct {
  obj_t *o;
  int is_list;
  elem_t *prev_el, *el;  /* list: element of `row`, and its predecessor (the head sentinel included) */
  uint32_t si;           /* map: index of the slot of `row` in o->sorted */
  row_t *row;            /* the document op; NULL = past the end of the object */
} cur_t;

static void cur_next(cur_t *c) {
  if (!c->row) return;
  if (c->row->next) { c->row = c->row->next; return; }
  if (c->is_list) {
    c->prev_el = c->el;
    c->el = c->el->next;
    c->row = c->el ? c->el->rows : NULL;
  } else {
    c->si++;
    c->row = c->si < c->o->n_sorted ? c->o->sorted[c->si]->rows : NULL;
  }
}

/* ---------------------------------------------------------------------------------------------------
 * appendEdit / appendUpdate live in am_oracle.c (append_edit, append_update); convertInsertToUpdate new.js:838-869
 * -------------------------------------------------------------------------------------------------*/
static int convert_insert_to_update(ictx_t *x, pobj_t *p, uint64_t index, opid_t elem) {
  pedit_t *ups = NULL;
  uint32_t n = 0, cap = 0;
  int rc = 0;
  while (p->n_edits > 0) {
    pedit_t *last = &p->edits[p->n_edits - 1];
    if (last->action == E_INSERT || last->action == E_UPDATE) {
      if (last->index != index) { rc = fail(x->e, "last edit has unexpected index"); break; }
      if (n == cap) { cap = cap ? cap * 2 : 4; ups = (pedit_t *)realloc(ups, sizeof(pedit_t) * cap); }
      ups[n++] = *last;
      p->n_edits--;
      if (ups[n - 1].action == E_INSERT) break;
    } else { rc = fail(x->e, "last edit has unexpected action"); break; }
  }
  /* `updates.unshift`: popped last-to-first, re-appended first-to-last */
  for (uint32_t i = n; i-- > 0 && !rc;) append_update(p, index, elem, ups[i].opid, ups[i].val, i == n - 1);
  free(ups);
  return rc;
}

/* ---------------------------------------------------------------------------------------------------
 * updatePatchProperty, incremental mode (new.js:884-1040)
 * -------------------------------------------------------------------------------------------------*/
static iprop_t *prop_state(ictx_t *x, const pkey_t *k, int *first) {
  for (uint32_t i = 0; i < x->n_props; i++)
    if (pkey_eq(k, x->props[i].k.is_elem, x->props[i].k.elem, x->props[i].k.key, x->props[i].k.key_len)) { *first = 0; return &x->props[i]; }
  if (x->n_props == x->cap_props) {
    x->cap_props = x->cap_props ? x->cap_props * 2 : 8;
    x->props = (iprop_t *)realloc(x->props, sizeof(iprop_t) * x->cap_props);
  }
  iprop_t *ip = &x->props[x->n_props++];
  memset(ip, 0, sizeof *ip);
  ip->k = *k;
  *first = 1;
  return ip;
}
static void prop_state_reset(ictx_t *x) {
  for (uint32_t i = 0; i < x->n_props; i++) { free(x->props[i].visible); free(x->props[i].ps.cmap); }
  x->n_props = 0;
}

/* index of the patches' properties: (patch object, key) -> position in patch->props */
typedef struct pidx {
  const pobj_t *patch;
  const uint8_t *key;
  uint32_t key_len;
  uint64_t pos;
  struct pidx *next;
} pidx_t;
static tab_t g_pidx;       /* (single-threaded test tool: one applyChanges call at a time) */
static pool_t g_pidx_pool;
static uint64_t pidx_key(const pobj_t *patch, const uint8_t *key, uint32_t len) { return mix64((uint64_t)(uintptr_t)patch) ^ ((uint64_t)hash_bytes(key, len) * 0x9e3779b97f4a7c15ull); }

/* patch.props[key], insertion-ordered; `reset`: `patch.props[key] = {}` (an existing key keeps its place among the keys) */
static pprop_t *patch_prop(pobj_t *patch, const uint8_t *key, uint32_t len, int create, int reset) {
  pprop_t *pp = NULL;
  uint64_t hk = pidx_key(patch, key, len);
  void **slot = tab_slot(&g_pidx, hk, 0);
  if (slot)
    for (pidx_t *n = (pidx_t *)*slot; n; n = n->next)
      if (n->patch == patch && n->key_len == len && memcmp(n->key, key, len) == 0) { pp = &patch->props[n->pos]; break; }
  if (!pp) {
    if (!create) return NULL;
    if (patch->n_props == patch->cap_props) {
      patch->cap_props = patch->cap_props ? patch->cap_props * 2 : 8;
      patch->props = (pprop_t *)realloc(patch->props, patch->cap_props * sizeof(pprop_t));
    }
    pp = &patch->props[patch->n_props++];
    memset(pp, 0, sizeof *pp);
    pp->key = key;
    pp->key_len = len;
    slot = tab_slot(&g_pidx, hk, 1);
    pidx_t *n = (pidx_t *)pool_alloc(&g_pidx_pool, sizeof *n);
    n->patch = patch; n->key = key; n->key_len = len; n->pos = patch->n_props - 1; n->next = (pidx_t *)*slot;
    *slot = n;
  } else if (reset) pp->n = 0;
  return pp;
}
static void prop_put(pprop_t *pp, opid_t opid, pval_t v) {
  uint32_t k = 0;
  while (k < pp->n && !same_id(pp->ents[k].opid, opid)) k++;
  if (k == pp->n) {
    if (pp->n == pp->cap) {
      pp->cap = pp->cap ? pp->cap * 2 : 2;
      pp->ents = (pent_t *)realloc(pp->ents, pp->cap * sizeof(pent_t));
    }
    pp->n++;
    pp->ents[k].opid = opid;
  }
  pp->ents[k].val = v;
}

static void touch_object(ictx_t *x, obj_t *o) {  /* objectIds.add(objectId) */
  if (o->touch_epoch == x->d->epoch) return;
  o->touch_epoch = x->d->epoch;
  if (x->n_touched == x->cap_touched) {
    x->cap_touched = x->cap_touched ? x->cap_touched * 2 : 16;
    x->touched = (obj_t **)realloc(x->touched, sizeof(obj_t *) * x->cap_touched);
  }
  x->touched[x->n_touched++] = o;
}

/* `r` is a document op (is_doc, old_succ = its succNum before this call touched it) or an op of the change just placed in the
 * document (oldSuccNum undefined) */
static int upp_inc(ictx_t *x, obj_t *o, const pkey_t *k, row_t *r, uint64_t list_index, int is_doc, uint32_t old_succ) {
  amo_doc *d = x->d;
  const int is_make = (r->action & 1) == 0;
  /* new.js:894-897 */
  if (is_make) {
    obj_t *child = obj_by_id(d, r->id);
    if (child && !child->has_meta) {
      child->has_meta = 1;
      child->parent = o;
      child->parent_elem = k->elem;
      child->parent_key = k->key;
      child->parent_key_len = k->key_len;
      chkey_t *ck = children_get(o, k, 1, &d->pool);
      uint32_t j = 0;
      while (j < ck->n && !same_id(ck->vals[j].opid, r->id)) j++;
      if (j == ck->n) {
        chval_t *nv = (chval_t *)pool_alloc(&d->pool, sizeof(chval_t) * (ck->n + 1));
        if (ck->n) memcpy(nv, ck->vals, sizeof(chval_t) * ck->n);
        ck->vals = nv;
        ck->n++;
      }
      memset(&ck->vals[j], 0, sizeof(chval_t));
      ck->vals[j].opid = r->id; ck->vals[j].is_obj = 1; ck->vals[j].type = (int)r->action;
    }
  }
  int first_op;
  iprop_t *ip = prop_state(x, k, &first_op);
  const int overwritten = is_doc && r->n_succ > 0;  /* new.js:904 */
  if (!overwritten) {
    if (ip->n_vis == ip->cap_vis) { ip->cap_vis = ip->cap_vis ? ip->cap_vis * 2 : 4; ip->visible = (row_t **)realloc(ip->visible, sizeof(row_t *) * ip->cap_vis); }
    ip->visible[ip->n_vis++] = r;
    ip->has_child = ip->has_child || is_make;
  }
  /* new.js:916-931 */
  chkey_t *prev = children_get(o, k, 0, NULL);
  if (ip->has_child || (prev && prev->n > 0)) {
    chkey_t *ck = children_get(o, k, 1, &d->pool);
    chval_t *nv = (chval_t *)pool_alloc(&d->pool, sizeof(chval_t) * (ip->n_vis ? ip->n_vis : 1));
    uint32_t n = 0;
    for (uint32_t i = 0; i < ip->n_vis; i++) {
      row_t *v = ip->visible[i];
      chval_t cv;
      memset(&cv, 0, sizeof cv);
      cv.opid = v->id;
      if (v->action == 1) { cv.tag_len = v->val_tag_len; cv.bytes = v->val; }
      else if ((v->action & 1) == 0) { cv.is_obj = 1; cv.type = (int)v->action; }
      else continue;
      /* `values[opId] = ...`: an id seen twice keeps its first position */
      uint32_t j = 0;
      while (j < n && !same_id(nv[j].opid, cv.opid)) j++;
      nv[j] = cv;
      if (j == n) n++;
    }
    ck->vals = nv;
    ck->n = n;
  }

  pstate_t *ps = &ip->ps;
  int have_val = 0;
  opid_t patch_key = r->id;
  pval_t pv;
  memset(&pv, 0, sizeof pv);
  if (overwritten && r->action == 1 && (r->val_tag_len & 15) == 8) {
    /* new.js:937-951 */
    cstate_t *st = (cstate_t *)pool_alloc(&x->pc.pool, sizeof *st);
    pval_t tmp = {0, r->val_tag_len, r->val, 0, NULL};
    st->opid = r->id;
    if (decode_int_value(&tmp, &st->value, x->e)) return -1;
    st->outstanding = r->n_succ;
    for (uint32_t i = 0; i < r->n_succ; i++) {
      uint32_t q = 0;
      while (q < ps->n_cmap && !same_id(ps->cmap[q].succ, r->succ[i])) q++;
      if (q == ps->n_cmap) {
        if (ps->n_cmap == ps->cap_cmap) { ps->cap_cmap = ps->cap_cmap ? ps->cap_cmap * 2 : 4; ps->cmap = (cmap_t *)realloc(ps->cmap, sizeof(cmap_t) * ps->cap_cmap); }
        ps->n_cmap++;
        ps->cmap[q].succ = r->succ[i];
      }
      ps->cmap[q].st = st;
    }
  } else if (r->action == 5) {
    /* new.js:952-965 */
    uint32_t q = 0;
    while (q < ps->n_cmap && !same_id(ps->cmap[q].succ, r->id)) q++;
    if (q == ps->n_cmap) {
      char b[160];
      fmt_opid(d, r->id, b, sizeof b);
      return fail(x->e, "increment operation %s for unknown counter", b);
    }
    cstate_t *st = ps->cmap[q].st;
    pval_t tmp = {0, r->val_tag_len, r->val, 0, NULL};
    int64_t inc = 0;
    uint64_t tag = r->val_tag_len & 15;
    if (tag == 3 || tag == 4 || tag == 8 || tag == 9) { if (decode_int_value(&tmp, &inc, x->e)) return -1; }
    else return fail(x->e, "unsupported: non-integer increment");
    st->value += inc;
    if (st->outstanding > 0) {
      st->outstanding--;
      ps->cmap[q].succ.ctr = 0;
      ps->cmap[q].succ.actor = NUL32;
    }
    if (st->outstanding == 0) {
      have_val = 1;
      patch_key = st->opid;
      pv.kind = 1;
      pv.counter = st->value;
    }
  } else if (!overwritten) {
    /* new.js:967-977 */
    if (r->action == 1) {
      have_val = 1;
      pv.kind = 0; pv.tag_len = r->val_tag_len; pv.bytes = r->val;
    } else if (is_make) {
      have_val = 1;
      pv.kind = 2;
      pv.obj = get_patch(&x->pc, r->id, (int)r->action);
    }
  }

  pobj_t *patch = patch_of(x, o);
  if (k->is_elem) {
    /* new.js:983-1033 */
    if (is_doc && old_succ == 0 && ps->action == 1) {
      ps->action = 2;
      if (convert_insert_to_update(x, patch, list_index, k->elem)) return -1;
    }
    if (have_val) {
      if (ps->action == 0 && !is_doc) {
        pedit_t e;
        memset(&e, 0, sizeof e);
        ps->action = 1;
        e.action = E_INSERT; e.index = list_index; e.elem = k->elem; e.opid = patch_key; e.val = pv;
        append_edit(patch, &e);
      } else if (ps->action == 3) {
        if (patch->n_edits == 0 || patch->edits[patch->n_edits - 1].action != E_REMOVE) return fail(x->e, "last edit has unexpected type");
        pedit_t *last = &patch->edits[patch->n_edits - 1];
        if (last->count > 1) last->count--; else patch->n_edits--;
        ps->action = 2;
        append_update(patch, list_index, k->elem, patch_key, pv, 1);
      } else {
        append_update(patch, list_index, k->elem, patch_key, pv, ps->action == 0);
        if (ps->action == 0) ps->action = 2;
      }
    } else if (is_doc && old_succ == 0 && ps->action == 0) {
      pedit_t e;
      memset(&e, 0, sizeof e);
      ps->action = 3;
      e.action = E_REMOVE; e.index = list_index; e.count = 1;
      append_edit(patch, &e);
    }
  } else {
    /* new.js:1035-1039 (`patchValue || !isWholeDoc` is always true here) */
    pprop_t *pp = patch_prop(patch, k->key, k->key_len, 1, first_op);
    if (have_val) prop_put(pp, patch_key, pv);
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * mergeDocChangeOps (new.js:1052-1290) preceded by the seek of applyOps (new.js:1304-1314, 227-317, 50-192)
 * -------------------------------------------------------------------------------------------------*/
static void fmt_elem(const amo_doc *d, opid_t id, char *buf, size_t cap) { fmt_opid(d, id, buf, cap); }

static int str_eq(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb) { return na == nb && memcmp(a, b, na) == 0; }

static obj_t *new_object(amo_doc *d, opid_t id, uint32_t action, err_t *e) {
  void **s = tab_slot(&d->objs, idkey(id), 1);
  if (*s) { char buf[160]; fmt_opid(d, id, buf, sizeof buf); fail(e, "duplicate operation ID: %s", buf); return NULL; }
  obj_t *no = (obj_t *)pool_alloc(&d->pool, sizeof *no);
  no->id = id;
  no->type = (int)action;
  *s = no;
  if (d->n_objs == d->cap_objs) {
    d->cap_objs = d->cap_objs ? d->cap_objs * 2 : 64;
    d->obj_list = (obj_t **)realloc(d->obj_list, d->cap_objs * sizeof(obj_t *));
  }
  d->obj_list[d->n_objs++] = no;
  return no;
}

static row_t *new_row(amo_doc *d, const cop_t *op) {
  row_t *r = (row_t *)pool_alloc(&d->pool, sizeof *r);
  r->id = op->id; r->insert = op->insert; r->action = op->action; r->val_tag_len = op->val_tag_len; r->val = op->val;
  d->n_rows++;
  return r;
}

/* the change op goes into the document right in front of the cursor (appendOperation in the take-change-ops branch, new.js:1259) */
static row_t *place_row(ictx_t *x, cur_t *c, const cop_t *op, pkey_t *k_out) {
  amo_doc *d = x->d;
  obj_t *o = c->o;
  char buf[160];
  if (op->action == 3) { fail(x->e, "unsupported: del operation stored as a row"); return NULL; }
  row_t *r = NULL;
  if (c->is_list) {
    if (op->insert) {
      /* a new element in front of the cursor's element */
      if (c->row && c->row != c->el->rows) { fail(x->e, "unsupported: insertion inside an element"); return NULL; }
      if (find_elem(d, o->id, op->id)) { fmt_opid(d, op->id, buf, sizeof buf); fail(x->e, "duplicate operation ID: %s", buf); return NULL; }
      elem_t *el = (elem_t *)pool_alloc(&d->pool, sizeof *el);
      r = new_row(d, op);
      el->rows = r;
      el->next = c->el;
      c->prev_el->next = el;
      if (o->t_built) t_insert_after(o, c->prev_el == &o->head ? NULL : c->prev_el, el);
      c->prev_el = el;
      o->n_elems++;
      index_elem(d, o->id, op->id, el);
      k_out->is_elem = 1; k_out->elem = op->id; k_out->key = NULL; k_out->key_len = 0;
    } else {
      elem_t *el = find_elem(d, o->id, op->key);
      if (!el) { fmt_opid(d, op->key, buf, sizeof buf); fail(x->e, "could not find list element with ID: %s", buf); return NULL; }
      r = new_row(d, op);
      if (c->row && c->el == el && c->row != el->rows) {
        row_t *p = el->rows;
        while (p->next != c->row) p = p->next;
        r->next = c->row;
        p->next = r;
      } else {
        if (c->row && c->el == el) { fail(x->e, "unsupported: update in front of its element"); return NULL; }
        row_t *p = el->rows;
        while (p->next) p = p->next;
        p->next = r;
      }
      t_refresh(o, el);
      k_out->is_elem = 1; k_out->elem = op->key; k_out->key = NULL; k_out->key_len = 0;
    }
  } else {
    slot_t *s = find_slot(d, o, op->key_str, op->key_len, 0);
    r = new_row(d, op);
    if (!s) {
      s = find_slot(d, o, op->key_str, op->key_len, 1);
      uint32_t at = sorted_lower_bound(o, op->key_str, op->key_len);
      uint32_t want = c->row ? c->si : o->n_sorted;
      if (at != want) { fail(x->e, "unsupported: key lands out of document order"); return NULL; }
      if (o->n_sorted == o->cap_sorted) {
        o->cap_sorted = o->cap_sorted ? o->cap_sorted * 2 : 16;
        o->sorted = (slot_t **)realloc(o->sorted, sizeof(slot_t *) * o->cap_sorted);
      }
      memmove(o->sorted + at + 1, o->sorted + at, sizeof(slot_t *) * (o->n_sorted - at));
      o->sorted[at] = s;
      o->n_sorted++;
      if (c->row) c->si++;
      s->rows = r;
    } else if (c->row && c->o->sorted[c->si] == s) {
      if (c->row == s->rows) { r->next = s->rows; s->rows = r; }
      else {
        row_t *p = s->rows;
        while (p->next != c->row) p = p->next;
        r->next = c->row;
        p->next = r;
      }
    } else {
      row_t *p = s->rows;
      while (p->next) p = p->next;
      p->next = r;
    }
    k_out->is_elem = 0; k_out->elem.ctr = 0; k_out->elem.actor = 0; k_out->key = s->key; k_out->key_len = s->key_len;
  }
  if ((op->action & 1) == 0 && !new_object(d, op->id, op->action, x->e)) return NULL;
  return r;
}

typedef struct {
  const cop_t *op;
  uint8_t *seen;  /* predSeen */
} chop_t;

static int merge_call(ictx_t *x, const cop_t *ops, uint64_t n_ops, uint64_t *pos) {
  amo_doc *d = x->d;
  err_t *e = x->e;
  char buf[160];
  const uint64_t i0 = *pos;
  uint64_t i = i0;
  const cop_t *first = &ops[i0];
  const int insert = first->insert;
  obj_t *o = obj_by_id(d, first->obj);
  if (!o) { fmt_opid(d, first->obj, buf, sizeof buf); return fail(e, "unsupported: operation on unknown object %s", buf); }
  const int is_list = is_list_type(o->type);
  if (first->key_str) {
    if (is_list) return fail(e, "unsupported: string key used in a list object");
    if (first->insert) return fail(e, "unsupported: insert flag on a map operation");
  } else if (!is_list) return fail(e, "unsupported: list operation on a map object");

  /* ---- seekToOp ---- */
  cur_t c;
  memset(&c, 0, sizeof c);
  c.o = o;
  c.is_list = is_list;
  uint64_t list_index = 0;
  if (!is_list) {
    ensure_sorted(o);
    c.si = sorted_lower_bound(o, first->key_str, first->key_len);
    c.row = c.si < o->n_sorted ? o->sorted[c.si]->rows : NULL;
  } else if (insert) {
    elem_t *ref = &o->head;
    if (first->key.ctr != 0) {
      ref = find_elem(d, o->id, first->key);
      if (!ref) { fmt_opid(d, first->key, buf, sizeof buf); return fail(e, "Reference element not found: %s", buf); }
    }
    /* new.js:144-163: skip every following element with a greater id (and the rows that are not insertions) */
    while (ref->next && cmp_opid(d, ref->next->rows->id, first->id) > 0) ref = ref->next;
    c.prev_el = ref;
    c.el = ref->next;
    c.row = c.el ? c.el->rows : NULL;
    list_index = vis_before(o, c.el);
  } else {
    if (first->key.ctr == 0) return fail(e, "unsupported: non-insert operation on _head");
    elem_t *el = find_elem(d, o->id, first->key);
    if (!el) { fmt_opid(d, first->key, buf, sizeof buf); return fail(e, "Reference element not found: %s", buf); }
    c.prev_el = NULL;
    c.el = el;
    c.row = el->rows;
    list_index = vis_before(o, el);
  }

  int found_list_elem = 0, elem_vis = 0, rc = 0;
  uint32_t doc_old_succ = c.row ? c.row->n_succ : 0;
  chop_t *ch = NULL;
  uint32_t n_ch = 0, cap_ch = 0;
  const cop_t *change_op = NULL;
  const uint8_t *last_key = NULL;
  uint32_t last_key_len = 0;
  int have_last_key = 0;
  prop_state_reset(x);
  touch_object(x, o);

  for (;;) {
    if (n_ch == 0) {
      found_list_elem = 0;
      while (i < n_ops) {
        const cop_t *nx = &ops[i];
        if (!(nx->id.actor == first->id.actor && nx->insert == insert && same_id(nx->obj, first->obj))) break;
        const cop_t *last = n_ch ? ch[n_ch - 1].op : NULL;
        int is_overwrite = 0;
        for (uint32_t p = 0; p < nx->pred_num; p++)
          for (uint32_t q = 0; q < n_ch; q++)
            if (same_id(nx->preds[p], ch[q].op->id)) is_overwrite = 1;
        if (i == i0) {
          /* the first change op of the call is always used */
        } else if (insert && last && !nx->key_str && same_id(nx->key, last->id)) {
          /* consecutive insertions */
        } else if (!insert && last && nx->key_str && last->key_str && str_eq(nx->key_str, nx->key_len, last->key_str, last->key_len) && !is_overwrite) {
          /* several updates of the same key */
        } else if (!insert && last && !nx->key_str && !last->key_str && same_id(nx->key, last->key) && !is_overwrite) {
          /* several updates of the same list element */
        } else if (!insert && !last && !nx->key_str && c.row && is_list && c.row->insert && same_id(c.row->id, nx->key)) {
          /* the next element of the change is the next element of the document */
        } else if (!insert && !last && nx->key_str && have_last_key && cmp_utf16(last_key, last_key_len, nx->key_str, nx->key_len) < 0) {
          /* several keys of one object in ascending order */
        } else break;
        if (nx->key_str ? is_list : !is_list) { rc = fail(e, "unsupported: key type does not match the object"); goto done; }
        have_last_key = nx->key_str != NULL;
        last_key = nx->key_str;
        last_key_len = nx->key_len;
        if (n_ch == cap_ch) { cap_ch = cap_ch ? cap_ch * 2 : 8; ch = (chop_t *)realloc(ch, sizeof(chop_t) * cap_ch); }
        ch[n_ch].op = nx;
        ch[n_ch].seen = (uint8_t *)calloc(nx->pred_num ? nx->pred_num : 1, 1);
        n_ch++;
        i++;
      }
    }
    if (n_ch > 0) change_op = ch[0].op;
    row_t *doc = c.row;
    const int in_obj = doc != NULL;
    int key_matches = 0, elem_matches = 0;
    if (doc && !is_list && change_op->key_str) key_matches = str_eq(o->sorted[c.si]->key, o->sorted[c.si]->key_len, change_op->key_str, change_op->key_len);
    if (doc && is_list && !change_op->key_str) elem_matches = same_id(c.el->rows->id, change_op->key);
    if (n_ch == 0 && !(in_obj && (key_matches || elem_matches))) break;

    int take_doc = 0;
    uint32_t take_ch = 0;
    if (insert || !in_obj ||
        (!is_list && change_op->key_str && cmp_utf16(change_op->key_str, change_op->key_len, o->sorted[c.si]->key, o->sorted[c.si]->key_len) < 0)) {
      take_ch = n_ch;
      if (!in_obj && !found_list_elem && !change_op->key_str && !change_op->insert) {
        fmt_elem(d, change_op->key, buf, sizeof buf);
        rc = fail(e, "could not find list element with ID: %s", buf);
        goto done;
      }
    } else if (key_matches || elem_matches || found_list_elem) {
      /* pred -> succ (new.js:1173-1188) */
      for (uint32_t q = 0; q < n_ch; q++) {
        const cop_t *op = ch[q].op;
        for (uint32_t p = 0; p < op->pred_num; p++)
          if (same_id(op->preds[p], doc->id)) {
            add_succ(d, doc, op->id);
            if (is_list) t_refresh(o, c.el);
            ch[q].seen[p] = 1;
            break;
          }
      }
      if (elem_matches) found_list_elem = 1;
      if (found_list_elem && !elem_matches) take_ch = n_ch;
      else if (n_ch == 0 || cmp_opid(d, doc->id, change_op->id) < 0) {
        take_doc = 1;
        pkey_t k;
        memset(&k, 0, sizeof k);
        if (is_list) { k.is_elem = 1; k.elem = c.el->rows->id; }
        else { k.key = o->sorted[c.si]->key; k.key_len = o->sorted[c.si]->key_len; }
        if ((rc = upp_inc(x, o, &k, doc, list_index, 1, doc_old_succ))) goto done;
        /* a deletion leaves no row, only its entries in the succ lists (new.js:1205-1217) */
        for (uint32_t q = n_ch; q-- > 0;) {
          int deleted = 1;
          for (uint32_t p = 0; p < ch[q].op->pred_num; p++) if (!ch[q].seen[p]) deleted = 0;
          if (ch[q].op->action == 3 && deleted) {
            free(ch[q].seen);
            memmove(ch + q, ch + q + 1, sizeof(chop_t) * (n_ch - q - 1));
            n_ch--;
          }
        }
      } else if (cmp_opid(d, doc->id, change_op->id) == 0) {
        fmt_opid(d, change_op->id, buf, sizeof buf);
        rc = fail(e, "duplicate operation ID: %s", buf);
        goto done;
      } else take_ch = 1;
    } else {
      take_doc = 1;
      if (!is_list && n_ch > 0 && change_op->key_str) {
        /* the document ops in front of the change op's key go by without a word (new.js:1226-1229): step over them in one go
         * (the reference decodes them one by one; nothing but the position changes) */
        uint32_t at = sorted_lower_bound(o, change_op->key_str, change_op->key_len);
        if (at > c.si) {
          c.si = at;
          c.row = c.si < o->n_sorted ? o->sorted[c.si]->rows : NULL;
          if (c.row) doc_old_succ = c.row->n_succ;
          continue;
        }
      }
    }

    if (take_doc) {
      if (doc->insert && elem_vis) { elem_vis = 0; list_index++; }
      if (doc->n_succ == 0) elem_vis = 1;
      cur_next(&c);
      if (c.row) doc_old_succ = c.row->n_succ;
    }
    if (take_ch > 0) {
      for (uint32_t q = 0; q < take_ch; q++) {
        const cop_t *op = ch[q].op;
        for (uint32_t p = 0; p < op->pred_num; p++)
          if (!ch[q].seen[p]) {
            fmt_opid(d, op->preds[p], buf, sizeof buf);
            rc = fail(e, "no matching operation for pred: %s", buf);
            goto done;
          }
        pkey_t k;
        row_t *r = place_row(x, &c, op, &k);
        if (!r) { rc = -1; goto done; }
        if ((rc = upp_inc(x, o, &k, r, list_index, 0, 0))) goto done;
        if (op->insert) { elem_vis = 0; list_index++; } else elem_vis = 1;
      }
      for (uint32_t q = 0; q < take_ch; q++) free(ch[q].seen);
      memmove(ch, ch + take_ch, sizeof(chop_t) * (n_ch - take_ch));
      n_ch -= take_ch;
    }
  }
done:
  for (uint32_t q = 0; q < n_ch; q++) free(ch[q].seen);
  free(ch);
  *pos = i;
  return rc;
}

/* ---------------------------------------------------------------------------------------------------
 * setupPatches (new.js:1461-1528)
 * -------------------------------------------------------------------------------------------------*/
static pval_t chval_to_pval(ictx_t *x, const chval_t *v) {
  pval_t pv;
  memset(&pv, 0, sizeof pv);
  if (v->is_obj) { pv.kind = 2; pv.obj = get_patch(&x->pc, v->opid, v->type); }
  else { pv.kind = 0; pv.tag_len = v->tag_len; pv.bytes = v->bytes; }
  return pv;
}

static int setup_patches(ictx_t *x) {
  amo_doc *d = x->d;
  for (uint32_t t = 0; t < x->n_touched; t++) {
    obj_t *o = x->touched[t], *child = NULL;
    int patch_exists = 0;
    for (;;) {
      chkey_t *ck = NULL;
      if (child) {
        pkey_t k;
        memset(&k, 0, sizeof k);
        if (is_list_type(o->type)) { k.is_elem = 1; k.elem = child->parent_elem; }
        else { k.key = child->parent_key; k.key_len = child->parent_key_len; }
        ck = children_get(o, &k, 0, NULL);
        if (!ck) return fail(x->e, "unsupported: objectMeta without the child's property");
      }
      const int has_children = child && ck->n > 0;
      pobj_t *patch = patch_of(x, o);
      if (child && has_children) {
        if (is_list_type(o->type)) {
          for (uint64_t i = 0; i < patch->n_edits; i++) {
            pedit_t *ed = &patch->edits[i];
            if (ed->action != E_INSERT && ed->action != E_UPDATE) continue;  /* `edit.opId` */
            for (uint32_t j = 0; j < ck->n; j++) if (same_id(ck->vals[j].opid, ed->opid)) patch_exists = 1;
          }
          if (!patch_exists) {
            elem_t *el = find_elem(d, o->id, child->parent_elem);
            if (!el) { char buf[160]; fmt_opid(d, child->parent_elem, buf, sizeof buf); return fail(x->e, "Reference element not found: %s", buf); }
            uint64_t vc = vis_before(o, el);
            for (uint32_t j = 0; j < ck->n; j++) {
              pedit_t ed;
              memset(&ed, 0, sizeof ed);
              ed.action = E_UPDATE; ed.index = vc; ed.opid = ck->vals[j].opid; ed.val = chval_to_pval(x, &ck->vals[j]);
              append_edit(patch, &ed);
            }
          }
        } else {
          pprop_t *pp = patch_prop(patch, child->parent_key, child->parent_key_len, 1, 0);
          for (uint32_t j = 0; j < ck->n; j++) {
            int present = 0;
            for (uint32_t q = 0; q < pp->n; q++) if (same_id(pp->ents[q].opid, ck->vals[j].opid)) present = 1;
            if (present) patch_exists = 1;
            else prop_put(pp, ck->vals[j].opid, chval_to_pval(x, &ck->vals[j]));
          }
        }
      }
      if (patch_exists || !o->parent || (child && !has_children)) break;
      child = o;
      o = o->parent;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * objectMeta of a loaded document = what documentPatch leaves behind (new.js:1604-1635 calling 884-931 with isWholeDoc)
 * -------------------------------------------------------------------------------------------------*/
static void meta_visit(amo_doc *d, obj_t *o, const pkey_t *k, row_t *rows) {
  row_t **vis = NULL;
  uint32_t n_vis = 0, cap = 0;
  int has_child = 0;
  for (row_t *r = rows; r; r = r->next) {
    const int is_make = (r->action & 1) == 0;
    if (is_make) {
      obj_t *child = obj_by_id(d, r->id);
      if (child && !child->has_meta) {
        child->has_meta = 1; child->parent = o; child->parent_elem = k->elem; child->parent_key = k->key; child->parent_key_len = k->key_len;
        chkey_t *ck = children_get(o, k, 1, &d->pool);
        chval_t *nv = (chval_t *)pool_alloc(&d->pool, sizeof(chval_t) * (ck->n + 1));
        if (ck->n) memcpy(nv, ck->vals, sizeof(chval_t) * ck->n);
        nv[ck->n].opid = r->id; nv[ck->n].is_obj = 1; nv[ck->n].type = (int)r->action;
        ck->vals = nv;
        ck->n++;
      }
    }
    if (r->n_succ == 0) {
      if (n_vis == cap) { cap = cap ? cap * 2 : 4; vis = (row_t **)realloc(vis, sizeof(row_t *) * cap); }
      vis[n_vis++] = r;
      has_child = has_child || is_make;
    }
    chkey_t *prev = children_get(o, k, 0, NULL);
    if (has_child || (prev && prev->n > 0)) {
      chkey_t *ck = children_get(o, k, 1, &d->pool);
      chval_t *nv = (chval_t *)pool_alloc(&d->pool, sizeof(chval_t) * (n_vis ? n_vis : 1));
      uint32_t n = 0;
      for (uint32_t i = 0; i < n_vis; i++) {
        chval_t cv;
        memset(&cv, 0, sizeof cv);
        cv.opid = vis[i]->id;
        if (vis[i]->action == 1) { cv.tag_len = vis[i]->val_tag_len; cv.bytes = vis[i]->val; }
        else if ((vis[i]->action & 1) == 0) { cv.is_obj = 1; cv.type = (int)vis[i]->action; }
        else continue;
        nv[n++] = cv;
      }
      ck->vals = nv;
      ck->n = n;
    }
  }
  free(vis);
}

static void build_meta(amo_doc *d) {
  d->root.has_meta = 1;
  obj_t **objs = sorted_objs(d);
  for (uint64_t oi = 0; oi <= d->n_objs; oi++) {
    obj_t *o = objs[oi];
    pkey_t k;
    memset(&k, 0, sizeof k);
    if (is_list_type(o->type)) {
      for (elem_t *el = o->head.next; el; el = el->next) { k.is_elem = 1; k.elem = el->rows->id; meta_visit(d, o, &k, el->rows); }
    } else {
      ensure_sorted(o);
      for (uint32_t i = 0; i < o->n_sorted; i++) { k.is_elem = 0; k.key = o->sorted[i]->key; k.key_len = o->sorted[i]->key_len; meta_visit(d, o, &k, o->sorted[i]->rows); }
    }
  }
  free(objs);
  d->meta_built = 1;
}

/* ---------------------------------------------------------------------------------------------------
 * BackendDoc.applyChanges (new.js:1797-1879) with the scheduling passes of applyChanges (new.js:1550-1597)
 * -------------------------------------------------------------------------------------------------*/
static int head_find(const uint8_t **heads, uint32_t n, const uint8_t *h) {
  for (uint32_t i = 0; i < n; i++) if (heads[i] && memcmp(heads[i], h, 32) == 0) return (int)i;
  return -1;
}

static void release_patches(ictx_t *x) {
  pctx_t *c = &x->pc;
  for (uint64_t i = 0; i < c->patches.cap; i++)
    if (c->patches.vals && c->patches.vals[i]) {
      pobj_t *p = (pobj_t *)c->patches.vals[i];
      for (uint64_t k = 0; k < p->n_props; k++) free(p->props[k].ents);
      for (uint64_t k = 0; k < p->n_edits; k++) if (p->edits[k].action == E_MULTI) free(p->edits[k].vals);
      free(p->props);
      free(p->edits);
    }
  for (uint64_t k = 0; k < c->root->n_props; k++) free(c->root->props[k].ents);
  free(c->root->props);
  tab_free(&c->patches);
  pool_free(&c->pool);
  tab_free(&g_pidx);
  pool_free(&g_pidx_pool);
  prop_state_reset(x);
  free(x->props);
  free(x->touched);
}

/* computeHashGraph (new.js:1887-1912): changeIndexByHash becomes the index of the document as saved BEFORE the running call -- the
 * document's own changes and what earlier calls applied; the hashes of the changes the running call has applied so far are not in it */
static void rebuild_hash_graph(amo_doc *d) {
  tab_free(&d->known);
  memset(&d->known, 0, sizeof d->known);
  for (uint32_t i = 0; i < d->n_doc_hashes; i++) hset_add(&d->pool, &d->known, d->doc_hashes + 32 * (size_t)i);
  for (uint32_t i = 0; i < d->n_since; i++) hset_add(&d->pool, &d->known, d->since[i]);
  d->have_graph = 1;
}

static void enter_session(amo_doc *d) {
  if (d->session) return;
  /* a document made by amo_load_document enters session mode: the heads are the only change hashes it knows (new.js:1711-1730) */
  d->session = 1;
  d->loaded = 1;
  d->actors_read = d->n_actors;
  for (uint32_t i = 0; i < d->n_heads; i++) hset_add(&d->pool, &d->known, d->heads + 32 * i);
}

void amo_set_document_history(amo_doc *d, const uint8_t *hashes, uint32_t n, int rebuilt) {
  enter_session(d);
  d->doc_hashes = hashes;
  d->n_doc_hashes = n;
  if (rebuilt && !d->have_graph) rebuild_hash_graph(d);
}

const char *amo_apply_changes(amo_doc *d, const uint8_t *arena, const uint64_t *offsets, uint32_t n, int is_local, size_t *len,
                              char *errbuf, size_t errcap) {
  err_t e = {{0}, 0};
  int rc = 0;
  enter_session(d);
  if (!d->meta_built) build_meta(d);
  d->json_done = 0;
  d->json.len = 0;
  d->epoch++;

  ictx_t x;
  memset(&x, 0, sizeof x);
  x.d = d;
  x.e = &e;
  x.pc.d = d;
  x.pc.e = &e;
  pobj_t root;
  memset(&root, 0, sizeof root);
  root.is_root = 1;
  x.pc.root = &root;

  /* decodeChangeColumns for every buffer (new.js:1806-1810); the buffers are copied, the document keeps pointers into them */
  uint32_t qn = n + d->n_queue;
  change_t *queue = (change_t *)calloc(qn ? qn : 1, sizeof(change_t));
  change_t *next_q = (change_t *)calloc(qn ? qn : 1, sizeof(change_t));
  change_t **applied = (change_t **)calloc(qn ? qn : 1, sizeof(change_t *));
  for (uint32_t i = 0; i < n && !rc; i++) {
    size_t l = (size_t)(offsets[i + 1] - offsets[i]);
    uint8_t *copy = (uint8_t *)pool_alloc(&d->pool, l ? l : 1);
    memcpy(copy, arena + offsets[i], l);
    rc = parse_change(&d->pool, copy, l, &queue[i], &e);
  }
  for (uint32_t i = 0; i < d->n_queue; i++) queue[n + i] = d->queue[i].c;  /* decodedChanges.concat(this.queue) */
  change_t first_decoded;
  memset(&first_decoded, 0, sizeof first_decoded);
  if (n == 1 && !rc) first_decoded = queue[0];

  /* heads as a set of pointers */
  uint32_t head_cap = d->n_heads + qn + 1, n_heads = 0;
  const uint8_t **heads = (const uint8_t **)calloc(head_cap, sizeof(uint8_t *));
  for (uint32_t i = 0; i < d->n_heads; i++) heads[n_heads++] = d->heads + 32 * i;
  uint32_t *atab = NULL;
  size_t atab_cap = 0;
  cop_t *cops = NULL;
  uint64_t cap_cops = 0;
  int any_applied = 0;

  const uint8_t **call_applied = (const uint8_t **)calloc(qn ? qn : 1, sizeof(uint8_t *));  /* hashes this call has applied (all rounds) */
  uint32_t n_call_applied = 0;
  while (!rc && qn > 0) {
    uint32_t na = 0, nq = 0;
    /* a round commits its clock / heads / actor table only when it is not abandoned (new.js:1551-1552 works on copies, :1581) */
    const uint32_t snap_actors = d->n_actors, snap_clock_n = d->n_clock, snap_heads = n_heads, snap_call = n_call_applied;
    uint64_t *snap_clock = (uint64_t *)malloc(8 * (size_t)(d->n_actors ? d->n_actors : 1));
    memcpy(snap_clock, d->clock, 8 * (size_t)d->n_actors);
    const uint8_t **snap_head_ptrs = (const uint8_t **)malloc(sizeof(uint8_t *) * (size_t)(n_heads ? n_heads : 1));
    memcpy(snap_head_ptrs, heads, sizeof(uint8_t *) * (size_t)n_heads);
    tab_t round_known;  /* changeHashes of this round (new.js:1551): committed to changeIndexByHash after the round (:1828-1830) */
    memset(&round_known, 0, sizeof round_known);
    int abandoned = 0;
    for (uint32_t qi = 0; qi < qn && !rc; qi++) {
      change_t *c = &queue[qi];
      if (hset_has(&d->known, c->hash) || hset_has(&round_known, c->hash)) continue;
      int ai = doc_actor_index(d, c->actors[0]);
      uint64_t expected = (ai >= 0 ? d->clock[ai] : 0) + 1;
      int ready = 1;
      for (uint32_t k = 0; k < c->n_deps; k++) if (!hset_has(&d->known, c->deps + 32 * k) && !hset_has(&round_known, c->deps + 32 * k)) ready = 0;
      if (!ready) { next_q[nq++] = *c; continue; }
      char hex[80];
      size_t hl = 0;
      for (size_t k = 0; k < c->actors[0].len && hl + 2 < sizeof hex; k++) hl += snprintf(hex + hl, sizeof hex - hl, "%02x", c->actors[0].p[k]);
      hex[hl] = 0;
      if (c->seq < expected && !d->have_graph) {
        /* a change the document may hold already, whose hash is not known yet: nothing of this round is applied, the whole queue
         * waits for the hash graph (new.js:1578-1582) */
        abandoned = 1;
        break;
      }
      if (c->seq < expected) { rc = fail(&e, "Reuse of sequence number %llu for actor %s", (unsigned long long)c->seq, hex); break; }
      if (c->seq > expected) { rc = fail(&e, "Skipped sequence number %llu for actor %s", (unsigned long long)expected, hex); break; }
      if (ai < 0) {
        if (d->n_actors == d->cap_actors) {
          d->cap_actors = d->cap_actors ? d->cap_actors * 2 : 16;
          d->actors = (span_t *)realloc(d->actors, sizeof(span_t) * d->cap_actors);
          d->clock = (uint64_t *)realloc(d->clock, 8 * d->cap_actors);
          if (d->clock_order) d->clock_order = (uint32_t *)realloc(d->clock_order, 4 * d->cap_actors);
        }
        if (d->n_actors >= MAX_ACTORS) { rc = fail(&e, "unsupported: too many actors"); break; }
        ai = (int)d->n_actors++;
        d->actors[ai] = c->actors[0];
        d->clock[ai] = 0;
      }
      if (d->clock_order && d->clock[ai] == 0) {
        int seen = 0;
        for (uint32_t k = 0; k < d->n_clock; k++) seen |= d->clock_order[k] == (uint32_t)ai;
        if (!seen) d->clock_order[d->n_clock++] = (uint32_t)ai;
      }
      d->clock[ai] = c->seq;
      uint8_t *hcopy = (uint8_t *)pool_alloc(&d->pool, 32);  /* the decoded change lives only as long as this call */
      memcpy(hcopy, c->hash, 32);
      hset_add(&d->pool, &round_known, hcopy);
      call_applied[n_call_applied++] = hcopy;
      for (uint32_t k = 0; k < c->n_deps; k++) {
        int h = head_find(heads, n_heads, c->deps + 32 * k);
        if (h >= 0) heads[h] = NULL;
      }
      if (head_find(heads, n_heads, c->hash) < 0) heads[n_heads++] = hcopy;
      applied[na++] = c;
    }
    if (abandoned) {
      d->n_actors = snap_actors;
      d->n_clock = snap_clock_n;
      memcpy(d->clock, snap_clock, 8 * (size_t)snap_actors);
      memcpy(heads, snap_head_ptrs, sizeof(uint8_t *) * (size_t)snap_heads);
      n_heads = snap_heads;
      n_call_applied = snap_call;
      na = 0;
      memcpy(next_q, queue, sizeof(change_t) * qn);
      nq = qn;
    } else {
      for (uint32_t k = snap_call; k < n_call_applied; k++) hset_add(&d->pool, &d->known, call_applied[k]);
    }
    free(snap_clock);
    free(snap_head_ptrs);
    tab_free(&round_known);
    if (rc) break;
    if (na > 0) {
      any_applied = 1;
      /* every op of every accepted change, in order (new.js:1588-1591) */
      uint64_t n_cops = 0;
      dops_t *dops = (dops_t *)calloc(na, sizeof(dops_t));
      for (uint32_t k = 0; k < na && !rc; k++) {
        change_t *c = applied[k];
        if (c->n_actors > atab_cap) { atab_cap = c->n_actors * 2; atab = (uint32_t *)realloc(atab, atab_cap * 4); }
        for (uint32_t a = 0; a < c->n_actors && !rc; a++) {
          int idx = doc_actor_index(d, c->actors[a]);
          if (a == 0 && (uint32_t)idx + 1 > d->actors_read) d->actors_read = (uint32_t)idx + 1;
          if (idx < 0 || (uint32_t)idx >= d->actors_read) rc = fail(&e, "actorId is not known to document");
          else atab[a] = (uint32_t)idx;
        }
        if (rc) break;
        if ((rc = decode_ops(c, &dops[k], &e))) break;
        if (n_cops + dops[k].n_ops > cap_cops) {
          cap_cops = (n_cops + dops[k].n_ops) * 2 + 16;
          cops = (cop_t *)realloc(cops, sizeof(cop_t) * cap_cops);
        }
        for (uint64_t j = 0; j < dops[k].n_ops && !rc; j++) {
          const dop_t *op = &dops[k].ops[j];
          cop_t *co = &cops[n_cops++];
          memset(co, 0, sizeof *co);
          co->id.ctr = c->start_op + j;
          co->id.actor = atab[0];
          if (co->id.ctr >= ((uint64_t)1 << 44)) { rc = fail(&e, "unsupported: op counter too large"); break; }
          if (co->id.ctr > d->max_op) d->max_op = co->id.ctr;
          if (op->obj_ctr != NUL64) { co->obj.ctr = op->obj_ctr; co->obj.actor = atab[op->obj_actor]; }
          if (op->key_len != NUL32) { co->key_str = op->key; co->key_len = op->key_len; if (!co->key_str) co->key_str = (const uint8_t *)""; }
          else if (op->key_ctr == NUL64) { rc = fail(&e, "unsupported: operation without key"); break; }
          else if (op->key_ctr != 0) { co->key.ctr = op->key_ctr; co->key.actor = atab[op->key_actor]; }
          co->insert = op->insert; co->action = op->action; co->val_tag_len = op->val_tag_len; co->val = op->val;
          co->pred_num = op->pred_num;
          co->preds = (opid_t *)pool_alloc(&x.pc.pool, sizeof(opid_t) * (op->pred_num ? op->pred_num : 1));
          for (uint32_t p = 0; p < op->pred_num; p++) {
            co->preds[p].ctr = dops[k].pred_ctr[op->pred_first + p];
            co->preds[p].actor = atab[dops[k].pred_actor[op->pred_first + p]];
          }
          if (op->action == 3 && op->pred_num == 0) { rc = fail(&e, "unsupported: del operation without pred"); break; }
        }
        d->n_ops += dops[k].n_ops;
      }
      uint64_t pos = 0;
      while (!rc && pos < n_cops) rc = merge_call(&x, cops, n_cops, &pos);
      for (uint32_t k = 0; k < na; k++) dops_free(&dops[k]);
      free(dops);
      d->n_applied += na;
    }
    memcpy(queue, next_q, sizeof(change_t) * nq);
    qn = nq;
    if (na == 0) {
      /* a round that applies nothing ends the call -- unless the hash graph has not been rebuilt yet (new.js:1833-1840) */
      if (d->have_graph || qn == 0 || !d->doc_hashes) break;
      rebuild_hash_graph(d);
    }
  }
  if (!rc && d->loaded && !d->have_graph && qn > 0) rc = fail(&e, "unsupported: a change waits for a dependency the loaded document may hold (hash graph not rebuilt)");
  /* the changes of this call enter changeIndexByHash whatever happened to it in between (new.js:1846-1850) */
  if (!rc) {
    for (uint32_t k = 0; k < n_call_applied; k++) {
      if (!hset_has(&d->known, call_applied[k])) hset_add(&d->pool, &d->known, call_applied[k]);
      if (d->loaded) {
        if (d->n_since == d->cap_since) { d->cap_since = d->cap_since ? 2 * d->cap_since : 64; d->since = (const uint8_t **)realloc(d->since, sizeof(uint8_t *) * d->cap_since); }
        d->since[d->n_since++] = call_applied[k];
      }
    }
  }
  free(call_applied);
  if (!rc) rc = setup_patches(&x);

  if (!rc) {
    free(d->queue);
    d->queue = (qchange_t *)calloc(qn ? qn : 1, sizeof(qchange_t));
    for (uint32_t i = 0; i < qn; i++) d->queue[i].c = queue[i];
    d->n_queue = qn;
    d->n_pending = qn;
    if (any_applied) {
      uint8_t *nh = (uint8_t *)pool_alloc(&d->pool, 32 * (size_t)(n_heads ? n_heads : 1));
      uint32_t k = 0;
      for (uint32_t i = 0; i < n_heads; i++) if (heads[i]) memcpy(nh + 32 * k++, heads[i], 32);
      qsort(nh, k, 32, cmp_hash32);
      d->heads = nh;
      d->n_heads = k;
    }
    sbuf_t *b = &d->apply_json;
    b->len = 0;
    json_envelope(d, b);
    rc = json_pobj(&x.pc, b, &root);
    if (!rc) {
      if (is_local && n == 1) {
        /* new.js:1874-1877 */
        static const char hx[] = "0123456789abcdef";
        char t[48];
        sb_puts(b, ",\"actor\":\"");
        for (size_t i = 0; i < first_decoded.actors[0].len; i++) { sb_putc(b, hx[first_decoded.actors[0].p[i] >> 4]); sb_putc(b, hx[first_decoded.actors[0].p[i] & 15]); }
        snprintf(t, sizeof t, "\",\"seq\":%llu", (unsigned long long)first_decoded.seq);
        sb_puts(b, t);
      }
      sb_putc(b, '}');
    }
  }
  release_patches(&x);
  free(queue); free(next_q); free(applied); free(heads); free(atab); free(cops);
  if (rc) {
    if (errbuf && errcap) snprintf(errbuf, errcap, "%s", e.msg);
    return NULL;
  }
  if (len) *len = d->apply_json.len;
  return d->apply_json.p;
}
