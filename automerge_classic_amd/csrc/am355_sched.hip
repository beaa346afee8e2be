// General causal scheduler on the device.
//
// Reference: BackendDoc.applyChanges (backend/new.js:1797-1879) hands its queue -- the batch followed by the changes still waiting --
// to applyChanges (:1550-1597), which applies, in queue order, every change whose dependencies are all known (applied before the call
// or earlier in the same pass) and returns the rest; the loop at :1822-1841 repeats that until a pass applies nothing. A change is
// therefore applied in pass
//     pass(c) = max over its dependencies d of  pass(d) + [d stands behind c in the queue]          (0 without dependencies)
// and never, if a dependency is not in the queue or is never applied itself; later copies of a change are dropped once the first copy
// is known (:1566). The application order is (pass, position in the queue).
//
// The dependency hashes have been resolved to change indexes by k_deps_resolve (stream B), so this is a longest-path computation over
// integers: ks_pass relaxes it in ONE workgroup (the state of up to 8192 changes lives in registers and LDS: a batch is thousands of
// changes, and every round of a multi-workgroup version would cost a launch or a grid-wide barrier for microseconds of work); a
// change looks at its dependencies in order and stops at the first one whose pass is not known yet, so every dependency edge is
// visited once plus one LDS read per waiting change and sweep. Then: stable radix sort of the changes by pass (am355_prims.hip) =
// application order; ks_plan_sums / ks_plan_apply = the decode plans in that order (what k_actor_check / k_plan_apply build for the
// in-order path); ks_checks = heads and the actor rule (new.js:1442-1449), one wavefront per applied change.
// What is left to the host is what the in-order path leaves to it as well (plan_fast with an order, am355_replay.hip, beside the decode kernels):
// sequence numbers, clock, per-actor span tables -- O(changes) over 32-byte digests.
#include "am355_sched.h"
#include <algorithm>
#include "am355_prims.h"

namespace am355 {

static size_t al256(size_t b) { return carve_round(b); }

size_t sched_bytes(uint32_t n, uint32_t slot_mask) {
  size_t n1 = (size_t)n + 1;
  return 8 * al256(4 * n1) + 3 * al256(8 * n1) + 2 * al256(4 * n1) + al256(sort_workspace_bytes((uint32_t)n1)) + al256(4 * ((size_t)slot_mask + 1)) + al256(n1) + al256(4 * SW_NUM) +
         al256(8 * 8 * ((n1 + BLOCK - 1) / BLOCK + 1)) + 4096;
}

void sched_bind(SchedBufs& s, void* block, uint32_t n, uint32_t slot_mask) {
  canary_scope("general scheduler (sched_bind)");
  canary_forget(block, sched_bytes(n, slot_mask));
  uint8_t* p = (uint8_t*)block;
  size_t n1 = (size_t)n + 1;
  auto take = [&](size_t bytes) { void* r = p; canary_note(p, bytes); p += al256(bytes); return r; };
  s.n = n;
  s.pass = (uint32_t*)take(4 * n1); s.cursor = (uint32_t*)take(4 * n1); s.curmax = (uint32_t*)take(4 * n1);
  s.dfirst = (uint32_t*)take(4 * n1); s.dcnt = (uint32_t*)take(4 * n1); s.rank_of = (uint32_t*)take(4 * n1);
  s.apos = (uint32_t*)take(4 * n1); s.left = (uint32_t*)take(4 * n1); s.best = (unsigned long long*)take(8 * n1);
  s.key_a = (uint64_t*)take(8 * n1); s.key_b = (uint64_t*)take(8 * n1);
  s.val_a = (uint32_t*)take(4 * n1); s.val_b = (uint32_t*)take(4 * n1);
  s.sort_ws = take(sort_workspace_bytes((uint32_t)n1));
  s.first_rank = (uint32_t*)take(4 * ((size_t)slot_mask + 1));
  s.is_head = (uint8_t*)take(n1);
  s.words = (uint32_t*)take(4 * SW_NUM);
  s.block_sums = (unsigned long long*)take(8 * 8 * ((n1 + BLOCK - 1) / BLOCK + 1));
}

// ---------------------------------------------------------------------------------------------------------
// pass numbers
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t SP_THREADS = 1024;
constexpr uint32_t SP_PER = 6;
constexpr uint32_t SP_SMALL_MAX = SP_THREADS * SP_PER;  // changes whose state fits the LDS (six words each: 144 KiB of the CU's 160)

// Copies of one change (the same hash delivered several times) form a GROUP, named by its first copy F (k_deps_resolve gives every
// copy F in self_idx and resolves dependency hashes to F). Every copy is a node of its own: a copy standing behind the dependencies
// its first copy stands in front of is ready a pass EARLIER, and the reference applies whichever copy becomes ready first --
// (pass, position) minimal -- and drops the others as duplicates once that one is known (new.js:1566). G[F] = pass in which the
// group is applied (SCHED_UNSET until every copy knows its own), A[F] = position of the copy that is applied.
struct SchedGroups {
  volatile uint32_t* G;       // [n] at group heads
  volatile uint32_t* A;       // [n] at group heads (a change without copies: itself)
  uint32_t* left;             // [n] global: copies of the group still undecided (0: the change has no copies)
  unsigned long long* best;   // [n] global: min over the decided copies of (pass << 32 | position)
};

// copy c has decided: it would be applied in pass v (SCHED_NEVER: never)
// (alone: the change has no copies -- known from LDS, so that the common case touches no global memory: F == c, A[c] == c from the start)
__device__ __forceinline__ void sched_decide(const SchedGroups& g, uint32_t c, uint32_t F, uint32_t v, bool alone) {
  if (alone) { g.G[c] = v; return; }
  if (v != SCHED_NEVER) atomicMin(&g.best[F], (unsigned long long)v << 32 | c);
  __threadfence();
  if (atomicSub(&g.left[F], 1u) == 1u) {  // the last copy to decide publishes the group
    __threadfence();
    const unsigned long long b = atomicMin(&g.best[F], ~0ull);
    if (b == ~0ull) g.G[F] = SCHED_NEVER;
    else {
      g.A[F] = (uint32_t)b;
      __threadfence();
      g.G[F] = (uint32_t)(b >> 32);
    }
  }
}

// SMALL: n <= SP_SMALL_MAX -- group results, the copies' dependency ranges and the work list live in LDS (six words per change).
// Otherwise the same loop over global arrays.
//
// A sweep has two phases. CHECK, one lane per copy: a copy that has not decided sits on ONE dependency it knows to be undecided
// (wait_on); only when that one has decided is the copy put on the work list -- an LDS read per waiting copy and sweep. RESOLVE, one
// wavefront per listed copy, one lane per dependency: the dependency indexes arrive in one coalesced load, every lane looks its
// dependency's group up, a ballot finds a dependency that is still undecided (the copy then waits on that one) and a wave maximum
// gives the pass. In a log of synced rounds a sweep settles one round: 64 copies, four per wavefront.
// (The first version walked a copy's dependencies with ONE lane, four indexes per round trip: 8.4 ms for the shuffled headline log --
// 262 k dependent loads spread over the lanes of sixteen wavefronts -- against 3.4 ms for the host's walk; profiles/r04_b_*.)
template <bool SMALL>
__global__ __launch_bounds__(SP_THREADS) void ks_pass(const ChangeMeta* __restrict__ metas, uint32_t n, const uint32_t* __restrict__ dep_idx,
                                                      const uint32_t* __restrict__ self_idx, uint32_t* __restrict__ pass_g, uint32_t* __restrict__ apos_g,
                                                      uint32_t* __restrict__ g_wait, uint32_t* __restrict__ g_list, uint32_t* __restrict__ g_first,
                                                      uint32_t* __restrict__ g_cnt, uint32_t* __restrict__ g_left, unsigned long long* __restrict__ g_best,
                                                      uint64_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t* __restrict__ words, uint32_t max_sweeps) {
  wave_priority_high();
  constexpr uint32_t CAP = SMALL ? SP_SMALL_MAX : 1;
  __shared__ uint32_t s_G[CAP], s_A[CAP], s_first[CAP], s_cnt[CAP], s_wait[CAP], s_list[CAP];
  __shared__ uint32_t s_n_list, s_pending, s_progress, s_applied, s_maxpass;
  const uint32_t t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
  constexpr uint32_t N_WAVES = SP_THREADS / WAVE;
  // (pass_g / apos_g double as G / A of the global-memory form: the results are written over them at the end)
  SchedGroups g{SMALL ? (volatile uint32_t*)s_G : (volatile uint32_t*)pass_g, SMALL ? (volatile uint32_t*)s_A : (volatile uint32_t*)apos_g, g_left, g_best};
  uint32_t *const st_first = SMALL ? s_first : g_first, *const st_cnt = SMALL ? s_cnt : g_cnt, *const st_list = SMALL ? s_list : g_list;
  volatile uint32_t* const st_wait = SMALL ? (volatile uint32_t*)s_wait : (volatile uint32_t*)g_wait;  // a dependency known to be undecided; NONE32: the copy has decided
  if (t == 0) { s_applied = 0; s_maxpass = 0; }
  // ---- groups: every later copy counts itself at its first copy (g_left was cleared by the caller) ----
  for (uint32_t ci = t; ci < n; ci += SP_THREADS) {
    const uint32_t F = self_idx[ci] < n ? self_idx[ci] : ci;
    if (F != ci) atomicAdd(&g_left[F], 1u);
    g.G[ci] = SCHED_UNSET;
    g.A[ci] = ci;
    g_best[ci] = ~0ull;
    const ChangeMeta* m = &metas[ci];
    st_first[ci] = (uint32_t)((m->base + m->deps_off) >> 5);
    st_cnt[ci] = m->n_deps;
  }
  __threadfence();
  __syncthreads();
  constexpr uint32_t COPIES = 0x80000000u;   // in st_cnt: the change is one of several copies (its group is decided through global memory)
  for (uint32_t ci = t; ci < n; ci += SP_THREADS) {
    volatile uint32_t* left = g_left;
    const bool first = self_idx[ci] >= n || self_idx[ci] == ci;
    if (first && left[ci] != 0) g_left[ci] = left[ci] + 1;  // + the first copy itself
    if (!first || left[ci] != 0) st_cnt[ci] |= COPIES;
  }
  __threadfence();
  __syncthreads();
  // a copy without dependencies is applied in the first pass; every other copy starts out waiting on its first dependency (so the
  // first sweep does not resolve all n copies -- one coalesced load each, but sixteen wavefronts' worth at a time -- only to find
  // nearly all of them blocked)
  for (uint32_t ci = t; ci < n; ci += SP_THREADS) {
    const bool alone = !(st_cnt[ci] & COPIES);
    const uint32_t cnt = st_cnt[ci] & ~COPIES;
    const uint32_t d0 = cnt ? dep_idx[st_first[ci]] : NONE32;
    if (cnt == 0 || d0 >= n) {
      sched_decide(g, ci, alone ? ci : (self_idx[ci] < n ? self_idx[ci] : ci), cnt == 0 ? 0u : SCHED_NEVER, alone);
      st_wait[ci] = NONE32;
    } else st_wait[ci] = d0;
  }
  __threadfence();
  __syncthreads();
  uint32_t unfinished = 0;
  for (uint32_t sweep = 0;; sweep++) {
    if (t == 0) { s_n_list = 0; s_pending = 0; s_progress = 0; }
    __syncthreads();
    // ---- check: which undecided copies may have become ready ----
    uint32_t pending = 0;
    for (uint32_t ci = t; ci < n; ci += SP_THREADS) {
      const uint32_t w = st_wait[ci];
      if (w == NONE32) continue;
      pending = 1;
      if (g.G[w] != SCHED_UNSET) st_list[atomicAdd(&s_n_list, 1u)] = ci;
    }
    if (pending) s_pending = 1;
    if (!SMALL) __threadfence();
    __syncthreads();
    const uint32_t any_pending = s_pending, n_list = s_n_list;
    if (!any_pending) break;
    // ---- resolve: a wavefront per listed copy, a lane per dependency (the loads of a wavefront's next copy are under way while it
    //      works on this one: the index loads are what a sweep waits for) ----
    uint32_t progress = 0;
    constexpr uint32_t RES = 4;   // copies a wavefront has in flight: the index loads of four copies are one round trip, not four
    for (uint32_t w0 = wave; w0 < n_list; w0 += N_WAVES * RES) {
      uint32_t ci_[RES], first_[RES], cnt_[RES], d0_[RES];
#pragma unroll
      for (uint32_t j = 0; j < RES; j++) {
        const uint32_t w = w0 + j * N_WAVES;
        ci_[j] = w < n_list ? st_list[w] : NONE32;
        first_[j] = ci_[j] != NONE32 ? st_first[ci_[j]] : 0u;
        cnt_[j] = ci_[j] != NONE32 ? st_cnt[ci_[j]] : 0u;   // (bit 31: COPIES)
      }
#pragma unroll
      for (uint32_t j = 0; j < RES; j++) d0_[j] = lane < (cnt_[j] & ~COPIES) ? dep_idx[first_[j] + lane] : NONE32;
#pragma unroll
      for (uint32_t j = 0; j < RES; j++) {
        const uint32_t ci = ci_[j], first = first_[j], cnt = cnt_[j] & ~COPIES;
        const bool alone = !(cnt_[j] & COPIES);
        if (ci == NONE32) continue;
        uint32_t mx = 0, waits = NONE32;
        bool never = false;
        for (uint32_t k0 = 0; k0 < cnt && waits == NONE32 && !never; k0 += WAVE) {
          const uint32_t k = k0 + lane;
          uint32_t q = 0;
          bool unset = false, nev = false;
          uint32_t d = NONE32;
          if (k < cnt) {
            d = k0 == 0 ? d0_[j] : dep_idx[first + k];
            if (d >= n) nev = true;   // a dependency outside the queue
            else {
              const uint32_t p = g.G[d];
              if (p == SCHED_UNSET) unset = true;
              else if (p == SCHED_NEVER) nev = true;
              else q = p + (g.A[d] > ci ? 1u : 0u);   // applied behind this copy: the next pass at the earliest
            }
          }
          if (__ballot(nev)) never = true;
          const unsigned long long mu = __ballot(unset);
          if (mu && !never) waits = __shfl(d, (int)__ffsll(mu) - 1);
          for (int o = WAVE / 2; o >= 1; o >>= 1) {
            const uint32_t x = __shfl_xor(q, o);
            q = x > q ? x : q;
          }
          mx = q > mx ? q : mx;
        }
        if (lane == 0) {
          if (never || waits == NONE32) {
            sched_decide(g, ci, alone ? ci : (self_idx[ci] < n ? self_idx[ci] : ci), never ? SCHED_NEVER : mx, alone);
            st_wait[ci] = NONE32;
          } else st_wait[ci] = waits;
        }
        progress = 1;   // (decided, or waiting on another dependency now: either way something moved)
      }
    }
    if (progress && lane == 0) s_progress = 1;
    if (!SMALL) __threadfence();
    __syncthreads();
    const uint32_t any_progress = s_progress;
    __syncthreads();  // (thread 0 clears the words at the top of the next sweep: everybody has read them by then)
    // nothing on the list: what still waits, waits for itself (a dependency cycle takes a hash collision) -- never applied. Out of
    // sweeps: the host's scheduler takes over (SW_UNFINISHED)
    if (!any_progress || sweep + 1 >= max_sweeps) {
      if (any_progress) unfinished = 1;
      for (uint32_t ci = t; ci < n; ci += SP_THREADS)
        if (st_wait[ci] != NONE32) sched_decide(g, ci, self_idx[ci] < n ? self_idx[ci] : ci, SCHED_NEVER, !(st_cnt[ci] & COPIES));
      break;
    }
  }
  __threadfence();
  __syncthreads();
  // ---- results: the pass of every COPY (the applied copy of a group: the group's; every other copy: never), the position at which
  //      a group is applied, sort keys (a copy never applied sorts behind every pass), counts ----
  // (the global-memory form keeps G / A in pass_g / apos_g themselves: everybody reads the heads it needs first, then -- behind a
  // barrier -- writes)
  uint32_t applied = 0, maxp = 0;
  for (uint32_t ci = t; ci < n; ci += SP_THREADS) {
    const uint32_t F = self_idx[ci] < n ? self_idx[ci] : ci;
    const uint32_t gp = g.G[F];
    const uint32_t p = (gp != SCHED_NEVER && gp != SCHED_UNSET && g.A[F] == ci) ? gp : SCHED_NEVER;
    key[ci] = p == SCHED_NEVER ? (uint64_t)n : (uint64_t)p;
    val[ci] = ci;
    if (SMALL) apos_g[ci] = g.A[ci];
    if (p != SCHED_NEVER) { applied++; maxp = p > maxp ? p : maxp; }
  }
  __syncthreads();
  for (uint32_t ci = t; ci < n; ci += SP_THREADS) pass_g[ci] = key[ci] == (uint64_t)n ? SCHED_NEVER : (uint32_t)key[ci];
  if (applied) { atomicAdd(&s_applied, applied); atomicMax(&s_maxpass, maxp); }
  __syncthreads();
  if (t == 0) {
    words[SW_N_APPLIED] = s_applied;
    words[SW_MAX_PASS] = s_maxpass;
    words[SW_UNFINISHED] = unfinished;
  }
}

// ---------------------------------------------------------------------------------------------------------
// plans in application order
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t SCHED_SUMS = 8;  // words per workgroup: ops, preds, entries, small, large, serial plans

// exclusive scans of three values over the workgroup (ex) and their totals, from wave prefix sums
__device__ __forceinline__ void sched_scan3(unsigned long long a, unsigned long long b, unsigned long long c, unsigned long long (*s)[3], unsigned long long ex[3],
                                            unsigned long long tot[3]) {
  const uint32_t lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  unsigned long long v[3] = {a, b, c}, inc[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    unsigned long long x = v[q];
    for (int d = 1; d < WAVE; d <<= 1) {
      unsigned long long o = __shfl_up(x, (unsigned)d);
      if ((int)lane >= d) x += o;
    }
    inc[q] = x;
  }
  __syncthreads();  // (s may still be read by the previous call)
  if (lane == WAVE - 1) { s[w][0] = inc[0]; s[w][1] = inc[1]; s[w][2] = inc[2]; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; q++) {
    unsigned long long before = 0, all = 0;
    for (uint32_t j = 0; j < BLOCK / WAVE; j++) {
      if (j < w) before += s[j][q];
      all += s[j][q];
    }
    ex[q] = before + inc[q] - v[q];
    tot[q] = all;
  }
}

struct SchedItem {
  ChangeBrief br;
  uint32_t ci;
  bool applied, small, large, serial;
};
__device__ __forceinline__ SchedItem sched_item(uint32_t t, uint32_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ pass,
                                                const ChangeBrief* __restrict__ briefs) {
  SchedItem it{};
  it.ci = NONE32;
  if (t < n) {
    it.ci = order[t];
    it.applied = pass[it.ci] != SCHED_NEVER;
    if (it.applied) it.br = briefs[it.ci];
  }
  const bool has = it.applied && it.br.n_ops != 0;
  it.small = has && (it.br.flags_fits & 0x40000000u);
  it.large = has && !it.small && (it.br.flags_fits & 0x80000000u);
  it.serial = has && !it.small && !it.large;
  return it;
}

// per workgroup of BLOCK application ranks: sums for ks_plan_apply; per applied change its rank, the first rank of its author, and
// the head mark ks_checks clears again for every change somebody depends on
__global__ __launch_bounds__(BLOCK) void ks_plan_sums(const ChangeBrief* __restrict__ briefs, uint32_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ pass,
                                                      uint32_t slot_mask, uint32_t* __restrict__ rank_of, uint32_t* __restrict__ first_rank, uint8_t* __restrict__ is_head,
                                                      unsigned long long* __restrict__ block_sums, uint32_t* __restrict__ words) {
  wave_priority_high();
  __shared__ unsigned long long s_scan[BLOCK / WAVE][3];
  const uint32_t t = gtid();
  const SchedItem it = sched_item(t, n, order, pass, briefs);
  if (t < n) {
    rank_of[it.ci] = it.applied ? t : NONE32;
    is_head[it.ci] = it.applied ? 1 : 0;
    if (it.applied && it.br.author_slot <= slot_mask) atomicMin(&first_rank[it.br.author_slot], t);
  }
  unsigned long long ex[3], t1[3], t2[3];
  sched_scan3(it.applied ? it.br.n_ops : 0u, it.applied ? it.br.n_preds : 0u, it.applied ? it.br.n_entries : 0u, s_scan, ex, t1);
  sched_scan3(it.small ? 1u : 0u, it.large ? 1u : 0u, it.serial ? 1u : 0u, s_scan, ex, t2);
  uint32_t mx_op = (it.applied && it.br.n_ops) ? it.br.start_op + it.br.n_ops - 1 : 0u;
  for (int d = WAVE / 2; d >= 1; d >>= 1) {
    uint32_t o = __shfl_xor(mx_op, d);
    mx_op = o > mx_op ? o : mx_op;
  }
  if ((threadIdx.x & (WAVE - 1)) == 0 && mx_op) atomicMax(&words[SW_MAX_OP], mx_op);
  if (threadIdx.x == 0) {
    unsigned long long* out = block_sums + (size_t)blockIdx.x * SCHED_SUMS;
    out[0] = t1[0]; out[1] = t1[1]; out[2] = t1[2]; out[3] = t2[0]; out[4] = t2[1]; out[5] = t2[2];
  }
}

// one wavefront per application rank: the changes it depends on are no heads (new.js:1582-1583); every actor its table mentions must
// be the author of a change applied no later than itself -- the reference reads the applied changes in application order and a
// change may only name actors the document knows by then (new.js:1442-1449)
__global__ __launch_bounds__(WAVE) void ks_checks(const ChangeMeta* __restrict__ metas, uint32_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ pass,
                                                  const uint32_t* __restrict__ dep_idx, const uint32_t* __restrict__ amap, const uint32_t* __restrict__ amap_base,
                                                  uint32_t amap_cap, const uint32_t* __restrict__ first_rank, const uint32_t* __restrict__ apos, uint8_t* __restrict__ is_head,
                                                  uint32_t* __restrict__ words) {
  wave_priority_high();
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  if (t >= n) return;
  const uint32_t ci = order[t];
  if (pass[ci] == SCHED_NEVER) return;
  const ChangeMeta* m = &metas[ci];
  const uint32_t dfirst = (uint32_t)((m->base + m->deps_off) >> 5), nd = m->n_deps;
  for (uint32_t k = lane; k < nd; k += WAVE) {
    const uint32_t d = dep_idx[dfirst + k];  // (the first copy of the dependency: the copy that was applied may be another)
    if (d < n) is_head[apos[d]] = 0;
  }
  const uint32_t base = amap_base[ci], ne = m->n_entries;
  uint32_t late = 0;
  if ((uint64_t)base + ne <= amap_cap)
    for (uint32_t k = lane; k < ne; k += WAVE)
      if (first_rank[amap[base + k]] > t) late = 1;  // (NONE32: no applied change by that actor at all)
  if (late) atomicOr(&words[SW_FLAGS], (uint32_t)F_UNKNOWN_ACTOR_DEV);
}

// prefix sums in application order -> the ChangePlan of every applied change with ops, by decoder class (as k_plan_apply); the last
// workgroup reports the totals and the scheduler's words through HostSignals
__global__ __launch_bounds__(BLOCK) void ks_plan_apply(const ChangeBrief* __restrict__ briefs, uint32_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ pass,
                                                       const uint32_t* __restrict__ amap_base, const uint32_t* __restrict__ slot_rank, uint32_t slot_mask,
                                                       const unsigned long long* __restrict__ block_sums, ChangePlan* __restrict__ plans, ChangePlan* __restrict__ plans_serial,
                                                       const uint32_t* __restrict__ stage_words, const uint32_t* __restrict__ words, const uint32_t* __restrict__ distinct,
                                                       HostSignals* sig, uint32_t seq) {
  wave_priority_high();
  __shared__ unsigned long long s_scan[BLOCK / WAVE][3];
  __shared__ unsigned long long s_base[6];
  {
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += BLOCK) {
      const unsigned long long* q = block_sums + (size_t)j * SCHED_SUMS;
      p0 += q[0]; p1 += q[1]; p2 += q[2]; p3 += q[3]; p4 += q[4]; p5 += q[5];
    }
    unsigned long long ex[3], t1[3], t2[3];
    sched_scan3(p0, p1, p2, s_scan, ex, t1);
    sched_scan3(p3, p4, p5, s_scan, ex, t2);
    if (threadIdx.x == 0) { s_base[0] = t1[0]; s_base[1] = t1[1]; s_base[2] = t1[2]; s_base[3] = t2[0]; s_base[4] = t2[1]; s_base[5] = t2[2]; }
    __syncthreads();
  }
  const uint32_t t = gtid();
  const SchedItem it = sched_item(t, n, order, pass, briefs);
  unsigned long long e1[3], e2[3], t1[3], t2[3];
  sched_scan3(it.applied ? it.br.n_ops : 0u, it.applied ? it.br.n_preds : 0u, it.applied ? it.br.n_entries : 0u, s_scan, e1, t1);
  sched_scan3(it.small ? 1u : 0u, it.large ? 1u : 0u, it.serial ? 1u : 0u, s_scan, e2, t2);
  if (it.small || it.large || it.serial) {
    // (the actor table of a change stays where k_actor_intern put it: at the prefix sum of the table sizes in INPUT order)
    ChangePlan pl{it.ci, (uint32_t)(s_base[0] + e1[0]), (uint32_t)(s_base[1] + e1[1]), amap_base[it.ci],
                  it.br.author_slot <= slot_mask ? slot_rank[it.br.author_slot] : 0u, it.br.n_entries};
    if (it.small) plans[(uint32_t)(s_base[3] + e2[0])] = pl;
    else if (it.large) plans[n - 1 - (uint32_t)(s_base[4] + e2[1])] = pl;
    else plans_serial[(uint32_t)(s_base[5] + e2[2])] = pl;
  }
  if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) {
    unsigned long long ops = s_base[0] + t1[0], preds = s_base[1] + t1[1], ent = s_base[2] + t1[2];
    PlanTotals z{};
    z.n_ops = (uint32_t)ops; z.n_preds = (uint32_t)preds; z.n_entries = (uint32_t)ent;
    z.n_small = (uint32_t)(s_base[3] + t2[0]); z.n_large = (uint32_t)(s_base[4] + t2[1]); z.n_serial = (uint32_t)(s_base[5] + t2[2]);
    z.max_op = words[SW_MAX_OP];
    z.fallback = (ops >= 0x7ffffff0ull || preds >= 0xfffffff0ull || ent >= 0xfffffff0ull) ? 1u : 0u;
    z.flags_a = words[SW_FLAGS]; z.fast_a = stage_words[1]; z.total_entries = stage_words[2]; z.n_distinct = distinct[0];
    z.reserved[1] = words[SW_N_APPLIED]; z.reserved[2] = words[SW_MAX_PASS]; z.reserved[3] = words[SW_UNFINISHED];
    signal_host((uint32_t*)&sig->plan, (const uint32_t*)&z, sizeof(PlanTotals) / 4, &sig->plan_seq, seq);
  }
}

static int sched_bits_for(uint64_t v) {
  int b = 1;
  while (b < 64 && (v >> b)) b++;
  return b;
}

void launch_sched_general(const ChangeMeta* metas, const ChangeBrief* briefs, uint32_t n, const uint32_t* dep_idx, const uint32_t* self_idx,
                          const uint32_t* amap, const uint32_t* amap_base, uint32_t amap_cap, const uint32_t* slot_rank, uint32_t slot_mask, SchedBufs& s,
                          uint32_t** order_out, ChangePlan* plans, ChangePlan* plans_serial, const uint32_t* stage_words, const uint32_t* distinct, HostSignals* sig,
                          uint32_t seq, hipStream_t st) {
  // (read per call: the tests switch them inside one process)
  const char* e_sweeps = getenv("AM355_SCHED_SWEEPS");
  // A sweep settles one dependency level and looks at all n changes, in ONE workgroup: a chain of 10^5 changes delivered in reverse
  // order would hold that workgroup for n sweeps x n changes (~10 s). The sweeps are bounded by the work they may cost (n x sweeps
  // <= 2^32 change visits, a few tens of milliseconds); beyond it SW_UNFINISHED hands the batch to the host's scheduler, which walks
  // such a chain in O(n).
  const uint32_t sweeps_by_work = (uint32_t)std::min<uint64_t>(1u << 16, std::max<uint64_t>(256, (1ull << 32) / std::max<uint32_t>(n, 1u)));
  const uint32_t max_sweeps = e_sweeps && atoi(e_sweeps) > 0 ? (uint32_t)atoi(e_sweeps) : sweeps_by_work;
  const bool force_big = getenv("AM355_SCHED_BIG") != nullptr;  // (tests: the global-memory variant on small batches too)
  (void)hipMemsetAsync(s.words, 0, 4 * SW_NUM, st);
  (void)hipMemsetAsync(s.first_rank, 0xff, 4 * ((size_t)slot_mask + 1), st);
  (void)hipMemsetAsync(s.left, 0, 4 * (size_t)n, st);
  if (n <= SP_SMALL_MAX && !force_big)
    hipLaunchKernelGGL(ks_pass<true>, dim3(1), dim3(SP_THREADS), 0, st, metas, n, dep_idx, self_idx, s.pass, s.apos, s.cursor, s.curmax, s.dfirst, s.dcnt, s.left, s.best, s.key_a, s.val_a, s.words,
                       max_sweeps);
  else
    hipLaunchKernelGGL(ks_pass<false>, dim3(1), dim3(SP_THREADS), 0, st, metas, n, dep_idx, self_idx, s.pass, s.apos, s.cursor, s.curmax, s.dfirst, s.dcnt, s.left, s.best, s.key_a, s.val_a, s.words,
                       max_sweeps);
  // application order = (pass, position): a STABLE sort by pass of the changes in queue order
  int res = radix_sort_pairs(s.key_a, s.val_a, s.key_b, s.val_b, n, 0, sched_bits_for(n), s.sort_ws, st);
  uint32_t* order = res ? s.val_b : s.val_a;
  *order_out = order;
  const dim3 grid((n + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL(ks_plan_sums, grid, dim3(BLOCK), 0, st, briefs, n, (const uint32_t*)order, (const uint32_t*)s.pass, slot_mask, s.rank_of, s.first_rank, s.is_head,
                     s.block_sums, s.words);
  hipLaunchKernelGGL(ks_checks, dim3(n), dim3(WAVE), 0, st, metas, n, (const uint32_t*)order, (const uint32_t*)s.pass, dep_idx, amap, amap_base, amap_cap,
                     (const uint32_t*)s.first_rank, (const uint32_t*)s.apos, s.is_head, s.words);
  hipLaunchKernelGGL(ks_plan_apply, grid, dim3(BLOCK), 0, st, briefs, n, (const uint32_t*)order, (const uint32_t*)s.pass, amap_base, slot_rank, slot_mask,
                     (const unsigned long long*)s.block_sums, plans, plans_serial, stage_words, (const uint32_t*)s.words, distinct, sig, seq);
}

}  // namespace am355
