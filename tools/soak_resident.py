"""Randomised sessions of am355_apply_changes onto a kept state against the oracle session, on the GPU (or, with AM355_TOOL_LIB, on the
CPU emulation): text typed by one author, concurrent text with deletions over one or more objects, maps with conflicts; batches of
random sizes (1 .. 40 changes), the in-place list merge in random chunk sizes, AM355_RESORDER_VERIFY on. Every incremental patch, the
whole-document patch every few calls and at the end, and Backend.save against a bulk replay are compared.
  python tools/soak_resident.py <first seed> <sessions>"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["AM355_RESORDER_VERIFY"] = "1"
from automerge_classic_amd import engine, loggen  # noqa: E402
from automerge_classic_amd.loggen import ChangeLog  # noqa: E402
import oracle_lib  # noqa: E402
from test_apply_engine import _ordered  # noqa: E402
from test_apply_vectors import same_patch  # noqa: E402

LIB = os.environ.get("AM355_TOOL_LIB")
first, count = int(sys.argv[1]), int(sys.argv[2])
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"   # long concurrent-text logs delivered in batches of up to 144 changes: several chunks of the in-place merge, tens of thousands of edit items
totals = {"sessions": 0, "calls": 0, "served": 0, "fell_back": 0, "in_place": 0, "maps_only": 0, "refused": 0}
for seed in range(first, first + count):
    rnd = random.Random(seed)
    kind = rnd.choice(["typing", "concurrent", "concurrent", "concurrent_small", "map", "mixed", "mixed"])
    if BIG:
        kind = "big"
    if kind == "big":
        log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=rnd.randint(8, 24), n_rounds=rnd.randint(6, 14), ins_per_change=rnd.randint(40, 320),
                              del_per_change=rnd.randint(0, 60), n_objects=rnd.randint(1, 3), seed=seed)
    elif kind == "typing":
        log = loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=rnd.randint(200, 3000), ops_per_change=rnd.randint(1, 60), seed=seed)
    elif kind == "concurrent":
        log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=rnd.randint(2, 12), n_rounds=rnd.randint(3, 8), ins_per_change=rnd.randint(5, 120),
                              del_per_change=rnd.randint(0, 30), n_objects=rnd.randint(1, 3), seed=seed)
    elif kind == "concurrent_small":
        log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=rnd.randint(2, 6), n_rounds=rnd.randint(4, 12), ins_per_change=rnd.randint(1, 4),
                              del_per_change=rnd.randint(0, 2), n_objects=rnd.randint(1, 2), seed=seed)
    elif kind == "mixed":
        # a Text (or several) AND root-map keys in one document: two logs side by side, their changes interleaved at random (each log's own
        # order kept) -- batches of map rows only (the map half of the merge alone), of list rows only (in place), of both (the whole merge)
        log = None
        la = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=rnd.randint(2, 6), n_rounds=rnd.randint(3, 7), ins_per_change=rnd.randint(3, 60),
                             del_per_change=rnd.randint(0, 10), n_objects=rnd.randint(1, 2), seed=seed)
        lb = loggen.generate(loggen.KIND_MAP_LWW, n_actors=rnd.randint(2, 4), n_rounds=rnd.randint(4, 12), n_keys=rnd.randint(5, 40), seed=seed + 500000)
    else:
        log = loggen.generate(loggen.KIND_MAP_LWW, n_actors=rnd.randint(2, 6), n_rounds=rnd.randint(3, 8), n_keys=rnd.randint(5, 60), seed=seed)
    os.environ.pop("AM355_RESORDER_CHUNK", None)
    if rnd.random() < 0.4:
        os.environ["AM355_RESORDER_CHUNK"] = str(rnd.choice([3, 7, 50, 400]) if not BIG else rnd.choice([700, 2000, 5000]))
    def changes_of(lg):
        arena, offs = bytes(lg.arena), [int(x) for x in lg.offsets]
        return [arena[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    if kind == "mixed":
        ca, cb = changes_of(la), changes_of(lb)
        ch, ia, ib = [], 0, 0
        run_a = True
        while ia < len(ca) or ib < len(cb):   # stretches of one log, then of the other
            take = rnd.randint(1, 6)
            if run_a and ia < len(ca): ch += ca[ia:ia + take]; ia += take
            elif ib < len(cb): ch += cb[ib:ib + take]; ib += take
            else: ch += ca[ia:ia + take]; ia += take
            run_a = not run_a
    else:
        ch = changes_of(log)
    k = max(1, rnd.randint(1, max(1, len(ch) // 2)))
    batches = [ch[:k]]
    while k < len(ch):
        size = rnd.choice([1, 1, 1, 2, 3, 5, 8, 13, 40]) if not BIG else rnd.choice([1, 7, 25, 40, 64, 96, 120, 144])
        batches.append(ch[k:k + size])
        k += size
    # deliveries the resident path must hand to the full replay AFTER it has enqueued the batch's device work (the hash-dependent checks run
    # behind it): an applied change again inside a batch, two batches in the wrong order (the first one's changes wait in the queue)
    perturbed = False
    if rnd.random() < 0.3 and len(batches) > 3:
        perturbed = True
        j = rnd.randint(2, len(batches) - 1)
        done = [x for bb in batches[:j] for x in bb]
        batches[j] = list(batches[j])
        batches[j].insert(rnd.randint(0, len(batches[j])), rnd.choice(done))
    if rnd.random() < 0.2 and len(batches) > 4:
        perturbed = True
        j = rnd.randint(1, len(batches) - 2)
        batches[j], batches[j + 1] = batches[j + 1], batches[j]
    eng = engine.Engine(0, LIB) if LIB else engine.Engine(0)
    session = oracle_lib.OracleSession()
    try:
        for i, batch in enumerate(batches):
            want = session.apply(batch)
            try:
                eng.apply_changes(ChangeLog.from_changes(batch))
            except engine.UnsupportedChanges:
                totals["refused"] += 1
                break
            got = eng.apply_patch_json()
            assert same_patch(got, want), f"seed {seed} ({kind}) batch {i}:\n{got[:2000]}\n{want[:2000]}"
            totals["calls"] += 1
            if rnd.random() < 0.15:
                assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"seed {seed} ({kind}) getPatch after batch {i}"
        else:
            assert dict(_ordered(eng.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"seed {seed} ({kind}) final getPatch"
            doc = bytes(eng.save())
            if not perturbed and log is not None:   # (a document holds its changes in application order: the generator's order only when delivered in it)
                bulk = engine.Engine(0, LIB) if LIB else engine.Engine(0)
                bulk.load_changes(log)
                bulk.replay()
                assert doc == bytes(bulk.save()), f"seed {seed} ({kind}) save differs from the bulk replay's"
                bulk.close()
            else:   # (the saved document loads into the same whole-document patch)
                back = oracle_lib.OracleSession(doc)
                assert dict(_ordered(back.patch_json()))["diffs"] == dict(_ordered(session.patch_json()))["diffs"], f"seed {seed} ({kind}) saved document"
                back.close()
        s, f, p = eng.resident_counters()
        totals["served"] += s; totals["fell_back"] += f; totals["in_place"] += p
        totals["maps_only"] += eng.resident_maps_only_calls()
        totals["sessions"] += 1
    finally:
        eng.close()
print("soak_resident seeds %d..%d:" % (first, first + count - 1), totals)

# ---- the committed applyChanges campaigns (nested maps, tables, lists with assigned elements, counters in lists, wide conflicts), their
#      changes given in the recorded order but in batches of OTHER sizes than the recorded calls: engine == oracle session per batch ----
if len(sys.argv) > 3 and sys.argv[3] == "campaigns":
    from test_apply_engine import load_campaign  # noqa: E402
    ct = {"sessions": 0, "calls": 0, "served": 0, "in_place": 0, "refused": 0, "oracle_rejects": 0}
    for fixture in ("apply_campaign.json.gz", "apply_campaign_lists.json.gz", "apply_campaign_quirks.json.gz"):
        sessions, pool = load_campaign(fixture)
        for si, s in enumerate(sessions):
            if "doc" in s:
                continue
            for variant in range(2):
                rnd = random.Random(first * 1000003 + si * 7 + variant)
                ch = [pool[k] for call in s["calls"] for k in call]
                batches, k = [], 0
                while k < len(ch):
                    size = rnd.choice([1, 1, 2, 3, 5])
                    batches.append(ch[k:k + size])
                    k += size
                eng = engine.Engine(0, LIB) if LIB else engine.Engine(0)
                session = oracle_lib.OracleSession()
                try:
                    for i, batch in enumerate(batches):
                        try:
                            want = session.apply(batch)
                        except oracle_lib.OracleError:
                            ct["oracle_rejects"] += 1
                            break
                        try:
                            eng.apply_changes(ChangeLog.from_changes(batch))
                        except engine.UnsupportedChanges:
                            ct["refused"] += 1
                            break
                        got = eng.apply_patch_json()
                        assert same_patch(got, want), f"{fixture} session {s['name']} variant {variant} batch {i}:\n{got[:2000]}\n{want[:2000]}"
                        ct["calls"] += 1
                    a, _, c = eng.resident_counters()
                    ct["served"] += a; ct["in_place"] += c; ct["sessions"] += 1
                finally:
                    eng.close()
    print("soak_resident campaigns re-split:", ct)
