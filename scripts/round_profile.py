"""Per-round kernel durations from a rocprofv3 --kernel-trace CSV (development aid)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
kinds = {"wf_trav": "trav", "wf_hit": "hit", "wf_miss": "miss"}
seq = []
for r in rows:
    for k, v in kinds.items():
        if k in r["Kernel_Name"]:
            seq.append((v, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
# keep the last render only (the probe renders a warm-up frame first): split at big gaps
rounds = []
cur = {}
for v, s, e in seq:
    if v == "trav" and cur:
        rounds.append(cur); cur = {}
    cur[v] = (s, e)
if cur: rounds.append(cur)
n = len(rounds)
print("rounds", n)
tot = collections.Counter()
for i, r in enumerate(rounds):
    t0 = r["trav"][0]; t1 = max(v[1] for v in r.values())
    nxt = rounds[i + 1]["trav"][0] if i + 1 < n else t1
    d = {k: (v[1] - v[0]) / 1e3 for k, v in r.items()}
    gap = (nxt - t1) / 1e3
    for k, v in d.items(): tot[k] += v
    tot["gap"] += gap
    if i % max(1, n // 40) == 0 or i > n - 6:
        print(f"round {i:4d}: trav {d.get('trav',0):8.1f} hit {d.get('hit',0):8.1f} miss {d.get('miss',0):8.1f} gap {gap:7.1f} us")
print({k: round(v / 1e3, 2) for k, v in tot.items()}, "ms")
