"""Which launch of a unit's preparation is late?  From a rocprofv3 kernel trace of a bench step (see pipeline_timeline.py):
per preparation stream (= hardware queue) the kernels of each launch unit in order -- gate, events, tbuild, compose, publish,
gate, seg_fill -- with the time each one spent DISPATCHED BUT NOT STARTED (start - end of its predecessor in the queue) and
running.  Prints the units whose preparation took longest, and the swap / scoring kernels that overlapped them.
usage: python3 scripts/unit_latency.py TRACE_DIR [N]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r["Queue_Id"])))
rows.sort()
i0 = [i for i, r in enumerate(rows) if r[2] == "k_raw_stream"][-1]
seg = rows[i0:]
base = seg[0][0]
ms = lambda t: (t - base) / 1e6
queues = {}
for s, e, n, q in seg:
    queues.setdefault(q, []).append((s, e, n))
units = []
for q, ks in queues.items():
    if not any(n == "k_phi_events" for _, _, n in ks):
        continue
    for i, (s, e, n) in enumerate(ks):
        if n != "k_phi_events":
            continue
        u = {"queue": q, "prev_end": ks[i - 1][1] if i else s, "prev": ks[i - 1][2] if i else "-", "events": (s, e)}
        for s2, e2, n2 in ks[i + 1:i + 5]:
            if n2 in ("k_phi_tbuild", "k_phi_compose", "k_publish") and n2 not in u:
                u[n2] = (s2, e2)
        units.append(u)
units.sort(key=lambda u: u["events"][0])
for k, u in enumerate(units):
    u["no"] = k
def total(u):
    return (u.get("k_publish", u["events"])[1] - u["events"][0]) / 1e6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
big = [r for r in seg if r[2].startswith("void k_apply_swaps") or r[2].startswith("void k_moran_score") or r[2] in ("k_expand", "k_block_exact")]
print("preparation of a unit, events start -> publish end: median %.2f ms" % sorted(total(u) for u in units)[len(units) // 2])
for u in sorted(units, key=total, reverse=True)[:N]:
    print("unit #%d (queue %d): %.2f ms from %.2f" % (u["no"], u["queue"], total(u), ms(u["events"][0])))
    last = u["prev_end"]
    for name in ("events", "k_phi_tbuild", "k_phi_compose", "k_publish"):
        if name in u:
            s, e = u[name]
            print("    %-14s waited %6.2f ms behind its predecessor, ran %6.2f ms (%.2f .. %.2f)" % (name, (s - last) / 1e6, (e - s) / 1e6, ms(s), ms(e)))
            last = e
    a, b = u["events"][0], u.get("k_publish", u["events"])[1]
    for s, e, n, q in big:
        if s < b and e > a:
            print("      beside it: %-28s %.2f .. %.2f (queue %d)" % (n[:28], ms(s), ms(e), q))
# everything that started or ended within 0.6 ms of the end of the slowest unit's compose kernel, and what was running then
u = max(units, key=total)
if "k_phi_compose" in u:
    t = u["k_phi_compose"][1]
    print("around the end of that unit's k_phi_compose (%.2f):" % ms(t))
    for s, e, n, q in seg:
        if abs(s - t) < 6e5 or abs(e - t) < 6e5:
            print("    %8.2f .. %8.2f  queue %2d  %s" % (ms(s), ms(e), q, n[:50]))
    print("running across it:")
    for s, e, n, q in seg:
        if s < t - 6e5 and e > t + 6e5:
            print("    %8.2f .. %8.2f  queue %2d  %s" % (ms(s), ms(e), q, n[:50]))
