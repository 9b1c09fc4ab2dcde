"""Timeline of one pipelined bench step from a rocprofv3 kernel trace.

usage (GPU box): rocprofv3 --kernel-trace -d gpurun_out/tr --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
                 python3 scripts/pipeline_timeline.py gpurun_out/tr
Prints busy time / span / gaps per kernel family of the LAST step (generator chain, preparation, swaps, scoring),
how much chain idle time is due to late preparation, and the kernels of the fill and tail phases."""
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
t0=rows[0][0]
# take the last bench step: find last k_raw_stream
starts=[i for i,r in enumerate(rows) if r[2]=="k_raw_stream"]
i0=starts[-1]; seg=rows[i0:]
base=seg[0][0]
def summarize(name):
    ks=[r for r in seg if r[2]==name]
    if not ks: return
    dur=sum(e-s for s,e,_ in ks)/1e6
    span=(ks[-1][1]-ks[0][0])/1e6
    gaps=sum(max(0,ks[i+1][0]-ks[i][1]) for i in range(len(ks)-1))/1e6
    print(f"{name:18s} n={len(ks):4d} busy {dur:7.1f} ms  span {span:7.1f} ms  gaps {gaps:7.1f} ms  first start {(ks[0][0]-base)/1e6:6.1f}  last end {(ks[-1][1]-base)/1e6:6.1f}")
names = sorted({r[2] for r in seg})
def family(prefix):
    return [n for n in names if n.startswith(prefix) or n.startswith("void " + prefix)]
for n in ["k_chain", "k_phi_events", "k_phi_tbuild", "k_phi_compose", "k_seg_fill", "k_gate", "k_publish", "k_block_exact", "k_expand"] + family("k_apply_swaps") + family("k_moran_score") + family("k_moran_finalize"):
    summarize(n)
# chain gap analysis: for each chain kernel, when did its prep (tbuild with same index) end?
ch=[r for r in seg if r[2]=="k_chain"]; tb=[r for r in seg if r[2]=="k_phi_tbuild"]; evk=[r for r in seg if r[2]=="k_phi_events"]
wait_prep=0; n=min(len(ch),len(tb)) if len(ch) == len(tb) else 0   # (r01 / early r02: one chain launch per unit)
for i in range(n):
    if i>0:
        gap=ch[i][0]-ch[i-1][1]
        late=tb[i][1]-ch[i-1][1]
        if late>0: wait_prep+=min(gap,late)
if n: print(f"chain idle attributable to late preparation: {wait_prep/1e6:.1f} ms")
print("chain launches: " + ", ".join(f"{(s_-base)/1e6:.1f}-{(e_-base)/1e6:.1f}" for s_, e_, _ in ch))
print("events kernel avg %.0f us, tbuild avg %.0f us, chain avg %.0f us" % (sum(e-s for s,e,_ in evk)/len(evk)/1e3, sum(e-s for s,e,_ in tb)/len(tb)/1e3, sum(e-s for s,e,_ in ch)/len(ch)/1e3))
cmp_ = [r for r in seg if r[2] == "k_phi_compose"]
if cmp_ and len(cmp_) == len(evk):   # a unit's preparation: from the start of its events kernel to the end of its compose kernel
    lat = [(c[1] - e[0]) / 1e3 for e, c in zip(sorted(evk), sorted(cmp_))]
    print("compose kernel avg %.0f us; preparation of a unit (events start -> compose end) avg %.0f us, max %.0f us" % (sum(e-s for s,e,_ in cmp_)/len(cmp_)/1e3, sum(lat)/len(lat), max(lat)))
for fam in ("k_apply_swaps", "k_moran_score", "k_block_exact", "k_expand"):
    for nme in family(fam) or ([fam] if fam in names else []):
        ks = [r for r in seg if r[2] == nme]
        print(f"{nme[:34]} launches: " + ", ".join(f"{(s_-base)/1e6:.1f}-{(e_-base)/1e6:.1f}" for s_, e_, _ in ks))
print("--- kernels other than generator in the first 100 ms and the last 80 ms of the step")
end=seg[-1][1]
for s_,e_,n_ in seg:
    if n_ in ("k_chain","k_phi_events","k_phi_tbuild","k_phi_compose","k_seg_fill","k_gate","k_publish"): continue
    if (s_-base)/1e6 < 45 or (end-s_)/1e6 < 30:
        print(f"{(s_-base)/1e6:8.2f} -> {(e_-base)/1e6:8.2f}  {n_[:40]}")
