"""At bench size: the seeded pipeline with the uint8 / uint16 / fp64 sources against each other, listing where the
per-permutation statistics differ by more than rounding (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from spatialcore_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 500
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
coords, X = bench.synth_inputs(N, G, 0)
ctx = _lib.Context(0)
ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1.0 / 15)
runs = {}
for tag, bits in (("u8", 8), ("u8 again", 8), ("u16", 16), ("u16 again", 16)):
    ctx.set_moran_source_bits(bits)
    ctx.set_expression(X, np.arange(G))
    w = _lib.rng_state_words(np.random.default_rng(0))
    out = ctx.moran_seeded(w, P)
    runs[tag] = (out["sims"].copy(), out["count_ge"].copy(), ctx.moran_source_bits(), ctx.permgen_stats(), out["I"].copy())
    print(tag, "bits", runs[tag][2], "permgen", runs[tag][3], flush=True)
base = runs["u16"][0]
for tag, (sims, cnt, bits, _, I) in runs.items():
    err = np.abs(sims - base) / (np.abs(base) + 1e-300)
    scale = np.abs(base).std(axis=0)
    abs_err = np.abs(sims - base) / scale            # in units of the gene's own spread
    bad = np.argwhere(abs_err > 1e-9)
    print(f"{tag}: max |diff| / sd(sims of the gene) = {abs_err.max():.3e}; {len(bad)} entries above 1e-9; "
          f"count_ge differs from u16 for {int((cnt != runs['u16'][1]).sum())} genes")
    if len(bad):
        print("   permutations:", np.unique(bad[:, 0])[:20].tolist(), " genes:", np.unique(bad[:, 1])[:40].tolist())
        for p_, g_ in bad[:8]:
            print(f"   perm {p_} gene {g_}: {sims[p_, g_]:.17g} vs {base[p_, g_]:.17g}")
    host = (sims >= I[None, :]).sum(axis=0)
    print(f"   I equal to u16's: {np.array_equal(I, runs['u16'][4], equal_nan=True)}; device count == host recount of the returned sims: "
          f"{int((host != cnt).sum())} genes differ")
    for g_ in np.flatnonzero(cnt != runs['u16'][1])[:6]:
        near = np.sort(np.abs(sims[:, g_] - I[g_]))[:2]
        print(f"   gene {g_}: count {cnt[g_]} vs {runs['u16'][1][g_]}, host recount {host[g_]}, I {I[g_]:.17g}, nearest |sim - I| {near.tolist()}")
sims, cnt, _, _, I = runs["u16"]
gap = np.abs(sims - I[None, :])
pm = gap.argmin(axis=0)
rel = gap.min(axis=0) / np.abs(sims).std(axis=0)
tied = np.flatnonzero(rel < 1e-9)
print(f"{tied.size} of {G} genes have a permutation whose statistic equals the observed one up to rounding; permutation indices: "
      f"{np.unique(pm[tied]).tolist()[:30]}")
print("   genes:", tied[:60].tolist())
print("   signs of (sim - I) u16:", np.sign(sims[pm[tied], tied] - I[tied])[:40].astype(int).tolist())
s8 = runs["u8"][0]
print("   signs of (sim - I) u8 :", np.sign(s8[pm[tied], tied] - I[tied])[:40].astype(int).tolist())
c8, c16 = runs["u8"][1], runs["u16"][1]
d = np.flatnonzero(c8 != c16)
print("genes whose counts differ:", d.tolist(), " all among the tied ones:", bool(np.isin(d, tied).all()))
g_ = G - 1
print(f"gene {g_}: counts {c8[g_]} / {c16[g_]}, nearest |sim - I| / sd = {rel[g_]:.3e} at permutation {pm[g_]}, I {I[g_]:.17g}")
