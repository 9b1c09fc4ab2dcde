import numpy as np, sys
a = {t: np.load(f"gpurun_out/conc_{t}.npz") for t in sys.argv[1:]}
ref = a[sys.argv[1]]["two_sims"]
for t, z in a.items():
    d = z["sims"] != ref
    print(t, "seeded reps vs solo two-step: differing entries", d.reshape(3, -1).sum(1), "of", ref.size,
          "two-step vs ref", int((z["two_sims"] != ref).sum()),
          "max rel", float(np.max(np.abs(z["sims"] - ref) / (np.abs(ref) + 1e-300))))
    if d.any():
        r, p, g = np.argwhere(d)[0]
        print("  first diff rep", r, "perm", p, "gene", g, z["sims"][r, p, g], ref[p, g], "perms with diffs:", np.unique(np.argwhere(d)[:, 1])[:20])
