import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
k, pool, n_rec = 31, 0, 120859
s = kmc.Synth(seed=908266423, pool=pool)
hb, ho = kmc.synth_reads_host(s, 291, n_rec)
rng = np.random.default_rng(1)
hbN = hb.copy(); hbN[rng.integers(0, hb.size, size=hb.size // 5000)] = ord("N")
ALG = {"sort": kmc.ALGO_SORT, "auto": kmc.ALGO_AUTO, "walk": kmc.ALGO_WALK}
for name, bases in (("clean", hb), ("withN", hbN)):
    want = oracle_py.count_kmers(bases, ho, k, True, method=1)
    for algo in ("walk", "auto", "sort"):
        for two in (False, True):
            try:
              with kmc.KmerCounter(k=k, algo=ALG[algo]) as kc:
                if two:
                    cut = 11624; c0 = int(ho[cut])
                    kc.add_batch(bases[:c0], ho[:cut + 1]); kc.add_batch(bases[c0:], ho[cut:] - ho[cut])
                else:
                    kc.add_batch(bases, ho)
                got = kc.export(); st = kc.stats()
                print(name, algo, "two" if two else "one", "ok" if got.equals(want) else "MISMATCH", got.n_distinct, want.n_distinct, got.n_total, want.n_total, "algo_last", st.algo_last, "launches", st.launches_last, flush=True)
            except Exception as e:
                print(name, algo, "two" if two else "one", "EXC", str(e)[:120], flush=True)
