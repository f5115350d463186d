"""CPU model (numpy; no GPU, no library): would a CHAMPION-LIST starting bound help the staged route?  For the BASELINE query set on zipf-N, per
query: the rank-table starting bound (what the library uses), the bound from the exact scores of the R highest-factor postings of each of its
terms (k-th best over distinct documents), and the true k-th best score; and the candidate postings (sum of df over the essential terms) each
of them leaves.  Result at 1 M docs: k = 10, R = 64 reaches the candidates of perfect bounds (-26 %), k = 32 -17 %, k = 100 -3 %.

    python scripts/champion_seed_model.py [docs]
"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from searcharray_amd import synth
N=int(sys.argv[1]) if len(sys.argv)>1 else 1_000_000
V=100_000
lens, terms = synth.zipf_batch_tokens(0, N, V, 32, 1234, fast=True)
doc = np.repeat(np.arange(N, dtype=np.int64), lens)
key = terms.astype(np.int64) * N + doc
uk, tf = np.unique(key, return_counts=True)
pt, pd = uk // N, uk % N
off = np.searchsorted(pt, np.arange(V + 1))
dl = lens.astype(np.float32); avgdl = np.float32(dl.mean())
norm = np.float32(1.2) * ((np.float32(1) - np.float32(0.75)) + np.float32(0.75) * (dl[pd] / avgdl))
fac = tf.astype(np.float32) / (tf.astype(np.float32) + norm)
df = np.diff(off)
idf = np.log(1 + (N - df + 0.5) / (df + 0.5)).astype(np.float32)
qs = synth.bm25_queries(256, V)
for k in (10, 32, 100):
  for R in (16, 64):
    tot = {'seed':0,'champ':0,'true':0}; post=0; ratios={'seed':[],'champ':[]}
    for q in qs[:128]:
        sl = [slice(off[t], off[t + 1]) for t in q]
        w = idf[q]
        sc = np.zeros(N, dtype=np.float32)
        for wi, s in zip(w, sl): sc[pd[s]] += fac[s] * wi
        true_kth = np.partition(sc, -k)[-k]
        kth = [np.partition(fac[s], -k)[-k] if df[t] >= k else 0.0 for s, t in zip(sl, q)]
        seed = max(wi * f for wi, f in zip(w, kth))
        RR = max(R, k)
        champs = set()
        for s, t in zip(sl, q):
            n = df[t]
            if n == 0: continue
            idx = np.argpartition(fac[s], -min(RR, n))[-min(RR, n):]
            champs.update(pd[s][idx].tolist())
        cs = np.sort(sc[np.fromiter(champs, dtype=np.int64)])[::-1]
        champ = max(seed, cs[k-1] if len(cs) >= k else 0.0)
        ub = np.array([wi * (fac[s].max() if df[t] else 0.0) for wi, s, t in zip(w, sl, q)], dtype=np.float32)
        order = np.argsort(-ub); sfx = np.cumsum(ub[order][::-1])[::-1]
        for name, G in (('seed',seed),('champ',champ),('true',true_kth)):
            n_ess = 4
            for i in range(4):
                if sfx[i] * 1.00001 < G: n_ess = i; break
            tot[name] += sum(int(df[q[order[i]]]) for i in range(n_ess))
        post += int(df[q].sum())
        ratios['seed'].append(seed/true_kth); ratios['champ'].append(champ/true_kth)
    print(f"k={k} R={R}: candidates seed {tot['seed']:,} champ {tot['champ']:,} true {tot['true']:,} of {post:,}; seed/true median {np.median(ratios['seed']):.3f} min {min(ratios['seed']):.3f}; champ/true median {np.median(ratios['champ']):.3f} min {min(ratios['champ']):.3f}", flush=True)
