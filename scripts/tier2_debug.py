import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, textcomp
ctx = textcomp.Context(0)
def genome_like(seed, n):
    rng = np.random.default_rng(seed)
    t = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
    fam = rng.choice(np.frombuffer(b"ACGT", np.uint8), 300)
    ncopy = n // 3000
    for p in rng.integers(0, n - 400, ncopy):
        c = fam.copy(); mut = rng.random(300) < 0.15
        c[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(mut.sum())); t[p:p + 300] = c
    for p in rng.integers(0, n - 100, n // 20000):
        t[p:p + rng.integers(15, 60)] = 65
    return t.tobytes()
for lg in (22, 26):
    n = 1 << lg
    t = genome_like(lg, n)
    os.environ["TC_SA_TIER2"] = "1"
    sa1 = ctx.suffix_array(t); st = ctx.stats(); info = (st.finish_pass, st.rounds, [int(st.m[i]) for i in range(st.rounds)], [int(st.h[i]) for i in range(st.rounds)])
    os.environ["TC_SA_TIER2"] = "0"
    os.environ["TC_SA_H_START"] = "12"
    sah = ctx.suffix_array(t); st = ctx.stats(); infoh = (st.finish_pass, st.rounds, [int(st.m[i]) for i in range(st.rounds)], [int(st.h[i]) for i in range(st.rounds)])
    del os.environ["TC_SA_H_START"]
    sa0 = ctx.suffix_array(t); st = ctx.stats(); info0 = (st.finish_pass, st.rounds, [int(st.m[i]) for i in range(st.rounds)])
    keep_sah = sah
    bad = np.nonzero(sa0 != sa1)[0]
    print("full path with h_start=12:", infoh, "mismatches vs full:", int((keep_sah != sa0).sum()))
    print("n=2^%d tier2 %s | full %s | mismatches %d first %s" % (lg, info, info0, len(bad), bad[:6]), flush=True)
    if len(bad):
        j = int(bad[0]); a, b = int(sa1[j]), int(sa0[j])
        print("  slot", j, "tier2 suffix", a, t[a:a+40], "full suffix", b, t[b:b+40])
        # how long is the common prefix with the neighbours
        def lcp(x, y):
            k = 0
            while x + k < n and y + k < n and t[x + k] == t[y + k]: k += 1
            return k
        def brange(k):
            lo = j
            while lo > 0 and lcp(int(sa0[lo - 1]), b) >= k: lo -= 1
            hi = j
            while hi + 1 <= n and lcp(int(sa0[hi + 1]), b) >= k: hi += 1
            return lo, hi + 1
        print("  12-symbol bucket of slot j:", brange(12), " 18-symbol group:", brange(18), " 24:", brange(24), "27:", brange(27))
        lo, hi = brange(18)
        print("  full  order:", [int(x) for x in sa0[lo:hi]])
        print("  tier2 order:", [int(x) for x in sa1[lo:hi]])
        for x in sa0[lo:hi]: print("   ", int(x), t[int(x):int(x)+60])
        print("  lcp(tier2[j], full[j]) =", lcp(a, b), " lcp(full[j-1], full[j]) =", lcp(int(sa0[j-1]), b) if j else -1)
        break
