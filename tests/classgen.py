"""Counter-based generators of the input classes away from iid ACGTN (test infrastructure).

Every byte is a function of (class, seed, position) through splitmix64 -- the generator of
tc_generate_dev / SURVEY.md 8(d) -- in integer arithmetic only, so the oracle's host
(tests/long/classes_digest.py, here) and the GPU box's host (tests/test_gpu_classes_digest.py)
produce the same record from numpy alone; nothing is drawn from a library RNG or from floating point.

Classes (what path of the library each one is there for is said in classes_digest.py):
  zipf_words  natural-language-like: Zipf(1)-distributed words of 2..9 lower-case letters from a
              20 000-word vocabulary, single spaces (sigma = 28 with the sentinel)
  bytes256    iid uniform over all 256 byte values (sigma = 257)
  ascii96     iid printable ASCII, the `kind 1` stream of tc_generate_dev (sigma = 96)
  acgt4       iid uniform over ACGT
  genome_like iid ACGT with a 300-bp repeat family (15 % divergence, ~10 % of the sequence), poly-A
              tracts and (CA)n microsatellites
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_CHUNK = 1 << 24


def splitmix64_at(seed, i):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (i.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(seed, n, fn, dtype=np.uint8):
    """fn(x: u64 splitmix values of positions lo..hi) -> values; evaluated in bounded chunks"""
    out = np.empty(n, dtype=dtype)
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        out[lo:hi] = fn(splitmix64_at(seed, np.arange(lo, hi, dtype=np.uint64)))
    return out


def _scaled(x, m):
    """((x >> 32) * m) >> 32: uniform over 0..m-1, the form SURVEY.md 8(d) uses"""
    return ((x >> np.uint64(32)) * np.uint64(m)) >> np.uint64(32)


def bytes256(n, seed=0xB256):
    return _stream(seed, n, lambda x: (x >> np.uint64(56)).astype(np.uint8))


def ascii96(n, seed=0xA596):
    return _stream(seed, n, lambda x: (np.uint64(0x20) + _scaled(x, 95)).astype(np.uint8))


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def acgt4(n, seed=0xAC64):
    return _stream(seed, n, lambda x: _ACGT[_scaled(x, 4).astype(np.int64)])


def zipf_words(n, seed=0x21BF, vocab=20000):
    k = np.arange(vocab, dtype=np.uint64)
    wl = (2 + splitmix64_at(seed + 1, k) % np.uint64(8)).astype(np.int64)     # letters per word: 2..9
    off = np.concatenate([[0], np.cumsum(wl + 1)])                             # word k = flat[off[k] : off[k+1]], space last
    flat = (97 + splitmix64_at(seed + 2, np.arange(int(off[-1]), dtype=np.uint64)) % np.uint64(26)).astype(np.uint8)
    flat[off[1:] - 1] = 32
    w = (np.uint64(1) << np.uint64(40)) // (k + np.uint64(1))                  # Zipf(1) weights, integers
    cw = np.cumsum(w, dtype=np.uint64)
    nw = n // 4 + 1024                                                         # mean word + space is > 4 bytes
    u = _stream(seed + 3, nw, lambda x: (x >> np.uint64(20)) % cw[-1], dtype=np.uint64)
    ids = np.searchsorted(cw, u, side="right").astype(np.int64)
    del u
    lens = (wl + 1)[ids]
    ends = np.cumsum(lens)
    nwords = int(np.searchsorted(ends, n, side="left")) + 1                    # words that cover n bytes
    assert nwords <= nw
    ids, lens, ends = ids[:nwords], lens[:nwords], ends[:nwords]
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 21                                                             # words per piece (bounded index arrays)
    for a in range(0, nwords, step):
        b = min(nwords, a + step)
        p0 = int(ends[a] - lens[a])
        src = np.repeat(off[:-1][ids[a:b]] - (ends[a:b] - lens[a:b]), lens[a:b])   # source offset minus position
        pos = np.arange(p0, p0 + len(src), dtype=np.int64)
        m = min(len(src), n - p0)
        if m <= 0:
            break
        out[p0:p0 + m] = flat[src[:m] + pos[:m]]
    return out


def genome_like(n, seed=0x6E0E):
    t = acgt4(n, seed)
    fam = _ACGT[_scaled(splitmix64_at(seed + 1, np.arange(300, dtype=np.uint64)), 4).astype(np.int64)]
    ncopy = n // 3000
    pos = (splitmix64_at(seed + 2, np.arange(ncopy, dtype=np.uint64)) % np.uint64(n - 400)).astype(np.int64)
    order = np.argsort(pos, kind="stable")       # copies are laid down in position order: later ones overwrite
    pos = pos[order]
    offs = np.arange(300, dtype=np.int64)
    for a in range(0, ncopy, 1 << 16):
        b = min(ncopy, a + (1 << 16))
        dst = (pos[a:b, None] + offs[None, :]).reshape(-1)
        cid = np.repeat(order[a:b].astype(np.uint64), 300) * np.uint64(300) + np.tile(offs.astype(np.uint64), b - a)
        x = splitmix64_at(seed + 3, cid)
        mut = _scaled(x, 100) < np.uint64(15)
        rnd = _ACGT[((x >> np.uint64(8)) & np.uint64(3)).astype(np.int64)]
        src = np.where(mut, rnd, np.tile(fam, b - a))
        t[dst] = src                             # numpy assigns in index order: the last writer of a slot wins
    npoly = n // 20000
    j = np.arange(npoly, dtype=np.uint64)
    pos = (splitmix64_at(seed + 4, j) % np.uint64(n - 100)).astype(np.int64)
    ln = (15 + splitmix64_at(seed + 5, j) % np.uint64(45)).astype(np.int64)
    o60 = np.arange(60, dtype=np.int64)
    msk = o60[None, :] < ln[:, None]
    t[(pos[:, None] + o60[None, :])[msk]] = 65
    ntr = n // 100000
    pos = (splitmix64_at(seed + 6, np.arange(ntr, dtype=np.uint64)) % np.uint64(n - 200)).astype(np.int64)
    dst = (pos[:, None] + np.arange(100, dtype=np.int64)[None, :]).reshape(-1)
    t[dst] = np.tile(np.tile(np.array([67, 65], dtype=np.uint8), 50), ntr)
    return t


# ---- the device-side generators of the bench's classes leg (tc_generate_dev kinds 2 .. 6, csrc/textcomp.hip), restated in
# numpy: the same integer function of (kind, seed, position), so that the oracle can encode what the device generates

def _mix(seed, i):
    return splitmix64_at(seed, np.asarray(i, dtype=np.uint64))


def dev_periodic(n, seed=0x4B1B, period=4096):
    blk = _ACGT[_scaled(_mix(seed, np.arange(period)), 4).astype(np.int64)]
    return np.resize(blk, n)


def dev_runs(n, seed=0x9A75):
    out = np.empty(n, dtype=np.uint8)
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        a = max(0, lo - 512)
        i = np.arange(a, hi, dtype=np.uint64)
        start = (_scaled(_mix(seed, i), 10) == 0) | (i == 0)
        # the run head of position i: the last start at or before it, at most 512 back (the device walks back 512 steps)
        last = np.maximum.accumulate(np.where(start, i, np.uint64(0)))
        head = np.where(i - last > np.uint64(512), i - np.uint64(512), last)
        head = np.where((i < np.uint64(512)) & (last == 0), np.uint64(0), head)
        out[lo:hi] = _ACGT[((_mix(seed + 1, head) >> np.uint64(40)) & np.uint64(3)).astype(np.int64)][lo - a:]
    return out


def dev_genome_like(n, seed=0x6E0E):
    out = np.empty(n, dtype=np.uint8)
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        i = np.arange(lo, hi, dtype=np.uint64)
        b = _ACGT[_scaled(_mix(seed, i), 4).astype(np.int64)]
        c3, o3 = i // np.uint64(3000), i % np.uint64(3000)
        f0 = _scaled(_mix(seed + 2, c3), 2700)
        infam = (o3 >= f0) & (o3 < f0 + np.uint64(300))
        x = _mix(seed + 3, i)
        fam = _ACGT[_scaled(_mix(seed + 1, np.where(infam, o3 - f0, np.uint64(0))), 4).astype(np.int64)]
        mut = _ACGT[((x >> np.uint64(8)) & np.uint64(3)).astype(np.int64)]
        b = np.where(infam, np.where(_scaled(x, 100) < np.uint64(15), mut, fam), b)
        c2, o2 = i // np.uint64(20000), i % np.uint64(20000)
        a0 = _scaled(_mix(seed + 4, c2), 19900)
        al = np.uint64(15) + _scaled(_mix(seed + 5, c2), 45)
        b = np.where((o2 >= a0) & (o2 < a0 + al), np.uint8(65), b)
        c1, o1 = i // np.uint64(100000), i % np.uint64(100000)
        m0 = _scaled(_mix(seed + 6, c1), 99800)
        inm = (o1 >= m0) & (o1 < m0 + np.uint64(100))
        b = np.where(inm, np.where(((o1 - m0) & np.uint64(1)) == 1, np.uint8(65), np.uint8(67)), b)
        out[lo:hi] = b
    return out


def dev_zipf_words(n, seed=0x21BF, vocab=20000):
    k = np.arange(vocab, dtype=np.uint64)
    cw = np.cumsum((np.uint64(1) << np.uint64(40)) // (k + np.uint64(1)), dtype=np.uint64)
    wlen = (2 + _mix(seed + 1, k) % np.uint64(8)).astype(np.int64)
    letters = (97 + _mix(seed + 2, (k[:, None] * np.uint64(16) + np.arange(16, dtype=np.uint64)[None, :])) % np.uint64(26)).astype(np.uint8)
    ncell = (n + 4095) // 4096
    out = np.zeros(ncell * 4096, dtype=np.uint8)
    pos = np.zeros(ncell, dtype=np.int64)                      # bytes written in each cell so far
    cells = np.arange(ncell, dtype=np.uint64)
    w = 0
    while True:
        live = np.nonzero(pos < 4096)[0]
        if live.size == 0:
            break
        u = (_mix(seed + 3, cells[live] * np.uint64(4096) + np.uint64(w)) >> np.uint64(20)) % cw[-1]
        ids = np.searchsorted(cw, u, side="right")
        ln = wlen[ids]
        for t in range(10):                                    # letters 0 .. len - 1, then the space
            ch = np.where(t < ln, letters[ids, min(t, 15)], np.uint8(32))
            ok = (t <= ln) & (pos[live] + t < 4096)
            out[(live * 4096 + pos[live] + t)[ok]] = ch[ok]
        pos[live] += ln + 1
        w += 1
    return out[:n]


def dev_gaps(n, seed=0x6A95):
    out = np.empty(n, dtype=np.uint8)
    g0, gl, cell, sl = n // 3, n // 64, n // 40, n // 4096
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        i = np.arange(lo, hi, dtype=np.uint64)
        b = _ACGT[_scaled(_mix(seed, i), 4).astype(np.int64)]
        big = (i >= np.uint64(g0)) & (i < np.uint64(g0 + gl))
        if cell:
            c = i // np.uint64(cell)
            small = ((c & np.uint64(1)) == 1) & (c < np.uint64(32)) & (i - c * np.uint64(cell) < np.uint64(sl))
        else:
            small = np.zeros(hi - lo, dtype=bool)
        out[lo:hi] = np.where(big | small, np.uint8(78), b)
    return out


CLASSES = {"zipf_words": zipf_words, "bytes256": bytes256, "ascii96": ascii96, "acgt4": acgt4, "genome_like": genome_like,
           "dev_genome_like": dev_genome_like, "dev_zipf_words": dev_zipf_words, "dev_runs": dev_runs, "dev_periodic": dev_periodic, "dev_gaps": dev_gaps}
DEV_KINDS = {"dev_genome_like": (2, 0x6E0E), "dev_zipf_words": (3, 0x21BF), "dev_runs": (4, 0x9A75), "dev_periodic": (5, 0x4B1B), "dev_gaps": (6, 0x6A95)}


def make(name, n):
    return np.ascontiguousarray(CLASSES[name](n))
