"""Synthetic overlap graphs for the step-4 tests: chains between branching nodes (parallel ones, closed ones, tips), cycles, two-cycles,
nodes whose two edges do not combine, multi-edges -- with the ids scattered at random, since the reference's sweeps are id-ordered.
Edges come out as graph3 holds them: from < to, one per (from, to, type), ascending; reads all have one length, so the twin's
lengthOfEdge equals the edge's (overlapGraph.cpp:147-150)."""
import numpy as np

EDGE_DTYPE = np.dtype([("from", "<u8"), ("to", "<u8"), ("length", "<u4"), ("length_twin", "<u4"), ("type", "u1"), ("pad", "u1", (7,))])


def _rev(t):
    return {0: 3, 3: 0, 1: 1, 2: 2}[t]


def random_graph(seed, n_anchor=30, n_paths=60, max_len=12, n_cycles=3, p_bad=0.03, len_hi=25):
    rng = np.random.default_rng(seed)
    nodes = 0
    def new():
        nonlocal nodes
        nodes += 1; return nodes
    anchors = [new() for _ in range(n_anchor)]
    raw = []                                           # (x, y, ox, oy, len): x -> y leaving x with orientation ox, entering y with oy
    def path(seq, closed=False):
        o = {v: int(rng.integers(0, 2)) for v in seq}
        pairs = list(zip(seq[:-1], seq[1:])) + ([(seq[-1], seq[0])] if closed else [])
        for j, (x, y) in enumerate(pairs):
            ox = o[x] if (j > 0 or closed) else int(rng.integers(0, 2))         # path ends join their node in any orientation
            oy = o[y] if (j < len(pairs) - 1 or closed) else int(rng.integers(0, 2))
            if rng.random() < p_bad: oy ^= 1
            raw.append((x, y, ox, oy, int(rng.integers(1, len_hi))))
    for _ in range(n_paths):
        kind = rng.random()
        u = anchors[int(rng.integers(0, n_anchor))]
        v = u if kind < 0.12 else anchors[int(rng.integers(0, n_anchor))]
        if kind > 0.85: v = new()                      # a tip
        m = int(rng.integers(0, max_len + 1))
        if u == v and m < 2: m = 2
        path([u] + [new() for _ in range(m)] + [v])
    for _ in range(n_cycles):
        m = int(rng.integers(2, max_len + 3))
        path([new() for _ in range(m)], closed=True)
    N = nodes
    perm = rng.permutation(N) + 1                      # scatter the ids
    seen, out = set(), []
    for x, y, ox, oy, ln in raw:
        a, b = int(perm[x - 1]), int(perm[y - 1])
        if a == b: continue
        t = (ox << 1) | oy
        if a > b: a, b, t = b, a, _rev(t)
        if (a, b, t) in seen: continue
        seen.add((a, b, t)); out.append((a, b, t, ln))
    out.sort()
    e = np.zeros(len(out), dtype=EDGE_DTYPE)
    for i, (a, b, t, ln) in enumerate(out):
        e[i]["from"], e[i]["to"], e[i]["type"], e[i]["length"], e[i]["length_twin"] = a, b, t, ln, ln
    return N, e


def write_graph3(path, N, e, read_len=100):
    with open(path, "w") as f:
        f.write(f"0\n{N}\n{read_len}\n")
        for r in e:
            f.write(f"{r['from']}\t{r['to']}\t{r['type']}\t1\t{r['length']}\t0\t0\n\n")
            f.write(f"{r['to']}\t{r['from']}\t{_rev(int(r['type']))}\t1\t{r['length_twin']}\t0\t0\n\n")


def write_reads(path, N, read_len=100, seed=0):
    """a .reads file with N distinct reads (readLoader.cpp:29-36: frequency, length, forward, reverse complement)"""
    rng = np.random.default_rng(seed + 12345)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    seqs = set()
    while len(seqs) < N:
        seqs.add("".join(rng.choice(list("ACGT"), size=read_len)))
    with open(path, "w") as f:
        f.write(f"{N}\n")
        for s in sorted(seqs):
            f.write(f"1\t{read_len}\t{s}\t{''.join(comp[c] for c in reversed(s))}\n")
