"""Seeded synthetic genome workloads shaped like BASELINE.json's configs.

Real assemblies cannot be fetched (no network), so every workload is generated from a
counter-based integer hash (splitmix64 finaliser over the base index): fully deterministic,
independent of numpy's RNG streams, cheap to regenerate on the GPU box.

  m1: 8 genomes x 5 Mbp, genome 0 uniform ACGT, genomes 1..7 = genome 0 with 1 % i.i.d.
      substitutions (BASELINE.json configs[1], k=25 f=32)
  m2: 62 E. coli-like genomes x 5 Mbp: 6 clades at 2 % from the root, members at 0.2 % from
      their clade ancestor, N runs (length 1..100) covering ~0.1 % (configs[2], k=25 f=36)
  m3: 7 "human-like" genomes x 160 Mbp (1.12 Gbp: more than 2^30 positions, so the partitioned query needs
      several tile batches), 0.1 % substitutions from a common root, N runs (configs[3]'s shape, k=25 f=38)
  m2r: m2 made less kind (round 5: repeat families, fragmentation, low complexity, mixed strands -- what real assemblies have and
      i.i.d. substitutions on a random root do not): 20 repeat families of 1-5 kbp copied 5-50 times into every genome (high-
      multiplicity k-mers: hot exact-filter keys, address skew), poly-A and dinucleotide tracts of 50-500 bp, two genomes
      reverse-complemented, and every genome cut into 50-300 contigs (records).  workload() then returns the records of all genomes
      and p["files"] = the record range of every genome's FASTA file.
Codes: A0 C1 G2 T3, N = 4.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(x):
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return x


def _stream(seed, n, salt=0):
    with np.errstate(over="ignore"):
        base = np.uint64((seed * 0x9E3779B97F4A7C15 + salt * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        return _mix(np.arange(n, dtype=np.uint64) * _G + base)


def _native():
    """host/synthgen.cpp: the same streams on all host threads (a 1 Gbp workload in seconds instead of minutes).
    TPC_SYNTH_NUMPY=1 forces the numpy code below (the definition; the two are compared bit for bit in the tests)."""
    import os
    if os.environ.get("TPC_SYNTH_NUMPY"):
        return None
    try:
        from . import capi
        return capi.host()
    except Exception:  # library not built yet
        return None


def random_genome(n, seed):
    L = _native()
    if L is not None:
        out = np.empty(n, dtype=np.uint8)
        L.tpch_synth_genome(seed, n, out.ctypes.data)
        return out
    return (_stream(seed, n) >> np.uint64(62)).astype(np.uint8)


def substitute(codes, rate, seed):
    """i.i.d. substitutions at `rate`, uniform over the three other letters."""
    L = _native()
    if L is not None:
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        out = np.empty_like(codes)
        L.tpch_synth_substitute(codes.ctypes.data, codes.size, int(rate * 2.0 ** 64), seed, out.ctypes.data)
        return out
    r = _stream(seed, codes.size, 1)
    hit = r < np.uint64(int(rate * 2.0 ** 64))
    shift = (_stream(seed, codes.size, 2) % np.uint64(3)).astype(np.uint8) + np.uint8(1)
    out = codes.copy()
    out[hit] = (codes[hit] + shift[hit]) & np.uint8(3)
    return out


def add_n_runs(codes, start_rate, seed, max_len=100):
    L = _native()
    if L is not None:
        out = np.array(codes, dtype=np.uint8, copy=True)
        L.tpch_synth_n_runs(out.ctypes.data, out.size, int(start_rate * 2.0 ** 64), seed, max_len)
        return out
    r = _stream(seed, codes.size, 3)
    starts = np.nonzero(r < np.uint64(int(start_rate * 2.0 ** 64)))[0]
    lens = (_stream(seed, starts.size, 4) % np.uint64(max_len)).astype(np.int64) + 1
    out = codes.copy()
    for s, l in zip(starts, lens):
        out[s:s + l] = 4
    return out


def workload(name, seed=12345, scale=1.0, m2r_features=("families", "tracts", "strands", "contigs")):
    """Returns (list of uint8 code arrays, dict(k, L, q)).  scale shrinks the genome length.  m2r_features: which of m2r's four departures
    from m2 to apply (tools/m2r_ablate.py times them one at a time; the workload and its golden are all four)."""
    if name == "m1":
        n = int(5_000_000 * scale)
        base = random_genome(n, seed)
        recs = [base] + [substitute(base, 0.01, seed + 100 + i) for i in range(1, 8)]
        return recs, dict(k=25, L=32, q=5)
    if name == "m2":
        n = int(5_000_000 * scale)
        root = random_genome(n, seed)
        recs = []
        for g in range(62):
            clade = g % 6
            anc = substitute(root, 0.02, seed + 1000 + clade)
            mem = substitute(anc, 0.002, seed + 2000 + g)
            recs.append(add_n_runs(mem, 2e-5, seed + 3000 + g))
        return recs, dict(k=25, L=36, q=5)
    if name == "m3":
        n = int(160_000_000 * scale)
        root = random_genome(n, seed)
        recs = [add_n_runs(substitute(root, 0.001, seed + 4000 + g), 2e-6, seed + 5000 + g) for g in range(7)]
        return recs, dict(k=25, L=38, q=5)
    if name == "m2r2":  # m2r + minisatellite tracts (period 7..60): the periodic windows k_periodic_build does NOT skip (it stops at 6)
        return workload("m2r", seed=seed, scale=scale, m2r_features=tuple(m2r_features) + ("minisat",))
    if name == "m2r":
        n = int(5_000_000 * scale)
        root = random_genome(n, seed)
        fam_len = [1000 + int(x % np.uint64(4001)) for x in _stream(seed, 20, 11)]
        # (seeds far apart: _stream(seed + 1, i) == _stream(seed, i + 1), so neighbouring seeds would make the families shifted copies
        #  of ONE sequence -- and, below, put copy i of family f + 1 where copy i + 1 of family f went)
        fams = [(_stream(seed + 7000, fam_len[f], 100 + f) >> np.uint64(62)).astype(np.uint8) for f in range(20)]
        recs, files = [], []
        for g in range(62):
            clade = g % 6
            anc = substitute(root, 0.02, seed + 1000 + clade)
            mem = np.array(substitute(anc, 0.002, seed + 2000 + g), dtype=np.uint8, copy=True)
            # (i) repeat families: every family copied 5..50 times over random places of this genome (scaled with the genome)
            copies = _stream(seed + g, 20, 12) % np.uint64(46) + np.uint64(5)
            for f in range(20 if "families" in m2r_features else 0):
                c = max(1, int(int(copies[f]) * min(1.0, scale * 4)))
                at = _stream(seed + 8000 + g, c, 200 + f) % np.uint64(max(1, n - fam_len[f]))
                for a in at:
                    mem[int(a):int(a) + fam_len[f]] = fams[f][:max(0, min(fam_len[f], n - int(a)))]
            # (iii) low-complexity tracts: poly-A, poly-T and (CA)n / (GT)n of 50..500 bp
            nt = max(1, int(24 * min(1.0, scale * 4))) if "tracts" in m2r_features else 0
            t_at = _stream(seed + 9000 + g, nt, 14) % np.uint64(max(1, n - 500))
            t_len = _stream(seed + 9000 + g, nt, 15) % np.uint64(451) + np.uint64(50)
            t_kind = _stream(seed + 9000 + g, nt, 16) % np.uint64(4)
            for a, l, kd in zip(t_at, t_len, t_kind):
                a, l, kd = int(a), int(l), int(kd)
                l = min(l, n - a)
                unit = [[0], [3], [1, 0], [2, 3]][kd]
                mem[a:a + l] = np.resize(np.array(unit, dtype=np.uint8), l)
            # (v, m2r2 only) minisatellites: a random unit of 7..60 bp repeated over 200..2000 bp, 24 per genome (scaled)
            nm = max(1, int(24 * min(1.0, scale * 4))) if "minisat" in m2r_features else 0
            m_at = _stream(seed + 9700 + g, nm, 24) % np.uint64(max(1, n - 2000))
            m_len = _stream(seed + 9700 + g, nm, 25) % np.uint64(1801) + np.uint64(200)
            m_per = _stream(seed + 9700 + g, nm, 26) % np.uint64(54) + np.uint64(7)
            for j, (a, l, per) in enumerate(zip(m_at, m_len, m_per)):
                a, l, per = int(a), int(l), int(per)
                l = min(l, n - a)
                unit = (_stream(seed + 9800 + g, per, 300 + j) >> np.uint64(62)).astype(np.uint8)
                mem[a:a + l] = np.resize(unit, l)
            mem = add_n_runs(mem, 2e-5, seed + 3000 + g)
            # (iv) two genomes on the other strand
            if g in (7, 31) and "strands" in m2r_features:
                mem = np.where(mem == 4, 4, 3 - mem).astype(np.uint8)[::-1].copy()
            # (ii) contigs: 50..300 records per genome (scaled), cut at random places
            nc = max(1, int((50 + int(_stream(seed + g, 1, 17)[0] % np.uint64(251))) * min(1.0, scale * 4))) if "contigs" in m2r_features else 1
            cuts = np.unique(_stream(seed + 9500 + g, nc - 1, 18) % np.uint64(max(1, n))) if nc > 1 else np.zeros(0, dtype=np.uint64)
            edges = [0] + [int(x) for x in cuts if 0 < int(x) < n] + [n]
            first = len(recs)
            for a, b in zip(edges[:-1], edges[1:]):
                if b > a:
                    recs.append(mem[a:b].copy())
            files.append((first, len(recs)))
        return recs, dict(k=25, L=36, q=5, files=files)
    raise ValueError("unknown workload " + name)


def fasta_files(recs, p, directory, prefix="g"):
    """Writes the workload as FASTA files -- one record per file, or p["files"] = the record range of every file (m2r: a genome's
    contigs) -- and returns the paths."""
    import os
    groups = p.get("files") or [(i, i + 1) for i in range(len(recs))]
    paths = []
    for i, (a, b) in enumerate(groups):
        path = os.path.join(str(directory), "%s%d.fa" % (prefix, i))
        write_fasta(path, recs[a:b], first_id=a)
        paths.append(path)
    return paths


def n_kmers(recs, k):
    """Vertex positions with an N-free window (what the first pass hashes)."""
    total = 0
    for r in recs:
        if r.size < k:
            continue
        bad = np.concatenate([[0], np.cumsum(r == 4)])
        total += int(np.count_nonzero(bad[k:] - bad[:-k] == 0))
    return total


_LETTERS = np.frombuffer(b"ACGTN", dtype=np.uint8)


def write_fasta(path, recs, first_id=0, width=80):
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">%d\n" % (first_id + i))
            s = _LETTERS[r]
            full = (s.size // width) * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = s[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if s.size > full:
                f.write(s[full:].tobytes() + b"\n")
