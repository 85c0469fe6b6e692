#!/usr/bin/env python3
"""How many DISTINCT filter bits does a rank's share of the text set?  The figure behind the link model of the combined
exchange (every rank combines its inserts in LDS first and routes only the distinct set bits of every
filter slice).  For W = 1, 2, 4, 8 the text of the workload is cut into W contiguous chunks of genomes (the chunk a rank of a
W-rank run hashes); each chunk is inserted alone and the set bits of the resulting filter are counted.
   python tools/distinct_bits.py [workload] [scale] > profiles/r06_distinct_bits.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from twopaco_amd import capi, synth
    workload = sys.argv[1] if len(sys.argv) > 1 else "m2"
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    recs, p = synth.workload(workload, scale=scale)
    ctx = capi.Context(0)
    ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=20240229))
    out = {"workload": workload, "scale": scale, "k": p["k"], "L": p["L"], "q": p["q"], "records": len(recs), "chunks": {}}
    for W in (1, 2, 4, 8):
        per = (len(recs) + W - 1) // W
        rows = []
        for r in range(W if W <= 2 else 2):  # the first two chunks say enough (the genomes are exchangeable)
            part = recs[r * per:(r + 1) * per]
            if not part:
                continue
            text = capi.PackedText.from_codes(part)
            ctx.seq_upload(text)
            ctx.run_begin()
            ctx.filter_reset()
            n = ctx.pass1_insert(0, None, count=True)
            words = ctx.filter_download()
            even = (words.size // 2) * 2
            bits = int(np.bitwise_count(words[:even].view(np.uint64)).sum()) + int(np.bitwise_count(words[even:]).sum())
            nz = int(np.count_nonzero(words))
            rows.append({"rank": r, "genomes": len(part), "kmers": int(n), "addresses": int(n) * p["q"], "distinct_bits": bits, "nonzero_words": nz,
                         "bytes_as_4B_bits": 4 * bits, "bytes_as_8B_words": 8 * nz, "bytes_dense": int(words.size) * 4})
            del words
        out["chunks"][str(W)] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
