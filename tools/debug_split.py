import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import oracle as O
from twopaco_amd import capi, synth
recs, _ = synth.workload("m2", scale=0.004)
letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
for (k, L, nrec) in [(51, 40, 62), (51, 30, 62), (25, 40, 62), (51, 40, 3)]:
    rr = recs[:nrec]
    o = O.Oracle(k, L, 5, O.seed_table(31337, 5, L))
    for r in rr:
        o.add_record(letters[r].tobytes())
    bins = o.split_bins()
    text = capi.PackedText.from_codes(rr)
    ctx = capi.Context(0)
    ctx.set_params(k, L, 5, capi.seed_table(5, L, seed=31337))
    ctx.seq_upload(text)
    rs, rl = text.rec_start, text.rec_length
    keep = rl >= k
    got = ctx.pass1_split_hist(rs[keep], rl[keep])
    d = got.astype(np.int64) - bins.astype(np.int64)
    nz = np.nonzero(d)[0]
    print(k, L, nrec, "sum got", got.sum(), "oracle", bins.sum(), "bins differing", nz.size, "diff sum", d.sum(), "first", nz[:6], d[nz[:6]])
    ctx.close(); o.close()
