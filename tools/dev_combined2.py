#!/usr/bin/env python3
"""Development: the combined calls in ONE process, ranks run one after the other; decodes the exports on the host."""
import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    import torch
    import test_gpu_combined as T
    from twopaco_amd import capi
    name, sb, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    tmp = pathlib.Path(tempfile.mkdtemp())
    spec, o = T.golden_spec(name, sb, tmp)
    lo, hi = spec["ranges"][0]
    o.fill_only(lo, hi)
    text = capi.PackedText.from_fasta(spec["files"])
    dev = torch.device("cuda", 0)
    ctxs, exports = [], []
    for r in range(W):
        c = capi.Context(0)
        c.set_option("slice_bits", sb); c.set_option("replicate_filter", 1)
        c.shard_config(r, W)
        c.set_params(spec["k"], spec["L"], spec["q"], capi.seed_table(spec["q"], spec["L"], seed=spec["seed"]))
        c.seq_upload(text)
        c.filter_reset(); c.pass1_insert(lo, hi, count=False)
        info = c.combine_info(W)
        print("rank", r, "info", info)
        cap = info["cap_units"]
        pay = torch.zeros(W * cap * 8, dtype=torch.int16, device=dev)
        d = torch.zeros(info["slices"] * info["windows"], dtype=torch.int64, device=dev)
        units = c.combine_export(W, pay.data_ptr(), cap, d.data_ptr())
        print("  units", units)
        exports.append((pay.cpu().numpy().view(np.uint16), d.cpu().numpy().view(np.uint64), units, cap, info))
        ctxs.append(c)
    # decode on the host: union of all exports -> filter
    info = exports[0][4]
    F = info["b1"] + info["b2"]; nb2 = 1 << info["b2"]; n_win = info["windows"]; spd = info["slices"] // W
    words = 1 << (sb - 5)
    # permutation inverse from a shard plan is not exposed: recover from library constant
    mult = 0x9E3779B1; x = mult
    for _ in range(5): x = (x * (2 - mult * x)) & 0xFFFFFFFF
    inv = x
    filt = np.zeros_like(o.filter)
    for r, (pay, d, units, cap, _) in enumerate(exports):
        for dest in range(W):
            for key in range(spd):
                bl, b2 = key >> info["b2"], key & (nb2 - 1)
                b1 = bl * W + dest
                sp = (b1 << info["b2"]) | b2
                s = (sp * inv) & ((1 << F) - 1)
                for w in range(n_win):
                    e = int(d[(dest * spd + key) * n_win + w])
                    n, unit = e & 0xFFFFFF, e >> 24
                    ent = pay[(dest * cap + unit) * 8:(dest * cap + unit) * 8 + n].astype(np.int64)
                    off = (w << 16) | ent
                    np.bitwise_or.at(filt, s * words + (off >> 5), (1 << (off & 31)).astype(np.uint32))
    print("host decode of the exports: missing", int(np.bitwise_count(o.filter & ~filt).sum()), "extra", int(np.bitwise_count(filt & ~o.filter).sum()), "of", int(np.bitwise_count(o.filter).sum()))
    # import all blocks on every rank (n_src = W * W, n_owner = W), query, compare
    most = max(sum(u) for _, _, u, _, _ in exports)
    allp = np.zeros((W, most * 8), dtype=np.uint16)
    alld = np.zeros((W, info["slices"] * n_win), dtype=np.uint64)
    base = []
    for r, (pay, d, units, cap, _) in enumerate(exports):
        o_ = 0
        for dest in range(W):
            allp[r, o_ * 8:(o_ + units[dest]) * 8] = pay[dest * cap * 8:(dest * cap + units[dest]) * 8]
            base.append(r * most + o_)
            o_ += units[dest]
        alld[r] = d
    # host emulation of tpc_lists_apply (n_src = W * W, n_owner = W) on exactly these arrays
    filt2 = np.zeros_like(o.filter)
    flatp, flatd = allp.reshape(-1), alld.reshape(-1)
    for sp in range(info["slices"]):
        b1, b2 = sp >> info["b2"], sp & (nb2 - 1)
        s = (sp * inv) & ((1 << F) - 1)
        key = ((b1 // W) << info["b2"]) | b2
        for i in range(W):
            src = (b1 & (W - 1)) + i * W
            for w in range(n_win):
                e = int(flatd[src * spd * n_win + key * n_win + w])
                n, unit = e & 0xFFFFFF, e >> 24
                ent = flatp[(base[src] + unit) * 8:(base[src] + unit) * 8 + n].astype(np.int64)
                off = (w << 16) | ent
                np.bitwise_or.at(filt2, s * words + (off >> 5), (1 << (off & 31)).astype(np.uint32))
    print("host emulation of the import: missing", int(np.bitwise_count(o.filter & ~filt2).sum()), "extra", int(np.bitwise_count(filt2 & ~o.filter).sum()), "base", base)
    tp = torch.from_numpy(allp.view(np.int16)).to(dev); td = torch.from_numpy(alld.view(np.int64)).to(dev)
    for r, c in enumerate(ctxs):
        c.combine_import(W * W, W, tp.data_ptr(), base, td.data_ptr(), spd * n_win)
        if len(sys.argv) > 4:
            f0 = c.filter_download()
            print("rank", r, "peek filter: missing", int(np.bitwise_count(o.filter & ~f0).sum()), "extra", int(np.bitwise_count(f0 & ~o.filter).sum()))
        m = c.pass1_query(lo, hi)
        f = c.filter_download()
        print("rank", r, "marks", m, "fused", c.stat("fused_lookups"), "filter: missing", int(np.bitwise_count(o.filter & ~f).sum()), "extra", int(np.bitwise_count(f & ~o.filter).sum()))

if __name__ == "__main__":
    main()
