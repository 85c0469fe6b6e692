#!/usr/bin/env python3
"""The combined exchange's device work of W ranks, one after the other in ONE process on one GPU (no transport: the blocks are moved
with torch on the device): per-call wall times per rank, and -- run under `rocprofv3 --kernel-trace --stats` -- the kernels behind them.
   python tools/combine_profile.py [W] [workload] [scale] [mode]      mode: scatter (default for W > 2) | gather"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from twopaco_amd import capi, synth
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    workload = sys.argv[2] if len(sys.argv) > 2 else "m2"
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    mode = sys.argv[4] if len(sys.argv) > 4 else ("gather" if W == 2 else "scatter")
    recs, p = synth.workload(workload, scale=scale)
    text = capi.PackedText.from_codes(recs)
    dev = torch.device("cuda", 0)
    ms = [dict() for _ in range(W)]

    def timed(r, name, fn, *a):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a)
        torch.cuda.synchronize()
        ms[r][name] = ms[r].get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return out

    ctxs = []
    for r in range(W):
        c = capi.Context(0)
        c.set_option("replicate_filter", 1)
        c.set_option("text_window", 1)
        c.shard_config(r, W)
        c.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=20240229))
        c.seq_upload(text)
        ctxs.append(c)
    out = {"world": W, "mode": mode, "workload": workload, "steps": []}
    for step in range(2):  # the second step is the warm one
        for m in ms:
            m.clear()
        exports = []
        for r, c in enumerate(ctxs):
            timed(r, "filter_reset", c.filter_reset)
            timed(r, "pass1_insert", c.pass1_insert, 0, None, False)
            info = timed(r, "combine_info", c.combine_info, W)
            assert info["sparse"], info
            cap = info["cap_units"]
            pay = torch.empty(W * cap * 2, dtype=torch.int64, device=dev)
            d = torch.empty(info["slices"] * info["windows"], dtype=torch.int64, device=dev)
            units = timed(r, "combine_export", c.combine_export, W, pay.data_ptr(), cap, d.data_ptr())
            blocks = [pay[2 * k * cap:2 * (k * cap + units[k])].clone() for k in range(W)]
            exports.append((blocks, d, units, info))
            del pay
        info = exports[0][3]
        spd, n_win = info["slices"] // W, info["windows"]
        if mode == "gather":
            allp = torch.cat([b for blocks, _, _, _ in exports for b in blocks])
            alld = torch.cat([d for _, d, _, _ in exports])
            base, o = [], 0
            for _, _, units, _ in exports:
                for u in units:
                    base.append(o)
                    o += u
            for r, c in enumerate(ctxs):
                timed(r, "combine_import", c.combine_import, W * W, W, allp.data_ptr(), base, alld.data_ptr(), spd * n_win)
        else:
            merged = []
            for r, c in enumerate(ctxs):
                recv = torch.cat([exports[s][0][r] for s in range(W)])
                rdir = torch.cat([exports[s][1][r * spd * n_win:(r + 1) * spd * n_win] for s in range(W)])
                base, o = [], 0
                for s in range(W):
                    base.append(o)
                    o += exports[s][2][r]
                mp = torch.empty(max(o, 1) * 2, dtype=torch.int64, device=dev)
                md = torch.empty(spd * n_win, dtype=torch.int64, device=dev)
                mu = timed(r, "combine_merge", c.combine_merge, W, recv.data_ptr(), base, rdir.data_ptr(), mp.data_ptr(), max(o, 1), md.data_ptr())
                merged.append((mp[:2 * mu].clone(), md, mu))
            allp = torch.cat([m[0] for m in merged])
            alld = torch.cat([m[1] for m in merged])
            base, o = [], 0
            for m in merged:
                base.append(o)
                o += m[2]
            for r, c in enumerate(ctxs):
                timed(r, "combine_import", c.combine_import, W, W, allp.data_ptr(), base, alld.data_ptr(), spd * n_win)
        marks = 0
        for r, c in enumerate(ctxs):
            marks += timed(r, "pass1_query", c.pass1_query, 0, None)
            ms[r]["kernel_ms"] = {k: c.kernel_ms(k) for k in ("insert", "query", "fused", "lookup", "combine")}
        out["steps"].append({"marks": marks, "export_units": [e[2] for e in exports], "merged_units": [m[2] for m in merged] if mode != "gather" else None,
                             "call_ms": [dict(m) for m in ms]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
