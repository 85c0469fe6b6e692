#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (one or more result databases):
   python tools/pmc_sq.py gpurun_out/prof_X_sq1/*.db gpurun_out/prof_X_sq2/*.db > gpurun_out/X_sq.csv
One row per kernel, one column per counter (average per dispatch, summed over the dimensions rocprofv3 reports)."""
import re
import sqlite3
import sys


def short_name(name):
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short).replace("void ", "")
    return short


def main():
    table = {}
    counters = []
    for db in sys.argv[1:]:
        c = sqlite3.connect(db)
        per = {}
        for name, counter, value, disp in c.execute("select name, counter_name, counter_value, dispatch_id from pmc_events"):
            k = short_name(name)
            if "rocprim" in k or k.startswith("__amd"):
                continue
            if counter not in counters:
                counters.append(counter)
            d = per.setdefault((k, counter), {})
            d[disp] = d.get(disp, 0.0) + value
        for (k, counter), d in per.items():
            table.setdefault(k, {})[counter] = sum(d.values()) / len(d)
    print("kernel," + ",".join(counters))
    for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", table[k].get(counters[0], 0))):
        print('"%s",' % k + ",".join("%.4g" % table[k].get(cn, float("nan")) for cn in counters))


if __name__ == "__main__":
    main()
