#!/usr/bin/env python
"""Exhaustive check of the glibc sinf / cosf restatement (oracle trig = glibc / glibc-nofma) against the host's libm: all 2^32
float arguments, both functions.  CPU only, about a minute on 8 cores.  Writes profiles/r03_glibc_trig_exhaustive.json.

    python tools/check_glibc_trig.py [out.json]
"""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import oracle as O

    O.build()
    O.set_num_threads(O.max_threads())
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_glibc_trig_exhaustive.json")
    res = {"host_libc": list(platform.libc_ver()), "machine": platform.machine(), "threads": O.max_threads(), "ranges": []}
    t0 = time.time()
    for name, lo, hi in (("+finite", 0, 0x7f800000), ("-finite", 0x80000000, 0xff800000), ("+inf / nan", 0x7f800000, 0x80000000),
                         ("-inf / nan", 0xff800000, 0x100000000)):
        row = {"range": name, "arguments": hi - lo}
        for label, mode in (("glibc", O.TRIG_GLIBC), ("glibc-nofma", O.TRIG_GLIBC_NOFMA), ("exact", O.TRIG_EXACT)):
            if label == "exact" and "finite" not in name:
                continue
            # hi == 2^32 does not fit the uint32 argument: split off the last pattern
            n, ex = O.trig_compare_range(O.TRIG_LIBM, mode, lo, min(hi, 0xffffffff), 3, 8)
            row[label] = {"differ_from_host_libm": n, "first": [hex(e) for e in ex]}
            print(name, label, n, [hex(e) for e in ex], "%.0f s" % (time.time() - t0), flush=True)
        res["ranges"].append(row)
    res["seconds"] = round(time.time() - t0, 1)
    res["summary"] = {k: sum(r.get(k, {}).get("differ_from_host_libm", 0) for r in res["ranges"]) for k in ("glibc", "glibc-nofma", "exact")}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["summary"]))


if __name__ == "__main__":
    main()
