#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into small text files that
can be committed under profiles/.

  python tools/rocpd_summary.py trace  <trace_results.db>            -> per-kernel stats
  python tools/rocpd_summary.py pmc    <pmc_results.db> [more.db...] -> per-kernel counter means
  python tools/rocpd_summary.py hbm    <fetch.db> <write.db> <kernel-substring> <out.json>
        -> per-launch HBM bytes for one kernel, corrected as MI355X_MICROARCH.md
           prescribes (FETCH_SIZE is KiB and reads 1/2 of wide coalesced reads on gfx950:
           bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024)
"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    name = re.sub(r"lsa::Fp<lsa::FqParams>\s*", "Fq", name)
    name = re.sub(r"lsa::Fp<lsa::FrParams>\s*", "Fr", name)
    name = name.replace("lsa::", "")
    if len(name) > 70:
        name = name[:67] + "..."
    return name


def trace(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("%-72s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        print("%-72s %7d %12.1f %12.3f %6.2f%%" % (short(name), calls, total, avg, pct))



def spread(db_path, needle=""):
    """per-kernel min / median / max duration (us) from the `kernels` view"""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    dur = "duration" if "duration" in cols else "(end - start)"
    rows = db.execute("select name, %s from kernels" % dur).fetchall()
    by = {}
    for name, d in rows:
        by.setdefault(short(name), []).append(d / 1e3)
    print("%-60s %6s %10s %10s %10s" % ("kernel", "calls", "min_us", "median_us", "max_us"))
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        if needle and needle not in name:
            continue
        v.sort()
        print("%-60s %6d %10.1f %10.1f %10.1f" % (name[:60], len(v), v[0], v[len(v) // 2], v[-1]))


def timeline(db_path, last=40):
    """the last `last` kernel dispatches in start order: start (us, relative to the first of them), duration, gap to
    the previous kernel's end -- where a single call's time goes between its kernels"""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    extra = ", stream_id" if "stream_id" in cols else (", queue_id" if "queue_id" in cols else "")
    rows = db.execute("select name, %s, %s%s from kernels order by %s" % (st, en, extra, st)).fetchall()[-last:]
    t0, prev_end = rows[0][1], rows[0][1]
    print("%-52s %10s %10s %10s %s" % ("kernel", "start_us", "dur_us", "gap_us", "queue"))
    for r in rows:
        name, a, b = r[0], r[1], r[2]
        print("%-52s %10.1f %10.1f %10.1f %s" % (short(name)[:52], (a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, r[3] if len(r) > 3 else ""))
        prev_end = max(prev_end, b)


def pmc_means(db_path):
    db = sqlite3.connect(db_path)
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration), max(vgpr_count), max(sgpr_count), "
         "max(lds_block_size), max(scratch_size) from counters_collection group by kernel_name, counter_name")
    return db.execute(q).fetchall()


def pmc_largest(db_path):
    """per kernel and counter: the launches whose counter value is within 2 % of the largest (a tool that runs several sizes:
    the figures of the largest one), count, mean value, mean duration"""
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
    groups = {}
    for name, ctr, val, dur in rows:
        groups.setdefault((name, ctr), []).append((val, dur))
    out = []
    for (name, ctr), lst in sorted(groups.items()):
        top = max(v for v, _ in lst)
        sel = [(v, d) for v, d in lst if v >= 0.98 * top]
        out.append((name, ctr, len(sel), sum(v for v, _ in sel) / len(sel), sum(d for _, d in sel) / len(sel)))
    return out


def pmc(paths):
    print("%-60s %-18s %6s %16s %12s %5s %5s %7s %7s" % ("kernel", "counter", "calls", "mean_value", "avg_us", "vgpr", "sgpr", "lds", "scratch"))
    for p in paths:
        for name, ctr, cnt, val, dur, vg, sg, lds, scr in pmc_means(p):
            if "lsa::" not in name:
                continue
            print("%-60s %-18s %6d %16.3f %12.3f %5d %5d %7d %7d" % (short(name)[:60], ctr, cnt, val, dur / 1e3, vg, sg, lds, scr))
    print("# the largest launches of each kernel (value within 2 % of its maximum)")
    print("%-60s %-18s %6s %16s %12s" % ("kernel", "counter", "calls", "mean_value", "avg_us"))
    for p in paths:
        for name, ctr, cnt, val, dur in pmc_largest(p):
            if "lsa::" not in name:
                continue
            print("%-60s %-18s %6d %16.3f %12.3f" % (short(name)[:60], ctr, cnt, val, dur / 1e3))


def hbm(fetch_db, write_db, needle, out):
    def mean(dbp, ctr):
        for name, c, cnt, val, dur, *_ in pmc_means(dbp):
            if needle in name and c == ctr:
                return val, cnt, dur
        raise SystemExit("kernel %r / counter %s not found in %s" % (needle, ctr, dbp))
    f, fc, fd = mean(fetch_db, "FETCH_SIZE")
    w, wc, wd = mean(write_db, "WRITE_SIZE")
    res = {
        "kernel": needle, "fetch_size_kib_mean": f, "write_size_kib_mean": w, "launches": [fc, wc],
        "kernel_us_under_pmc": [fd / 1e3, wd / 1e3],
        "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
        "note": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md HBM section); "
                "inputs fit the 256 MiB Infinity Cache, so memory-side counters can under-report re-reads",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    elif mode == "spread":
        spread(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    elif mode == "trace":
        trace(sys.argv[2])
    elif mode == "pmc":
        pmc(sys.argv[2:])
    elif mode == "hbm":
        hbm(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
