#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs into small text/JSON files for profiles/.

usage: rocprof_summary.py --trace <kernel-trace.db> [--fetch <pmc FETCH_SIZE .db>] [--write <pmc WRITE_SIZE .db>]
                          --out profiles/<name>   (writes <name>.txt and <name>.json)

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes (they do not fit one pass), are reported in KiB, and on gfx950
FETCH_SIZE counts 128-byte requests of a wide coalesced stream as 64 bytes, so the read side is doubled.
"""
import argparse
import json
import sqlite3


def kernels_from_trace(path):
    cur = sqlite3.connect(path).cursor()
    # only the full-size launches of each kernel (bench.py also runs a one-frame priming batch)
    rows = cur.execute("select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), "
                       "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                       "from kernels k join (select name n, max(grid_x) g from kernels group by name) m on k.name = m.n and k.grid_x = m.g "
                       "group by name order by sum(duration) desc").fetchall()  # (a join, not a correlated subquery: minutes -> seconds)
    out = {}
    for r in rows:
        out[r[0]] = dict(calls=r[1], avg_us=r[2] / 1e3, min_us=r[3] / 1e3, max_us=r[4] / 1e3, vgpr=r[5], agpr=r[6], sgpr=r[7],
                         lds=r[8], scratch=r[9], grid=r[10], workgroup=r[11])
    return out


def counter_avg(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, avg(value), count(*) from counters_collection c join "
                       "(select kernel_name n, max(grid_size_x) g from counters_collection group by kernel_name) m "
                       "on c.kernel_name = m.n and c.grid_size_x = m.g where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--insts", help="pmc pass with SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES")
    ap.add_argument("--out", required=True)
    ap.add_argument("--prefix", default="k_", help="only kernels whose name starts with this prefix")
    ap.add_argument("--note", default="")
    ap.add_argument("--traffic-out", help="write the per-kernel HBM bytes in the shape of profiles/traffic.json (bench.py reads it)")
    ap.add_argument("--build", default=None, help="nvh_version() of the library that was profiled (bench.py compares it with its own)")
    ap.add_argument("--calibration-from", default=None, help="an older traffic.json whose `calibration` block is carried over")
    a = ap.parse_args()
    ks = {k: v for k, v in kernels_from_trace(a.trace).items() if k.startswith(a.prefix)}
    fetch = counter_avg(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = counter_avg(a.write, "WRITE_SIZE") if a.write else {}
    for k, v in ks.items():
        if k in fetch:
            v["fetch_size_kib_raw"] = fetch[k][0]
            v["hbm_read_bytes"] = fetch[k][0] * 1024 * 2  # gfx950 FETCH_SIZE correction (x2)
        if k in write:
            v["write_size_kib_raw"] = write[k][0]
            v["hbm_write_bytes"] = write[k][0] * 1024
        if "hbm_read_bytes" in v and "hbm_write_bytes" in v:
            v["hbm_bytes"] = v["hbm_read_bytes"] + v["hbm_write_bytes"]
    if a.insts:
        for cname in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAVES"):
            for k, (val, _) in counter_avg(a.insts, cname).items():
                if k in ks:
                    ks[k][cname.lower()] = val
    lines = ["# rocprofv3 summary  " + a.note,
             "%-22s %6s %10s %10s %10s %5s %5s %7s %8s %14s %14s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "vgpr", "sgpr",
                                                                      "lds_B", "grid", "hbm_read_B", "hbm_write_B")]
    for k, v in ks.items():
        lines.append("%-22s %6d %10.3f %10.3f %10.3f %5d %5d %7d %8d %14s %14s" % (
            k, v["calls"], v["avg_us"], v["min_us"], v["max_us"], v["vgpr"], v["sgpr"], v["lds"], v["grid"],
            ("%.0f" % v["hbm_read_bytes"]) if "hbm_read_bytes" in v else "-",
            ("%.0f" % v["hbm_write_bytes"]) if "hbm_write_bytes" in v else "-"))
    if a.insts:
        lines.append("")
        lines.append("%-22s %14s %14s %14s %16s" % ("kernel", "insts_valu", "insts_salu", "insts_lds", "wave_cycles(x4)"))
        for k, v in ks.items():
            lines.append("%-22s %14.0f %14.0f %14.0f %16.0f" % (k, v.get("sq_insts_valu", 0), v.get("sq_insts_salu", 0),
                                                                 v.get("sq_insts_lds", 0), v.get("sq_wave_cycles", 0)))
    if a.traffic_out:
        tj = {"source": a.note, "build": a.build,
              "kernels": {k: {kk: v[kk] for kk in ("avg_us", "hbm_bytes", "hbm_read_bytes", "hbm_write_bytes") if kk in v} for k, v in ks.items()}}
        if a.calibration_from:
            try:
                tj["calibration"] = json.load(open(a.calibration_from)).get("calibration")
            except Exception:
                pass
        json.dump(tj, open(a.traffic_out, "w"), indent=1, sort_keys=True)
    open(a.out + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(ks, open(a.out + ".json", "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
