"""Condense the rocprofv3 output of tools/profile_bench.sh into the small files kept under
profiles/: per-kernel duration statistics with the warm-up launches EXCLUDED, HBM traffic per
launch from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction: FETCH_SIZE x 2 for wide
coalesced streams, MI355X_MICROARCH.md section HBM), and the SQ counters per launch.

    python3 tools/rocprof_summary.py <prof dir> <tag> <warm-up launches in the trace run>
"""
import csv, glob, json, os, sys
import numpy as np


def find(root, suffix):
    hits = sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def main():
    root, tag, warm = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = os.path.join(root, "summary")
    os.makedirs(out, exist_ok=True)
    lines = []
    # ---- kernel trace: durations per kernel, first `warm` launches of each dropped
    trace = find(os.path.join(root, "trace"), "kernel_trace.csv")
    dom = None
    if trace:
        per = {}
        meta = {}
        for r in csv.DictReader(open(trace)):
            k = r["Kernel_Name"]
            per.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[k] = (r["VGPR_Count"], r.get("Accum_VGPR_Count", "0"), r["SGPR_Count"], r["Scratch_Size"],
                       r["LDS_Block_Size"], r["Workgroup_Size_X"], r["Grid_Size_X"])
        dom = max(per, key=lambda k: sum(per[k]))
        lines.append("# rocprofv3 --kernel-trace --stats, %s; durations in ns; the first %d launches of the dominant kernel"
                     " (warm-up: clock ramp, first-touch) are EXCLUDED from its timed row" % (tag, warm))
        lines.append("kernel,calls,avg_ns,median_ns,min_ns,max_ns,std_ns,vgpr,agpr,sgpr,scratch_bytes,lds_bytes,wg_size,grid")
        for k in sorted(per, key=lambda k: -sum(per[k])):
            d = np.array(per[k][warm:] if k == dom and len(per[k]) > warm else per[k], dtype=np.float64)
            lines.append('"%s",%d,%.1f,%.1f,%d,%d,%.1f,%s' % (k, len(d), d.mean(), np.median(d), d.min(), d.max(), d.std(),
                                                             ",".join(meta[k])))
        allc = np.array(per[dom], dtype=np.float64)
        lines.append('# all %d launches of the dominant kernel incl. warm-up: avg %.1f ns' % (len(allc), allc.mean()))
        bj = os.path.join(root, "bench_under_rocprof.json")
        if os.path.exists(bj):
            try:
                b = json.loads([l for l in open(bj) if l.startswith("{")][-1])
                lines.append("# bench.py line printed in the same run: kernel_ms %.6f (HIP events over the %d timed launches), "
                             "ms_per_step %.6f" % (b["roofline"]["kernel_ms"], b["steps"], b["ms_per_step"]))
            except Exception as e:
                lines.append("# (bench line unreadable: %s)" % e)
        open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w").write("\n".join(lines) + "\n")
    # ---- PMC passes
    def pmc(sub):
        f = find(os.path.join(root, sub), "counter_collection.csv")
        vals = {}
        if not f:
            return vals
        for r in csv.DictReader(open(f)):
            if dom and r["Kernel_Name"] != dom:
                continue
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        return {k: float(np.median(v[2:] if len(v) > 4 else v)) for k, v in vals.items()}
    fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
    if fetch or write:
        fk, wk = fetch.get("FETCH_SIZE"), write.get("WRITE_SIZE")
        tj = {"round": os.environ.get("GP_ROUND", "round 3"), "kernel": dom,
              "comment": "HBM bytes per launch, rocprofv3 PMC, one counter per pass, median over the launches after 2 "
                         "warm-ups. FETCH_SIZE is in KiB and on gfx950 reads half the bytes of a coalesced stream "
                         "(MI355X_MICROARCH.md, HBM): x 1024 x 2.  WRITE_SIZE reads exactly: x 1024.",
              "fetch_size_kib": fk, "write_size_kib": wk,
              "fetch_bytes_corrected": int(fk * 2048) if fk else None,
              "write_bytes": int(wk * 1024) if wk else None,
              "hbm_bytes_per_launch": int(fk * 2048 + wk * 1024) if fk and wk else None}
        json.dump(tj, open(os.path.join(out, "traffic_%s.json" % tag), "w"), indent=1)
    if sq:
        with open(os.path.join(out, "%s_pmc_sq.txt" % tag), "w") as fh:
            fh.write("# SQ counters per launch of %s (median over launches after 2 warm-ups)\n" % dom)
            for k in sorted(sq):
                fh.write("%s %.6g\n" % (k, sq[k]))
    print("\n".join(lines[:8]))
    print(json.dumps({"fetch": fetch, "write": write, "sq": sq}))


if __name__ == "__main__":
    main()
