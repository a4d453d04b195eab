#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <tag>
# For each scalar HashMode (hpc = the headline, regular):
#  1. rocprofv3 --kernel-trace --stats of the bench command            -> gpurun_out/prof_<tag>/<mode>/stats
#  2. separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ sets), kernel-trace only, as the pool requires -> .../pmc
# and a traffic.json with the HBM bytes of ALL kernels of a step (FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md, WRITE_SIZE exact).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in hpc regular; do
  args="--mode $mode --contexts 1 --no-other-mode --no-cpu-baseline --no-end-to-end --verify-reads 0"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$mode/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $args > $out/$mode.stats.log 2>&1 || echo "stats pass failed ($mode)"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$mode/pmc/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $args > $out/$mode.pmc_p$i.log 2>&1 || echo "pmc pass $i failed ($mode)"
  done
done
# the headline configuration itself (two chained contexts): kernel stats + trace, so that chunks x kernel time <= ms_per_step can be checked for `value`
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/hpc2ctx/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --mode hpc --contexts 2 --no-other-mode --no-cpu-baseline --no-end-to-end --verify-reads 0 > $out/hpc2ctx.stats.log 2>&1 || echo "stats pass failed (two contexts)"
# the same two commands unprofiled, on this box: what the profiler itself costs
python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --mode hpc --contexts 1 --no-other-mode --no-cpu-baseline --no-end-to-end --verify-reads 0 > $out/hpc.unprofiled.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --mode hpc --contexts 2 --no-other-mode --no-cpu-baseline --no-end-to-end --verify-reads 0 > $out/hpc2ctx.unprofiled.log 2>&1
python3 - <<PY
import csv, glob, json, collections, os
out = "$out"
res = {}
for mode in ("hpc", "regular"):
    # kernel stats
    rows = []
    for f in glob.glob(out + "/%s/stats/*/*kernel_stats.csv" % mode):
        rows = list(csv.DictReader(open(f)))
    with open(out + "/%s_kernel_stats.csv" % mode, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --mode %s --contexts 1 --no-other-mode --no-cpu-baseline --verify-reads 0\n" % mode)
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows:
            n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
            o.write('"%s",%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
    # counters per kernel, averaged per launch
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/%s/pmc/p*/*/*counter_collection.csv" % mode):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            n = n.replace("void s2k::", "").replace("s2k::", "")
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = []
    fetch = write = 0.0
    sq = {}
    per_kernel = {}
    for k in sorted(agg):
        vals = {c: sum(v) / len(v) for c, v in agg[k].items()}
        calls_per_step = len(agg[k].get("FETCH_SIZE", [])) / 3.0  # 1 warm-up + 2 steps per profiled run
        lines.append("== %s  (launches per step %.2f)" % (k, calls_per_step))
        for c in sorted(vals): lines.append("   %-24s %.6g" % (c, vals[c]))
        f_kb, w_kb = vals.get("FETCH_SIZE", 0.0), vals.get("WRITE_SIZE", 0.0)
        per_kernel[k] = {"launches_per_step": calls_per_step, "fetch_size_kb": f_kb, "write_size_kb": w_kb,
                         "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024)}
        if "synth_kernel" in k or "at::native" in k or "elementwise_kernel" in k: continue  # run once per process, not per step
        fetch += f_kb * calls_per_step; write += w_kb * calls_per_step
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            sq[c] = sq.get(c, 0.0) + vals.get(c, 0.0) * calls_per_step
    open(out + "/%s_pmc_summary.txt" % mode, "w").write("\n".join(lines) + "\n")
    import sys
    sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
    import bench  # kernel_source_id(): the state of the device sources these counters were collected on (bench.py refuses a stale file)
    res[mode] = {"mode": mode, "n_bases": 10000000000, "kernel_source_id": bench.kernel_source_id(), "hbm_bytes_per_step": int((2 * fetch + write) * 1024),
                 "fetch_size_kb_per_step": fetch, "write_size_kb_per_step": write, "kernels": per_kernel,
                 "sq_per_step": dict(sq, all_insts=sum(sq.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))),
                 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on python3 bench.py --steps 2 --warmup 1 --mode %s --no-other-mode; "
                           "sum over all kernels of one step (input generator and torch's own kernels left out); FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE exact; collected %s" % (mode, os.popen("date -u +%Y-%m-%dT%H:%MZ").read().strip())}
    print(mode, "HBM bytes per step", res[mode]["hbm_bytes_per_step"])
# two contexts: kernel stats and the bench line of the profiled run
rows = []
for f in glob.glob(out + "/hpc2ctx/stats/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
with open(out + "/hpc2ctx_kernel_stats.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 6 --warmup 2 --mode hpc --contexts 2 --no-other-mode --no-cpu-baseline --verify-reads 0\n")
    o.write("# (the one-context run of the same process comes first: (2 warm-up + 6 timed) calls x 6 chunks; then 2 warm-up and 6 timed calls x 3 chunks through two chained contexts)\n")
    try:
        o.write("# bench line of this run: " + [l for l in open(out + "/hpc2ctx.stats.log") if l.startswith("{")][-1][:600] + "\n")
    except Exception:
        pass
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
        o.write('"%s",%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
# ---- round 6: statistics over the TIMED launches only (the warm-up launches of a run are slower and do not belong in the average), the kernel timeline of
#      the last two chained calls, and the profiler's own overhead (the same command unprofiled, on the same box, see below)
def trace_rows(d):
    ev = []
    for f in glob.glob(out + "/%s/stats/*/*kernel_trace.csv" % d):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void s2k::", "").replace("s2k::", "")
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    return sorted(ev)
def bench_line(log):
    try:
        return json.loads([l for l in open(log) if l.startswith("{")][-1])
    except Exception:
        return None
with open(out + "/timed_kernel_stats.txt", "w") as o:
    for d, steps, warm in (("hpc", 5, 2), ("regular", 5, 2), ("hpc2ctx", 6, 2)):
        ev = trace_rows(d)
        bl = bench_line(out + "/%s.stats.log" % d)
        o.write("== %s: rocprofv3 --kernel-trace of bench.py --steps %d --warmup %d; launches of the TIMED steps only (the last %d of %d calls of a kernel per call)\n" % (d, steps, warm, steps, steps + warm))
        if bl:
            o.write("   bench line of the profiled run: ms_per_step %.3f (value %.1f), one_context %s\n" % (bl["ms_per_step"], bl["value"], bl.get("one_context")))
        by = collections.defaultdict(list)
        for st, en, n in ev:
            if "tile_minimizer" in n or "desc_kminmer" in n or "desc_scan" in n or "read_table" in n or "finalize" in n:
                by[n].append((st, en))
        for n in sorted(by):
            L = by[n]
            if d == "hpc2ctx":  # the one-context leg comes first in this process: (steps + warm) calls; then the chained legs
                per1 = len(L) // 1  # (reported as they are: the chained calls are the LAST steps x chunks launches)
            k = len(L) * steps // (steps + warm) if d != "hpc2ctx" else None
            if d == "hpc2ctx":
                # one-context calls: 6 chunks each, (steps + warm) of them; chained: 3 chunks each, (steps + warm') of them -- take the last steps x 3
                chunks2 = 3 if "tile_minimizer" in n or "desc_kminmer" in n else (15 if "desc_scan" in n else 1)
                k = min(len(L), steps * chunks2)
            T = L[-k:] if k else L
            durs = [(en - st) / 1e3 for st, en in T]
            o.write("   %-60s timed launches %4d  avg %9.1f us  min %9.1f  max %9.1f  (all %d launches avg %.1f us)\n" % (
                n[-60:], len(T), sum(durs) / len(durs), min(durs), max(durs), len(L), sum((en - st) / 1e3 for st, en in L) / len(L)))
    # timeline of the last two chained calls
    ev = trace_rows("hpc2ctx")
    mins = [e for e in ev if "tile_minimizer" in e[2]]
    if len(mins) >= 7:
        t0 = mins[-6][0]
        o.write("== hpc2ctx: kernel timeline of the last two chained calls (us from the first minimizer kernel of the second-to-last call)\n")
        for st, en, n in ev:
            if st >= t0 and ("tile_minimizer" in n or "desc_kminmer" in n or "finalize" in n or "read_table" in n):
                tag = "MIN" if "tile_minimizer" in n else "KMM" if "desc_kminmer" in n else "FIN" if "finalize" in n else "RT "
                o.write("   %s start %9.1f end %9.1f dur %8.1f\n" % (tag, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3))
    # profiler overhead: the unprofiled runs of the same commands (written by the shell below)
    for d in ("hpc", "hpc2ctx"):
        a, b = bench_line(out + "/%s.stats.log" % d), bench_line(out + "/%s.unprofiled.log" % d)
        if a and b:
            o.write("== profiler overhead (%s): ms_per_step %.3f under rocprofv3 --kernel-trace --stats, %.3f unprofiled on the same box (one_context %s / %s)\n" % (
                d, a["ms_per_step"], b["ms_per_step"], a.get("one_context", {}).get("ms_per_step"), b.get("one_context", {}).get("ms_per_step")))
json.dump(res["hpc"], open(out + "/traffic_hpc.json", "w"), indent=1)
json.dump(res["regular"], open(out + "/traffic_regular.json", "w"), indent=1)
PY
cat $out/hpc_kernel_stats.csv; cat $out/hpc_pmc_summary.txt | head -60; cat $out/timed_kernel_stats.txt
