#!/bin/bash
# kernel timeline of the two-context bench for a chunk count: tools/ab/timeline2ctx.sh <chunks>  -> gpurun_out/tl2ctx_<chunks>.txt
c=$1
out=$GRAFT_REPO_ROOT/gpurun_out/tl2ctx_$c
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S2K_DESC_CHUNKS=$c timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --mode hpc --contexts 2 --no-other-mode --no-cpu-baseline --no-end-to-end --verify-reads 0 > $out/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$out/*/*kernel_trace.csv")[0]
ev=[]
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    k='MIN' if 'tile_minimizer' in n else 'KMM' if 'desc_kminmer' in n else 'FIN' if 'finalize' in n else 'RT' if 'read_table' in n else None
    if k: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),k,r.get('Stream_Id','?')))
ev.sort()
mins=[e for e in ev if e[2]=='MIN']
t0=mins[-(4*int("$c")+1)][0]
with open("$GRAFT_REPO_ROOT/gpurun_out/tl2ctx_$c.txt","w") as o:
    o.write(open("$out/log.txt").read().strip().split("\n")[-1][:400]+"\n")
    for e in ev:
        if e[0]>=t0: o.write("%-4s start %9.1f end %9.1f dur %7.1f stream %s\n"%(e[2],(e[0]-t0)/1e3,(e[1]-t0)/1e3,(e[1]-e[0])/1e3,e[3]))
PY
