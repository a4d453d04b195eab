#!/usr/bin/env python3
"""Static instruction histogram of the tiled kernel per phase, from an ISA listing built with -DS2K_ISA_MARKS (the phase stamps of
s2k_tile_impl.h become `; S2K_PHASE n` comments): instructions are attributed to the last mark seen in text order -- an
approximation (loops and cold paths count once each), good enough to see where the instructions of the common path sit.
usage: phase_hist.py <listing.s> <kernel-substring, e.g. ILi31ELb1ELb1E>"""
import collections
import re
import sys

NAMES = {0: "staging(top)", 13: "c.marks", 14: "c.flags+scan", 15: "c.stores", 1: "c.halo+rest", 2: "hash loop", 3: "boundaries", 4: "counts+word+list",
         12: "rederive", 8: "r.list read", 9: "r.backmap", 10: "r.stores(legacy)", 5: "rounds(rest)", 6: "tail", -1: "prologue"}
path, kern = sys.argv[1], sys.argv[2]
infn, cur = False, -1
hist = collections.defaultdict(collections.Counter)
order = []
for ln in open(path):
    s = ln.strip()
    if re.match(r"_Z\S+:", s):
        infn = kern in s and "tile_minimizer_kernel" in s
        cur = -1
        continue
    if not infn:
        continue
    if s.startswith(".Lfunc_end"):
        infn = False
        continue
    m = re.match(r"; S2K_PHASE (\d+)", s)
    if m:
        cur = int(m.group(1))
        continue
    if not s or s[0] in ".;" or re.match(r"\S+:", s):
        continue
    op = s.split()[0]
    kind = ("readlane" if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) else "valu" if op.startswith("v_") else
            "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "salu" if op.startswith("s_") else
            "lds" if op.startswith("ds_") else "vmem" if op.split("_")[0] in ("global", "scratch", "buffer", "flat") else "other")
    if cur not in hist:
        order.append(cur)
    hist[cur][kind] += 1
tot = collections.Counter()
kinds = ("valu", "readlane", "lds", "vmem", "salu", "branch", "wait")
print("%-20s " % "phase (after mark)" + " ".join("%8s" % k for k in kinds))
for ph in order:
    c = hist[ph]
    print("%-20s " % NAMES.get(ph, str(ph)) + " ".join("%8d" % c[k] for k in kinds))
    tot.update(c)
print("%-20s " % "total" + " ".join("%8d" % tot[k] for k in kinds))
