#!/usr/bin/env python3
"""Checks the hand-counted `s_waitcnt vmcnt(n)` of the tiled kernel against the ISA the compiler emitted.

The kernel loads the next tile with LDS-DMA (global_load_lds_dwordx4, inline asm) and, at the top of the next iteration, waits
with vmcnt(n), n = stores_per_round() x (rounds of the last batch of hits; 1 store per round on the descriptor path, 3 on the
legacy one): the vector-memory operations issued AFTER the DMA
loads.  vmcnt counts loads, LDS-DMA loads and stores in order, so MORE operations after the DMA than counted only make the
wait longer; FEWER would let it return before the tile has landed (stale LDS, silently wrong minimizers).  The kernel marks
the body of its one-lane-per-hit round loop with `; S2K_MARK round_begin / round_end` comments; this script counts the
vector-memory instructions between them in every instantiation and fails if there are fewer than the kernel assumes.

usage: check_vmcnt.py <kernel.s> [...]   (files made by `make -C rust-seq2kminmers_amd/csrc isa-check`)
Validated against: ROCm 7.2.0 (hipcc / AMD clang of /opt/rocm-7.2.0), gfx950."""
import re
import sys

VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)(store|load|atomic)")


def check(path):
    """The round loop is found structurally: the innermost backward branch whose span contains the `round_begin many=0`
    marker (the compiler rotates the loop, so the markers are not in program order in the text)."""
    lines = open(path).read().split("\n")
    funcs = []  # (name, first line, last line)
    cur = None
    for i, ln in enumerate(lines):
        s = ln.strip()
        m = re.match(r"(_Z\S*tile_minimizer_kernel\S*):", s)
        if m:
            cur = [m.group(1), i, None]
        elif s.startswith(".Lfunc_end") and cur:
            cur[2] = i
            funcs.append(tuple(cur))
            cur = None
    errors, report = [], []
    if not funcs:
        errors.append("%s: no tile_minimizer_kernel instantiation found" % path)
    for name, a, b in funcs:
        short = re.search(r"tile_minimizer_kernelILi(\d+)ELb(\d)ELb(\d)", name)
        tag = "L=%s hpc=%s desc=%s" % short.groups() if short else name
        labels, branches, marks, need, dma, waits = {}, [], [], None, 0, set()
        for i in range(a, b):
            s = lines[i].strip()
            m = re.match(r"(\.LBB\d+_\d+):", s)
            if m:
                labels[m.group(1)] = i
            m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", s)
            if m:
                branches.append((i, m.group(1)))
            m = re.search(r"S2K_MARK (\w+) (\w+)=(\d+)", s)
            if m:
                if m.group(1) == "round_begin" and m.group(3) == "0":
                    marks.append(i)
                if m.group(1) == "counted_wait":
                    need = int(m.group(3))
            if "global_load_lds_dwordx4" in s:
                dma += 1
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)$", s)
            if m:
                waits.add(int(m.group(1)))
        if need is None:
            errors.append("%s: no counted_wait marker" % tag)
            continue
        if dma < 10:
            errors.append("%s: %d LDS-DMA loads, expected >= 10" % (tag, dma))
        for n in (need, 2 * need, 3 * need, 4 * need):
            if n not in waits:
                errors.append("%s: s_waitcnt vmcnt(%d) missing" % (tag, n))
        if not marks:
            errors.append("%s: round loop (many=0) not marked" % tag)
        for mk in marks:
            spans = [(bl - labels[t], labels[t], bl) for bl, t in branches if t in labels and labels[t] <= mk <= bl]
            if not spans:
                errors.append("%s: no loop around the round_begin marker" % tag)
                continue
            _, lo, hi = min(spans)  # innermost loop header around the marker ...
            hi = max(bl for bl, t in branches if t in labels and labels[t] == lo)  # ... and its last back edge
            ops = [lines[i].split()[0] for i in range(lo, hi + 1) if VMEM.match(lines[i])]
            stores = [o for o in ops if "store" in o or "atomic" in o]
            report.append("%s: round loop = lines %d..%d, %d vector-memory ops per round (%s), kernel counts %d" % (
                tag, lo + 1, hi + 1, len(ops), " ".join(ops), need))
            if len(stores) < need:
                errors.append("%s: only %d vector-memory stores in the round loop, the counted wait assumes %d" % (tag, len(stores), need))
            if any("load" in o for o in ops):
                errors.append("%s: a vector-memory LOAD inside the round loop (it makes the compiler drain the stores)" % tag)
    return errors, report


def main():
    bad = []
    for p in sys.argv[1:]:
        e, r = check(p)
        for x in r:
            print(x)
        bad += e
    for x in bad:
        print("ERROR:", x, file=sys.stderr)
    return 1 if bad or len(sys.argv) < 2 else 0


if __name__ == "__main__":
    sys.exit(main())
