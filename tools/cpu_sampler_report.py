#!/usr/bin/env python3
"""Resolves the samples of tools/cpu_sampler.c: cpu_sampler_report.py samples.txt [top]
Prints the leaf symbols by share, and for each the first frame inside libkslam_hip.so that called it."""
import bisect, collections, subprocess, sys
tables = {}


def table(obj):
    if obj not in tables:
        syms = []
        if obj != "?":
            for flags in (["-C"], ["-C", "-D"]):
                try:
                    for l in subprocess.run(["nm"] + flags + [obj], capture_output=True, text=True).stdout.splitlines():
                        p = l.split(" ", 2)
                        if len(p) == 3 and p[0] and p[1] in "tTwWiI":
                            syms.append((int(p[0], 16), p[2]))
                except OSError:
                    pass
        syms.sort()
        tables[obj] = ([a for a, _ in syms], syms)
    return tables[obj]


def name(frame):
    obj, off = frame.split()
    addrs, syms = table(obj)
    k = bisect.bisect_right(addrs, int(off, 16)) - 1
    return obj.rsplit("/", 1)[-1], (syms[k][1] if k >= 0 else "?")


top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
leaf = collections.Counter()
via = collections.defaultdict(collections.Counter)
total = 0
by_thread = collections.defaultdict(collections.Counter)
for line in open(sys.argv[1]):
    frames = [f for f in line.strip().split(" | ") if f]
    thread = "?"
    if frames and frames[0].startswith("@"):
        thread = frames.pop(0)[1:].strip()
    if not frames:
        continue
    total += 1
    names = [name(f) for f in frames]
    leaf[names[0]] += 1
    by_thread[thread][(names[0][0], names[0][1][:70])] += 1
    inside = next((n for n in names[1:] if n[0].startswith("libkslam")), None) if not names[0][0].startswith("libkslam") else None
    if inside is None and not names[0][0].startswith("libkslam"):   # no frame of ours: whatever named code the stack words point into
        others = [n for n in names[1:] if n[1] != "?" and n[0] != names[0][0]][:3]
        inside = ("", " < ".join("%s:%s" % (o[:12], f[:40]) for o, f in others)) if others else None
    via[names[0]][inside[1][:150] if inside else "-"] += 1
print("%d samples" % total)
for (o, s), c in leaf.most_common(top):
    print("%6.2f%%  %-22s %s" % (100.0 * c / total, o, s[:120]))
    for w, k in via[(o, s)].most_common(3):
        if w != "-":
            print("            %5.2f%% from %s" % (100.0 * k / total, w))
print()
for th, c in sorted(by_thread.items(), key=lambda kv: -sum(kv[1].values())):
    n = sum(c.values())
    print("thread %-16s %6.2f%%" % (th, 100.0 * n / total))
    for (o, f), k in c.most_common(6):
        print("      %5.2f%%  %-20s %s" % (100.0 * k / total, o[:20], f))
