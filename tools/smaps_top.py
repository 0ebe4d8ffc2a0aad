#!/usr/bin/env python3
"""Largest resident mappings of a finished `fade annotate -T` run: FADE_SMAPS_DUMP=path makes the driver copy
/proc/self/smaps there just before it returns; this prints the top entries."""
import sys
rows, cur = [], None
for line in open(sys.argv[1]):
    f = line.split()
    if "-" in f[0] and len(f) >= 5 and ":" not in f[0]:
        cur = [0, line.strip()]
        rows.append(cur)
    elif f[0] == "Rss:" and cur:
        cur[0] = int(f[1])
rows.sort(reverse=True)
print("total resident %d MB in %d mappings" % (sum(r[0] for r in rows) >> 10, len(rows)))
for kb, name in rows[:25]:
    print("%8d MB  %s" % (kb >> 10, name))
if len(sys.argv) > 2:  # the whole entry of the largest mapping
    want, on = rows[0][1], False
    for line in open(sys.argv[1]):
        if line.strip() == want:
            on = True
        elif on and "-" in line.split()[0] and ":" not in line.split()[0]:
            break
        if on:
            print("   ", line.rstrip())
