"""Timeline of ONE isolated query (N = 1) from a rocprofv3 kernel trace: per kernel its duration and the GAP to the previous kernel's
end, averaged over the last 20 queries.    python tools/experiments/latency_timeline.py <results.db>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60] for r in rows]
# a query = the run of kernels from one leg_front_kernel to the next
starts = [i for i, n in enumerate(names) if n.startswith("leg_front")]
qs = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-20:]
n = qs[0][1] - qs[0][0]
qs = [q for q in qs if q[1] - q[0] == n]
acc = [[0.0, 0.0] for _ in range(n)]
tot = 0.0
for a, b in qs:
    for k in range(n):
        r = rows[a + k]
        acc[k][0] += (r[2] - r[1]) / 1e3
        acc[k][1] += (r[1] - rows[a + k - 1][2]) / 1e3
    tot += (rows[b][1] - rows[a][1]) / 1e3
print("%d queries of %d kernels; period %.1f us" % (len(qs), n, tot / len(qs)))
for k in range(n):
    print("  %-60s dur %7.1f us   gap before %6.1f us" % (names[qs[0][0] + k], acc[k][0] / len(qs), acc[k][1] / len(qs)))
print("  sum of durations %.1f us, sum of gaps %.1f us" % (sum(a[0] for a in acc) / len(qs), sum(a[1] for a in acc) / len(qs)))
