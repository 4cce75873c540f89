"""Per-kernel mean durations of each runner of `two_runners.py repeat N K` out of a rocprofv3 kernel trace (runners are separated by the long gaps of their
construction).   python tools/probe/two_runners_trace.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
groups, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 150e6:  # > 150 ms without a kernel: the next runner is being built
        groups.append(cur); cur = []
    cur.append(r); last_end = max(e, last_end or 0)
groups.append(cur)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^>]*>)?)", n); return (m.group(1) if m else n)[:44]
stats = []
for g in groups:
    d = collections.defaultdict(list)
    for r in g: d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    stats.append(d)
names = sorted({n for d in stats for n in d}, key=lambda n: -sum(sum(d.get(n, [])) for d in stats))[:16]
print(f"{'kernel':44s} " + " ".join(f"{'runner ' + str(k) + ' mean us (calls)':>26s}" for k in range(len(stats))))
for n in names:
    print(f"{n:44s} " + " ".join(f"{(sum(d[n]) / len(d[n]) if d.get(n) else 0):14.1f} ({len(d.get(n, [])):5d})    " for d in stats))
