"""One mini-epoch of the update phase out of a rocprofv3 --kernel-trace CSV: start / end / duration (us) and queue of every kernel between two
consecutive optimiser launches (tail_adam_kernel, or optimizer_step_kernel; the last complete pair of the trace).  python tools/timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tail_adam_kernel" in r["Kernel_Name"] or r["Kernel_Name"].startswith("optimizer_step")]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:52]
for r in rows[a : b + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f} q{r['Queue_Id']:>2} {short(r['Kernel_Name'])}")
