"""One train iteration as a timeline: start / end / duration (us), hardware queue and name of every kernel between two Adam launches, from a
rocprofv3 --kernel-trace csv:   python tools/iteration_timeline.py <..._kernel_trace.csv>   (profiles/r4_timeline_*.txt)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last adam_multi and print the iteration before it
idx = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q{r.get('Queue_Id','?'):>3} {name}")
