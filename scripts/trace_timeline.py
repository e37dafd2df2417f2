"""Timeline of the last full step in a rocprofv3 --kernel-trace CSV (a step = from one k_ibf_query* start to the next): every kernel with
its start offset, duration, queue and the idle gap in front of it on the whole device.  usage: python scripts/trace_timeline.py <kernel_trace.csv> [step_from_end [marker_kernel]]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marker = sys.argv[3] if len(sys.argv) > 3 else "k_ibf_query"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")) for r in rows), key=lambda x: x[0])
starts = [i for i, e in enumerate(ev) if marker in e[2]]
back = min(back, len(starts) - 1)
a, b = starts[-back - 1], starts[-back]
seg = ev[a:b]
t0 = seg[0][0]
end = t0
print("step of %.3f ms, %d kernels" % ((ev[b][0] - t0) / 1e6, len(seg)))
for s, e, n, q in seg:
    gap = s - end
    print("%9.1f us  +%7.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n, ("   [idle %.1f us before]" % (gap / 1e3)) if gap > 3000 else ""))
    end = max(end, e)
print("tail idle %.1f us" % ((ev[b][0] - end) / 1e3))
