"""Idle time on the device inside the timed steps: union of the kernel intervals of a rocprofv3 --kernel-trace CSV against the span of a step
(a step = from a k_ibf_query_* start to the next one).  usage: python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
starts = [i for i, e in enumerate(ev) if "k_ibf_query" in e[2]]
for a, b in list(zip(starts, starts[1:]))[-4:]:
    seg = ev[a:b]
    t0, t1 = seg[0][0], ev[b][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    prev = seg[0][2]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    busy += cur_e - cur_s
    tail = t1 - cur_e
    print("step %.3f ms, device busy %.3f ms, idle %.3f ms (%d gaps; after the last kernel %.3f ms)" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), tail / 1e6))
    for g, p, n in sorted(gaps, reverse=True)[:6]:
        print("   %.1f us between %s and %s" % (g / 1e3, p[:45], n[:45]))
