"""ad-hoc: per-queue view of a rocprofv3 kernel trace (last window): what do the other queues do while kr_lac_scan_kernel runs alone?"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
best, cut, end = 0, 0, ev[0][1]
for i in range(1, len(ev)):
    gap = ev[i][0] - end
    if gap > best and i > len(ev) // 4: best, cut = gap, i
    end = max(end, ev[i][1])
ev = ev[cut:]
t0 = ev[0][0]
byq = collections.defaultdict(list)
for e in ev: byq[(e[3], e[4])].append(e)
for q, l in byq.items():
    busy = sum(e[1] - e[0] for e in l)
    print("queue/stream", q, "dispatches", len(l), "busy ms %.1f" % (busy / 1e6), "first %.2f last %.2f ms" % ((l[0][0] - t0) / 1e6, (l[-1][1] - t0) / 1e6))
# a slice of the timeline in the middle: 60 consecutive dispatches in start order with queue, start, duration
mid = len(ev) // 2
print("timeline slice (start ms, dur us, queue, kernel):")
for e in ev[mid - 60:mid + 200]:
    print("  %9.3f %9.3f %8.1f  q%s  %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e3, e[3], e[2]))
print("idle gaps > 120 us per queue: (gap end ms, gap us, prev kernel -> next kernel | kernel on another queue that ended within 8 us before the gap end)")
allev = ev
import bisect
ends = sorted((e[1], e[2], e[3]) for e in allev)
endt = [x[0] for x in ends]
cnt = collections.Counter()
for q, l in byq.items():
    for a, b in zip(l, l[1:]):
        gap = b[0] - a[1]
        if gap > 120000:
            i = bisect.bisect_right(endt, b[0])
            rel = [x for x in ends[max(0, i - 6):i] if x[2] != q[0] and b[0] - x[0] < 8000]
            key = (a[2], b[2], rel[-1][1] if rel else "-")
            cnt[key] += 1
            if cnt[key] <= 2: print("  q%s %9.3f %7.0f  %s -> %s | %s" % (q[0], (b[0] - t0) / 1e6, gap / 1e3, a[2], b[2], rel[-1][1] if rel else "-"))
print("gap patterns (prev -> next | released by): count")
for k, v in cnt.most_common(20): print("  %4d  %s -> %s | %s" % (v, k[0], k[1], k[2]))
