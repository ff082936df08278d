"""Developer tool: reads a rocprofv3 kernel-trace CSV and reports how much the kernels of different streams / queues overlapped:
busy time (union of kernel intervals), sum of kernel durations, average concurrency, per-kernel mean duration."""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
if not rows:
    raise SystemExit("no kernel trace rows")
t0 = rows[0][0]
tot = sum(e - s for s, e, *_ in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, *_ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e, *_ in rows) - t0
print("kernels %d, span %.1f ms, busy (union) %.1f ms, sum of durations %.1f ms, average concurrency while busy %.2f" % (len(rows), span / 1e6, busy / 1e6, tot / 1e6, tot / busy))
per = collections.defaultdict(list)
for s, e, n, q, st in rows:
    per[n].append(e - s)
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  %-60s calls %5d mean %.3f ms max %.3f ms" % (n, len(v), sum(v) / len(v) / 1e6, max(v) / 1e6))
print("queues:", collections.Counter(q for *_, q, st in rows))
# a window of the timeline in the middle: who runs when
mid = rows[len(rows) // 2][0]
print("timeline around the middle (ms from there): start, duration, queue, kernel")
for s, e, n, q, st in rows:
    if mid <= s < mid + 12_000_000:
        print("  %8.3f %7.3f q%s %s" % ((s - mid) / 1e6, (e - s) / 1e6, q, n[:40]))
