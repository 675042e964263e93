import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
R = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Stream_Id"], r["Thread_Id"], r["Kernel_Name"].split("(")[0][-34:]) for r in rows]
R.sort()
byq = collections.defaultdict(list)
for x in R: byq[x[2]].append(x)
q2 = max(byq.items(), key=lambda kv: sum(1 for x in kv[1] if "ln_mod" in x[5]))[0]
v = byq[q2]
gaps = [(v[i+1][0]-v[i][1], i) for i in range(len(v)-1)]
gaps.sort(reverse=True)
print("flow queue", q2, "streams on it:", collections.Counter(x[3] for x in v).most_common(6), "threads:", collections.Counter(x[4] for x in v).most_common(6))
for g, i in gaps[3:6]:
    a, b = v[i], v[i+1]
    print(f"\n gap {g/1e6:.1f} ms: after {a[5]} (stream {a[3]} thread {a[4]}) before {b[5]} (stream {b[3]} thread {b[4]})")
    # the next 6 dispatches on this queue
    for x in v[i+1:i+8]: print(f"     +{(x[0]-a[1])/1e6:7.2f} ms  {x[5]:36s} stream {x[3]} thread {x[4]} dur {(x[1]-x[0])/1e3:.1f} us")
    # what the other queues do during the hole
    for q, w in byq.items():
        if q == q2: continue
        inside = [x for x in w if x[0] >= a[1] and x[0] < b[0]]
        if inside:
            busy = sum(x[1]-x[0] for x in inside)
            print(f"     queue {q}: {len(inside)} dispatches inside the hole, {busy/1e6:.1f} ms in kernels; streams {collections.Counter(x[3] for x in inside).most_common(3)}")
