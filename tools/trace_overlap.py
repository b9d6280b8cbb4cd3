import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-4000:]
ov=0; tot=0; pairs={}
prev_end=0; prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    tot+=e-s
    if s<prev_end:
        o=min(e,prev_end)-s; ov+=o
        k=(prev["Kernel_Name"][:50], r["Kernel_Name"][:50]); pairs[k]=pairs.get(k,0)+o
    if e>prev_end: prev_end=e; prev=r
print("busy ms",tot/1e6,"overlap ms",ov/1e6, "queues", set(r.get("Queue_Id") for r in rows), "streams", set(r.get("Stream_Id") for r in rows))
for k,v in sorted(pairs.items(), key=lambda kv:-kv[1])[:8]: print(v/1e3, k)
