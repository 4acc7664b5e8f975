import csv, sys, collections
f=sys.argv[1]
rows=list(csv.DictReader(open(f)))
by=collections.defaultdict(list)
for r in rows:
    name=r["Kernel_Name"].split("(")[0]
    if "k_trdb" in name:
        dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        by[(name,int(r["Grid_Size_Y"]))].append((int(r["Grid_Size_X"]),dur))
for k,v in sorted(by.items()):
    v.sort()
    n=len(v)
    print(k, "n=",n)
    for q in (0,n//4,n//2,3*n//4,n-1):
        print("   grid_x=%d dur=%.2f us"%v[q])
# gaps between consecutive kernels on the stream
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
gaps=[]
for a,b in zip(rows[:-1],rows[1:]):
    if "k_trdb" in a["Kernel_Name"] and "k_trdb" in b["Kernel_Name"]:
        gaps.append((int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3)
import statistics
print("gaps: n=%d mean=%.2f median=%.2f max=%.2f"%(len(gaps),statistics.mean(gaps),statistics.median(gaps),max(gaps)))
