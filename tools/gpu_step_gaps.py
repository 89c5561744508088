import csv, sys
rows=[]
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","")))
rows.sort()
t0=rows[0][0]
im=[i for i,r in enumerate(rows) if "im2col" in r[2]]
start=im[-8]; stop=im[-4]
cur_end=rows[start][1]
print("window %.2f .. %.2f ms" % ((rows[start][0]-t0)/1e6, (rows[stop][0]-t0)/1e6))
tot=0
for i in range(start+1,stop):
    s,e,n,q=rows[i]
    if s>cur_end:
        g=s-cur_end; tot+=g
        if g>60000:
            print("gap %7.1f us at %9.2f ms | before: %-46s | after: %-46s q=%s" % (g/1e3, (s-t0)/1e6, rows[i-1][2][:46], n[:46], q))
    cur_end=max(cur_end,e)
print("idle in window %.2f ms of %.2f ms" % (tot/1e6, (rows[stop][0]-rows[start][0])/1e6))
