import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[]
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
# find steps by the embed_fwd kernel
emb=[i for i,r in enumerate(rows) if "embed_fwd" in r[3]]
lo,hi=emb[-2],emb[-1]
t0=rows[lo][0]
print("step span %.1f us"%((rows[hi][0]-t0)/1e3))
n=int(sys.argv[2]) if len(sys.argv)>2 else 120
for r in rows[lo-8:lo+n]:
    print("%9.1f %9.1f q%d %7.1f %s"%((r[0]-t0)/1e3,(r[1]-t0)/1e3,r[2],(r[1]-r[0])/1e3,r[3][:56]))
