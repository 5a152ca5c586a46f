import sqlite3, sys, collections
c=sqlite3.connect(sys.argv[1])
rows=list(c.execute("select name,start,end from kernels order by start"))
# take the last third (last solve) by splitting on k_ba_iter_begin count
acc=collections.defaultdict(lambda:[0,0])
for n,s,e in rows:
    k=n.split('(')[0]; acc[k][0]+=1; acc[k][1]+=e-s
tot=sum(v[1] for v in acc.values())
span=rows[-1][2]-rows[0][1]
print("kernels total busy ms %.2f  span ms %.2f"%(tot/1e6,span/1e6))
for k,v in sorted(acc.items(), key=lambda x:-x[1][1]): print("%-40s n=%-6d tot %.3f ms avg %.1f us"%(k,v[0],v[1]/1e6,v[1]/v[0]/1e3))
