"""Where does a Tracking step wait when other host threads use the GPU?  Reads the rocpd database of
`rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -- python -m pytest tests/test_gpu_concurrency.py` (tools/trace_concurrency.sh)
and prints, for the Tracking thread: the dispatch delay of its kernels (end of hipLaunchKernel -> start on the GPU), its longest HIP
API calls with what the other threads and the GPU were doing meanwhile, and the longest gaps between its API calls.
Round 4 found the 7-10 ms outliers with it: hipMemcpyAsync calls blocking two threads at once, GPU idle."""
import sqlite3, sys, collections
c=sqlite3.connect(sys.argv[1])
kc=[r[1] for r in c.execute("pragma table_info(kernels)")]; kx={n:i for i,n in enumerate(kc)}
rc=[r[1] for r in c.execute("pragma table_info(regions)")]; rx={n:i for i,n in enumerate(rc)}
K=list(c.execute("select * from kernels order by start"))
R=list(c.execute("select * from regions order by start"))
api={}
for r in R:
    if r[rx['name']].startswith('hipLaunchKernel') or r[rx['name']].startswith('hipExtLaunch') or r[rx['name']].startswith('hipModuleLaunch'): api.setdefault(r[rx['stack_id']], r)
nm=lambda k:k[kx['name']].split('(')[0].replace('orbhip::','')
# tracker thread = the tid that launched k_trk_greedy most
trk_tid=sorted(set(k[kx['tid']] for k in K if 'k_trk_greedy' in nm(k)))[1]
rows=[]
for k in K:
    if k[kx['tid']]!=trk_tid: continue
    a=api.get(k[kx['stack_id']])
    if not a: continue
    rows.append((k[kx['start']]-a[rx['end']], a[rx['end']]-a[rx['start']], nm(k), a[rx['name']], k[kx['start']], a[rx['start']]))
rows.sort(reverse=True)
print("tracker tid", trk_tid, "launches", len(rows))
for d,ad,n,an,ks,as_ in rows[:12]:
    # what were other threads doing (API calls overlapping [api end, kernel start])?
    ov=[r for r in R if r[rx['tid']]!=trk_tid and r[rx['start']]<ks and r[rx['end']]>as_ and (r[rx['end']]-r[rx['start']])>200000]
    print("dispatch delay %.0f us (api %s took %.0f us) kernel %s | other threads' long API calls meanwhile: %s" % (d/1e3, an, ad/1e3, n, [(r[rx['tid']], r[rx['name']], round((r[rx['end']]-r[rx['start']])/1e3)) for r in ov][:6]))
T=[r for r in R if r[rx['tid']]==trk_tid and not r[rx['name']].startswith('__hip')]
T.sort(key=lambda r:r[rx['start']])
t_first=T[0][rx['start']]
longest=sorted([r for r in T if r[rx['start']]>t_first], key=lambda r:-(r[rx['end']]-r[rx['start']]))[:8]
print("tracker thread API span %.1f ms, %d calls" % ((T[-1][rx['end']]-T[0][rx['start']])/1e6, len(T)))
print("longest API calls of the tracker thread:", [(r[rx['name']], round((r[rx['end']]-r[rx['start']])/1e3), "at %.1f ms" % ((r[rx['start']]-T[0][rx['start']])/1e6)) for r in longest])
gaps=[]
for a,b in zip(T[:-1],T[1:]):
    if a[rx['start']]>t_first: gaps.append((b[rx['start']]-a[rx['end']], a[rx['name']], b[rx['name']], a[rx['end']], b[rx['start']]))
gaps.sort(reverse=True)
for g,an,bn,t0,t1 in gaps[:8]:
    ov=[r for r in R if r[rx['tid']]!=trk_tid and r[rx['start']]<t1 and r[rx['end']]>t0 and (r[rx['end']]-r[rx['start']])>100000 and not r[rx['name']].startswith('__hip')]
    print("host gap %.0f us at %.1f ms between %s and %s; other threads meanwhile: %s" % (g/1e3, (t0-T[0][rx['start']])/1e6, an, bn, [(r[rx['tid']], r[rx['name']], round((r[rx['end']]-r[rx['start']])/1e3)) for r in ov][:5]))
for r in longest[:4]:
    t0,t1=r[rx['start']],r[rx['end']]
    if t1-t0<1e6 or (t0-T[0][rx['start']])<2e7: continue
    ov=[q for q in R if q[rx['tid']]!=trk_tid and q[rx['start']]<t1 and q[rx['end']]>t0 and not q[rx['name']].startswith('__hip') and (q[rx['end']]-q[rx['start']])>50000]
    print("DURING tracker %s (%.1f ms):" % (r[rx['name']], (t1-t0)/1e6), [(q[rx['tid']], q[rx['name']], round((q[rx['end']]-q[rx['start']])/1e3), "starts %+.1f ms" % ((q[rx['start']]-t0)/1e6)) for q in ov][:12])
    kk=[k for k in K if k[kx['start']]<t1 and k[kx['end']]>t0]
    print("   kernels running meanwhile:", collections.Counter(nm(k) for k in kk).most_common(8))
    mc=[m for m in c.execute("select tid,name,size,start,end from memory_copies where start<? and end>?", (t1,t0))]
    print("   copies meanwhile:", [(m[0], m[1], m[2], round((m[4]-m[3])/1e3)) for m in mc][:8])
