import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split('(')[0][-40:]) for r in rows)
segs=[];cur=[iv[0]];ce=iv[0][1]
for x in iv[1:]:
    if x[0]>ce+150000: segs.append(cur);cur=[]
    cur.append(x);ce=max(ce,x[1])
segs.append(cur)
for s in segs:
    w=max(e for _,e,_ in s)-s[0][0]
    if w<3e6: continue
    busy=0;cs,ce=s[0][0],s[0][1]
    for a,b,_ in s[1:]:
        if a>ce: busy+=ce-cs;cs,ce=a,b
        else: ce=max(ce,b)
    busy+=ce-cs
    ks=sum(b-a for a,b,_ in s)
    print("segment: %d kernels wall %.3f ms busy %.3f idle %.3f (%.1f%%) ksum %.3f"%(len(s),w/1e6,busy/1e6,(w-busy)/1e6,100*(w-busy)/w,ks/1e6))
    d=collections.defaultdict(lambda:[0,0])
    for a,b,n in s: d[n][0]+=1;d[n][1]+=b-a
    for n,(c,t) in sorted(d.items(),key=lambda x:-x[1][1])[:12]: print("   %-42s %4d avg %.1f us  total %.2f ms"%(n,c,t/c/1e3,t/1e6))
