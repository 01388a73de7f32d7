import sys, re
a=[0,0,0]; n=0
for l in sys.stdin:
    m=re.search(r"host prep ([\d.]+) ms, enqueue ([\d.]+) ms, wait for k_parse ([\d.]+) ms", l)
    if m:
        for i in range(3): a[i]+=float(m.group(i+1))
        n+=1
    elif l.startswith("threads"): print(l.strip())
print("batches %d: mean host prep %.3f ms, enqueue %.3f ms, wait %.3f ms" % (n, a[0]/max(n,1), a[1]/max(n,1), a[2]/max(n,1)))
