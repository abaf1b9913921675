import csv,re,collections,sys,glob
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
rows=list(csv.DictReader(open(f)))
d=collections.OrderedDict()
for r in rows:
    k=r['Dispatch_Id']
    e=d.setdefault(k,{'name':r['Kernel_Name'],'grid':r['Grid_Size'],'t':int(r['End_Timestamp'])-int(r['Start_Timestamp'])})
    e[r['Counter_Name']]=float(r['Counter_Value'])
agg=collections.OrderedDict()
pat=sys.argv[2] if len(sys.argv)>2 else 'gemm_kernel'
for k,e in d.items():
    if not re.search(pat,e['name']): continue
    m=re.search(r'(\w+)<([^>]*)>',e['name'])
    key=(m.group(1)+'<'+m.group(2)+'>' if m else e['name'][:40],e['grid'])
    a=agg.setdefault(key,collections.Counter()); a['n']+=1
    for c,v in e.items():
        if c not in('name','grid'): a[c]+=v
ctrs=[c for c in next(iter(agg.values())) if c not in('n','t')]
print('%-52s %9s %3s %8s | '%('kernel','grid','n','us')+' '.join('%14s'%c[-14:] for c in ctrs))
for (kk,g),a in agg.items():
    n=a['n']
    print('%-52s %9s %3d %8.1f | '%(kk[:52],g,n,a['t']/n/1e3)+' '.join('%14.4g'%(a[c]/n) for c in ctrs))
