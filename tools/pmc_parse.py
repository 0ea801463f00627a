import csv,glob,collections,sys
d0=sys.argv[1]
names=['fwd full','fwd half','fwd half mask','dx full','dx half','dx half mask','dW full','dW half','dW half mask']
for d in sorted(glob.glob(d0+'/*/g_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    disp=collections.OrderedDict()
    for r in rows:
        if 'gemm_f16x2' not in r['Kernel_Name']: continue
        disp.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
    ids=list(disp); pos=0
    print(d.split('/')[-2])
    for nm in names:
        chunk=ids[pos:pos+7]; pos+=7
        if not chunk: break
        print('  %-14s'%nm, ' '.join('%s=%.4g'%(k,v) for k,v in disp[chunk[-1]].items()))
