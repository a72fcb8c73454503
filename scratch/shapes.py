import sys
rows=[]
for line in open(sys.argv[1]):
    M,N,K,kind,bm,bn,split,akf,bkf,count,ms=line.split()
    M,N,K,kind,bm,bn,split,akf,bkf,count=map(int,(M,N,K,kind,bm,bn,split,akf,bkf,count)); ms=float(ms)
    rows.append((ms,M,N,K,kind,bm,bn,split,akf,bkf,count,2.0*M*N*K*count))
tot=sum(r[0] for r in rows); totf=sum(r[-1] for r in rows)
print('total ms',round(tot,1),'GF',round(totf/1e9),'TF/s',round(totf/tot/1e9,1))
rows.sort(reverse=True); acc=0
n=int(sys.argv[2]) if len(sys.argv)>2 else 30
print('   ms    %   cum%  count      M       N      K kind bm  bn split akf bkf  TF/s  us/launch')
for ms,M,N,K,kind,bm,bn,split,akf,bkf,count,fl in rows[:n]:
    acc+=ms
    print(f'{ms:7.2f} {100*ms/tot:5.1f} {100*acc/tot:5.1f} {count:6d} {M:6d} {N:8d} {K:6d} {kind:3d} {bm:4d} {bn:4d} {split:4d} {akf:3d} {bkf:3d} {fl/ms/1e9:6.1f} {1e3*ms/count:8.1f}')
