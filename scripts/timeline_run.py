import numpy as np, sys, json
raw=np.fromfile(sys.argv[1],dtype=np.uint64)
g=raw[64*4*96:].reshape(-1,96)
rows=[r[:np.count_nonzero(r)].astype(np.int64) for r in g if r[0]!=0]
print("waves",len(rows),"stamps",len(rows[0]))
d=np.array([np.diff(r) for r in rows if len(r)==len(rows[0])])
m=d.mean(axis=0); tot=(np.array([r[-1]-r[0] for r in rows])).mean()
scale=float(sys.argv[2])/tot if len(sys.argv)>2 else 1.0
print("total",tot*scale)
print(" ".join(f"{x*scale:.2f}" for x in m))
