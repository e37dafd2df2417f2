import numpy as np, sys
n=int(sys.argv[2]); L=100; block=min(n,1000000)
rng=np.random.default_rng(1)
rec=np.empty((block,14+2*L),np.uint8)
names=np.char.zfill(np.arange(block).astype(str),7)
rec[:,0]=ord("@");rec[:,1]=ord("r");rec[:,2:9]=np.frombuffer("".join(names).encode(),np.uint8).reshape(block,7);rec[:,9]=10
rec[:,10:10+L]=np.frombuffer(b"ACGT",np.uint8)[rng.integers(0,4,(block,L))];rec[:,10+L]=10;rec[:,11+L]=ord("+");rec[:,12+L]=10
rec[:,13+L:13+2*L]=ord("I");rec[:,13+2*L]=10
with open(sys.argv[1],"wb") as f:
    for _ in range(n//block): f.write(rec.tobytes())
