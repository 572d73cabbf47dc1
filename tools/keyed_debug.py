import sys, os, ctypes
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
os.environ["LAMD_KEYED"]="1"
import numpy as np
from lightning_amd import Engine, workload
e=Engine(0)
w=workload.make_ecdsa(e, 2000, nkeys=7, publen=33, invalid_frac=0.0)
e.verify_ecdsa_device(w.dev[0],w.dev[1],w.dev[2],w.d_ok); e.synchronize()
print(e.info())
got=w.d_ok.cpu().numpy()
print("ok sum", got.sum(), "of", w.n)
def rd(which, nbytes, dtype):
    buf=np.zeros(nbytes,dtype=np.uint8)
    rc=e._lib.lamd_debug_read(e._ctx, which, 0, nbytes, buf.ctypes.data); assert rc==0, rc
    return buf.view(dtype)
nu=e.info()["last_unique_keys"]
print("rep", rd(2, 4*16, np.uint32)); print("uid", rd(3,4*16,np.uint32)); print("keyid", rd(4,4*32,np.uint32))
print("uniq_row", rd(5,4*nu,np.uint32)); print("keyok_u", rd(6,nu,np.uint8)); print("keyok_row", rd(1,32,np.uint8))
recs=rd(0, 80*4, np.uint32).reshape(4,20); print("rec flags", recs[:,16])
q=rd(7, 64*nu, np.uint32).reshape(nu,16)
import pyref
for u in range(min(nu,3)):
    row=int(rd(5,4*nu,np.uint32)[u]); pk=w.cols[2][row].tobytes(); pt=pyref.pubkey_parse(pk)
    x=sum(int(q[u,i])<<(32*i) for i in range(8)); y=sum(int(q[u,8+i])<<(32*i) for i in range(8))
    print("key",u,"row",row,"parse ok", (x,y)==pt)
    tab=rd(8, 4*6400*(u+1), np.uint32)[6400*u:6400*(u+1)]
    def ent(pos,d): 
        e_=tab[(pos*8+d-1)*24:(pos*8+d)*24]; return (sum(int(e_[i])<<(32*i) for i in range(8)), sum(int(e_[16+i])<<(32*i) for i in range(8)))
    for pos,d in ((0,1),(0,8),(1,1),(5,3),(32,1)):
        print("  tab",pos,d, ent(pos,d)==pyref.pmul(d*16**pos % pyref.N, pt))
