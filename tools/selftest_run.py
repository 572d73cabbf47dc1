import sys, json, ctypes
sys.path.insert(0,'.')
from lightning_amd import Engine
kat=json.load(open('tests/golden/kat.json'))
v=next(x for x in kat['ecdsa'] if x['name']=='KAT-B11')
H=bytes.fromhex
e=Engine(0)
buf=ctypes.create_string_buffer(8192)
rc=e._lib.lamd_x2_debug(e._ctx, buf, 8192); print("x2_debug mask=%#x"%rc, buf.value.decode()[:300])
rc,rep=e.inv_debug(); print("inv_debug mask=%#x"%rc, rep[:300])
rc,rep=e.selftest(H(v['hash']),H(v['sig']),H(v['pub'])); print("selftest rc=%#x"%rc, rep[:600])
for um in (0,1): print("chain use_mul=%d:"%um, e.chain_debug(um))
