import sys, os
os.environ["SPARTAN_HOST_LAPS"]="1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from spartan2_amd import frontend, hip, host
inst=frontend.sha256_circuit(bytes(2048))
ctx=hip.Context(0); sn=host.SpartanSNARK(ctx,inst)
tape=np.random.default_rng(1).integers(0,256,size=(4096,64),dtype=np.uint8)
sn.prep_prove(tape)
for i in range(4): sn.prove(tape)
sys.stderr.write("==== traced prove ====\n")
w,u,ph=sn.prove(tape)
print(ph)
