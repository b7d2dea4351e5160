import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from spartan2_amd import frontend, hip, host
inst=frontend.sha256_circuit(bytes(2048))
ctx=hip.Context(0); sn=host.SpartanSNARK(ctx,inst)
tape=np.random.default_rng(1).integers(0,256,size=(4096,64),dtype=np.uint8)
sn.prep_prove(tape)
for i in range(3): sn.prove(tape)
os.environ["SPARTAN_HOST_LAPS"]="1"
w,u,ph=sn.prove(tape)
print(ph)
