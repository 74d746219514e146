"""debug: n final exponentiations in one launch, per-workgroup cycles and placement (LSA_FE_STAMPS=1)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import legosnark_amd as lsa
from legosnark_amd import curve
lsa.init(0)
G1, G2 = curve.generator("g1"), curve.generator("g2")
f = lsa.miller_loop(lsa.normalize("g1", G1.reshape(1, 12)), lsa.normalize("g2", G2.reshape(1, 24)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fs = np.repeat(f.reshape(1, 48), n, axis=0)
lsa.final_exponentiation(fs)
