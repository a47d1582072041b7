import os, sys
os.environ["EDRGP_CALIB_ORDERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
print(h.calib())
