"""bench-shaped batch through the eigensolver (3 blocks), for the stamp dump of hip/trdp.hip (HELFEM_TRDP_STAMPS=2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import trdp_check
sizes = [int(a) for a in sys.argv[1:]] or [1380, 1470, 1380]
trdp_check.check_blocks(sizes, reps=1)
