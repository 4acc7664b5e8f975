import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helfem_amd as hf, bench
w = bench.WORKLOADS["n2_pbe_nbf4230"]
hf.default_context()
for dev in (True, False, True):
    basis, *_ = bench.build_basis(hf, w)
    t = time.time(); basis.compute_tei(False, device=dev); t1 = time.time() - t
    t = time.time(); basis.upload(92, 13); hf.default_context().synchronize(); t2 = time.time() - t
    print("compute_tei device=%s: %.3f s, upload %.3f s (HELFEM_NUM_THREADS=%s, cpus %d)" % (dev, t1, t2, os.environ.get("HELFEM_NUM_THREADS"), os.cpu_count()))
