#!/usr/bin/env python
"""Benchmark of the SCF hot path (Fock build + generalized eigensolve) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run
(one rank per GPU, RCCL).  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[3], the configuration the metric "SCF-iteration wall time at Nbf~4000" is
quoted on; it fits one GPU): diatomic N2, R=2.068 a0, gga_x_pbe-gga_c_pbe, 5 radial elements of 15-node
LIPs, nquad 75, lmmax=[20,20], lpad 10 -> Nrad 70, Nang 61, Nbf 4230; XC grid ldft 92 x mdft 13; default
--symmetry 1 (three m-blocks of 1470/1380/1380).  One "step" = one SCF iteration's hot path:
J(P) + XC(P) -> F = sym(H0+J+XC) -> eig_gsym_sub(F) -> P = C_occ C_occ^T, inputs resident in HBM.
Density: the core-Hamiltonian guess orbitals (deterministic, synthetic - no files).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: Z1, Z2, Rbond, lmmax, nelem, nnodes, method ids (x,c), nocc
    "n2_pbe_nbf4230": dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[20, 20], nelem=5, nnodes=15, x=101, c=130, nocc=7),
    # BASELINE configs[4] sizing: LiF, lmmax=[29,29], 5 x 15 -> Nang 88, Nbf 6102; symmetry blocks 2100/2001/2001
    "lif_pbe_nbf6102": dict(Z1=3, Z2=9, Rbond=2.955, lmmax=[29, 29], nelem=5, nnodes=15, x=101, c=130, nocc=6),
    "n2_pbe_small": dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[6, 6], nelem=3, nnodes=8, x=101, c=130, nocc=7),
    # exact-exchange kernel path (BASELINE configs[4]; the diatomic program has no range-separated exchange -- the reference
    # refuses it too, main.cpp:393 -- so the global hybrid PBE0: J + 0.25 K + XC(hyb_gga_xc_pbeh) + eig, main.cpp:820-877)
    "lif_pbe0_nbf6102": dict(Z1=3, Z2=9, Rbond=2.955, lmmax=[29, 29], nelem=5, nnodes=15, x=406, c=0, nocc=6, kfrac=0.25),
    "n2_pbe0_nbf4230": dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[20, 20], nelem=5, nnodes=15, x=406, c=0, nocc=7, kfrac=0.25),
    "n2_pbe0_small": dict(Z1=7, Z2=7, Rbond=2.068, lmmax=[6, 6], nelem=3, nnodes=8, x=406, c=0, nocc=7, kfrac=0.25),
}


# dominant kernel: the trailing-matrix sweep of the Householder tridiagonalisation (one launch per column)
TRD_KERNEL = "k_trdb_gemv" if os.environ.get("HELFEM_TRD") == "twokernel" else "k_trdf"


def kernel_sources_sha():
    """fingerprint of the HIP sources the profiled kernels come from: tools/pmc_traffic.py stores it with the PMC traffic
    it reduces, and the bench line drops a committed `traffic` figure that was measured on other kernel sources"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "helfem_amd", "csrc", "hip")
    for f in ("trdp.hip", "trd.hip", "gemm.hip", "exchange_lr.hip", "fock.hip", "wave.h"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_basis(hf, w):
    lval, mval = hf.lm_to_l_m(w["lmmax"])
    Rh = 0.5 * w["Rbond"]
    bval = hf.get_grid(float(np.arccosh(40.0 / Rh)), w["nelem"], 4, 1.0)
    b = hf.TwoDBasis(w["Z1"], w["Z2"], Rh, w["nnodes"], 5 * w["nnodes"], bval, lval, mval, 10)
    ldft = 4 * max(w["lmmax"]) + 12
    mdft = 4 * len(w["lmmax"]) + 5
    return b, bval, lval, mval, ldft, mdft


class MT19937_64(object):
    """std::mt19937_64 (Matsumoto / Nishimura 2000), for the seeded density BASELINE.md section 2 names"""

    def __init__(self, seed):
        self.mt = [0] * 312
        self.mt[0] = seed & 0xFFFFFFFFFFFFFFFF
        for i in range(1, 312):
            self.mt[i] = (6364136223846793005 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 62)) + i) & 0xFFFFFFFFFFFFFFFF
        self.idx = 312

    def next(self):
        if self.idx >= 312:
            mt, UM, LM = self.mt, 0xFFFFFFFF80000000, 0x7FFFFFFF
            for i in range(312):
                x = (mt[i] & UM) | (mt[(i + 1) % 312] & LM)
                mt[i] = mt[(i + 156) % 312] ^ (x >> 1) ^ (0xB5026F5AA96619E9 if x & 1 else 0)
            self.idx = 0
        x = self.mt[self.idx]
        self.idx += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        return x

    def uniform(self, lo, hi):
        """std::uniform_real_distribution<double>(lo, hi): one 64-bit draw per value (generate_canonical<double, 53>)"""
        u = min(self.next() / 18446744073709551616.0, 1.0 - 2.0 ** -53)
        return lo + (hi - lo) * u


def seeded_density(N, nocc, blocks, Sinvh, seed=20260130):
    """BASELINE.md section 2, kernel-only runs: P = 2 C C^T with C = Sinvh Q, Q orthonormalised columns of i.i.d.
    U(-1,1) numbers from std::mt19937_64(seed), built per symmetry block so that P is block diagonal in m; the nocc
    columns are dealt out as in a diatomic ground state (every further block one orbital, the rest in the first block)"""
    rng = MT19937_64(seed)
    per = [1] * len(blocks)
    per[0] = nocc - (len(blocks) - 1)
    if per[0] < 1:
        per = [nocc] + [0] * (len(blocks) - 1)
    C = np.zeros((N, nocc), order="F")
    col = 0
    for idx, k in zip(blocks, per):
        if k == 0:
            continue
        n = len(idx)
        Q = np.array([[rng.uniform(-1.0, 1.0) for _ in range(k)] for _ in range(n)])
        Q, _ = np.linalg.qr(Q)
        cols = np.where(np.max(np.abs(Sinvh[np.ix_(idx, range(N))]), axis=0) > 0)[0]
        C[:, col:col + k] = Sinvh[:, cols] @ Q
        col += k
    return 2.0 * (C @ C.T)


def stage_rooflines(basis, w, sizes, fams):
    """Coulomb against HBM, the eigensolve's N^3 products against the FP64 matrix peak.
    Coulomb: read P, write J, the (L,M) intermediates once, the in-element tables once:
    2 (Nd^2 + 2 N_LM R^2) 8 B + 4 N_lm E p^4 8 B.  Products per block: F X (2 n^3), the lower 128 x 128 tiles of
    X^T (F X), C = X Z (2 n^3)."""
    out = []
    try:
        lm = basis.lm_map()
        n_lm, n_LM = len(lm), sum(1 if m == 0 else 2 for (_, m) in lm)
        R, E, p = basis.Nrad(), w["nelem"], w["nnodes"]
        Nd = basis.Nang() * R
        cbytes = 2.0 * (Nd * Nd + 2.0 * n_LM * R * R) * 8 + 4.0 * n_lm * E * p ** 4 * 8
        ms = fams["coulomb"]["ms_per_step"]
        if ms > 0:
            gbs = cbytes / (ms * 1e-3) / 1e9
            out.append({"stage": "coulomb (4 kernels)", "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                        "frac": gbs / 8000.0, "algorithmic_bytes": cbytes, "ms": ms,
                        "note": "HIP events on the side stream: the four kernels run beside the XC kernels of the main "
                                "stream (HELFEM_FOCK_OVERLAP=0: alone, 0.41 ms = 3.3 TB/s)"})
    except Exception:
        pass
    ms = fams.get("eig_products", {}).get("ms_per_step", 0.0)
    if ms > 0:
        fl = 0.0
        for n in sizes:
            t = (n + 127) // 128
            low = sum(min(128, n - 128 * i) * min(128, n - 128 * j) for i in range(t) for j in range(i + 1))
            fl += 2.0 * n ** 3 + 2.0 * n * low + 2.0 * n ** 3
        tf = fl / (ms * 1e-3) / 1e12
        out.append({"stage": "eigensolve N^3 products (F X, lower tiles of X^T(F X), X Z)", "bound": "mfma", "achieved": tf,
                    "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "flops": fl, "ms": ms})
    return out


def usable_cores():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (a GPU box
    hands a 256-core host's affinity to a job that owns 16 CPUs' worth of time: 256 spinning BLAS threads on that are
    slower than one)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, int(q / float(fh.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 64))


def _progress(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(w, bval, lval, mval, ldft, mdft, P, F, Sinvh, blocks):
    """The CPU path timed on this host's cores, on a bounded sample of the same workload, extrapolated to one SCF
    iteration (kind "port+lapack"):
      * generalized eigensolve as scf::eig_gsym_sub does it (/root/reference/src/general/scf_helpers.cpp:131-186): per
        symmetry block Sinvh^T F Sinvh, LAPACK dsyevd, Sinvh C -- ALL blocks in full, through torch's CPU LAPACK/BLAS (MKL)
        on every core this process may use;
      * XC quadrature with the oracle (loop-for-loop restatement of the reference's dense algorithm), `cores` radial points
        at a time on `cores` threads, the sample points spread evenly over the radial elements;
      * Coulomb with the oracle, one thread -- the reference's coulomb() has no OpenMP either (basis.cpp:1359-1530)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading
    import torch
    import oracle_lib as orc
    cores = usable_cores()
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    torch.set_num_threads(cores)
    Rh = 0.5 * w["Rbond"]
    os.environ["HELFEM_NUM_THREADS"] = str(cores)
    _progress("cpu_baseline on %d cores (%s): tables" % (cores, cpu_model))
    ob = orc.OracleBasis(w["Z1"], w["Z2"], Rh, w["nnodes"], 5 * w["nnodes"], bval, lval, mval, 10)
    ob.compute_tei(False)
    t0 = time.time()
    ob.coulomb(P)
    tJ = time.time() - t0
    _progress("cpu_baseline: Coulomb %.2f s; XC sample" % tJ)
    # XC: one radial point per thread, points spread over the elements
    NQ = w["nelem"] * 5 * w["nnodes"]
    nth = max(1, min(cores, 32, NQ))
    npts = nth if NQ >= nth else NQ
    pts = sorted(set(int(round((k + 0.5) * NQ / npts)) % NQ for k in range(npts)))
    work = list(pts)
    lock = threading.Lock()

    def worker():
        while True:
            with lock:
                if not work:
                    return
                q = work.pop()
            ob.eval_Fxc(ldft, mdft, w["x"], w["c"], P, q_begin=q, q_end=q + 1)

    threads = [threading.Thread(target=worker) for _ in range(nth)]
    t0 = time.time()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    tXC_sample = time.time() - t0
    tXC = tXC_sample * NQ / float(len(pts))
    _progress("cpu_baseline: XC sample %.2f s for %d points; eigensolve" % (tXC_sample, len(pts)))
    # eigensolve: every block, LAPACK on all cores
    Ft = torch.from_numpy(np.ascontiguousarray(F))
    Xt = torch.from_numpy(np.ascontiguousarray(Sinvh))
    sizes = [len(b) for b in blocks]
    t0 = time.time()
    for idx in blocks:
        it = torch.as_tensor(np.asarray(idx, dtype=np.int64))
        cols = torch.nonzero(Xt[it, :].abs().sum(dim=0) > 0).flatten()
        Xb = Xt[it][:, cols]
        Fb = Ft[it][:, it]
        Forth = Xb.T @ (Fb @ Xb)
        Eb, Zb = torch.linalg.eigh(Forth)
        Cb = Xb @ Zb
        del Cb, Eb
    tE = time.time() - t0
    _progress("cpu_baseline: eigensolve %.2f s" % tE)
    # exact exchange (hybrid workloads): the oracle builds K output block by output block like the reference
    # (basis.cpp:1575-1579, no OpenMP over blocks there either: scratch is allocated per block); a sample of blocks spread
    # over the shell list, one per thread, scaled to the A^2 blocks
    tK, ksample = 0.0, ""
    if float(w.get("kfrac", 0.0)) != 0.0:
        ob.compute_tei(True)
        A = len(lval)
        nks = max(1, min(cores, 16))
        pairs = [((7 * q) % A, (11 * q + 3) % A) for q in range(nks)]
        Ph = np.asfortranarray(0.5 * P)
        kwork = list(pairs)

        def kworker():
            while True:
                with lock:
                    if not kwork:
                        return
                    jk = kwork.pop()
                ob.exchange_blocks(Ph, [jk])

        kth = [threading.Thread(target=kworker) for _ in range(nks)]
        t0 = time.time()
        for t in kth:
            t.start()
        for t in kth:
            t.join()
        tK_sample = time.time() - t0
        tK = tK_sample * (A * A) / float(len(pairs))
        ksample = ("; exchange: oracle, %d of %d output blocks (jang, kang) on %d threads (%.2f s wall, scaled x%.1f)"
                   % (len(pairs), A * A, nks, tK_sample, A * A / float(len(pairs))))
        _progress("cpu_baseline: exchange sample %.2f s for %d blocks" % (tK_sample, len(pairs)))
    total = tJ + tXC + tE + tK
    return dict(value=total * 1e3, unit="ms", cores=cores, kind="port+lapack", cpu=cpu_model,
                parts_ms=dict(coulomb=tJ * 1e3, xc=tXC * 1e3, eig=tE * 1e3, exchange=tK * 1e3),
                threads=dict(coulomb=1, xc=nth, eig=cores, exchange=(min(cores, 16) if tK else 0)),
                sample="Coulomb: oracle, full build, 1 thread as the reference (%.2f s); XC: oracle, %d of %d radial points spread "
                       "over the %d elements on %d threads (%.2f s wall, scaled x%.1f); eigensolve: all blocks %s with "
                       "LAPACK dsyevd + 3 GEMMs (torch CPU, MKL, %d threads, %.2f s)%s"
                       % (tJ, len(pts), NQ, w["nelem"], nth, tXC_sample, NQ / float(len(pts)), sizes, cores, tE, ksample))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="n2_pbe_nbf4230")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--symmetry", type=int, default=1, help="0: one unsymmetrised eigenproblem; 1 (reference default): m blocks")
    ap.add_argument("--density", default="core", choices=["core", "seeded"],
                    help="start density: occupied orbitals of the core Hamiltonian, or BASELINE.md's seeded mt19937_64 orbitals")
    args = ap.parse_args()

    import torch
    import helfem_amd as hf
    from helfem_amd import parallel
    rank, local_rank, world = parallel.init()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ncpu = os.cpu_count() or 8
    os.environ.setdefault("HELFEM_NUM_THREADS", str(max(1, min(16, ncpu) // max(1, world))))

    w = WORKLOADS[args.workload]
    basis, bval, lval, mval, ldft, mdft = build_basis(hf, w)
    kfrac = float(w.get("kfrac", 0.0))
    N = basis.Nbf()
    # one rank per GPU; HELFEM_BENCH_DEVICE pins every rank to one device (rehearsal of the N>1 path on a 1-GPU box
    # together with HELFEM_DIST_BACKEND=gloo)
    dev_index = int(os.environ.get("HELFEM_BENCH_DEVICE", local_rank))
    if kfrac == 0.0:
        basis.compute_tei(False)
    # hybrid workloads: the in-element tables and their exchange-ordered copies are built on the device (the 2-3 GB never
    # exist on the host)
    step = hf.DeviceSCFStep(basis, w["x"], w["c"], ldft, mdft, w["nocc"], symmetry=args.symmetry, device=dev_index, rank=rank,
                            nranks=world, kfrac=kfrac, device_tei=(kfrac != 0.0))
    ctx = step.ctx
    S = basis.overlap()
    H0 = basis.kinetic() + basis.nuclear()
    blocks = step.blocks
    Sinvh = hf.scf.form_Sinvh(S, False, blocks, ctx=ctx)
    step.set_matrices(H0, Sinvh)
    # core guess -> density (done through the same device path: F = sym(H0) by using a zero compact matrix)
    E0, C0 = hf.scf.eig_gsym_sub(H0, Sinvh, blocks, ctx=ctx)
    P0 = 2.0 * hf.scf.form_density(C0, w["nocc"], ctx=ctx)
    if args.density == "seeded":
        P0 = np.asfortranarray(seeded_density(N, w["nocc"], blocks, Sinvh))
    allred = parallel.allreduce_sum_ if (world > 1 or parallel.forced()) else None
    xblocks = parallel.broadcast_block_slots_ if (world > 1 or parallel.forced()) else None

    def one_step():
        step.set_density_scaled = None
        step.step(allred, xblocks)
        step.P.mul_(2.0)  # closed shell: P = Pa + Pb

    C_guess = C0 if args.density == "core" else None
    step.set_density(P0, C_guess)
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    step.set_density(P0, C_guess)
    ctx.profile(True)
    ctx.profile_reset()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, device=step.dev if (world > 1 or parallel.forced()) else "cpu")
    ms_per_step = dt / args.steps * 1e3

    fams = {}
    for name in ("coulomb", "xc", "scatter", "eig_reduce", "eig_tridiag", "eig_tridiag_solve", "eig_backtransform",
                 "gemm", "eig_products", "density", "k_trdp", "exchange", "exl_element_gemm", "exl_element_gemm_gflop"):
        ms, n = ctx.profile_get(name)
        fams[name] = dict(ms_per_step=ms / args.steps, calls=n)
    ctx.profile(False)
    # self-check of the timed path, independent of the number of ranks: ONE step from the fixed guess density
    # (the undamped iteration itself is chaotic from a core guess, its later iterates are not comparable)
    step.set_density(P0, C_guess)
    one_step()
    torch.cuda.synchronize()
    check = {"sum_lowest_eigenvalues_after_one_step": float(step.E[:w["nocc"]].sum().item()),
             "sum_all_eigenvalues_after_one_step": float(step.E.sum().item()),
             "xc_energy_after_one_step": float(step.scal[0].item())}

    # dominant kernel, measured live: all its launches of one eigensolve replayed back to back between two HIP
    # events on the launch stream (3 repetitions, the last is kept)
    # (rank 0 only: it always owns symmetry block 0; ranks beyond the number of blocks never ran the factorisation)
    ctx_gemv = (0.0, 0)
    trd_kernel = TRD_KERNEL
    trdp_shapes = {}
    if rank == 0:
        if fams["k_trdp"]["calls"] > 0:
            # persistent tridiagonalisation: ONE cooperative launch per eigensolve, timed by HIP events around each of
            # the launches of the timed region itself on the launch stream (nothing to replay: the kernel consumes its input)
            trd_kernel = "k_trdp"
            ctx_gemv = (fams["k_trdp"]["ms_per_step"] * args.steps, fams["k_trdp"]["calls"])
            # its launches by tile shape ("k_trdp<R, U>": one launch per phase of the eigensolve, see hip/trdp.hip)
            trdp_shapes = {}
            for nm in ctx.profile_names():
                if nm.startswith("k_trdp<"):
                    ms_s, calls_s = ctx.profile_get(nm)
                    trdp_shapes[nm] = {"us_per_launch": 1e3 * ms_s / max(1, calls_s), "launches_per_step": calls_s / float(args.steps)}
        else:
            for _ in range(3):
                ctx_gemv = ctx.measure_kernel(TRD_KERNEL)

    if rank == 0:
        sizes = [len(b) for b in blocks]
        my_sizes = [n for ib, n in enumerate(sizes) if ib % world == 0]
        # Dominant kernel: k_trdb_gemv, the trailing-matrix sweep y = A22 v of the Householder tridiagonalisation
        # (one launch per column, all symmetry blocks batched).  Algorithmic bytes: every Householder
        # column streams one triangle of the trailing matrix once, 4 (n-k)^2 B; summed over the columns of a
        # block that is (4/3) n^3 B.  Duration: HIP events around every launch on the launch stream.
        gemv_ms, gemv_launches = ctx_gemv
        launches_per_step = float(gemv_launches) / (args.steps if trd_kernel == "k_trdp" else 1.0)
        alg_bytes_step = sum(sum(4.0 * float(n - k - 1) ** 2 for k in range(n - 2)) for n in my_sizes)
        alg_bytes = alg_bytes_step / launches_per_step if launches_per_step else 0.0
        avg_ms = gemv_ms / gemv_launches if gemv_launches else 0.0
        trd_ms = fams["eig_tridiag"]["ms_per_step"]
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM-side bytes per launch of the same kernel from the committed PMC passes of this command
        # (tools/gpu_round.sh: separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as the gfx950 guide
        # prescribes; tools/pmc_traffic.py).  The counters cannot be collected from inside this process.
        traffic = None
        traffic_file = None
        try:
            if world == 1:
                import glob
                pat = "r[0-9][0-9]_pmc_traffic.json" if args.workload == "n2_pbe_nbf4230" else "r[0-9][0-9]_exchange_pmc_traffic_%s.json" % args.workload
                traffic_file = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))[-1]
                with open(traffic_file) as fh:
                    tj = json.load(fh)
                meta = tj.pop("_meta", {})
                if meta.get("kernel_sources_sha256", kernel_sources_sha()) == kernel_sources_sha():  # stale figures are dropped
                    if trdp_shapes:
                        # all launches of one eigensolve (one per phase, each its own template instance
                        # "void hfg::k_trdp<R, U, false>"), averaged per launch like `achieved`
                        tot, cnt = 0.0, 0.0
                        for nm, rec_s in trdp_shapes.items():
                            key = "void hfg::" + nm[:-1] + ", false>"
                            if key in tj and tj[key].get("traffic_bytes_per_launch") is not None:
                                tot += tj[key]["traffic_bytes_per_launch"] * rec_s["launches_per_step"]
                                cnt += rec_s["launches_per_step"]
                        if cnt == sum(r["launches_per_step"] for r in trdp_shapes.values()) and cnt > 0:
                            traffic = tot / cnt
                    else:
                        for kname, rec in tj.items():  # "hfg::k_trdf<1024>" or "hfg::k_trdb_gemv"
                            if ("hfg::" + trd_kernel) in kname:
                                traffic = rec.get("traffic_bytes_per_launch")
        except Exception:
            traffic = None
        out = {
            "metric": "scf_iteration_wall_time_fock_plus_geneig_nbf4230" if args.workload == "n2_pbe_nbf4230"
            else "scf_iteration_wall_time_" + args.workload,
            "value": ms_per_step, "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "diatomic Z1=%d Z2=%d R=%.3f %s, nelem=%d nnodes=%d nquad=%d lmmax=%s (Nbf=%d, "
                                   "Nang=%d, Nrad=%d), XC grid %dx%d, symmetry blocks %s: %s Fock build + "
                                   "eig_gsym_sub + density per step" % (
                                       w["Z1"], w["Z2"], w["Rbond"], "PBE0 (hyb_gga_xc_pbeh, 0.25 exact exchange)" if kfrac else "PBE",
                                       w["nelem"], w["nnodes"], 5 * w["nnodes"],
                                       str(w["lmmax"]).replace(" ", ""), N, basis.Nang(), basis.Nrad(), ldft, mdft, sizes,
                                       "J + 0.25 K + XC" if kfrac else "J+XC"),
                       "name": args.workload, "parallelism": "shard%d" % world,
                       "density": "occupied orbitals of the core Hamiltonian" if args.density == "core"
                       else "seeded std::mt19937_64(20260130) block-diagonal orbitals (BASELINE.md section 2)",
                       "streams": "two: the Coulomb kernels run beside the XC kernels, and X Q (the back-transformation folded "
                                  "into X) beside the divide-and-conquer stage, on a side stream of the context; stages_ms "
                                  "are HIP-event times on the stream a stage runs on and overlap where the stages do",
                       "timed_path": "device-resident step (hfg_*_dev entry points on HBM buffers: the loop body of hfg_scf_run, "
                                     "which the diatomic/atomic executables and helfem::gpu::run_scf call); the host-pointer "
                                     "entry points (hfg_coulomb, hfg_eig_gsym_sub, ...) add ~12 ms of PCIe per call"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "traffic_source": ("profiles/%s (committed rocprofv3 --pmc passes of this command, not "
                                            "measured in this run)" % os.path.basename(traffic_file)) if traffic is not None else None,
                         "kernel": "hfg::" + trd_kernel, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": avg_ms * 1e3, "launches_per_step": launches_per_step,
                         "launches_by_tile_shape": trdp_shapes,
                         "note": ("persistent cooperative launches, one per phase of an eigensolve (a phase ends when the "
                                  "trailing matrices fit the next narrower register tile), the trailing matrices resident in "
                                  "the register file: the algorithmic bytes (one triangle of the trailing matrix per Householder "
                                  "column, SURVEY 8d) never cross HBM; achieved = algorithmic bytes of an eigensolve / time of its "
                                  "launches; bound by one exchange between workgroups per column (DESIGN.md 3.4)")
                         if trd_kernel == "k_trdp" else
                         "latency-bound: one dependent launch per Householder column (see DESIGN.md 3.4)"},
            "stages_ms": {k: round(v["ms_per_step"], 4) for k, v in fams.items()},
            # the other stages against their own bounds (stage time of this run, algorithmic work)
            "stage_rooflines": stage_rooflines(basis, w, my_sizes, fams),
            # size-independent self-check of the timed path: must not depend on the number of ranks
            "check": check,
        }
        if kfrac != 0.0 and fams["exl_element_gemm"]["calls"] > 0:
            # exact-exchange workloads: the dominant kernel of the exchange build is the in-element task-list GEMM
            # (k_dgemm_tasklist_wl: C[p^2 x pairs] = ktei[p^2 x 4 p^2] RB[4 p^2 x pairs] per table slot and element).
            # achieved = USEFUL flops (no tile padding) / HIP-event time of its launches in the timed region.
            g_ms = fams["exl_element_gemm"]["ms_per_step"] * args.steps
            g_n = fams["exl_element_gemm"]["calls"]
            g_gflop = fams["exl_element_gemm_gflop"]["ms_per_step"] * args.steps  # accumulated GFLOP (see exchange_lr.hip)
            tf = g_gflop / g_ms if g_ms > 0 else 0.0  # GFLOP / ms = TFLOP/s
            ktraffic, ktraffic_file = None, None
            try:
                import glob
                cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_exchange_pmc_traffic_%s.json" % args.workload)))
                if cand:
                    ktraffic_file = cand[-1]
                    with open(ktraffic_file) as fh:
                        tj = json.load(fh)
                    meta = tj.pop("_meta", {})
                    if meta.get("kernel_sources_sha256", kernel_sources_sha()) == kernel_sources_sha():
                        for kname, rec in tj.items():  # the element GEMM is the <128, 128, false> (or <128, 64, false>) instance; <..., true> are the cross products
                            if "k_dgemm_tasklist_wl" in kname and "false>" in kname and rec.get("traffic_bytes_per_launch") is not None:
                                ktraffic = max(ktraffic or 0.0, rec["traffic_bytes_per_launch"])
            except Exception:
                ktraffic = None
            out["roofline_tridiagonalisation"] = out["roofline"]
            out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                               "traffic": ktraffic,
                               "traffic_source": ("profiles/%s (committed rocprofv3 --pmc passes, tools/exchange_pmc.sh; not measured "
                                                  "in this run)" % os.path.basename(ktraffic_file)) if ktraffic is not None else None,
                               "kernel": "hfg::k_dgemm_tasklist_wl (in-element GEMM of the exchange build)",
                               "useful_gflop_per_launch": g_gflop / g_n if g_n else 0.0, "avg_launch_us": g_ms / g_n * 1e3 if g_n else 0.0,
                               "launches_per_step": g_n / float(args.steps),
                               "note": "FP64 MFMA (v_mfma_f64_16x16x4_f64); peak 78.6 TFLOP/s is AMD's data-sheet figure, "
                                       "reproduced at 77 TFLOP/s by tests/gpu_probe/fp64_rate.hip"}
        if not args.no_cpu_baseline and world == 1:
            P = step.numpy(step.P, (N, N))
            F = step.numpy(step.F, (N, N))
            try:
                out["cpu_baseline"] = cpu_baseline(w, bval, lval, mval, ldft, mdft, P, F, Sinvh, blocks)
            except Exception as e:  # the checker library is test infrastructure; report rather than die
                out["cpu_baseline"] = {"value": None, "unit": "ms", "cores": 1, "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out))
    if world > 1 or parallel.forced():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
