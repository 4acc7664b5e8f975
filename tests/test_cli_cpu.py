"""Boundary pieces that need no GPU: the `diatomic` / `atomic` command lines (flag names and defaults of
/root/reference/src/diatomic/main.cpp:89-133 and src/atomic/main.cpp:63-119, refusal of out-of-scope options), the options
structure behind them, the checkpoint format (/root/reference/src/general/checkpoint.cpp) and that the C++ adapter header
compiles against an Armadillo-compatible matrix type."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "helfem_amd", "bin")


@pytest.fixture(scope="module")
def hf(native_libs):
    import helfem_amd
    helfem_amd.lib()
    from helfem_amd import build
    build.build_cli(verbose=False)
    build.build_adapter_test(verbose=False)
    return helfem_amd


def run(exe, *args):
    p = subprocess.run([os.path.join(BIN, exe)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


# flag = default pairs of the reference parsers; "*" = required
DIATOMIC_FLAGS = {"Z1": "*", "Z2": "*", "Rbond": "*", "angstrom": "0", "nela": "0", "nelb": "0", "Q": "0", "M": "0", "lmax": "*",
                  "mmax": "-1", "lpad": "10", "Rmax": "40.0", "grid": "4", "zexp": "1.0", "nelem": "*", "nnodes": "15", "nquad": "0",
                  "maxit": "50", "convthr": "1e-7", "Ez": "0.0", "Qzz": "0.0", "Bz": "0.0", "diag": "1", "finitenuc": "0",
                  "Rrms1": "0.0", "Rrms2": "0.0", "method": "HF", "ldft": "0", "mdft": "0", "dftthr": "1e-12", "restricted": "-1",
                  "symmetry": "1", "primbas": "4", "diiseps": "1e-2", "diisthr": "1e-3", "diisorder": "5", "readocc": "0",
                  "perturb": "0.0", "seed": "0", "load": "", "save": "helfem.chk", "x_pars": "", "c_pars": "", "maverage": "0"}
ATOMIC_EXTRA = {"Z": "*", "Zl": "", "Zr": "", "Rmid": "0.0", "mmax": "*", "grid0": "4", "zexp": "2.0", "zexp0": "2.0", "nelem0": "0",
                "Rrms": "0.0", "dampfock": "0.7", "dampthr": "0.1", "zeroder": "0", "taylor_order": "-1", "iconf": "0", "conf_N": "0",
                "conf_R": "0.0", "conf_barrier": "0.0", "shift_conf": "0.0", "add_conf": "1"}


def usage_flags(text):
    """{flag: default or '*'} from the usage text"""
    out = {}
    for line in text.splitlines():
        m = re.match(r"\s+--(\S+)\s+(.*)$", line)
        if not m:
            continue
        d = re.search(r"\[=(.*)\]$", m.group(2))
        out[m.group(1)] = d.group(1) if d else "*"
    return out


def test_diatomic_flags_and_defaults(hf):
    rc, out, err = run("diatomic", "--help")
    assert rc == 0
    flags = usage_flags(err)
    for k, v in DIATOMIC_FLAGS.items():
        assert k in flags, k
        assert flags[k] == v, (k, flags[k], v)
    # the one deliberate difference: --iguess defaults to the core guess, the SAP table (reference default 2) is outside the scope
    assert flags["iguess"] == "0"


def test_atomic_flags_and_defaults(hf):
    rc, out, err = run("atomic", "--help")
    assert rc == 0
    flags = usage_flags(err)
    want = dict(DIATOMIC_FLAGS)
    for k in ("Z1", "Z2", "Rbond", "lpad", "finitenuc", "Rrms1", "Rrms2"):
        want.pop(k)
    want.update(ATOMIC_EXTRA)
    want["finitenuc"] = "0"
    for k, v in want.items():
        assert k in flags, k
        assert flags[k] == v, (k, flags[k], v)


def test_required_and_unknown_options(hf):
    rc, out, err = run("diatomic", "--Z1", "H", "--Z2", "H", "--Rbond", "1.4", "--lmax", "4")
    assert rc == 1 and "need option: --nelem" in err
    rc, out, err = run("diatomic", "--Z1", "H", "--Z2", "H", "--Rbond", "1.4", "--lmax", "4", "--nelem", "2", "--frobnicate", "1")
    assert rc == 1 and "undefined option: --frobnicate" in err
    rc, out, err = run("atomic", "--Z", "Xx", "--lmax", "0", "--mmax", "0", "--nelem", "2")
    assert rc == 1 and 'Element "Xx" not found' in err


@pytest.mark.parametrize("extra,msg", [(["--Ez", "0.01"], "fields are not supported"), (["--finitenuc", "1"], "Finite nuclear"),
                                       (["--readocc", "3"], "occs.dat"), (["--primbas", "3"], "LIP primitive basis"),
                                       (["--iguess", "2"], "SAP"),
                                       (["--iguess", "0", "--load", "x.chk"], "projection of the stored Fock matrix"),
                                       (["--lmax", "6,x", "--mmax", "-1"], "option value is invalid: --lmax=x"),
                                       (["--method", "hyb_lda_xc_cam_lda0"], "Range separated functionals are not supported"),
                                       (["--M", "2"], "Requested multiplicity 2 with 2 electrons"),
                                       (["--method", "no_such_functional"], "")])
def test_out_of_scope_options_are_refused_before_any_device_work(hf, extra, msg):
    rc, out, err = run("diatomic", "--Z1", "H", "--Z2", "H", "--Rbond", "1.4", "--lmax", "4", "--nelem", "2", *extra)
    assert rc == 1 and msg in err, err
    assert "no usable HIP device" not in err  # refused by the option check, not by the missing GPU


def test_readocc_reads_and_checks_occs_dat(hf, tmp_path):
    """--readocc: occs.dat of the working directory (diatomic/main.cpp:219-224), its sums against the spin state (:369-380)"""
    (tmp_path / "occs.dat").write_text("1 1 0\n1 0 1\n0 0 -1\n")
    p = subprocess.run([os.path.join(BIN, "diatomic"), "--Z1", "H", "--Z2", "H", "--Rbond", "1.4", "--lmax", "4", "--mmax", "1", "--nelem", "2",
                        "--readocc", "-1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, cwd=str(tmp_path))
    assert p.returncode == 1 and "Occupying 2 orbitals but should have 1 orbitals" in p.stderr.decode(), p.stderr.decode()
    (tmp_path / "occs.dat").write_text("1 1\n")
    p = subprocess.run([os.path.join(BIN, "diatomic"), "--Z1", "H", "--Z2", "H", "--Rbond", "1.4", "--lmax", "4", "--nelem", "2",
                        "--readocc", "-1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, cwd=str(tmp_path))
    assert p.returncode == 1 and "three columns" in p.stderr.decode(), p.stderr.decode()


def test_options_structure_defaults(hf):
    L = hf.lib()

    class Opt(ctypes.Structure):
        _fields_ = [("program", ctypes.c_int), ("Z1", ctypes.c_int), ("Z2", ctypes.c_int), ("Rbond", ctypes.c_double),
                    ("nela", ctypes.c_int), ("nelb", ctypes.c_int), ("Q", ctypes.c_int), ("M", ctypes.c_int),
                    ("lmmax", ctypes.c_int * 16), ("nlm", ctypes.c_int), ("lmax", ctypes.c_int), ("mmax", ctypes.c_int),
                    ("lpad", ctypes.c_int), ("Rmax", ctypes.c_double), ("grid", ctypes.c_int), ("zexp", ctypes.c_double),
                    ("nelem", ctypes.c_int), ("nnodes", ctypes.c_int), ("nquad", ctypes.c_int), ("maxit", ctypes.c_int),
                    ("convthr", ctypes.c_double), ("diag", ctypes.c_int), ("method", ctypes.c_char * 128), ("ldft", ctypes.c_int),
                    ("mdft", ctypes.c_int), ("dftthr", ctypes.c_double), ("restricted", ctypes.c_int), ("symmetry", ctypes.c_int),
                    ("primbas", ctypes.c_int), ("diiseps", ctypes.c_double), ("diisthr", ctypes.c_double), ("diisorder", ctypes.c_int),
                    ("iguess", ctypes.c_int)]

    o = Opt()
    buf = ctypes.create_string_buffer(4096)  # the structure is larger than the prefix declared here
    for prog, zexp in ((0, 1.0), (1, 2.0)):
        assert L.hfg_scf_options_default(buf, prog) == 0
        ctypes.memmove(ctypes.byref(o), buf, ctypes.sizeof(o))
        assert (o.lpad, o.Rmax, o.grid, o.zexp, o.nnodes, o.nquad, o.maxit) == (10, 40.0, 4, zexp, 15, 0, 50)
        assert (o.convthr, o.diag, o.method, o.dftthr, o.restricted, o.symmetry, o.primbas) == (1e-7, 1, b"HF", 1e-12, -1, 1, 4)
        assert (o.diiseps, o.diisthr, o.diisorder) == (1e-2, 1e-3, 5)


def test_parse_xc_params_and_elements(hf, tmp_path):
    L = hf.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    L.hfg_parse_xc_params.argtypes = [ctypes.c_char_p, dp, ctypes.POINTER(ctypes.c_int)]
    f = tmp_path / "xpars.dat"
    f.write_text("0.804\n0.2195149727645171\n")
    v = np.zeros(8)
    n = ctypes.c_int(8)
    assert L.hfg_parse_xc_params(str(f).encode(), v.ctypes.data_as(dp), ctypes.byref(n)) == 0
    assert n.value == 2 and v[0] == 0.804 and v[1] == 0.2195149727645171
    n = ctypes.c_int(8)
    assert L.hfg_parse_xc_params(b"1.5 2.5 3.5", v.ctypes.data_as(dp), ctypes.byref(n)) == 0 and n.value == 3 and v[2] == 3.5
    n = ctypes.c_int(8)
    assert L.hfg_parse_xc_params(b"", v.ctypes.data_as(dp), ctypes.byref(n)) == 0 and n.value == 0
    L.hfg_get_Z.argtypes = [ctypes.c_char_p]
    assert [L.hfg_get_Z(s) for s in (b"H", b"he", b"N", b"Ar", b"F", b"Li", b"18", b"", b"Og")] == [1, 2, 7, 18, 9, 3, 18, 0, 118]
    assert L.hfg_get_Z(b"Qq") < 0


def test_adapter_header_compiles_against_a_plain_matrix_type(hf):
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_test")
    p = subprocess.run([exe, "compile-only"], stdout=subprocess.PIPE, timeout=60)
    assert p.returncode == 0 and b"adapter compiled" in p.stdout


# ---- checkpoint format --------------------------------------------------------------------------------------------------
def h5dump_header(path):
    out = subprocess.run(["/opt/conda/bin/h5dump", "-H", path], stdout=subprocess.PIPE, timeout=60).stdout.decode()
    sets = {}
    for m in re.finditer(r'DATASET "([^"]+)" \{\s*DATATYPE\s+(\S+)\s*DATASPACE\s+(SCALAR|SIMPLE \{ \( ([0-9, ]+) \))', out):
        sets[m.group(1)] = (m.group(2), None if m.group(3) == "SCALAR" else tuple(int(x) for x in m.group(4).split(",")))
    return sets


@pytest.fixture()
def chk(hf):
    L = hf.lib()
    if not L.hfg_chk_available():
        pytest.skip("no libhdf5 in this environment")
    dp = ctypes.POINTER(ctypes.c_double)
    i64 = ctypes.POINTER(ctypes.c_int64)
    L.hfg_chk_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.hfg_chk_close.argtypes = [ctypes.c_void_p]
    L.hfg_chk_exist.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.hfg_chk_write_mat.argtypes = [ctypes.c_void_p, ctypes.c_char_p, dp, ctypes.c_int64, ctypes.c_int64]
    L.hfg_chk_write_ivec.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int64]
    L.hfg_chk_write_double.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
    L.hfg_chk_write_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.hfg_chk_write_basis.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.hfg_chk_read_mat.argtypes = [ctypes.c_void_p, ctypes.c_char_p, dp, i64, i64]
    L.hfg_chk_read_ivec.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), i64]
    L.hfg_chk_read_double.argtypes = [ctypes.c_void_p, ctypes.c_char_p, dp]
    L.hfg_chk_read_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.hfg_chk_read_diatomic_basis.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    return L


def test_checkpoint_round_trip_and_layout(hf, chk, tmp_path):
    import common
    L = chk
    dp = ctypes.POINTER(ctypes.c_double)
    path = str(tmp_path / "t.chk").encode()
    h = ctypes.c_void_p()
    assert L.hfg_chk_open(path, 1, ctypes.byref(h)) == 0, L.hfg_last_error()
    A = np.asfortranarray(np.arange(12, dtype=float).reshape(3, 4) + 0.25)  # 3 x 4, column-major like arma::mat
    v = np.array([1.5, -2.0, 3.25, 7.0, 9.0])
    iv = (ctypes.c_int * 4)(3, -1, 0, 2)
    assert L.hfg_chk_write_mat(h, b"A", A.ctypes.data_as(dp), 3, 4) == 0
    assert L.hfg_chk_write_mat(h, b"v", v.ctypes.data_as(dp), 5, 1) == 0
    assert L.hfg_chk_write_ivec(h, b"iv", iv, 4) == 0
    assert L.hfg_chk_write_double(h, b"Etot", -1.133629) == 0
    assert L.hfg_chk_write_int(h, b"nela", 7) == 0
    assert L.hfg_chk_write_double(h, b"Etot", -2.5) == 0  # overwriting an entry replaces it (checkpoint.cpp:127)
    gb, _ = common.make_bases(7, 7, 2.068, (3, 2), 2, 5, oracle=False)
    assert L.hfg_chk_write_basis(h, gb.h) == 0
    L.hfg_chk_close(h)

    # layout as the reference writes it
    sets = h5dump_header(path.decode())
    assert sets["A"] == ("H5T_IEEE_F64LE", (4, 3))       # dims = {n_cols, n_rows}
    assert sets["v"] == ("H5T_IEEE_F64LE", (1, 5))       # arma::vec = n x 1 matrix
    assert sets["iv"] == ("H5T_STD_I32LE", (4, 1))       # arma::ivec: {n_rows, n_cols}
    assert sets["Etot"] == ("H5T_IEEE_F64LE", None) and sets["nela"] == ("H5T_STD_I32LE", None)
    for name, shape in (("HelFEM_ID", None), ("Z1", None), ("Z2", None), ("Rhalf", None), ("bval", (1, 3)), ("n_quad", None),
                        ("poly_id", None), ("poly_nnodes", None), ("lval", (len(gb.lval), 1)), ("mval", (len(gb.mval), 1))):
        assert name in sets and sets[name][1] == shape, (name, sets.get(name))

    # and back
    h = ctypes.c_void_p()
    assert L.hfg_chk_open(path, 0, ctypes.byref(h)) == 0
    r, c = ctypes.c_int64(), ctypes.c_int64()
    assert L.hfg_chk_read_mat(h, b"A", None, ctypes.byref(r), ctypes.byref(c)) == 0 and (r.value, c.value) == (3, 4)
    B = np.zeros((3, 4), order="F")
    assert L.hfg_chk_read_mat(h, b"A", B.ctypes.data_as(dp), ctypes.byref(r), ctypes.byref(c)) == 0 and np.array_equal(A, B)
    d = ctypes.c_double()
    assert L.hfg_chk_read_double(h, b"Etot", ctypes.byref(d)) == 0 and d.value == -2.5
    i = ctypes.c_int()
    assert L.hfg_chk_read_int(h, b"HelFEM_ID", ctypes.byref(i)) == 0 and i.value == 2
    n = ctypes.c_int64(8)
    out = (ctypes.c_int * 8)()
    assert L.hfg_chk_read_ivec(h, b"iv", out, ctypes.byref(n)) == 0 and list(out[:n.value]) == [3, -1, 0, 2]
    assert L.hfg_chk_exist(h, b"A") == 1 and L.hfg_chk_exist(h, b"nothing") == 0
    assert L.hfg_chk_read_double(h, b"nothing", ctypes.byref(d)) != 0 and b"does not exist" in L.hfg_last_error()
    # the basis comes back from its constructor arguments with identical one-electron matrices
    hb = ctypes.c_void_p()
    assert L.hfg_chk_read_diatomic_basis(h, 10, ctypes.byref(hb)) == 0, L.hfg_last_error()
    S2 = np.zeros((gb.Nbf(), gb.Nbf()), order="F")
    assert L.hfg_basis_overlap(hb, S2.ctypes.data_as(dp)) == 0
    assert np.array_equal(S2, gb.overlap())
    L.hfg_basis_destroy(hb)
    L.hfg_chk_close(h)
    # writing into a read-only file is refused
    h = ctypes.c_void_p()
    assert L.hfg_chk_open(path, 0, ctypes.byref(h)) == 0
    assert L.hfg_chk_write_int(h, b"x", 1) != 0 and b"reading only" in L.hfg_last_error()
    L.hfg_chk_close(h)


def test_interbasis_overlap_projects_between_bases(hf):
    """TwoDBasis::overlap(const TwoDBasis &) (basis.cpp:713-750, RadialBasis::overlap basis.cpp:104-200): with the basis
    itself it is the overlap matrix; onto a refinement (every element split in two, same polynomial order) the functions of
    the coarse basis are represented exactly, so S11^-1 S12 maps S22-orthonormal vectors to S11-orthonormal ones"""
    lval, mval = hf.lm_to_l_m([3, 2])
    Rh = 1.0
    bcoarse = np.array([0.0, 0.6, 1.5, 3.0])
    bfine = np.array([0.0, 0.3, 0.6, 1.05, 1.5, 2.25, 3.0])
    b2 = hf.TwoDBasis(3, 9, Rh, 6, 30, bcoarse, lval, mval, 10)
    b1 = hf.TwoDBasis(3, 9, Rh, 6, 30, bfine, lval, mval, 10)
    S22, S11 = b2.overlap(), b1.overlap()
    assert np.max(np.abs(b2.overlap_with(b2) - S22)) < 1e-12 * np.max(np.abs(S22))
    S12 = b1.overlap_with(b2)
    assert S12.shape == (b1.Nbf(), b2.Nbf())
    # S22-orthonormal vectors of the coarse basis
    w, U = np.linalg.eigh(S22)
    C2 = U / np.sqrt(w)
    C1 = np.linalg.solve(S11, S12 @ C2)
    assert np.max(np.abs(C1.T @ S11 @ C1 - np.eye(b2.Nbf()))) < 1e-9
    # and an unrelated grid loses norm (a projection)
    b3 = hf.TwoDBasis(3, 9, Rh, 6, 30, np.array([0.0, 1.0, 3.0]), lval, mval, 10)
    C3 = np.linalg.solve(b3.overlap(), b3.overlap_with(b2) @ C2)
    nrm = np.diag(C3.T @ b3.overlap() @ C3)
    assert np.all(nrm < 1.0 + 1e-9) and np.min(nrm) < 0.999


def test_atomic_interbasis_overlap(hf):
    """atomic TwoDBasis::overlap(const TwoDBasis &) (src/atomic/TwoDBasis.cpp:330-344): radial int B_i B'_j dr on matching
    (l, m) shells; same properties as the diatomic one, plus: an angular shell missing from the other basis gives zero rows"""
    lval, mval = [0, 1, 1, 1], [0, -1, 0, 1]
    bcoarse = np.array([0.0, 0.5, 2.0, 8.0])
    bfine = np.array([0.0, 0.25, 0.5, 1.25, 2.0, 5.0, 8.0])
    b2 = hf.AtomicTwoDBasis(4, 8, 40, bcoarse, lval, mval)
    b1 = hf.AtomicTwoDBasis(4, 8, 40, bfine, lval, mval)
    S22, S11 = b2.overlap(), b1.overlap()
    assert np.max(np.abs(b2.overlap_with(b2) - S22)) < 1e-12 * np.max(np.abs(S22))
    S12 = b1.overlap_with(b2)
    assert S12.shape == (b1.Nbf(), b2.Nbf())
    w, U = np.linalg.eigh(S22)
    C2 = U / np.sqrt(w)
    C1 = np.linalg.solve(S11, S12 @ C2)
    assert np.max(np.abs(C1.T @ S11 @ C1 - np.eye(b2.Nbf()))) < 1e-9
    # s-only basis against the s+p one: the p shells of the larger basis have no partner
    bs = hf.AtomicTwoDBasis(4, 8, 40, bcoarse, [0], [0])
    Ssp = b2.overlap_with(bs)
    R = b2.Nbf() // 4
    assert np.max(np.abs(Ssp[:R] - bs.overlap())) < 1e-12 and np.max(np.abs(Ssp[R:])) == 0.0

