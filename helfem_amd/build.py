"""In-tree build of the native libraries.

  helfem_amd/lib/libhelfem_amd.so   product: host setup code + HIP kernels + C ABI (hipcc, gfx950)
  oracle/liboracle.so               test-only CPU checker (g++)
  oracle/_ref/libref_legendre.so    the one buildable piece of the reference (only if /root/reference exists)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only container too.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "helfem_amd", "csrc")
LIBDIR = os.path.join(ROOT, "helfem_amd", "lib")
OBJDIR = os.path.join(ROOT, "helfem_amd", "build")

HOST_SRCS = ["host/fem.cpp", "host/special.cpp", "host/atomic_basis.cpp", "host/diatomic_basis.cpp", "host/scf.cpp", "host/diis.cpp", "host/checkpoint.cpp", "host/dftfuncs.cpp"]
HIP_SRCS = ["hip/tables.cpp", "hip/capi.cpp", "hip/fock.hip", "hip/exchange.hip", "hip/exchange_lr.hip", "hip/gemm.hip", "hip/eig.hip", "hip/dc.hip", "hip/trd.hip", "hip/trdp.hip",
            "hip/misc.hip", "hip/scf_gpu.cpp", "hip/scf_device.hip", "hip/tei_dev.hip"]


# Kernel arguments preloaded into SGPRs at wave launch (gfx940+): every launch of the eigensolver's dependent chains
# otherwise starts with a scalar load from the kernarg segment before it can even read its descriptor; measured at the
# bench workload: 9.14 -> 8.93 us per tridiagonalisation column, 22.85 -> 22.4 ms per step.
# HELFEM_HIPCC_FLAGS replaces these flags for A/B builds (empty string: none).
EXTRA_FLAGS = os.environ.get("HELFEM_HIPCC_FLAGS", "-mllvm -amdgpu-kernarg-preload-count=16").split()


def _newer(src, dst, extra_deps=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra_deps))


def _headers():
    hs = []
    for d in ("host", "hip"):
        dd = os.path.join(CSRC, d)
        hs += [os.path.join(dd, f) for f in os.listdir(dd) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "helfem_gpu.h"))
    return hs


def build_product(verbose=True, force=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = _headers()
    objs = []
    procs = []
    for rel in HOST_SRCS + HIP_SRCS:
        src = os.path.join(CSRC, rel)
        obj = os.path.join(OBJDIR, rel.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _newer(src, obj, hdrs):
            cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result"] + EXTRA_FLAGS + ["-c", src, "-o", obj]
            if not rel.endswith(".hip"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((rel, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for rel, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("---- %s ----\n%s\n" % (rel, out))
        elif verbose and "error" in out:
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    lib = os.path.join(LIBDIR, "libhelfem_amd.so")
    if force or procs or not os.path.exists(lib):
        cmd = [hipcc, "-shared", "--offload-arch=gfx950", "-o", lib] + objs + ["-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_cli(verbose=verbose, force=force or bool(procs))
    build_adapter_test(verbose=verbose)
    return lib


def build_cli(verbose=True, force=False):
    """helfem_amd/bin/diatomic and helfem_amd/bin/atomic: the reference's command lines in front of libhelfem_amd.so"""
    bindir = os.path.join(ROOT, "helfem_amd", "bin")
    os.makedirs(bindir, exist_ok=True)
    lib = os.path.join(LIBDIR, "libhelfem_amd.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    for name in ("diatomic", "atomic"):
        src = os.path.join(CSRC, "cli", name + "_main.cpp")
        exe = os.path.join(bindir, name)
        deps = (os.path.join(CSRC, "cli", "options.h"), os.path.join(ROOT, "include", "helfem_gpu.h"), lib)
        if force or _newer(src, exe, deps):
            # linked through hipcc so that the HIP runtime the library needs is found the same way as for the library itself
            cmd = [hipcc, "-O2", "-std=c++17", src, "-o", exe, "-L" + LIBDIR, "-lhelfem_amd", "-Wl,-rpath,$ORIGIN/../lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)


def build_adapter_test(verbose=True):
    """tests/cpp/adapter_test: calls the hot path through include/helfem_gpu_arma.hpp (built next to the library; the GPU box
    runs it from tests/test_gpu_adapter.py)"""
    src = os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_test")
    lib = os.path.join(LIBDIR, "libhelfem_amd.so")
    deps = (os.path.join(ROOT, "include", "helfem_gpu_arma.hpp"), os.path.join(ROOT, "include", "helfem_gpu.h"), lib)
    if _newer(src, exe, deps):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        cmd = [hipcc, "-O2", "-std=c++17", "-Wall", src, "-o", exe, "-L" + LIBDIR, "-lhelfem_amd", "-Wl,-rpath,$ORIGIN/../../helfem_amd/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return exe


def build_probe(verbose=True):
    """tests/gpu_probe/libtwostage_probe.so: the two-stage tridiagonalisation of round 2 (measured slower than the product
    path, kept as a probe with its test), linked against the product library for the GEMM task lists"""
    src = os.path.join(ROOT, "tests", "gpu_probe", "two_stage.hip")
    out = os.path.join(ROOT, "tests", "gpu_probe", "libtwostage_probe.so")
    lib = os.path.join(LIBDIR, "libhelfem_amd.so")
    if _newer(src, out, tuple(_headers()) + (lib,)):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-unused-result"] + EXTRA_FLAGS + [
            src, "-o", out, "-L" + LIBDIR, "-lhelfem_amd", "-Wl,-rpath,$ORIGIN/../../helfem_amd/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_oracle(verbose=True):
    cmd = ["make", "-C", os.path.join(ROOT, "oracle"), "-j8"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    ref = os.path.join(ROOT, "oracle", "build_ref.sh")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["bash", ref], stdout=None if verbose else subprocess.DEVNULL)
    return os.path.join(ROOT, "oracle", "liboracle.so")


if __name__ == "__main__":
    build_product(force="--force" in sys.argv)
    build_probe()
    build_oracle()
