"""A/B builds of one kernel source: variant libraries next to the product library, for measurements on the GPU box.

  python tools/ab_build.py build hip/trdp.hip  base:  late:-DTP_EXP_LATE=1  direct:-DTP_EXP_DIRECT=1
      compiles the named source once per variant with the extra flags and links it with the product's other objects into
      helfem_amd/build/variants/libhelfem_amd_<name>.so (helfem_amd/build/ is git-ignored; the .so files travel with gpurun)
  python tools/ab_build.py run <name>... -- <command>
      runs the command once per variant with HELFEM_AMD_LIB pointing at it (child processes: one library per process)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helfem_amd import build as hb  # noqa: E402

VDIR = os.path.join(hb.OBJDIR, "variants")


def build(rel, variants):
    hb.build_product(verbose=False)
    os.makedirs(VDIR, exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    objs = [os.path.join(hb.OBJDIR, r.replace("/", "_") + ".o") for r in hb.HOST_SRCS + hb.HIP_SRCS]
    mine = os.path.join(hb.OBJDIR, rel.replace("/", "_") + ".o")
    assert mine in objs, mine
    procs = []
    for v in variants:
        name, _, flags = v.partition(":")
        obj = os.path.join(VDIR, name + "_" + rel.replace("/", "_") + ".o")
        cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result"] + hb.EXTRA_FLAGS + flags.split() + [
            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(hb.CSRC, rel), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise SystemExit("variant %s failed" % name)
        open(os.path.join(VDIR, name + ".resources.txt"), "w").write(out)
        lib = os.path.join(VDIR, "libhelfem_amd_%s.so" % name)
        subprocess.check_call([hipcc, "-shared", "--offload-arch=gfx950", "-o", lib] + [obj if o == mine else o for o in objs] + ["-lpthread", "-ldl"])
        print("built", lib, flush=True)


def run(names, cmd):
    for name in names:
        lib = os.path.join(VDIR, "libhelfem_amd_%s.so" % name)
        print("==== variant %s ====" % name, flush=True)
        env = dict(os.environ, HELFEM_AMD_LIB=lib)
        subprocess.call(cmd, env=env)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        i = sys.argv.index("--")
        run(sys.argv[2:i], sys.argv[i + 1:])
