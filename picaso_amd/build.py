"""Build the gfx950 shared library (hipcc cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpicaso_hip.so")
ROCM = os.environ.get("ROCM_PATH") or os.environ.get("ROCM_HOME") or "/opt/rocm"
HIPCC = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


FLAGS += os.environ.get("PICASO_HIPCC_EXTRA", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


STAMP = LIB + ".srchash"


def source_hash():
    """sha256 over every input of the build: csrc/*.hip, csrc/*.hpp, the C header and the flags."""
    import hashlib
    h = hashlib.sha256()
    deps = sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + \
        [os.path.join(os.path.dirname(HERE), "include", "picaso_hip.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def stale():
    """The library is current when the hash stored beside it equals the hash of the sources (file
    times say nothing after a checkout or a copy to another box)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not stale():
        # progress goes to stderr: stdout belongs to the caller (bench.py owes its caller ONE JSON line)
        print("picaso_amd.build: libpicaso_hip.so reused (source hash %s matches)" % source_hash()[:12],
              file=sys.stderr)
        return LIB
    objs = []
    procs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    # librccl.so: the multi-GPU layer (csrc/comm.hip) calls RCCL directly
    rocm_lib = os.path.join(ROCM, "lib")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs +
                          ["-L" + rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib])
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    print("picaso_amd.build: libpicaso_hip.so compiled for gfx950 (source hash %s)" % source_hash()[:12],
          file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
