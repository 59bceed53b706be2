"""Builds liburso_hip.so (gfx950) in-tree with hipcc.  `python -m ursonet_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# Kernel experiments: URSO_LIB_VARIANT=<name> builds / loads lib/liburso_hip_<name>.so compiled with URSO_VARIANT_FLAGS
# (e.g. "-DURSO_PW_NT=1"), so that two compile-time variants can be compared inside ONE gpurun call on the same box.
VARIANT = os.environ.get("URSO_LIB_VARIANT", "")
LIB = os.path.join(LIBDIR, "liburso_hip%s.so" % (("_" + VARIANT) if VARIANT else ""))
SOURCES = ["runtime.hip", "conv_igemm.hip", "conv_pw.hip", "conv_pwx.hip", "conv_dense.hip", "conv_bneck.hip", "conv_halo.hip", "conv_halo2.hip", "conv_winograd.hip", "conv_pair.hip", "conv_pairw.hip", "conv_pairx.hip", "conv_pairs.hip", "conv_c3.hip", "conv_c3g.hip", "conv_hwgrad.hip", "conv_stem.hip", "conv_stemw.hip", "conv_wgrad.hip", "prep.hip", "pool_loss_optim.hip", "augment.hip", "bn_train.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_hash():
    """SHA-256 over everything the library is a function of: every file of csrc/ (name + bytes), the public header, the compile flags
    (incl. URSO_VARIANT_FLAGS) and the compiler's version string."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "ursonet_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS + os.environ.get("URSO_VARIANT_FLAGS", "").split() + SOURCES).encode())
    try:
        h.update(subprocess.run([_hipcc(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout)
    except Exception:
        pass
    return h.hexdigest()


def needs_build():
    """Stale unless the hash recorded next to the .so equals the hash of the sources as they are now (mtimes are meaningless after a
    checkout or a snapshot push; a .so shipped without its hash file is rebuilt where a compiler exists and trusted where none does)."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(LIB + ".srchash") as fh:
            return fh.read().strip() != source_hash()
    except IOError:
        return True


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library (no GPU needed)."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace(".hip", (("_" + VARIANT) if VARIANT else "") + ".o"))
        objs.append(o)
        cmd = [_hipcc()] + FLAGS + os.environ.get("URSO_VARIANT_FLAGS", "").split() + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode(errors="replace")))
        if verbose and out.strip():
            print(out.decode(errors="replace"))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(LIB + ".srchash", "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
