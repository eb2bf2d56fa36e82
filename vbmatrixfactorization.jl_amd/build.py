"""Build the HIP shared library in-tree (hipcc, gfx950 only).

    python vbmatrixfactorization.jl_amd/build.py

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to
the GPU box.  There is deliberately no other backend: no CPU fallback, no second architecture.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "vbmf_hip.hip")
OUT = os.path.join(HERE, "libvbmf_hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc")))]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "vbmf_hip.h"))


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libvbmf_hip.so)")


STAMP = OUT + ".srchash"


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17"] + os.environ.get("VBMF_HIPCC_FLAGS", "").split()


def source_hash():
    """SHA-256 over every source the library is built from (csrc/*, include/vbmf_hip.h) and the compile flags."""
    import hashlib
    h = hashlib.sha256(" ".join(_flags()).encode())
    for d in DEPS:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def up_to_date():
    """The in-tree library was built from exactly these sources (content hash recorded beside it at build time -- not file
    times: a .so that travelled with the tree but belongs to other sources is rebuilt)."""
    if not (os.path.exists(OUT) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == source_hash()


def build(force=False, verbose=False, out=OUT):
    if not force and out == OUT and up_to_date():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-unused-result", SRC, "-o", out + ".tmp",
           "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    cmd[1:1] = os.environ.get("VBMF_HIPCC_FLAGS", "").split()      # tuning switches for A/B builds (e.g. -DVBMF_EPI_PV_AHEAD=0)
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libvbmf_hip.so")
    if verbose:
        sys.stderr.write(r.stderr)
    os.replace(out + ".tmp", out)
    if out == OUT:
        with open(STAMP, "w") as f:
            f.write(source_hash() + "\n")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
