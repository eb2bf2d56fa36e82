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


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


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
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
