"""Build the in-tree native pieces: the gfx950 solver library and (test infrastructure) the oracle."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libfrp_nmpc_amd.so")
SOURCES = ["frp_kernels.hip", "frp_ipm_lds.hip", "frp_capi.hip", "frp_pack.hip", "frp_tube.hip", "frp_corridor.hip", "frp_reference.hip"]
HEADERS = ["frp_kernels.h", "frp_device.hpp", "frp_model.hpp", "frp_adapter.hpp", os.path.join(ROOT, "include", "frp_nmpc.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -> forces_resilient_planner_amd/libfrp_nmpc_amd.so (in-tree)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc" if False else "-Wall",
           "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_ubench(verbose=True):
    """Micro-benchmarks / probes behind the numbers in DESIGN.md (tools/ubench/*.hip -> binaries next to the sources):
    MFMA layout and latency probes, the mat-vec building block, the FETCH_SIZE / WRITE_SIZE calibration kernel."""
    d = os.path.join(ROOT, "tools", "ubench")
    for f in sorted(os.listdir(d)):
        if not f.endswith(".hip"):
            continue
        src, exe = os.path.join(d, f), os.path.join(d, f[:-4])
        deps = [src]
        if "csrc/" in open(src).read():  # a probe that includes the product kernels is rebuilt with them
            c = os.path.join(ROOT, "forces_resilient_planner_amd", "csrc")
            deps += [os.path.join(c, g) for g in os.listdir(c)]
        if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(g) for g in deps):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-Wno-unused-value", src, "-o", exe]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


def build_oracle(verbose=True):
    """Test infrastructure: oracle/liboracle.so and, when /root/reference is present, oracle/_ref."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"],
                          stdout=None if verbose else subprocess.DEVNULL)


if __name__ == "__main__":
    build_native(force=True)
    build_ubench()
    build_oracle()
