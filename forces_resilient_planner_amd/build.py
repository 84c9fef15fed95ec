"""Build the in-tree native pieces: the gfx950 solver library and (test infrastructure) the oracle."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libfrp_nmpc_amd.so")
SOURCES = ["frp_kernels.hip", "frp_ipm_lds.hip", "frp_ipm_lds_mem.hip", "frp_ipm_lds_q4.hip", "frp_ipm_lds_q30.hip", "frp_ipm_lds_s2.hip", "frp_capi.hip", "frp_pack.hip", "frp_tube.hip", "frp_corridor.hip", "frp_reference.hip", "frp_astar.hip"]
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


# Code-generation flags of the solver kernel's main translation unit (the test that inspects its generated code uses the same).
# The kernel sits at the 168-VGPR cap of three workgroups per CU with role loops full of live state, and what it issues is what
# it costs: every flag below trades recomputation or code size for fewer live registers / spills.  Measured one after
# the other on the same box, B = 4096 (tools/ab_variants.sh): 1.388 ms per launch with the defaults,
#   -amdgpu-use-amdgpu-trackers=1                 1.335   (the scheduler tracks pressure with the AMDGPU-specific trackers)
#   -disable-machine-licm                         1.310   (no hoisting of loop invariants into registers)
#   -disable-machine-cse                          1.296
#   -amdgpu-enable-rewrite-partial-reg-uses=0     1.278
#   -amdgpu-load-store-vectorizer=0               1.268
# (not additive beyond this: -greedy-reverse-local-assignment=1 alone 1.273, together with the last one 1.304; max-ilp, O2 / Os,
# no post-RA scheduler, no machine sinking, no LSR: all slower).  The variants that re-read the corridor rows from the
# parameters lose 2-10 % with the first flag already, so they are a translation unit of their own (frp_ipm_lds_mem.hip)
# on the defaults; the other files are indifferent and stay on the defaults too.
CODEGEN_FLAGS = ["-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-mllvm", "-disable-machine-licm", "-mllvm", "-disable-machine-cse",
                 "-mllvm", "-amdgpu-enable-rewrite-partial-reg-uses=0", "-mllvm", "-amdgpu-load-store-vectorizer=0"]
# The re-reading solver variants and the corridor kernel (168 VGPRs, three workgroups per CU as well) gain from not hoisting
# loop invariants into registers: (20, 10) variant in the full tick 1.68 -> 1.58 ms, (64, 30) 6.48 -> 5.76 ms, corridor
# kernel 1.09 -> 0.96 ms.  CAUTION, measured: "-disable-machine-licm -disable-machine-cse" WITHOUT the other flags of
# CODEGEN_FLAGS makes the kernels report a wrong objective (the iterates stay right) -- a code-generation problem of that
# combination in this compiler; tests/test_gpu_parity.py::test_every_kernel_variant_reports_the_oracles_numbers checks every
# variant's reported quantities against the oracle, and it is green for the sets below on all eight variants.
# ROOT CAUSE of that wrong objective (round 3, tools/dbg_objective.py + the disassembly of the object): with exactly those two
# switches this compiler selects  s_mov_b64 s[18:19], 0x4028000000000000  for the FP64 constants 12.0 / 10.0 / 20.0 of stage_cost()
# -- gfx9 encodes only 32-bit literals, the assembler keeps the LOW dword, and the register pair holds 0.0: the yaw term, the
# first-stage term and the final model's terminal term vanish from the REPORTED objective (the iteration uses other forms of
# the same weights).  A compiler bug, not undefined behaviour in the source.  check_device_code() below looks for that
# encoding in every object that goes into a library, and the flags are only used with the compiler they were tuned on.
NO_HOIST = ["-mllvm", "-disable-machine-licm"]
# Round 4, after the re-reading variants fetch their corridor rows in chunks (frp_ipm_lds.hip: for_faces): CODEGEN_FLAGS without the
# pressure trackers is their best set -- solve of the full tick ((20, 10) variant, 31 rows per stage) 1.062 ms with NO_HOIST alone,
# 1.043 with the vectorizer off as well, 1.019-1.026 with all five, 1.011-1.017 without the trackers, 1.121-1.125 with the compiler's
# defaults (tools/ab_tick.sh, two rounds on one box).
MEM_FLAGS = [f for i, f in enumerate(CODEGEN_FLAGS) if "trackers" not in f and not (f == "-mllvm" and "trackers" in CODEGEN_FLAGS[i + 1])]
TUNED_COMPILER = "roc-7.2.0"  # `hipcc --version` of the toolchain CODEGEN_FLAGS / NO_HOIST were measured and checked on


def codegen_flags_trusted():
    """The -mllvm switches are internal to one compiler release: with any other hipcc the sources are built with its defaults."""
    try:
        out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout
    except Exception:
        return False
    return TUNED_COMPILER in out


def check_device_code(obj):
    """Refuse an object whose gfx950 code holds an s_mov_b64 with a literal dword of zero: a 64-bit immediate that was cut to its
    low 32 bits (a real zero is an inline constant, never a literal).  Returns the number of instructions checked."""
    import re
    import tempfile
    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib", "llvm", "bin")
    if not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as d:
        co, fb = os.path.join(d, "dev.co"), os.path.join(d, "fatbin.bin")
        # a host object carries the device code as an offload bundle in its .hip_fatbin section (a --cuda-device-only object is the bundle itself)
        r = subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, obj, os.path.join(d, "copy.o")], capture_output=True, text=True)
        bundle = fb if r.returncode == 0 and os.path.exists(fb) and os.path.getsize(fb) > 0 else obj
        r = subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + bundle,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
        if r.returncode != 0 and "Can't find bundles" in r.stderr:
            return 0  # a translation unit without kernels
        if r.returncode != 0 or not os.path.exists(co):
            raise RuntimeError(f"cannot extract the gfx950 code of {obj}: {r.stderr}")
        dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
    n = 0
    for line in dis.splitlines():
        if "s_mov_b64" not in line:
            continue
        n += 1
        m = re.search(r"//\s*[0-9A-Fa-f]+:\s*([0-9A-Fa-f]{8})\s+([0-9A-Fa-f]{8})\s*$", line)
        if m and m.group(1).upper().endswith("FF") and int(m.group(2), 16) == 0:
            raise RuntimeError(f"{obj}: miscompiled 64-bit scalar immediate (literal cut to 32 bits): {line.strip()}")
    return n
PER_SOURCE_FLAGS = {"frp_ipm_lds.hip": CODEGEN_FLAGS + ["-DFRP_LDS_SPLIT_TU"],
                    "frp_ipm_lds_mem.hip": MEM_FLAGS,
                    # (the factorisation sweep inlined: as a function of its own it saves and restores 38 callee-saved registers per call
                    # whether or not the caller holds anything in them -- 0.858 -> 0.853 ms per 4096-problem launch; inlining the
                    # vector sweeps as well loses that again, profiles/r05_q4_flags.txt)
                    "frp_ipm_lds_q4.hip": CODEGEN_FLAGS + ["-DFRP_INLINE_FACTOR"],
                    "frp_ipm_lds_q30.hip": CODEGEN_FLAGS,
                    "frp_ipm_lds_s2.hip": CODEGEN_FLAGS + ["-DFRP_INLINE_FACTOR", "-DFRP_INLINE_SWEEPS"],
                    "frp_corridor.hip": NO_HOIST,
                    # the A* agrees with its oracle to the bit (node order depends on comparisons of nearly equal costs): no a * b + c contraction
                    "frp_astar.hip": ["-ffp-contract=off"]}
OBJDIR = os.path.join(PKG, "_build")


def build_native(force=False, verbose=True, lib=None, extra_flags=(), solver_flags=None, objdir=None, check=True, mem_flags=None, q4_flags=None):
    """hipcc --offload-arch=gfx950 -> forces_resilient_planner_amd/libfrp_nmpc_amd.so (in-tree): one object per source
    (compiled in parallel, per-source flags), linked into the shared library.
    Experiment builds (tools/build_variant.sh): `lib` = another output file, `extra_flags` for every source, `solver_flags`
    in place of CODEGEN_FLAGS for the solver kernel's translation unit (e.g. [] = the compiler's defaults)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    variant = lib is not None
    lib = lib or LIB
    if not variant and not force and not _stale(LIB, deps):
        return LIB
    objdir = objdir or OBJDIR
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")] + list(extra_flags)
    trusted = codegen_flags_trusted()
    if not trusted and verbose:
        print(f"[build] hipcc is not the {TUNED_COMPILER} toolchain the code-generation flags were tuned on: building with the compiler's defaults", flush=True)
    jobs = []
    for name, src in zip(SOURCES, srcs):
        obj = os.path.join(objdir, name + ".o")
        flags = PER_SOURCE_FLAGS.get(name, [])
        if not trusted:
            flags = [f for i, f in enumerate(flags) if f != "-mllvm" and (i == 0 or flags[i - 1] != "-mllvm")]
        if solver_flags is not None and name == "frp_ipm_lds.hip":
            flags = list(solver_flags) + ["-DFRP_LDS_SPLIT_TU"]
        if mem_flags is not None and name == "frp_ipm_lds_mem.hip":  # (experiments: the re-reading variants' translation unit)
            flags = list(mem_flags)
        if q4_flags is not None and name == "frp_ipm_lds_q4.hip":    # (experiments: the four-per-CU variants' translation unit)
            flags = list(q4_flags)
        cmd = common + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, obj, subprocess.Popen(cmd)))
    for cmd, obj, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if check:
        for cmd, obj, pr in jobs:
            check_device_code(obj)
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [j[1] for j in jobs] + ["-o", lib]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return lib


def build_variant(name, extra_flags=(), solver_flags=None, verbose=False, check=True, mem_flags=None, q4_flags=None):
    """forces_resilient_planner_amd/lib_<name>.so: the product sources with extra flags (selected with FRP_LIB=...)."""
    return build_native(force=True, verbose=verbose, lib=os.path.join(PKG, f"lib_{name}.so"), extra_flags=extra_flags,
                        solver_flags=solver_flags, objdir=os.path.join(OBJDIR, "variant_" + name), check=check, mem_flags=mem_flags, q4_flags=q4_flags)


DROPIN_DIR = os.path.join(PKG, "dropin", "lib")


def build_dropin_archives(verbose=False):
    """libFORCESNLPsolver_normal.a / libFORCESNLPsolver_final.a: the file names plan_manage links (CMakeLists.txt:64-65, 82-83:
    link_directories(.../lib) + target_link_libraries(... libFORCESNLPsolver_normal.a libFORCESNLPsolver_final.a)).  Both hold
    the same objects (either one resolves both solve symbols; the linker takes every member once); the consumer adds the HIP
    runtime to its link line (-L/opt/rocm/lib -lamdhip64), see INTEGRATION.md."""
    objs = [os.path.join(OBJDIR, n + ".o") for n in SOURCES]
    os.makedirs(DROPIN_DIR, exist_ok=True)
    out = []
    for name in ("libFORCESNLPsolver_normal.a", "libFORCESNLPsolver_final.a"):
        dst = os.path.join(DROPIN_DIR, name)
        if _stale(dst, objs):
            if os.path.exists(dst):
                os.remove(dst)
            cmd = ["ar", "rcs", dst] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        out.append(dst)
    # when the reference tree is here (build container): a stand-in for plan_manage's link line -- a program compiled against the
    # REFERENCE's generated headers, linked by plain g++ against the two archives + the HIP runtime; -m gpu tests run it on the box
    ref_inc = ["/root/reference/src/resilient_planner/plan_manage/solver/%s/FORCESNLPsolver_%s/include" % (m, m) for m in ("normal", "final")]
    stub_src = os.path.join(ROOT, "tests", "cpp", "static_dropin_stub.cpp")
    stub = os.path.join(os.path.dirname(DROPIN_DIR), "planner_stub")
    if all(os.path.isdir(d) for d in ref_inc) and os.path.exists(stub_src) and _stale(stub, out + [stub_src]):
        rocm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib")
        subprocess.check_call(["g++", "-O1", "-I" + ref_inc[0], "-I" + ref_inc[1], stub_src, "-L" + DROPIN_DIR, "-l:libFORCESNLPsolver_normal.a",
                               "-l:libFORCESNLPsolver_final.a", "-L" + rocm, "-lamdhip64", "-lpthread", "-Wl,-rpath," + rocm, "-o", stub])
    return out


DEFAULT_FLAGS_LIB = os.path.join(PKG, "lib_defaultflags.so")


def build_default_flags_lib(verbose=False):
    """The same library with the solver kernel's translation unit compiled WITHOUT CODEGEN_FLAGS (the other objects are the
    product's own): tests/test_gpu_parity.py runs the variant and horizon parity tests on it as well, so that a result never
    depends on an internal compiler switch."""
    src = os.path.join(CSRC, "frp_ipm_lds.hip")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "frp_nmpc.h"), os.path.abspath(__file__), LIB]
    if not _stale(DEFAULT_FLAGS_LIB, deps):
        return DEFAULT_FLAGS_LIB
    d = os.path.join(OBJDIR, "defaultflags")
    os.makedirs(d, exist_ok=True)
    obj = os.path.join(d, "frp_ipm_lds.hip.o")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
           "-DFRP_LDS_SPLIT_TU", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    check_device_code(obj)
    objs = [obj if n == "frp_ipm_lds.hip" else os.path.join(OBJDIR, n + ".o") for n in SOURCES]
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", DEFAULT_FLAGS_LIB])
    return DEFAULT_FLAGS_LIB


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
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-Wno-unused-value"] + (CODEGEN_FLAGS if "csrc/" in open(src).read() else []) + [src, "-o", exe]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


def build_oracle(verbose=True):
    """Test infrastructure: oracle/liboracle.so and, when /root/reference is present, oracle/_ref."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"],
                          stdout=None if verbose else subprocess.DEVNULL)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "variant":  # python -m forces_resilient_planner_amd.build variant <name> [--solver-flags=...] [flags...]
        sf = None
        mf = None
        qf = None
        rest = []
        chk = True
        for a_ in sys.argv[3:]:
            if a_.startswith("--solver-flags="):
                sf = a_.split("=", 1)[1].split()
            elif a_.startswith("--mem-flags="):
                mf = a_.split("=", 1)[1].split()
            elif a_.startswith("--q4-flags="):
                qf = a_.split("=", 1)[1].split()
            elif a_ == "--no-check":
                chk = False
            else:
                rest.append(a_)
        print(build_variant(sys.argv[2], rest, sf, check=chk, mem_flags=mf, q4_flags=qf))
    else:
        build_native(force=True)
        build_ubench()
        build_oracle()
