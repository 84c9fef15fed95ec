"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/frp_nmpc.h declares,
the drop-in structs have the reference's sizes, the host-side adapter mirrors pack identically in Python
and C++, and -- with no GPU -- every compute entry point fails loudly instead of falling back."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from forces_resilient_planner_amd import adapter, layout as L, solver, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        return solver.lib().frp_nmpc_device_count() > 0
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "frp_nmpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(frp_nmpc_\w+|FORCESNLPsolver_\w+_solve)\s*\(", hdr))
    declared -= {"frp_nmpc_options", "frp_nmpc_batch"}
    assert {"FORCESNLPsolver_normal_solve", "FORCESNLPsolver_final_solve", "frp_nmpc_solve_batch"} <= declared
    lib = solver.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/frp_nmpc.h but not exported"
    assert set(solver.EXPORTS) <= declared
    assert b"gfx950" in lib.frp_nmpc_version()


def test_dropin_struct_layouts_match_reference_sizes():
    # FORCESNLPsolver_normal.h:153-301 ([probed] sizes in SURVEY 8b): params 23600, output 2720, info 136
    assert ctypes.sizeof(solver.ForcesParams) == 23600
    assert solver.ForcesParams.x0.offset == 72 and solver.ForcesParams.all_parameters.offset == 2792
    assert solver.ForcesParams.num_of_threads.offset == 23592
    assert ctypes.sizeof(solver.ForcesOutput) == 2720
    assert ctypes.sizeof(solver.ForcesInfo) == 136
    assert solver.ForcesInfo.res_eq.offset == 8 and solver.ForcesInfo.solvetime.offset == 120


REF_INC = "/root/reference/src/resilient_planner/plan_manage/solver/{m}/FORCESNLPsolver_{m}/include"


@pytest.mark.skipif(not os.path.isdir(REF_INC.format(m="normal")), reason="reference tree absent (GPU box)")
@pytest.mark.parametrize("m", ["normal", "final"])
def test_dropin_matches_the_references_own_header(m, tmp_path):
    """Build-container check against the header plan_manage compiles with (FORCESNLPsolver_normal.h:153-323, included by
    forces_normal.cpp:2-3): one translation unit sees BOTH headers, compares every struct field, the callback type and
    the solve prototype, and then links the reference's prototype against libfrp_nmpc_amd.so."""
    src = tmp_path / "hdr.cpp"
    R = f"FORCESNLPsolver_{m}"
    src.write_text(f"""
#include "{R}.h"
// this repo's header declares the same two C symbols with its own (layout-compatible) struct names
#define FORCESNLPsolver_normal_solve frp_shadow_normal_solve
#define FORCESNLPsolver_final_solve frp_shadow_final_solve
#include "frp_nmpc.h"
#undef FORCESNLPsolver_normal_solve
#undef FORCESNLPsolver_final_solve
#include <cstddef>
#include <cstdio>
#include <type_traits>
#define SAME_FIELD(RT, FT, rf, ff) static_assert(offsetof(RT, rf) == offsetof(FT, ff) && sizeof(((RT *)0)->rf) == sizeof(((FT *)0)->ff), #rf)
static_assert(sizeof({R}_params) == sizeof(frp_forces_params), "params");
SAME_FIELD({R}_params, frp_forces_params, xinit, xinit);
SAME_FIELD({R}_params, frp_forces_params, x0, x0);
SAME_FIELD({R}_params, frp_forces_params, all_parameters, all_parameters);
SAME_FIELD({R}_params, frp_forces_params, num_of_threads, num_of_threads);
static_assert(sizeof({R}_output) == sizeof(frp_forces_output), "output");
static_assert(offsetof({R}_output, x01) == 0 && offsetof({R}_output, x02) == 17 * sizeof(double) && offsetof({R}_output, x20) == 19 * 17 * sizeof(double), "x01..x20 contiguous");
static_assert(sizeof({R}_info) == sizeof(frp_forces_info), "info");
SAME_FIELD({R}_info, frp_forces_info, it, it); SAME_FIELD({R}_info, frp_forces_info, it2opt, it2opt);
SAME_FIELD({R}_info, frp_forces_info, res_eq, res_eq); SAME_FIELD({R}_info, frp_forces_info, res_ineq, res_ineq);
SAME_FIELD({R}_info, frp_forces_info, rsnorm, rsnorm); SAME_FIELD({R}_info, frp_forces_info, rcompnorm, rcompnorm);
SAME_FIELD({R}_info, frp_forces_info, pobj, pobj); SAME_FIELD({R}_info, frp_forces_info, dobj, dobj);
SAME_FIELD({R}_info, frp_forces_info, dgap, dgap); SAME_FIELD({R}_info, frp_forces_info, rdgap, rdgap);
SAME_FIELD({R}_info, frp_forces_info, mu, mu); SAME_FIELD({R}_info, frp_forces_info, mu_aff, mu_aff);
SAME_FIELD({R}_info, frp_forces_info, sigma, sigma); SAME_FIELD({R}_info, frp_forces_info, lsit_aff, lsit_aff);
SAME_FIELD({R}_info, frp_forces_info, lsit_cc, lsit_cc); SAME_FIELD({R}_info, frp_forces_info, step_aff, step_aff);
SAME_FIELD({R}_info, frp_forces_info, step_cc, step_cc); SAME_FIELD({R}_info, frp_forces_info, solvetime, solvetime);
SAME_FIELD({R}_info, frp_forces_info, fevalstime, fevalstime);
static_assert(std::is_same<{R}_extfunc, frp_forces_extfunc>::value, "callback type");
static_assert(std::is_same<decltype(&{R}_solve), int (*)({R}_params *, {R}_output *, {R}_info *, FILE *, {R}_extfunc)>::value, "reference prototype");
static_assert(std::is_same<decltype(&frp_shadow_{m}_solve), int (*)(frp_forces_params *, frp_forces_output *, frp_forces_info *, FILE *, frp_forces_extfunc)>::value, "this repo's prototype");
// return codes the caller distinguishes (nmpc_solver.cpp:398-421)
static_assert(OPTIMAL_{R} == FRP_EXIT_OPTIMAL && MAXITREACHED_{R} == FRP_EXIT_MAXIT && FACTORIZATION_ERROR_{R} == FRP_EXIT_FACTORIZATION &&
              BADFUNCEVAL_{R} == FRP_EXIT_BADFUNCEVAL && NOPROGRESS_{R} == FRP_EXIT_NOPROGRESS && PARAM_VALUE_ERROR_{R} == FRP_EXIT_PARAM_VALUE, "exit codes");
// "no usable device" is reported with the reference's own "solver not valid on this machine" value; the runtime-fault code is
// outside the reference's set
static_assert(LICENSE_ERROR_{R} == FRP_EXIT_NO_DEVICE && FRP_EXIT_DEVICE_FAULT < LICENSE_ERROR_{R}, "machine-level codes");
int main() {{
    // the reference's prototype resolved by this repo's library, exactly what plan_manage's link step does
    int (*solve)({R}_params *, {R}_output *, {R}_info *, FILE *, {R}_extfunc) = &{R}_solve;
    return solve ? 0 : 1;
}}
""")
    exe = tmp_path / "hdr"
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I" + REF_INC.format(m=m), "-I" + os.path.join(ROOT, "include"), str(src),
           "-L" + os.path.dirname(solver.LIB_PATH), "-l:" + os.path.basename(solver.LIB_PATH), "-Wl,-rpath," + os.path.dirname(solver.LIB_PATH),
           "-Wl,--unresolved-symbols=ignore-in-shared-libs", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_abi_check_rejects_a_caller_built_against_another_header():
    """FRP_INFO_STRIDE went 8 -> 12 and frp_nmpc_options grew with only the version string bumped.  The header now
    carries FRP_NMPC_ABI_VERSION; frp_nmpc_abi_check compares version, struct sizes and info stride with the library's own,
    the Python loader and the C++ adapter call it before anything else."""
    import ctypes, re
    lib = solver.lib()
    hdr = open(os.path.join(ROOT, "include", "frp_nmpc.h")).read()
    ver = int(re.search(r"#define FRP_NMPC_ABI_VERSION (\d+)", hdr).group(1))
    stride = int(re.search(r"#define FRP_INFO_STRIDE (\d+)", hdr).group(1))
    assert lib.frp_nmpc_abi_version() == ver == solver.ABI_VERSION and stride == solver.INFO_STRIDE
    so, sb = ctypes.sizeof(solver.Options), ctypes.sizeof(solver.Batch)
    assert lib.frp_nmpc_abi_check(ver, so, sb, stride) == 0
    for bad in ((ver - 1, so, sb, stride), (ver, so - 8, sb, stride), (ver, so, sb + 8, stride), (ver, so, sb, 8)):
        assert lib.frp_nmpc_abi_check(*bad) == -1003  # FRP_ERR_ARG


def test_workspace_size_formula():
    """The solver keeps its per-iteration state in LDS and registers: the device workspace is the work queue (counter 256 B,
    per-CU counters 8 KB, one key and one order entry per problem = 12 B per problem), whatever the horizon and the face
    count -- plus, for the launches the four-per-CU variants take (N <= 20, at most 6 corridor rows), the packed Riccati
    blocks S_xx of the RESIDENT workgroups (20 x 48 doubles each, at most four per CU; 128 bytes of alignment slack) and, for the launches
    the three-per-CU variant of round 6 takes (20 < N <= 30, at most 16 rows), 30 x 48 doubles for at most three workgroups per CU: bounded by
    the chip, not by B."""
    lib = solver.lib()
    ws = lib.frp_nmpc_workspace_bytes
    queue = lambda B: 8 * (32 + 1024 + B + (B + 1) // 2)
    slot = 8 * 20 * 48  # S_xx of 20 stages: 48 doubles each
    cap = (ws(10 ** 6, 20, 6) - queue(10 ** 6) - 128) // slot
    assert cap % 4 == 0 and 4 <= cap <= 4 * 1024  # four workgroups per CU (256 CUs assumed without a device)
    slot30 = 8 * 30 * 48
    cap30 = (ws(10 ** 6, 30, 15) - queue(10 ** 6) - 128) // slot30
    assert cap30 % 3 == 0 and cap30 * 4 == cap * 3
    for B in (1, 7, 4096, 10 ** 6):
        assert ws(B, 64, 30) == ws(B, 20, 7) == ws(B, 31, 6) == ws(B, 30, 17) == queue(B)
        assert ws(B, 20, 6) == queue(B) + 128 + min(B, cap) * slot
        assert ws(B, 21, 6) == ws(B, 30, 16) == queue(B) + 128 + min(B, cap30) * slot30


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_device_means_loud_failure_not_cpu_fallback():
    w = workloads.config2(2)
    with pytest.raises(RuntimeError):
        solver.solve_batch_host(w)
    w0 = workloads.config0()
    p = solver.ForcesParams(); o = solver.ForcesOutput(); info = solver.ForcesInfo()
    p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel()
    flag = solver.lib().FORCESNLPsolver_normal_solve(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), None, None)
    assert flag == -100  # FRP_EXIT_NO_DEVICE = the reference's "solver not valid on this machine", not PARAM_VALUE
    assert np.all(np.array(o.x) == 0.0)


def _harness():
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_harness")
    src = exe + ".cpp"
    # (rebuilt whenever the source, the C-ABI header, the C++ adapter or the library is newer: the harness sizes its buffers from the
    # header's constants -- a binary built against an older FRP_INFO_STRIDE corrupts its heap)
    deps = [src, os.path.join(ROOT, "include", "frp_nmpc.h"), os.path.join(ROOT, "forces_resilient_planner_amd", "csrc", "frp_adapter.hpp"), solver.LIB_PATH]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps if os.path.exists(d)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe, "-L" + os.path.dirname(solver.LIB_PATH),
                               "-lfrp_nmpc_amd", "-Wl,-rpath," + os.path.dirname(solver.LIB_PATH),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _write_harness_input(path, w, weights, F):
    B, N = w["B"], w["N"]
    with open(path, "wb") as f:
        np.array([B, N, F, w["model"]], dtype=np.int32).tofile(f)
        np.asarray(weights, dtype=np.float64).tofile(f)
        for key in ("mpc_output", "f_ext", "ref_pos", "ref_yaw", "E", "poly_A", "poly_b"):
            np.ascontiguousarray(w[key], dtype=np.float64).tofile(f)
        np.ascontiguousarray(w["nfaces"], dtype=np.int32).tofile(f)


def test_cpp_adapter_packs_exactly_like_python_adapter(tmp_path):
    """G2 packing vectors: forces_normal.cpp:62-136 restated twice (numpy, C++) must agree bit for bit,
    including the robust tightening b - ||E a||, zero padding and the 30-row truncation."""
    w = workloads.config2(5)
    rng = np.random.default_rng(3)
    # make it harder: 33 faces on some stages (the reference silently drops rows >= 30), ragged counts
    B, N = 5, w["N"]
    F = 33
    A = rng.normal(size=(B, N, F, 3)); b = rng.uniform(1, 3, size=(B, N, F))
    nf = rng.integers(0, F + 1, size=(B, N)).astype(np.int32)
    w = dict(w, poly_A=A, poly_b=b, nfaces=nf)
    weights = (7.0, 1.0, 80.0, 12.0, 0.5)
    ad = adapter.ForcesAdapter(B, w["model"], N, 30)
    ad.set_paras(*weights)
    xinit, x0, params, nfo = ad.pack(w["mpc_output"], w["f_ext"], w["ref_pos"], w["ref_yaw"], w["E"], A, b, nf)
    inp = tmp_path / "in.bin"; out = tmp_path / "out.bin"
    _write_harness_input(inp, w, weights, F)
    subprocess.check_call([_harness(), "pack", str(inp), str(out)])
    raw = np.fromfile(out, dtype=np.uint8)
    nd = B * 9 + B * N * 17 + B * N * 130
    d = raw[:nd * 8].view(np.float64); ni = raw[nd * 8:].view(np.int32)
    assert np.array_equal(d[:B * 9], xinit.ravel())
    assert np.array_equal(d[B * 9:B * 9 + B * N * 17], x0.ravel())
    pc = d[B * 9 + B * N * 17:]
    assert np.array_equal(pc == 0, params.ravel() == 0)            # same padding pattern
    assert np.max(np.abs(pc - params.ravel())) <= 4e-16 * 4        # b - ||E a||: summation order only
    assert np.array_equal(ni, nfo.ravel())
    assert nfo.max() == 30  # truncated
    # padded rows are exactly zero, live rows carry the tightened offsets
    p = params.reshape(B, N, 130)
    for bb in range(B):
        for k in range(N):
            m = nfo[bb, k]
            assert np.all(p[bb, k, 10 + 3 * m:100] == 0) and np.all(p[bb, k, 100 + m:] == 0)
            if m:
                Ea = w["E"][bb, k] @ A[bb, k, 0]
                assert abs(p[bb, k, 100] - (b[bb, k, 0] - np.linalg.norm(Ea))) < 1e-15


def test_adapter_shift_warm_start_and_yaw_wrap():
    st = np.array([[0.1, 0.2, 1.0, 0, 0, 0, 0, 0, 0.3]])
    mpc = adapter.init_mpc_output(st, 20)
    assert mpc.shape == (1, 21, 17) and np.all(mpc[0, :, 3] == 7.3) and np.all(mpc[0, :, 7] == 7.3)
    mpc[0, :, 16] = np.linspace(-4, 4, 21)
    mpc[0, :, 0] = np.arange(21)
    ad = adapter.ForcesAdapter(1)
    A = np.zeros((1, 20, 1, 3)); b = np.zeros((1, 20, 1)); nf = np.zeros((1, 20), np.int32)
    xinit, x0, params, _ = ad.pack(mpc, np.zeros((1, 3)), np.zeros((1, 20, 3)), np.zeros((1, 20)), np.zeros((1, 20, 3, 3)), A, b, nf)
    assert np.array_equal(x0[0, :, 0], np.arange(1, 21))          # stage i <- previous plan stage i+1
    assert np.array_equal(xinit[0], mpc[0, 1, 8:17])                # NOT row 0 / odometry (forces_normal.cpp:62-72)
    out = adapter.update_forces_results(mpc.copy())
    assert np.all(np.abs(out[0, :20, 16]) <= np.pi + 1e-12)
    assert np.array_equal(out[0, 20], out[0, 19])


def test_workloads_are_seeded_and_well_formed():
    a = workloads.config2(16); b = workloads.config2(16)
    assert np.array_equal(a["params"], b["params"]) and np.array_equal(a["xinit"], b["xinit"])
    assert a["params"].shape == (16, 20, 130) and a["nfaces"].min() == 6 == a["nfaces"].max()
    c = workloads.config3(8)
    assert c["N"] == 30 and c["M"] == 15 and c["params"].shape == (8, 30, 70) and c["nfaces"].max() <= 15
    # the stage reference point is strictly inside its tightened polytope
    p = c["params"]
    for bb in range(8):
        for k in range(30):
            m = c["nfaces"][bb, k]
            A = p[bb, k, 10:10 + 3 * m].reshape(m, 3); bt = p[bb, k, 10 + 45:10 + 45 + m]
            assert np.all(A @ c["ref_pos"][bb, k] < bt)
    d = workloads.config1(4)
    assert d["nfaces"].max() == 0 and np.all(d["params"][:, :, 10:] == 0)


def test_lds_kernel_has_no_spill_behind_a_lane_divergent_loop(tmp_path):
    """Compiler-hazard guard.  The gfx950 backend may place a VGPR spill at the exit of a lane-divergent loop, where the
    EXEC mask is empty, so the spill saves nothing and the later reload returns whatever the scratch slot held (this
    crashed a solve of the round-1 kernel through a reloaded zero offset).  The solver kernel (frp_ipm_lds.hip) has one
    lane-divergent loop (the padding-row detection): it is a function of its own and must use no scratch."""
    import re
    import subprocess
    from forces_resilient_planner_amd import build
    src = os.path.join(build.CSRC, "frp_ipm_lds.hip")
    out = tmp_path / "lds.s"
    subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + build.PER_SOURCE_FLAGS["frp_ipm_lds.hip"] + [
                           "-I" + os.path.join(build.ROOT, "include"), src, "-o", str(out)], stderr=subprocess.DEVNULL)
    txt = out.read_text()
    scratch = {name: int(sz) for name, sz in re.findall(r"^(_ZN3frp\w+):.*?^; ScratchSize: (\d+)", txt, flags=re.M | re.S)}
    det = [n for n in scratch if "count_live_faces" in n]
    assert len(det) == 1 and scratch[det[0]] == 0, scratch
    # (every other loop of that file runs over the wave-uniform horizon length or a compile-time face count: the kernel
    # bodies, which do use scratch, contain no per-lane trip count)
    assert len([n for n in scratch if "nmpc_ipm_lds_kernel" in n]) >= 5  # (the main translation unit: the five FREG variants)
    _check_asm_lds_loads_land_before_edges(txt)


def _check_asm_lds_loads_land_before_edges(txt):
    """The vector sweeps issue their LDS loads through inline asm and wait for them with hand-counted s_waitcnt: the
    compiler believes a loaded register is valid at once.  Invariant that keeps this safe (frp_ipm_lds.hip, "the two
    vector sweeps"): between a ds_read of the sweeps and the s_waitcnt that retires it, no instruction reads the
    destination registers, and no label / branch is crossed (the register allocator may copy values on an edge).  Checked
    on the generated code: a linear scan per function with the set of registers whose load is in flight."""
    import re

    def regs(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()

    checked = 0
    for fn in ("sweep_forward", "sweep_backvec"):
        m = re.search(r"^(_ZN3frp2lr\d+%s\w*):\s" % fn, txt, flags=re.M)
        assert m, fn
        body = txt[m.end():]
        body = body[:body.index("s_setpc_b64")]
        inflight = set()
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line or "#ASMEND" in line:  # (the compiler's own LDS loads are waited for by the compiler)
                in_asm = "#ASMSTART" in line
                continue
            line = line.split(";")[0].strip()
            if not line or line.startswith("."):
                if line.startswith(".LBB"):
                    assert not inflight, (fn, "label reached with loads in flight", line, sorted(inflight))
                continue
            op, _, rest = line.partition(" ")
            toks = [t.strip() for t in rest.replace(" offset", ", offset").split(",")]
            if op == "s_waitcnt":
                m2 = re.search(r"lgkmcnt\((\d+)\)", rest)
                if m2 and int(m2.group(1)) <= 2:  # at most the two stores of the running step stay outstanding
                    inflight.clear()
                continue
            if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64"):
                assert not inflight, (fn, "branch with loads in flight", line, sorted(inflight))
                continue
            if op == "ds_read_b64" and in_asm:
                dst = regs(toks[0])
                src = set().union(*[regs(t) for t in toks[1:]])
                assert not (src & inflight), (fn, line)
                inflight |= dst
                checked += 1
                continue
            used = set().union(*[regs(t) for t in toks]) if toks else set()
            assert not (used & inflight), (fn, "register of a load in flight touched", line, sorted(used & inflight))
    assert checked > 100


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: include/frp_nmpc.h must compile as strict C99 on its own (no C++, no torch types)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "frp_nmpc.h"\n'
                   'int main(void) { frp_nmpc_batch b; frp_nmpc_pack p; frp_nmpc_tube t; frp_nmpc_corridor c; frp_nmpc_reference r;\n'
                   '  frp_forces_params fp; (void)b; (void)p; (void)t; (void)c; (void)r; (void)fp;\n'
                   '  return sizeof(frp_forces_params) == 23600 ? 0 : 1; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(src)])


def test_python_mirrors_of_the_c_structs_have_the_headers_layout(tmp_path):
    """solver.py drives the C-ABI through ctypes mirrors of the header's structs: their sizes, and the offsets of the fields that were
    appended last, must be the compiler's (a field added to the header only would shift every argument behind it silently)."""
    import subprocess
    pairs = [("frp_nmpc_options", solver.Options, "twist"), ("frp_nmpc_batch", solver.Batch, "order_hint"), ("frp_nmpc_astar", solver.Astar, "retry_vel"),
             ("frp_nmpc_pack", solver.Pack, None), ("frp_nmpc_tube", solver.Tube, None), ("frp_nmpc_corridor", solver.Corridor, None),
             ("frp_nmpc_reference", solver.Reference, None),
             ("frp_forces_params", solver.ForcesParams, "num_of_threads"), ("frp_forces_info", solver.ForcesInfo, None), ("frp_forces_output", solver.ForcesOutput, None)]
    src = tmp_path / "layout.c"
    body = "".join(f'  printf("%zu %zu\\n", sizeof({c}), {("offsetof(" + c + ", " + f + ")") if f else "(size_t)0"});\n' for c, _, f in pairs)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "frp_nmpc.h"\nint main(void) {\n' + body + '  return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    for i, (cname, py, field) in enumerate(pairs):
        size, off = int(out[2 * i]), int(out[2 * i + 1])
        assert ctypes.sizeof(py) == size, (cname, ctypes.sizeof(py), size)
        if field:
            assert getattr(py, field).offset == off, (cname, field, getattr(py, field).offset, off)


def test_widened_entry_points_reject_bad_arguments_before_touching_a_device():
    """Argument checks of the f-1 .. f-4 entry points run on the host, so they are testable without a GPU: NULL
    buffers, horizons beyond 64 stages, polytope capacities outside [6, 64], oversized clouds / grids, non-positive
    physical constants all return FRP_ERR_ARG (-1003)."""
    l = solver.lib()
    ERR = -1003
    one = ctypes.c_void_p(8)  # any non-NULL address: the checks must fail before it would be dereferenced on the device
    tb = solver.Tube(4, 20, None, 0.74, 0.33, 0.27, 0.0425, (ctypes.c_double * 3)(0.5, 0.5, 0.5), 0.06, 0.05, one)
    assert l.frp_nmpc_tube_batch(ctypes.byref(tb), None) == ERR                 # no plan
    tb.mpc_output = 8; tb.N = 65
    assert l.frp_nmpc_tube_batch(ctypes.byref(tb), None) == ERR                 # horizon > 64
    tb.N = 20; tb.mass = 0.0
    assert l.frp_nmpc_tube_batch(ctypes.byref(tb), None) == ERR                 # mass
    tb.mass = 0.74; tb.noise = (ctypes.c_double * 3)(0.5, 0.0, 0.5)
    assert l.frp_nmpc_tube_batch(ctypes.byref(tb), None) == ERR                 # a zero noise bound makes tr(X) = 0
    cr = solver.Corridor()
    cr.B, cr.N, cr.F, cr.P = 4, 20, 64, 100
    for f in ("cloud", "ref_pos", "ref_yaw", "ellipsoid", "poly_A", "poly_b", "poly_nfaces", "poly_index"):
        setattr(cr, f, 8)
    cr.bbox = (ctypes.c_double * 3)(2, 2, 1); cr.seed_len = 0.1; cr.inflation = 1.1
    for field, bad in (("F", 5), ("F", 65), ("N", 0), ("N", 65), ("P", 65537), ("seed_len", 0.0), ("poly_index", None), ("cloud", None)):
        keep = getattr(cr, field)
        setattr(cr, field, bad)
        assert l.frp_nmpc_corridor_batch(ctypes.byref(cr), None) == ERR, field
        setattr(cr, field, keep)
    cr.grid_start = 8                                                            # a grid without its other arrays / cell size
    assert l.frp_nmpc_corridor_batch(ctypes.byref(cr), None) == ERR
    rf = solver.Reference(4, 20, 50, None, 0, None, 8, 8, 0.05, 3.1415926, 8, 8, None)
    assert l.frp_nmpc_reference_batch(ctypes.byref(rf), None) == ERR           # no path
    rf.kino_path = 8; rf.Ts = 0.0
    assert l.frp_nmpc_reference_batch(ctypes.byref(rf), None) == ERR
    assert l.frp_nmpc_coldstart_batch(4, 20, None, None, 7.3, None, None) == ERR
    d3, i3 = (ctypes.c_double * 3)(0, 0, 0), (ctypes.c_int * 3)(1 << 11, 1 << 11, 2)   # 2^23 cells > FRP_CORRIDOR_MAX_CELLS
    assert l.frp_nmpc_cloud_grid_build(one, 10, d3, 0.5, i3, one, one, one, one, None) == ERR
    assert l.frp_nmpc_cloud_grid_build(one, 10, d3, 0.0, (ctypes.c_int * 3)(4, 4, 4), one, one, one, one, None) == ERR


def test_product_objects_hold_no_truncated_scalar_immediates(monkeypatch):
    """Guard against the miscompile behind round 2's "wrong objective" build: with -disable-machine-licm -disable-machine-cse alone
    this compiler emits s_mov_b64 with a 64-bit FP immediate, gfx9 encodes the low dword only, and 12.0 / 10.0 / 20.0 become 0.0
    (build.py has the whole story).  Every object of the product library is disassembled and checked; and the code-generation
    switches are dropped when hipcc is not the release they were checked on."""
    from forces_resilient_planner_amd import build
    build.build_native(force=False, verbose=False)
    checked = 0
    for n in build.SOURCES:
        obj = os.path.join(build.OBJDIR, n + ".o")
        if os.path.exists(obj):
            checked += build.check_device_code(obj)
    assert checked > 500
    import subprocess
    real = subprocess.run
    monkeypatch.setattr(build.subprocess, "run", lambda *a, **k: type("R", (), {"stdout": "HIP version: 9.9\nAMD clang version 99 (roc-9.9.0)", "returncode": 0})())
    assert not build.codegen_flags_trusted()
    monkeypatch.setattr(build.subprocess, "run", real)
    assert build.codegen_flags_trusted()


def _build_static_dropin_program(tmp_path):
    """A C++ program compiled against the REFERENCE's generated headers and linked, by plain g++, against static archives with
    the reference's file names (plan_manage/CMakeLists.txt:64-65, 82-83) + the HIP runtime."""
    import subprocess
    from forces_resilient_planner_amd import build
    build.build_native(force=False, verbose=False)
    build.build_dropin_archives()
    inc = [REF_INC.format(m=m) for m in ("normal", "final")]
    if not all(os.path.isdir(d) for d in inc):
        pytest.skip("reference headers absent (GPU box)")
    src = os.path.join(ROOT, "tests", "cpp", "static_dropin_stub.cpp")
    exe = tmp_path / "planner_stub"
    rocm = "/opt/rocm/lib"
    r = subprocess.run(["g++", "-O1", "-I" + inc[0], "-I" + inc[1], str(src), "-L" + build.DROPIN_DIR, "-l:libFORCESNLPsolver_normal.a",
                        "-l:libFORCESNLPsolver_final.a", "-L" + rocm, "-lamdhip64", "-lpthread", "-Wl,-rpath," + rocm, "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour of the statically linked stub")
def test_static_archives_with_the_references_file_names_link_and_fail_loudly_without_a_device(tmp_path):
    import subprocess
    exe = _build_static_dropin_program(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split()[:2] == ["-100", "-100"], r.stdout + r.stderr
