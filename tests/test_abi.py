"""CPU: the C-ABI library loads, exports every symbol include/neptune_backend.h declares, and
its records have the layout the ctypes mirror assumes.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from neptune_amd import _lib, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "neptune_backend.h")).read()
    declared = set(re.findall(r"^(?:int|void|double|int64_t|const char\*|nep_backend_t\*|nep_batch_t\*|nep_comm_t\*)\s+(nep_[a-z_0-9]+)\(", hdr, re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    # test hooks, measurement aids and A/B knobs live in a header of their own: the drop-in surface has none of them
    dbg = open(os.path.join(ROOT, "include", "neptune_backend_debug.h")).read()
    declared_dbg = set(re.findall(r"^(?:int|void|double|int64_t)\s+(nep_[a-z_0-9]+)\(", dbg, re.M))
    assert declared_dbg == set(_lib.DEBUG_EXPORTS) and not (declared & declared_dbg)
    assert not [n for n in declared if "debug" in n]
    for name in declared_dbg:
        assert hasattr(L, name), name


def test_struct_layouts(L):
    for k, t in enumerate([abi.nep_pwp, abi.nep_traj_rec, abi.nep_backend_cfg, abi.nep_stats, abi.nep_batch_cfg,
                           abi.nep_guess, abi.nep_solution, abi.nep_ent_view]):
        assert C.sizeof(t) == L.nep_abi_sizeof(k), t.__name__
    assert abi.TRAJ_REC_DTYPE.itemsize == C.sizeof(abi.nep_traj_rec)
    assert abi.GUESS_DTYPE.itemsize == C.sizeof(abi.nep_guess)
    assert abi.SOLUTION_DTYPE.itemsize == C.sizeof(abi.nep_solution)
    for k, t in ((11, abi.nep_fe_cfg), (12, abi.nep_fe_start), (13, abi.nep_fe_result)):      # include/neptune_frontend.h
        assert C.sizeof(t) == L.nep_abi_sizeof(k), t.__name__
    hdr = open(os.path.join(ROOT, "include", "neptune_frontend.h")).read()
    declared = set(re.findall(r"^int\s+(nep_[a-z_0-9]+)\(", hdr, re.M))
    assert declared == set(_lib.FE_EXPORTS) and all(hasattr(L, n) for n in declared)


def test_fails_loudly_without_a_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pb = np.zeros((1, 2))
    cfg = abi.nep_backend_cfg(8, 3, 1, 1, 0.5, 1000.0, 0.5, 1, 0, abi.dptr(pb))
    assert not L.nep_backend_create(C.byref(cfg))
    assert b"no HIP device" in L.nep_last_error() or b"hip" in L.nep_last_error().lower()
    nd = np.zeros(3); ok = np.zeros(1, dtype=np.int32); off = np.array([0, 1], dtype=np.int32); xy = np.zeros((1, 2))
    rc = L.nep_separator_batch(1, abi.iptr(off), abi.dptr(xy), abi.iptr(off), abi.dptr(xy), abi.dptr(nd), abi.iptr(ok))
    assert rc < 0


def test_the_library_reads_no_environment_variable_that_changes_what_it_computes(L):
    """Round-5 review: 17 getenv names in the product (NEP_QP_KERNEL, NEP_SEP_SKIP, NEP_SEP_NO_REDO, NEP_CORR_*, ...), several of which
    changed which solves succeed.  They are explicit setters of include/neptune_backend_debug.h now.  Here: the sources call getenv
    once (NEP_QP_PROFILE: the per-phase cycle counters of `make PROFILE=1`, no result depends on it), the built library holds no other
    NEP_* name to look up, and the option setters exist and refuse unknown names.  (The GPU suite runs a replan under a hostile
    environment and compares bytes: tests/test_gpu_robustness.py.)"""
    src = os.path.join(ROOT, "neptune_amd", "csrc")
    sites = []
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".cpp", ".h")):
            for n, line in enumerate(open(os.path.join(src, f)), 1):
                code = line.split("//")[0]
                sites += [(f, n, m) for m in re.findall(r'getenv\("([A-Z_0-9]+)"\)', code)]
                assert "getenv(" not in re.sub(r'getenv\("[A-Z_0-9]+"\)', "", code), (f, n)      # no computed names
    assert {m for _, _, m in sites} == {"NEP_QP_PROFILE"} and len(sites) <= 3, sites
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(re.findall(rb"NEP_[A-Z][A-Z_0-9]{3,}", blob)) - {b"NEP_QP_PROFILE"}
    old = {b"NEP_QP_KERNEL", b"NEP_QP_AUTOCULL", b"NEP_SEP_SKIP", b"NEP_SEP_NO_REDO", b"NEP_QP_LPT", b"NEP_FE_LPT", b"NEP_QP_KEY_DECAY", b"NEP_FE_KEY_DECAY",
           b"NEP_SEP_UNPACKED", b"NEP_SEP_PACK", b"NEP_CORR_FROM", b"NEP_CORR_MAX", b"NEP_HULL_KERNEL", b"NEP_FE_THREE", b"NEP_FE_XCD", b"NEP_POLISH_GRID"}
    assert not (names & old), names & old
    assert L.nep_debug_set_global_option(b"fe_xcd", 1) == 0 and L.nep_debug_set_global_option(b"polish_grid", 256) == 0
    assert L.nep_debug_set_global_option(b"no_such_option", 1) < 0 and b"unknown" in L.nep_last_error()
    assert L.nep_batch_debug_set_option(None, b"qp_kernel", 1) < 0


def test_oracle_is_not_reachable_from_the_product():
    """The product package must not import or link the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "neptune_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith(("//", "#", "*", "/*")))
                assert "from oracle" not in code and "import oracle" not in code and "neptune_oracle" not in code, f


def test_cpp_host_class_compiles_and_links(L):
    """include/neptune_poly_solver.hpp (the PolySolverGurobi-shaped C++ class) builds with plain g++."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "replan_example")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "replan_example.cpp"),
                           "-L" + os.path.join(ROOT, "neptune_amd"), "-lneptune_backend",
                           "-Wl,-rpath,$ORIGIN/../../neptune_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    assert os.path.exists(exe)


def test_reduced_qp_tables_are_consistent(tmp_path):
    """nep_tables.h on the host: Zp is the left inverse of Th (start point), theta(z) is C2-continuous, starts at the
    initial state and (mode 0) ends at rest, and projecting a feasible theta returns its z — for every K and mode."""
    import subprocess
    exe = str(tmp_path / "tables_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "tables_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout


def test_headers_are_plain_c99():
    """The boundary is a C ABI: every header under include/ (except the C++ class) must compile as strict C99."""
    import subprocess
    for h in ("neptune_backend.h", "neptune_plan.h", "neptune_entangle.h", "neptune_frontend.h"):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                            "-x", "c", "-"], input='#include "%s"\n' % h, text=True, capture_output=True)
        assert r.returncode == 0, (h, r.stderr)


def _find_eigen():
    for d in ("/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/eigen3/include/eigen3", os.environ.get("EIGEN3_INCLUDE_DIR", "")):
        if d and os.path.exists(os.path.join(d, "Eigen", "Dense")):
            return d
    return None


def test_exact_signature_shim_compiles_against_the_reference_headers(tmp_path):
    """include/neptune_poly_solver.hpp's `class PolySolverGurobi` (the signatures of solver_gurobi_poly.hpp:27-52) built
    with -DNEPTUNE_AMD_REFERENCE_SHIM against the reference's own mader_types.hpp / entangle_utils.hpp.  Needs Eigen and
    the reference tree: SKIPPED (not passed) where either is absent — this image has no Eigen, so the claim "neptune.cpp
    compiles unchanged against the shim" stays a claim until a maintainer runs this in the reference's build environment."""
    import subprocess
    eigen = _find_eigen()
    ref = os.environ.get("NEPTUNE_REFERENCE_DIR", "/root/reference")
    inc = os.path.join(ref, "neptune", "include")
    if eigen is None:
        pytest.skip("Eigen not installed: the exact-signature shim cannot be compiled here")
    if not os.path.exists(os.path.join(inc, "mader_types.hpp")):
        pytest.skip("reference headers not available")
    src = tmp_path / "shim.cpp"
    src.write_text('#define NEPTUNE_AMD_REFERENCE_SHIM 1\n#include "neptune_poly_solver.hpp"\n'
                   'int main() { std::vector<Eigen::Vector2d> pb(1, Eigen::Vector2d(0, 0)); PolySolverGurobi s(8, 3, 1, 0.5, pb, 1000.0, 0.5, true); '
                   'double o = 0; (void)&PolySolverGurobi::optimize; (void)s; (void)o; return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-I" + eigen, "-I" + inc, str(src)])


def _build_shim_check(out):
    import subprocess
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "cpp", "ref_types_min"), os.path.join(ROOT, "tests", "cpp", "shim_signature_check.cpp"),
                           "-L" + os.path.join(ROOT, "neptune_amd"), "-lneptune_backend", "-Wl,-rpath," + os.path.join(ROOT, "neptune_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(out)])
    return subprocess.run([str(out)], capture_output=True, text=True, timeout=300)


def test_exact_signature_shim_compiles_against_minimal_type_declarations(L, tmp_path):
    """`class PolySolverGurobi` of include/neptune_poly_solver.hpp — the exact signatures of solver_gurobi_poly.hpp:28-49 —
    compiled (-Wall -Wextra -Werror), linked against the library and run, against the stand-in declarations of
    tests/cpp/ref_types_min/ (this repo's own few dozen lines naming what the class touches of Eigen, mader_types.hpp and
    entangle_utils.hpp; NOT the reference's headers, NOT Eigen: syntax + signatures + conversions only).  The program
    static_asserts every method's type and makes the calls in Neptune's order (neptune.cpp:102-107, 663, 1514-1527); without a
    GPU the constructor throws (no CPU path) — the `-m gpu` twin in tests/test_gpu_per_agent_api.py checks the solve."""
    if _find_eigen() is not None:
        pytest.skip("Eigen present: the stand-in <Eigen/Dense> would shadow it; the reference-header test above is the check")
    r = _build_shim_check(tmp_path / "shim_signature_check")
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0 and "optimize -> 1" in r.stdout, (r.returncode, r.stdout, r.stderr)
    else:
        assert r.returncode == 3 and "no HIP device" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_shim_macro_without_eigen_is_an_error(tmp_path):
    """setting NEPTUNE_AMD_REFERENCE_SHIM without Eigen must fail the build, not silently drop the class"""
    import subprocess
    if _find_eigen() is not None:
        pytest.skip("Eigen present")
    src = tmp_path / "shim.cpp"
    src.write_text('#define NEPTUNE_AMD_REFERENCE_SHIM 1\n#include "neptune_poly_solver.hpp"\nint main() { return 0; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode != 0 and "needs Eigen" in r.stderr


def test_small_divisor_magic_numbers_are_exact():
    """qp_common.h's cDivMagic (x / d as a multiply and a shift in the QP kernel's set-up) and nep_device.h's hull_pb_magic
    formula: exact over the ranges the kernels use them on."""
    import re
    src = open(os.path.join(ROOT, "neptune_amd", "csrc", "qp_common.h")).read()
    m = re.search(r"cDivMagic\[65\]\s*=\s*\{([^}]*)\}", src)
    tab = [int(v) for v in m.group(1).split(",")]
    assert len(tab) == 65
    for d in range(1, 65):
        for x in range(1024):
            assert (x * tab[d]) >> 16 == x // d, (x, d)
    for d in list(range(1, 300)) + [512, 1000, 4096, 65535]:
        magic = (1 << 32) // d + 1
        for j in list(range(0, 3000, 7)) + [65535]:
            assert (j * magic) >> 32 == j // d, (j, d)


def test_bench_line_stays_short():
    """bench.py's last stdout line is what the driver parses out of an 8 KB tail: the compact form of a full record (here: round 4's
    27 KB line, which the driver could not read) must stay below 4 KB and keep the contract's fields"""
    import json
    from bench_legs import compact
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_v53_bench_line.json")))
    line = json.dumps(compact.compact_line(full, "bench_detail.json"))
    assert len(line) < 4096
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert out[k] == full[k] or abs(out[k] - full[k]) <= 1e-5 * abs(full[k]), k
    assert out["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4) and out["cpu_baseline"]["kind"] == "port"
    assert out["config"]["workload"] == full["config"]["workload"]
    # a record with 64 ranks' entries still fits (the optional parts are shed, largest first)
    full["per_rank"] = [{"rank": r, "kernel_ms": {"hull": 0.1, "separator": 0.4, "qp": 1.0, "exchange_wait": 0.01}, "step_ms_p50": 1.6, "wall_s": 0.03} for r in range(64)]
    assert len(json.dumps(compact.compact_line(full))) <= compact.LIMIT
