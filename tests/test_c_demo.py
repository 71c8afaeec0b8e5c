"""The C driver examples/aprilsam_demo_amd.c (counterpart of the reference's examples/aprilsam_demo.c, same flags)
compiled with gcc against include/aprilsam_amd.h and linked to libaprilsam_amd.so."""
import os
import re
import subprocess

import numpy as np
import pytest

from aprilsam_amd import datasets
from tests.conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The goldens were recorded on the reference's DETERMINISTIC schedule (wall-clock rule aprilsam.c:557-559 off); the library
# follows the reference by default (rule on), so the programs below run with the rule switched off through the environment.
DET_ENV = dict(os.environ, APRILSAM_AMD_DETERMINISTIC="1")


def _build(tmp_path, name="aprilsam_demo_amd"):
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "aprilsam_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
                           "-L" + libdir, "-laprilsam_amd", "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    return exe


def test_c_driver_compiles_as_plain_c_and_parses_its_flags(built, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage:" in r.stderr
    txt = str(tmp_path / "m.txt")
    datasets.write_vertex_edge_text(txt, *datasets.m3500_arrays())
    st, fa, fb, z, W = datasets.parse_vertex_edge_text(txt)             # lossless round trip of the loader format
    a = datasets.m3500_arrays()
    assert np.array_equal(st, a[0]) and np.array_equal(z, a[3]) and np.array_equal(W, a[4]) and np.array_equal(fa, a[1])
    # loader + writer of the C program, no solver call (--max_poses 0): text in, the reference's own data/M3500.graph out
    import hashlib
    from tests.test_graph_io import REF_M3500_GRAPH_SHA256
    out = str(tmp_path / "loaded.graph")
    r = subprocess.run([exe, "--datapath", txt, "--savepath", out, "--max_poses", "0"], capture_output=True, text=True)
    assert r.returncode == 0 and "3500 nodes,  factors: 5453" in r.stdout, r.stderr
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == REF_M3500_GRAPH_SHA256
    r = subprocess.run([exe, "--graphpath", out, "--max_poses", "0"], capture_output=True, text=True)     # and back in
    assert r.returncode == 0 and "3500 nodes,  factors: 5453" in r.stdout
    r = subprocess.run([exe, "--graphpath", str(tmp_path / "nope.graph")], capture_output=True, text=True)
    assert r.returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["inc", "batch", "inc_from_graph_file"])
def test_c_driver_reproduces_the_reference_demo(built, tmp_path, mode):
    exe = _build(tmp_path)
    txt = str(tmp_path / "m.txt")
    datasets.write_vertex_edge_text(txt, *datasets.m3500_arrays())
    n = 120 if mode == "batch" else 300
    src = ["--datapath", txt]
    if mode == "inc_from_graph_file":           # the reference demo's default input is the .graph file (examples/aprilsam_demo.c:249,262)
        gpath = str(tmp_path / "m.graph")
        assert subprocess.run([exe, "--datapath", txt, "--savepath", gpath, "--max_poses", "0"], capture_output=True).returncode == 0
        src = ["--graphpath", gpath]
    args = [exe] + src + ["--max_poses", str(n), "--nthreshold", "100", "--delta_xy", "0.1", "--delta_theta", "0.1"]
    if mode == "batch":
        args.append("--batch_update_only")
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=DET_ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    chi2 = np.array([float(x) for x in re.findall(r"Chi squared error: ([-0-9.eE+]+)", r.stdout)])
    assert len(chi2) == n
    if mode != "batch":
        G = golden("m3500_inc_demo.npz")["chi2"][:n]
        assert np.max(np.abs(chi2 - G) / np.maximum(G, 1e-6)) < 1e-5      # the driver prints %f: 6 decimals
    else:
        assert chi2[0] == 0.0 and np.all(np.isfinite(chi2)) and chi2[-1] < 50.0


def test_tutorial_driver_compiles_and_parses_its_flags(built, tmp_path):
    exe = _build(tmp_path, "aprilsam_tutorial_amd")
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["inc", "batch"])
def test_tutorial_driver_matches_reference_golden(built, tmp_path, mode):
    """examples/aprilsam_tutorial.c:80-266 scenario; the driver prints chi^2 with %f and states with %.2f (:67-76)"""
    exe = _build(tmp_path, "aprilsam_tutorial_amd")
    r = subprocess.run([exe] + (["--batch_update_only"] if mode == "batch" else []), capture_output=True, text=True, timeout=300, env=DET_ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    G = golden("tutorial_batch.npz" if mode == "batch" else "tutorial_inc.npz")
    chi2 = np.array([float(x) for x in re.findall(r"Chi squared error: ([-0-9.eE+]+)", r.stdout)])
    assert len(chi2) == 6 and np.max(np.abs(chi2 - G["chi2"])) < 2e-6
    blocks = r.stdout.split("==================== Step:")[1:]
    for k, blk in enumerate(blocks):
        st = np.array([[float(v) for v in m] for m in re.findall(r"node_\d+ = \{([-0-9.]+), ([-0-9.]+), ([-0-9.]+)\}", blk)])
        assert st.shape == (k + 1, 3) and np.max(np.abs(st - G[f"states_{k}"])) < 0.0051


# ---- INTEGRATION.md Option A, for real: the reference's OWN example programs on the GPU library --------------------
# oracle/Makefile compiles /root/reference/examples/aprilsam_demo.c and aprilsam_tutorial.c where they lie, links them
# with the reference's own objects minus aprilsam.o (the solver) and with libaprilsam_amd.so instead (oracle/_ref/, test
# infrastructure, built in the container, travels to the GPU box).  Unresolved symbols of those programs that our
# library satisfies: APRILSAM_VERSION, april_graph_cholesky_param_init / _destory, april_graph_cholesky, _inc.
REF_DEMO = os.path.join(ROOT, "oracle", "_ref", "ref_demo_on_amd")
REF_TUTORIAL = os.path.join(ROOT, "oracle", "_ref", "ref_tutorial_on_amd")


@pytest.mark.gpu
def test_reference_demo_program_linked_against_our_library_reproduces_its_own_trace(built, tmp_path):
    """all 3500 poses of the M3500 demo, incremental mode, through the reference's unmodified main(): the printed
    chi^2 after every step equals the trace the all-reference build printed (golden, deterministic schedule) -- which
    also means the 49 batch fall-backs happened at the same steps"""
    if not os.path.exists(REF_DEMO):
        pytest.skip("oracle/_ref/ref_demo_on_amd not built (needs /root/reference at build time)")
    txt = str(tmp_path / "m.txt")
    datasets.write_vertex_edge_text(txt, *datasets.m3500_arrays())
    r = subprocess.run([REF_DEMO, "--datapath", txt, "--nthreshold", "100", "--delta_xy", "0.1", "--delta_theta", "0.1"],
                       capture_output=True, text=True, timeout=600, env=DET_ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "aprilsam_amd" in r.stdout.splitlines()[1]                    # our banner: the solver really is this library
    chi2 = np.array([float(x) for x in re.findall(r"Chi squared error: ([-0-9.eE+]+)", r.stdout)])
    G = golden("m3500_inc_demo.npz")["chi2"]
    assert len(chi2) == len(G) == 3500
    assert np.max(np.abs(chi2 - G) / np.maximum(G, 1e-6)) < 1e-5          # printed with %f: 6 decimals
    assert abs(chi2[-1] - 68.965608) < 2e-6                               # SURVEY.md section 6, deterministic schedule


@pytest.mark.gpu
def test_reference_tutorial_program_linked_against_our_library(built):
    if not os.path.exists(REF_TUTORIAL):
        pytest.skip("oracle/_ref/ref_tutorial_on_amd not built (needs /root/reference at build time)")
    r = subprocess.run([REF_TUTORIAL], capture_output=True, text=True, timeout=300, env=DET_ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    chi2 = np.array([float(x) for x in re.findall(r"Chi squared error: ([-0-9.eE+]+)", r.stdout)])
    G = golden("tutorial_inc.npz")
    assert len(chi2) == 6 and np.max(np.abs(chi2 - G["chi2"])) < 2e-6
    assert abs(chi2[-1] - 7.805041) < 2e-6
